#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats of bench.py and
# separate PMC passes (HBM traffic, SQ issue mix).  Writes under gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -- $BENCH --workload C2 > "$OUT/pmc_${C}_bench.json" 2> "$OUT/pmc_$C.err"
  echo "pmc $C rc=$?"
  # the default workload too: its per-launch HBM traffic is what bench.py reports as roofline.traffic
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmcdef_$C" -- $BENCH > "$OUT/pmcdef_${C}_bench.json" 2> "$OUT/pmcdef_$C.err"
  echo "pmcdef $C rc=$?"
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_SQ" -- $BENCH --workload C2 > "$OUT/pmc_SQ_bench.json" 2> "$OUT/pmc_SQ.err"
echo "pmc SQ rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_SQ2" -- $BENCH --workload C2 > "$OUT/pmc_SQ2_bench.json" 2> "$OUT/pmc_SQ2.err"
echo "pmc SQ2 rc=$?"
find "$OUT" -name "*.csv" | head -40
du -sh "$OUT"
