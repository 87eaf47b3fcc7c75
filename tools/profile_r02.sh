#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats of bench.py (default workload:
# main leg + dense leg + tn93 legs) and separate PMC passes (HBM traffic) of the main leg.  Writes under
# gpurun_out/prof_<tag>/; tools/prof_summary_r02.py condenses it into profiles/<tag>/.
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -- $BENCH > "$OUT/pmc_${C}_bench.json" 2> "$OUT/pmc_$C.err"
  echo "pmc $C rc=$?"
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_SQ" -- $BENCH --no-extra > "$OUT/pmc_SQ_bench.json" 2> "$OUT/pmc_SQ.err"
echo "pmc SQ rc=$?"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_TCC" -- $BENCH --no-extra > "$OUT/pmc_TCC_bench.json" 2> "$OUT/pmc_TCC.err"
echo "pmc TCC rc=$?"
du -sh "$OUT"
