#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root: rocprofv3 of bench.py, one clean pass set per dominant kernel —
#   prof_r03       default workload (50,000 x 30,000 raw), --no-extra: the main leg's kernels (+ the verify pass' dense slabs)
#   prof_r03_tn93  --workload C3 (tn93), --no-extra
#   prof_r03_full  the default run with every leg (dense, tn93, clades, nruns): kernel-trace stats only
# each with --kernel-trace --stats and, in passes of their own, --pmc FETCH_SIZE / WRITE_SIZE (HBM traffic) and SQ counters.
# tools/prof_summary_r02.py condenses a directory into profiles/r03/.
set -u
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
run_set() {   # name, bench arguments
  local OUT=$REPO/gpurun_out/$1; shift
  rm -rf "$OUT"; mkdir -p "$OUT"
  local BENCH="python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace_bench.json" 2> "$OUT/trace.err"; echo "$OUT trace rc=$?"
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -- $BENCH > "$OUT/pmc_${C}_bench.json" 2> "$OUT/pmc_$C.err"; echo "pmc $C rc=$?"
  done
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_SQ" -- $BENCH > "$OUT/pmc_SQ_bench.json" 2> "$OUT/pmc_SQ.err"; echo "pmc SQ rc=$?"
}
run_set prof_r03 --no-extra
run_set prof_r03_tn93 --workload C3 --no-extra
OUT=$REPO/gpurun_out/prof_r03_full; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/trace_bench.json" 2> "$OUT/trace.err"; echo "full trace rc=$?"
du -sh $REPO/gpurun_out/prof_r03*
