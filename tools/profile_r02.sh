#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats of bench.py (default
# workload, main leg only) and separate PMC passes (HBM traffic).  Writes under gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r02}
shift || true
EXTRA="$*"
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -- $BENCH > "$OUT/pmc_${C}_bench.json" 2> "$OUT/pmc_$C.err"
  echo "pmc $C rc=$?"
done
find "$OUT" -name "*.csv" | head -20
du -sh "$OUT"
