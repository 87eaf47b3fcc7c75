#!/bin/bash
# per-kernel averages + kernel trace (step timeline) of bench.py at C2 and C3raw, in-tree library: gpurun_out/sf_<workload>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in C2 C3raw; do
  rm -rf $R/gpurun_out/sf_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sf_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $R/gpurun_out/sf_$w.log 2>&1 || exit 1
  tail -1 $R/gpurun_out/sf_$w.log | cut -c1-160
done
