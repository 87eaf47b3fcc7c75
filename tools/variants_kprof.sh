#!/bin/bash
# per-kernel average times (rocprofv3 --kernel-trace --stats) of bench.py on every measurement build under build/variants
# usage: tools/variants_kprof.sh [workload ...]   (default: C2 C3raw)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${@:-C2 C3raw}
for so in $R/build/variants/libdistance_hip_*.so; do
  name=$(basename $so .so); name=${name#libdistance_hip_}
  for w in $W; do
    out=$R/gpurun_out/kprof_${name}_$w
    rm -rf $out
    DST_LIB_PATH=$so rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out.log 2>&1 || exit 1
    echo "== $name $w $(tail -1 $out.log | cut -c1-120)"
  done
done
