#!/bin/bash
# rocprofv3 kernel stats of tools/cbench.py for every measurement build under build/variants: one kernel's average per build
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for so in $R/build/variants/libdistance_hip_*.so; do
  name=$(basename $so .so); name=${name#libdistance_hip_}
  rm -rf $R/gpurun_out/vprof_$name
  DST_LIB_PATH=$so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vprof_$name -- python3 $R/tools/cbench.py --measures ${MEASURES:-n_high} --paths consensus --reps 4 > /dev/null 2>&1 || exit 1
  echo "== $name"
  python3 - "$R/gpurun_out/vprof_$name" "${KERNEL:-site_fill}" <<'PY'
import csv,glob,sys
for p in glob.glob(sys.argv[1]+'/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(p)):
        if sys.argv[2] in r['Name']:
            print(f"{r['Name'][:60]:60s} {r['Calls']:>4s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
