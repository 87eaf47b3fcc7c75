#!/bin/bash
# run tools/cbench.py on every measurement build under build/variants (one GPU process after the other)
cd "$(dirname "$0")/.."
for so in build/variants/libdistance_hip_*.so; do
  name=$(basename $so .so); name=${name#libdistance_hip_}
  echo "== $name"
  DST_LIB_PATH=$PWD/$so timeout -k 10 120 python tools/cbench.py --measures ${MEASURES:-n_high,raw} --paths consensus --reps ${REPS:-5} | grep -v "^#" || exit 1
done
