#!/bin/bash
# tools/calibrate.py on every measurement build under build/variants
cd "$(dirname "$0")/.."
for so in build/variants/libdistance_hip_*.so; do
  name=$(basename $so .so); name=${name#libdistance_hip_}
  echo "== $name"
  DST_LIB_PATH=$PWD/$so timeout -k 10 300 python tools/calibrate.py ${CALIB_ARGS:---measures raw,tn93 --rates 0.001,0.01,0.03} | grep -v "^#" || exit 1
done
