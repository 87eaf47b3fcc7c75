#!/bin/bash
# Measurement builds of the consensus pair kernel: tools/variants.sh name "-DMACRO ..." [name "-D..."]...
# -> build/variants/libdistance_hip_<name>.so (select with DST_LIB_PATH; tools/variants_run.sh / variants_prof.sh /
# variants_calib.sh run cbench / rocprofv3 / calibrate on each).  Only dst_consensus.hip is rebuilt.
# Knobs (all compiled out of the product build; what each one answered is in DESIGN.md 3b):
#   (the work-skipping knobs of r02 — NO_EVENTS / NO_STORE / NO_APPLY — are gone from the kernel: r03)
#   -DDST_DBG_PLAIN_STORES  default-policy stores instead of nontemporal ones
#   -DDST_DBG_OLDMAP        panel-relative column mapping for every family (no address-aligned quarters)
#   -DDST_DBG_EVWAVES=k     k event waves + 8-k output waves for every launch
#   -DDST_DBG_RB=k          k rows per batch
set -e
cd "$(dirname "$0")/../distance_amd/csrc"
mkdir -p ../../build/variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function"
SRC=${VARIANT_SRC:-dst_consensus}   # which translation unit gets the macros (dst_consensus or dst_kernels)
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs -x hip -c $SRC.hip -o ../../build/variants/${SRC}_$name.o
  objs=""
  for o in dst_kernels dst_consensus dst_text dst_api dst_stream dst_host dst_gather dst_shared; do
    if [ $o = $SRC ]; then objs="$objs ../../build/variants/${SRC}_$name.o"; else objs="$objs ../../build/obj/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -no-hip-rt $objs \
     -o ../../build/variants/libdistance_hip_$name.so -L../../build/hipstub -lamdhip64 -ldl -Wl,-rpath,/opt/rocm/lib
  echo built $name
done
