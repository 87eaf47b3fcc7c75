#!/bin/bash
# Measurement builds of the consensus pair kernel: tools/variants.sh name "-DMACRO ..." [name "-D..."]...
# -> build/variants/libdistance_hip_<name>.so (select with DST_LIB_PATH).  Only dst_consensus.hip is rebuilt.
set -e
cd "$(dirname "$0")/../distance_amd/csrc"
mkdir -p ../../build/variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs -x hip -c dst_consensus.hip -o ../../build/variants/dst_consensus_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -no-hip-rt ../../build/obj/dst_kernels.o ../../build/variants/dst_consensus_$name.o \
     ../../build/obj/dst_api.o ../../build/obj/dst_stream.o ../../build/obj/dst_host.o ../../build/obj/dst_gather.o \
     -o ../../build/variants/libdistance_hip_$name.so -L../../build/hipstub -lamdhip64 -ldl -Wl,-rpath,/opt/rocm/lib
  echo built $name
done
