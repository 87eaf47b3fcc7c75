#!/bin/bash
# gpurun_out/refresh + gpurun_out/prof_<tag> (tools/refresh_r02.sh, tools/profile_r02.sh, tools/calib_all.sh) -> profiles/r02
R=$(cd "$(dirname "$0")/.." && pwd); TAG=${1:-r02d}; P=$R/profiles/r02; O=$R/gpurun_out/refresh
python3 $R/tools/prof_summary_r02.py $R/gpurun_out/prof_$TAG $P > /dev/null
cp $R/gpurun_out/prof_$TAG/trace_bench.json $P/bench_under_rocprof.json
for w in C2 C3 C5 C4; do cp $O/bench_$w.json $P/bench_$(echo $w | tr A-Z a-z).json; done
cp $O/bench_C3raw.json $P/bench_default.json
for f in cbench_paths_c3 nsweep_paths consensus_calibration ubench_store_rate; do [ -f $O/$f.txt ] && grep -v amdgpu.ids $O/$f.txt > $P/$f.txt; done
head -18 $P/summary.txt
