#!/bin/bash
# Run ON the GPU box (via gpurun) from the repo root: every bench workload + the side tables that profiles/r02 holds.
R=$(pwd); O=$R/gpurun_out/refresh; mkdir -p $O
for w in C3raw C2 C3 C5; do
  timeout -k 10 300 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err || echo "bench $w failed"
  tail -c 200 $O/bench_$w.json | head -c 100; echo
done
timeout -k 10 300 python bench.py --workload C4 > $O/bench_C4.json 2> $O/bench_C4.err || echo "bench C4 failed"
timeout -k 10 400 python tools/cbench.py --reps 3 > $O/cbench_paths_c3.txt 2>&1
timeout -k 10 300 python tools/nsweep.py > $O/nsweep_paths.txt 2>&1
make -s -C tools/ubench store_rate && timeout -k 5 120 tools/ubench/store_rate 50000 32 1 > $O/ubench_store_rate.txt 2>&1
echo refresh done
