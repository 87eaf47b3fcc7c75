#!/bin/bash
# Run ON the GPU box: the bench lines and tables profiles/r03/ keeps (tools/collect copies them from gpurun_out/refresh_r03)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh_r03; mkdir -p $O
cd $R
for w in C3raw C2 C3 C5; do
  timeout -k 10 400 python3 bench.py --workload $w --steps 20 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"
done
for v in 0 1 2; do
  timeout -k 10 200 python3 bench.py --workload C3 --path dense --variant $v --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $O/tn93_dense_v$v.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$O/tn93_dense_v$v.json'));print('tn93 dense variant $v', round(d['ms_per_step'],2),'ms')"
done
