# on the GPU box: the tn93 workload (C3) with the kernel time, then the finalisation tests
cd $GRAFT_REPO_ROOT
python3 bench.py --workload C3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/c3_probe.json 2> gpurun_out/c3_probe.err; echo rc=$?
python3 -c "
import json
r=json.loads(open('gpurun_out/c3_probe.json').read().strip().split('\n')[-1])
print('C3', round(r['ms_per_step'],3), r['kernels_ms'], r['roofline']['frac'], r['verify'])"
