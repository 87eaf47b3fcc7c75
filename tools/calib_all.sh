#!/bin/bash
# every table of profiles/r02/consensus_calibration.txt (run on the GPU box from the repo root)
O=gpurun_out/refresh/consensus_calibration.txt; mkdir -p gpurun_out/refresh
{
echo "## unstructured substitutions"
timeout -k 10 400 python tools/calibrate.py --measures n_high,raw,tn93 --rates 0.001,0.003,0.01,0.02,0.03,0.05,0.1
echo; echo "## clade-structured: 33 % of the records share substitutions at 2 % of the sites"
timeout -k 10 400 python tools/calibrate.py --structured --measures n_high,raw,tn93 --rates 0.001,0.01,0.03
echo; echo "## clade-structured, 10 % of the sites"
timeout -k 10 400 python tools/calibrate.py --structured --clade-sites 0.1 --measures raw,tn93 --rates 0.001
echo; echo "## clade-structured at C3 size"
timeout -k 10 400 python tools/calibrate.py --n 50000 --len 30000 --structured --measures raw,tn93 --rates 0.001
echo; echo "## 3000 x 30000"
timeout -k 10 400 python tools/calibrate.py --n 3000 --len 30000 --measures raw --rates 0.001,0.01,0.03
} > $O 2>&1
grep -c MISPICK $O; tail -5 $O
