#!/usr/bin/env python3
"""On the GPU box: one HIP runtime per process whatever the import/initialisation order."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = {
    "engine_then_torch": "import distance_amd as da; e = da.Engine(0); import torch; torch.cuda.init(); x = torch.ones(4, device='cuda'); print(float(x.sum()))",
    "torch_then_engine": "import torch; torch.cuda.init(); import distance_amd as da; e = da.Engine(0); x = torch.ones(4, device='cuda'); print(float(x.sum()))",
}
for name, code in CODE.items():
    tail = "; import re; libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)); print('HIPLIBS', libs)"
    r = subprocess.run([sys.executable, "-c", code + tail], capture_output=True, text=True, cwd=ROOT)
    print(name, "rc", r.returncode, r.stdout.strip().replace("\n", " | "), r.stderr.strip().splitlines()[-1:] if r.returncode else "")
