#!/usr/bin/env python3
"""CLI end to end at C3 scale on the GPU box: 50,000 x 30,000 FASTA (SURVEY 8(d) generator) -> `distance -m <m>`
-> 1.25e9 TSV lines to /dev/null, with DISTANCE_TIMING=1 phase times.  python tools/cli_c3_perf.py [n] [measure]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
measure = sys.argv[2] if len(sys.argv) > 2 else "raw"
L = 30_000
root = synth.root(synth.SEED ^ 3, L)
path = "/tmp/c3.fasta"
t0 = time.time()
with open(path, "wb") as fh:
    for r0 in range(0, n, 2000):
        codes = synth.records(synth.SEED ^ 3, root, r0, min(2000, n - r0))
        fh.write(synth.fasta_bytes(synth.SEED ^ 3, codes, first=r0))
print(f"# wrote {path}: {os.path.getsize(path) / 1e9:.2f} GB in {time.time() - t0:.1f} s", flush=True)
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "distance_amd", "cli", "distance")
runs = (([], {}, "TSV lines formatted on the GPU (default)"),
        ([], {"DISTANCE_HOST_FORMAT": "1"}, "host formatter pool, default threads"),
        (["-t", "32"], {"DISTANCE_HOST_FORMAT": "1"}, "host formatter pool, -t 32"))
for extra, env, what in runs:
    args = ["-m", measure, path, "-o", "/dev/null"] + extra
    t0 = time.time()
    r = subprocess.run([cli] + args, env=dict(os.environ, DISTANCE_TIMING="1", **env), capture_output=True)
    dt = time.time() - t0
    pairs = n * (n - 1) // 2
    print(f"# {what}: distance {' '.join(args)}: rc={r.returncode} {dt:.2f} s wall, {pairs / dt:.3e} pairs/s ({pairs} TSV lines)")
    print(r.stderr.decode())
os.remove(path)
