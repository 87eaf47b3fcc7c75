#!/bin/bash
# On the GPU box: end-to-end CLI timing on a synthetic FASTA (host parse -> GPU -> TSV on disk).
set -e
N=${1:-4000}; L=${2:-30000}; M=${3:-raw}
cd "$(dirname "$0")/.."
python3 - "$N" "$L" <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
from helpers import random_alignment, to_fasta_bytes
n, L = int(sys.argv[1]), int(sys.argv[2])
codes = random_alignment(n, L, 5, p_ambig=1e-4, p_gap=1e-3, divergence=1e-3)
with open("/tmp/cli_perf.fasta", "wb") as fh:
    for i, r in enumerate(codes):
        fh.write(b">seq%d\n" % i + to_fasta_bytes(r) + b"\n")
PY
ls -la /tmp/cli_perf.fasta
for T in 16; do
  S=$(date +%s%N)
  ./distance_amd/cli/distance -m $M -t $T /tmp/cli_perf.fasta -o /tmp/cli_perf.tsv
  E=$(date +%s%N)
  echo "CLI -m $M -t $T: $(( (E - S) / 1000000 )) ms wall"
  ls -la /tmp/cli_perf.tsv; wc -l /tmp/cli_perf.tsv | awk -v n=$N '{printf "%d lines (expected %d)\n", $1, n*(n-1)/2+1}'
done
head -3 /tmp/cli_perf.tsv; md5sum /tmp/cli_perf.tsv
