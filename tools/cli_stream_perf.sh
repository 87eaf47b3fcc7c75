#!/bin/bash
# On the GPU box: CLI stream mode end to end: loaded NL x L against a streamed NS x L file.
set -e
NL=${1:-1000}; NS=${2:-20000}; L=${3:-30000}; M=${4:-n_high}
cd "$(dirname "$0")/.."
python3 - "$NL" "$NS" "$L" <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
from helpers import random_alignment, to_fasta_bytes
nl, ns, L = map(int, sys.argv[1:4])
for name, n, seed in (("/tmp/cli_loaded.fasta", nl, 5), ("/tmp/cli_streamed.fasta", ns, 6)):
    codes = random_alignment(n, L, seed, p_ambig=1e-4, p_gap=1e-3, divergence=1e-3)
    with open(name, "wb") as fh:
        for i, r in enumerate(codes):
            fh.write(b">%s%d\n" % (b"L" if n == nl else b"S", i) + to_fasta_bytes(r) + b"\n")
PY
ls -la /tmp/cli_loaded.fasta /tmp/cli_streamed.fasta
S=$(date +%s%N)
./distance_amd/cli/distance -m $M -t 16 -i /tmp/cli_loaded.fasta -s /tmp/cli_streamed.fasta -o /tmp/cli_stream.tsv
E=$(date +%s%N)
echo "CLI stream -m $M: $NL loaded x $NS streamed x $L: $(( (E - S) / 1000000 )) ms wall"
wc -l /tmp/cli_stream.tsv | awk -v a=$NL -v b=$NS '{printf "%d lines (expected %d)\n", $1, a*b+1}'
