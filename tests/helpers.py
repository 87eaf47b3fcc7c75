"""Shared test inputs: seeded synthetic alignments over the full 17-code Paradis alphabet."""
import numpy as np

# the 17 valid codes (src/encoding.rs:7-38)
CODES = np.array([136, 72, 40, 24, 192, 160, 144, 96, 80, 48, 224, 176, 208, 112, 240, 244, 242],
                 np.uint8)
LETTERS = b"AGCTRMWSKYVHDBN-?"
KNOWN = (136, 72, 40, 24)


def random_alignment(n, L, seed, p_ambig=0.02, p_gap=0.02, divergence=0.05):
    """Root sequence + per-row substitutions + N/gap/IUPAC noise; returns uint8 codes (n, L)."""
    rng = np.random.default_rng(seed)
    root = rng.choice(np.array(KNOWN, np.uint8), size=L, p=[0.30, 0.20, 0.18, 0.32])
    codes = np.tile(root, (n, 1))
    if n and L:
        mut = rng.random((n, L)) < divergence
        codes[mut] = rng.choice(np.array(KNOWN, np.uint8), size=int(mut.sum()))
        amb = rng.random((n, L)) < p_ambig
        codes[amb] = rng.choice(CODES[4:14], size=int(amb.sum()))
        gap = rng.random((n, L)) < p_gap
        codes[gap] = rng.choice(CODES[14:], size=int(gap.sum()))
    return np.ascontiguousarray(codes)


def uniform_codes(n, L, seed):
    """Every one of the 17 codes equally likely at every site (adversarial for the predicates)."""
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(rng.choice(CODES, size=(n, L)))


def to_fasta_bytes(codes_row):
    lut = {int(c): LETTERS[k] for k, c in enumerate(CODES)}
    return bytes(lut[int(c)] for c in codes_row)
