#!/usr/bin/env python3
"""Writes tests/golden/reference_vectors.json: every known-answer vector the reference's own
unit tests hold for the hot path, as DATA (inputs + expected outputs), each citing the
reference file:line it was read from.

The Rust reference cannot be executed here (no cargo/rustc), so nothing in this file was produced
by running it.  Integer/byte expectations are literal values from the reference's assert_eq!
lines.  The four float expectations are the closed-form f64 expressions the reference's tests
assert against (measures.rs:244, 251-254, 263-268, 282-307), evaluated here in IEEE f64 with
Python's math.log/sqrt (glibc libm, the same libm the reference binary links) in the same
operation order, and stored as C99 hex floats.
"""
import json
import math
import os

FASTA = "ATGATGATGATGCCC"   # fastaio.rs:344-346, measures.rs:202-204 (TARGET_FASTA)
OTHER = "ATTATTATGATGCCC"   # fastaio.rs:348-350, measures.rs:206-208 (QUERY_FASTA)


def tn93_expected():
    # measures.rs:282-304, same names and operation order
    g_A = 8.0 / 30.0
    g_T = 10.0 / 30.0
    g_C = 6.0 / 30.0
    g_G = 6.0 / 30.0
    g_R = (8.0 + 6.0) / 30.0
    g_Y = (7.0 + 9.0) / 30.0
    k1 = 2.0 * g_A * g_G / g_R
    k2 = 2.0 * g_T * g_C / g_Y
    k3 = 2.0 * (g_R * g_Y - g_A * g_G * g_Y / g_R - g_T * g_C * g_R / g_Y)
    P1 = 0.0 / 15.0
    P2 = 0.0 / 15.0
    Q = (2.0 - (0.0 + 0.0)) / 15.0
    w1 = 1.0 - P1 / k1 - Q / (2.0 * g_R)
    w2 = 1.0 - P2 / k2 - Q / (2.0 * g_Y)
    w3 = 1.0 - Q / (2.0 * g_R * g_Y)
    return -k1 * math.log(w1) - k2 * math.log(w2) - k3 * math.log(w3)


def main():
    P = 0.0 / 15.0
    Q = 2.0 / 15.0
    fasta_codes = [136, 24, 72, 136, 24, 72, 136, 24, 72, 136, 24, 72, 40, 40, 40]
    vectors = {
        "_about": "Known-answer vectors held by the reference's own unit tests; see make_reference_vectors.py",
        "sequences": {"FASTA": FASTA, "OTHER": OTHER},
        "encode": [
            {"seq": FASTA, "codes": fasta_codes, "cite": "src/fastaio.rs:380-389,413-422"},
        ],
        "count_bases": [
            {"seq": FASTA, "A": 4, "T": 4, "C": 3, "G": 4, "cite": "src/fastaio.rs:358-367,392-400"},
        ],
        "get_differences": [
            {"seq": FASTA, "other": OTHER, "differences": [2, 5], "cite": "src/fastaio.rs:369-377,402-411"},
        ],
        "consensus": [
            {"rows": [FASTA, OTHER], "codes": fasta_codes, "cite": "src/fastaio.rs:431-437 (1-1 tie at sites 2 and 5 resolves to G)"},
            {"rows": [FASTA, FASTA], "codes": fasta_codes, "cite": "src/fastaio.rs:439-445"},
            {"rows": [OTHER, OTHER],
             "codes": [136, 24, 24, 136, 24, 24, 136, 24, 72, 136, 24, 72, 40, 40, 40],
             "cite": "src/fastaio.rs:447-453"},
        ],
        "measures": [
            {"measure": "n_high", "query": FASTA, "target": OTHER, "int": 2, "cite": "src/measures.rs:219-224 (snp(&target,&query))"},
            {"measure": "n", "query": FASTA, "target": OTHER, "int": 2, "cite": "src/measures.rs:226-238"},
            {"measure": "raw", "query": OTHER, "target": FASTA, "hex": (2.0 / 15.0).hex(), "cite": "src/measures.rs:240-245"},
            {"measure": "jc69", "query": OTHER, "target": FASTA,
             "hex": (-0.75 * math.log(1.0 - (4.0 / 3.0) * (2.0 / 15.0))).hex(), "cite": "src/measures.rs:247-255"},
            {"measure": "k80", "query": OTHER, "target": FASTA,
             "hex": (-0.5 * math.log((1.0 - 2.0 * P - Q) * math.sqrt(1.0 - 2.0 * Q))).hex(), "cite": "src/measures.rs:257-269"},
            {"measure": "tn93", "query": FASTA, "target": OTHER, "hex": tn93_expected().hex(), "cite": "src/measures.rs:271-308"},
        ],
        "pairs_square": [
            {"n": 4, "pairs": [[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]],
             "batches_b1": [1, 1, 1, 1, 1, 1], "batches_b4": [4, 2], "cite": "src/lib.rs:651-793"},
        ],
        "pairs_rectangle": [
            {"n1": 2, "n2": 2, "pairs": [[0, 0], [0, 1], [1, 0], [1, 1]],
             "batches_b1": [1, 1, 1, 1], "batches_b4": [4], "cite": "src/lib.rs:796-904"},
        ],
        "tsv": [
            {"mode": "square", "measure": "n", "loaded": [["seq1", "ATGATG"], ["seq2", "ATGATC"]],
             "expected": "sequence1\tsequence2\tdistance\nseq1\tseq2\t1\n", "cite": "src/lib.rs:906-1002 (threads 1|2, batchsize 1|2)"},
            {"mode": "stream", "measure": "n_high", "loaded": [["seq1", "ATGATG"], ["seq2", "ATGATC"]],
             "streamed": [["seqA", "ATGATG"]],
             "expected": "sequence1\tsequence2\tdistance\nseq1\tseqA\t0\nseq2\tseqA\t1\n", "cite": "src/lib.rs:1004-1068"},
            {"mode": "rect", "measure": "n_high", "loaded": [["seq1", "ATGATG"], ["seq2", "ATGATC"]],
             "second": [["seqA", "ATGATG"]],
             "expected": "sequence1\tsequence2\tdistance\nseq1\tseqA\t0\nseq2\tseqA\t1\n", "cite": "src/lib.rs:1070-1111"},
            {"mode": "rect", "measure": "n_high", "loaded": [["seqA", "ATGATG"]],
             "second": [["seq1", "ATGATG"], ["seq2", "ATGATC"]],
             "expected": "sequence1\tsequence2\tdistance\nseqA\tseq1\t0\nseqA\tseq2\t1\n", "cite": "src/lib.rs:1113-1153"},
        ],
    }
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
    with open(out, "w") as fh:
        json.dump(vectors, fh, indent=1)
        fh.write("\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
