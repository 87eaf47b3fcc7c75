"""CPU: the host-only parts of the `distance` CLI (no GPU needed): FASTA tokenisation, the exact
`{:.12}` formatter against libc/Python formatting, argument surface and error exits."""
import math
import os
import random
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "distance_amd", "cli", "distance")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(CLI):
        subprocess.run(["make", "-C", os.path.dirname(CLI)], check=True)


def run(args, stdin=b""):
    return subprocess.run([CLI] + args, input=stdin, capture_output=True)


def test_fasta_tokenisation_follows_rust_bio():
    text = b">s1 some description\nACGT\nAC\n\n>s2\r\nAAAA\r\nCC\r\n>s3\tdesc2\nGG"
    out = run(["--host-selftest", "fasta"], text).stdout.decode().splitlines()
    assert out == ["s1\tsome description\tACGTAC", "s2\t<none>\tAAAACC", "s3\tdesc2\tGG"]
    # the reference's own fixtures (src/lib.rs:906-914)
    out = run(["--host-selftest", "fasta"], b">seq1\nATGATG\n>seq2\nATGATC\n").stdout.decode().splitlines()
    assert out == ["seq1\t<none>\tATGATG", "seq2\t<none>\tATGATC"]
    out = run(["--host-selftest", "fasta"], b"ACGT\n>x\nAC\n").stdout.decode().splitlines()
    assert out == ["ERROR\tExpected > at record start."]
    assert run(["--host-selftest", "fasta"], b"").stdout == b""


def _py_fixed12(v):
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "-inf" if v < 0 else "inf"
    return "%.12f" % v          # libc-exact, half-to-even on the exact binary value, keeps -0.0


def test_fixed12_matches_exact_decimal_rounding():
    rng = random.Random(7)
    vals = [0.0, -0.0, 2.0 / 15.0, 0.5, 1.0, 5e-13, 2.5e-12, 1.5e-12, 1e-13, 0.1, 0.2, 0.3, 1e-300, 4.9e-324,
            123456.7890123456789, 1e15, 4503599627370495.5, 9007199254740993.0, 1e22, float("nan"),
            float("inf"), float("-inf"), 0.0000000000005, 0.9999999999995, 0.99999999999949996]
    for _ in range(20000):
        kind = rng.random()
        if kind < 0.5:
            vals.append(rng.random() * 10 ** rng.randint(-14, 2))
        elif kind < 0.8:
            vals.append(rng.randint(0, 10 ** 13) / 1e12 + rng.choice([0.0, 5e-13, -5e-13]))   # near ties
        else:
            vals.append(struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0])       # any bit pattern
    vals += [-v for v in vals if not math.isnan(v)]
    text = "\n".join(float.hex(v) if not (math.isnan(v) or math.isinf(v)) else repr(v) for v in vals) + "\n"
    out = run(["--host-selftest", "format"], text.encode()).stdout.decode().splitlines()
    assert len(out) == len(vals)
    for v, got in zip(vals, out):
        assert got == _py_fixed12(v), (v, float.hex(v) if not math.isnan(v) else v)


def test_argument_surface_matches_clap_definition():
    out = run(["--host-selftest", "args", "-t", "8", "-m", "jc69", "a.fasta", "-o", "o.tsv"]).stdout.decode()
    assert "measure=jc69 threads=8 batchsize=1" in out and "[pos a.fasta]" in out and "output=o.tsv" in out
    out = run(["--host-selftest", "args", "-i", "a", "b", "-b", "1000", "--measure=tn93"]).stdout.decode()
    assert "[-i a][-i b]" in out and "batchsize=1000" in out and "measure=tn93" in out
    out = run(["--host-selftest", "args", "small.fasta", "-s", "-"]).stdout.decode()
    assert "stream=-" in out and "[pos small.fasta]" in out
    out = run(["--host-selftest", "args"]).stdout.decode()
    assert "measure=raw" in out and "threads=0(default)" in out     # defaults: src/lib.rs:104-124
    r = run(["-m", "hamming"])
    assert r.returncode == 2 and b"possible values: n, n_high, raw, jc69, k80, tn93" in r.stderr
    r = run(["-t", "x"])
    assert r.returncode == 2
    assert run(["-V"]).stdout.decode().strip() == "distance 0.3.1"
    assert b"Usage: All sequences across all input files must be the same length." in run(["-h"]).stdout


def test_set_up_errors_before_any_gpu_work(tmp_path):
    f = tmp_path / "a.fasta"
    f.write_text(">a\nACGT\n")
    r = run([str(f), "-i", str(f)])       # src/lib.rs:182-184
    assert r.returncode == 1 and b"don't use both positional arguments and the -i/--input flag" in r.stderr
    r = run(["-s", str(f)])               # src/lib.rs:196-199 (no loaded file named)
    assert r.returncode == 1 and b"you must also provide exactly one other file to be loaded" in r.stderr
    r = run([str(tmp_path / "missing.fasta")])
    assert r.returncode == 1 and b"NotFound" in r.stderr
    bad = tmp_path / "bad.fasta"
    bad.write_text(">a\nACGU\n")
    r = run([str(bad)])                   # src/fastaio.rs:89-91
    assert r.returncode == 1 and b"Invalid nucleotide character in record 'a': 'U'" in r.stderr
    ragged = tmp_path / "ragged.fasta"
    ragged.write_text(">a\nACGT\n>b\nACG\n")
    r = run([str(ragged)])                # src/fastaio.rs:93-95
    assert r.returncode == 1 and b"Different length sequences in alignment(s): 3 vs 4" in r.stderr
    empty = tmp_path / "empty.fasta"
    empty.write_text("")
    r = run([str(empty)])                 # src/fastaio.rs:97-99
    assert r.returncode == 1 and b"Empty FASTA file" in r.stderr
    # load_fasta() encodes a record before it compares widths (src/fastaio.rs:182-190): a record that is both
    # too short and holds an invalid character reports the character — also when it sits in a later parse block
    both = tmp_path / "both.fasta"
    both.write_text(">a\nACGT\n>b\nACX\n")
    r = run([str(both)])
    assert r.returncode == 1 and b"Invalid nucleotide character in record 'b': 'X'" in r.stderr
    later = tmp_path / "later.fasta"
    later.write_text(">a\nACGT\n>b\nACGT\n>c\nAX\n")
    r = subprocess.run([CLI, str(later)], capture_output=True, env=dict(os.environ, DISTANCE_PARSE_BLOCK_BYTES="6"))
    assert r.returncode == 1 and b"Invalid nucleotide character in record 'c': 'X'" in r.stderr


def test_block_parallel_parser_matches_the_sequential_reader():
    """BlockReader + parse_stream (blocks of 5..64 bytes) see the same records as FastaReader."""
    texts = [
        b">s1 d\nACGT\nAC\n>s2\r\nAAAAAA\r\n>s3\nCC\nGG\nTT\n",
        b">only\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n",
        b">a\nAC>GT\n>b\nACGGT\n",                    # '>' inside a line is sequence text, not a header
        b">a\nAAAA\n\n\n>b\nCCCC",                     # blank lines, no final newline
        b"",
    ]
    for text in texts:
        seq = [l.split("\t") for l in run(["--host-selftest", "fasta"], text).stdout.decode().splitlines()]
        want = [f"{r[0]}\t{len(r[2])}" for r in seq] + [f"records\t{len(seq)}"]
        for blk in ("5", "9", "64"):
            got = run(["--host-selftest", "fasta-blocks", "--slab-pairs", blk], text).stdout.decode().splitlines()
            assert got == want, (text, blk)
    # ragged widths are an error in file order, whichever block they fall in (src/fastaio.rs:188-190)
    r = run(["--host-selftest", "fasta-blocks", "--slab-pairs", "6"], b">a\nACGT\n>b\nACG\n>c\nACGT\n")
    assert r.returncode == 1 and b"Different length sequences in alignment(s): 3 vs 4" in r.stderr
