"""GPU (-m gpu): the `distance` CLI end to end — TSV text identical to what the reference's
gather_write would print for the oracle's distances (same header, order, `{}` / `{:.12}`)."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from helpers import LETTERS, CODES, random_alignment

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "distance_amd", "cli", "distance")
ALL = ("n", "n_high", "raw", "jc69", "k80", "tn93")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(CLI):
        subprocess.run(["make", "-C", os.path.dirname(CLI)], check=True)


def to_text(codes, rng=None, lower_frac=0.0):
    lut = {int(c): chr(LETTERS[k]) for k, c in enumerate(CODES)}
    rows = []
    for r in codes:
        s = "".join(lut[int(c)] for c in r)
        if rng is not None and lower_frac:
            s = "".join(ch.lower() if rng.random() < lower_frac else ch for ch in s)
        rows.append(s)
    return rows


def write_fasta(path, ids, seqs, width=0):
    with open(path, "w") as fh:
        for i, s in zip(ids, seqs):
            fh.write(f">{i} some description\n")
            if width:
                for k in range(0, len(s), width):
                    fh.write(s[k:k + width] + "\n")
            else:
                fh.write(s + "\n")


def cli(args, stdin=None):
    """Every run twice: the TSV lines formatted by the GPU (the default in load mode, dst_text_*) and by the host's
    formatter pool (DISTANCE_HOST_FORMAT=1) — the two must be the same bytes."""
    r = subprocess.run([CLI] + args, input=stdin, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    h = subprocess.run([CLI] + args, input=stdin, capture_output=True, env=dict(os.environ, DISTANCE_HOST_FORMAT="1"))
    assert h.returncode == 0, h.stderr.decode()
    assert r.stdout == h.stdout, "GPU-formatted and host-formatted TSV differ"
    if "-s" in args:   # stream mode sends the codes' high nibbles by default (DST_WIRE_NIBBLES): the byte format must agree
        w = subprocess.run([CLI] + args, input=stdin, capture_output=True, env=dict(os.environ, DISTANCE_WIRE="codes"))
        assert w.returncode == 0, w.stderr.decode()
        assert r.stdout == w.stdout, "4-bit and 8-bit wire formats give different TSV"
    return r.stdout.decode()


def expected_square(measure, ids, codes):
    d = oracle.all_pairs_square(measure, codes)
    ij = oracle.pairs_square(len(codes))
    vals = [int(x) for x in d] if measure in oracle.INT_MEASURES else list(d)
    return oracle.tsv(ids, ids, ij, vals)


def test_golden_tsv_through_the_cli(tmp_path, golden):
    for k, v in enumerate(golden["tsv"]):
        a = tmp_path / f"a{k}.fasta"
        write_fasta(a, [r[0] for r in v["loaded"]], [r[1] for r in v["loaded"]])
        if v["mode"] == "square":
            for extra in ([], ["-t", "2", "-b", "2"]):   # src/lib.rs:919-1002: output independent of -t/-b
                assert cli(["-m", v["measure"], str(a)] + extra) == v["expected"]
        else:
            other = v.get("streamed") or v["second"]
            b = tmp_path / f"b{k}.fasta"
            write_fasta(b, [r[0] for r in other], [r[1] for r in other])
            if v["mode"] == "stream":
                assert cli(["-m", v["measure"], "-i", str(a), "-s", str(b)]) == v["expected"]
                assert cli(["-m", v["measure"], str(a), "-s", "-"], stdin=b.read_bytes()) == v["expected"]
            else:
                assert cli(["-m", v["measure"], str(a), str(b)]) == v["expected"]
                assert cli(["-m", v["measure"], "-i", str(a), str(b)]) == v["expected"]


@pytest.mark.parametrize("measure", ALL)
def test_square_matches_oracle_text(tmp_path, measure):
    codes = random_alignment(57, 701, seed=3)
    ids = [f"seq{i}" for i in range(len(codes))]
    f = tmp_path / "a.fasta"
    write_fasta(f, ids, to_text(codes), width=60)     # multi-line records
    want = expected_square(measure, ids, codes)
    assert cli(["-m", measure, str(f)]) == want
    # slabs, formatting threads and -b never change the bytes
    assert cli(["-m", measure, str(f), "--slab-pairs", "100", "-t", "3", "-b", "7"]) == want
    out = tmp_path / "o.tsv"
    assert cli(["-m", measure, "-i", str(f), "-o", str(out)]) == ""
    assert out.read_text() == want
    assert cli(["-m", measure], stdin=f.read_bytes()) == want      # stdin when no input is named


def test_several_contexts_keep_canonical_order(tmp_path):
    """The multi-GPU slab pipeline (slab k on context k mod G, ordered writer), exercised with
    three contexts on the one GPU of the test box."""
    codes = random_alignment(120, 300, seed=8)
    ids = [f"r{i}" for i in range(len(codes))]
    fa = tmp_path / "a.fasta"
    write_fasta(fa, ids, to_text(codes))
    for m in ("n", "tn93"):
        want = expected_square(m, ids, codes)
        assert cli(["-m", m, str(fa), "--devices", "0,0,0", "--slab-pairs", "500", "-t", "4"]) == want
    b = random_alignment(50, 300, seed=9)
    idb = [f"q{i}" for i in range(len(b))]
    fb = tmp_path / "b.fasta"
    write_fasta(fb, idb, to_text(b))
    one = cli(["-m", "k80", "-i", str(fa), "-s", str(fb)])
    assert cli(["-m", "k80", "-i", str(fa), "-s", str(fb), "--devices", "0,0", "--slab-pairs", "700"]) == one


def test_stream_batches_sharded_over_contexts(tmp_path):
    """Stream mode: 41 streamed records in batches of 2 dealt round-robin to three contexts, each with its own
    overlapped pipeline (dst_stream_*); output in input order; tn93 with lower-case streamed letters."""
    a = random_alignment(30, 500, seed=21)
    b = random_alignment(41, 500, seed=22)
    ida, idb = [f"a{i}" for i in range(len(a))], [f"b{i}" for i in range(len(b))]
    fa, fb = tmp_path / "a.fasta", tmp_path / "b.fasta"
    rng = np.random.default_rng(2)
    tb = to_text(b, rng, lower_frac=0.4)
    write_fasta(fa, ida, to_text(a))
    write_fasta(fb, idb, tb, width=70)
    ca = oracle.count_bases_matrix(a)
    cb = np.stack([oracle.encode_count_bases(s.encode())[1] for s in tb])
    for m in ("n_high", "tn93"):
        d = oracle.all_pairs_rect(m, a, b, counts_a=ca, counts_b=cb) if m == "tn93" else oracle.all_pairs_rect(m, a, b)
        ij = [(i, j) for j in range(len(b)) for i in range(len(a))]
        vals = [int(d[i, j]) if m in oracle.INT_MEASURES else d[i, j] for i, j in ij]
        want = oracle.tsv(ida, idb, ij, vals)
        for extra in (["--slab-pairs", "60"], ["--slab-pairs", "60", "--devices", "0,0,0", "-t", "3"], []):
            assert cli(["-m", m, "-i", str(fa), "-s", str(fb)] + extra) == want, (m, extra)
        assert cli(["-m", m, "-i", str(fa), "-s", "-"], stdin=open(fb, "rb").read()) == want


def test_default_measure_is_raw(tmp_path):
    codes = random_alignment(9, 100, seed=5)
    ids = [f"s{i}" for i in range(9)]
    f = tmp_path / "a.fasta"
    write_fasta(f, ids, to_text(codes))
    assert cli([str(f)]) == expected_square("raw", ids, codes)


@pytest.mark.parametrize("measure", ALL)
def test_two_files_and_stream_match_oracle_text(tmp_path, measure):
    a = random_alignment(23, 333, seed=11)
    b = random_alignment(31, 333, seed=12)
    ida = [f"a{i}" for i in range(len(a))]
    idb = [f"b{i}" for i in range(len(b))]
    fa, fb = tmp_path / "a.fasta", tmp_path / "b.fasta"
    rng = np.random.default_rng(1)
    ta, tb = to_text(a), to_text(b, rng, lower_frac=0.3)     # lower case in the second file
    write_fasta(fa, ida, ta)
    write_fasta(fb, idb, tb)
    ca = oracle.count_bases_matrix(a)
    # two loaded files: counts by code for both (src/lib.rs:233-239)
    d = oracle.all_pairs_rect(measure, a, b)
    ij = [(i, j) for i in range(len(a)) for j in range(len(b))]
    vals = [int(d[i, j]) if measure in oracle.INT_MEASURES else d[i, j] for i, j in ij]
    assert cli(["-m", measure, str(fa), str(fb)]) == oracle.tsv(ida, idb, ij, vals)
    # stream: streamed-record-major; streamed tn93 counts only upper-case characters
    # (src/fastaio.rs:136-142)
    cb = np.stack([oracle.encode_count_bases(s.encode())[1] for s in tb])
    d = oracle.all_pairs_rect(measure, a, b, counts_a=ca, counts_b=cb) if measure == "tn93" else d
    ij = [(i, j) for j in range(len(b)) for i in range(len(a))]
    vals = [int(d[i, j]) if measure in oracle.INT_MEASURES else d[i, j] for i, j in ij]
    want = oracle.tsv(ida, idb, ij, vals)
    assert cli(["-m", measure, "-i", str(fa), "-s", str(fb)]) == want
    assert cli(["-m", measure, "-i", str(fa), "-s", str(fb), "--slab-pairs", "50", "-b", "3"]) == want


def test_special_float_text(tmp_path):
    f = tmp_path / "a.fasta"
    write_fasta(f, ["x", "y", "z", "n"], ["ACGTACGTACGTACGTAAAA", "ACGTACGTACGTACGTAAAA",
                                           "CATGCATGCATGCATTAAAA", "NNNNNNNNNNNNNNNNNNNN"])
    out = cli(["-m", "jc69", str(f)]).splitlines()
    assert out[1] == "x\ty\t-0.000000000000"        # -0.75*ln(1): Rust prints the sign of -0.0
    assert out[2] == "x\tz\tinf"                    # p = 0.75
    assert out[3] == "x\tn\tNaN"                    # 0/0
    out = cli(["-m", "tn93", str(f)]).splitlines()
    assert out[1] == "x\ty\t0.000000000000"         # src/measures.rs:188-190 normalises -0.0


def test_fasta_corner_cases_through_the_cli(tmp_path):
    """Header-only records (width 0), CRLF line ends, wrapped + lower-case sequence, descriptions."""
    f = tmp_path / "empty_seqs.fasta"
    f.write_text(">a\n>b some words\n>c\n")
    out = cli(["-m", "raw", str(f)]).splitlines()
    assert out == ["sequence1\tsequence2\tdistance", "a\tb\tNaN", "a\tc\tNaN", "b\tc\tNaN"]
    assert cli(["-m", "n", str(f)]).splitlines()[1:] == ["a\tb\t0", "a\tc\t0", "b\tc\t0"]
    g = tmp_path / "crlf.fasta"
    g.write_bytes(b">x desc one\r\nACGT\r\nacgt\r\n>y\r\nACGTAC\r\nGA\r\n")
    out = cli(["-m", "n_high", str(g)]).splitlines()
    assert out[1] == "x\ty\t1"          # ACGTacgt vs ACGTACGA: one difference, case-insensitive
    assert cli(["-m", "raw", str(g)]).splitlines()[1] == "x\ty\t0.125000000000"


def test_single_record_and_error_paths_with_gpu(tmp_path):
    one = tmp_path / "one.fasta"
    one.write_text(">only\nACGT\n")
    assert cli(["-m", "raw", str(one)]) == "sequence1\tsequence2\tdistance\n"     # 0 pairs: header only
    a = tmp_path / "a.fasta"
    a.write_text(">a1\nACGT\n>a2\nACGA\n")
    b = tmp_path / "b.fasta"
    b.write_text(">b1\nACG\n")
    r = subprocess.run([CLI, str(a), str(b)], capture_output=True)       # src/fastaio.rs:206-208
    assert r.returncode == 1 and b"Different length sequences in alignment(s): 4 vs 3" in r.stderr
    r = subprocess.run([CLI, "-i", str(a), "-s", str(b)], capture_output=True)   # src/fastaio.rs:246-248
    assert r.returncode == 1 and b"Different length sequences in alignment(s): 3 vs 4" in r.stderr
    bad = tmp_path / "bad.fasta"
    bad.write_text(">s1\nACGT\n>s2\nAC*T\n")
    r = subprocess.run([CLI, "-i", str(a), "-s", str(bad)], capture_output=True)
    assert r.returncode == 1 and b"Invalid nucleotide character in record 's2': '*'" in r.stderr
    # stream_fasta() compares widths BEFORE encoding (src/fastaio.rs:246-254): short + invalid reports the width
    both = tmp_path / "both.fasta"
    both.write_text(">s1\nACGT\n>s2\nA*T\n")
    r = subprocess.run([CLI, "-i", str(a), "-s", str(both)], capture_output=True)
    assert r.returncode == 1 and b"Different length sequences in alignment(s): 3 vs 4" in r.stderr
    empty = tmp_path / "empty.fasta"
    empty.write_text("")
    r = subprocess.run([CLI, "-i", str(a), "-s", str(empty)], capture_output=True)  # src/fastaio.rs:281-283
    assert r.returncode == 1 and b"Empty FASTA file" in r.stderr
    assert r.stdout == b"sequence1\tsequence2\tdistance\n"             # the header is already out, as in the reference


def test_broken_pipe_exits_zero(tmp_path):
    """handle_broken_pipe(): src/lib.rs:598-608 — a closed stdout is not an error."""
    codes = random_alignment(400, 200, seed=13)
    f = tmp_path / "a.fasta"
    write_fasta(f, [f"s{i}" for i in range(len(codes))], to_text(codes))
    p = subprocess.Popen([CLI, "-m", "n", str(f), "--slab-pairs", "2000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    first = p.stdout.readline()
    p.stdout.close()
    rc = p.wait(timeout=60)
    assert first == b"sequence1\tsequence2\tdistance\n" and rc == 0, p.stderr.read()
