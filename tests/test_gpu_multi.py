"""GPU (-m gpu): the multi-rank job of bench.py with the ranks SHARING the one GPU of the test box and the exchange
over gloo (host-staged): the same partition, sub-slab indexing, uint16 wire format and rank-0 finalisation as the
RCCL run, checked by bench.py's own `verify` (sampled rows of the gathered result against a single-engine dense
run).  RCCL itself needs one GPU per rank and has not run on N>1 hardware."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(world, port, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--rehearse-gloo", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("world,extra", [(3, ["--workload", "C2"]),                                  # raw: uint16 tallies on the wire
                                         (2, ["--workload", "C2", "--measure", "tn93", "--path", "dense"]),  # f64 on the wire, dense kernels
                                         (2, ["--workload", "C2", "--measure", "n_high"])])
def test_gathered_multi_rank_job_matches_a_single_engine(world, extra):
    d = run_ranks(world, 29500 + world * 7 + len(extra), ["--exchange", "gather"] + extra)
    assert d["n_gpus"] == world and d["scaling"] == "strong"
    assert d["verify"]["rows_bad"] == 0 and d["verify"]["rows_checked"] >= 5
    assert d["value"] > 0 and d["config"]["pairs"] == 10_000 * 9_999 // 2


@pytest.mark.parametrize("world,extra", [(3, ["--workload", "C2"]), (2, ["--workload", "C2", "--measure", "tn93"])])
def test_sharded_multi_rank_job_matches_a_single_engine(world, extra):
    """bench.py's default for N>1: every rank's slab stays in its own HBM; every rank checks rows of its slab."""
    d = run_ranks(world, 29600 + world * 7 + len(extra), extra)
    assert d["n_gpus"] == world and d["scaling"] == "strong"
    assert d["verify"]["rows_bad"] == 0 and d["verify"]["rows_checked"] >= 5
    assert d["value"] > 0 and d["config"]["pairs"] == 10_000 * 9_999 // 2
