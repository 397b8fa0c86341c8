"""bench.py itself: the N = 1 line and the N > 1 path (two ranks on this one GPU over the host-file transport, MECAT_BENCH_BACKEND=gloo —
what the driver launches with torch.distributed.run on a multi-GPU node, minus RCCL) give the same candidates and overlaps."""
import json
import os
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu
FLAGS = ["--workload", "config1", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e", "--no-extras"]


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_line_one_rank_and_two_ranks_agree():
    bench = os.path.join(H.ROOT, "bench.py")
    r1 = subprocess.run([sys.executable, bench, "--gpus", "1"] + FLAGS, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    a = _line(r1.stdout)
    assert a["n_gpus"] == 1 and a["value"] > 0 and a["roofline"]["achieved"] > 0
    env = dict(os.environ, MECAT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29541", bench, "--gpus", "2"] + FLAGS, capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    b = _line(r2.stdout)
    assert b["n_gpus"] == 2 and b["scaling"] == "strong"
    assert b["candidates"] == a["candidates"] and b["overlaps_ok"] == a["overlaps_ok"]
