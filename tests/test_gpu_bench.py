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
    # the line says what the communicator was and what the exchanges cost (the first run on a node must show rccl / N ranks here)
    x = b["exchange"]
    assert x["transport"].startswith("host files") and x["rccl_ranks"] == 0 and x["ms"] > 0 and x["calls_per_step"] >= 2
    # --simulate-ranks 2 (every rank's share alone on the GPU, no transport: the scaling bound of profiles/r06_simulated_scaling.json) hands
    # each simulated rank the reads and finds the candidates the real 2-rank run's ranks had
    r3 = subprocess.run([sys.executable, bench, "--gpus", "1", "--simulate-ranks", "2"] + FLAGS, capture_output=True, text=True, timeout=600)
    assert r3.returncode == 0, r3.stderr[-2000:]
    s = _line(r3.stdout)["P"]["2"]
    assert s["sum_of_rank_candidates"] == a["candidates"]
    assert [k["candidates"] for k in s["ranks"]] == [k["local_candidates"] for k in sorted(x["per_rank"], key=lambda k: k["rank"])]
    assert all(k["seed_ms"] > 0 and k["align_ms"] > 0 and k["index_slice_ms"] > 0 for k in s["ranks"])
    assert 0.2 < s["compute_only_speedup_bound"]["index_rebuilt_on_every_rank"] <= 2.05      # (config 1 is a few milliseconds of work: launch overheads)


@pytest.mark.parametrize("wl,tech", [("grid_tiny", 0), ("grid_tiny_ont", 1)])
def test_bench_grid_workload_equals_the_drop_in_binary(tmp_path, wl, tech):
    """bench.py's multi-volume runner (bench_grid.py: the workloads config3 / config5_cell) on a three-volume toy set: its cells hold the
    candidates and overlaps the drop-in binary writes for the same reads cut at the same volume size (MECAT_HIP_MCS), one rank and two."""
    from mecat_amd import workload as W
    name, nvols, cells, mcs = W.GRIDS[wl]
    n, L, err, G, seed, ont = W.CONFIGS[name]
    codes, lens = W.synth_reads(n, L, err, G, seed, ont)
    fa = str(tmp_path / "r.fa")
    W.write_fasta(fa, codes, lens)
    exe = os.path.join(H.ROOT, "mecat_amd", "bin", "mecat2pw")
    want = {}
    for task in (0, 1):
        out = str(tmp_path / ("o%d" % task))
        r = subprocess.run([exe, "-j", str(task), "-d", fa, "-o", out, "-w", str(tmp_path / ("w%d" % task)), "-t", "4", "-g", "1", "-x", str(tech)],
                           capture_output=True, text=True, env=dict(os.environ, MECAT_HIP_MCS=str(mcs)), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        want[task] = open(out, "rb").read().splitlines(keepends=True)
    assert len(open(str(tmp_path / "w0" / "fileindex.txt")).read().split()) == nvols
    bench = os.path.join(H.ROOT, "bench.py")
    flags = ["--workload", wl, "--steps", "1", "--warmup", "0", "--no-cpu"]
    r1 = subprocess.run([sys.executable, bench, "--gpus", "1"] + flags, capture_output=True, text=True, timeout=600,
                        env=dict(os.environ, MECAT_BENCH_PARITY="1"))
    assert r1.returncode == 0, r1.stderr[-3000:]
    a = _line(r1.stdout)
    assert a["n_gpus"] == 1 and a["value"] > 0 and a["roofline"]["achieved"] > 0 and set(a["roofline"]["phases"]) == {"index", "seed", "align"}
    assert sorted(a["cells"]) == sorted("%d,%d" % c for c in cells)
    assert a["candidates"] == len(want[0]) == sum(c["can_lines"] for c in a["cells"].values())
    # cell by cell: the binary's lines whose query / subject ids fall into the cell's volumes, sorted and hashed as the bench hashes its own
    import bisect
    import hashlib
    starts = [v["start_read_id"] for v in a["config"]["volumes"]]
    per = {}
    for ln in want[0]:
        f = ln.split(b"\t")
        key = "%d,%d" % (bisect.bisect_right(starts, int(f[1])) - 1, bisect.bisect_right(starts, int(f[0])) - 1)
        per.setdefault(key, []).append(ln)
    for key, c in a["cells"].items():
        h = hashlib.sha256()
        for ln in sorted(per.get(key, [])):
            h.update(ln)
        assert (c["can_lines"], c["can_sorted_sha256"]) == (len(per.get(key, [])), h.hexdigest()), key
    # the -j 1 side: every extension that reaches min_align_size is an overlap line before the per-read containment filter
    assert a["overlaps_ok"] >= len(want[1]) > 0
    # two ranks, every cell sharded over them (three rows over two ranks would be dealt out as rows: the cells mode is asked for)
    env = dict(os.environ, MECAT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MECAT_HIP_SHARD="cells")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29543", bench, "--gpus", "2"] + flags, capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    b = _line(r2.stdout)
    assert b["n_gpus"] == 2 and b["candidates"] == a["candidates"] and b["overlaps_ok"] == a["overlaps_ok"]
    assert b["exchange"]["transport"].startswith("host files") and b["exchange"]["calls_per_step"] >= 2 * len(cells)


def test_bench_grid_rows_mode_two_ranks_equal_one():
    """N > 1 with at least as many volumes as ranks: bench_grid deals the grid ROWS out by cost (mhip_shard_deal_rows), as the driver's rows mode
    does, and moves no data — five toy volumes over two ranks give the candidates and overlaps of the one-rank run."""
    bench = os.path.join(H.ROOT, "bench.py")
    flags = ["--workload", "grid_tiny_rows", "--steps", "1", "--warmup", "0", "--no-cpu"]
    r1 = subprocess.run([sys.executable, bench, "--gpus", "1"] + flags, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]
    a = _line(r1.stdout)
    assert len(a["config"]["volumes"]) == 5 and len(a["cells"]) == 15
    env = dict(os.environ, MECAT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29545", bench, "--gpus", "2"] + flags, capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    b = _line(r2.stdout)
    assert b["n_gpus"] == 2 and b["candidates"] == a["candidates"] and b["overlaps_ok"] == a["overlaps_ok"]
    x = b["exchange"]
    assert x["transport"].startswith("none") and sum(x["cells_of_rank"]) == 15 and max(x["cells_of_rank"]) == x["heaviest_rank_cells"] == 8
