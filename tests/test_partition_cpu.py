"""SURVEY.md §8f row N4 — mecat2cns' candidate partition files written by mecat_amd/host/partition.cpp (standalone tool
mecat2cns_partition; the mecat2pw driver feeds the same writer, tests/test_gpu_cli.py) against the compiled reference
(oracle/_ref/libref_part.so, this container and the GPU box) and the restatement oracle/partition_oracle.py."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

sys.path.insert(0, os.path.join(H.ROOT, "oracle"))
import partition_oracle as PO  # noqa: E402

TOOL = os.path.join(H.ROOT, "mecat_amd", "bin", "mecat2cns_partition")
REFLIB = os.path.join(H.ROOT, "oracle", "_ref", "libref_part.so")


def _tool():
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-C", H.ROOT, "host"], check=True, capture_output=True)
    return TOOL


def _random_can(rng, n, nreads, min_size):
    lines = []
    for _ in range(n):
        qid, sid = int(rng.integers(0, nreads)), int(rng.integers(0, nreads))
        qs = int(rng.integers(min_size - 300, min_size + 5000))
        ss = int(rng.integers(min_size - 300, min_size + 5000))
        lines.append("%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d" % (qid, sid, int(rng.integers(0, 2)), int(rng.integers(0, 2)),
                                                             int(rng.integers(0, qs)), int(rng.integers(0, ss)), int(rng.integers(1, 90)), qs, ss))
    return "\n".join(lines) + ("\n" if lines else "")


def _read_parts(can):
    """-> ({k: int32 array (n, 13)}, index lines with the directory stripped)"""
    d, base = os.path.dirname(can), os.path.basename(can)
    files = {}
    for f in os.listdir(d):
        if f.startswith(base + ".part") and not f.endswith("partition_files"):
            a = np.fromfile(os.path.join(d, f), dtype=np.int32)
            assert a.size % 13 == 0, f
            files[int(f[len(base) + 5:])] = a.reshape(-1, 13)
    idx = []
    for ln in open(can + ".partition_files").read().splitlines():
        name, lo, hi = ln.split("\t")
        assert os.path.dirname(name) == d
        idx.append((os.path.basename(name), int(lo), int(hi)))
    return files, idx


def _check_against_oracle(can, text, batch, min_size):
    files, idx = _read_parts(can)
    ofiles, oidx = PO.partition(PO.parse_can(text), batch, min_size)
    assert sorted(files) == sorted(ofiles)
    for k in ofiles:
        want = np.array(ofiles[k], dtype=np.int32).reshape(-1, 13)
        assert files[k].shape == want.shape, k
        assert np.array_equal(files[k][:, PO.DEFINED], want[:, PO.DEFINED]), k
    assert idx == [(os.path.basename(can) + ".part%d" % k, lo, hi) for k, lo, hi in oidx]
    return files, idx


@pytest.mark.parametrize("n,nreads,batch,threads", [(0, 10, 5, 1), (1, 10, 5, 2), (5000, 2300, 500, 4), (20000, 977, 100, 7)])
def test_tool_equals_oracle(tmp_path, n, nreads, batch, threads):
    rng = np.random.default_rng(n + batch)
    text = _random_can(rng, n, nreads, 2000)
    can = str(tmp_path / "x.can")
    open(can, "w").write(text)
    r = subprocess.run([_tool(), can, str(batch), "2000", str(threads)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    files, _ = _check_against_oracle(can, text, batch, 2000)
    if n:
        assert not np.any(np.concatenate(list(files.values()))[:, [4, 5, 10, 11]])      # the undefined ints are written as 0


def test_same_batch_line_writes_query_side_first(tmp_path):
    text = "3\t4\t0\t1\t10\t20\t7\t9000\t9001\n"
    can = str(tmp_path / "y.can")
    open(can, "w").write(text)
    assert subprocess.run([_tool(), can, "100", "5000"], capture_output=True).returncode == 0
    files, idx = _check_against_oracle(can, text, 100, 5000)
    # seen from the query's side the template is read 3 (forward already); as it stands the template 4 is reversed -> both strands flip
    assert files[0].tolist() == [[1, 4, 20, 9001, 0, 0, 0, 3, 10, 9000, 0, 0, 7], [1, 3, 10, 9000, 0, 0, 0, 4, 20, 9001, 0, 0, 7]]
    assert idx == [("y.can.part0", 3, 4)]


def test_malformed_line_is_an_error(tmp_path):
    can = str(tmp_path / "z.can")
    open(can, "w").write("1\t2\t0\t0\t5\t6\t7\t9000\n")
    r = subprocess.run([_tool(), can, "100", "5000"], capture_output=True, text=True)
    assert r.returncode != 0 and "malformed" in r.stderr


@pytest.mark.skipif(not os.path.exists(REFLIB), reason="oracle/_ref/libref_part.so not built (needs /root/reference)")
@pytest.mark.parametrize("num_files", [3, 64])
def test_tool_and_oracle_equal_reference(tmp_path, num_files):
    """the compiled, unmodified partition_candidates: same files, same record order, same index (num_files = 3 makes it take
    several passes over the text; the bytes do not depend on that)"""
    rng = np.random.default_rng(11)
    text = _random_can(rng, 30000, 1500, 2000)
    for sub in ("ref", "ours"):
        os.mkdir(tmp_path / sub)
        open(tmp_path / sub / "c.can", "w").write(text)
    code = ("import ctypes as C; L = C.CDLL(%r); L.refp_partition_candidates.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_int]; "
            "L.refp_partition_candidates(%r, 200, 2000, %d)" % (REFLIB, str(tmp_path / "ref" / "c.can").encode(), num_files))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([_tool(), str(tmp_path / "ours" / "c.can"), "200", "2000", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rf, ri = _read_parts(str(tmp_path / "ref" / "c.can"))
    of, oi = _check_against_oracle(str(tmp_path / "ours" / "c.can"), text, 200, 2000)
    assert ri == oi and sorted(rf) == sorted(of)
    for k in rf:
        assert rf[k].shape == of[k].shape
        assert np.array_equal(rf[k][:, PO.DEFINED], of[k][:, PO.DEFINED]), k
