"""End-to-end drop-in test on the GPU box: the mecat2pw driver binary produces the same multiset of .can / .m4 lines as
the unmodified reference (golden files generated in the build container), keeps the wrk/ resume protocol."""
import json
import os
import subprocess

import pytest

import helpers as H

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(H.GOLDEN, "golden.json")))
BIN = os.path.join(H.ROOT, "mecat_amd", "bin", "mecat2pw")


def _fasta(tmp_path, name):
    g = G["sets"][name]["gen"]
    codes, lens = H.synth_reads(g["nreads"], g["L"], g["err"], g["genome"], g["seed"], g["ont"])
    fa = str(tmp_path / (name + ".fa"))
    H.write_fasta(fa, codes, lens)
    return fa


def _run(tmp_path, fa, args, name):
    out = str(tmp_path / name)
    r = subprocess.run([BIN, "-d", fa, "-o", out, "-w", str(tmp_path / ("w_" + name)), "-t", "4"] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return sorted(open(out).read().splitlines()), r


def test_cli_can_tiny(tmp_path):
    fa = _fasta(tmp_path, "tiny")
    lines, r = _run(tmp_path, fa, ["-j", "0"], "t.can")
    assert lines == open(os.path.join(H.GOLDEN, "tiny.can.sorted")).read().splitlines()
    assert "number of kmers: %d" % G["sets"]["tiny"]["index"]["num_kmers"] in r.stdout


def test_cli_can_nanopore_mode(tmp_path):
    fa = _fasta(tmp_path, "tiny_ont")
    lines, _ = _run(tmp_path, fa, ["-j", "0", "-x", "1"], "o.can")
    assert lines == open(os.path.join(H.GOLDEN, "tiny_ont.can.sorted")).read().splitlines()


@pytest.mark.parametrize("g", [0, 1])
def test_cli_m4_tiny(tmp_path, g):
    fa = _fasta(tmp_path, "tiny")
    lines, _ = _run(tmp_path, fa, ["-j", "1", "-g", str(g)], "t%d.m4" % g)
    assert lines == open(os.path.join(H.GOLDEN, "tiny.g%d.m4.sorted" % g)).read().splitlines()


def test_cli_default_task_is_align_and_small_slabs(tmp_path):
    """-j defaults to 1 (pw_options.cpp:33); result independent of the host slab size"""
    fa = _fasta(tmp_path, "tiny")
    env = dict(os.environ, MECAT_HIP_SLAB="37")
    out = str(tmp_path / "d.m4")
    r = subprocess.run([BIN, "-d", fa, "-o", out, "-w", str(tmp_path / "wd")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted(open(out).read().splitlines()) == open(os.path.join(H.GOLDEN, "tiny.g0.m4.sorted")).read().splitlines()


def test_cli_resume_skips_finished_volume(tmp_path):
    fa = _fasta(tmp_path, "tiny")
    wrk = tmp_path / "w_resume"
    wrk.mkdir()
    (wrk / "r_0").write_text("sentinel\n")
    out = str(tmp_path / "r.can")
    r = subprocess.run([BIN, "-j", "0", "-d", fa, "-o", out, "-w", str(wrk)], capture_output=True, text=True)
    assert r.returncode == 0 and "volume 0 has been finished" in r.stderr
    assert open(out).read() == "sentinel\n"


def test_cli_multi_volume_grid(tmp_path):
    """three volumes (MECAT_HIP_MCS test knob) -> 6 grid cells (i, j >= i), r_<i> files, cat merge.  Expected output:
    the oracle run cell by cell on the same volumes (the reference's MCS is a compile-time constant, so the reference
    binary itself cannot be made to split a small input)."""
    import ctypes as C
    g = G["sets"]["tiny"]["gen"]
    codes, lens = H.synth_reads(g["nreads"], g["L"], g["err"], g["genome"], g["seed"], g["ont"])
    fa = str(tmp_path / "tiny.fa")
    H.write_fasta(fa, codes, lens)
    wrk = tmp_path / "w_mv"
    out = str(tmp_path / "mv.can")
    env = dict(os.environ, MECAT_HIP_MCS="250000")
    r = subprocess.run([BIN, "-j", "0", "-d", fa, "-o", out, "-w", str(wrk)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    names = open(os.path.join(str(wrk), "fileindex.txt")).read().split()
    assert len(names) == 3
    O = H.orc()
    vols = [O.orc_volume_load(n.encode()) for n in names]
    assert sum(v.contents.num_reads for v in vols) == len(lens)
    assert [v.contents.start_read_id for v in vols][0] == 0
    want = []
    p = H.orc_params(tech=0)
    for i, ref in enumerate(vols):
        oidx = O.orc_index_build(ref)
        ro, _ = H.vol_arrays(ref)
        for j in range(i, len(vols)):
            rd = vols[j]
            cands = H.orc_seed_all(ref, rd, oidx, p)
            qo, _ = H.vol_arrays(rd)
            want += H.can_lines_from_cands(cands, qo, ro, rd.contents.start_read_id, ref.contents.start_read_id)
        O.orc_index_free(oidx)
    got = sorted(open(out).read().splitlines())
    assert got == sorted(want)
    assert len(got) > 300
    for i in range(3):
        assert os.path.exists(os.path.join(str(wrk), "r_%d" % i))
    # the driver keeps query volumes resident across the grid rows (round 6); with the cache off (every cell loads, uploads and frees its
    # query volume, as the reference does) and with room for one volume only the lines are the same, for both tasks
    for task in ("0", "1"):
        outs = []
        for cache in (None, "0", "1"):
            o = str(tmp_path / ("c%s_%s" % (cache, task)))
            e = dict(env)
            if cache is not None:
                e["MECAT_HIP_VOLCACHE_MB"] = cache
            r = subprocess.run([BIN, "-j", task, "-g", "1", "-d", fa, "-o", o, "-w", str(tmp_path / ("wc%s_%s" % (cache, task))), "-t", "4"], capture_output=True,
                               text=True, env=e)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(sorted(open(o).read().splitlines()))
        assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 300
        if task == "0":
            assert outs[0] == got


@pytest.mark.parametrize("task", ["0", "1"])
def test_cli_two_processes_share_the_grid_rows(tmp_path, task):
    """multi-GPU mode of the driver: two processes (here both on GPU 0) deal out the rows of a 3-volume grid; the merged
    output is byte-identical to the single-process run (rows are merged in volume order)"""
    fa = _fasta(tmp_path, "tiny")
    env = dict(os.environ, MECAT_HIP_MCS="250000")
    one = str(tmp_path / "one.out")
    r = subprocess.run([BIN, "-j", task, "-d", fa, "-o", one, "-w", str(tmp_path / "w_one"), "-t", "4"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    two = str(tmp_path / "two.out")
    wrk = str(tmp_path / "w_two")
    procs = []
    for rank in (1, 0):          # rank 1 first: it has to wait for rank 0's split
        e = dict(env, MECAT_HIP_WORLD="2", MECAT_HIP_RANK=str(rank), MECAT_HIP_DEVICE="0")
        procs.append(subprocess.Popen([BIN, "-j", task, "-d", fa, "-o", two, "-w", wrk, "-t", "4"], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=e))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-2000:]
    assert open(two).read() == open(one).read()
    assert len(open(two).read().splitlines()) > 300
    assert sorted(f for f in os.listdir(wrk) if f.startswith("r_")) == ["r_0", "r_1", "r_2"]


@pytest.mark.parametrize("task,nproc,vols,ishard", [("0", 2, 1, "1"), ("1", 2, 1, "0"), ("1", 3, 3, "1"), ("0", 2, 3, "0")])
def test_cli_processes_share_the_grid_cells(tmp_path, task, nproc, vols, ishard):
    """multi-GPU mode of the driver, cell sharding: every rank works on every (reference volume, query volume)
    cell — the query reads dealt out in chunks, chunk c of query volume j to rank (c + j) mod P — the candidate lists and
    extension results are all-gathered (mhip_seed_reads_sharded / mhip_align_sharded), every rank formats and writes the lines of
    its own reads and rank 0 strings the parts together into r_<i>.  Here the ranks share GPU 0 and exchange through the host-file
    transport (RCCL refuses two ranks on one device); a one-volume input, which the row sharding cannot split, is spread over all
    ranks.  ishard = "1": the ranks also build each reference volume's look-up table together (mhip_index_build_sharded: key-range
    shards + all-gather; the driver's default from four ranks on).  The output must be the single-process run's as a multiset of lines."""
    import uuid
    fa = _fasta(tmp_path, "config1" if vols == 1 else "tiny")
    env = dict(os.environ)
    if vols > 1:
        env["MECAT_HIP_MCS"] = "250000"
    args = ["-j", task, "-g", "1"]
    one = str(tmp_path / "one.out")
    r = subprocess.run([BIN, "-d", fa, "-o", one, "-w", str(tmp_path / "w_one"), "-t", "4"] + args, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    many = str(tmp_path / "many.out")
    wrk = str(tmp_path / "w_many")
    run = uuid.uuid4().hex[:10]
    procs = []
    for rank in reversed(range(nproc)):          # rank 0 last: the others have to wait for its split
        e = dict(env, MECAT_HIP_WORLD=str(nproc), MECAT_HIP_RANK=str(rank), MECAT_HIP_DEVICE="0", MECAT_HIP_SHARD="cells",
                 MECAT_HIP_COMM="file", MECAT_HIP_RUN_ID=run, MECAT_HIP_SHARD_CHUNK="100" if vols > 1 else "500",
                 MECAT_HIP_COMM_TIMEOUT_S="60", MECAT_HIP_WAIT_S="120", MECAT_HIP_INDEX_SHARD=ishard)
        procs.append(subprocess.Popen([BIN, "-d", fa, "-o", many, "-w", wrk, "-t", "4"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=e))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-2000:]
    # every rank writes the lines of its own reads and rank 0 strings the parts together: the multiset of lines is the contract
    # (the reference's own line order depends on its thread timing)
    assert sorted(open(many).read().splitlines()) == sorted(open(one).read().splitlines())
    assert len(open(many).read().splitlines()) > (5000 if vols == 1 else 300)
    assert sorted(f for f in os.listdir(wrk) if f.startswith("r_")) == ["r_%d" % i for i in range(vols)]      # no part files left behind


def test_cli_dead_rank_does_not_hang_rank_0(tmp_path):
    """rows mode: rank 1 owns grid row 1 and never shows up; rank 0 must give up (no heartbeat) instead of polling forever"""
    import uuid
    fa = _fasta(tmp_path, "tiny")
    env = dict(os.environ, MECAT_HIP_MCS="250000", MECAT_HIP_WORLD="2", MECAT_HIP_RANK="0", MECAT_HIP_DEVICE="0", MECAT_HIP_SHARD="rows",
               MECAT_HIP_RUN_ID=uuid.uuid4().hex[:10], MECAT_HIP_WAIT_S="8")
    r = subprocess.run([BIN, "-j", "0", "-d", fa, "-o", str(tmp_path / "o"), "-w", str(tmp_path / "w"), "-t", "2"], capture_output=True, text=True,
                       env=env, timeout=120)
    assert r.returncode != 0 and "volume 1" in r.stderr, r.stderr[-1000:]


def test_cli_multi_volume_grid_m4(tmp_path):
    """-j 1 over a 3-volume grid: off-diagonal cells align reads of volume j against volume i (local ids, start_read_id
    offsets in every record).  Expected output: the oracle's whole `-j 1` body (orc_map_read) run cell by cell."""
    import ctypes as C
    g = G["sets"]["tiny"]["gen"]
    codes, lens = H.synth_reads(g["nreads"], g["L"], g["err"], g["genome"], g["seed"], g["ont"])
    fa = str(tmp_path / "tiny.fa")
    H.write_fasta(fa, codes, lens)
    wrk = tmp_path / "w_mv4"
    out = str(tmp_path / "mv.m4")
    env = dict(os.environ, MECAT_HIP_MCS="250000")
    r = subprocess.run([BIN, "-j", "1", "-g", "1", "-d", fa, "-o", out, "-w", str(wrk), "-t", "4"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    names = open(os.path.join(str(wrk), "fileindex.txt")).read().split()
    assert len(names) == 3
    O = H.orc()
    vols = [O.orc_volume_load(n.encode()) for n in names]
    p = H.orc_params(tech=0)
    want = []
    outm = (H.OrcM4 * 200)()
    buf = C.create_string_buffer(512)
    al = O.orc_aligner_new()
    for i, ref in enumerate(vols):
        oidx = O.orc_index_build(ref)
        bk = O.orc_bk_new(ref.contents.num_bases)
        for j in range(i, len(vols)):
            rd = vols[j]
            for rid in range(rd.contents.num_reads):
                k = O.orc_map_read(ref, rd, oidx, bk, al, rid, C.byref(p), outm)
                for x in range(k):
                    n = O.orc_m4_line(C.byref(outm[x]), 1, buf)
                    want.append(buf.raw[:n].decode().rstrip("\n"))
        O.orc_bk_free(bk)
        O.orc_index_free(oidx)
    O.orc_aligner_free(al)
    got = sorted(open(out).read().splitlines())
    assert got == sorted(want)
    assert len(got) > 300


def test_cli_candidate_partition_files(tmp_path):
    """MECAT_HIP_PARTITION: mecat2cns' partition files straight from the candidate arrays == the threaded
    text partitioner on the produced .can == the restatement (oracle/partition_oracle.py, pinned to the reference by
    tests/test_partition_cpu.py); a resumed run (every row already on disk) takes the text path and writes the same bytes."""
    import sys

    import numpy as np
    sys.path.insert(0, os.path.join(H.ROOT, "oracle"))
    import partition_oracle as PO
    fa = _fasta(tmp_path, "tiny")
    out = str(tmp_path / "p.can")
    wrk = str(tmp_path / "w_p")
    env = dict(os.environ, MECAT_HIP_PARTITION="7,1000", MECAT_HIP_SLAB="41")
    cmd = [BIN, "-j", "0", "-d", fa, "-o", out, "-w", wrk, "-t", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    assert sorted(text.splitlines()) == open(os.path.join(H.GOLDEN, "tiny.can.sorted")).read().splitlines()

    def parts(can):
        d, base = os.path.dirname(can), os.path.basename(can)
        return {f[len(base):]: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d)) if f.startswith(base + ".part")}
    direct = parts(out)
    assert len(direct) > 3
    ofiles, oidx = PO.partition(PO.parse_can(text), 7, 1000)
    assert sum(len(v) for v in ofiles.values()) > 100
    for k, recs in ofiles.items():
        assert np.array_equal(np.frombuffer(direct[".part%d" % k], dtype=np.int32).reshape(-1, 13), np.array(recs, dtype=np.int32).reshape(-1, 13))
    assert direct[".partition_files"].decode().splitlines() == ["%s.part%d\t%d\t%d" % (out, k, lo, hi) for k, lo, hi in oidx]
    # the standalone tool on a copy of the text
    os.mkdir(tmp_path / "t")
    can2 = str(tmp_path / "t" / "p.can")
    open(can2, "w").write(text)
    tool = os.path.join(H.ROOT, "mecat_amd", "bin", "mecat2cns_partition")
    assert subprocess.run([tool, can2, "7", "1000", "2"], capture_output=True).returncode == 0
    t = parts(can2)
    assert sorted(t) == sorted(direct)
    assert all(t[k] == direct[k] for k in t if k != ".partition_files")
    # resume: rows are skipped, the merged text is partitioned instead
    for f in list(os.listdir(tmp_path)):
        if f.startswith("p.can.part"):
            os.remove(tmp_path / f)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "has been finished" in r.stderr, r.stderr[-2000:]
    assert parts(out) == direct
    # -j 1 without -g 1 has no start points for mecat2cns
    r = subprocess.run([BIN, "-j", "1", "-d", fa, "-o", str(tmp_path / "q.m4"), "-w", str(tmp_path / "w_q")], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "MECAT_HIP_PARTITION" in r.stderr
    # -j 1 -g 1: partition_m4records (coverage filter with mecat2cns' -r 0.9 minus 0.02; tiny reads overlap end to end often enough)
    env4 = dict(os.environ, MECAT_HIP_PARTITION="9,1000,0.5", MECAT_HIP_SLAB="41")
    out4 = str(tmp_path / "p.m4")
    cmd4 = [BIN, "-j", "1", "-g", "1", "-d", fa, "-o", out4, "-w", str(tmp_path / "w_p4"), "-t", "3"]
    r = subprocess.run(cmd4, capture_output=True, text=True, env=env4)
    assert r.returncode == 0, r.stderr[-2000:]
    text4 = open(out4).read()
    d4 = parts(out4)
    ofiles, oidx = PO.partition_m4(PO.parse_m4(text4), 0.5 - 0.02, 9, 1000)
    assert sum(len(v) for v in ofiles.values()) > 50
    for k, recs in ofiles.items():
        assert np.array_equal(np.frombuffer(d4[".part%d" % k], dtype=np.int32).reshape(-1, 13), np.array(recs, dtype=np.int32).reshape(-1, 13))
    assert d4[".partition_files"].decode().splitlines() == ["%s.part%d\t%d\t%d" % (out4, k, lo, hi) for k, lo, hi in oidx]
    for f in list(os.listdir(tmp_path)):
        if f.startswith("p.m4.part"):
            os.remove(tmp_path / f)
    r = subprocess.run(cmd4, capture_output=True, text=True, env=env4)          # resumed: the text path
    assert r.returncode == 0 and "has been finished" in r.stderr, r.stderr[-2000:]
    assert parts(out4) == d4


def test_cli_cells_mode_stops_when_a_peer_fails(tmp_path):
    """cells mode: a rank that dies (here: killed) leaves a failure marker (or a heartbeat that goes stale); the surviving rank must
    stop with a message instead of waiting in the exchange for ever (watchdog of the heartbeat thread)"""
    import signal
    import time
    import uuid
    fa = _fasta(tmp_path, "config1")
    wrk = str(tmp_path / "w")
    run = uuid.uuid4().hex[:10]
    env = dict(os.environ, MECAT_HIP_WORLD="2", MECAT_HIP_DEVICE="0", MECAT_HIP_SHARD="cells", MECAT_HIP_COMM="file", MECAT_HIP_RUN_ID=run,
               MECAT_HIP_COMM_TIMEOUT_S="300", MECAT_HIP_WAIT_S="300")
    # rank 1 is started and killed while it waits for rank 0's split: its SIGTERM handler leaves the failure marker
    p1 = subprocess.Popen([BIN, "-j", "1", "-d", fa, "-o", str(tmp_path / "o"), "-w", wrk, "-t", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, env=dict(env, MECAT_HIP_RANK="1"))
    time.sleep(1.5)
    p1.send_signal(signal.SIGTERM)
    p1.communicate(timeout=60)
    t0 = time.time()
    r0 = subprocess.run([BIN, "-j", "1", "-d", fa, "-o", str(tmp_path / "o"), "-w", wrk, "-t", "2"], capture_output=True, text=True,
                        env=dict(env, MECAT_HIP_RANK="0"), timeout=200)
    assert r0.returncode != 0 and "a peer" in r0.stderr, r0.stderr[-1500:]
    assert time.time() - t0 < 120


def _part_files(can):
    """{suffix: bytes} of <can>.part* and the index file with the output path taken out of its lines"""
    d, base = os.path.dirname(can), os.path.basename(can)
    out = {}
    for f in sorted(os.listdir(d)):
        if f.startswith(base + ".part"):
            b = open(os.path.join(d, f), "rb").read()
            out[f[len(base):]] = b.replace(can.encode(), b"<out>") if f.endswith(".partition_files") else b
    return out


@pytest.mark.parametrize("task,nproc,vols,mode", [("0", 2, 1, "cells"), ("1", 2, 1, "cells"), ("0", 2, 3, "rows"), ("1", 3, 3, "cells")])
def test_cli_partition_files_of_a_multi_process_run(tmp_path, task, nproc, vols, mode):
    """SURVEY.md §8f row N4 in a multi-process run: every rank writes the partition records of its own lines as
    streams of its own (with a (row, query read) key per record), rank 0 merges the streams by key — no text is parsed — and the
    files are, byte for byte, those of the one-process run (whose record order is the order of ITS output lines: rows, then reads)."""
    import uuid
    fa = _fasta(tmp_path, "config1" if vols == 1 else "tiny")
    env = dict(os.environ, MECAT_HIP_PARTITION="40,1000,0.5" if vols == 1 else "7,1000,0.5", MECAT_HIP_SLAB="300" if vols == 1 else "41")
    if vols > 1:
        env["MECAT_HIP_MCS"] = "250000"
    args = ["-j", task, "-g", "1"]
    os.mkdir(tmp_path / "a")
    os.mkdir(tmp_path / "b")
    one = str(tmp_path / "a" / "o.out")
    r = subprocess.run([BIN, "-d", fa, "-o", one, "-w", str(tmp_path / "w_one"), "-t", "4"] + args, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    many = str(tmp_path / "b" / "o.out")
    wrk = str(tmp_path / "w_many")
    run = uuid.uuid4().hex[:10]
    procs = []
    for rank in reversed(range(nproc)):
        e = dict(env, MECAT_HIP_WORLD=str(nproc), MECAT_HIP_RANK=str(rank), MECAT_HIP_DEVICE="0", MECAT_HIP_SHARD=mode, MECAT_HIP_COMM="file",
                 MECAT_HIP_RUN_ID=run, MECAT_HIP_SHARD_CHUNK="100", MECAT_HIP_COMM_TIMEOUT_S="60", MECAT_HIP_WAIT_S="120")
        procs.append(subprocess.Popen([BIN, "-d", fa, "-o", many, "-w", wrk, "-t", "4"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=e))
    errs = []
    for p in procs:
        out, err = p.communicate(timeout=300)
        errs.append(err)
        assert p.returncode == 0, err[-2000:]
    assert "partition_files(text)" not in "".join(errs)
    a, b = _part_files(one), _part_files(many)
    assert sorted(a) == sorted(b), (sorted(a), sorted(b))          # (no rank streams, keys or meta files left behind either)
    assert sum(len(v) for k, v in a.items() if k != ".partition_files") > 52 * 200
    for k in a:
        assert a[k] == b[k], k
