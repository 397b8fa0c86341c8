"""SURVEY.md §8f row N3, first device piece: the look-up table of mecat2canu's overlappers (creat_ref_index, mecat2asmpw.c:422-512) is
mecat2pw's table with a bucket cap of 256 instead of 128 (sumvalue_x, :307-314) — mhip_index_build_ex(.., 256, ..) against the
table the UNMODIFIED reference function built for the same reads (tests/golden/asmpw_index.npz, tests/golden/make_golden_asmpw_index.py)."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def test_index_with_cap_256_equals_the_reference_table():
    import mecat_amd.hip as M
    from mecat_amd import workload as W
    g = np.load(os.path.join(H.GOLDEN, "asmpw_index.npz"))
    pac, offs, nb = W.pack_volume(g["codes"], g["lens"])
    ctx = M.Context(0)
    vol = M.Volume(ctx, pac, offs, nb, 0)
    idx = M.Index(ctx, vol, max_bucket=256)
    counts, offsets = idx.download()
    want = np.zeros(1 << 26, dtype=np.int32)
    want[g["ids"]] = g["counts"]
    assert np.array_equal(counts, want)
    # bucket contents: ids ascending == the order of offsets[]
    assert np.array_equal(offsets, g["positions"])
    assert int((g["counts"] > 128).sum()) >= 5
    # the default cap drops those buckets and nothing else
    idx128 = M.Index(ctx, vol)
    c128, o128 = idx128.download()
    want128 = np.where(want > 128, 0, want)
    assert np.array_equal(c128, want128)
    assert np.array_equal(o128, g["positions"][np.repeat(g["counts"] <= 128, g["counts"])])
    with pytest.raises(M.MhipError):
        M.Index(ctx, vol, max_bucket=257)
    idx.free(); idx128.free(); vol.free(); ctx.close()


@pytest.mark.parametrize("start", [1, 2])
def test_candidate_stage_equals_the_restatement(start):
    """mhip_asm_seed_reads (asm_seed.hip) against oracle/asmpw_oracle.c in its fresh-per-read mode — which tests/test_asmpw_ref_cpu.py
    pins to the UNMODIFIED pairwise_mapping of mecat2asmpw.c (same candidates; the reference's thread history can move a score by a few
    votes) — on the golden set of corrected reads laid out as canu's two overlap blocks: every field of every candidate, in list order.
    start = the indexed block (-S); queries = the reads of that block and of the later ones, as the tool maps them."""
    import ctypes as C
    import sys
    import mecat_amd.hip as M
    from mecat_amd import workload as W
    sys.path.insert(0, H.GOLDEN)
    import make_golden_asmpw as G
    codes, lens = H.synth_reads(G.GEN["nreads"], G.GEN["L"], G.GEN["err"], G.GEN["genome"], G.GEN["seed"], G.GEN["ont"])
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    b, e = G.BLOCKS[start - 1]
    # the device side: block and query reads as volumes (read numbers from 1, as canu numbers them)
    ctx = M.Context(0)
    bpac, boffs, bnb = W.pack_volume(codes[starts[b - 1]: starts[e]], lens[b - 1: e])
    block = M.Volume(ctx, bpac, boffs, bnb, b)
    idx = M.Index(ctx, block, max_bucket=256)
    qpac, qoffs, qnb = W.pack_volume(codes[starts[b - 1]:], lens[b - 1:])
    reads = M.Volume(ctx, qpac, qoffs, qnb, b)
    nq = G.GEN["nreads"] - b + 1
    got, cnt = M.asm_seed_reads(ctx, idx, block, reads, 0, nq)
    # the restatement
    parts, st, off = [], [], 0
    for rid in range(b, e + 1):
        s = codes[starts[rid - 1]: starts[rid]]
        st.append(off)
        parts.append(bytes(b"ACGT"[c] for c in s) + b"\0")
        off += len(s) + 1
    text = b"".join(parts)
    st = np.array(st, dtype=np.int32)
    assert np.array_equal(st, np.asarray(boffs).reshape(-1, 2)[:, 0])      # one pad base after a read == the tool's one NUL: same offsets

    class Cand(C.Structure):
        _fields_ = [(n, C.c_int) for n in ("loc1", "loc2", "left1", "left2", "right1", "right2", "score", "num1", "num2", "readno", "readstart")] + [("chain", C.c_char)]
    O = C.CDLL(os.path.join(H.ROOT, "oracle", "liboracle.so"))
    O.asm_block_new.restype = C.c_void_p
    O.asm_block_new.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    O.asm_candidates.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    O.asm_block_fresh.argtypes = [C.c_void_p, C.c_int]
    O.asm_block_free.argtypes = [C.c_void_p]
    tbuf = C.create_string_buffer(text, len(text))
    B = O.asm_block_new(tbuf, len(text), st.ctypes.data, e - b + 1, b)
    O.asm_block_fresh(B, 1)
    out = (Cand * 100)()
    total = 0
    names = ("loc1", "loc2", "left1", "left2", "right1", "right2", "score", "num1", "num2", "readno", "readstart")
    for q in range(nq):
        rid = b + q
        fwd = bytes(b"ACGT"[c] for c in codes[starts[rid - 1]: starts[rid]])
        n = O.asm_candidates(B, fwd, len(fwd), rid, out)
        assert cnt[q] == n, (rid, cnt[q], n)
        for i in range(n):
            want = tuple(getattr(out[i], f) for f in names) + (0 if out[i].chain == b"F" else 1,)
            assert tuple(int(got[q, i][f]) for f in names + ("chain",)) == want, (rid, i)
        total += n
    O.asm_block_free(B)
    assert total > 2000
    idx.free(); block.free(); reads.free(); ctx.close()


def test_reads_that_outgrow_their_record_pool_are_redone_with_full_pools(capfd):
    """asm_seed hands a wave's segment records out of a pool of ASM_POOL records; a read that needs more gives up and is redone by a second
    launch whose pools hold one record per segment.  With pools of 96 records most reads of the golden set take that way: same candidates,
    field for field, as with the default pools (themselves pinned to the restatement above), and the same again on a context whose buffers
    were laid out for the other pool size before."""
    import sys
    import mecat_amd.hip as M
    from mecat_amd import workload as W
    sys.path.insert(0, H.GOLDEN)
    import make_golden_asmpw as G
    codes, lens = H.synth_reads(G.GEN["nreads"], G.GEN["L"], G.GEN["err"], G.GEN["genome"], G.GEN["seed"], G.GEN["ont"])
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    b, e = G.BLOCKS[0]
    ctx = M.Context(0)
    bpac, boffs, bnb = W.pack_volume(codes[starts[b - 1]: starts[e]], lens[b - 1: e])
    block = M.Volume(ctx, bpac, boffs, bnb, b)
    idx = M.Index(ctx, block, max_bucket=256)
    qpac, qoffs, qnb = W.pack_volume(codes[starts[b - 1]:], lens[b - 1:])
    reads = M.Volume(ctx, qpac, qoffs, qnb, b)
    nq = G.GEN["nreads"] - b + 1
    want, wcnt = M.asm_seed_reads(ctx, idx, block, reads, 0, nq)
    assert int(wcnt.sum()) > 2000
    capfd.readouterr()
    os.environ["MECAT_ASM_POOL"] = "96"
    os.environ["MECAT_ASM_STATS"] = "1"
    try:
        got, cnt = M.asm_seed_reads(ctx, idx, block, reads, 0, nq)
        err = capfd.readouterr().err
    finally:
        del os.environ["MECAT_ASM_POOL"], os.environ["MECAT_ASM_STATS"]
    given_up = int(err.split("reads given up")[1].split()[0])
    assert given_up > nq // 4, err
    assert np.array_equal(cnt, wcnt)
    mask = np.arange(want.shape[1])[None, :] < wcnt[:, None]
    assert got[mask].tobytes() == want[mask].tobytes()
    again, acnt = M.asm_seed_reads(ctx, idx, block, reads, 0, nq)          # back to the default layout on the same buffers
    assert np.array_equal(acnt, wcnt) and again[mask].tobytes() == want[mask].tobytes()
    idx.free(); block.free(); reads.free(); ctx.close()


@pytest.mark.parametrize("tool,start", [("mecat2asmpw", 1), ("mecat2asmpw", 2), ("mecat2trimpw", 1), ("mecat2trimpw", 2)])
def test_drop_in_tool_equals_the_reference_output(tmp_path, tool, start):
    """mecat_amd/bin/mecat2asmpw / mecat2trimpw through the tools' own command line (-P<dir> -T<n> -S<start> -E<last>, reading ovlprep and
    %06d.fasta as canu lays them out) against the sorted output of the UNMODIFIED tools on the same blocks (tests/golden/*.S<start>.sorted,
    tests/golden/make_golden_asmpw.py): index (cap 256), seeding, candidates and the O(ND) extension on the device, string_check,
    coordinates, jscore and the 12-field lines on the host."""
    import hashlib
    import json
    import subprocess
    import sys
    sys.path.insert(0, H.GOLDEN)
    import make_golden_asmpw as G
    meta = json.load(open(os.path.join(H.GOLDEN, "asmpw.json")))
    d = str(tmp_path)
    G.layout(d)
    exe = os.path.join(H.ROOT, "mecat_amd", "bin", tool)
    r = subprocess.run([exe, "-P" + d, "-T3", "-S%d" % start, "-E%d" % len(G.BLOCKS)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = []
    for t in range(3):
        lines += open(os.path.join(d, "%d_%d.r" % (start, t))).read().splitlines()
    lines.sort()
    m = meta["outputs"]["%s.S%d.sorted" % (tool, start)]
    golden = os.path.join(H.GOLDEN, "%s.S%d.sorted" % (tool, start))
    if os.path.exists(golden):
        want = open(golden).read().splitlines()
        assert len(lines) == len(want)
        bad = [(a, b) for a, b in zip(lines, want) if a != b]
        assert not bad, bad[:3]
    assert len(lines) == m["lines"]
    assert hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest() == m["sha256"]


@pytest.mark.parametrize("tool,start", [("mecat2asmpw", 1), ("mecat2trimpw", 2)])
def test_drop_in_tool_equals_the_reference_run_on_this_machine(tmp_path, tool, start):
    """Beyond the golden set: 4 000 corrected reads (32 Mbases, 32x of a 1 Mb genome) in two blocks, the UNMODIFIED tool (oracle/_ref/<tool>,
    built in the container from /root/reference/mecat2canu/src/mecat2asmpw by oracle/Makefile; the binary travels with the repo) and the
    drop-in run one after the other on this machine: sorted outputs equal line by line (~118 000 / ~40 000 lines)."""
    from mecat_amd import workload as W
    ref = os.path.join(H.ROOT, "oracle", "_ref", tool)
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/%s is not built (make -C oracle ref, in the container)" % tool)
    d = str(tmp_path)
    blocks, bases = W.asm_blocks_layout(d, 4000, 8000, 1_000_000, 2, 77)
    want, ref_s, _ = W.asm_tool_run(ref, d, 32, start, 2)
    got, dev_s, _ = W.asm_tool_run(os.path.join(H.ROOT, "mecat_amd", "bin", tool), d, 32, start, 2)
    assert len(want) > 30000
    assert len(got) == len(want)
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert not bad, bad[:3]
    print("%s -S%d: %d lines, reference %.1f s on 32 threads, device tool %.1f s" % (tool, start, len(want), ref_s, dev_s))


@pytest.mark.parametrize("tool,start", [("mecat2asmpw", 1), ("mecat2trimpw", 1), ("mecat2asmpw", 2)])
def test_reads_with_n_equal_the_reference_run_on_this_machine(tmp_path, tool, start):
    """Bases other than A, C, G, T: the tools restart their k-mer at such a base in table and query and compare it as a character in the
    extension (mecat2asmpw.c:445, 486, 316-335).  2 000 corrected reads in two blocks, every fifth read with Ns (single, three, a run of five,
    at both ends), the UNMODIFIED tool and the drop-in side by side on this machine: sorted outputs equal line by line; and the Ns matter
    (the same reads without them give other lines)."""
    from mecat_amd import workload as W
    ref = os.path.join(H.ROOT, "oracle", "_ref", tool)
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/%s is not built (make -C oracle ref, in the container)" % tool)
    d = str(tmp_path / "n")
    os.makedirs(d)
    W.asm_blocks_layout(d, 2000, 8000, 500_000, 2, 78, n_every=5)
    want, _, _ = W.asm_tool_run(ref, d, 16, start, 2)
    got, _, _ = W.asm_tool_run(os.path.join(H.ROOT, "mecat_amd", "bin", tool), d, 16, start, 2)
    assert len(want) > 10000
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert len(got) == len(want) and not bad, (len(got), len(want), bad[:3])
    d0 = str(tmp_path / "plain")
    os.makedirs(d0)
    W.asm_blocks_layout(d0, 2000, 8000, 500_000, 2, 78)
    plain, _, _ = W.asm_tool_run(ref, d0, 16, start, 2)
    assert plain != want


@pytest.mark.parametrize("tool,start", [("mecat2asmpw", 1), ("mecat2trimpw", 2)])
def test_reads_with_iupac_codes_equal_the_reference_run_on_this_machine(tmp_path, tool, start):
    """Every character outside A, C, G, T — not only N: atcttrans gives them all 4 (mecat2asmpw.c:296-304: no k-mer over one), the
    extension compares them as the characters they are (a letter equals itself and nothing else), the mapped strand keeps them
    (:583-590), lower case is upper-cased (:398, 992).  2 000 corrected reads in two blocks with R / Y / K / m behind fixed 7-mers of
    every read (so overlapping reads of one strand agree on them) and S, W, B, D, H, V, n sprinkled in; the UNMODIFIED tool and the
    drop-in side by side on this machine: sorted outputs equal line by line, and different from the run on the plain reads."""
    from mecat_amd import workload as W
    ref = os.path.join(H.ROOT, "oracle", "_ref", tool)
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/%s is not built (make -C oracle ref, in the container)" % tool)
    d = str(tmp_path / "iupac")
    os.makedirs(d)
    W.asm_blocks_layout(d, 2000, 8000, 500_000, 2, 79, iupac=True)
    letters = set(open(os.path.join(d, "000001.fasta"), "rb").read()) - set(b">0123456789\nACGT")
    assert len(letters) >= 8, letters
    want, _, _ = W.asm_tool_run(ref, d, 16, start, 2)
    got, _, _ = W.asm_tool_run(os.path.join(H.ROOT, "mecat_amd", "bin", tool), d, 16, start, 2)
    assert len(want) > 10000
    bad = [(a, b) for a, b in zip(got, want) if a != b]
    assert len(got) == len(want) and not bad, (len(got), len(want), bad[:3])
    d0 = str(tmp_path / "plain")
    os.makedirs(d0)
    W.asm_blocks_layout(d0, 2000, 8000, 500_000, 2, 79)
    plain, _, _ = W.asm_tool_run(ref, d0, 16, start, 2)
    assert plain != want


@pytest.mark.parametrize("tool,start", [("mecat2asmpw50", 1), ("mecat2asmpw50", 2), ("mecat2trimpw50", 1), ("mecat2trimpw50", 2)])
def test_50_candidate_variants_equal_the_reference_output(tmp_path, tool, start):
    """the `*50` names (MAXC 50, mecat2asmpw50.c:23) on a set dense enough that the top-MAXC cut decides which candidates survive (the
    100-candidate tool writes 42 423 lines where this one writes 25 975; tests/golden/make_golden_asmpw50.py): the sorted output of the
    drop-in equals the UNMODIFIED tool's, which on this set does not depend on its thread count (-T1 / -T2 / -T3 agree line for line),
    i.e. the device's all-zero start state per read selects the same candidates as the reference's thread history."""
    import hashlib
    import json
    import subprocess
    import sys
    sys.path.insert(0, H.GOLDEN)
    import make_golden_asmpw50 as G
    meta = json.load(open(os.path.join(H.GOLDEN, "asmpw50.json")))
    m = meta["outputs"]["%s.S%d.sorted" % (tool, start)]
    assert m["lines"] == m["lines_T2"] == m["lines_T3"] and not m["thread_dependent_lines"]
    d = str(tmp_path)
    G.layout(d)
    exe = os.path.join(H.ROOT, "mecat_amd", "bin", tool)
    r = subprocess.run([exe, "-P" + d, "-T3", "-S%d" % start, "-E%d" % len(G.BLOCKS)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = []
    for t in range(3):
        lines += open(os.path.join(d, "%d_%d.r" % (start, t))).read().splitlines()
    lines.sort()
    golden = os.path.join(H.GOLDEN, "%s.S%d.sorted" % (tool, start))
    if os.path.exists(golden):
        want = open(golden).read().splitlines()
        bad = [(a, b) for a, b in zip(lines, want) if a != b]
        assert len(lines) == len(want) and not bad, (len(lines), len(want), bad[:3])
    assert len(lines) == m["lines"]
    assert hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest() == m["sha256"]
