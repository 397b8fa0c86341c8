"""SURVEY.md §8f row N3 (mecat2canu's mecat2asmpw / mecat2trimpw), first step: the reference pin.  tests/golden/asmpw.json and the
*.S2.sorted files hold what the UNMODIFIED tools (oracle/_ref/mecat2asmpw, mecat2trimpw: gcc on the reference's single C files)
print for a seeded set of corrected reads laid out as canu lays out its overlap blocks.  Where the reference binaries exist
(the build container) this test re-runs them and checks the committed goldens; everywhere it checks the fixtures' integrity.
(The restatement of the candidate stage is oracle/asmpw_oracle.c, the device path mecat_amd/csrc/asm_seed.hip: tests/test_gpu_asmpw.py.)"""
import hashlib
import json
import os
import sys

import pytest

import helpers as H

META = json.load(open(os.path.join(H.GOLDEN, "asmpw.json")))


def test_committed_outputs_match_their_hashes():
    for name, m in META["outputs"].items():
        p = os.path.join(H.GOLDEN, name)
        if os.path.exists(p):
            txt = open(p).read()
            assert hashlib.sha256(txt.encode()).hexdigest() == m["sha256"] and len(txt.splitlines()) == m["lines"]
    assert META["outputs"]["mecat2asmpw.S1.sorted"]["lines"] > 5000


@pytest.mark.skipif(not os.path.exists(os.path.join(H.ROOT, "oracle", "_ref", "mecat2asmpw")), reason="reference binaries are built in the build container only")
def test_reference_tools_reproduce_the_goldens(tmp_path):
    sys.path.insert(0, H.GOLDEN)
    import make_golden_asmpw as G
    assert G.GEN == META["gen"]
    for tool in ("mecat2asmpw", "mecat2trimpw"):
        d = str(tmp_path / tool)
        os.makedirs(d)
        G.layout(d)
        lines = G.run(tool, d, 2)
        m = META["outputs"]["%s.S2.sorted" % tool]
        assert len(lines) == m["lines"] and hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest() == m["sha256"]
        assert lines == open(os.path.join(H.GOLDEN, "%s.S2.sorted" % tool)).read().splitlines()


# ---- the look-up table of one block (creat_ref_index, mecat2asmpw.c:422-512)
IDX = os.path.join(H.GOLDEN, "asmpw_index.npz")


def asmpw_index_restated(codes, lens, k=13, cap=256):
    """numpy restatement of creat_ref_index + sumvalue_x (mecat2asmpw.c:422-512, 307-314): every k-mer start inside a read, bucketed
    by its id (here in mecat2pw's A, C, G, T digit order), buckets of more than `cap` occurrences emptied, positions ascending.
    Positions are 0-based offsets into the block's text with one separator after every read (the tool stores them + 1)."""
    import numpy as np
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64) + 1)])[:-1]
    ids, pos = [], []
    off = 0
    for i, L in enumerate(lens):
        r = codes[off: off + L].astype(np.int64)
        off += L
        if L < k:
            continue
        km = np.zeros(L - k + 1, dtype=np.int64)
        for d in range(k):
            km = (km << 2) | r[d: L - k + 1 + d]
        ids.append(km)
        pos.append(starts[i] + np.arange(L - k + 1, dtype=np.int64))
    ids, pos = np.concatenate(ids), np.concatenate(pos)
    order = np.lexsort((pos, ids))
    ids, pos = ids[order], pos[order]
    uid, first, cnt = np.unique(ids, return_index=True, return_counts=True)
    keep = cnt <= cap
    mask = np.repeat(keep, cnt)
    return uid[keep], cnt[keep], pos[mask], int((~keep).sum())


def test_index_restatement_matches_the_reference_table():
    import numpy as np
    g = np.load(IDX)
    ids, cnt, pos, dropped = asmpw_index_restated(g["codes"], g["lens"])
    assert np.array_equal(ids, g["ids"]) and np.array_equal(cnt, g["counts"]) and np.array_equal(pos, g["positions"])
    assert dropped >= 5 and int((cnt > 128).sum()) >= 5        # both planted repeats do what they were planted for


@pytest.mark.skipif(not os.path.exists(os.path.join(H.ROOT, "oracle", "_ref", "libref_asmpw.so")), reason="reference harness is built in the build container only")
def test_reference_reproduces_the_index_golden():
    import numpy as np
    sys.path.insert(0, H.GOLDEN)
    import make_golden_asmpw_index as G
    g = np.load(IDX)
    codes, lens = G.reads()
    assert np.array_equal(codes, g["codes"]) and np.array_equal(lens, g["lens"])
    ids, cnt, pos = G.reference_index(codes, lens)
    pw = G.to_pw_id(ids)
    order = np.argsort(pw, kind="stable")
    assert np.array_equal(pw[order], g["ids"]) and np.array_equal(cnt[order], g["counts"])
    assert np.array_equal(np.concatenate([pos[i] - 1 for i in order]), g["positions"])


# ---- the candidate stage of pairwise_mapping (mecat2asmpw.c:580-718): the restatement (oracle/asmpw_oracle.c) against the unmodified file
def _block_text(codes, starts, lens, b, e):
    import numpy as np
    parts, st, off = [], [], 0
    for rid in range(b, e + 1):
        s = codes[starts[rid - 1]: starts[rid]]
        st.append(off)
        parts.append(bytes(b"ACGT"[c] for c in s) + b"\0")
        off += len(s) + 1
    return b"".join(parts), np.array(st, dtype=np.int32), lens[b - 1: e].astype(np.int32)


def _expected_calls(text, cand, fwd, rev):
    """the two `align` calls the reference makes for a candidate when the first block of either extension reports no alignment
    (mecat2asmpw.c:741-749, :800-809): slices of the block text and of the query strand, min(num, 500) long (num > 600: 500)"""
    q = fwd if cand.chain == b"F" else rev
    out = []
    n1 = 500 if cand.num1 > 600 else cand.num1
    a, b = cand.loc1 + 13 - 2, cand.loc2 + 13 - 1                    # seq_pr1 / seq_pr2, walked backwards
    out.append((bytes(text[a - i] for i in range(n1)), bytes(q[b - i] for i in range(n1)), int(0.10 * n1)))
    n2 = 500 if cand.num2 > 600 else cand.num2
    a, b = cand.loc1 - 1, cand.loc2
    out.append((bytes(text[a: a + n2]), bytes(q[b: b + n2]), int(0.10 * n2)))
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(H.ROOT, "oracle", "_ref", "libref_asmpw_cand.so")), reason="the reference harness is built in the build container only")
@pytest.mark.parametrize("start,fresh", [(1, 0), (2, 0), (1, 1), (2, 1)])
def test_candidate_stage_equals_reference(start, fresh):
    """Every candidate of every query read of the golden set — subject position, query position, strand, num1, num2, in list order —
    as the UNMODIFIED pairwise_mapping selects them (observed through its `align` calls, ref_harness_asmpw_cand.c) and as
    oracle/asmpw_oracle.c restates them.  start = the indexed block (-S); the queries are the reads of that block and of the later ones.
    fresh = 1: the restatement starts every read from an all-zero segment array (asm_block_fresh: the deterministic definition the
    device path implements) — the reference, whose worker threads keep the stale seeds of their earlier reads, selects the same
    candidates on these sets."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, H.GOLDEN)
    import make_golden_asmpw as G
    codes, lens = H.synth_reads(G.GEN["nreads"], G.GEN["L"], G.GEN["err"], G.GEN["genome"], G.GEN["seed"], G.GEN["ont"])
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    b, e = G.BLOCKS[start - 1]
    text, st, ln = _block_text(codes, starts, lens, b, e)
    R = C.CDLL(os.path.join(H.ROOT, "oracle", "_ref", "libref_asmpw_cand.so"))
    R.refasmc_setup.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    R.refasmc_candidates.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
    tbuf = C.create_string_buffer(text, len(text))
    R.refasmc_setup(tbuf, len(text), st.ctypes.data, ln.ctypes.data, e - b + 1, b)

    class Cand(C.Structure):
        _fields_ = [(n, C.c_int) for n in ("loc1", "loc2", "left1", "left2", "right1", "right2", "score", "num1", "num2", "readno", "readstart")] + [("chain", C.c_char)]
    O = C.CDLL(os.path.join(H.ROOT, "oracle", "liboracle.so"))
    O.asm_block_new.restype = C.c_void_p
    O.asm_block_new.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    O.asm_candidates.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    O.asm_block_free.argtypes = [C.c_void_p]
    B = O.asm_block_new(tbuf, len(text), st.ctypes.data, e - b + 1, b)
    O.asm_block_fresh.argtypes = [C.c_void_p, C.c_int]
    O.asm_block_fresh(B, fresh)
    rec = np.zeros(8_000_000, dtype=np.uint8)
    out = (Cand * 100)()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    total = reordered = 0
    for rid in range(b, G.GEN["nreads"] + 1):
        fwd = bytes(b"ACGT"[c] for c in codes[starts[rid - 1]: starts[rid]])
        rev = fwd[::-1].translate(comp)
        used = C.c_long()
        ncalls = R.refasmc_candidates(C.create_string_buffer(fwd), rid, rec.ctypes.data, len(rec), C.byref(used))
        assert ncalls >= 0 and ncalls % 2 == 0
        n = O.asm_candidates(B, fwd, len(fwd), rid, out)
        assert 2 * n == ncalls, (rid, n, ncalls)
        raw, at = rec[: used.value].tobytes(), 0
        got, want = [], []
        for i in range(n):
            pair = []
            for _ in range(2):
                ql, tl, bd = np.frombuffer(raw[at: at + 12], dtype=np.int32)
                # (the reference passes the subject slice as `query_seq` and the query read's slice as `target_seq`)
                pair.append((raw[at + 12: at + 12 + ql], raw[at + 12 + ql: at + 12 + ql + tl], int(bd)))
                at += 12 + ql + tl
            got.append(tuple(pair))
            want.append(tuple(_expected_calls(text, out[i], fwd, rev)))
        if fresh:      # the stale seeds of earlier reads can move a candidate's score by a few votes, i.e. change the list ORDER: same candidates
            assert sorted(got) == sorted(want), rid
            reordered += got != want
        else:
            assert got == want, rid
        total += n
    O.asm_block_free(B)
    assert total > 2000      # (7 000+ candidates with block 1 indexed, 2 400+ with block 2)
    assert reordered <= 3    # (reads whose list order differs between the reference's thread history and a fresh start: 0 and 1 on these sets)
