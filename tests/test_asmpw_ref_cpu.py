"""SURVEY.md §8f row N3 (mecat2canu's mecat2asmpw / mecat2trimpw), first step: the reference pin.  tests/golden/asmpw.json and the
*.S2.sorted files hold what the UNMODIFIED tools (oracle/_ref/mecat2asmpw, mecat2trimpw: gcc on the reference's single C files)
print for a seeded set of corrected reads laid out as canu lays out its overlap blocks.  Where the reference binaries exist
(the build container) this test re-runs them and checks the committed goldens; everywhere it checks the fixtures' integrity.
The restatement and the kernels for this path are not written yet (DESIGN.md §6 lists the deltas to mecat2pw)."""
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
