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
