"""Every index-build path the product can take, against the oracle and against the default build:

  MECAT_IDX_STRICT=1   explicit ballot ranks instead of the lane order of one LDS atomic (index_part.hip wave_rank) — the build the
                       product falls back to when ix_fill finds a bucket out of ascending order; no default run ever reaches it
  MECAT_IDX_S1_XCD=0/1 ix_scatter1's two tile orders (tile = block / per-XCD ranges); without the knob the first large build of a
                       context times both and keeps the faster one, so a default run exercises whichever won

on `tiny`, `config1`, the ragged-edge set, a repeat-structured set (buckets at and beyond the cap of 128) and one volume of more than
8 192 level-1 tiles (where the default build's timing race runs).  Reference: lookup_table.cpp:63-160 (ascending positions inside a
bucket, buckets of more than 128 dropped)."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

VARIANTS = [
    dict(MECAT_IDX_STRICT="1"),
    dict(MECAT_IDX_S1_XCD="0"),
    dict(MECAT_IDX_S1_XCD="1"),
    dict(MECAT_IDX_STRICT="1", MECAT_IDX_S1_XCD="1"),
]


class _Env:
    def __init__(self, kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _ragged():
    rng = np.random.default_rng(3)
    rep = rng.integers(0, 4, size=13).astype(np.uint8)
    reads = [rng.integers(0, 4, size=n).astype(np.uint8) for n in (1, 5, 12, 13, 14, 40, 300, 16, 17, 31, 32, 33)]
    reads += [np.concatenate([rep, rng.integers(0, 4, size=7).astype(np.uint8)]) for _ in range(140)]
    reads += [rng.integers(0, 4, size=int(n)).astype(np.uint8) for n in rng.integers(1, 200, size=50)]
    return np.concatenate(reads), np.array([len(r) for r in reads], dtype=np.int32)


def _reads(name):
    if name == "tiny":
        return H.synth_reads(200, 3000, 0.15, 30000, 11)
    if name == "config1":
        return H.synth_reads(1000, 10000, 0.15, 500000, 1)
    if name == "ragged":
        return _ragged()
    if name == "rep_ont":
        return H.rep_set("rep_ont")[:2]
    if name == "big":      # 9 100 x 15 kb = 139 Mbase: 8 500 level-1 tiles of 16 384 positions (the timing race needs >= 8 192)
        return H.synth_reads(9100, 15000, 0.15, 4_500_000, 41)
    raise KeyError(name)


@pytest.fixture(scope="module")
def ctx():
    import mecat_amd.hip as M
    c = M.Context(0)
    yield c
    c.close()


def _tables(M, ctx, gv):
    gi = M.Index(ctx, gv)
    counts, offsets = gi.download()
    slots, recs, cut = gi.download_aux()
    n = gi.num_kmers
    gi.free()
    return n, counts, offsets, slots, recs, cut


@pytest.mark.parametrize("name", ["tiny", "config1", "ragged", "rep_ont", "big"])
def test_every_build_path_gives_the_oracle_table(name, ctx):
    import mecat_amd.hip as M
    codes, lens = _reads(name)
    ov = H.orc_pack(codes, lens)
    oidx = H.orc().orc_index_build(ov)
    oi = oidx.contents
    ocounts = np.ctypeslib.as_array(oi.counts, shape=(H.NK,))
    ooffs = np.ctypeslib.as_array(oi.offsets, shape=(oi.num_kmers,))
    offs, pac = H.vol_arrays(ov)
    gv = M.Volume(ctx, pac, offs, ov.contents.num_bases, 0)
    if name == "big":
        assert (ov.contents.num_bases + 16383) // 16384 >= 8192
    for k in ("MECAT_IDX_STRICT", "MECAT_IDX_S1_XCD"):
        assert k not in os.environ
    base = _tables(M, ctx, gv)
    assert base[0] == oi.num_kmers and np.array_equal(base[1], ocounts) and np.array_equal(base[2], ooffs), "default build"
    assert np.array_equal(base[3], ((ooffs // 2000) & 0x7FFF).astype(np.uint16))
    if name in ("ragged", "rep_ont"):
        assert int(base[1].max()) == 128 or name == "ragged"      # a bucket exactly at the cap is kept
    for env in VARIANTS:
        with _Env(env):
            got = _tables(M, ctx, gv)
        assert got[0] == base[0], env
        assert np.array_equal(got[1], ocounts) and np.array_equal(got[2], ooffs), env
        assert np.array_equal(got[3], base[3]) and got[5] == base[5], env
        assert (got[4] is None) == (base[4] is None) and (got[4] is None or np.array_equal(got[4], base[4])), env
    gv.free()
    H.orc().orc_index_free(oidx)
    H.orc().orc_volume_free(ov)


def test_strict_build_feeds_the_same_candidates(ctx):
    """the seeding stage on a STRICT-built index of a repeat-structured set == the oracle's candidates"""
    import mecat_amd.hip as M
    codes, lens, ont, _ = H.rep_set("rep_pb")
    ov = H.orc_pack(codes, lens)
    oidx = H.orc().orc_index_build(ov)
    offs, pac = H.vol_arrays(ov)
    gv = M.Volume(ctx, pac, offs, ov.contents.num_bases, 0)
    with _Env(dict(MECAT_IDX_STRICT="1")):
        gi = M.Index(ctx, gv)
    p = M.default_params(ont)
    got, cnt = M.seed_reads(ctx, gi, gv, gv, 0, len(lens), p)
    want = H.orc_seed_all(ov, ov, oidx, H.orc_params(tech=ont))
    bad = [r for r, w in enumerate(want) if not (cnt[r] == len(w) and all(np.array_equal(got[r][: cnt[r]][f], w[f]) for f in H.CAND_DTYPE.names))]
    assert not bad, bad[:5]
    gi.free()
    gv.free()
