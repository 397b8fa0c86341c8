"""The device side of the accept stage's strings (mecat_amd/csrc/cns_strings.hip):

  * cns_push_gaps against the UNMODIFIED normalize_gaps(push = true) of the reference (reads_correction_aux.cpp:3-81) on 700 adversarial
    pairs of gapped strings (tests/golden/pushgaps.npz, make_golden_pushgaps.py: long gap runs, runs to the end of the string, adjacent
    query / template gaps, homopolymers), laid out back to back as the accept stage lays its strings out — so that every alignment of
    the two strings to the kernel's 32-byte blocks occurs, and neighbours share blocks
  * the whole path (mhip_cns_accept_templates) is pinned to the reference's accepted strings in test_gpu_cns_accept.py; here its
    slicing: one slice == many slices == a result buffer that has to grow, byte for byte."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import mecat_amd.hip as M
    c = M.Context(0)
    yield c
    c.close()


def test_push_gaps_kernel_equals_the_reference(ctx):
    import ctypes as C
    import mecat_amd.hip as M
    g = np.load(os.path.join(H.GOLDEN, "pushgaps.npz"))
    lens = g["lens"]
    starts = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    for lead in (0, 1, 7, 13):             # shifts every string against the 16- and 32-byte grids
        off = np.zeros(len(lens), dtype=np.int64)
        pos = lead
        for p, n in enumerate(lens):
            off[p] = pos
            pos += 2 * (int(n) + 1)
        buf = np.zeros(pos + 8, dtype=np.uint8)
        want = np.zeros(pos + 8, dtype=np.uint8)
        for p, n in enumerate(lens):
            n = int(n)
            o = int(off[p])
            buf[o: o + n] = g["qin"][starts[p]: starts[p + 1]]
            buf[o + n + 1: o + 2 * n + 1] = g["tin"][starts[p]: starts[p + 1]]
            want[o: o + n] = g["qout"][starts[p]: starts[p + 1]]
            want[o + n + 1: o + 2 * n + 1] = g["tout"][starts[p]: starts[p + 1]]
        lib = M.lib()
        lib.mhip_debug_push_gaps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        ln = np.ascontiguousarray(lens, dtype=np.int32)
        assert lib.mhip_debug_push_gaps(ctx.h, buf.ctypes.data, len(buf), off.ctypes.data, ln.ctypes.data, len(lens)) == 0, lib.mhip_last_error()
        bad = [p for p in range(len(lens)) if not np.array_equal(buf[off[p]: off[p] + 2 * (lens[p] + 1)], want[off[p]: off[p] + 2 * (lens[p] + 1)])]
        assert not bad, (lead, bad[:5])
        assert np.array_equal(buf, want)      # (nothing outside the strings was touched)


def test_slices_and_a_growing_result_buffer_change_nothing(ctx, monkeypatch):
    import mecat_amd.hip as M
    from mecat_amd import workload as W
    n = 1500
    codes, lens = W.synth_reads(n, 8000, 0.15, n * 8000 // 30, 71, 0)
    pac, offs, nb = W.pack_volume(codes, lens)
    vol = M.Volume(ctx, pac, offs, nb, 0)
    idx = M.Index(ctx, vol)
    p = M.default_params(0)
    cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, n, p)
    idx.free()
    ec = W.ext_candidates_from_table(cands, cnt, lens)
    rec, tb, ids = W.cns_templates(ec, n)
    assert len(ids) > 500

    def run():
        r = rec.copy()
        acc, strs, nj = M.cns_accept_templates(ctx, vol, pac, r, tb, 0, 2000, 0.9, threads=8)
        return acc.copy(), bytes(strs), nj

    want = run()
    assert len(want[0]) > 2000 and len(want[1]) > (64 << 20)
    for env in (dict(MECAT_CNS_SLICE_JOBS="3000"), dict(MECAT_CNS_SLICE_JOBS="7000", MECAT_CNS_STR_ESTIMATE="30"), dict(MECAT_CNS_SLICE_JOBS="1")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = run()
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(got[0], want[0]), env
        assert got[1] == want[1], env
        assert got[2] == want[2]
    vol.free()
