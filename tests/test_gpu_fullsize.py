"""Parity at BASELINE.json's FULL size (configs[1]: 100 000 reads x 15 kb @ 15 % error, 1.58 Gbase), where the oracle is
too slow to run: size-independent properties plus bit-for-bit agreement of independent implementations of each stage

  * index:      binned build (LDS partition)  ==  direct atomic build           (MECAT_IDX_BUILD=1)
  * candidates: relevance-filtered pipeline   ==  all-hits pipeline             (MECAT_SEED_FILTER=0)
                contiguous seeding            ==  two strided halves merged     (the multi-GPU shard)
  * extension:  two units per wave (dw_extend2) == one unit per wave (dw_extend) (MECAT_DW_KERNEL=1)

The right-hand variants are the ones checked against the oracle / golden vectors at small sizes (test_gpu_parity.py,
test_gpu_align.py); the left-hand ones are the defaults measured by bench.py.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_SAMPLE = 200_000


class _Env:
    def __init__(self, **kw):
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


@pytest.fixture(scope="module")
def full():
    import mecat_amd.hip as M
    from mecat_amd import workload as W
    n, L, err, G, seed, ont = W.CONFIGS["config2"]
    if os.environ.get("MECAT_FULLSIZE_READS"):       # e.g. 142000: one volume right below the 2.14 Gbase volume limit
        n = int(os.environ["MECAT_FULLSIZE_READS"])
        G = int(G * n / 100000)
    codes, lens = W.synth_reads(n, L, err, G, seed, ont)
    pac, offs, nb = W.pack_volume(codes, lens)
    ctx = M.Context(0)
    vol = M.Volume(ctx, pac, offs, nb, 0)
    d = dict(M=M, W=W, ctx=ctx, vol=vol, codes=codes, lens=lens, offs=offs, nb=nb, p=M.default_params(0))
    yield d
    vol.free()
    ctx.close()


def test_index_full_size(full):
    M, ctx, vol, codes, lens, offs = full["M"], full["ctx"], full["vol"], full["codes"], full["lens"], full["offs"]
    idx = M.Index(ctx, vol)
    counts, offsets = idx.download()
    full["idx"] = idx
    # properties
    assert counts.max() <= 128 and counts.min() >= 0
    assert int(counts.sum(dtype=np.int64)) == idx.num_kmers == len(offsets)
    starts = np.zeros(len(counts) + 1, dtype=np.int64)
    np.cumsum(counts, out=starts[1:])
    inner = np.ones(len(offsets), dtype=bool)
    inner[starts[:-1][counts > 0]] = False                      # first entry of every bucket
    assert np.all(np.diff(offsets)[inner[1:]] > 0), "bucket contents must ascend"
    assert offsets.min() >= 0 and offsets.max() < full["nb"]
    # every sampled entry really is an occurrence of its bucket's k-mer and lies inside one read
    rng = np.random.default_rng(1)
    pick = np.sort(rng.integers(0, len(offsets), size=N_SAMPLE))
    bucket = np.searchsorted(starts, pick, side="right") - 1
    pos = offsets[pick].astype(np.int64)
    rid = np.searchsorted(offs[:, 0], pos, side="right") - 1
    assert np.all(pos + 13 <= offs[rid, 0].astype(np.int64) + offs[rid, 1])
    read_start_in_codes = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])[rid]
    cpos = read_start_in_codes + (pos - offs[rid, 0])
    kmer = np.zeros(len(pick), dtype=np.int64)
    for j in range(13):
        kmer = (kmer << 2) | codes[cpos + j]
    assert np.array_equal(kmer, bucket)
    # independent implementation: direct atomic build
    with _Env(MECAT_IDX_BUILD="1"):
        idx2 = M.Index(ctx, vol)
    c2, o2 = idx2.download()
    assert np.array_equal(counts, c2) and np.array_equal(offsets, o2)
    idx2.free()


def test_candidates_full_size(full):
    M, ctx, vol, lens, p = full["M"], full["ctx"], full["vol"], full["lens"], full["p"]
    idx = full.get("idx") or M.Index(ctx, vol)
    full["idx"] = idx
    n = len(lens)
    cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, n, p)
    full["cands"], full["cnt"] = cands, cnt
    assert cnt.min() >= 0 and cnt.max() <= p.maxc and int(cnt.sum()) > 20 * n
    mask = np.arange(p.maxc)[None, :] < cnt[:, None]
    qid = np.broadcast_to(np.arange(n)[:, None], mask.shape)[mask]
    c = cands[mask]
    assert np.all(c["readno"] < qid), "subject id must be smaller than the query id (sid > qid dropped, sid == qid scrubbed)"
    assert set(np.unique(c["chain"])) <= {0, 1}
    ssize = lens[c["readno"]]
    assert np.all((c["loc1"] >= 0) & (c["loc1"] < ssize)) and np.all((c["loc2"] >= 0) & (c["loc2"] < lens[qid]))
    assert np.all(c["num1"] + c["num2"] >= p.min_kmer_dist) and np.all(c["score"] >= 2 * p.min_kmer_match + 2)
    sc = np.where(mask, cands["score"], np.iinfo(np.int32).max)
    sc_next = np.where(mask[:, 1:], cands["score"][:, 1:], -1)
    assert np.all(sc[:, :-1] >= sc_next), "candidate lists are sorted by score, descending"
    # the strand-resident pipeline took (practically) every strand with hits; the kernel chain gives the same lists
    took, left = ctx.debug_counter(13), ctx.debug_counter(14)
    assert took > 0 and left <= 0.01 * took, (took, left)
    with _Env(MECAT_SEED_FUSED="0"):
        c2, n2 = M.seed_reads(ctx, idx, vol, vol, 0, n, p)
    assert np.array_equal(cnt, n2)
    assert np.array_equal(cands[mask], c2[mask])
    del c2
    # buckets walked to their ends (no cut behind the read's own copy)
    walked, hits = ctx.debug_counter(15), ctx.counters()["hits"]
    assert 0.4 * hits < walked < 0.7 * hits, (walked, hits)
    with _Env(MECAT_SEED_CUTS="0"):
        c2, n2 = M.seed_reads(ctx, idx, vol, vol, 0, n, p)
    assert np.array_equal(cnt, n2)
    assert np.array_equal(cands[mask], c2[mask])
    del c2
    # every gated segment listed (no early drop of the segments whose subjects all have higher ids)
    with _Env(MECAT_SEED_PREDROP="0"):
        c2, n2 = M.seed_reads(ctx, idx, vol, vol, 0, n, p)
    assert np.array_equal(cnt, n2)
    assert np.array_equal(cands[mask], c2[mask])
    del c2
    # independent implementation: no relevance filter
    with _Env(MECAT_SEED_FILTER="0"):
        c2, n2 = M.seed_reads(ctx, idx, vol, vol, 0, n, p)
    assert np.array_equal(cnt, n2)
    assert np.array_equal(cands[mask], c2[mask])
    del c2
    # the multi-GPU shard: two strided halves == contiguous
    import torch
    dev = torch.device("cuda", 0)
    for rank in range(2):
        nl = (n - rank + 1) // 2
        dc = torch.zeros((nl, p.maxc, 12), dtype=torch.int32, device=dev)
        dn = torch.zeros((nl,), dtype=torch.int32, device=dev)
        M.seed_reads_strided_dev(ctx, idx, vol, vol, rank, 2, nl, p, dc.data_ptr(), dn.data_ptr())
        ctx.sync()
        assert np.array_equal(dn.cpu().numpy(), cnt[rank::2])
        got = dc.cpu().numpy().reshape(nl, p.maxc, 12)
        m2 = mask[rank::2]
        want = np.stack([cands[rank::2][f] for f in M.CAND_DTYPE.names], axis=2)
        assert np.array_equal(got[m2], want[m2])


def test_extension_full_size(full):
    M, W, ctx, vol, lens, p = full["M"], full["W"], full["ctx"], full["vol"], full["lens"], full["p"]
    if "cands" not in full:
        pytest.skip("needs the candidates of test_candidates_full_size")
    jobs = W.jobs_from_candidates(full["cands"], full["cnt"], 0)
    res = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size)
    qs, ts = lens[jobs["qid_local"]], lens[jobs["sid_local"]]
    assert np.all((0 <= res["query_start"]) & (res["query_start"] <= jobs["qstart"]) & (jobs["qstart"] <= res["query_end"]) & (res["query_end"] <= qs))
    assert np.all((0 <= res["target_start"]) & (res["target_start"] <= jobs["sstart"]) & (jobs["sstart"] <= res["target_end"]) & (res["target_end"] <= ts))
    qspan, tspan = res["query_end"] - res["query_start"], res["target_end"] - res["target_start"]
    assert np.all(res["matches"] <= np.minimum(qspan, tspan)) and np.all(res["columns"] >= np.maximum(qspan, tspan))
    assert np.all(res["columns"] == qspan + tspan - res["matches"]), "O(ND) paths have no mismatch columns"
    assert np.array_equal(res["ok"] != 0, res["columns"] >= p.min_align_size)
    ok = res["ok"] != 0
    assert ok.mean() > 0.99
    ident = 100.0 * res["matches"][ok] / res["columns"][ok]
    assert 70.0 < ident.mean() < 80.0          # two reads at 15 % error each
    # independent implementation: one unit per wave
    with _Env(MECAT_DW_KERNEL="1"):
        res1 = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size)
    for f in ("ok", "query_start", "query_end", "target_start", "target_end", "matches", "columns", "blocks"):
        assert np.array_equal(res[f], res1[f]), f
    full["idx"].free()
