"""Multi-GPU entry points of the C ABI (mhip_comm_*, mhip_seed_reads_sharded, mhip_align_sharded) on ONE GPU:
two or three ranks as threads of this process, each with its own context (stream + scratch) and a communicator on the
host-file transport — the test hook for boxes with fewer GPUs than ranks (RCCL refuses two ranks on one device).  Same call
sequence, same kernels and the same count-then-payload exchange as with RCCL; only the bytes travel through files.

Checks: every rank ends up with exactly the table / results of the single-GPU calls, for the reference's chunk of 500 reads and
for ragged shards (chunk 64, a read count that is not a multiple of anything, a slab that starts inside the volume)."""
import os
import tempfile
import threading
import uuid

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
os.environ.setdefault("MECAT_HIP_COMM_TIMEOUT_S", "30")      # a failed rank must not leave its peer waiting for minutes


@pytest.fixture(scope="module")
def data():
    import mecat_amd.hip as M
    from mecat_amd import workload as W
    codes, lens = W.synth_reads(2731, 3000, 0.15, 300_000, 21)
    pac, offs, nb = W.pack_volume(codes, lens)
    ctx = M.Context(0)
    vol = M.Volume(ctx, pac, offs, nb, 0)
    idx = M.Index(ctx, vol)
    p = M.default_params(0)
    cands, cnt = M.seed_reads(ctx, idx, vol, vol, 0, len(lens), p)
    jobs = W.jobs_from_candidates(cands, cnt, 0)
    res = M.align_candidates(ctx, vol, vol, jobs, p.min_align_size)
    d = dict(M=M, ctx=ctx, vol=vol, idx=idx, p=p, n=len(lens), cands=cands, cnt=cnt, res=res)
    yield d
    idx.free()
    vol.free()
    ctx.close()


def _run_ranks(nranks, fn):
    out, err = [None] * nranks, [None] * nranks

    def body(r):
        try:
            out[r] = fn(r)
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc()
            err[r] = e

    th = [threading.Thread(target=body, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("nranks,chunk,rb,re,shift", [(2, 500, 0, None, 0), (3, 64, 0, None, 1), (2, 64, 640, 2500, 5), (3, 500, 500, None, 2),
                                                        (4, 64, 0, None, 3), (8, 500, 0, None, 0), (8, 64, 128, 2600, 7)])
def test_sharded_seeding_and_extension_equal_single_gpu(data, nranks, chunk, rb, re, shift):
    M, vol, idx, p = data["M"], data["vol"], data["idx"], data["p"]
    re = data["n"] if re is None else re
    d = tempfile.mkdtemp(prefix="mecat_comm_")
    run = uuid.uuid4().hex[:8]

    def rank_body(r):
        ctx = M.Context(0)
        cm = M.Comm(ctx, nranks, r, hostfile_dir=d, run_id=run)
        cm.barrier()
        cands, cnt = cm.seed_reads_sharded(idx, vol, vol, rb, re, p, chunk=chunk, cell_shift=shift)
        res, nj = cm.align_sharded(vol, vol, p.min_align_size)
        got = (cands.copy(), cnt.copy(), res.copy(), nj, cm.bytes_received())
        cm.barrier()
        cm.close()
        ctx.close()
        return got

    outs = _run_ranks(nranks, rank_body)
    want_c, want_n = data["cands"][rb:re], data["cnt"][rb:re]
    first = np.concatenate([[0], np.cumsum(data["cnt"].astype(np.int64))])
    want_r = data["res"][first[rb]: first[re]]
    mask = np.arange(p.maxc)[None, :] < want_n[:, None]
    for r, (cands, cnt, res, nj, nbytes) in enumerate(outs):
        assert np.array_equal(cnt, want_n), r
        assert np.array_equal(cands[mask], want_c[mask]), r
        assert nj == int(want_n.sum()) == len(res)
        assert res.tobytes() == want_r.tobytes(), r
        # count-then-payload: what a rank receives is the other ranks' records and counts, not max-padded slabs
        total = int(want_n.sum())
        assert nbytes < (48 + 32) * total + 4 * (re - rb + nranks * chunk) * 2 + 4096, (nbytes, total)


@pytest.mark.parametrize("nranks", [2, 3, 4, 8])      # (bench.py and the driver shard the index from four ranks on)
def test_sharded_index_build_equals_single_build(data, nranks):
    """mhip_index_build_sharded: every rank builds the buckets of its own k-mer key range, positions and table slices are gathered,
    and every rank must hold the table mhip_index_build makes — bucket boundaries, positions, segment slots and bucket records, array
    for array (host-file transport on one GPU)"""
    M, vol, idx = data["M"], data["vol"], data["idx"]
    want_c, want_o = idx.download()
    want_s, want_r, want_cs = idx.download_aux()
    assert want_r is not None
    d = tempfile.mkdtemp(prefix="mecat_comm_")
    run = uuid.uuid4().hex[:8]

    def rank_body(r):
        ctx = M.Context(0)
        cm = M.Comm(ctx, nranks, r, hostfile_dir=d, run_id=run)
        cm.barrier()
        ix = cm.index_build_sharded(vol)
        got = ix.download() + ix.download_aux() + (ix.num_kmers, cm.bytes_received())
        # ... and it seeds like the single build
        cands, cnt = M.seed_reads(ctx, ix, vol, vol, 0, 300, data["p"])
        ix.free()
        cm.barrier()
        cm.close()
        ctx.close()
        return got + (cands, cnt)

    for r, (c, o, s, rec, cs, nk, nbytes, cands, cnt) in enumerate(_run_ranks(nranks, rank_body)):
        assert nk == idx.num_kmers, r
        assert np.array_equal(c, want_c) and np.array_equal(o, want_o), r
        assert np.array_equal(s, want_s) and cs == want_cs and np.array_equal(rec, want_r), r
        assert np.array_equal(cnt, data["cnt"][:300]), r
        mask = np.arange(data["p"].maxc)[None, :] < cnt[:, None]
        assert np.array_equal(cands[mask], data["cands"][:300][mask]), r
        # a rank receives the other ranks' positions and table slices, not the table it built itself
        assert nbytes < 4 * idx.num_kmers + (4 + 16) * (1 << 26) + 4096, (nbytes, idx.num_kmers)


@pytest.mark.parametrize("nranks,force", [(2, None), (3, None), (2, "1"), (2, "0")])
def test_index_build_auto_measures_then_keeps_one_way(data, nranks, force, monkeypatch):
    """mhip_index_build_auto: the first call on a communicator builds the table both ways, timed barrier to barrier,
    every rank arrives at the same choice (the slowest rank's times), later calls build that way without measuring; the table is the
    single build's either way.  MECAT_HIP_INDEX_SHARD decides without measuring."""
    M, vol, idx = data["M"], data["vol"], data["idx"]
    want_c, want_o = idx.download()
    if force is None:
        monkeypatch.delenv("MECAT_HIP_INDEX_SHARD", raising=False)
    else:
        monkeypatch.setenv("MECAT_HIP_INDEX_SHARD", force)
    d = tempfile.mkdtemp(prefix="mecat_comm_")
    run = uuid.uuid4().hex[:8]

    def rank_body(r):
        ctx = M.Context(0)
        cm = M.Comm(ctx, nranks, r, hostfile_dir=d, run_id=run)
        cm.barrier()
        got = []
        for call in range(2):
            ix, how = cm.index_build_auto(vol)
            got.append((how, ix.download(), ix.num_kmers))
            ix.free()
        cm.barrier()
        cm.close()
        ctx.close()
        return got

    outs = _run_ranks(nranks, rank_body)
    first = outs[0][0][0]
    assert first["measured"] == (force is None)
    if force is None:
        assert first["replicated_ms"] > 0 and first["sharded_ms"] > 0
        assert first["chosen"] == ("sharded" if first["sharded_ms"] < first["replicated_ms"] else "replicated")
    else:
        assert first["chosen"] == ("sharded" if force == "1" else "replicated")
    for r, calls in enumerate(outs):
        for how, (c, o), nk in calls:
            assert how == first, (r, how, first)              # same times and the same choice on every rank, unchanged by the second call
            assert nk == idx.num_kmers and np.array_equal(c, want_c) and np.array_equal(o, want_o), r


def test_rccl_loads_and_moves_bytes_on_this_device(data):
    """the RCCL transport itself cannot run two ranks on one GPU; this checks what can be checked here: librccl is found,
    every symbol the library uses resolves, a communicator comes up on the context's device and both transport forms
    (ncclAllGather, grouped ncclSend / ncclRecv) move the right bytes on the context's stream"""
    M, ctx = data["M"], data["ctx"]
    M._chk(M.lib().mhip_comm_selftest(ctx.h))
    uid = M.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    cm = M.Comm(ctx, 1, 0, unique_id=uid)          # one rank: same calls, nothing to exchange
    cands, cnt = cm.seed_reads_sharded(data["idx"], data["vol"], data["vol"], 0, data["n"], data["p"])
    assert np.array_equal(cnt, data["cnt"])
    res, nj = cm.align_sharded(data["vol"], data["vol"], data["p"].min_align_size)
    assert res.tobytes() == data["res"].tobytes()
    cm.close()
