"""N > 1 path on CPU: world_size 2 over gloo.  Each rank takes its chunked shard of a grid cell (shard arithmetic from the
library's own mhip_shard_* exports — they need no GPU), computes the candidate lists of its reads with the oracle as the
stand-in compute, and runs the count-then-payload exchange of mecat_amd/shard.py — the torch mirror of the protocol
libmecat_hip.so runs over RCCL (comm.hip; the GPU-side twin of this test is tests/test_gpu_comm.py).  Every rank must end
up with exactly the single-process table, and a rank must receive the other ranks' records, not max-padded slabs."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

N_READS, MAXC, CHUNK, SHIFT = 61, 100, 8, 1     # 61 reads in chunks of 8: ragged last chunk, uneven shards


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dataset():
    codes, lens = H.synth_reads(N_READS, 3000, 0.15, 12000, 31)
    ov = H.orc_pack(codes, lens)
    oidx = H.orc().orc_index_build(ov)
    return ov, oidx


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(H.ROOT))
    from mecat_amd import shard as S
    ov, oidx = _dataset()
    p = H.orc_params(tech=0, maxc=MAXC)
    rids = S.local_reads(0, N_READS, CHUNK, SHIFT, rank, world)
    mine = H.orc_seed_all(ov, ov, oidx, p, rids=rids)
    cands = torch.zeros((max(len(rids), 1), MAXC, 12), dtype=torch.int32)
    counts = torch.zeros((max(len(rids), 1),), dtype=torch.int32)
    for i, a in enumerate(mine):
        counts[i] = len(a)
        if len(a):
            cands[i, : len(a)] = torch.from_numpy(np.stack([a[f] for f in H.CAND_DTYPE.names], axis=1).astype(np.int32))
    full_cands, full_counts, totals, received = S.all_gather_candidates(cands, counts, 0, N_READS, CHUNK, SHIFT)
    # extension stage: fake "results" = (rank, read id, slot) for this rank's candidates, local read-major
    res = torch.zeros((totals[rank] + 3, 8), dtype=torch.int32)
    k = 0
    for i, rid in enumerate(rids):
        for j in range(int(counts[i])):
            res[k, 0], res[k, 1], res[k, 2] = rank, rid, j
            k += 1
    allres = S.all_gather_results(res, full_counts, 0, N_READS, CHUNK, SHIFT)
    torch.save({"cands": full_cands, "counts": full_counts, "res": allres, "totals": totals, "received": received, "rids": rids},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_exchange_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ov, oidx = _dataset()
    want = H.orc_seed_all(ov, ov, oidx, H.orc_params(tech=0, maxc=MAXC))
    outs = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    # the shards partition the reads: chunk c -> rank (c + SHIFT) % world
    assert sorted(outs[0]["rids"] + outs[1]["rids"]) == list(range(N_READS))
    for r, o in enumerate(outs):
        assert all(((rid // CHUNK) + SHIFT) % world == r for rid in o["rids"])
    total = sum(len(a) for a in want)
    for r, o in enumerate(outs):
        assert [int(x) for x in o["counts"]] == [len(a) for a in want]
        for rid, a in enumerate(want):
            got = o["cands"][rid, : len(a)].numpy()
            exp = np.stack([a[f] for f in H.CAND_DTYPE.names], axis=1) if len(a) else np.zeros((0, 12), np.int32)
            assert np.array_equal(got, exp), rid
        # results come back dense and read-major: read 0's slots, read 1's slots, ...
        exp = [(((rid // CHUNK) + SHIFT) % world, rid, j) for rid, a in enumerate(want) for j in range(len(a))]
        assert [tuple(int(v) for v in row[:3]) for row in o["res"]] == exp
        assert o["totals"] == [sum(len(want[rid]) for rid in outs[q]["rids"]) for q in range(world)]
        # count-then-payload: bytes received = the peer's counts + its occupied records (far below a max-padded slab)
        peer = 1 - r
        assert o["received"] == 4 * max(len(outs[0]["rids"]), len(outs[1]["rids"])) + 48 * o["totals"][peer]
        assert o["received"] < 48 * MAXC * len(outs[peer]["rids"]) / 3
    assert torch.equal(outs[0]["cands"], outs[1]["cands"]) and torch.equal(outs[0]["res"], outs[1]["res"])
    assert total > 50


def test_rows_mode_deal_is_balanced_and_deterministic():
    """rows mode (#volumes >= ranks): grid row i holds num_vols - i cells (pw.cpp:65-81, pw_impl.cpp:859-879); the cost-aware static
    deal keeps the heaviest rank within 10 % of the mean at config 5's 19 rows over 8 ranks (the cyclic deal i mod P: 33 of a mean
    23.75 cells, i.e. at most 5.76x on 8 GPUs), deals every todo row exactly once, and ignores finished rows."""
    from mecat_amd import hip as M
    owner, heaviest = M.deal_rows(19, np.arange(19), 8)
    cells = np.bincount(owner, weights=19 - np.arange(19), minlength=8)
    assert heaviest == cells.max() == 26 and cells.sum() == 190
    assert cells.max() <= 1.1 * cells.mean()
    cyc = np.bincount(np.arange(19) % 8, weights=19 - np.arange(19), minlength=8)
    assert cyc.max() == 33
    # a resume: only some rows left; the others are -1 and the rest is balanced over the ranks again
    todo = [3, 4, 9, 10, 11, 17, 18]
    owner2, h2 = M.deal_rows(19, todo, 4)
    assert sorted(np.nonzero(owner2 >= 0)[0].tolist()) == todo and (owner2[[0, 1, 2, 5]] == -1).all()
    c2 = np.bincount(owner2[todo], weights=19 - np.array(todo), minlength=4)
    assert h2 == c2.max() == 17                       # 16 | 15 | 10 + 2 + 1 | 9 + 8 (a perfect split of 61 cells would be 16)
    # sweep: never worse than the cyclic deal, every rank count
    for nv in (2, 5, 8, 19, 40):
        for P in (2, 3, 4, 8):
            o, h = M.deal_rows(nv, np.arange(nv), P)
            w = nv - np.arange(nv)
            assert h == np.bincount(o, weights=w, minlength=P).max() <= np.bincount(np.arange(nv) % P, weights=w, minlength=P).max()
    # bad arguments
    assert M.lib().mhip_shard_deal_rows(3, np.array([0, 0], np.int32).ctypes.data, 2, 2, np.zeros(3, np.int32).ctypes.data) == -1
