"""N > 1 path on CPU: world_size 2 over gloo.  Each rank computes the candidate lists of its cyclic shard (with the
oracle as the stand-in compute, this is a test of the sharding / exchange / re-sharding plumbing in mecat_amd/shard.py
that bench.py runs over RCCL), all-gathers the slabs and must end up with exactly the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

N_READS, MAXC = 61, 100     # 61: not divisible by 2 -> exercises the padded slab row


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
    n_local, n_pad = S.local_count(N_READS, rank, world), S.padded_count(N_READS, world)
    mine = H.orc_seed_all(ov, ov, oidx, p, rids=range(rank, N_READS, world))
    assert len(mine) == n_local
    cands = torch.zeros((n_pad, MAXC, 12), dtype=torch.int32)
    counts = torch.zeros((n_pad,), dtype=torch.int32)
    for i, a in enumerate(mine):
        counts[i] = len(a)
        if len(a):
            cands[i, : len(a)] = torch.from_numpy(np.stack([a[f] for f in H.CAND_DTYPE.names], axis=1).astype(np.int32))
    full_cands, full_counts = S.all_gather_candidates(cands, counts, N_READS, world)
    # extension stage re-shard: every world-th candidate; fake "results" = (global job index, read id) to check the merge
    total = int(full_counts.sum())
    mask = torch.arange(MAXC)[None, :] < full_counts[:, None]
    read_of_job = torch.arange(N_READS)[:, None].expand(N_READS, MAXC)[mask]
    my_jobs = torch.arange(total)[rank::world]
    assert len(my_jobs) == S.my_job_count(total, rank, world)
    res = torch.zeros((len(my_jobs) + 3, 8), dtype=torch.int32)
    res[: len(my_jobs), 0] = my_jobs.int()
    res[: len(my_jobs), 1] = read_of_job[my_jobs].int()
    allres = S.all_gather_results(res, len(my_jobs), total, world)
    # overlapped form used by bench.py: asynchronous gather, the rank extends its own reads, results gathered rank-major
    pending = S.start_all_gather_candidates(cands, counts, world)
    n_mine = int(counts.sum())
    res2 = torch.zeros((n_mine + 5, 8), dtype=torch.int32)
    k = 0
    for i in range(n_local):
        for j in range(int(counts[i])):
            res2[k, 0], res2[k, 1], res2[k, 2] = rank, rank + i * world, j      # (rank, read id, slot)
            k += 1
    fc2, fn2, per_rank = S.finish_all_gather_candidates(pending, N_READS)
    assert torch.equal(fc2, full_cands) and torch.equal(fn2, full_counts) and int(per_rank[rank]) == n_mine
    allres2 = S.all_gather_results_by_rank(res2, per_rank)
    torch.save({"cands": full_cands, "counts": full_counts, "res": allres, "res2": allres2, "per_rank": per_rank},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_exchange_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ov, oidx = _dataset()
    want = H.orc_seed_all(ov, ov, oidx, H.orc_params(tech=0, maxc=MAXC))
    outs = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    for o in outs:
        assert [int(x) for x in o["counts"]] == [len(a) for a in want]
        for rid, a in enumerate(want):
            got = o["cands"][rid, : len(a)].numpy()
            exp = np.stack([a[f] for f in H.CAND_DTYPE.names], axis=1) if len(a) else np.zeros((0, 12), np.int32)
            assert np.array_equal(got, exp), rid
        total = sum(len(a) for a in want)
        assert o["res"].shape[0] == total
        assert [int(x) for x in o["res"][:, 0]] == list(range(total))          # global job order restored
        rid_of = [rid for rid, a in enumerate(want) for _ in range(len(a))]
        assert [int(x) for x in o["res"][:, 1]] == rid_of
        # rank-major result table: rank r's reads r, r + world, ... with their slots in order
        exp2 = [(r, rid, j) for r in range(world) for rid in range(r, N_READS, world) for j in range(len(want[rid]))]
        assert [tuple(int(v) for v in row[:3]) for row in o["res2"]] == exp2
        assert [int(x) for x in o["per_rank"]] == [sum(len(want[rid]) for rid in range(r, N_READS, world)) for r in range(world)]
    assert torch.equal(outs[0]["cands"], outs[1]["cands"])
    assert sum(len(a) for a in want) > 50
