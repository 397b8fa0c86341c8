"""Python mirror of the multi-GPU exchange protocol of libmecat_hip.so (mecat_amd/csrc/comm.hip), on
torch.distributed tensors: the chunked static shard of a grid cell and the count-then-payload all-gather of the per-read
candidate lists and of the extension results.  The measured path (bench.py, the mecat2pw driver) runs the C ABI
(mhip_seed_reads_sharded / mhip_align_sharded over RCCL); this module exists so that the protocol — shard arithmetic taken
from the library's own exported mhip_shard_* functions, packing, displacements, the scatter back into read-major order —
is also exercised by a world-size-2 gloo test on machines without a GPU (tests/test_dist_cpu.py).  No compute here.

Shard: the reads [rid_begin, rid_end) of a query volume in chunks of `chunk` reads (the reference's CHUNK_SIZE = 500,
mecat2pw/pw_impl.h:15); chunk c belongs to rank (c + cell_shift) mod P.  Local index i of a rank is read
first + (i // chunk) * chunk * P + i % chunk.
"""
import torch
import torch.distributed as dist

from . import hip as M


def local_count(rid_begin, rid_end, chunk, cell_shift, rank, world):
    return M.lib().mhip_shard_local_count(rid_begin, rid_end, chunk, cell_shift, rank, world)


def first_read(rid_begin, rid_end, chunk, cell_shift, rank, world):
    return M.lib().mhip_shard_first_read(rid_begin, rid_end, chunk, cell_shift, rank, world)


def local_reads(rid_begin, rid_end, chunk, cell_shift, rank, world):
    """read ids owned by `rank`, ascending (local index order)"""
    n, f = local_count(rid_begin, rid_end, chunk, cell_shift, rank, world), first_read(rid_begin, rid_end, chunk, cell_shift, rank, world)
    return [f + (i // chunk) * chunk * world + i % chunk for i in range(n)]


def _allgatherv(parts_of_me, sizes, rank):
    """every rank contributes a [sizes[rank], w] tensor; returns the list of all ranks' tensors (exact sizes: one broadcast
    per rank — the gloo stand-in for the grouped send/recv of the RCCL transport)"""
    out = []
    for r, n in enumerate(sizes):
        t = parts_of_me if r == rank else torch.empty((n,) + tuple(parts_of_me.shape[1:]), dtype=parts_of_me.dtype)
        if n:
            dist.broadcast(t, src=r)
        out.append(t)
    return out


def all_gather_candidates(local_cands, local_counts, rid_begin, rid_end, chunk, cell_shift):
    """local_cands [n_local, maxc, 12] int32, local_counts [n_local] int32 -> (cands [n, maxc, 12], counts [n], per_rank
    totals, bytes_received) with n = rid_end - rid_begin, read-major, identical on every rank"""
    world, rank = dist.get_world_size(), dist.get_rank()
    maxc, w = local_cands.shape[1], local_cands.shape[2]
    nloc = [local_count(rid_begin, rid_end, chunk, cell_shift, r, world) for r in range(world)]
    n_pad = max(max(nloc), 1)
    # 1. counts: one int32 per read, padded to the longest shard
    pad = torch.zeros(n_pad, dtype=torch.int32)
    pad[: nloc[rank]] = local_counts[: nloc[rank]]
    g = [torch.empty(n_pad, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(g, pad)
    totals = [int(x.sum()) for x in g]
    # 2. payload: only the occupied records, dense, local read-major
    mask = torch.arange(maxc)[None, :] < local_counts[: nloc[rank], None]
    pack = local_cands[: nloc[rank]][mask]                       # [totals[rank], 12]
    dense = _allgatherv(pack, totals, rank)
    received = sum(4 * n_pad + 48 * totals[r] for r in range(world) if r != rank)
    # 3. scatter back into the read-major table
    n = rid_end - rid_begin
    cands = torch.zeros((n, maxc, w), dtype=local_cands.dtype)
    counts = torch.zeros(n, dtype=torch.int32)
    for r in range(world):
        rids = local_reads(rid_begin, rid_end, chunk, cell_shift, r, world)
        pos = 0
        for i, rid in enumerate(rids):
            c = int(g[r][i])
            counts[rid - rid_begin] = c
            cands[rid - rid_begin, :c] = dense[r][pos: pos + c]
            pos += c
    return cands, counts, totals, received


def all_gather_results(local_res, all_counts, rid_begin, rid_end, chunk, cell_shift):
    """local_res [totals[rank], 8]: results of this rank's candidates, local read-major.  -> [sum(counts), 8] dense, read-major"""
    world, rank = dist.get_world_size(), dist.get_rank()
    shards = [local_reads(rid_begin, rid_end, chunk, cell_shift, r, world) for r in range(world)]
    totals = [int(sum(int(all_counts[rid - rid_begin]) for rid in s)) for s in shards]
    dense = _allgatherv(local_res[: totals[rank]], totals, rank)
    first = torch.zeros(rid_end - rid_begin + 1, dtype=torch.int64)
    first[1:] = torch.cumsum(all_counts.to(torch.int64), 0)
    out = torch.zeros((int(first[-1]), local_res.shape[1]), dtype=local_res.dtype)
    for r in range(world):
        pos = 0
        for rid in shards[r]:
            c = int(all_counts[rid - rid_begin])
            out[int(first[rid - rid_begin]): int(first[rid - rid_begin]) + c] = dense[r][pos: pos + c]
            pos += c
    return out
