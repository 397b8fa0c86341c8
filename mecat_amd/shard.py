"""Static multi-GPU sharding of one (reference volume, query volume) grid cell and the candidate exchange
(SURVEY.md §8e).  Pure tensor plumbing on top of torch.distributed (backend "nccl" = RCCL on the GPUs, "gloo" in the
CPU tests); no compute.

Shard: rank r of P owns query reads r, r+P, r+2P, ... (cyclic: candidates with sid > qid are dropped,
pw_impl.cpp:370 of the reference, so work per read grows with the read id inside a diagonal cell).
Exchange: one all-gather of the fixed-size per-read candidate slabs [ceil(n/P)][MAXC] x 48-byte candidate_save records
plus the per-read counts; afterwards every rank holds the complete read-major table.  The gather is started
asynchronously (start_all_gather_candidates) and runs on RCCL's stream while the rank extends the candidates of its own
reads (the cyclic read shard balances that work to ~1 %), then the per-rank result slabs are gathered
(all_gather_results_by_rank).  The blocking variants re-shard the extension stage by candidate instead.
"""
import torch
import torch.distributed as dist


def local_count(n_reads, rank, world):
    """number of reads owned by `rank`: r, r + world, ... < n_reads"""
    return (n_reads - rank + world - 1) // world if rank < n_reads else 0


def padded_count(n_reads, world):
    return (n_reads + world - 1) // world


def all_gather_candidates(local_cands, local_counts, n_reads, world):
    """local_cands [n_pad, maxc, 12] int32, local_counts [n_pad] int32 (row i = read rank + i * world; rows past the
    rank's share must have count 0).  Returns read-major (cands [n_reads, maxc, 12], counts [n_reads])."""
    if world == 1:
        return local_cands[:n_reads], local_counts[:n_reads]
    n_pad, maxc, w = local_cands.shape
    g_counts = torch.empty((world, n_pad), dtype=local_counts.dtype, device=local_counts.device)
    g_cands = torch.empty((world, n_pad, maxc, w), dtype=local_cands.dtype, device=local_cands.device)
    # flat views: the gloo backend (CPU tests) only accepts a concatenated 1-D output; RCCL takes either
    dist.all_gather_into_tensor(g_counts.view(-1), local_counts.contiguous().view(-1))
    dist.all_gather_into_tensor(g_cands.view(-1), local_cands.contiguous().view(-1))
    # table row i of rank r is read r + i * world  ->  read id = i * world + r
    counts = g_counts.transpose(0, 1).reshape(-1)[:n_reads].contiguous()
    cands = g_cands.transpose(0, 1).reshape(n_pad * world, maxc, w)[:n_reads].contiguous()
    return cands, counts


def my_job_count(total_jobs, rank, world):
    """jobs g with g % world == rank, stored at slot g // world (mhip_jobs_from_candidates_dev part_index/part_count)"""
    return total_jobs // world + (1 if rank < total_jobs % world else 0)


def all_gather_results(local_res, n_local, total_jobs, world):
    """local_res [cap, 8] int32 with the first n_local rows valid (job g = slot * world + rank).
    Returns [total_jobs, 8] in global job order on every rank."""
    if world == 1:
        return local_res[:total_jobs]
    m = (total_jobs + world - 1) // world
    slab = torch.zeros((m, local_res.shape[1]), dtype=local_res.dtype, device=local_res.device)
    slab[:n_local] = local_res[:n_local]
    g = torch.empty((world, m, local_res.shape[1]), dtype=local_res.dtype, device=local_res.device)
    dist.all_gather_into_tensor(g.view(-1), slab.view(-1))
    return g.transpose(0, 1).reshape(m * world, local_res.shape[1])[:total_jobs].contiguous()


def start_all_gather_candidates(local_cands, local_counts, world):
    """Asynchronous form of all_gather_candidates: returns a pending object for finish_all_gather_candidates.  The
    collective waits for the work already queued on the current stream and then proceeds on the backend's own stream."""
    n_pad, maxc, w = local_cands.shape
    g_counts = torch.empty((world, n_pad), dtype=local_counts.dtype, device=local_counts.device)
    g_cands = torch.empty((world, n_pad, maxc, w), dtype=local_cands.dtype, device=local_cands.device)
    h1 = dist.all_gather_into_tensor(g_counts.view(-1), local_counts.contiguous().view(-1), async_op=True)
    h2 = dist.all_gather_into_tensor(g_cands.view(-1), local_cands.contiguous().view(-1), async_op=True)
    return (h1, h2, g_counts, g_cands)


def finish_all_gather_candidates(pending, n_reads):
    """-> (cands [n_reads, maxc, 12], counts [n_reads]) read-major, and the per-rank candidate totals [world]"""
    h1, h2, g_counts, g_cands = pending
    h1.wait()
    h2.wait()
    world, n_pad = g_counts.shape
    maxc, w = g_cands.shape[2], g_cands.shape[3]
    counts = g_counts.transpose(0, 1).reshape(-1)[:n_reads].contiguous()
    cands = g_cands.transpose(0, 1).reshape(n_pad * world, maxc, w)[:n_reads].contiguous()
    return cands, counts, g_counts.sum(dim=1)


def all_gather_results_by_rank(local_res, per_rank_jobs):
    """local_res [cap, 8] with the first per_rank_jobs[rank] rows valid (the rank's own reads, read-major).  Returns the
    rows of all ranks, rank-major, [sum(per_rank_jobs), 8] on every rank.  per_rank_jobs: 1-D tensor or list, length world."""
    per = [int(x) for x in per_rank_jobs]
    world = len(per)
    rank = dist.get_rank()
    m = max(max(per), 1)
    slab = torch.zeros((m, local_res.shape[1]), dtype=local_res.dtype, device=local_res.device)
    slab[: per[rank]] = local_res[: per[rank]]
    g = torch.empty((world, m, local_res.shape[1]), dtype=local_res.dtype, device=local_res.device)
    dist.all_gather_into_tensor(g.view(-1), slab.view(-1))
    return torch.cat([g[r, : per[r]] for r in range(world)], dim=0)
