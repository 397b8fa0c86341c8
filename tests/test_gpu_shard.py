"""Device-resident (_dev) entry points and the multi-GPU shard arithmetic, exercised on ONE GPU by playing both ranks:
strided seeding + merge == contiguous seeding; part_index/part_count job split covers every job once; _dev alignment
== host-pointer alignment."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def test_two_virtual_ranks_on_one_gpu():
    import torch
    import mecat_amd.hip as M
    from mecat_amd import shard as S
    from mecat_amd import workload as W
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = M.Context(0, stream.cuda_stream)
    codes, lens = H.synth_reads(201, 3000, 0.15, 30000, 41)
    pac, offs, nb = W.pack_volume(codes, lens)
    vol = M.Volume(ctx, pac, offs, nb, 0)
    idx = M.Index(ctx, vol)
    p = M.default_params(0)
    n, maxc, world = len(lens), p.maxc, 2
    full, cnt = M.seed_reads(ctx, idx, vol, vol, 0, n, p)
    n_pad = S.padded_count(n, world)
    merged_c = torch.zeros((n_pad * world, maxc, 12), dtype=torch.int32, device=dev)
    merged_n = torch.zeros((n_pad * world,), dtype=torch.int32, device=dev)
    for rank in range(world):
        nl = S.local_count(n, rank, world)
        dc = torch.zeros((n_pad, maxc, 12), dtype=torch.int32, device=dev)
        dn = torch.zeros((n_pad,), dtype=torch.int32, device=dev)
        M.seed_reads_strided_dev(ctx, idx, vol, vol, rank, world, nl, p, dc.data_ptr(), dn.data_ptr())
        ctx.sync()
        merged_c[rank::world] = dc
        merged_n[rank::world] = dn
    merged_c, merged_n = merged_c[:n].contiguous(), merged_n[:n].contiguous()
    assert np.array_equal(merged_n.cpu().numpy(), cnt)
    got = merged_c.cpu().numpy()
    for r in range(n):
        exp = np.stack([full[r][: cnt[r]][f] for f in M.CAND_DTYPE.names], axis=1) if cnt[r] else np.zeros((0, 12), np.int32)
        assert np.array_equal(got[r, : cnt[r]], exp), r
    # job split
    total = int(cnt.sum())
    want_jobs = W.jobs_from_candidates(full, cnt, 0)
    allj = np.zeros(total, dtype=M.JOB_DTYPE)
    res_parts = []
    for rank in range(world):
        dj = torch.zeros((n * maxc + maxc, 5), dtype=torch.int32, device=dev)
        k = M.jobs_from_candidates_dev(ctx, merged_c.data_ptr(), merged_n.data_ptr(), n, maxc, 0, 1, 0, rank, world, dj.data_ptr())
        assert k == S.my_job_count(total, rank, world)
        dr = torch.zeros((k + 1, 8), dtype=torch.int32, device=dev)
        M.align_candidates_dev(ctx, vol, vol, dj.data_ptr(), k, p.min_align_size, dr.data_ptr())
        ctx.sync()
        allj[rank::world] = dj[:k].cpu().numpy().view(M.JOB_DTYPE).reshape(-1)
        res_parts.append(dr[:k].cpu().numpy())
    assert np.array_equal(allj, want_jobs)
    host = M.align_candidates(ctx, vol, vol, want_jobs, p.min_align_size)
    for rank in range(world):
        exp = np.stack([host[rank::world][f] for f in M.ALN_DTYPE.names], axis=1)
        assert np.array_equal(res_parts[rank], exp)
    assert int(host["ok"].sum()) > 100
    idx.free()
    vol.free()
    ctx.close()
