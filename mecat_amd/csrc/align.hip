// align.hip — "dw": banded O(ND) furthest-reaching d-path local aligner in 500-bp blocks.
//
// Replaces, for PacBio-mode candidates (pairwise_mapping, mecat2pw/pw_impl.cpp:674-698):
//   DiffAligner::go          common/diff_gapalign.cpp:294-349   left + right extension from the seed point, stitching
//   dw_in_one_direction      common/diff_gapalign.cpp:221-292   block chaining with the 4-match tail anchor
//   retrieve_next_aln_block  common/gapalign.cpp:9-45           500-bp blocks, last-block sizing
//   Align                    common/diff_gapalign.cpp:107-219   O(ND) wavefront with the adaptive band
//   GetAlignString           common/diff_gapalign.cpp:39-104    traceback
//   trim_mismatch_end        common/gapalign.cpp:47-68          tail anchor
//
// mecat2pw consumes only the four end coordinates and the identity (matches / columns) of an alignment, never the
// aligned strings, so the traceback is reduced to what those need.  An O(ND) path is a chain of d+1 match runs
// ("snakes") separated by single-base indels, hence per block
//     columns = (x + y + d) / 2,  matches = (x + y - d) / 2                       (Align: aln_str_size)
// and trim_mismatch_end (4 consecutive equal columns from the tail) is "walk the path back to the first snake of
// length >= 4".  Only that tail of the path is traced.
//
// Mapping: one wave per (candidate, direction); a direction is a sequential chain of blocks; inside a block the
// d-loop is sequential and the <= 217 diagonals of the band are the lanes (up to 4 per lane).  Everything hot lives
// in LDS: V/U wavefront arrays (diagonal -> furthest x, x + y), the two 2-bit->byte block sequences, and a 16-row ring
// of the most recent d-rows (u16 x per diagonal) for the tail traceback.  A path whose last >= 4 snake is more than
// 16 rows back (p ~ 3e-5 at 15 % error) re-runs the block with the rows spilled to a per-wave global scratch.
// Waves are persistent and pull (candidate, direction) units from an atomic cursor; a second tiny kernel stitches the
// two directions (query_start = qstart - left bases, ...).
//
// Roofline: integer compare + LDS; HBM traffic is the two block sequences (2 bits per base) and a 32-byte result, so the
// HBM fraction is small by construction; cells and snake bases are counted for the VALU/LDS view.
#include <algorithm>

#include "common.h"

#define AL_BLOCK 256
#define AL_WAVES (AL_BLOCK / WAVE)
#define SEG_BLK 500            // DiffAlignParameters::segment_size, diff_gapalign.h:35
#define MAX_QB 736             // last block (gapalign.cpp:24-30): one side < 600, the other <= int(599 * 1.2) = 718
#define MAX_TB 736
#define MAX_D 400              // int(0.3 * (599 + 718)) = 395
#define VU_LEN (2 * MAX_D + 8)
#define ROW_W 224              // diagonals per d-row: (2 * int(0.3 * 718)) / 2 + 1 = 216
#define RING 16

struct DirResult {
    int32_t qbases, tbases, matches, columns, blocks, pad;
};

struct AlnWaveLds {
    int16_t V[VU_LEN];
    int16_t U[VU_LEN];
    int16_t rmin[MAX_D + 8];
    int16_t rmax[MAX_D + 8];
    uint16_t ring[RING * ROW_W];
    uint8_t Q[MAX_QB];
    uint8_t T[MAX_TB];
};

struct SeqView {
    const uint32_t* pac;
    int off;        // volume offset of the read
    int len;        // read length
    int rc;         // 1: reverse-complemented view
    int start;      // first logical position of the extension (qstart or qstart - 1)
    int step;       // +1 right extension, -1 left extension
};

// code of logical extension position i (0-based from the seed point)
__device__ __forceinline__ uint32_t seq_at(const SeqView& s, int i) {
    int p = s.start + s.step * i;                       // position in the strand's coordinate
    if (s.rc) return 3u - pac_base(s.pac, (int64_t)s.off + (s.len - 1 - p));
    return pac_base(s.pac, (int64_t)s.off + p);
}

struct BlockOut {
    int aligned_or_best;   // 1 when an alignment string exists (aln_str_size > 0 possible)
    int qe, te, dist;      // aln_q_e, aln_t_e, dist
    int qcnt, tcnt, acnt;  // trim_mismatch_end outputs
    int trim_ok;
    int fallback;          // ring too short for the tail traceback
};

// One block: Align + tail traceback + trim_mismatch_end.  `grow` == nullptr -> rows in the LDS ring, else global rows.
__device__ void align_block(AlnWaveLds& S, int q_len, int t_len, uint16_t* __restrict__ grow, BlockOut& o,
                            unsigned long long& cells, unsigned long long& snake) {
    const int lane = lane_id();
    const int band_tol = (int)(0.3 * (q_len > t_len ? q_len : t_len));      // dw_in_one_direction passes 0.3 * max(qblk, tblk)
    const int max_d = (int)(.3 * (q_len + t_len));
    const int k_offset = max_d;
    const int band_size = band_tol * 2;
    for (int i = lane; i < 2 * max_d + 4 && i < VU_LEN; i += 64) { S.V[i] = 0; S.U[i] = 0; }
    __builtin_amdgcn_wave_barrier();
    int best_m = -1, best_x = -1, best_y = -1, best_d = 0, best_k = 0;
    int min_k = 0, max_k = 0;
    int aligned = 0, end_x = 0, end_y = 0, end_d = 0, end_k = 0;
    o.fallback = 0;
    int d, last_row = -1;
    for (d = 0; d < max_d; ++d) {
        if (max_k - min_k > band_size) break;
        last_row = d;
        const int nslot = (max_k - min_k) / 2 + 1;
        if (lane == 0) { S.rmin[d] = (int16_t)min_k; S.rmax[d] = (int16_t)max_k; }
        uint16_t* row = grow ? grow + (size_t)d * ROW_W : S.ring + (d % RING) * ROW_W;
        int my_m = -1, my_x = 0, my_k = 0;          // best (first max) among this lane's diagonals
        int hit_k = 0x7fffffff, hit_x = 0;          // lowest diagonal of this lane that reached an end
        int xs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = lane + 64 * j;
            xs[j] = -1;
            if (t < nslot) {
                const int k = min_k + 2 * t;
                int x;
                const int vl = S.V[k - 1 + k_offset], vr = S.V[k + 1 + k_offset];
                if (k == min_k || (k != max_k && vl < vr)) x = vr; else x = vl + 1;
                int y = x - k;
                const int x0 = x;
                while (x < q_len && y < t_len && S.Q[x] == S.T[y]) { ++x; ++y; }
                snake += (unsigned long long)(x - x0);
                xs[j] = x;
                if (x + y > my_m) { my_m = x + y; my_x = x; my_k = k; }
                if ((x >= q_len || y >= t_len) && k < hit_k) { hit_k = k; hit_x = x; }
            }
        }
        cells += (unsigned long long)__popcll(__ballot(xs[0] >= 0)) + __popcll(__ballot(xs[1] >= 0)) +
                 __popcll(__ballot(xs[2] >= 0)) + __popcll(__ballot(xs[3] >= 0));
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = lane + 64 * j;
            if (t < nslot) {
                const int k = min_k + 2 * t;
                S.V[k + k_offset] = (int16_t)xs[j];
                S.U[k + k_offset] = (int16_t)(2 * xs[j] - k);
                row[t] = (uint16_t)xs[j];
            }
        }
        // first maximum of x + y in (d, k) order (diff_gapalign.cpp:160-167)
        int rm = my_m, rk = my_k, rx = my_x;
        for (int off = 32; off > 0; off >>= 1) {
            int om = __shfl_xor(rm, off), ok = __shfl_xor(rk, off), ox = __shfl_xor(rx, off);
            if (om > rm || (om == rm && ok < rk)) { rm = om; rk = ok; rx = ox; }
        }
        if (rm > best_m) { best_m = rm; best_x = rx; best_y = rx - rk; best_d = d; best_k = rk; }
        // lowest diagonal that reached an end (the sequential k loop breaks there, :168-169)
        int hk = hit_k, hx = hit_x;
        for (int off = 32; off > 0; off >>= 1) {
            int ok = __shfl_xor(hk, off), ox = __shfl_xor(hx, off);
            if (ok < hk) { hk = ok; hx = ox; }
        }
        __builtin_amdgcn_wave_barrier();
        // band update (:172-179)
        int nmin = max_k, nmax = min_k;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = lane + 64 * j;
            if (t < nslot) {
                const int k2 = min_k + 2 * t;
                if (S.U[k2 + k_offset] >= best_m - band_tol) { nmin = min(nmin, k2); nmax = max(nmax, k2); }
            }
        }
        for (int off = 32; off > 0; off >>= 1) { nmin = min(nmin, __shfl_xor(nmin, off)); nmax = max(nmax, __shfl_xor(nmax, off)); }
        max_k = nmax + 1;
        min_k = nmin - 1;
        if (hk != 0x7fffffff) { aligned = 1; end_x = hx; end_y = hx - hk; end_d = d; end_k = hk; break; }
    }
    o.trim_ok = 0; o.qcnt = o.tcnt = o.acnt = 0;
    if (!aligned) {
        if (best_x > 0) { end_x = best_x; end_y = best_y; end_d = best_d; end_k = best_k; }
        else { o.aligned_or_best = 0; o.qe = o.te = o.dist = 0; return; }
    }
    o.aligned_or_best = 1; o.qe = end_x; o.te = end_y; o.dist = end_d;
    // ---- tail traceback == trim_mismatch_end(.., 4, ..) on the alignment string (gapalign.cpp:47-68)
    const int aln_size = (end_x + end_y + end_d) / 2;
    int cd = end_d, ck = end_k, cx2 = end_x;
    int qcnt = 0, tcnt = 0, acnt = 0, found = 0;
    while (true) {
        int x1, pre_k = 0, takes_q = 0;
        if (cd == 0) x1 = 0;   // V is zero-filled: the d = 0 point starts at (0, 0)
        else {
            if (!grow && last_row - (cd - 1) >= RING) { o.fallback = 1; return; }
            const uint16_t* prow = grow ? grow + (size_t)(cd - 1) * ROW_W : S.ring + ((cd - 1) % RING) * ROW_W;
            const int pmin = S.rmin[cd - 1], pmax = S.rmax[cd - 1];
            const int cmin = S.rmin[cd], cmax = S.rmax[cd];
            // values the forward pass read: row d-1 inside its band, the zero fill outside (never hit in practice)
            const int kl = ck - 1, kr = ck + 1;
            const int vl = (kl >= pmin && kl <= pmax) ? (int)prow[(kl - pmin) >> 1] : 0;
            const int vr = (kr >= pmin && kr <= pmax) ? (int)prow[(kr - pmin) >> 1] : 0;
            if (ck == cmin || (ck != cmax && vl < vr)) { x1 = vr; pre_k = kr; takes_q = 0; }
            else { x1 = vl + 1; pre_k = kl; takes_q = 1; }
        }
        const int s = cx2 - x1;     // snake length
        if (s >= 4) { acnt += 4; qcnt += 4; tcnt += 4; found = 1; break; }
        acnt += s; qcnt += s; tcnt += s;
        if (cd == 0) break;
        acnt += 1;                  // the indel column
        if (takes_q) { qcnt += 1; cx2 = x1 - 1; } else { tcnt += 1; cx2 = x1; }
        ck = pre_k;
        --cd;
    }
    o.qcnt = qcnt; o.tcnt = tcnt; o.acnt = acnt;
    o.trim_ok = found && (aln_size - acnt >= 2);     // "m == mat_cnt && k > 0"
}

__global__ __launch_bounds__(AL_BLOCK) void dw_extend(const uint32_t* __restrict__ rpac, const mhip_offset_t* __restrict__ roffs,
                                                      const uint32_t* __restrict__ qpac, const mhip_offset_t* __restrict__ qoffs,
                                                      const mhip_aln_job* __restrict__ jobs, int n, DirResult* __restrict__ dres,
                                                      uint16_t* __restrict__ gscratch, unsigned int* __restrict__ cursor,
                                                      unsigned long long* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    AlnWaveLds& S = ((AlnWaveLds*)smem_raw)[threadIdx.x >> 6];
    const int lane = lane_id();
    const int gw = blockIdx.x * AL_WAVES + (threadIdx.x >> 6);
    uint16_t* grow = gscratch + (size_t)gw * ((size_t)MAX_D * ROW_W);
    unsigned long long cells = 0, snake = 0, nblocks = 0, nfallback = 0;
    while (true) {
        unsigned int unit = 0;
        if (lane == 0) unit = atomicAdd(cursor, 1u);
        unit = __shfl(unit, 0);
        if (unit >= 2u * (unsigned)n) break;
        const mhip_aln_job jb = jobs[unit >> 1];
        const int right = unit & 1;
        const int qsize = qoffs[jb.qid_local].size, tsize = roffs[jb.sid_local].size;
        SeqView q, t;
        q.pac = qpac; q.off = qoffs[jb.qid_local].offset; q.len = qsize; q.rc = jb.chain;
        t.pac = rpac; t.off = roffs[jb.sid_local].offset; t.len = tsize; t.rc = 0;
        int query_size, target_size;
        if (right) { q.start = jb.qstart; q.step = 1; t.start = jb.sstart; t.step = 1; query_size = qsize - jb.qstart; target_size = tsize - jb.sstart; }
        else { q.start = jb.qstart - 1; q.step = -1; t.start = jb.sstart - 1; t.step = -1; query_size = jb.qstart; target_size = jb.sstart; }
        int qidx = 0, tidx = 0;
        DirResult R = {0, 0, 0, 0, 0, 0};
        while (true) {
            // retrieve_next_aln_block (gapalign.cpp:9-45)
            const int qleft = query_size - qidx, tleft = target_size - tidx;
            int qblk, tblk, last_block;
            if (qleft < SEG_BLK + 100 || tleft < SEG_BLK + 100) {
                qblk = min(qleft, (int)(tleft + tleft * 0.2));
                tblk = min(tleft, (int)(qleft + qleft * 0.2));
                last_block = 1;
            } else { qblk = SEG_BLK; tblk = SEG_BLK; last_block = 0; }
            if (qblk < 0) qblk = 0;
            if (tblk < 0) tblk = 0;
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < qblk; i += 64) S.Q[i] = (uint8_t)seq_at(q, qidx + i);
            for (int i = lane; i < tblk; i += 64) S.T[i] = (uint8_t)seq_at(t, tidx + i);
            __builtin_amdgcn_wave_barrier();
            BlockOut o;
            align_block(S, qblk, tblk, nullptr, o, cells, snake);
            if (o.fallback) { ++nfallback; align_block(S, qblk, tblk, grow, o, cells, snake); }
            ++nblocks;
            R.blocks += 1;
            if (!o.aligned_or_best || !o.trim_ok) break;
            const int full_map = (qblk - o.qe <= 20 || tblk - o.te <= 20);
            int qcnt = o.qcnt, tcnt = o.tcnt, acnt = o.acnt;
            if (last_block || !full_map) { qcnt -= 4; tcnt -= 4; acnt -= 4; }
            const int cols = (o.qe + o.te + o.dist) / 2 - acnt;
            const int mcut = qcnt + tcnt - acnt;
            R.columns += cols;
            R.matches += (o.qe + o.te - o.dist) / 2 - mcut;
            R.qbases += o.qe - qcnt;
            R.tbases += o.te - tcnt;
            if (last_block || !full_map) break;
            qidx += o.qe - qcnt;
            tidx += o.te - tcnt;
        }
        if (lane == 0) dres[unit] = R;
    }
    if (lane == 0) {
        atomicAdd(&counters[3], nblocks);
        atomicAdd(&counters[4], cells);
        atomicAdd(&counters[8], nfallback);       // debug slot: blocks re-run with global rows
    }
    // snake bases are per lane
    for (int off = 32; off > 0; off >>= 1) snake += __shfl_xor(snake, off);
    if (lane == 0) atomicAdd(&counters[5], snake);
}

// stitch the two directions (diff_gapalign.cpp:309-348)
__global__ void dw_stitch(const mhip_aln_job* __restrict__ jobs, const DirResult* __restrict__ dres, int n, int min_aln,
                          mhip_aln_result* __restrict__ out, unsigned long long* __restrict__ counters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DirResult L = dres[2 * i], R = dres[2 * i + 1];
    mhip_aln_result r;
    r.query_start = jobs[i].qstart - L.qbases;
    r.target_start = jobs[i].sstart - L.tbases;
    r.query_end = jobs[i].qstart + R.qbases;
    r.target_end = jobs[i].sstart + R.tbases;
    r.matches = L.matches + R.matches;
    r.columns = L.columns + R.columns;
    r.blocks = L.blocks + R.blocks;
    r.ok = r.columns >= min_aln;
    out[i] = r;
    if (r.ok) {
        atomicAdd(&counters[6], (unsigned long long)(r.query_end - r.query_start));
        atomicAdd(&counters[7], 1ull);
    }
}

// candidate table -> job list (pw_impl.cpp:674-686).  Single block: running exclusive scan of the counts over tiles.
__global__ __launch_bounds__(1024) void dw_make_jobs(const mhip_candidate* __restrict__ cands, const int32_t* __restrict__ counts,
                                                     int n_reads, int maxc, int rid_begin, int rid_stride, int ref_start_id,
                                                     int part_index, int part_count, mhip_aln_job* __restrict__ jobs,
                                                     int* __restrict__ num_jobs) {
    __shared__ unsigned int wtot[16];
    __shared__ unsigned int carry_all, carry_mine;
    if (threadIdx.x == 0) { carry_all = 0; carry_mine = 0; }
    __syncthreads();
    for (int t0 = 0; t0 < n_reads; t0 += 1024) {
        const int i = t0 + threadIdx.x;
        const unsigned int c = i < n_reads ? (unsigned int)counts[i] : 0u;
        unsigned int incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            unsigned int v = __shfl_up(incl, o);
            if (lane_id() >= o) incl += v;
        }
        if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned int base = carry_all;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wtot[w];
        const unsigned int first = base + incl - c;          // global index of this read's first candidate
        for (unsigned int j = 0; j < c; ++j) {
            const unsigned int g = first + j;
            if (part_count > 1 && (int)(g % (unsigned)part_count) != part_index) continue;
            const unsigned int slot = part_count > 1 ? g / (unsigned)part_count : g;
            const mhip_candidate cd = cands[(size_t)i * maxc + j];
            mhip_aln_job jb;
            jb.qid_local = rid_begin + i * rid_stride;
            jb.sid_local = cd.readno - ref_start_id;
            jb.chain = cd.chain;
            int qstart = cd.loc2, sstart = cd.loc1;
            if (qstart && sstart) { qstart += MHIP_KMER_SIZE / 2; sstart += MHIP_KMER_SIZE / 2; }
            jb.qstart = qstart;
            jb.sstart = sstart;
            jobs[slot] = jb;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_all = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned int tot = carry_all;
        unsigned int mine = tot;
        if (part_count > 1) mine = tot / (unsigned)part_count + ((unsigned)part_index < tot % (unsigned)part_count ? 1u : 0u);
        *num_jobs = (int)mine;
        (void)carry_mine;
    }
}

extern "C" {

int mhip_jobs_from_candidates_dev(mhip_ctx* c, const void* d_cands, const void* d_counts, int n_reads, int maxc, int rid_begin,
                                  int rid_stride, int ref_start_read_id, int part_index, int part_count, void* d_jobs, int* num_jobs) {
    HIPCHK(hipSetDevice(c->device));
    *num_jobs = 0;
    if (n_reads <= 0) return 0;
    if (part_count < 1) part_count = 1;
    int* d_n;
    if (c->scratch("al_njobs", 64, (void**)&d_n)) return -1;
    LAUNCH(c, "dw_make_jobs", dw_make_jobs, 1, 1024, 0, (const mhip_candidate*)d_cands, (const int32_t*)d_counts, n_reads, maxc,
           rid_begin, rid_stride, ref_start_read_id, part_index, part_count, (mhip_aln_job*)d_jobs, d_n);
    HIPCHK(hipMemcpyAsync(num_jobs, d_n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int mhip_align_candidates_dev(mhip_ctx* c, const mhip_volume* ref, const mhip_volume* reads, const void* d_jobs, int n,
                              int min_align_size, void* d_out) {
    HIPCHK(hipSetDevice(c->device));
    if (n <= 0) return 0;
    const int waves_per_cu = 12;                        // LDS: 3 blocks x 4 waves x 13.4 KB per CU
    int grid = c->num_cus * waves_per_cu / AL_WAVES;
    grid = std::min(grid, (2 * n + AL_WAVES - 1) / AL_WAVES);
    DirResult* d_dres;
    uint16_t* d_g;
    unsigned int* d_cur;
    if (c->scratch("al_dres", sizeof(DirResult) * 2 * (size_t)n, (void**)&d_dres)) return -1;
    if (c->scratch("al_rows", sizeof(uint16_t) * (size_t)MAX_D * ROW_W * (size_t)(c->num_cus * waves_per_cu), (void**)&d_g)) return -1;
    if (c->scratch("al_cursor", 64, (void**)&d_cur)) return -1;
    HIPCHK(hipMemsetAsync(d_cur, 0, 4, c->stream));
    const size_t lds = sizeof(AlnWaveLds) * AL_WAVES;
    LAUNCH(c, "dw_extend", dw_extend, grid, AL_BLOCK, lds, (const uint32_t*)ref->d_pac, (const mhip_offset_t*)ref->d_offs,
           (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, (const mhip_aln_job*)d_jobs, n, d_dres, d_g, d_cur,
           (unsigned long long*)c->d_counters);
    LAUNCH(c, "dw_stitch", dw_stitch, (n + 255) / 256, 256, 0, (const mhip_aln_job*)d_jobs, (const DirResult*)d_dres, n,
           min_align_size, (mhip_aln_result*)d_out, (unsigned long long*)c->d_counters);
    HIPCHK(hipGetLastError());
    return 0;
}

int mhip_align_candidates(mhip_ctx* c, const mhip_volume* ref, const mhip_volume* reads, const mhip_aln_job* jobs, int n,
                          int min_align_size, mhip_aln_result* out) {
    HIPCHK(hipSetDevice(c->device));
    if (n <= 0) return 0;
    for (int i = 0; i < n; ++i) {
        const mhip_aln_job& j = jobs[i];
        if (j.qid_local < 0 || j.qid_local >= reads->num_reads || j.sid_local < 0 || j.sid_local >= ref->num_reads) {
            mhip_set_error("alignment job %d: read index out of range", i);
            return -1;
        }
        const int qs = reads->h_offs[(size_t)j.qid_local].size, ts = ref->h_offs[(size_t)j.sid_local].size;
        if (j.qstart < 0 || j.qstart > qs || j.sstart < 0 || j.sstart > ts) {
            mhip_set_error("alignment job %d: start point outside the reads", i);
            return -1;
        }
    }
    mhip_aln_job* d_jobs;
    mhip_aln_result* d_out;
    if (c->scratch("al_jobs", sizeof(mhip_aln_job) * (size_t)n, (void**)&d_jobs)) return -1;
    if (c->scratch("al_out", sizeof(mhip_aln_result) * (size_t)n, (void**)&d_out)) return -1;
    HIPCHK(hipMemcpyAsync(d_jobs, jobs, sizeof(mhip_aln_job) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if (mhip_align_candidates_dev(c, ref, reads, d_jobs, n, min_align_size, d_out)) return -1;
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(mhip_aln_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
