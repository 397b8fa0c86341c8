// asm_seed.hip — seeding + candidate selection of mecat2canu's overlapper for corrected reads.
//
// Replaces, for one block of reads indexed with mhip_index_build_ex(.., 256, ..) and a batch of query reads, the part of
// pairwise_mapping in front of its extension loop (/root/reference/mecat2canu/src/mecat2asmpw/mecat2asmpw.c:580-718; same in
// mecat2trimpw.c): the same algorithm family as mecat2pw's seeding, in the older single-file code base, with
//   segments of 1 000 text positions, 60 stored seeds per segment and NO overflow replacement (the insert_loc call is commented out,
//   :621: a segment past 60 seeds only counts), positions 1-based (:503), gate `index_score > 10` (:644), second gate `< 6` (:662),
//   find_location at a 0.10 cutoff (:355-386), the subject read by its own bisection (binary, :283-297; the entry behind the last
//   read start is never written by the tool and reads as 0), sweeps that start two segments to the left / one to the right and run
//   over `score` entries although 60 exist (:689-703).
// That last point decides the data layout: the sweeps read past loczhi[60] / seedno[60] into the rest of the segment record and
// into the following records, so the segments are kept as the reference's own `struct Back_List` images (124 shorts, 248 bytes) in
// one dense array per wave in HBM (block bases / 1000 + 5 records: 248 KB per million bases; 288 GB of HBM is what lets every
// resident wave own one), and the sweeps index it as a flat array of shorts exactly like the reference's out-of-range indices do.
// Every read starts from an all-zero array (the reference's worker threads keep the stale seeds of the reads they mapped before,
// which can move a candidate's score by a few votes: INTEGRATION.md, divergences); the forward strand's leftovers stay for
// the reverse strand, as there.
//
// Mapping: one wave per query read, persistent, pulling reads from an atomic cursor.  Everything order dependent runs in the
// reference's order with the wave's lanes inside each step:
//   seeding        the bucket of a query k-mer 64 entries at a time (ascending positions: the entries of one segment are adjacent
//                  lanes; the first lane of a run is the hit that can be a seed), score / stored seed / seednum updates by those lanes,
//                  then index_score = own + left neighbour's score as of that moment (the neighbour's update of the same step comes
//                  from a lower lane and is already written), first-touch order by a ballot prefix
//   candidates     the touched segments one after the other (their state changes under the loop: self-hit scrub, neighbours zeroed
//                  by the sweeps); find_location's vote with one list entry per lane and the per-entry `tempi` chain in registers,
//                  the sweeps 64 entries per step, the top-100 list in LDS
// Parity: tests/test_gpu_asmpw.py against the CPU restatement of the test-suite (fresh mode; itself pinned to the unmodified
// pairwise_mapping), and the whole tool against the sorted output of the unmodified binaries.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

#define AZV 1000
#define ASM 60
#define AMAXC 100
#define AREC 124                     // shorts of one struct Back_List: score, loczhi[60], seedno[60], seednum, index (two shorts)
#define A_SCORE 0
#define A_LOC 1
#define A_SEED (1 + ASM)
#define A_SEEDNUM (1 + 2 * ASM)
#define A_INDEX (2 + 2 * ASM)        // int at short offset 122 (byte 244)
#define ASM_BLOCK 256
#define ASM_WAVES (ASM_BLOCK / WAVE)

namespace {

struct AsmWaveLds {
    int tl[128], ts[128], tsc[128];
    mhip_asm_candidate cand[AMAXC];
};

typedef volatile short vshort;       // the segment records are re-read after other lanes wrote them: no L1 / register caching

__device__ __forceinline__ int rec_index(vshort* r) { return (int)(uint16_t)r[A_INDEX] | ((int)r[A_INDEX + 1] << 16); }
__device__ __forceinline__ void rec_set_index(vshort* r, int v) { r[A_INDEX] = (short)(v & 0xffff); r[A_INDEX + 1] = (short)(v >> 16); }

// the reference's bisection over the read starts (binary, mecat2asmpw.c:283-297), statement for statement: its returns decide
// which read a position belongs to.  lloc(i) = start of read i for i < n, 0 for i == n (the entry the tool never writes).
template <typename F>
__device__ __forceinline__ int asm_find_read(F lloc, int key, int n) {
    int left = 0, right = n - 1, mid = (left + right) / 2;
    if (lloc(right) < key) return right;
    while (left <= right && lloc(mid) != key) {
        if (lloc(mid) < key && lloc(mid + 1) > key) return mid;
        if (lloc(mid) < key && lloc(mid + 1) == key) return mid + 1;
        else if (lloc(mid) < key && lloc(mid + 1) < key) left = mid + 1;
        else if (lloc(mid) > key && lloc(mid - 1) <= key) return mid - 1;
        else if (lloc(mid) > key && lloc(mid - 1) > key) right = mid - 1;
        mid = (left + right) / 2;
    }
    return mid;
}

// all-f32 DDF test of find_location with the cutoff compared in f64 (mecat2asmpw.c:358: fabs(float expression) < 0.10)
__device__ __forceinline__ bool asm_ddf(int dloc, int dseed) {
    const float r = (float)dloc / ((float)dseed * 10.0f) - 1.0f;
    return (double)fabsf(r) < 0.10;
}

__global__ void asm_init_records(short* img, size_t nrec) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrec) return;
    img[i * AREC + A_INDEX] = (short)-1;
    img[i * AREC + A_INDEX + 1] = (short)-1;
}

__global__ __launch_bounds__(ASM_BLOCK) void asm_seed(const uint32_t* __restrict__ bpac, const mhip_offset_t* __restrict__ boffs, int bnreads,
                                                     int bstart_id, const uint32_t* __restrict__ starts, const int32_t* __restrict__ offsets,
                                                     const uint32_t* __restrict__ qpac, const uint32_t* __restrict__ qnpac, const mhip_offset_t* __restrict__ qoffs, int qstart_id,
                                                     int rid_begin, int n, short* __restrict__ img_all, int* __restrict__ ilist_all,
                                                     short* __restrict__ iscore_all, int* __restrict__ dirty_all, int nseg, int gate, int maxc,
                                                     unsigned int* __restrict__ cursor, mhip_asm_candidate* __restrict__ out, int32_t* __restrict__ out_counts) {
    __shared__ AsmWaveLds lds[ASM_WAVES];
    AsmWaveLds& S = lds[threadIdx.x >> 6];
    const int lane = lane_id();
    const size_t gw = (size_t)blockIdx.x * ASM_WAVES + (threadIdx.x >> 6);
    vshort* img = img_all + gw * (size_t)(nseg + 8) * AREC;
    int* ilist = ilist_all + gw * (size_t)nseg;
    volatile short* iscore = iscore_all + gw * (size_t)nseg;
    int* dirty = dirty_all + gw * (size_t)nseg * 2;
    const unsigned long long below = (1ull << lane) - 1ull;
    auto lloc = [&](int i) { return i < bnreads ? boffs[i].offset : 0; };
    auto rec = [&](int seg) { return img + (size_t)seg * AREC; };

    while (true) {
        unsigned int u = 0;
        if (lane == 0) u = atomicAdd(cursor, 1u);
        u = __shfl(u, 0);
        if (u >= (unsigned)n) break;
        const int rid = rid_begin + (int)u;
        const int L = qoffs[rid].size, read_name = qstart_id + rid;
        const int64_t qoff = qoffs[rid].offset;
        const int K = L < MHIP_KMER_SIZE ? 0 : (L - MHIP_KMER_SIZE) / BC + 1;
        int ncand = 0, ndirty = 0;
        for (int strand = 0; strand < 2; ++strand) {
            int touched = 0;
            // ---- seeding (:601-641).  What a step looks up (the query k-mer, its bucket's bounds, the bucket's first 64 positions) does not
            // depend on the records, only the updates are order dependent: the bounds are fetched two steps ahead and the positions one step
            // ahead, so their memory round trips run beside the record round trips of the step in hand instead of in front of them.
            auto bounds = [&](int k, uint32_t& s0, uint32_t& s1) {
                s0 = s1 = 0;
                if (k >= K) return;
                const int64_t kpos = !strand ? qoff + (int64_t)k * BC : qoff + L - MHIP_KMER_SIZE - (int64_t)k * BC;
                // a query k-mer that holds a base other than A, C, G, T is not looked up (transnum_buchang gives it -1, :316-335, :601)
                if (qnpac && pac_kmer(qnpac, kpos) != 0u) return;
                const uint32_t id = !strand ? pac_kmer(qpac, kpos) : kmer_revcomp(pac_kmer(qpac, kpos));
                s0 = starts[id];
                s1 = starts[id + 1];
            };
            uint32_t s0, s1, n0, n1;
            bounds(0, s0, s1);
            bounds(1, n0, n1);
            int pfirst = s0 + (uint32_t)lane < s1 ? offsets[s0 + (uint32_t)lane] : 0;
            for (int k = 0; k < K; ++k) {
                const int pnext = n0 + (uint32_t)lane < n1 ? offsets[n0 + (uint32_t)lane] : 0;
                uint32_t m0, m1;
                bounds(k + 2, m0, m1);
                int carry_seg = -1;
                for (uint32_t base = s0; base < s1; base += 64) {
                    const uint32_t e = base + (uint32_t)lane;
                    const bool valid = e < s1;
                    const int p = valid ? (base == s0 ? pfirst : offsets[e]) + 1 : 0;          // 1-based text position (:503)
                    const int seg = valid ? p / AZV : -2, off = p % AZV;
                    int seg_prev = __shfl_up(seg, 1);
                    if (lane == 0) seg_prev = carry_seg;
                    carry_seg = __shfl(seg, 63);
                    const bool head = valid && seg != seg_prev;        // the first hit of this k-mer in its segment
                    bool ev = false;
                    int newscore = 0;
                    if (head) {
                        vshort* r = rec(seg);
                        const int sc = r[A_SCORE], sn = r[A_SEEDNUM];
                        ev = sc == 0 || sn < k + 1;
                        if (ev) {
                            newscore = sc + 1;
                            r[A_SCORE] = (short)newscore;
                            if (newscore <= ASM) { r[A_LOC + newscore - 1] = (short)off; r[A_SEED + newscore - 1] = (short)(k + 1); }
                        }
                        r[A_SEEDNUM] = (short)(k + 1);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    int idx = 0;
                    if (ev) idx = rec_index(rec(seg));
                    const unsigned long long fresh = __ballot(ev && idx == -1);
                    if (ev) {
                        const int sk = seg > 0 ? newscore + (int)rec(seg - 1)[A_SCORE] : newscore;
                        if (idx == -1) {
                            const int t = touched + (int)__popcll(fresh & below);
                            ilist[t] = seg;
                            iscore[t] = (short)sk;
                            rec_set_index(rec(seg), t);
                        } else iscore[idx] = (short)sk;
                    }
                    touched += (int)__popcll(fresh);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                s0 = n0; s1 = n1; pfirst = pnext;
                n0 = m0; n1 = m1;
            }
            // ---- candidates (:643-716)
            for (int i = 0; i < touched; ++i) {
                const int seg = ((volatile int*)ilist)[i];
                if (!(iscore[i] > gate)) continue;        // :644: > 10 (mecat2asmpw), > 8 (mecat2trimpw)
                vshort* g = rec(seg);
                const int score = g[A_SCORE];
                if (score == 0) continue;
                const int prev = seg > 0 ? (int)rec(seg - 1)[A_SCORE] : 0;
                const int start_loc = prev > 0 ? (seg - 1) * AZV : seg * AZV;
                const int np = prev > 0 ? min(prev, ASM) : 0, ns = min(score, ASM), nl = np + ns;
                for (int j = lane; j < nl; j += 64) {
                    if (j < np) { S.tl[j] = rec(seg - 1)[A_LOC + j]; S.ts[j] = rec(seg - 1)[A_SEED + j]; }
                    else { S.tl[j] = g[A_LOC + j - np] + (prev > 0 ? AZV : 0); S.ts[j] = g[A_SEED + j - np]; }
                    S.tsc[j] = 0;
                }
                __builtin_amdgcn_wave_barrier();
                // find_location (:355-386): entry i per lane, its j loop in order (the `tempi` chain), votes for j by LDS atomics
                for (int i0 = 0; i0 < nl - 1; i0 += 64) {
                    const int ii = i0 + lane;
                    if (ii < nl - 1) {
                        const int li = S.tl[ii], si = S.ts[ii];
                        int tempi = si, mine = 0;
                        for (int j = ii + 1; j < nl; ++j) {
                            const int lj = S.tl[j], sj = S.ts[j];
                            if (tempi != sj && sj - si > 0 && lj - li > 0 && lj - li < L && asm_ddf(lj - li, sj - si)) {
                                ++mine;
                                atomicAdd(&S.tsc[j], 1);
                                tempi = sj;
                            }
                        }
                        if (mine) atomicAdd(&S.tsc[ii], mine);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                int flag = 0, rep_loc = 0, loc0 = 0, loc1 = 0;
                {
                    int maxval = 0, maxi = 0, rep = 0;
                    for (int j = 0; j < nl; ++j) {
                        const int v = S.tsc[j];
                        if (maxval < v) { maxval = v; maxi = j; rep = 0; }
                        else if (maxval == v) ++rep;
                    }
                    if (maxval >= 5 && rep == maxval) { flag = 1; loc0 = S.tl[maxi]; loc1 = S.ts[maxi]; rep_loc = maxi; }
                    else if (maxval >= 5) {
                        // the entries that agree with the best one, in index order (before it, itself, after it): each is taken as
                        // the start point while none is held — "none" is `loc[0] == 0` in the reference, so an entry at list position 0
                        // does not hold the place (:371-383)
                        flag = 1;
                        const int lm = S.tl[maxi], sm = S.ts[maxi];
                        for (int j = 0; j < maxi; ++j) {
                            const int lj = S.tl[j], sj = S.ts[j];
                            if (sm - sj > 0 && lm - lj > 0 && lm - lj < L && asm_ddf(lm - lj, sm - sj) && loc0 == 0) { loc0 = lj; loc1 = sj; rep_loc = j; }
                        }
                        if (loc0 == 0) { loc0 = lm; loc1 = sm; rep_loc = maxi; }
                        for (int j = maxi + 1; j < nl; ++j) {
                            const int lj = S.tl[j], sj = S.ts[j];
                            if (sj - sm > 0 && lj - lm > 0 && lj - lm <= L && asm_ddf(lj - lm, sj - sm) && loc0 == 0) { loc0 = lj; loc1 = sj; rep_loc = j; }
                        }
                    }
                }
                if (!flag) continue;
                const int vote = S.tsc[rep_loc];
                if (vote < 6) continue;
                const int loc_seed = S.ts[rep_loc];
                loc0 += start_loc;
                const int loc_list = loc0;
                const int rd = asm_find_read(lloc, loc0, bnreads);
                const int rstart = lloc(rd), rend = lloc(rd + 1);
                const int subj = bstart_id + rd;
                if (subj > read_name) continue;
                if (subj == read_name) {
                    // the read meets itself: its own stretch of the block is scrubbed (:651-658); lane 0 (short, sequential)
                    if (lane == 0) {
                        int useg = rstart / AZV, cut = rstart % AZV, kk = 0;
                        vshort* h = rec(useg);
                        int sc = h[A_SCORE];
                        for (int j = 0; j < sc && j < ASM; ++j) if (h[A_LOC + j] < cut) { h[A_LOC + kk] = h[A_LOC + j]; ++kk; }
                        h[A_SCORE] = (short)kk;
                        const int last = rend / AZV;
                        for (++useg; useg < last; ++useg) rec(useg)[A_SCORE] = 0;
                        h = rec(useg);
                        sc = h[A_SCORE];
                        cut = rend % AZV;
                        kk = 0;
                        for (int j = 0; j < sc && j < ASM; ++j) if (h[A_LOC + j] > cut) { h[A_LOC + kk] = h[A_LOC + j]; ++kk; }
                        h[A_SCORE] = (short)kk;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    continue;
                }
                mhip_asm_candidate c;
                loc1 = (loc1 - 1) * BC;
                c.readno = rd; c.readstart = rstart;
                c.left1 = loc0 - rstart + MHIP_KMER_SIZE - 1; c.right1 = rend - loc0;
                c.left2 = loc1 + MHIP_KMER_SIZE - 1; c.right2 = L - loc1;
                c.num1 = c.left1 >= c.left2 ? c.left2 : c.left1;
                c.num2 = c.right1 >= c.right2 ? c.right2 : c.right1;
                if (c.num1 + c.num2 < 400) continue;
                c.loc1 = loc0; c.loc2 = loc1;
                int seedcount = 0;
                // sweeps (:689-703): entries j < score of the neighbour's record, read as a flat array of shorts
                for (int useg = seg - 2, kk = c.num1 / AZV; useg >= 0 && kk >= 0; --useg, --kk) {
                    vshort* o = rec(useg);
                    const int osc = o[A_SCORE];
                    if (osc <= 0) continue;
                    const int sl = useg * AZV;
                    int agree = 0;
                    const int jend = min(osc, (nseg + 8 - useg) * AREC - A_SEED);      // entries inside this wave's image (the reference reads its own heap beyond)
                    for (int j0 = 0; j0 < jend; j0 += 64) {
                        const int j = j0 + lane;
                        bool q = false;
                        if (j < jend) q = fabs((double)(loc_list - sl - (int)o[A_LOC + j]) / ((double)(loc_seed - (int)o[A_SEED + j]) * BC * 1.0) - 1.0) < 0.10;
                        agree += (int)__popcll(__ballot(q));
                    }
                    seedcount += agree;
                    if (agree * 1.0 / osc > 0.4 && lane == 0) o[A_SCORE] = 0;
                }
                for (int useg = seg + 1, kk = c.num2 / AZV; kk > 0; ++useg, --kk) {
                    vshort* o = rec(useg);
                    const int osc = o[A_SCORE];
                    if (osc <= 0) continue;
                    const int sl = useg * AZV;
                    int agree = 0;
                    const int jend = min(osc, (nseg + 8 - useg) * AREC - A_SEED);      // (as above)
                    for (int j0 = 0; j0 < jend; j0 += 64) {
                        const int j = j0 + lane;
                        bool q = false;
                        if (j < jend) q = fabs((double)(sl + (int)o[A_LOC + j] - loc_list) / ((double)((int)o[A_SEED + j] - loc_seed) * BC * 1.0) - 1.0) < 0.10;
                        agree += (int)__popcll(__ballot(q));
                    }
                    seedcount += agree;
                    if (agree * 1.0 / osc > 0.4 && lane == 0) o[A_SCORE] = 0;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                c.score = vote + seedcount;
                c.chain = strand;
                // stable insertion into the descending top-100 list (:707-716)
                int lo = 0, hi = ncand - 1;
                while (lo <= hi) {
                    const int mid = (lo + hi) / 2;
                    if (mid >= ncand || S.cand[mid].score < c.score) hi = mid - 1; else lo = mid + 1;
                }
                const int top = ncand < maxc ? ncand - 1 : ncand - 2;        // last entry that moves one place down
                __builtin_amdgcn_wave_barrier();
                // (entries hi + 1 .. top move to hi + 2 .. top + 1: every lane reads its entries first, then writes)
                mhip_asm_candidate mv[2];
                bool has[2];
                for (int q = 0; q < 2; ++q) {
                    const int j = hi + 1 + lane + 64 * q;
                    has[q] = j <= top;
                    if (has[q]) mv[q] = S.cand[j];
                }
                __builtin_amdgcn_wave_barrier();
                for (int q = 0; q < 2; ++q) {
                    const int j = hi + 1 + lane + 64 * q;
                    if (has[q]) S.cand[j + 1] = mv[q];
                }
                if (lane == 0 && hi + 1 < maxc) S.cand[hi + 1] = c;
                if (ncand < maxc) ++ncand;
                __builtin_amdgcn_wave_barrier();
            }
            // ---- the strand's touched segments: score and index reset (:717); remembered for the clean-up behind the read
            for (int t = lane; t < touched; t += 64) {
                const int seg = ((volatile int*)ilist)[t];
                rec(seg)[A_SCORE] = 0;
                rec_set_index(rec(seg), -1);
                dirty[ndirty + t] = seg;
            }
            ndirty += touched;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        // every read starts from an all-zero array: the records both strands touched are cleared whole
        for (int t = 0; t < ndirty; ++t) {
            vshort* r = rec(((volatile int*)dirty)[t]);
            for (int j = lane; j < AREC - 2; j += 64) r[j] = 0;
            if (lane == 0) rec_set_index(r, -1);
        }
        for (int j = lane; j < ncand; j += 64) out[(size_t)u * AMAXC + j] = S.cand[j];
        if (lane == 0) out_counts[u] = ncand;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

}  // namespace

extern "C" {

int mhip_asm_seed_reads(mhip_ctx* c, const mhip_index* idx, const mhip_volume* block, const mhip_volume* reads, int rid_begin, int rid_end,
                        mhip_asm_candidate* out, int32_t* out_counts) {
    return mhip_asm_seed_reads_ex(c, idx, block, reads, rid_begin, rid_end, 10, AMAXC, out, out_counts);
}

int mhip_asm_seed_reads_ex(mhip_ctx* c, const mhip_index* idx, const mhip_volume* block, const mhip_volume* reads, int rid_begin, int rid_end,
                           int gate, int maxc, mhip_asm_candidate* out, int32_t* out_counts) {
    HIPCHK(hipSetDevice(c->device));
    if (maxc < 1 || maxc > AMAXC || gate < 0) { mhip_set_error("bad gate %d / list length %d (1..100)", gate, maxc); return -1; }
    if (rid_begin < 0 || rid_end > reads->num_reads || rid_begin > rid_end) { mhip_set_error("bad read range [%d,%d)", rid_begin, rid_end); return -1; }
    if (idx->max_bucket != 256) { mhip_set_error("mhip_asm_seed_reads needs an index built with bucket cap 256 (mhip_index_build_ex), not %d", idx->max_bucket); return -1; }
    if (idx->num_bases != block->num_bases) { mhip_set_error("the index was not built from this block"); return -1; }
    const int n = rid_end - rid_begin;
    if (n == 0) return 0;
    const int nseg = block->num_bases / AZV + 5;
    // one dense array of segment records per resident wave: as many waves as fit a budget of the free memory, at most 16 per CU
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t per_wave = (size_t)(nseg + 8) * AREC * sizeof(short) + (size_t)nseg * (sizeof(int) + sizeof(short) + 2 * sizeof(int));
    // (measured on 80 Mbase blocks, 20 MB of records per wave: 24 GB and 48 GB of records are equally fast, 12 GB 45 % slower; with 48 GB
    // one run in three took four times as long, with 80 GB every run — the random record updates then miss every level of address
    // translation.  MECAT_ASM_RECORDS_GB overrides the 24.)
    const char* gb = getenv("MECAT_ASM_RECORDS_GB");
    const size_t budget = std::min<size_t>(free_b / 2, (size_t)(gb && atoi(gb) > 0 ? atoi(gb) : 24) << 30);
    size_t waves = std::min<size_t>((size_t)c->num_cus * 16, std::max<size_t>(1, budget / per_wave));
    waves = std::min<size_t>(waves, (size_t)n);
    const unsigned grid = (unsigned)((waves + ASM_WAVES - 1) / ASM_WAVES);
    waves = (size_t)grid * ASM_WAVES;
    short *d_img, *d_iscore;
    int *d_ilist, *d_dirty;
    unsigned int* d_cur;
    mhip_asm_candidate* d_out;
    int32_t* d_cnt;
    if (c->scratch("as_img", sizeof(short) * waves * (size_t)(nseg + 8) * AREC, (void**)&d_img)) return -1;
    if (c->scratch("as_ilist", sizeof(int) * waves * (size_t)nseg, (void**)&d_ilist)) return -1;
    if (c->scratch("as_iscore", sizeof(short) * waves * (size_t)nseg, (void**)&d_iscore)) return -1;
    if (c->scratch("as_dirty", sizeof(int) * waves * (size_t)nseg * 2, (void**)&d_dirty)) return -1;
    if (c->scratch("as_cursor", 64, (void**)&d_cur)) return -1;
    if (c->scratch("as_out", sizeof(mhip_asm_candidate) * (size_t)n * AMAXC, (void**)&d_out)) return -1;
    if (c->scratch("as_cnt", sizeof(int32_t) * (size_t)n, (void**)&d_cnt)) return -1;
    const size_t nrec = waves * (size_t)(nseg + 8);
    // The records are initialised (all zero, index -1) when the buffer is new or a call needs more of them than were initialised: the
    // kernel clears what a read touched behind the read, so a completed call leaves the image as it found it — whatever the block's
    // segment count was (the image is one flat array of records; only the waves' shares of it move).  It was 24 GB of fill per call.
    if (d_img != c->as_clean_base || nrec > c->as_clean_nrec) {
        c->as_clean_base = nullptr;
        HIPCHK(hipMemsetAsync(d_img, 0, sizeof(short) * nrec * AREC, c->stream));
        LAUNCH(c, "asm_init_records", asm_init_records, (unsigned)((nrec + 255) / 256), 256, 0, d_img, nrec);
    }
    const size_t nrec_clean = std::max(nrec, c->as_clean_base ? c->as_clean_nrec : (size_t)0);
    c->as_clean_base = nullptr;               // (until this call has completed)
    HIPCHK(hipMemsetAsync(d_cur, 0, 16, c->stream));
    LAUNCH(c, "asm_seed", asm_seed, grid, ASM_BLOCK, 0, (const uint32_t*)block->d_pac, (const mhip_offset_t*)block->d_offs, block->num_reads,
           block->start_read_id, (const uint32_t*)idx->d_starts, (const int32_t*)idx->d_offsets, (const uint32_t*)reads->d_pac, (const uint32_t*)reads->d_npac,
           (const mhip_offset_t*)reads->d_offs, reads->start_read_id, rid_begin, n, d_img, d_ilist, d_iscore, d_dirty, nseg, gate, maxc, d_cur, d_out, d_cnt);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_counts, d_cnt, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(mhip_asm_candidate) * (size_t)n * AMAXC, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->as_clean_base = d_img;
    c->as_clean_nrec = nrec_clean;
    return 0;
}

}  // extern "C"
