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
// into the following records, so the segments are kept as the reference's own `struct Back_List` images (124 shorts, 248 bytes) and the
// sweeps index them as a flat array of shorts exactly like the reference's out-of-range indices do.  The array is VIRTUAL: a read touches
// one or two thousand of a block's 80 000 segments, and a dense 20 MB image per wave meant every record access missed every level of
// address translation (24 GB of images: 2.6 us per dependent access).  Each wave has a directory (one 32-bit slot number per segment, 0 =
// untouched) and a pool of records handed out in order of first touch; slot 0 is the untouched record (all zero, index -1), never
// written with anything else, so reads need no branch and the zero-over-zero writes the algorithm makes to untouched segments land there.
// A wave's working set is its directory (320 KB) plus the first few hundred KB of its pool, whatever the block size.  A pool holds
// ASM_POOL records; a read that needs more gives up and is redone by a second launch whose pools hold one record per segment.
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
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

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
#define ASM_POOL 8192                 // records of a wave's pool: 2 MB, a read of 30 kb (one that needs more is redone with a worst-case pool)
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

__global__ __launch_bounds__(ASM_BLOCK) void asm_seed(const uint32_t* __restrict__ bpac, const mhip_offset_t* __restrict__ boffs, int bnreads,
                                                     int bstart_id, const uint32_t* __restrict__ starts, const int32_t* __restrict__ offsets,
                                                     const uint32_t* __restrict__ qpac, const uint32_t* __restrict__ qnpac, const mhip_offset_t* __restrict__ qoffs, int qstart_id,
                                                     int rid_begin, int n, const int* __restrict__ rid_list, short* __restrict__ pool_all, uint32_t* __restrict__ dir_all,
                                                     int* __restrict__ hw_all, int* __restrict__ ilist_all, short* __restrict__ iscore_all, int* __restrict__ slotseg_all,
                                                     int nseg, int pcap, int gate, int maxc, unsigned int* __restrict__ cursor, mhip_asm_candidate* __restrict__ out,
                                                     int32_t* __restrict__ out_counts) {
    __shared__ AsmWaveLds lds[ASM_WAVES];
    AsmWaveLds& S = lds[threadIdx.x >> 6];
    const int lane = lane_id();
    const size_t gw = (size_t)blockIdx.x * ASM_WAVES + (threadIdx.x >> 6);
    vshort* pool = pool_all + gw * (size_t)pcap * AREC;                    // pcap records; record 0 = the untouched record
    volatile uint32_t* dir = dir_all + gw * (size_t)(nseg + 8);
    int* ilist = ilist_all + gw * (size_t)pcap;
    volatile short* iscore = iscore_all + gw * (size_t)pcap;
    volatile int* slotseg = slotseg_all + gw * (size_t)pcap;
    const unsigned long long below = (1ull << lane) - 1ull;
    auto lloc = [&](int i) { return i < bnreads ? boffs[i].offset : 0; };
    auto rec = [&](int seg) { return pool + (size_t)dir[seg] * AREC; };
    // records [0, hw) of the pool have been initialised (by this or an earlier launch on the same buffer) and are in the untouched state
    int hw = hw_all[gw];
    auto init_records = [&](int from, int to) {
        for (int r = from; r < to; ++r) {
            vshort* q = pool + (size_t)r * AREC;
            for (int j = lane; j < AREC - 2; j += 64) q[j] = 0;
            if (lane == 0) rec_set_index(q, -1);
        }
    };
    if (hw == 0) { init_records(0, 1); hw = 1; }
    int nslots = 1;                                                      // records handed out to the read in hand: [1, nslots)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    while (true) {
        unsigned int u = 0;
        if (lane == 0) u = atomicAdd(cursor, 1u);
        u = __shfl(u, 0);
        if (u >= (unsigned)n) break;
        const int rid = rid_list ? rid_list[u] : rid_begin + (int)u;
        const size_t uo = (size_t)(rid - rid_begin);                       // the read's place in the output
        const int L = qoffs[rid].size, read_name = qstart_id + rid;
        const int64_t qoff = qoffs[rid].offset;
        const int K = L < MHIP_KMER_SIZE ? 0 : (L - MHIP_KMER_SIZE) / BC + 1;
        int ncand = 0;
        bool ovf = false;                    // the read needs more records than the pool has: given up here, redone by the launch with full pools
        for (int strand = 0; strand < 2 && !ovf; ++strand) {
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
                    // a segment's first hit of the read takes the next record of the pool (records are handed out in order; what lies beyond the
                    // initialised ones is initialised first)
                    const bool fresh_rec = head && dir[seg] == 0u;
                    const unsigned long long fm = __ballot(fresh_rec);
                    if (fm && nslots + (int)__popcll(fm) > pcap) { ovf = true; break; }
                    if (fm) {
                        const int nnew = (int)__popcll(fm);
                        if (nslots + nnew > hw) {
                            init_records(hw, nslots + nnew);
                            hw = nslots + nnew;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        }
                        if (fresh_rec) {
                            const int sl = nslots + (int)__popcll(fm & below);
                            dir[seg] = (uint32_t)sl;
                            slotseg[sl] = seg;
                        }
                        nslots += nnew;
                    }
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
                if (ovf) break;
                s0 = n0; s1 = n1; pfirst = pnext;
                n0 = m0; n1 = m1;
            }
            if (ovf) break;
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
                // sweeps (:689-703): entries j < score of the neighbour's record, read as a flat array of shorts — past the record's end that
                // is the record of the segment behind it (and so on), wherever the pool holds it
                auto flat = [&](int useg, vshort* o, int fi) -> int {
                    if (fi < AREC) return (int)o[fi];
                    return (int)rec(useg + fi / AREC)[fi % AREC];
                };
                for (int useg = seg - 2, kk = c.num1 / AZV; useg >= 0 && kk >= 0; --useg, --kk) {
                    vshort* o = rec(useg);
                    const int osc = o[A_SCORE];
                    if (osc <= 0) continue;
                    const int sl = useg * AZV;
                    int agree = 0;
                    const int jend = min(osc, (nseg + 8 - useg) * AREC - A_SEED);      // entries inside this wave's (virtual) image (the reference reads its own heap beyond)
                    for (int j0 = 0; j0 < jend; j0 += 64) {
                        const int j = j0 + lane;
                        bool q = false;
                        if (j < jend) q = fabs((double)(loc_list - sl - flat(useg, o, A_LOC + j)) / ((double)(loc_seed - flat(useg, o, A_SEED + j)) * BC * 1.0) - 1.0) < 0.10;
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
                        if (j < jend) q = fabs((double)(sl + flat(useg, o, A_LOC + j) - loc_list) / ((double)(flat(useg, o, A_SEED + j) - loc_seed) * BC * 1.0) - 1.0) < 0.10;
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
            // ---- the strand's touched segments: score and index reset (:717)
            for (int t = lane; t < touched; t += 64) {
                const int seg = ((volatile int*)ilist)[t];
                rec(seg)[A_SCORE] = 0;
                rec_set_index(rec(seg), -1);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        // every read starts from an all-zero array: the records the read took go back to the untouched state (they are one contiguous
        // stretch of the pool) and their segments' directory entries to 0
        {
            volatile uint32_t* w = (volatile uint32_t*)(pool + AREC);           // (AREC shorts = 62 words per record)
            const int nw = (nslots - 1) * (AREC / 2);
            for (int j = lane; j < nw; j += 64) w[j] = (j % (AREC / 2)) == (AREC / 2 - 1) ? 0xffffffffu : 0u;
            for (int sl = 1 + lane; sl < nslots; sl += 64) dir[slotseg[sl]] = 0u;
            nslots = 1;
        }
        if (ovf) ncand = 0;
        for (int j = lane; j < ncand; j += 64) out[uo * AMAXC + j] = S.cand[j];
        if (lane == 0) {
            out_counts[uo] = ovf ? -1 : ncand;
            atomicMax(cursor + 4, (unsigned int)hw);
            if (ovf) atomicAdd(cursor + 5, 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (lane == 0) hw_all[gw] = hw;
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
    // One directory + record pool per resident wave, 16 waves per CU.  A pool holds ASM_POOL records (2 MB; a read of 8 kb takes about
    // 2 500, and records are touched from the front) or the worst case if that is less; a read that needs more gives up (count -1) and
    // is redone below with worst-case pools on a few waves.  Allocating for the worst case everywhere (20 MB per wave, 86 GB) made the
    // allocation itself the cost: device memory that another process has just given back, or that is handed out for the first time, is
    // cleared by the driver when it is allocated — a second and more for tens of GB, a tenth of that for these 10 GB.  MECAT_ASM_POOL overrides.
    const char* pe = getenv("MECAT_ASM_POOL");
    const int pcap = std::min(nseg + 9, std::max(64, pe && atoi(pe) > 0 ? atoi(pe) : ASM_POOL));
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    auto per_wave = [&](int cap) { return (size_t)cap * (AREC * sizeof(short) + sizeof(int) + sizeof(short) + sizeof(int)) + (size_t)(nseg + 8) * sizeof(uint32_t); };
    const size_t max_waves = (size_t)c->num_cus * 16;
    size_t waves = std::min<size_t>(max_waves, std::max<size_t>(1, free_b / 2 / per_wave(pcap)));
    waves = std::min<size_t>(waves, (size_t)n);
    unsigned grid = (unsigned)((waves + ASM_WAVES - 1) / ASM_WAVES);
    waves = (size_t)grid * ASM_WAVES;
    short *d_pool, *d_iscore;
    uint32_t* d_dir;
    int *d_ilist, *d_slotseg, *d_hw;
    unsigned int* d_cur;
    mhip_asm_candidate* d_out;
    int32_t* d_cnt;
    if (c->scratch("as_pool", sizeof(short) * waves * (size_t)pcap * AREC, (void**)&d_pool)) return -1;
    if (c->scratch("as_dir", sizeof(uint32_t) * waves * (size_t)(nseg + 8), (void**)&d_dir)) return -1;
    if (c->scratch("as_hw", sizeof(int) * max_waves, (void**)&d_hw)) return -1;
    if (c->scratch("as_ilist", sizeof(int) * waves * (size_t)pcap, (void**)&d_ilist)) return -1;
    if (c->scratch("as_iscore", sizeof(short) * waves * (size_t)pcap, (void**)&d_iscore)) return -1;
    if (c->scratch("as_slotseg", sizeof(int) * waves * (size_t)pcap, (void**)&d_slotseg)) return -1;
    if (c->scratch("as_cursor", 64, (void**)&d_cur)) return -1;
    if (c->scratch("as_out", sizeof(mhip_asm_candidate) * (size_t)n * AMAXC, (void**)&d_out)) return -1;
    if (c->scratch("as_cnt", sizeof(int32_t) * (size_t)n, (void**)&d_cnt)) return -1;
    // State that outlives a call: a completed call leaves every directory all zero and every record it initialised (the first hw of a
    // wave's pool) untouched, so nothing is set up again for as long as the buffers and the layout (segments and records per wave) are
    // the same.  A new layout moves the waves' shares: the directories are zeroed and the pools count as uninitialised (hw = 0; the
    // kernel initialises records as it hands them out).
    // (keyed on the buffers' allocation generations as well as their addresses: a buffer that grew may come back at its old address
    // with undefined contents)
    const uint64_t gens[3] = {c->scratch_generation("as_pool"), c->scratch_generation("as_dir"), c->scratch_generation("as_hw")};
    const bool same = d_pool == c->as_clean_base && d_dir == c->as_clean_dir && nseg == c->as_clean_nseg && pcap == c->as_clean_pcap &&
                      gens[0] == c->as_clean_gen[0] && gens[1] == c->as_clean_gen[1] && gens[2] == c->as_clean_gen[2];
    size_t waves_clean = same ? c->as_clean_nrec : 0;
    if (!same) HIPCHK(hipMemsetAsync(d_hw, 0, sizeof(int) * max_waves, c->stream));
    if (waves > waves_clean) {
        // (more waves than any call on these buffers had: the directories of the new ones)
        HIPCHK(hipMemsetAsync(d_dir + waves_clean * (size_t)(nseg + 8), 0, sizeof(uint32_t) * (waves - waves_clean) * (size_t)(nseg + 8), c->stream));
        waves_clean = waves;
    }
    c->as_clean_base = nullptr;               // (until this call has completed)
    HIPCHK(hipMemsetAsync(d_cur, 0, 64, c->stream));
    LAUNCH(c, "asm_seed", asm_seed, grid, ASM_BLOCK, 0, (const uint32_t*)block->d_pac, (const mhip_offset_t*)block->d_offs, block->num_reads,
           block->start_read_id, (const uint32_t*)idx->d_starts, (const int32_t*)idx->d_offsets, (const uint32_t*)reads->d_pac, (const uint32_t*)reads->d_npac,
           (const mhip_offset_t*)reads->d_offs, reads->start_read_id, rid_begin, n, (const int*)nullptr, d_pool, d_dir, d_hw, d_ilist, d_iscore, d_slotseg, nseg, pcap,
           gate, maxc, d_cur, d_out, d_cnt);
    HIPCHK(hipGetLastError());
    unsigned int st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(st, d_cur, sizeof(st), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(out_counts, d_cnt, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->as_clean_base = d_pool;
    c->as_clean_dir = d_dir;
    c->as_clean_gen[0] = gens[0]; c->as_clean_gen[1] = gens[1]; c->as_clean_gen[2] = gens[2];
    c->as_clean_nseg = nseg;
    c->as_clean_pcap = pcap;
    c->as_clean_nrec = waves_clean;
    if (getenv("MECAT_ASM_STATS")) fprintf(stderr, "[asm_seed] %d reads on %zu waves, pools of %d records: most records a wave initialised %u, reads given up %u\n", n, waves, pcap, st[4], st[5]);
    if (st[5]) {
        // the reads that ran out of records, again, with pools that cannot (one record per segment of the block): few waves, their own pool
        // buffer; directories (all zero again), lists and output are the first launch's
        std::vector<int> redo;
        for (int i = 0; i < n; ++i)
            if (out_counts[i] < 0) redo.push_back(rid_begin + i);
        const int full = nseg + 9;
        size_t w2 = std::min<size_t>(std::min<size_t>(waves, redo.size()), std::max<size_t>(1, ((size_t)8 << 30) / ((size_t)full * (AREC * sizeof(short) + 10))));
        const unsigned grid2 = (unsigned)((w2 + ASM_WAVES - 1) / ASM_WAVES);
        w2 = (size_t)grid2 * ASM_WAVES;
        short *d_pool2, *d_iscore2;
        int *d_ilist2, *d_slotseg2, *d_redo, *d_hw2;
        if (w2 > waves) { mhip_set_error("asm_seed: no room for the worst-case pools of %zu reads", redo.size()); return -1; }
        if (c->scratch("as_pool_full", sizeof(short) * w2 * (size_t)full * AREC, (void**)&d_pool2)) return -1;
        if (c->scratch("as_ilist_full", sizeof(int) * w2 * (size_t)full, (void**)&d_ilist2)) return -1;
        if (c->scratch("as_iscore_full", sizeof(short) * w2 * (size_t)full, (void**)&d_iscore2)) return -1;
        if (c->scratch("as_slotseg_full", sizeof(int) * w2 * (size_t)full, (void**)&d_slotseg2)) return -1;
        if (c->scratch("as_hw_full", sizeof(int) * max_waves, (void**)&d_hw2)) return -1;
        if (c->scratch("as_redo", sizeof(int) * redo.size(), (void**)&d_redo)) return -1;
        HIPCHK(hipMemsetAsync(d_hw2, 0, sizeof(int) * max_waves, c->stream));        // (these pools always start uninitialised)
        HIPCHK(hipMemsetAsync(d_cur, 0, 64, c->stream));
        HIPCHK(hipMemcpyAsync(d_redo, redo.data(), sizeof(int) * redo.size(), hipMemcpyHostToDevice, c->stream));
        c->as_clean_base = nullptr;
        LAUNCH(c, "asm_seed_full", asm_seed, grid2, ASM_BLOCK, 0, (const uint32_t*)block->d_pac, (const mhip_offset_t*)block->d_offs, block->num_reads,
               block->start_read_id, (const uint32_t*)idx->d_starts, (const int32_t*)idx->d_offsets, (const uint32_t*)reads->d_pac, (const uint32_t*)reads->d_npac,
               (const mhip_offset_t*)reads->d_offs, reads->start_read_id, rid_begin, (int)redo.size(), (const int*)d_redo, d_pool2, d_dir, d_hw2, d_ilist2, d_iscore2,
               d_slotseg2, nseg, full, gate, maxc, d_cur, d_out, d_cnt);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out_counts, d_cnt, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->as_clean_base = d_pool;
    }
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(mhip_asm_candidate) * (size_t)n * AMAXC, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
