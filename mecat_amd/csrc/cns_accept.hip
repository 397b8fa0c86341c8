// cns_accept.hip — mecat2cns' candidate accept loop on top of the device re-aligner.
//
// Replaces, for a batch of template reads, the front half of consensus_one_read_can_pacbio / consensus_one_read_can_nanopore
// (mecat2cns/mecat_correction.cpp:388-450, 452-515): everything up to the point where an accepted alignment is handed to the
// consensus table (meap_add_one_aln) and to CnsAlns::add_aln.  The reference walks a template's candidates one by one —
//     sort by (score desc, qid, qext)                                             :362-370, :409
//     for each, while fewer than 60 (PacBio) / 100 (nanopore) are accepted and fewer than 200 were looked at:
//         skip a query read that was already accepted (std::set used_ids)         :424
//         GetAlignment(error_rate 0.15 / 0.20, min_align_size)                    :431 (dw.cpp:482-553)
//         check_ovlp_mapping_range with min_mapping_ratio - 0.02                  :191-200, :432
//         check_cov_stats: the aligned template range must have >= 200 positions below coverage 20, then ++coverage   :372-386
//         normalize_gaps(qaln, saln, push = true) -> meap_add_one_aln, add_aln    reads_correction_aux.cpp:3-81
// — and every alignment depends on nothing but the two reads, so the <= 200 candidates of every template of the batch are
// re-aligned speculatively on the device (mhip_cns_align_candidates_dev, a slice of the batch's templates per launch), the sequential
// accept decisions are replayed over the results on host threads, and the gap-normalised strings of the ACCEPTED alignments are
// built on the device as well (cns_strings.hip) and copied into the result buffer while the next slice re-aligns.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <set>
#include <thread>
#include <vector>

#include <functional>

#include "common.h"
#include "cns_strings.h"

namespace {

struct CmpByScore {      // CmpExtensionCandidateByScore, mecat_correction.cpp:362-370
    bool operator()(const mhip_ext_candidate& a, const mhip_ext_candidate& b) const {
        if (a.score != b.score) return a.score > b.score;
        if (a.qid != b.qid) return a.qid < b.qid;
        return a.qext < b.qext;
    }
};

template <typename F>
void parallel_for(int64_t n, int nthreads, F f) {
    nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, n));
    std::atomic<int64_t> next{0};
    auto body = [&]() {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n) return;
            f(i);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(body);
    body();
    for (auto& x : th) x.join();
}

}  // namespace

extern "C" {

// The strings of a batch are gigabytes (24 GB at config 2), and pages that are touched for the first time cost more than the copy that
// fills them.  So the library keeps ONE released string buffer and
// hands it out again when the next batch fits it (mecat2cns works through its partitions batch after batch): pages that are mapped
// already.  mhip_cns_free parks a buffer it knows instead of freeing it; a larger request replaces the parked one.
namespace {
std::mutex g_strbuf_mu;
char* g_strbuf_parked = nullptr;          // released, reusable
size_t g_strbuf_parked_cap = 0;
std::map<void*, size_t> g_strbuf_out;     // handed to a caller: address -> capacity
std::set<void*> g_strbuf_reg;             // page-locked (hipHostRegister): the copy engine fills them without a staging copy, asynchronously
void strbuf_free(void* p) {               // (g_strbuf_mu held or not: only the set is shared)
    if (!p) return;
    bool reg;
    { std::lock_guard<std::mutex> lk(g_strbuf_mu); reg = g_strbuf_reg.erase(p) != 0; }
    if (reg) (void)hipHostUnregister(p);
    free(p);
}
char* strbuf_get(size_t bytes, int num_threads) {
    {
        std::lock_guard<std::mutex> lk(g_strbuf_mu);
        if (g_strbuf_parked && g_strbuf_parked_cap >= bytes) {
            char* p = g_strbuf_parked;
            g_strbuf_out[p] = g_strbuf_parked_cap;
            g_strbuf_parked = nullptr;
            g_strbuf_parked_cap = 0;
            return p;
        }
    }
    void* p = nullptr;
    const size_t two_mb = (size_t)2 << 20, cap = (bytes + bytes / 16 + two_mb - 1) & ~(two_mb - 1);
    if (posix_memalign(&p, two_mb, cap) != 0) return nullptr;
    (void)madvise(p, cap, MADV_HUGEPAGE);
    // first touch on the host threads (huge pages: 8 GB in 30 ms on 32 threads), then page-locked — 70 ms for 8 GB of touched pages, against
    // 0.4 s untouched and 1.9 s for a hipHostMalloc of the size (tools/dev/probes/pin_probe.hip)
    parallel_for((int64_t)(cap / two_mb), num_threads, [&](int64_t pg) { ((volatile char*)p)[(size_t)pg * two_mb] = 0; });
    const bool reg = hipHostRegister(p, cap, hipHostRegisterDefault) == hipSuccess;
    if (!reg) (void)hipGetLastError();      // stays pageable: the copies still work, through the runtime's staging
    std::lock_guard<std::mutex> lk(g_strbuf_mu);
    g_strbuf_out[p] = cap;
    if (reg) g_strbuf_reg.insert(p);
    return (char*)p;
}
}  // namespace

void mhip_cns_free(void* p) {
    if (!p) return;
    void* to_free = p;
    {
        std::lock_guard<std::mutex> lk(g_strbuf_mu);
        auto it = g_strbuf_out.find(p);
        if (it != g_strbuf_out.end()) {
            const size_t cap = it->second;
            g_strbuf_out.erase(it);
            if (cap > g_strbuf_parked_cap) {      // park this one, free what was parked (the smaller of the two)
                to_free = g_strbuf_parked;
                g_strbuf_parked = (char*)p;
                g_strbuf_parked_cap = cap;
            }
        }
    }
    strbuf_free(to_free);      // (a plain malloc'ed buffer — the accepted records, small string buffers — is just freed)
}

// gives the parked string buffer (see above) back to the system; buffers still in a caller's hands are not touched
void mhip_cns_release_parked(void) {
    void* p;
    {
        std::lock_guard<std::mutex> lk(g_strbuf_mu);
        p = g_strbuf_parked;
        g_strbuf_parked = nullptr;
        g_strbuf_parked_cap = 0;
    }
    strbuf_free(p);
}

int mhip_cns_accept_templates(mhip_ctx* c, const mhip_volume* vol, const uint8_t* /*host_pac: not read any more (the strings are built on the device)*/, mhip_ext_candidate* cands, const int64_t* tmpl_begin,
                              int num_templates, int tech, int min_align_size, double min_mapping_ratio, int num_threads,
                              mhip_cns_accepted** out_accepted, int64_t* out_count, char** out_strings, int64_t* out_strings_bytes,
                              int64_t* out_jobs) {
    HIPCHK(hipSetDevice(c->device));
    *out_accepted = nullptr; *out_count = 0; *out_strings = nullptr; *out_strings_bytes = 0;
    if (out_jobs) *out_jobs = 0;
    if (num_templates <= 0) return 0;
    const int max_ext = 200;                                         // mecat_correction.cpp:412
    const int max_added = tech == 0 ? 60 : 100;                      // :407 ; MAX_CNS_OVLPS, reads_correction_aux.h:32
    const double error_rate = tech == 0 ? 0.15 : 0.20;               // :431 / :494
    const double ratio = min_mapping_ratio - 0.02;                   // :406
    const int start_id = vol->start_read_id, nreads = vol->num_reads;
    num_threads = std::max(1, num_threads);
    // MECAT_CNS_TIMES=1: where the call's wall time went, on stderr
    const bool times = getenv("MECAT_CNS_TIMES") != nullptr;
    double tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = now();
    auto lap = [&](int k) { const double t = now(); tk[k] += t - t_last; t_last = t; };

    // 1. order of the reference's walk
    std::atomic<int> bad{0};
    parallel_for(num_templates, num_threads, [&](int64_t t) {
        mhip_ext_candidate* b = cands + tmpl_begin[t];
        mhip_ext_candidate* e = cands + tmpl_begin[t + 1];
        // std::sort, not stable_sort: the reference sorts the same array (file order of the partition) with the same comparator and
        // the same libstdc++ introsort (mecat_correction.cpp:409 / :472), so records that tie on (score, qid, qext) come out in the
        // reference's order exactly when the input order is the reference's — which is what the caller hands over
        std::sort(b, e, CmpByScore());
        for (mhip_ext_candidate* p = b; p < e; ++p) {
            if (p->sdir != 0 || p->qid < start_id || p->qid >= start_id + nreads || p->sid < start_id || p->sid >= start_id + nreads ||
                p->sid != b->sid) {
                bad = 1;
                continue;
            }
            // the record's read lengths are the volume's: the replay indexes the coverage array and the reads with them (the
            // reference has a fixed MAX_SEQ_SIZE array there; a record that disagrees with the volume is refused, not trusted)
            if (p->qsize != vol->h_offs[(size_t)(p->qid - start_id)].size || p->ssize != vol->h_offs[(size_t)(p->sid - start_id)].size) bad = 2;
        }
    });
    if (bad.load() == 2) { mhip_set_error("cns accept: a candidate's qsize / ssize differs from the read lengths of the volume"); return -1; }
    if (bad.load()) { mhip_set_error("cns accept: a candidate is outside the volume, has sdir != 0 or sits in another template's range"); return -1; }

    lap(0);
    // 2. the first <= 200 candidates of every template, as alignment jobs
    std::vector<int64_t> jfirst((size_t)num_templates + 1, 0);
    for (int t = 0; t < num_templates; ++t) jfirst[(size_t)t + 1] = jfirst[(size_t)t] + std::min<int64_t>(max_ext, tmpl_begin[t + 1] - tmpl_begin[t]);
    const int64_t nj = jfirst[(size_t)num_templates];
    if (out_jobs) *out_jobs = nj;
    if (nj == 0) return 0;
    if (nj > 0x7fffffffLL) { mhip_set_error("cns accept: too many jobs in one batch"); return -1; }
    std::vector<mhip_aln_job> jobs((size_t)nj);
    parallel_for(num_templates, num_threads, [&](int64_t t) {
        for (int64_t k = 0; k < jfirst[(size_t)t + 1] - jfirst[(size_t)t]; ++k) {
            const mhip_ext_candidate& ec = cands[tmpl_begin[t] + k];
            mhip_aln_job j;
            j.qid_local = ec.qid - start_id;
            j.sid_local = ec.sid - start_id;
            j.chain = ec.qdir != 0;
            j.qstart = ec.qdir != 0 ? ec.qsize - 1 - ec.qext : ec.qext;      // :428-429
            j.sstart = ec.sext;
            jobs[(size_t)(jfirst[(size_t)t] + k)] = j;
        }
    });
    int max_len = 16;
    for (int64_t i = 0; i < nj; ++i)
        max_len = std::max(max_len, std::max(vol->h_offs[(size_t)jobs[(size_t)i].qid_local].size, vol->h_offs[(size_t)jobs[(size_t)i].sid_local].size));
    // columns of one direction <= bases of both reads on that side; 16-column words
    const int cap = (int)(((int64_t)max_len * 2 + 64 + 15) / 16 * 16);
    const int row_words = 2 * (cap / 16);

    // 3. The batch goes through the device in SLICES of whole templates (at most MECAT_CNS_SLICE_JOBS jobs, default 1.2 M: four slices
    // at config 2) with two sets of buffers in turn:
    //     slice k + 1 is re-aligned on the GPU      while   a host thread replays the accept decisions of slice k
    //     the strings of slice k are built on the GPU behind it, and cross the PCIe link on a second stream while slice k + 2 re-aligns
    // The host side is the replay only; the string buffer is filled by the copy engine.
    int64_t slice_jobs = 1200000;
    if (const char* e = getenv("MECAT_CNS_SLICE_JOBS")) slice_jobs = std::max<int64_t>(1, atoll(e));
    std::vector<int> sl_t;                                    // first template of every slice, and one behind the last
    for (int t = 0; t < num_templates;) {
        sl_t.push_back(t);
        int u = t + 1;
        while (u < num_templates && jfirst[(size_t)u + 1] - jfirst[(size_t)t] <= slice_jobs) ++u;
        t = u;
    }
    sl_t.push_back(num_templates);
    const int nslices = (int)sl_t.size() - 1;
    int64_t max_slice = 0;
    for (int k = 0; k < nslices; ++k) max_slice = std::max(max_slice, jfirst[(size_t)sl_t[(size_t)k + 1]] - jfirst[(size_t)sl_t[(size_t)k]]);

    mhip_aln_job* d_jobs;
    if (c->scratch("ca_jobs", sizeof(mhip_aln_job) * (size_t)nj, (void**)&d_jobs)) return -1;
    HIPCHK(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(mhip_aln_job) * (size_t)nj, hipMemcpyHostToDevice, c->stream));
    mhip_cns_result* d_res[2] = {nullptr, nullptr};
    uint32_t* d_ops[2] = {nullptr, nullptr};
    for (int b = 0; b < std::min(2, nslices); ++b) {
        if (c->scratch(b ? "ca_res1" : "ca_res", sizeof(mhip_cns_result) * (size_t)max_slice, (void**)&d_res[b])) return -1;
        if (c->scratch(b ? "ca_ops1" : "ca_ops", sizeof(uint32_t) * (size_t)row_words * (size_t)max_slice, (void**)&d_ops[b])) return -1;
    }
    std::vector<mhip_cns_result> res((size_t)nj);           // every slice's results land in their place
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_built[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    struct Cleanup {
        hipStream_t& s; hipEvent_t* a; hipEvent_t* b;
        ~Cleanup() {
            if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
            for (int i = 0; i < 2; ++i) { if (a[i]) (void)hipEventDestroy(a[i]); if (b[i]) (void)hipEventDestroy(b[i]); }
        }
    } cleanup{copy_stream, ev_built, ev_copied};
    HIPCHK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
        HIPCHK(hipEventCreateWithFlags(&ev_built[b], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_copied[b], hipEventDisableTiming));
    }

    // the result buffer: sized after the first slice's replay from its bytes per template (+ 15 %); a later slice that does not fit
    // gets a larger one, the strings copied so far moved over (MECAT_CNS_STR_ESTIMATE=<percent> scales the estimate: test knob)
    char* S = nullptr;
    size_t S_cap = 0, S_used = 0;
    bool S_big = false;
    auto s_release = [&]() {
        if (S && copy_stream) (void)hipStreamSynchronize(copy_stream);      // no copy may still be writing into it
        if (S) { if (S_big) mhip_cns_free(S); else free(S); }
        S = nullptr; S_cap = 0;
    };
    struct SGuard { std::function<void()> f; bool armed = true; ~SGuard() { if (armed) f(); } } s_guard{s_release};
    auto s_reserve = [&](size_t want) -> int {
        if (want <= S_cap) return 0;
        if (hipStreamSynchronize(copy_stream) != hipSuccess) { mhip_set_error("cns accept: copy stream failed"); return -1; }      // the copies into the old buffer have landed
        const bool big = want >= ((size_t)64 << 20);
        char* nS = big ? strbuf_get(want, num_threads) : (char*)malloc(std::max<size_t>(want, 1));
        if (!nS) { mhip_set_error("out of memory (%lld bytes of aligned strings)", (long long)want); return -1; }
        if (S_used) {
            const size_t piece = (size_t)64 << 20;
            parallel_for((int64_t)((S_used + piece - 1) / piece), num_threads, [&](int64_t pc) {
                const size_t o = (size_t)pc * piece;
                memcpy(nS + o, S + o, std::min(piece, S_used - o));
            });
        }
        s_release();
        S = nS; S_cap = want; S_big = big;
        return 0;
    };
    double est_scale = 1.15;
    if (const char* e = getenv("MECAT_CNS_STR_ESTIMATE")) est_scale = std::max(0.01, atof(e) / 100.0);

    struct Slice {
        int t0 = 0, t1 = 0;
        int64_t j0 = 0, nj = 0;
        std::vector<std::vector<int32_t>> acc;      // per template: accepted job indices (batch-wide), in acceptance order
        std::vector<int64_t> afirst;               // per template: first accepted record of the slice
        std::vector<CnsStrItem> items;
        size_t sbytes = 0;
    };
    std::vector<mhip_cns_accepted> Avec;
    lap(1);

    auto align_slice = [&](Slice& sl, int b) -> int {
        if (mhip_cns_align_candidates_dev(c, vol, vol, d_jobs + sl.j0, (int)sl.nj, error_rate, min_align_size, cap, d_res[b], d_ops[b])) return -1;
        HIPCHK(hipMemcpyAsync(res.data() + sl.j0, d_res[b], sizeof(mhip_cns_result) * (size_t)sl.nj, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return 0;
    };
    // the sequential accept decisions, per template (host threads)
    auto replay_slice = [&](Slice& sl) {
        const int nt = sl.t1 - sl.t0;
        sl.acc.assign((size_t)nt, std::vector<int32_t>());
        parallel_for(nt, num_threads, [&](int64_t tl) {
            const int64_t t = sl.t0 + tl;
            const int64_t b = tmpl_begin[t], n = tmpl_begin[t + 1] - b;
            if (n == 0) return;
            const int ssize = vol->h_offs[(size_t)(cands[b].sid - start_id)].size;      // (== every candidate's ssize: checked above)
            std::vector<uint8_t> cov((size_t)std::max(ssize, 1), 0);
            std::set<int> used;
            int num_added = 0, num_ext = 0;
            for (int64_t i = 0; i < n && num_added < max_added && num_ext < max_ext; ++i) {
                ++num_ext;
                const mhip_ext_candidate& ec = cands[b + i];
                if (used.find(ec.qid) != used.end()) continue;
                const int64_t ji = jfirst[(size_t)t] + i;          // i < max_ext here: the job exists
                const mhip_cns_result& r = res[(size_t)ji];
                if (!r.ok) continue;
                const int oq = r.qend - r.qoff, qqs = (int)(ec.qsize * ratio), os = r.send - r.soff, qss = (int)(ec.ssize * ratio);      // :191-200
                if (!(oq >= qqs || os >= qss)) continue;
                int full = 0;                                        // check_cov_stats, :372-386
                for (int p = r.soff; p < r.send; ++p) full += cov[(size_t)p] >= 20;
                if (!(r.send - r.soff >= full + 200)) continue;
                for (int p = r.soff; p < r.send; ++p) ++cov[(size_t)p];
                ++num_added;
                used.insert(ec.qid);
                sl.acc[(size_t)tl].push_back((int32_t)ji);
            }
        });
        sl.afirst.assign((size_t)nt + 1, 0);
        for (int tl = 0; tl < nt; ++tl) sl.afirst[(size_t)tl + 1] = sl.afirst[(size_t)tl] + (int64_t)sl.acc[(size_t)tl].size();
        const int64_t na = sl.afirst[(size_t)nt];
        sl.items.resize((size_t)na);
        size_t off = 0;
        for (int tl = 0; tl < nt; ++tl)
            for (size_t k = 0; k < sl.acc[(size_t)tl].size(); ++k) {
                const int64_t ji = sl.acc[(size_t)tl][k];
                const mhip_cns_result& r = res[(size_t)ji];
                CnsStrItem& it = sl.items[(size_t)(sl.afirst[(size_t)tl] + (int64_t)k)];
                it.job = (int32_t)(ji - sl.j0);
                it.aln_size = r.last_col - r.first_col;          // O(ND) columns are matches or indels: normalising adds no columns
                it.off = (unsigned long long)off;
                off += 2 * ((size_t)it.aln_size + 1);
            }
        sl.sbytes = off;
    };
    // strings of an (aligned, replayed) slice: built on the device behind whatever the stream holds, copied on the second stream
    auto strings_slice = [&](Slice& sl, int b, int k) -> int {
        const int64_t na = (int64_t)sl.items.size();
        if (na == 0) return 0;
        size_t want = S_used + sl.sbytes;
        if (want > S_cap && k + 1 < nslices)
            want = std::max(want, (size_t)((double)want / (double)sl.t1 * (double)num_templates * est_scale) + ((size_t)1 << 20));
        if (s_reserve(want)) return -1;
        // the accepted records (what the caller gets beside the strings)
        const size_t a0 = Avec.size();
        Avec.resize(a0 + (size_t)na);
        parallel_for(sl.t1 - sl.t0, num_threads, [&](int64_t tl) {
            const int64_t t = sl.t0 + tl;
            for (size_t kk = 0; kk < sl.acc[(size_t)tl].size(); ++kk) {
                const int64_t ji = sl.acc[(size_t)tl][kk];
                const mhip_cns_result& r = res[(size_t)ji];
                const CnsStrItem& it = sl.items[(size_t)(sl.afirst[(size_t)tl] + (int64_t)kk)];
                const mhip_ext_candidate& ec = cands[tmpl_begin[t] + (ji - jfirst[(size_t)t])];
                mhip_cns_accepted& o = Avec[a0 + (size_t)(sl.afirst[(size_t)tl] + (int64_t)kk)];
                o.template_index = (int32_t)t;
                o.cand_index = tmpl_begin[t] + (ji - jfirst[(size_t)t]);
                o.qid = ec.qid; o.sid = ec.sid;
                o.qoff = r.qoff; o.qend = r.qend; o.soff = r.soff; o.send = r.send;
                o.aln_size = it.aln_size;
                o.str_offset = (int64_t)(S_used + it.off);
            }
        });
        if (k >= 2) HIPCHK(hipEventSynchronize(ev_copied[b]));      // the copy out of this set's string buffer two slices ago (the buffer may move)
        char* d_str;
        CnsStrItem* d_items;
        if (c->scratch(b ? "ca_str1" : "ca_str", sl.sbytes + 128, (void**)&d_str)) return -1;
        if (c->scratch(b ? "ca_items1" : "ca_items", sizeof(CnsStrItem) * (size_t)na, (void**)&d_items)) return -1;
        HIPCHK(hipMemcpyAsync(d_items, sl.items.data(), sizeof(CnsStrItem) * (size_t)na, hipMemcpyHostToDevice, c->stream));
        if (cns_strings_launch(c, vol, d_jobs + sl.j0, d_res[b], d_ops[b], row_words, d_items, (int)na, d_str)) return -1;
        HIPCHK(hipEventRecord(ev_built[b], c->stream));
        HIPCHK(hipStreamWaitEvent(copy_stream, ev_built[b], 0));
        HIPCHK(hipMemcpyAsync(S + S_used, d_str, sl.sbytes, hipMemcpyDeviceToHost, copy_stream));
        HIPCHK(hipEventRecord(ev_copied[b], copy_stream));
        S_used += sl.sbytes;
        return 0;
    };

    std::vector<Slice> slices((size_t)nslices);
    for (int k = 0; k < nslices; ++k) {
        Slice& sl = slices[(size_t)k];
        sl.t0 = sl_t[(size_t)k]; sl.t1 = sl_t[(size_t)k + 1];
        sl.j0 = jfirst[(size_t)sl.t0]; sl.nj = jfirst[(size_t)sl.t1] - sl.j0;
    }
    if (slices[0].nj > 0 && align_slice(slices[0], 0)) return -1;
    lap(2);
    for (int k = 0; k < nslices; ++k) {
        Slice& sl = slices[(size_t)k];
        std::thread rp([&]() { replay_slice(sl); });
        struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join{rp};
        int rc = 0;
        if (k + 1 < nslices && slices[(size_t)k + 1].nj > 0) rc = align_slice(slices[(size_t)k + 1], (k + 1) & 1);
        lap(2);
        rp.join();
        lap(3);
        if (rc) return -1;
        if (strings_slice(sl, k & 1, k)) return -1;
        sl.acc.clear(); sl.acc.shrink_to_fit(); sl.items.clear(); sl.items.shrink_to_fit();
        lap(4);
    }
    HIPCHK(hipStreamSynchronize(copy_stream));
    const int64_t na = (int64_t)Avec.size();
    if (na == 0) return 0;
    mhip_cns_accepted* A = (mhip_cns_accepted*)malloc(sizeof(mhip_cns_accepted) * (size_t)na);
    if (!A) { mhip_set_error("out of memory"); return -1; }
    memcpy(A, Avec.data(), sizeof(mhip_cns_accepted) * (size_t)na);
    const int64_t sbytes = (int64_t)S_used;
    s_guard.armed = false;
    lap(5);
    if (times)
        fprintf(stderr, "[cns_accept] %d templates, %lld jobs in %d slices, %lld accepted, %.2f GB of strings: sort + checks %.3f s, jobs %.3f, re-alignment on the "
                        "device %.3f, accept replay beyond it %.3f, strings launched %.3f, last copies %.3f\n", num_templates, (long long)nj, nslices, (long long)na,
                (double)sbytes / 1e9, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5]);
    *out_accepted = A;
    *out_count = na;
    *out_strings = S;
    *out_strings_bytes = sbytes;
    return 0;
}

}  // extern "C"
