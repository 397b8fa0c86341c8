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
// re-aligned speculatively in ONE device launch (mhip_cns_align_candidates_dev), the sequential accept decisions are replayed
// over the results on host threads, only the 2-bit column strings of the ACCEPTED alignments are gathered on the device and
// brought back, and the gap-normalised strings are rebuilt from them and the host copy of the reads.
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

#include "common.h"
#include "aln_strings.h"

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

inline int host_base(const uint8_t* pac, int64_t idx) { return (pac[idx >> 2] >> ((~idx & 3) << 1)) & 3; }      // packed_db.h:103-107

// normalize_gaps (reads_correction_aux.cpp:3-81), push = true; q / t are NUL-terminated work buffers of equal length n
void push_gaps(char* q, char* t, int64_t n) {
    for (int64_t i = 0; i + 1 < n; ++i) {
        if (t[i] == '-') {
            int64_t j = i;
            for (;;) {
                const char c = t[++j];
                if (c != '-' || j > n - 1) {
                    if (c == q[i]) { t[i] = c; t[j] = '-'; }
                    break;
                }
            }
        }
        if (q[i] == '-') {
            int64_t j = i;
            for (;;) {
                const char c = q[++j];
                if (c != '-' || j > n - 1) {
                    if (c == t[i]) { q[i] = c; q[j] = '-'; }
                    break;
                }
            }
        }
    }
}

// the 2-bit columns of the accepted jobs -> one dense buffer: one wave per accepted alignment, its left words then its right words at
// woff[i] (a direction fills a fraction of its worst-case stride: 119 MB instead of 424 MB cross the PCIe link at config 2's 28 000 accepted)
__global__ __launch_bounds__(256) void cns_gather_ops(const uint32_t* __restrict__ ops, const int32_t* __restrict__ sel, const int32_t* __restrict__ lw,
                                                      const int32_t* __restrict__ rw, const unsigned long long* __restrict__ woff, int nsel, int row_words,
                                                      uint32_t* __restrict__ out) {
    const int lane = (int)threadIdx.x & 63;
    for (size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < (size_t)nsel; i += (size_t)gridDim.x * 4) {
        const uint32_t* src = ops + (size_t)sel[i] * row_words;
        uint32_t* dst = out + woff[i];
        const int nl = lw[i], nr = rw[i];
        for (int j = lane; j < nl; j += 64) dst[j] = src[j];
        for (int j = lane; j < nr; j += 64) dst[nl + j] = src[row_words / 2 + j];
    }
}

// four bases of a packed byte as characters (first base in the high bits: packed_db.h:103-107), forward and reverse-complemented
struct BaseLut {
    uint32_t fwd[256], rc[256];
    BaseLut() {
        for (int b = 0; b < 256; ++b) {
            char f[4], r[4];
            for (int k = 0; k < 4; ++k) {
                const int c = (b >> ((3 - k) << 1)) & 3;
                f[k] = "ACGT"[c];
                r[3 - k] = "ACGT"[3 - c];
            }
            memcpy(&fwd[b], f, 4);
            memcpy(&rc[b], r, 4);
        }
    }
};
const BaseLut g_lut;

// bases [from, from + n) of the read at `off` as characters (n + up to 7 bytes of dst are written)
void decode_fwd(const uint8_t* pac, int64_t off, int from, int n, char* dst) {
    int64_t idx = off + from;
    int k = 0;
    for (; k < n && (idx & 3); ++k, ++idx) dst[k] = "ACGT"[host_base(pac, idx)];
    for (; k < n; k += 4, idx += 4) memcpy(dst + k, &g_lut.fwd[pac[idx >> 2]], 4);
}
// the same range of the read's reverse complement: position i of the strand is the complement of base size - 1 - i
void decode_rc(const uint8_t* pac, int64_t off, int size, int from, int n, char* dst) {
    // strand positions from .. from + n - 1  <->  read positions size - 1 - from  down to  size - from - n
    int64_t idx = off + size - 1 - from;                 // read index of the first character, walking down
    int k = 0;
    for (; k < n && ((idx & 3) != 3); ++k, --idx) dst[k] = "ACGT"[3 - host_base(pac, idx)];
    for (; k + 4 <= n; k += 4, idx -= 4) memcpy(dst + k, &g_lut.rc[pac[idx >> 2]], 4);
    for (; k < n; ++k, --idx) dst[k] = "ACGT"[3 - host_base(pac, idx)];
}

}  // namespace

extern "C" {

// The strings of a batch are gigabytes (14 GB at config 2) that the host threads touch for the first time while they write them: fresh
// pages from the kernel at 3.5 GB/s on this host — 4 of a batch's 6 seconds.  So the library keeps ONE released string buffer and
// hands it out again when the next batch fits it (mecat2cns works through its partitions batch after batch): pages that are mapped
// already.  mhip_cns_free parks a buffer it knows instead of freeing it; a larger request replaces the parked one.
namespace {
std::mutex g_strbuf_mu;
char* g_strbuf_parked = nullptr;          // released, reusable
size_t g_strbuf_parked_cap = 0;
std::map<void*, size_t> g_strbuf_out;     // handed to a caller: address -> capacity
char* strbuf_get(size_t bytes) {
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
    std::lock_guard<std::mutex> lk(g_strbuf_mu);
    g_strbuf_out[p] = cap;
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
    free(to_free);
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
    free(p);
}

int mhip_cns_accept_templates(mhip_ctx* c, const mhip_volume* vol, const uint8_t* host_pac, mhip_ext_candidate* cands, const int64_t* tmpl_begin,
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
    int max_len = 16;
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
    for (int64_t i = 0; i < nj; ++i)
        max_len = std::max(max_len, std::max(vol->h_offs[(size_t)jobs[(size_t)i].qid_local].size, vol->h_offs[(size_t)jobs[(size_t)i].sid_local].size));
    // columns of one direction <= bases of both reads on that side; 16-column words
    const int cap = (int)(((int64_t)max_len * 2 + 64 + 15) / 16 * 16);
    const int row_words = 2 * (cap / 16);

    lap(1);
    // 3. one speculative device launch
    mhip_aln_job* d_jobs;
    mhip_cns_result* d_res;
    uint32_t* d_ops;
    if (c->scratch("ca_jobs", sizeof(mhip_aln_job) * (size_t)nj, (void**)&d_jobs)) return -1;
    if (c->scratch("ca_res", sizeof(mhip_cns_result) * (size_t)nj, (void**)&d_res)) return -1;
    if (c->scratch("ca_ops", sizeof(uint32_t) * (size_t)row_words * (size_t)nj, (void**)&d_ops)) return -1;
    HIPCHK(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(mhip_aln_job) * (size_t)nj, hipMemcpyHostToDevice, c->stream));
    if (mhip_cns_align_candidates_dev(c, vol, vol, d_jobs, (int)nj, error_rate, min_align_size, cap, d_res, d_ops)) return -1;
    std::vector<mhip_cns_result> res((size_t)nj);
    HIPCHK(hipMemcpyAsync(res.data(), d_res, sizeof(mhip_cns_result) * (size_t)nj, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));

    lap(2);
    // 4. the sequential accept decisions, per template
    std::vector<std::vector<int32_t>> acc((size_t)num_templates);      // accepted job indices, in acceptance order
    parallel_for(num_templates, num_threads, [&](int64_t t) {
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
            acc[(size_t)t].push_back((int32_t)ji);
        }
    });
    std::vector<int64_t> afirst((size_t)num_templates + 1, 0);
    for (int t = 0; t < num_templates; ++t) afirst[(size_t)t + 1] = afirst[(size_t)t] + (int64_t)acc[(size_t)t].size();
    const int64_t na = afirst[(size_t)num_templates];
    if (na == 0) return 0;

    lap(3);
    // 5. the accepted alignments' columns, packed: left words then right words of every accepted alignment
    std::vector<int32_t> sel((size_t)na), lw((size_t)na), rw((size_t)na);
    std::vector<unsigned long long> woff((size_t)na + 1);
    for (int t = 0; t < num_templates; ++t) std::copy(acc[(size_t)t].begin(), acc[(size_t)t].end(), sel.begin() + afirst[(size_t)t]);
    woff[0] = 0;
    for (int64_t a = 0; a < na; ++a) {
        const mhip_cns_result& r = res[(size_t)sel[(size_t)a]];
        lw[(size_t)a] = (r.left_cols + 15) >> 4;
        rw[(size_t)a] = (r.right_cols + 15) >> 4;
        woff[(size_t)a + 1] = woff[(size_t)a] + (unsigned long long)(lw[(size_t)a] + rw[(size_t)a]);
    }
    const size_t dense_words = (size_t)woff[(size_t)na];
    int32_t *d_sel, *d_lw, *d_rw;
    unsigned long long* d_woff;
    uint32_t* d_pack;
    if (c->scratch("ca_sel", sizeof(int32_t) * (size_t)na, (void**)&d_sel)) return -1;
    if (c->scratch("ca_lw", sizeof(int32_t) * (size_t)na, (void**)&d_lw)) return -1;
    if (c->scratch("ca_rw", sizeof(int32_t) * (size_t)na, (void**)&d_rw)) return -1;
    if (c->scratch("ca_woff", sizeof(unsigned long long) * ((size_t)na + 1), (void**)&d_woff)) return -1;
    if (c->scratch("ca_pack", sizeof(uint32_t) * std::max<size_t>(dense_words, 1), (void**)&d_pack)) return -1;
    HIPCHK(hipMemcpyAsync(d_sel, sel.data(), sizeof(int32_t) * (size_t)na, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_lw, lw.data(), sizeof(int32_t) * (size_t)na, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_rw, rw.data(), sizeof(int32_t) * (size_t)na, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_woff, woff.data(), sizeof(unsigned long long) * ((size_t)na + 1), hipMemcpyHostToDevice, c->stream));
    LAUNCH(c, "cns_gather_ops", cns_gather_ops, (unsigned)std::min<size_t>(((size_t)na + 3) / 4, (size_t)c->num_cus * 32), 256, 0, (const uint32_t*)d_ops,
           (const int32_t*)d_sel, (const int32_t*)d_lw, (const int32_t*)d_rw, (const unsigned long long*)d_woff, (int)na, row_words, d_pack);
    uint32_t* ops = nullptr;                                   // page-locked: the copy is one DMA
    HIPCHK(hipHostMalloc((void**)&ops, sizeof(uint32_t) * std::max<size_t>(dense_words, 1), hipHostMallocDefault));
    struct HostFree { uint32_t* p; ~HostFree() { (void)hipHostFree(p); } } ops_guard{ops};
    HIPCHK(hipMemcpyAsync(ops, d_pack, sizeof(uint32_t) * dense_words, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));

    lap(4);
    // 6. strings: m5qaln / m5saln from the columns and the reads, then normalize_gaps(push = true)
    mhip_cns_accepted* A = (mhip_cns_accepted*)malloc(sizeof(mhip_cns_accepted) * (size_t)na);
    if (!A) { mhip_set_error("out of memory"); return -1; }
    int64_t sbytes = 0;
    for (int64_t a = 0; a < na; ++a) {
        const mhip_cns_result& r = res[(size_t)sel[(size_t)a]];
        A[a].aln_size = r.last_col - r.first_col;          // O(ND) columns are matches or indels: normalising adds no columns
        A[a].str_offset = sbytes;
        sbytes += 2 * ((int64_t)A[a].aln_size + 1);
    }
    // (gigabytes at config 2: the buffer a caller released before is handed out again when it fits, see strbuf_get)
    char* S = (size_t)sbytes >= ((size_t)64 << 20) ? strbuf_get((size_t)sbytes) : (char*)malloc((size_t)std::max<int64_t>(sbytes, 1));
    if (!S) { free(A); mhip_set_error("out of memory (%lld bytes of aligned strings)", (long long)sbytes); return -1; }
    parallel_for(num_templates, num_threads, [&](int64_t t) {
        std::vector<char> qbuf, tbuf, qtmp, ttmp;
        for (int64_t a = afirst[(size_t)t]; a < afirst[(size_t)t + 1]; ++a) {
            const int64_t ji = sel[(size_t)a];
            const mhip_cns_result& r = res[(size_t)ji];
            const mhip_aln_job& jb = jobs[(size_t)ji];
            const mhip_ext_candidate& ec = cands[tmpl_begin[t] + (ji - jfirst[(size_t)t])];
            mhip_cns_accepted& o = A[a];
            o.template_index = (int32_t)t;
            o.cand_index = tmpl_begin[t] + (ji - jfirst[(size_t)t]);
            o.qid = ec.qid; o.sid = ec.sid;
            o.qoff = r.qoff; o.qend = r.qend; o.soff = r.soff; o.send = r.send;
            char* qa = S + o.str_offset;
            char* sa = qa + o.aln_size + 1;
            const uint32_t* left = ops + woff[(size_t)a];
            const uint32_t* right = left + lw[(size_t)a];
            const mhip_offset_t qo = vol->h_offs[(size_t)jb.qid_local], so = vol->h_offs[(size_t)jb.sid_local];
            // the bases the columns cover, as characters: query (in the mapped strand's orientation) from the untrimmed start point,
            // template likewise — the column loop below then only picks from them
            const int nq = r.query_end - r.query_start, nt = r.target_end - r.target_start;
            qbuf.resize((size_t)nq + 24);             // (8 bytes in front of the bases and 8 + the decoders' overrun behind them: aln_strings.h)
            tbuf.resize((size_t)nt + 24);
            if (jb.chain) decode_rc(host_pac, qo.offset, qo.size, r.query_start, nq, qbuf.data() + 8);
            else decode_fwd(host_pac, qo.offset, r.query_start, nq, qbuf.data() + 8);
            decode_fwd(host_pac, so.offset, r.target_start, nt, tbuf.data() + 8);
            // merged columns: reverse(left) then right, four at a time (aln_strings.h); columns [first_col, last_col) are kept (GetAlignment's
            // trimming)
            {
                const int ncols = r.left_cols + r.right_cols;
                qtmp.resize((size_t)ncols + 16);
                ttmp.resize((size_t)ncols + 16);
                alnstr::build(left, r.left_cols, right, r.right_cols, qbuf.data() + 8, tbuf.data() + 8, qtmp.data() + 8, ttmp.data() + 8);
                memcpy(qa, qtmp.data() + 8 + r.first_col, (size_t)o.aln_size);
                memcpy(sa, ttmp.data() + 8 + r.first_col, (size_t)o.aln_size);
            }
            qa[o.aln_size] = 0;
            sa[o.aln_size] = 0;
            push_gaps(qa, sa, o.aln_size);
        }
    });
    lap(5);
    if (times)
        fprintf(stderr, "[cns_accept] %d templates, %lld jobs, %lld accepted: sort + checks %.3f s, jobs %.3f, re-alignment on the device %.3f, accept replay %.3f, "
                        "columns gathered + copied %.3f, strings %.3f\n", num_templates, (long long)nj, (long long)na, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5]);
    *out_accepted = A;
    *out_count = na;
    *out_strings = S;
    *out_strings_bytes = sbytes;
    return 0;
}

}  // extern "C"
