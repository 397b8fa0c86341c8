// main.cpp — mecat2pw: all-vs-all long-read overlapper, MI355X host driver (drop-in for the reference binary).
//
// Same process boundary as the reference: command line (options.cpp), wrk/vol<k> + wrk/fileindex.txt
// (volume.cpp), per-reference-volume result files wrk/r_<i>.working -> wrk/r_<i> with skip-if-finished resume, final
// `cat` in volume order (reference: mecat2pw/pw.cpp:14-85), `.can` lines (common/alignment.cpp:18-32) and `.m4` lines
// (mecat2pw/pw_impl.cpp:509-531).  What the reference does in its pthread workers (pw_impl.cpp:623-818) is done by three
// calls into libmecat_hip.so per (reference volume, query volume) grid cell: index build, seed_reads, align_candidates.
// Record assembly, the per-read m4 post-filter (std::sort + containment, pw_impl.cpp:539-610) and text output stay on
// the host (`-t` threads; the same option sizes the FASTA reader's thread pool).  There is no CPU fallback for the kernels: any failure aborts with the library's message.
//
// Additive, environment-only knobs:  MECAT_HIP_DEVICE=<n> (default 0),  MECAT_HIP_SLAB=<reads per seed call>,
// WORLD_SIZE / RANK / LOCAL_RANK (or MECAT_HIP_WORLD / MECAT_HIP_RANK): one process per GPU (see "Multi-GPU mode" below),
// MECAT_HIP_SHARD=rows|cells, MECAT_HIP_SHARD_CHUNK=<reads>, MECAT_HIP_RUN_ID=<token>, MECAT_HIP_COMM=file (test hook).
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <errno.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "mecat_hip.h"
#include "options.h"
#include "partition.h"
#include "volume.h"

// multi-process runs: a rank that dies leaves this marker so that the ranks waiting for its files stop too
static char g_fail_marker[1024] = "";
static void leave_fail_marker() {
    if (!g_fail_marker[0]) return;
    const int fd = open(g_fail_marker, O_CREAT | O_WRONLY, 0644);
    if (fd >= 0) close(fd);
}
#define DIE(...)                                                  \
    do {                                                          \
        fprintf(stderr, "[%s, %u] ", __func__, __LINE__);         \
        fprintf(stderr, __VA_ARGS__);                             \
        fprintf(stderr, "\n");                                    \
        leave_fail_marker();                                      \
        abort();                                                  \
    } while (0)
#define MCHK(call)                                                 \
    do {                                                           \
        if ((call) != 0) DIE("%s failed: %s", #call, mhip_last_error()); \
    } while (0)

struct ScopedTimer {   // DynamicTimer, common/defs.h:175-191
    std::string name;
    struct timeval t0;
    explicit ScopedTimer(const std::string& n) : name(n) { fprintf(stderr, "[%s] begins.\n", name.c_str()); gettimeofday(&t0, NULL); }
    ~ScopedTimer() {
        struct timeval t1;
        gettimeofday(&t1, NULL);
        fprintf(stderr, "[%s] takes %.2f secs.\n", name.c_str(), t1.tv_sec - t0.tv_sec + 1e-6 * (t1.tv_usec - t0.tv_usec));
    }
};

// extra phase timings on stderr, only with MECAT_TRACE set (the reference prints none of these)
struct TraceTimer {
    const char* name;
    struct timeval t0;
    bool on;
    explicit TraceTimer(const char* n) : name(n), on(getenv("MECAT_TRACE") != NULL) { if (on) gettimeofday(&t0, NULL); }
    ~TraceTimer() {
        if (!on) return;
        struct timeval t1;
        gettimeofday(&t1, NULL);
        fprintf(stderr, "[trace] %-16s %.3f s\n", name, t1.tv_sec - t0.tv_sec + 1e-6 * (t1.tv_usec - t0.tv_usec));
    }
};

struct M4Record {      // common/alignment.h:21-37
    int64_t qid, sid;
    double ident;
    int vscore, qdir;
    int64_t qoff, qend, qsize;
    int sdir;
    int64_t soff, send, ssize, qext, sext;
};

struct CmpM4ByQidAndOvlpSize {   // pw_impl.cpp:539-548 ; std::sort keeps libstdc++'s tie order like the reference
    bool operator()(const M4Record& a, const M4Record& b) const {
        if (a.qid != b.qid) return a.qid < b.qid;
        const int64_t qa = a.qend - a.qoff, sa = a.send - a.soff, qb = b.qend - b.qoff, sb = b.send - b.soff;
        const int o1 = (int)std::min(qa, sa), o2 = (int)std::min(qb, sb);
        return o1 > o2;
    }
};

// pw_impl.cpp:550-574
static void check_records_containment(const M4Record* v, int s, int e, std::vector<int>& valid) {
    const int soft = 100;
    for (int i = s; i < e; ++i) {
        if (!valid[i]) continue;
        const int qb1 = (int)v[i].qoff, qe1 = (int)v[i].qend, sb1 = (int)v[i].soff, se1 = (int)v[i].send;
        for (int j = i + 1; j < e; ++j) {
            if (!valid[j]) continue;
            if (v[i].sdir != v[j].sdir) continue;
            const int qb2 = (int)v[j].qoff, qe2 = (int)v[j].qend, sb2 = (int)v[j].soff, se2 = (int)v[j].send;
            if (qb2 + soft >= qb1 && qe2 - soft <= qe1 && sb2 + soft >= sb1 && se2 - soft <= se1) valid[j] = 0;
        }
    }
}

// "%d" of printf, without printf: the digits of v at p, returns the position behind them
static inline char* put_int(char* p, int v) {
    unsigned int u = (unsigned int)v;
    if (v < 0) { *p++ = '-'; u = 0u - u; }
    char tmp[12];
    int n = 0;
    do { tmp[n++] = (char)('0' + u % 10u); u /= 10u; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}

// growable array in page-locked host memory (contents are not preserved across a grow: every user refills it)
template <typename T>
struct PinnedBuf {
    // buffers that cross the PCIe link: 2 MB-aligned huge pages, touched here and then page-locked (mhip_host_register) — 100 MB in a few
    // milliseconds, where a hipHostMalloc of the size takes 25 - 60 ms (tools/dev/probes/pin_probe.hip)
    T* p = nullptr;
    size_t cap = 0, n = 0;
    bool locked = false;
    void release() {
        if (!p) return;
        if (locked) mhip_host_unregister(p);
        free(p);
        p = nullptr;
        locked = false;
    }
    ~PinnedBuf() { release(); }
    void resize(size_t want) {
        if (want > cap) {
            release();
            cap = want + want / 8 + 1024;
            const size_t huge = (size_t)2 << 20, bytes = (cap * sizeof(T) + huge - 1) & ~(huge - 1);
            void* q = nullptr;
            if (posix_memalign(&q, huge, bytes) != 0) DIE("out of memory (%zu bytes of transfer buffer)", bytes);
            (void)madvise(q, bytes, MADV_HUGEPAGE);
            for (size_t o = 0; o < bytes; o += huge) ((volatile char*)q)[o] = 0;
            locked = mhip_host_register(q, bytes) == 0;      // (not locked: the copies still work, slower)
            p = (T*)q;
        }
        n = want;
    }
    T* data() { return p; }
    size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

template <typename F>
static void run_threads(int nt, F f) {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(f, t);
    f(0);
    for (auto& x : th) x.join();
}

// Query volumes stay RESIDENT across the grid: row i visits volumes i .. V - 1, row i + 1 visits i + 1 .. V - 1 again, and loading a 535 MB
// volume file, page-locking it, uploading it and freeing it again cost ~0.09 s per cell — 16 of the 43 s of config 5's `-j 0` run (190
// cells).  A volume is uploaded once per process and kept (device: the packed bytes and read table; host: the read table the formatter
// needs) while the resident volumes stay inside a budget of a quarter of the device memory (MECAT_HIP_VOLCACHE_MB overrides; 0 turns
// the cache off); volumes beyond the budget are loaded per cell as before.
struct ResidentVolume {
    HostVolume hv;               // pac released once the bytes are on the device (and the volume file, if still being written, is done)
    mhip_volume* dv = NULL;
    size_t bytes = 0;
};
static std::vector<ResidentVolume*> g_resident;
static size_t g_resident_bytes = 0;
static void resident_clear() {
    for (ResidentVolume* r : g_resident)
        if (r) { if (r->dv) mhip_volume_free(r->dv); delete r; }
    g_resident.clear();
    g_resident_bytes = 0;
}
// volume `vid`, on the device: from the cache, or loaded + uploaded now (and kept when it fits the budget: *cached says so; a volume that
// is not kept is the caller's to free, host part in *own_host, device part in the return value)
static mhip_volume* resident_get(mhip_ctx* ctx, const std::vector<std::string>& vn, int vid, const HostVolume** hv_out, HostVolume* own_host, bool* cached) {
    if (g_resident.size() < vn.size()) g_resident.resize(vn.size(), NULL);
    if (g_resident[(size_t)vid]) { *hv_out = &g_resident[(size_t)vid]->hv; *cached = true; return g_resident[(size_t)vid]->dv; }
    size_t budget;
    if (const char* e = getenv("MECAT_HIP_VOLCACHE_MB")) budget = (size_t)std::max(0L, atol(e)) << 20;
    else {
        size_t free_b = 0, total_b = 0;
        MCHK(mhip_ctx_mem_info(ctx, &free_b, &total_b));
        budget = total_b / 4;
    }
    HostVolume tmp;
    { TraceTimer tt("load_volume"); load_volume(vn[(size_t)vid], &tmp); }
    mhip_volume* dv = NULL;
    {
        TraceTimer tt("volume_upload");
        // (the packed bytes sit in huge pages the packer / reader has touched: locking them takes a few milliseconds and the copy then runs
        // at the link's rate instead of through the runtime's staging buffers)
        const bool locked = !tmp.pac.empty() && mhip_host_register(tmp.pac.data(), tmp.pac.size()) == 0;
        MCHK(mhip_volume_upload(ctx, tmp.pac.data(), tmp.offs.data(), tmp.num_reads, tmp.num_bases, tmp.start_read_id, &dv));
        if (locked) mhip_host_unregister(tmp.pac.data());
    }
    const size_t bytes = tmp.pac.size() + sizeof(mhip_offset_t) * tmp.offs.size();
    if (g_resident_bytes + bytes <= budget) {
        ResidentVolume* r = new ResidentVolume();
        r->hv = std::move(tmp);
        // the bytes live on the device now — unless the file of the volume that stayed in memory is still being written from these very
        // bytes (one-volume runs: nothing waits for that write; the host copy then goes with the cache)
        if (!volume_dump_in_flight()) { std::vector<uint8_t, NoInitAlloc<uint8_t>> none; r->hv.pac.swap(none); }
        r->dv = dv;
        r->bytes = bytes;
        g_resident[(size_t)vid] = r;
        g_resident_bytes += bytes;
        *hv_out = &r->hv;
        *cached = true;
        return dv;
    }
    *own_host = std::move(tmp);
    *hv_out = own_host;
    *cached = false;
    return dv;
}

// comm == NULL: this process computes the whole grid row.  Otherwise every rank of the communicator runs this function for the
// same row: the reads of a slab are dealt out in chunks (chunk c of query volume j -> rank (c + j) mod P), each rank seeds and
// extends its own, the lists are all-gathered (mhip_seed_reads_sharded / mhip_align_sharded), and every rank formats and writes the
// lines of its own reads into its part of r_<i> (out = that part).
static void process_one_volume(const Options& opt, mhip_ctx* ctx, int svid, const std::vector<std::string>& vn, FILE* out, PartitionWriter* pw,
                               double part_ratio, mhip_comm* comm, int shard_chunk) {
    mhip_params P;
    mhip_params_default(&P, opt.tech);
    P.maxc = opt.num_candidates;
    P.min_align_size = opt.min_align_size;
    P.min_kmer_match = opt.min_kmer_match;

    HostVolume ref_own;
    const HostVolume* refp = NULL;
    bool ref_cached = false;
    mhip_volume* dref = resident_get(ctx, vn, svid, &refp, &ref_own, &ref_cached);
    const HostVolume& ref = *refp;

    const char* slab_env = getenv("MECAT_HIP_SLAB");
    // 20 000 reads per slab; nanopore extension: 60 000 — an X-drop call ends with the tail of its longest units (the waves pull units,
    // longest first, from one cursor: the last ones run on a chip that is emptying), so fewer, larger calls: 2 per config-5 cell instead
    // of 6.  (The second launch that used to end every call — rounds 1-3, the first reason for this size — is gone since round 4.)
    int slab = slab_env ? std::max(1, atoi(slab_env)) : (opt.tech == TECH_NANOPORE && opt.task == TASK_ALN ? 60000 : 20000);
    if (comm) slab = std::max(shard_chunk, slab - slab % shard_chunk);      // slabs start on chunk boundaries
    const bool writes = out != NULL;
    // One process, PacBio gates: slabs of SHRINKING size — 70 % of the reads that are left, down to 2 000.  Formatting a slab takes a third
    // of the time its extension takes, so slab s is always written out before slab s + 1 comes off the GPU and only the LAST slab's
    // formatting is exposed at the end of the volume: the smaller it is the better, while every slab costs a fixed few milliseconds on
    // the device (the second extension launch for the handed-over units, the tails of the launches, the copies): 70 000 / 21 000 /
    // 6 300 / 2 700 reads at config 2 instead of five slabs of 20 000.  MECAT_HIP_SLAB keeps a fixed size.
    const bool shrinking = !slab_env && !comm && !(opt.tech == TECH_NANOPORE && opt.task == TASK_ALN);
    auto slab_len = [&](int rb, int end) {
        const int left = end - rb;
        if (!shrinking) return std::min(slab, left);
        int s2 = std::max(2000, (int)(0.7 * left));
        if (left - s2 < 2000) s2 = left;
        return std::min(s2, left);
    };

    struct SlabBuf {
        PinnedBuf<mhip_candidate> cands;      // buffers that cross the PCIe link: page-locked
        PinnedBuf<int32_t> counts;
        PinnedBuf<mhip_aln_job> jobs;
        PinnedBuf<mhip_aln_result> res;
        std::vector<size_t> jfirst;       // first entry of read r's list in res[] (and in cands[] when packed)
        int rb = 0, nr = 0;
        const HostVolume* rd = NULL;      // the query volume the slab belongs to, and its number (the writer thread works across cells)
        int vid = 0;
        bool packed = false;              // cands[] holds only the occupied entries, read-major (one process); else [nr][maxc]
    };
    SlabBuf slabs[2];                         // slab s is written out while slab s + 1 is on the GPU
    // page-locking ~100 MB per slab buffer takes 30-60 ms each: done on a second thread while the volume goes up and is indexed
    std::thread prealloc;
    if (writes)
        prealloc = std::thread([&]() {
            const size_t rows = (size_t)std::max(1, slab_len(0, std::max(ref.num_reads, 1)));      // (other query volumes of the row are no larger; buffers grow when one is)
            for (SlabBuf& B : slabs) {
                B.cands.resize(comm ? rows * (size_t)P.maxc : rows * 32);       // packed lists in a one-process run (grown when a slab holds more)
                B.counts.resize(rows);
                if (opt.task != TASK_SEED) B.res.resize(rows * 32);       // (grown when a slab holds more candidates)
            }
        });
    mhip_index* idx = NULL;
    {
        ScopedTimer t("create_ref_index");
        // cells mode: the ranks either build the table together, each the buckets of its own k-mer key range, and gather positions and
        // table slices (mhip_index_build_sharded) — a replicated rebuild is the part of a sharded cell that does not shrink with the
        // number of GPUs — or every rank rebuilds it for itself (with few ranks the 4.7 GB of positions over one or two xGMI links can
        // cost more than the rebuild).  The first table of a run is built both ways, timed, and the faster way is kept
        // (mhip_index_build_auto; MECAT_HIP_INDEX_SHARD=0 / 1 decides without measuring).
        if (comm) MCHK(mhip_index_build_auto(comm, dref, &idx, NULL, NULL));
        else MCHK(mhip_index_build(ctx, dref, &idx));
    }
    printf("number of kmers: %lld\n", (long long)mhip_index_num_kmers(idx));

    if (prealloc.joinable()) prealloc.join();

        // One writer for the whole grid row: the text of cell n's last slabs is assembled while cell n + 1 is being seeded (a slab carries the
        // query volume it belongs to) — at `-j 0` a cell is one seeding call and then nothing but copies and formatting, which used to run with
        // the GPU idle: 3 of the 29 s of config 5's 190 cells.
        double st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_shown[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // MECAT_TRACE (… [6] waits for a slab buffer, [7] device memory query): seconds in seeding, job assembly, extension, formatting, writing, page-locked buffers
        struct StageClock {
            double* acc; double t0;
            static double now() { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec + 1e-6 * t.tv_usec; }
            explicit StageClock(double* a) : acc(a), t0(now()) {}
            ~StageClock() { *acc += now() - t0; }
        };
        // Two-stage pipeline over the slabs: this thread drives the GPU (seeding, job assembly, extension) for slab s + 1 while a
        // second thread formats and writes slab s (text assembly is per read and order preserving; one writer keeps the order).
        const int nt = std::max(1, std::min(opt.num_threads, 64));
        auto emit = [&](SlabBuf& B) {
            PinnedBuf<mhip_candidate>& cands = B.cands;
            PinnedBuf<int32_t>& counts = B.counts;
            PinnedBuf<mhip_aln_result>& res = B.res;
            std::vector<size_t>& jfirst = B.jfirst;
            const int rb = B.rb, nr = B.nr;
            // cells mode: every rank holds the slab's complete tables (the all-gather of the candidate lists and results) and formats
            // and writes the reads of its OWN chunks — chunk c of query volume vid belongs to rank (c + vid) mod P — into its own part
            // of r_<i>; rank 0 strings the parts together (main).  The line order of r_<i> is then "by rank" instead of "by read": the
            // multiset of lines is the contract, the reference's own order depends on its thread timing.
            const int c_rank = comm ? mhip_comm_rank(comm) : 0, c_world = comm ? mhip_comm_nranks(comm) : 1;
            auto mine = [&](int r) { return c_world == 1 || ((rb + r) / shard_chunk + B.vid) % c_world == c_rank; };
            std::vector<std::string> text((size_t)nt);
            auto range_of = [&](int t, int* lo, int* hi) { *lo = (int)((long long)nr * t / nt); *hi = (int)((long long)nr * (t + 1) / nt); };
            if (opt.task == TASK_SEED) {
                StageClock sc(&st[3]);
                // candidate_detect, pw_impl.cpp:767-801 ; line format alignment.cpp:18-32
                std::vector<std::vector<CanRec>> prec(pw ? (size_t)nt : 0);
                run_threads(nt, [&](int t) {
                    int lo, hi;
                    range_of(t, &lo, &hi);
                    std::string& o = text[(size_t)t];
                    char line[160];
                    for (int r = lo; r < hi; ++r) {
                        if (!mine(r)) continue;
                        const int qsize = B.rd->offs[(size_t)(rb + r)].size, qid = rb + r + B.rd->start_read_id;
                        const size_t c0 = B.packed ? jfirst[(size_t)r] : (size_t)r * P.maxc;
                        for (int k = 0; k < counts[(size_t)r]; ++k) {
                            const mhip_candidate& c = cands[c0 + k];
                            int qext = c.loc2, sext = c.loc1;
                            if (qext && sext) { qext += MHIP_KMER_SIZE / 2; sext += MHIP_KMER_SIZE / 2; }
                            const int ssize = ref.offs[(size_t)(c.readno - ref.start_read_id)].size;
                            if (c.chain == 1) qext = qsize - 1 - qext;
                            // ("%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n": nine integers a line, two million lines at config 2 — snprintf was most of the
                            // drop-in's -j 0 wall time behind the device)
                            char* p = line;
                            const int f[9] = {qid, c.readno, c.chain, 0, qext, sext, c.score, qsize, ssize};
                            for (int q = 0; q < 9; ++q) { p = put_int(p, f[q]); *p++ = q == 8 ? '\n' : '\t'; }
                            o.append(line, (size_t)(p - line));
                            if (pw) prec[(size_t)t].push_back(CanRec{qid, c.readno, c.chain, 0, qext, sext, c.score, qsize, ssize});
                        }
                    }
                });
                for (const std::string& o : text)
                    if (!o.empty() && fwrite(o.data(), 1, o.size(), out) != o.size()) DIE("write error!");
                if (pw)      // the same lines, in the same order, as records (no text round trip)
                    for (const std::vector<CanRec>& v : prec) pw->add(v.data(), v.size());
                return;
            }
            StageClock* sc_fmt = new StageClock(&st[3]);
            std::vector<std::vector<M4Rec>> mrec(pw ? (size_t)nt : 0);
            run_threads(nt, [&](int t) {
                int lo, hi;
                range_of(t, &lo, &hi);
                std::string& o = text[(size_t)t];
                std::vector<M4Record> m4v;
                std::vector<int> valid;
                char line[320];
                for (int r = lo; r < hi; ++r) {
                    if (!mine(r)) continue;
                    const int qsize = B.rd->offs[(size_t)(rb + r)].size, qid = rb + r + B.rd->start_read_id;
                    size_t ji = jfirst[(size_t)r];
                    const size_t c0 = B.packed ? jfirst[(size_t)r] : (size_t)r * P.maxc;
                    m4v.clear();
                    for (int k = 0; k < counts[(size_t)r]; ++k, ++ji) {
                        const mhip_aln_result& a = res[ji];
                        if (!a.ok) continue;
                        const mhip_candidate& c = cands[c0 + k];
                        // the job of this candidate (the loop head of pairwise_mapping, pw_impl.cpp:674-686): made on the device when
                        // the candidate lists stay there, so its fields are derived here rather than read
                        mhip_aln_job j;
                        j.sid_local = c.readno - ref.start_read_id;
                        j.qstart = c.loc2;
                        j.sstart = c.loc1;
                        if (j.qstart && j.sstart) { j.qstart += MHIP_KMER_SIZE / 2; j.sstart += MHIP_KMER_SIZE / 2; }
                        const int ssize = ref.offs[(size_t)j.sid_local].size;
                        M4Record m;     // fill_m4record, pw_impl.cpp:467-506
                        m.qid = c.readno;
                        m.sid = qid;
                        m.ident = a.columns == 0 ? 0.0 : 100.0 * a.matches / a.columns;   // OutputStore::calc_ident / XdropAligner::calc_ident
                        m.vscore = c.score;
                        m.qdir = 0;
                        m.qoff = a.target_start;
                        m.qend = a.target_end;
                        m.qsize = ssize;
                        m.ssize = qsize;
                        m.qext = j.sstart;
                        if (c.chain == 0) { m.sdir = 0; m.soff = a.query_start; m.send = a.query_end; m.sext = j.qstart; }
                        else { m.sdir = 1; m.soff = qsize - a.query_end; m.send = qsize - a.query_start; m.sext = qsize - 1 - j.qstart; }
                        m4v.push_back(m);
                    }
                    // append_m4v, pw_impl.cpp:576-610
                    std::sort(m4v.begin(), m4v.end(), CmpM4ByQidAndOvlpSize());
                    const int n = (int)m4v.size();
                    valid.assign((size_t)n, 1);
                    for (int i = 0; i < n;) {
                        int e = i + 1;
                        while (e < n && m4v[(size_t)e].qid == m4v[(size_t)i].qid) ++e;
                        if (e - i > 1) check_records_containment(m4v.data(), i, e, valid);
                        i = e;
                    }
                    for (int i = 0; i < n; ++i) {
                        if (!valid[(size_t)i]) continue;
                        const M4Record& m = m4v[(size_t)i];
                        int w = snprintf(line, 256, "%lld\t%lld\t%g\t%d\t%d\t%lld\t%lld\t%lld\t%d\t%lld\t%lld\t%lld", (long long)m.qid,
                                         (long long)m.sid, m.ident, m.vscore, m.qdir, (long long)m.qoff, (long long)m.qend,
                                         (long long)m.qsize, m.sdir, (long long)m.soff, (long long)m.send, (long long)m.ssize);
                        if (opt.output_gapped_start_point) w += snprintf(line + w, 64, "\t%lld\t%lld", (long long)m.qext, (long long)m.sext);
                        line[w++] = '\n';
                        o.append(line, (size_t)w);
                        if (pw)
                            mrec[(size_t)t].push_back(M4Rec{(int32_t)m.qid, (int32_t)m.sid, m.vscore, m.qdir, (int32_t)m.qoff, (int32_t)m.qend,
                                                            (int32_t)m.qsize, m.sdir, (int32_t)m.soff, (int32_t)m.send, (int32_t)m.ssize,
                                                            (int32_t)m.qext, (int32_t)m.sext});
                    }
                }
            });
            delete sc_fmt;
            {
                StageClock sc(&st[4]);
                for (const std::string& o : text)
                    if (!o.empty() && fwrite(o.data(), 1, o.size(), out) != o.size()) DIE("write error!");
                if (pw)
                    for (const std::vector<M4Rec>& v : mrec) pw->add_m4(v.data(), v.size(), part_ratio);
            }
        };
        std::mutex pm;
        std::condition_variable pcv;
        int produced = 0, consumed = 0;
        bool closing = false;
        std::thread writer([&]() {
            if (!writes) return;
            for (;;) {
                int s;
                {
                    std::unique_lock<std::mutex> lk(pm);
                    pcv.wait(lk, [&]() { return consumed < produced || closing; });
                    if (consumed >= produced) return;
                    s = consumed;
                }
                emit(slabs[s & 1]);
                {
                    std::lock_guard<std::mutex> lk(pm);
                    ++consumed;
                }
                pcv.notify_all();
            }
        });
    int sno = 0;      // slabs of the row so far (the two slab buffers alternate across cells as well)
    for (int vid = svid; vid < (int)vn.size(); ++vid) {
        char info[64];
        snprintf(info, sizeof(info), "process volume %d", vid);
        ScopedTimer t(info);
        fprintf(stderr, "[%s, %u] processing %s\n\n", __func__, __LINE__, vn[vid].c_str());
        HostVolume rd_store;
        const HostVolume* rd = &ref;
        mhip_volume* dreads = dref;
        bool rd_cached = true;
        if (vid != svid) dreads = resident_get(ctx, vn, vid, &rd, &rd_store, &rd_cached);
        // candidate_detect aborts on a read of MAX_SEQ_SIZE bases or more (pw_impl.cpp:743-746); pairwise_mapping would
        // overrun its MAX_SEQ_SIZE buffers there.  Same limit, same message, for both tasks.
        for (int r = 0; r < rd->num_reads; ++r)
            if (rd->offs[(size_t)r].size >= MHIP_MAX_SEQ_SIZE) {
                printf("rsize = %d\t%d\n", rd->offs[(size_t)r].size, MHIP_MAX_SEQ_SIZE);
                fflush(stdout);
                abort();
            }
        // One process: the candidate lists of the whole cell are made in one go and stay in HBM; a slab is then job assembly and
        // extension on the device plus the copies the text needs (a seeding call per slab cost 5 x 17 ms instead of 59 at config 2,
        // and the host-side job assembly kept the GPU waiting).  With a communicator the sharded calls below do all of this.
        // The resident table is [reads][MAXC] records of 48 bytes: a volume of short reads at a large -n would not fit (2 M reads at
        // -n 1024: 100 GB), so the cell is seeded in super-slabs — a whole number of slabs whose table stays inside a budget taken from
        // the free device memory (a quarter of it, at most 32 GB; MECAT_HIP_CELL_MB overrides) — one super-slab = the whole cell whenever
        // it fits (config 2: 0.48 GB).
        void *d_cell_cands = NULL, *d_cell_counts = NULL;
        int cell_first = 0, cell_reads = 0, super_reads = rd->num_reads;
        if (!comm) {
            size_t free_b = 0, total_b = 0;
            { StageClock sc(&st[7]); MCHK(mhip_ctx_mem_info(ctx, &free_b, &total_b)); }
            size_t budget = std::min<size_t>(free_b / 4, (size_t)32 << 30);
            if (const char* e = getenv("MECAT_HIP_CELL_MB")) budget = (size_t)std::max(1L, atol(e)) << 20;
            const size_t per_read = sizeof(mhip_candidate) * (size_t)P.maxc + sizeof(int32_t);
            const size_t fit = shrinking ? std::max<size_t>(2000, budget / per_read) : std::max<size_t>(1, budget / per_read / (size_t)slab) * (size_t)slab;
            super_reads = (int)std::min<size_t>((size_t)std::max(rd->num_reads, 1), fit);
        }
        auto seed_super = [&](int first) {
            StageClock sc(&st[0]);
            cell_first = first;
            cell_reads = std::min(super_reads, rd->num_reads - first);
            MCHK(mhip_ctx_buffer(ctx, "cell_cands", sizeof(mhip_candidate) * (size_t)std::min(super_reads, rd->num_reads) * P.maxc, &d_cell_cands));
            MCHK(mhip_ctx_buffer(ctx, "cell_counts", sizeof(int32_t) * (size_t)std::min(super_reads, rd->num_reads), &d_cell_counts));
            MCHK(mhip_seed_reads_dev(ctx, idx, dref, dreads, first, first + cell_reads, &P, d_cell_cands, d_cell_counts));
            MCHK(mhip_ctx_sync(ctx));
        };
        bool first_of_cell = true;
        for (int rb = 0, step = 0; rb < rd->num_reads; rb += step, ++sno) {
            // (a slab never straddles two super-slabs of the resident table: it ends where the super-slab that holds its first read ends)
            const int super_end = comm ? rd->num_reads : std::min(rd->num_reads, (rb / super_reads + 1) * super_reads);
            step = slab_len(rb, super_end);
            const int re = rb + step, nr = re - rb;
            {
                StageClock sc(&st[6]);
                std::unique_lock<std::mutex> lk(pm);                  // the buffers of slab sno - 2 must have been written out
                pcv.wait(lk, [&]() { return consumed >= sno - 1; });
            }
            SlabBuf& B = slabs[sno & 1];
            PinnedBuf<mhip_candidate>& cands = B.cands;
            PinnedBuf<int32_t>& counts = B.counts;
            PinnedBuf<mhip_aln_result>& res = B.res;
            std::vector<size_t>& jfirst = B.jfirst;
            B.rb = rb;
            B.nr = nr;
            B.packed = !comm;
            B.rd = rd;
            B.vid = vid;
            if (!comm && (first_of_cell || rb >= cell_first + cell_reads)) seed_super(rb);      // (slabs never straddle super-slabs)
            first_of_cell = false;
            const int cb = rb - cell_first;                                                // the slab inside the resident table
            if (writes) {
                StageClock sc(&st[5]);
                if (comm) cands.resize((size_t)nr * P.maxc);
                counts.resize((size_t)nr);
            }
            if (comm) {
                StageClock sc(&st[0]);
                MCHK(mhip_seed_reads_sharded(comm, idx, dref, dreads, rb, re, shard_chunk, vid, &P, writes ? cands.data() : NULL,
                                             writes ? counts.data() : NULL));
            } else {
                StageClock sc(&st[4]);      // (copies: booked with the writing)
                // the counts, and the occupied entries of the lists packed on the device (a list is ~22 of its 100 slots)
                MCHK(mhip_download(ctx, counts.data(), (const int32_t*)d_cell_counts + cb, sizeof(int32_t) * (size_t)nr));
                jfirst.assign((size_t)nr + 1, 0);
                for (int r = 0; r < nr; ++r) jfirst[(size_t)r + 1] = jfirst[(size_t)r] + (size_t)counts[(size_t)r];
                void* d_pack = NULL;
                int64_t total = 0;
                MCHK(mhip_ctx_buffer(ctx, "slab_pack", sizeof(mhip_candidate) * (size_t)nr * P.maxc, &d_pack));
                MCHK(mhip_pack_candidates_dev(ctx, (const mhip_candidate*)d_cell_cands + (size_t)cb * P.maxc, (const int32_t*)d_cell_counts + cb, nr, P.maxc,
                                              d_pack, &total));
                if ((size_t)total != jfirst[(size_t)nr]) DIE("%lld packed candidates for %zu counted", (long long)total, jfirst[(size_t)nr]);
                cands.resize((size_t)total);
                MCHK(mhip_download(ctx, cands.data(), d_pack, sizeof(mhip_candidate) * (size_t)total));
            }
            if (opt.task != TASK_SEED && comm && !writes) {
                StageClock sc(&st[2]);
                int64_t nj = 0;
                MCHK(mhip_align_sharded(comm, dref, dreads, opt.tech == TECH_NANOPORE ? 1 : 0, P.min_align_size, NULL, &nj));
            } else if (opt.task != TASK_SEED && !comm) {
                // pairwise_mapping, pw_impl.cpp:674-700, with the jobs made on the device from the lists that are there
                { StageClock sc(&st[5]); res.resize(jfirst[(size_t)nr]); }
                void *d_jobs = NULL, *d_res = NULL;
                int nj = 0;
                {
                    StageClock sc(&st[1]);
                    MCHK(mhip_ctx_buffer(ctx, "slab_jobs", sizeof(mhip_aln_job) * (size_t)nr * P.maxc, &d_jobs));
                    MCHK(mhip_jobs_from_candidates_dev(ctx, (const mhip_candidate*)d_cell_cands + (size_t)cb * P.maxc, (const int32_t*)d_cell_counts + cb, nr,
                                                       P.maxc, rb, 1, ref.start_read_id, 0, 1, d_jobs, &nj));
                    if ((size_t)nj != jfirst[(size_t)nr]) DIE("%d jobs for %zu candidates", nj, jfirst[(size_t)nr]);
                }
                {
                    StageClock sc(&st[2]);
                    MCHK(mhip_ctx_buffer(ctx, "slab_results", sizeof(mhip_aln_result) * (size_t)std::max(nj, 1), &d_res));
                    // aligner by technology (pw_impl.cpp:638-644): DiffAligner (dw) for PacBio, XdropAligner for nanopore
                    if (opt.tech == TECH_NANOPORE) MCHK(mhip_xalign_candidates_dev(ctx, dref, dreads, d_jobs, nj, P.min_align_size, d_res));
                    else MCHK(mhip_align_candidates_dev(ctx, dref, dreads, d_jobs, nj, P.min_align_size, d_res));
                    MCHK(mhip_download(ctx, res.data(), d_res, sizeof(mhip_aln_result) * (size_t)nj));
                }
            } else if (opt.task != TASK_SEED) {
                PinnedBuf<mhip_aln_job>& jobs = B.jobs;
                auto range_of = [&](int t, int* lo, int* hi) { *lo = (int)((long long)nr * t / nt); *hi = (int)((long long)nr * (t + 1) / nt); };
                // pairwise_mapping, pw_impl.cpp:674-700
                StageClock* sc_jobs = new StageClock(&st[1]);
                jfirst.assign((size_t)nr + 1, 0);
                for (int r = 0; r < nr; ++r) jfirst[(size_t)r + 1] = jfirst[(size_t)r] + (size_t)counts[(size_t)r];
                jobs.resize(jfirst[(size_t)nr]);
                run_threads(nt, [&](int t) {
                    int lo, hi;
                    range_of(t, &lo, &hi);
                    for (int r = lo; r < hi; ++r) {
                        size_t jn = jfirst[(size_t)r];
                        for (int k = 0; k < counts[(size_t)r]; ++k) {
                            const mhip_candidate& c = cands[(size_t)r * P.maxc + k];
                            mhip_aln_job j;
                            j.qid_local = rb + r;
                            j.sid_local = c.readno - ref.start_read_id;
                            j.chain = c.chain;
                            j.qstart = c.loc2;
                            j.sstart = c.loc1;
                            if (j.qstart && j.sstart) { j.qstart += MHIP_KMER_SIZE / 2; j.sstart += MHIP_KMER_SIZE / 2; }
                            jobs[jn++] = j;
                        }
                    }
                });
                res.resize(jobs.size());
                delete sc_jobs;
                {
                    StageClock sc(&st[2]);
                    // aligner by technology (pw_impl.cpp:638-644): DiffAligner (dw) for PacBio, XdropAligner for nanopore
                    if (comm) {
                        int64_t nj = 0;
                        MCHK(mhip_align_sharded(comm, dref, dreads, opt.tech == TECH_NANOPORE ? 1 : 0, P.min_align_size, res.data(), &nj));
                        if ((size_t)nj != jobs.size()) DIE("sharded extension returned %lld results for %zu candidates", (long long)nj, jobs.size());
                    } else if (opt.tech == TECH_NANOPORE) MCHK(mhip_xalign_candidates(ctx, dref, dreads, jobs.data(), (int)jobs.size(), P.min_align_size, res.data()));
                    else MCHK(mhip_align_candidates(ctx, dref, dreads, jobs.data(), (int)jobs.size(), P.min_align_size, res.data()));
                }
            }
            if (sno == 0) volume_release_input();      // every scratch array of the volume exists by now
            if (writes) {
                std::lock_guard<std::mutex> lk(pm);
                ++produced;
            } else {
                std::lock_guard<std::mutex> lk(pm);      // nothing to write on this rank: the buffers are free again at once
                ++produced;
                ++consumed;
            }
            pcv.notify_all();
        }
        if (!rd_cached) {      // this cell's query volume goes away with the cell: its slabs have to be written out first
            std::unique_lock<std::mutex> lk(pm);
            pcv.wait(lk, [&]() { return consumed >= produced; });
        }
        if (getenv("MECAT_TRACE")) {      // (what the clocks gathered since the last line; formatting of this cell's tail shows up in the next line)
            fprintf(stderr, "[trace] volume %d stages: seed %.3f s, jobs %.3f s, extend %.3f s, format %.3f s, write + copies %.3f s, page-locked buffers %.3f s, slab buffer waits %.3f s, memory query %.3f s\n",
                    vid, st[0] - st_shown[0], st[1] - st_shown[1], st[2] - st_shown[2], st[3] - st_shown[3], st[4] - st_shown[4], st[5] - st_shown[5], st[6] - st_shown[6], st[7] - st_shown[7]);
            for (int k = 0; k < 8; ++k) st_shown[k] = st[k];
        }
        if (dreads != dref && !rd_cached) mhip_volume_free(dreads);
    }
    {
        std::lock_guard<std::mutex> lk(pm);
        closing = true;
    }
    pcv.notify_all();
    writer.join();
    if (getenv("MECAT_TRACE") && (st[3] - st_shown[3] > 0.0005 || st[4] - st_shown[4] > 0.0005))
        fprintf(stderr, "[trace] volume -1 stages: seed %.3f s, jobs %.3f s, extend %.3f s, format %.3f s, write + copies %.3f s, page-locked buffers %.3f s\n",
                st[0] - st_shown[0], st[1] - st_shown[1], st[2] - st_shown[2], st[3] - st_shown[3], st[4] - st_shown[4], st[5] - st_shown[5]);
    mhip_index_free(idx);
    if (!ref_cached) mhip_volume_free(dref);
    volume_wait_pending();       // the volume's file is written from `ref`'s buffers
}

static std::string results_name(const char* wrk_dir, int vid, bool working) {
    std::string s(wrk_dir);
    if (s.empty() || s[s.size() - 1] != '/') s += '/';
    s += "r_" + std::to_string(vid);
    if (working) s += ".working";
    return s;
}

static int env_int(const char* a, const char* b, int dflt) {
    const char* e = getenv(a);
    if (!e && b) e = getenv(b);
    return e ? atoi(e) : dflt;
}

static double now_s() {
    struct timeval t;
    gettimeofday(&t, NULL);
    return t.tv_sec + 1e-6 * t.tv_usec;
}

// Multi-GPU mode (additive): P processes, one per GPU, started with WORLD_SIZE / RANK / LOCAL_RANK in the environment (e.g.
// `python -m torch.distributed.run --no-python --nproc-per-node 8 mecat2pw ...`) or MECAT_HIP_WORLD / MECAT_HIP_RANK.  Rank 0
// splits the input, merges the r_<i> files and writes the output; hand-offs between the processes are files in wrk_dir, like
// the resume protocol itself.  Two ways to share the volume x volume grid (MECAT_HIP_SHARD=rows|cells overrides the choice):
//   rows   (#volumes >= P)  the grid rows still to do are dealt out by cost (row i = num_vols - i cells, heaviest first to the least
//          loaded rank: mhip_shard_deal_rows), each row computed by one GPU exactly as in a single-GPU run (one index build per row, no
//          data moves between the processes);
//   cells  (#volumes <  P)  every rank works on every cell: the query reads of a cell are dealt out in chunks of 500
//          (MECAT_HIP_SHARD_CHUNK; chunk c of query volume j -> rank (c + j) mod P), each rank builds the index of
//          the row's reference volume itself, and the candidate lists / extension results are all-gathered over RCCL
//          (mhip_seed_reads_sharded, mhip_align_sharded); every rank formats and writes the lines of its own reads (r_<i>.part<rank>),
//          rank 0 strings the parts together into r_<i>.  A one-volume input (config 2) uses every GPU.
// A run is identified by a token (MECAT_HIP_RUN_ID, else the launcher's TORCHELASTIC_RUN_ID + the parent pid, which every rank
// of one launch shares): rank 0 puts it into the split marker, the other ranks accept no other marker.  Ranks above 0 keep a
// heartbeat file fresh and leave a failure marker when they abort; rank 0 stops waiting for a row whose owner has died.
struct RunFiles {
    std::string dir, token;
    std::string marker() const { return dir + "split_done"; }
    std::string alive(int r) const { return dir + "rank_" + std::to_string(r) + ".alive." + token; }
    std::string failed(int r) const { return dir + "rank_" + std::to_string(r) + ".failed." + token; }
};

static std::string run_token() {
    if (const char* e = getenv("MECAT_HIP_RUN_ID")) return std::string("x") + e;
    std::string t = "p" + std::to_string((long)getppid());
    if (const char* e = getenv("TORCHELASTIC_RUN_ID")) t += std::string("_") + e;
    if (const char* e = getenv("MASTER_PORT")) t += std::string("_") + e;
    // an elastic restart by the same launcher keeps all of the above: the restart count tells the attempts apart, so that a rank of
    // attempt k + 1 never accepts the split marker (communicator id, rows to do) attempt k left behind
    if (const char* e = getenv("TORCHELASTIC_RESTART_COUNT")) t += std::string("_r") + e;
    for (char& ch : t)
        if (!isalnum((unsigned char)ch) && ch != '_' && ch != '-') ch = '_';
    return t;
}

static std::string to_hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; ++i) { s += d[p[i] >> 4]; s += d[p[i] & 15]; }
    return s;
}
static bool from_hex(const std::string& s, uint8_t* p, size_t n) {
    if (s.size() != 2 * n) return false;
    auto v = [](char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : -1; };
    for (size_t i = 0; i < n; ++i) {
        const int a = v(s[2 * i]), b = v(s[2 * i + 1]);
        if (a < 0 || b < 0) return false;
        p[i] = (uint8_t)(a * 16 + b);
    }
    return true;
}

// r_<i> appended to (first: replacing) the output file, copied inside the kernel.  false = not an ordinary file (or the copy could
// not be made this way): the caller falls back to the reference's `cat`.
static bool merge_copy(const std::string& fin, const char* output, bool first) {
    struct stat so;
    const int have = lstat(output, &so);
    if (have == 0 && !S_ISREG(so.st_mode)) return false;
    if (have != 0 && errno != ENOENT) return false;
    if (have == 0 && first && unlink(output) != 0) return false;
    if (have == 0 && !first && so.st_nlink > 1) return false;
    const int in = open(fin.c_str(), O_RDONLY);
    if (in < 0) return false;
    const int out = open(output, O_WRONLY | O_CREAT | (first ? O_EXCL : 0), 0666);
    if (out < 0) { close(in); return false; }
    struct stat si;
    bool ok = fstat(in, &si) == 0 && lseek(out, 0, SEEK_END) >= 0;
    off_t left = ok ? si.st_size : 0;
    const off_t at0 = ok ? lseek(out, 0, SEEK_CUR) : 0;
    bool plain = false;
    while (ok && left > 0) {
        const ssize_t n = plain ? -1 : copy_file_range(in, NULL, out, NULL, (size_t)std::min<off_t>(left, (off_t)1 << 30), 0);
        if (n > 0) { left -= n; continue; }
        if (n == 0) break;
        // no copy_file_range between these two files (EXDEV, EINVAL, ENOSYS ...): plain read / write from where it stopped
        plain = true;
        static char buf[1 << 20];
        const ssize_t r = read(in, buf, sizeof(buf));
        if (r < 0) { ok = false; break; }
        if (r == 0) break;
        for (ssize_t w = 0; w < r;) {
            const ssize_t k = write(out, buf + w, (size_t)(r - w));
            if (k <= 0) { ok = false; break; }
            w += k;
        }
        left -= r;
    }
    ok = ok && left == 0;
    close(in);
    if (close(out) != 0) ok = false;
    if (!ok) {
        if (first) unlink(output);
        else if (truncate(output, at0) != 0) DIE("write error on '%s'", output);
    }
    return ok;
}

int main(int argc, char* argv[]) {
    Options opt;
    if (parse_arguments(argc, argv, &opt)) {
        print_usage(argv[0]);
        return 1;
    }
    const int world = std::max(1, env_int("MECAT_HIP_WORLD", "WORLD_SIZE", 1));
    const int rank = std::min(world - 1, std::max(0, env_int("MECAT_HIP_RANK", "RANK", 0)));
    const double t_start = now_s();
    RunFiles rf;
    rf.dir = opt.wrk_dir;
    if (rf.dir.empty() || rf.dir[rf.dir.size() - 1] != '/') rf.dir += '/';
    rf.token = run_token();
    const bool explicit_token = getenv("MECAT_HIP_RUN_ID") != NULL;
    const std::string marker = rf.marker();
    // The GPU context and the index-build scratch (two arrays of 8 bytes per base of a volume: hundreds of milliseconds to map
    // at volume size) are set up on a second thread while this one parses the input.
    mhip_ctx* ctx = NULL;
    std::atomic<int> ctx_state{0};                       // 0 pending, 1 ready, -1 failed
    std::string ctx_error;
    const int device = env_int("MECAT_HIP_DEVICE", world > 1 ? "LOCAL_RANK" : NULL, 0);
    long long est_bases = 0;
    {
        struct stat sb;
        FILE* f = fopen(opt.reads, "rb");
        if (f && fstat(fileno(f), &sb) == 0) {
            const int c0 = fgetc(f);
            est_bases = (long long)sb.st_size / (c0 == '@' ? 2 : 1);      // FASTQ carries a quality byte per base
        }
        if (f) fclose(f);
        est_bases = std::min(est_bases, (long long)kMaxVolumeBases + 64);
        if (getenv("MECAT_HIP_NO_RESERVE")) est_bases = 0;
    }
    std::thread gpu_setup([&]() {
        TraceTimer tt("ctx_create+reserve (background)");
        mhip_ctx* made = NULL;
        int rc = -1;
        for (int attempt = 0; attempt < 3 && rc != 0; ++attempt) {      // several processes opening a cold device at once can see a transient failure
            if (attempt) usleep(300 * 1000);
            rc = mhip_ctx_create(device, NULL, &made);
        }
        if (rc != 0) {
            ctx_error = mhip_last_error();
            ctx_state.store(-1);
            return;
        }
        ctx = made;
        ctx_state.store(1);
        if (est_bases > 0) (void)mhip_ctx_reserve_index(made, est_bases);      // best effort
    });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } gpu_setup_joiner{gpu_setup};

    // heartbeat + failure marker of every rank of a multi-process run, and a watchdog on the peers: a rank that sees another
    // rank's failure marker (or, once that rank's heartbeat has been seen, no beat for a minute) leaves its own marker and exits —
    // also when its main thread sits inside an RCCL collective the dead rank will never join (cells mode had no timeout)
    std::atomic<bool> beat_stop{false};
    std::thread beat;
    if (world > 1) {
        snprintf(g_fail_marker, sizeof(g_fail_marker), "%s", rf.failed(rank).c_str());
        unlink(g_fail_marker);
        signal(SIGSEGV, [](int) { leave_fail_marker(); _exit(139); });
        signal(SIGTERM, [](int) { leave_fail_marker(); _exit(143); });
        signal(SIGABRT, [](int) { leave_fail_marker(); _exit(134); });      // the reader's and the splitter's ERROR() abort, like the reference's
        signal(SIGBUS, [](int) { leave_fail_marker(); _exit(135); });       // (an mmap'd input that shrank under the reader)
        const std::string alive = rf.alive(rank);
        std::vector<std::string> peer_failed, peer_alive;
        for (int r = 0; r < world; ++r)
            if (r != rank) { peer_failed.push_back(rf.failed(r)); peer_alive.push_back(rf.alive(r)); }
        const bool watch = !getenv("MECAT_HIP_NO_WATCHDOG");
        // A token can be reused (a fixed MECAT_HIP_RUN_ID, ranks started by hand from one shell), and a failed attempt leaves every
        // rank's failure marker and possibly a stale heartbeat behind.  Every rank removes its OWN failure marker when it
        // starts, so a peer's marker says "the peer failed in the last attempt it started": younger than this process — this attempt, believed
        // at once; older — either the attempt before (and the peer of this one has not started yet: it will remove it), or this attempt
        // with the peer started, and dead, more than a launcher's skew before this rank (that peer also removed its heartbeat,
        // so nothing else would ever notice).  The two are told apart by waiting: an older marker that is still there `grace` seconds
        // after this rank started (MECAT_HIP_PEER_GRACE_S, default 120: the time after which a peer without a heartbeat counts as "not
        // running" anyway, so a peer that is merely late — slow launcher, GPU lease wait, ranks started by hand — is not declared dead any
        // earlier on account of a stale marker) is believed.  The 60 s silence rule applies to a peer only once a
        // heartbeat of THIS attempt has been seen from it.
        const double t_mine = t_start - 1.0;                     // st_mtime has one-second granularity on some file systems
        const double grace = env_int("MECAT_HIP_PEER_GRACE_S", NULL, 120);
        beat = std::thread([&beat_stop, alive, peer_failed, peer_alive, watch, rank, t_mine, grace]() {
            std::vector<char> seen(peer_alive.size(), 0);
            auto mtime_of = [](const struct stat& sb) { return (double)sb.st_mtim.tv_sec + 1e-9 * (double)sb.st_mtim.tv_nsec; };
            while (!beat_stop.load()) {
                const int fd = open(alive.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0644);
                if (fd >= 0) { (void)!write(fd, "1\n", 2); close(fd); }
                for (int i = 0; i < 20 && !beat_stop.load(); ++i) {
                    usleep(100 * 1000);
                    if (!watch || i % 5) continue;
                    for (size_t k = 0; k < peer_failed.size(); ++k) {
                        struct stat sb;
                        bool dead = stat(peer_failed[k].c_str(), &sb) == 0 && (mtime_of(sb) >= t_mine || now_s() - (t_mine + 1.0) > grace);
                        const char* why = "left a failure marker";
                        if (!dead && stat(peer_alive[k].c_str(), &sb) == 0) {
                            if (mtime_of(sb) >= t_mine) seen[k] = 1;
                            if (seen[k] && now_s() - mtime_of(sb) > 60.0) { dead = true; why = "stopped responding"; }
                        }
                        if (dead && !beat_stop.load()) {
                            fprintf(stderr, "[mecat2pw rank %d] a peer %s (%s): stopping\n", rank, why, peer_failed[k].c_str());
                            leave_fail_marker();
                            unlink(alive.c_str());
                            _exit(1);
                        }
                    }
                }
            }
            unlink(alive.c_str());
        });
    }
    struct BeatJoiner { std::atomic<bool>& stop; std::thread& t; ~BeatJoiner() { stop.store(true); if (t.joinable()) t.join(); } } beat_joiner{beat_stop, beat};

    const int shard_chunk = std::max(1, env_int("MECAT_HIP_SHARD_CHUNK", NULL, MHIP_SHARD_CHUNK));
    const bool file_comm = getenv("MECAT_HIP_COMM") && !strcmp(getenv("MECAT_HIP_COMM"), "file");
    int num_vols = 0;
    bool cells = false;
    std::vector<int> todo;                                // grid rows this run still has to compute
    uint8_t comm_id[MHIP_COMM_ID_BYTES];
    memset(comm_id, 0, sizeof(comm_id));
    if (rank == 0) {
        if (world > 1) unlink(marker.c_str());
        for (int r = 0; world > 1 && r < world; ++r) unlink(partition_meta_name(opt.output, r).c_str());      // (no rank has started its partition streams yet)
        volume_set_async_dump(world == 1);       // other ranks read the volume files as soon as the run marker exists
        volume_set_device_packer([&]() -> mhip_ctx* {      // (only used under MECAT_HIP_SPLIT=gpu)
            if (gpu_setup.joinable()) gpu_setup.join();      // (the set-up thread still uses the context after it has published it)
            return ctx_state.load() == 1 ? ctx : NULL;
        });
        num_vols = split_raw_dataset(opt.reads, opt.wrk_dir, opt.num_threads);
        for (int i = 0; i < num_vols; ++i)
            if (access(results_name(opt.wrk_dir, i, false).c_str(), F_OK) != 0) todo.push_back(i);
        if (world > 1) {
            const char* se = getenv("MECAT_HIP_SHARD");
            cells = se ? !strcmp(se, "cells") : num_vols < world;
            if (cells && !file_comm) MCHK(mhip_comm_unique_id(comm_id));
            FILE* m = fopen((marker + ".tmp").c_str(), "w");
            if (!m) DIE("cannot write '%s'", marker.c_str());
            fprintf(m, "%s %.3f %d %d %s\n", rf.token.c_str(), t_start, num_vols, cells ? 1 : 0, to_hex(comm_id, sizeof(comm_id)).c_str());
            for (int i : todo) fprintf(m, "%d ", i);
            fprintf(m, "\n");
            fclose(m);
            if (rename((marker + ".tmp").c_str(), marker.c_str()) != 0) DIE("cannot rename %s", marker.c_str());
        }
    } else {
        // wait for THIS run's split: the marker carries the run token (and, for a token that was not given explicitly, must be
        // younger than this process: a restart by the same parent would reuse the token)
        const double wait_limit = env_int("MECAT_HIP_WAIT_S", NULL, 6 * 3600);
        for (;;) {
            FILE* m = fopen(marker.c_str(), "r");
            char tok[256] = "", hex[2 * MHIP_COMM_ID_BYTES + 8] = "";      // 264 bytes: %263s below
            double t0 = 0;
            int nv = 0, cl = 0;
            const bool ok = m && fscanf(m, "%255s %lf %d %d %263s", tok, &t0, &nv, &cl, hex) == 5;
            if (ok && rf.token == tok && (explicit_token || t0 > t_start - 120.0) && from_hex(hex, comm_id, sizeof(comm_id))) {
                int v;
                while (fscanf(m, "%d", &v) == 1) todo.push_back(v);
                fclose(m);
                num_vols = nv;
                cells = cl != 0;
                break;
            }
            if (m) fclose(m);
            if (now_s() - t_start > wait_limit) DIE("rank %d: no split marker of run '%s' in %s after %.0f s", rank, rf.token.c_str(), opt.wrk_dir, wait_limit);
            usleep(20 * 1000);
        }
    }
    const std::string idx_name = index_file_name(opt.wrk_dir);
    if (rank == 0) printf("%s\n", idx_name.c_str());
    const std::vector<std::string> vn = load_volume_names(idx_name);
    if ((int)vn.size() != num_vols) DIE("assertion 'num_vols == vn->num_vols' failed");

    {
        TraceTimer tt("wait for ctx_create");
        // only the context is needed from here on; a reservation still in flight is waited for inside the library
        while (ctx_state.load() == 0) usleep(500);
        if (ctx_state.load() < 0) { if (gpu_setup.joinable()) gpu_setup.join(); DIE("cannot use the GPU: %s", ctx_error.c_str()); }
    }
    mhip_comm* comm = NULL;
    if (cells) {
        TraceTimer tt("comm_init");
        if (file_comm) MCHK(mhip_comm_init_hostfile(ctx, world, rank, opt.wrk_dir, rf.token.c_str(), &comm));
        else MCHK(mhip_comm_init(ctx, world, rank, comm_id, &comm));
        MCHK(mhip_comm_barrier(comm));
    }

    // MECAT_HIP_PARTITION=<batch_size>[,<min_read_size>[,<mapping_ratio>]] (additive): also write mecat2cns' partition files
    // <output>.part<k> + <output>.partition_files (partition.h) — partition_candidates for -j 0, partition_m4records for
    // -j 1 -g 1.  Records are taken straight from the result arrays: in a one-process run by the one writer, in a multi-process run by a
    // writer per rank whose streams rank 0 merges at the end (partition.h: same bytes as the one-process files, no text parsed).  Only
    // after a resume — finished rows' records are on disk as text alone — the merged text is partitioned.
    long part_batch = 0;
    int part_min = opt.tech == TECH_NANOPORE ? 2000 : 5000;      // mecat2cns defaults -l and -r, options.cpp:13-27
    double part_ratio = opt.tech == TECH_NANOPORE ? 0.4 : 0.9;
    if (const char* pe = getenv("MECAT_HIP_PARTITION")) {
        if (opt.task == TASK_ALN && !opt.output_gapped_start_point)
            DIE("MECAT_HIP_PARTITION with -j 1 needs -g 1 (mecat2cns reads the gapped start points)");
        part_batch = atol(pe);
        if (const char* comma = strchr(pe, ',')) {
            part_min = atoi(comma + 1);
            if (const char* c2 = strchr(comma + 1, ',')) part_ratio = atof(c2 + 1);
        }
        if (part_batch <= 0) DIE("MECAT_HIP_PARTITION: batch size must be positive");
    }
    part_ratio = part_ratio - 0.02;                               // reads_correction_m4.cpp:79
    // rows mode: the static, cost-aware deal of the rows still to do (mhip_shard_deal_rows: row i costs num_vols - i cells; every rank
    // derives it from the split marker's list alone)
    std::vector<int> row_owner((size_t)num_vols, 0);
    if (world > 1 && !cells) {
        const int heaviest = mhip_shard_deal_rows(num_vols, todo.data(), (int)todo.size(), world, row_owner.data());
        if (heaviest < 0) DIE("cannot deal %d rows to %d ranks", (int)todo.size(), world);
        if (rank == 0 && getenv("MECAT_TRACE")) {
            long total = 0;
            for (int i : todo) total += num_vols - i;
            fprintf(stderr, "[trace] rows dealt: heaviest rank %d cells of %ld (mean %.2f)\n", heaviest, total, (double)total / world);
        }
    }
    PartitionWriter* pw = part_batch > 0 ? new PartitionWriter(opt.output, part_batch, part_min, world > 1 ? rank : -1) : NULL;
    if (pw && (int)todo.size() != num_vols) { pw->abandon(); delete pw; pw = NULL; }      // finished rows' records are only on disk (every rank sees the same list)
    for (int i = 0; i < num_vols; ++i) {
        if (std::find(todo.begin(), todo.end(), i) == todo.end()) {
            if (rank == 0) fprintf(stderr, "[%s, %u] volume %d has been finished\n\n", __func__, __LINE__, i);
            continue;
        }
        if (!cells && world > 1 && row_owner[(size_t)i] != rank) continue;      // rows: dealt out by cost
        // rows: the owner of row i writes r_<i>.  cells: every rank writes the lines of its own reads to r_<i>.part<rank>, and once all
        // parts are closed rank 0 strings them together into r_<i>.working -> r_<i> (the resume protocol sees only complete rows)
        const std::string fin = results_name(opt.wrk_dir, i, false), wrk = results_name(opt.wrk_dir, i, true);
        const std::string mine = cells ? fin + ".part" + std::to_string(rank) : wrk;
        FILE* out = fopen(mine.c_str(), "w");
        if (!out) DIE("failed to open file '%s' with mode 'ios::out'", mine.c_str());
        if (pw) pw->set_row(i);
        process_one_volume(opt, ctx, i, vn, out, pw, part_ratio, comm, shard_chunk);
        if (fclose(out) != 0) DIE("write error!");
        if (cells) {
            MCHK(mhip_comm_barrier(comm));
            if (rank == 0) {
                TraceTimer tt("parts -> r_<i>");
                for (int r = 0; r < world; ++r) {
                    const std::string part = fin + ".part" + std::to_string(r);
                    if (!merge_copy(part, wrk.c_str(), r == 0)) DIE("cannot append '%s' to '%s'", part.c_str(), wrk.c_str());
                    unlink(part.c_str());
                }
            }
        }
        if (!cells || rank == 0)
            if (rename(wrk.c_str(), fin.c_str()) != 0) DIE("cannot rename %s", wrk.c_str());
    }
    if (comm) {
        MCHK(mhip_comm_barrier(comm));
        mhip_comm_destroy(comm);
    }
    if (gpu_setup.joinable()) gpu_setup.join();
    {
        TraceTimer tt("ctx_destroy");
        resident_clear();
        mhip_ctx_destroy(ctx);
    }
    if (getenv("MECAT_TRACE")) fprintf(stderr, "[trace] main up to here     %.3f s\n", now_s() - t_start);
    if (rank != 0) {
        if (pw) { pw->finish(); delete pw; }      // this rank's record streams, complete (rank 0 merges them)
        return 0;
    }

    // merge_results, pw.cpp:34-46 (rank 0; in rows mode it waits for the rows of the other ranks, but not for a dead one)
    TraceTimer* tt_merge = new TraceTimer("merge_results");
    const double merge_wait = env_int("MECAT_HIP_WAIT_S", NULL, 6 * 3600);
    for (int i = 0; i < num_vols; ++i) {
        const std::string fin = results_name(opt.wrk_dir, i, false);
        const int owner = world > 1 && !cells && row_owner[(size_t)i] >= 0 ? row_owner[(size_t)i] : 0;
        const double w0 = now_s();
        while (world > 1 && !cells && access(fin.c_str(), F_OK) != 0) {
            struct stat sb;
            // (by now every rank that started in this attempt has removed the failure marker of an earlier one: a marker that is there is
            // this attempt's, or that of a rank that never started — dead either way; same rule as the watchdog's)
            if (stat(rf.failed(owner).c_str(), &sb) == 0 && ((double)sb.st_mtime >= t_start - 1.0 || now_s() - t_start > env_int("MECAT_HIP_PEER_GRACE_S", NULL, 120)))
                DIE("rank %d failed before it finished volume %d", owner, i);
            const double now = now_s();
            if (stat(rf.alive(owner).c_str(), &sb) == 0) {
                if (now - (double)sb.st_mtime > 60.0) DIE("rank %d stopped responding (volume %d unfinished)", owner, i);
            } else if (now - w0 > 120.0 && access(fin.c_str(), F_OK) != 0) {
                DIE("rank %d is not running (no heartbeat; volume %d unfinished)", owner, i);
            }
            if (now - w0 > merge_wait) DIE("gave up waiting for volume %d of rank %d after %.0f s", i, owner, merge_wait);
            usleep(50 * 1000);
        }
        // The reference runs `cat r_<i> > output` / `>> output` through the shell (pw.cpp:34-46).  Same bytes here, copied inside the
        // kernel (copy_file_range: a reflink where the file system has one) when the output is, or is going to be, an ordinary file; an
        // existing output that is not a regular file (a FIFO, /dev/stdout) is written through by the shell as before.  The output never
        // shares an inode with r_<i> (a hard link would let a later `cat r_0 > output` of a resumed reference run truncate
        // both), and an old output is never written through: it is removed first, so a name that is a second link to some other file
        // cannot damage that file.  MECAT_HIP_MERGE=cat keeps the shell.
        if (!(getenv("MECAT_HIP_MERGE") && !strcmp(getenv("MECAT_HIP_MERGE"), "cat")) && merge_copy(fin, opt.output, i == 0)) continue;
        if (i == 0) {
            struct stat so;
            if (lstat(opt.output, &so) == 0 && S_ISREG(so.st_mode) && so.st_nlink > 1) unlink(opt.output);
        }
        const std::string cmd = std::string("cat ") + fin + (i == 0 ? " >" : " >> ") + opt.output;
        if (system(cmd.c_str()) != 0) DIE("'%s' failed", cmd.c_str());
    }
    delete tt_merge;
    if (pw) {
        TraceTimer tt("partition_files");
        pw->finish();
        delete pw;
        if (world > 1) {
            // the other ranks' streams are complete when their meta files are there (same patience, and the same eye on failure markers, as
            // for their rows above)
            const double w0 = now_s();
            for (int r = 1; r < world; ++r)
                while (access(partition_meta_name(opt.output, r).c_str(), F_OK) != 0) {
                    struct stat sb;
                    if (stat(rf.failed(r).c_str(), &sb) == 0 && ((double)sb.st_mtime >= t_start - 1.0 || now_s() - t_start > env_int("MECAT_HIP_PEER_GRACE_S", NULL, 120)))
                        DIE("rank %d failed before it finished its partition streams", r);
                    if (now_s() - w0 > merge_wait) DIE("gave up waiting for the partition streams of rank %d after %.0f s", r, merge_wait);
                    usleep(20 * 1000);
                }
            partition_merge_ranks(opt.output, world, part_batch);
        }
    } else if (part_batch > 0) {
        TraceTimer tt("partition_files(text)");
        if (opt.task == TASK_SEED) partition_candidates_text(opt.output, part_batch, part_min, opt.num_threads);
        else partition_m4_text(opt.output, part_ratio, part_batch, part_min, opt.num_threads);
    }
    if (getenv("MECAT_TRACE")) fprintf(stderr, "[trace] main returns at    %.3f s\n", now_s() - t_start);
    // Everything this run owes the caller is on disk and closed.  What a plain `return` would still do — the HIP runtime's own teardown
    // (code objects, queues, its threads) and the destructors of this file's statics — takes 40 - 80 ms and produces nothing: leave at once.
    fflush(NULL);
    _exit(0);
}
