// mecat2asmpw / mecat2trimpw (and their *50 variants) of mecat2canu on the MI355X path: a drop-in for
// /root/reference/mecat2canu/src/mecat2asmpw/{mecat2asmpw,mecat2trimpw,mecat2asmpw50,mecat2trimpw50}.c behind their own command line
//     <tool> -P<blocks dir> -T<threads> -S<start block> -E<last block>
// as canu's overlap jobs start them (mecat2canu/src/pipelines/canu/Overlapmecat2asmpw.pm:483-503): reads <dir>/ovlprep (one line
// per block: "-allreads -allbases -b <first read> -e <last read>") and <dir>/%06d.fasta (a header line and ONE sequence line per
// read), indexes block S, maps the reads of blocks S .. E against it and writes <dir>/<S>_<t>.r for t in [0, T) — the files the job
// script then concatenates.  Which of the four tools this is follows from the name it is started under:
//     ...trimpw...   seeding gate > 8 instead of > 10 (:644), jscore = mismatches / (4 * columns) instead of (2 cols - mism) * 120 / cols (:942-943)
//     ...50          at most 50 candidates per read instead of 100 (:23)
// On the device (C ABI, include/mecat_hip.h): the look-up table with bucket cap 256 (mhip_index_build_ex), seeding + candidate
// selection (mhip_asm_seed_reads_ex, asm_seed.hip), the O(ND) extension of every candidate in both directions (mhip_asm_extend_run /
// _fetch, cns_align.hip: the columns come back packed, into page-locked buffers).  On the host, per candidate and on -T threads: what the tool does with the two aligned string pairs afterwards —
// string_check's gap shuffling of the left pair (:199-281), the overlap of the two directions on the seed 13-mer, coordinates, the
// 450-base test, jscore and the 12-field line (:843-948).  Line order inside the .r files is by read here and by thread timing in the
// tool: the consumer (mecat2asmpwConvert) reads lines one by one.
// Letters outside A, C, G, T (N, the other IUPAC codes, anything else a read file holds): the tool restarts its k-mer there in table and
// query (atcttrans gives every one of them 4, :296-304, 445, 486) and aligns it as the character it is — equal only to itself, and its
// own complement (:583-590).  Here such a base travels as a symbol (code, plane): a non-zero value 1..3 in a second 2-bit plane beside
// the volume (mhip_volume_set_nplane) and one of the four codes in the volume itself, twelve symbols in all — N is (0, 3), the others
// are handed out in order of first appearance (the eleven other IUPAC codes fit; a thirteenth distinct character is refused).
#include <ctype.h>
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "mecat_hip.h"

#define DIE(...)                                  \
    do {                                          \
        fprintf(stderr, "[mecat2asmpw] ");        \
        fprintf(stderr, __VA_ARGS__);             \
        fprintf(stderr, "\n");                    \
        exit(1);                                  \
    } while (0)
#define MCHK(call)                                                    \
    do {                                                              \
        if ((call) != 0) DIE("%s failed: %s", #call, mhip_last_error()); \
    } while (0)

static const int SEED = 13;

// symbols of the characters that are not A, C, G, T (see the head of the file): sym = code | plane << 2, plane != 0
#include <mutex>
static std::mutex g_sym_mu;
static int g_sym_of[256];            // 0: none yet
static char g_sym_chr[16];           // the character of symbol (code | plane << 2)
static int g_sym_next = 0;
static int symbol_of(unsigned char ch, const char* path, int read_no) {
    static const int order[11] = {1 | 3 << 2, 2 | 3 << 2, 3 | 3 << 2, 0 | 1 << 2, 1 | 1 << 2, 2 | 1 << 2, 3 | 1 << 2, 0 | 2 << 2, 1 | 2 << 2, 2 | 2 << 2, 3 | 2 << 2};
    std::lock_guard<std::mutex> lk(g_sym_mu);
    if (!g_sym_chr[0 | 3 << 2]) { g_sym_chr[0 | 3 << 2] = 'N'; g_sym_of[(int)'N'] = 0 | 3 << 2; }
    if (g_sym_of[ch]) return g_sym_of[ch];
    if (g_sym_next >= 11)
        DIE("%s: character '%c' in read %d: a thirteenth distinct character besides A, C, G, T (this path carries twelve)", path, ch, read_no);
    const int sym = order[g_sym_next++];
    g_sym_of[ch] = sym;
    g_sym_chr[sym] = (char)ch;
    return sym;
}

struct Reads {                       // one fasta block: 2-bit volume (one pad base after every read) + its host copy
    std::vector<uint8_t> pac;
    std::vector<mhip_offset_t> offs;
    int num_bases = 0, first_no = 0;
    // bases that are not A, C, G, T (an N of a corrected read brought in from elsewhere): code 0 in pac, 3 in this second plane of the same
    // layout (empty without such bases); frag = the read table with a "read" ending at every such base — what the look-up table is built
    // from (the tools restart their k-mer there, :445, 486: no k-mer of the table holds one)
    std::vector<uint8_t> npac;
    std::vector<mhip_offset_t> frag;
    bool has_n = false;
    int base(int r, int i) const {
        const int64_t idx = (int64_t)offs[(size_t)r].offset + i;
        return (pac[(size_t)(idx >> 2)] >> ((~idx & 3) << 1)) & 3;
    }
    int plane(int r, int i) const {
        if (!has_n) return 0;
        const int64_t idx = (int64_t)offs[(size_t)r].offset + i;
        return (npac[(size_t)(idx >> 2)] >> ((~idx & 3) << 1)) & 3;
    }
    bool is_n(int r, int i) const { return plane(r, i) != 0; }
    char chr(int r, int i) const { const int k = plane(r, i); return k ? g_sym_chr[base(r, i) | k << 2] : "ACGT"[base(r, i)]; }
};

// load_read / load_fastq (:388-409, :982-1010): ">header" line, one sequence line; lower case is upper-cased
static void load_block(const std::string& path, int first_no, Reads* R) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) DIE("cannot open '%s': %s", path.c_str(), strerror(errno));
    std::vector<char> buf;
    {
        if (fseek(f, 0, SEEK_END) != 0) DIE("cannot read '%s'", path.c_str());
        const long sz = ftell(f);
        rewind(f);
        buf.resize((size_t)std::max(0L, sz));
        if (sz > 0 && fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) DIE("cannot read '%s'", path.c_str());
        fclose(f);
    }
    static int8_t code[256];
    static bool code_ready = false;
    if (!code_ready) {
        memset(code, -1, sizeof(code));
        code[(int)'A'] = code[(int)'a'] = 0; code[(int)'C'] = code[(int)'c'] = 1; code[(int)'G'] = code[(int)'g'] = 2; code[(int)'T'] = code[(int)'t'] = 3;
        code_ready = true;
    }
    R->first_no = first_no;
    R->offs.clear();
    R->frag.clear();
    R->has_n = false;
    R->pac.assign(buf.size() / 4 + 16, 0);       // (never more bases + pads than bytes in the file)
    R->npac.clear();
    int64_t at = 0;
    bool want_seq = false;
    const char* p = buf.data();
    const char* const end = p + buf.size();
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        const char* next = nl ? nl + 1 : end;
        while (le > p && le[-1] == '\r') --le;
        const int64_t n = le - p;
        if (!want_seq) {
            if (n > 0) {
                if (p[0] != '>') DIE("%s: a header line was expected", path.c_str());
                want_seq = true;
            }
            p = next;
            continue;
        }
        if (at + n + 1 > 2140000000LL) DIE("%s: more than 2.14 G bases in one block", path.c_str());
        mhip_offset_t o;
        o.offset = (int)at;
        o.size = (int)n;
        uint8_t* pac = R->pac.data();
        int64_t fstart = 0;                      // first base of the open fragment (read-local)
        for (int64_t i = 0; i < n; ++i) {
            int c = code[(unsigned char)p[i]];
            const int64_t idx = at + i;
            if (c < 0) {
                // not A, C, G, T: atcttrans() gives it 4 (:296-304: no k-mer over it); the extension compares it as the character it is,
                // upper-cased like everything the tool reads (:398, 992: every byte from 'a' up goes through toupper)
                unsigned char ch = (unsigned char)p[i];
                if (ch >= 'a') ch = (unsigned char)toupper(ch);
                const int sym = symbol_of(ch, path.c_str(), first_no + (int)R->offs.size());
                if (!R->has_n) { R->has_n = true; R->npac.assign(R->pac.size(), 0); R->frag = R->offs; }
                R->npac[(size_t)(idx >> 2)] |= (uint8_t)((sym >> 2) << ((~idx & 3) << 1));
                mhip_offset_t fr;
                fr.offset = (int)(at + fstart);
                fr.size = (int)(i - fstart);
                R->frag.push_back(fr);
                fstart = i + 1;
                c = sym & 3;
            }
            pac[idx >> 2] |= (uint8_t)(c << ((~idx & 3) << 1));
        }
        if (R->has_n) {
            mhip_offset_t fr;
            fr.offset = (int)(at + fstart);
            fr.size = (int)(n - fstart);
            R->frag.push_back(fr);
        }
        at += n + 1;                             // the pad base (code 0) = the tool's NUL behind every read
        R->offs.push_back(o);
        want_seq = false;
        p = next;
    }
    R->num_bases = (int)at;
    R->pac.resize(((size_t)at + 3) / 4);
    if (R->has_n) R->npac.resize(R->pac.size());
}

// string_check (:199-281): gaps of the left pair are moved over runs that also match one column further on.  a / b = the aligned
// pair in extension order, s1 / s2 = the same two sequences without gaps.  S(i) may be one past the end (the C string's NUL).
static void shuffle_gaps(const std::string& s1, const std::string& s2, std::string& a, std::string& b) {
    const int len1 = (int)s1.size() - 1, len2 = (int)s2.size() - 1;
    const char* S1 = s1.c_str();
    const char* S2 = s2.c_str();
    int loc1 = 0, loc2 = 0;
    for (int p = (int)a.size() - 1; p > -1; --p) {
        auto move_run = [&](int k, int o1, int o2) {
            int s = 0, j = p;
            while (s < k && j >= 0) { if (a[(size_t)j] != '-') { a[(size_t)j] = '-'; ++s; } --j; }
            s = 0; j = p;
            while (s < k && j >= 0) { if (b[(size_t)j] != '-') { b[(size_t)j] = '-'; ++s; } --j; }
            for (s = 0, j = p; s < k && j >= 0; --j, ++s) { a[(size_t)j] = S1[o1 - s]; b[(size_t)j] = S2[o2 - s]; }
        };
        if (a[(size_t)p] != '-') ++loc1;
        else if (loc1 <= len1 && loc2 <= len2 && S1[len1 - loc1] == S2[len2 - loc2]) {
            int k = 1;
            while (loc1 + k <= len1 && loc2 + k <= len2 && S1[len1 - loc1 - k] == S2[len2 - loc2 - k]) ++k;
            move_run(k, len1 - loc1, len2 - loc2);
            if (a[(size_t)p] != '-') ++loc1;
        }
        if (b[(size_t)p] != '-') ++loc2;
        else if (a[(size_t)p] != '-' && (loc1 - 1 <= len1 && loc2 <= len2 && S1[len1 - loc1 + 1] == S2[len2 - loc2])) {
            int k = 1;
            while (loc1 + k - 1 <= len1 && loc2 + k <= len2 && S1[len1 - loc1 + 1 - k] == S2[len2 - loc2 - k]) ++k;
            move_run(k, len1 - loc1 + 1, len2 - loc2);
            if (b[(size_t)p] != '-') ++loc2;
        } else if (a[(size_t)p] == '-' && (loc1 - 1 <= len1 && loc2 <= len2 && S1[len1 - loc1] == S2[len2 - loc2])) {
            int k = 1;
            while (loc1 + k <= len1 && loc2 + k <= len2 && S1[len1 - loc1 - k] == S2[len2 - loc2 - k]) ++k;
            move_run(k, len1 - loc1, len2 - loc2);
            if (b[(size_t)p] != '-') ++loc2;
        }
    }
}

struct Tool { int gate = 10, maxc = 100; bool trim = false; };

int main(int argc, char** argv) {
    std::string dir;
    int threads = -1, start = -1, last = -1;
    for (int i = 1; i < argc; ++i) {             // param_read (:1012-1056): -P<dir> -T<n> -S<n> -E<n>, value glued to the letter
        if (argv[i][0] != '-' || !argv[i][1]) { fprintf(stderr, "usage: %s -P<blocks dir> -T<threads> -S<start block> -E<last block>\n", argv[0]); return 1; }
        const char* v = argv[i] + 2;
        switch (argv[i][1]) {
        case 'P': dir = v; break;
        case 'T': threads = atoi(v); break;
        case 'S': start = atoi(v); break;
        case 'E': last = atoi(v); break;
        default: break;
        }
    }
    if (dir.empty() || threads < 1 || start < 1 || last < start) {
        fprintf(stderr, "usage: %s -P<blocks dir> -T<threads> -S<start block> -E<last block>\n", argv[0]);
        return 1;
    }
    Tool tool;
    {
        const char* b = strrchr(argv[0], '/');
        const std::string name = b ? b + 1 : argv[0];
        if (const char* e = getenv("MECAT_ASMPW_TOOL")) { if (strstr(e, "trim")) tool.trim = true; if (strstr(e, "50")) tool.maxc = 50; }
        else { if (name.find("trim") != std::string::npos) tool.trim = true; if (name.find("50") != std::string::npos) tool.maxc = 50; }
        if (tool.trim) tool.gate = 8;
    }
    // ovlprep (:1086-1094): " %s %s %s %d %s %d"
    std::vector<int> first_read, last_read;
    {
        const std::string p = dir + "/ovlprep";
        FILE* f = fopen(p.c_str(), "r");
        if (!f) DIE("cannot open '%s': %s", p.c_str(), strerror(errno));
        char a[300], b2[300], c2[300], d2[300];
        int k, e;
        while ((int)first_read.size() < last && fscanf(f, " %299s %299s %299s %d %299s %d", a, b2, c2, &k, d2, &e) == 6) { first_read.push_back(k); last_read.push_back(e); }
        fclose(f);
        if ((int)first_read.size() < last) DIE("%s names %zu blocks, -E is %d", p.c_str(), first_read.size(), last);
    }
    auto block_path = [&](int id) { char t[32]; snprintf(t, sizeof(t), "/%06d.fasta", id); return dir + t; };

    // MECAT_ASMPW_TIMES=1: where the wall time went, on stderr at the end
    const bool times = getenv("MECAT_ASMPW_TIMES") != NULL;
    double t_load = 0, t_index = 0, t_seed = 0, t_jobs = 0, t_extend = 0, t_host = 0, t_flush = 0, t_drain = 0, t_close = 0;
    size_t n_jobs = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    mhip_ctx* ctx = NULL;
    const int device = getenv("MECAT_HIP_DEVICE") ? atoi(getenv("MECAT_HIP_DEVICE")) : 0;
    // The context comes up (HIP runtime, device, stream) and the -T output files are opened beside the first block's read: neither needs
    // the other, and each is a tenth of a second of waiting (the files are needed by the first chunk's lines, the context by the upload).
    std::thread ctx_thread([&] { MCHK(mhip_ctx_create(device, NULL, &ctx)); });
    std::vector<FILE*> out((size_t)threads);
    std::thread open_thread([&] {
        for (int t = 0; t < threads; ++t) {
            const std::string p = dir + "/" + std::to_string(start) + "_" + std::to_string(t) + ".r";
            out[(size_t)t] = fopen(p.c_str(), "w");
            if (!out[(size_t)t]) DIE("cannot write '%s': %s", p.c_str(), strerror(errno));
        }
    });
    Reads blk;
    double t0 = now();
    load_block(block_path(start), first_read[(size_t)start - 1], &blk);
    t_load += now() - t0; t0 = now();
    ctx_thread.join();
    const double t_ctx = now() - t0;
    t0 = now();
    mhip_volume* dblk = NULL;
    MCHK(mhip_volume_upload(ctx, blk.pac.data(), blk.offs.data(), (int)blk.offs.size(), blk.num_bases, blk.first_no, &dblk));
    t_load += now() - t0; t0 = now();
    mhip_index* idx = NULL;
    if (!blk.has_n) MCHK(mhip_index_build_ex(ctx, dblk, 256, &idx));
    else {
        // the table of a block with such bases: built from the same bytes under a read table that ends a "read" at every one of them (the
        // index never starts a k-mer in the 13 positions in front of a read end: exactly the positions whose 13-mer would hold the base)
        MCHK(mhip_volume_set_nplane(ctx, dblk, blk.npac.data()));
        mhip_volume* dfrag = NULL;
        MCHK(mhip_volume_upload(ctx, blk.pac.data(), blk.frag.data(), (int)blk.frag.size(), blk.num_bases, blk.first_no, &dfrag));
        MCHK(mhip_index_build_ex(ctx, dfrag, 256, &idx));
        mhip_volume_free(dfrag);
    }
    t_index += now() - t0;
    int next_file = 0;

    // Candidates are found for `slab` query reads per launch (one wave per read, 4 096 by default = 16 waves per CU); their extensions run
    // in chunks of at most `chunk` candidates, and the string work of a chunk runs on the -T threads while the device works on the next
    // one — across block boundaries too: a chunk keeps its query block alive.
    const int slab = std::max(1, getenv("MECAT_ASMPW_SLAB") ? atoi(getenv("MECAT_ASMPW_SLAB")) : 4096);
    // MECAT_ASMPW_CHUNK_MB bounds the DEVICE array of a chunk (every direction at its worst-case length there); what crosses the PCIe
    // link is the columns the directions really have, packed (mhip_asm_extend_run / _fetch), into page-locked buffers that grow on demand
    const size_t chunk_bytes = (size_t)std::max(1, getenv("MECAT_ASMPW_CHUNK_MB") ? atoi(getenv("MECAT_ASMPW_CHUNK_MB")) : 256) << 20;
    mhip_asm_candidate* cands = NULL;                // page-locked: a slab's lists are 10 MB, of which a third is filled
    int32_t* counts = NULL;
    MCHK(mhip_host_alloc((size_t)slab * 100 * sizeof(mhip_asm_candidate), (void**)&cands));
    MCHK(mhip_host_alloc((size_t)slab * sizeof(int32_t), (void**)&counts));
    struct Chunk {                                   // one extension call and what the host stage needs of it
        std::vector<mhip_asm_job> jobs;
        std::vector<mhip_asm_candidate> cand;
        std::vector<int> qrid;                       // query read (index in its block) of every job
        std::shared_ptr<const Reads> qkeep;          // the query block, when it is not the subject block
        const Reads* q = NULL;
        int32_t* dirs = NULL;                        // page-locked: [2 jobs][6]
        uint64_t* woffs = NULL;                      // page-locked: [2 jobs + 1] first word of every direction in ops
        size_t jobs_cap = 0;
        uint32_t* ops = NULL;                        // page-locked, ops_cap words
        size_t ops_cap = 0;
        std::thread post;
    } ck[2];
    int cur = 0;
    // ---- per candidate: the tool's string work and its output line (:843-948)
    auto host_stage = [&](Chunk* C) {
        const double th0 = now();
        const size_t nj = C->jobs.size();
        std::vector<std::string> text((size_t)threads);
        std::atomic<size_t> next_job{0};
        auto worker = [&](int t) {
            std::string& o = text[(size_t)t];
            std::string L1, L2, R1, R2, g1, g2, O1, O2;
            char line[256];
            for (;;) {
                const size_t j0 = next_job.fetch_add(64);
                if (j0 >= nj) break;
                for (size_t ji = j0; ji < std::min(nj, j0 + 64); ++ji) {
                    const int qrid = C->qrid[ji], read_len = C->q->offs[(size_t)qrid].size, read_name = C->q->first_no + qrid;
                    const mhip_asm_candidate& c = C->cand[ji];
                    const mhip_asm_job& jb = C->jobs[ji];
                    // characters of the subject read and of the mapped strand (the complement of N is N, :583-590): bases and planes of
                    // the two reads addressed directly, 16 columns of a word at a time
                    const Reads& Q = *C->q;
                    const int64_t xo = blk.offs[(size_t)jb.xid].offset, yo = Q.offs[(size_t)qrid].offset;
                    const uint8_t* const xp = blk.pac.data();
                    const uint8_t* const yp = Q.pac.data();
                    const uint8_t* const xn = blk.has_n ? blk.npac.data() : NULL;
                    const uint8_t* const yn = Q.has_n ? Q.npac.data() : NULL;
                    auto at2 = [](const uint8_t* pl, int64_t idx) -> int { return (pl[(size_t)(idx >> 2)] >> ((~idx & 3) << 1)) & 3; };
                    auto xchr = [&](int pos) -> char {
                        const int64_t idx = xo + pos;
                        const int k = xn ? at2(xn, idx) : 0;
                        return k ? g_sym_chr[at2(xp, idx) | k << 2] : "ACGT"[at2(xp, idx)];
                    };
                    auto ychr = [&](int pos) -> char {
                        const int64_t idx = yo + (jb.chain ? read_len - 1 - pos : pos);
                        const int k = yn ? at2(yn, idx) : 0;
                        if (k) return g_sym_chr[at2(yp, idx) | k << 2];
                        return jb.chain ? "TGCA"[at2(yp, idx)] : "ACGT"[at2(yp, idx)];
                    };
                    auto build = [&](int d, std::string& s1, std::string& s2) {
                        const int cols = C->dirs[(ji * 2 + (size_t)d) * 6];
                        const uint32_t* w = C->ops + C->woffs[ji * 2 + (size_t)d];
                        s1.resize((size_t)cols); s2.resize((size_t)cols);
                        char* const a = &s1[0];
                        char* const b = &s2[0];
                        int x = d ? jb.rx : jb.lx, y = d ? jb.ry : jb.ly;
                        const int step = d ? 1 : -1;
                        for (int m0 = 0; m0 < cols; m0 += 16) {
                            const uint32_t word = w[m0 >> 4];
                            const int n = std::min(16, cols - m0);
                            if (word == 0u) {                              // sixteen columns with both bases
                                for (int m = 0; m < n; ++m) { a[m0 + m] = xchr(x); b[m0 + m] = ychr(y); x += step; y += step; }
                                continue;
                            }
                            for (int m = 0; m < n; ++m) {
                                const int op = (int)((word >> (m << 1)) & 3u);
                                char ca = '-', cb = '-';
                                if (op != 1) { ca = xchr(x); x += step; }
                                if (op != 2) { cb = ychr(y); y += step; }
                                a[m0 + m] = ca; b[m0 + m] = cb;
                            }
                        }
                    };
                    build(0, L1, L2);
                    build(1, R1, R2);
                    {
                        // the two sequences of the left pair without their gaps
                        const size_t n = L1.size();
                        g1.resize(n); g2.resize(n);
                        size_t k1 = 0, k2 = 0;
                        for (size_t m = 0; m < n; ++m) {
                            g1[k1] = L1[m]; k1 += L1[m] != '-';
                            g2[k2] = L2[m]; k2 += L2[m] != '-';
                        }
                        g1.resize(k1); g2.resize(k2);
                    }
                    shuffle_gaps(g1, g2, L1, L2);
                    const int u_k = (int)L1.size();
                    O1.assign(L1.rbegin(), L1.rend());
                    O2.assign(L2.rbegin(), L2.rend());
                    int nl1 = 0, nl2 = 0;
                    for (int m = 0; m < u_k; ++m) { nl1 += L1[(size_t)m] != '-'; nl2 += L2[(size_t)m] != '-'; }
                    int left_loc1, left_loc, right_loc1, right_loc;
                    if (u_k == SEED - 1) { left_loc1 = c.loc1 + SEED - nl1 - 1; left_loc = c.loc2 + SEED - nl2; }
                    else if (u_k > 0) { left_loc1 = c.loc1 + SEED - nl1; left_loc = c.loc2 + SEED - nl2 + 1; }
                    else { left_loc1 = c.loc1; left_loc = c.loc2 + 1; }
                    const int s_k = (int)R1.size();
                    int nr1 = 0, nr2 = 0;
                    for (int m = 0; m < s_k; ++m) { nr1 += R1[(size_t)m] != '-'; nr2 += R2[(size_t)m] != '-'; }
                    if (s_k > 0) { right_loc1 = c.loc1 + nr1 - 1; right_loc = c.loc2 + nr2; }
                    else { right_loc1 = c.loc1 + SEED - 1; right_loc = c.loc2 + SEED; }
                    if (s_k >= SEED && u_k >= SEED) { O1.append(R1, SEED, std::string::npos); O2.append(R2, SEED, std::string::npos); }
                    else if (u_k < SEED) { O1 = R1; O2 = R2; }
                    left_loc1 -= c.readstart;
                    right_loc1 -= c.readstart;
                    if (!(right_loc1 - left_loc1 > 450)) continue;
                    int mism = 0;
                    const int cols = (int)O1.size();
                    for (int m = 0; m < cols; ++m) mism += !(O1[(size_t)m] == O2[(size_t)m] && O2[(size_t)m] != '-');
                    float jscore;
                    if (!tool.trim) { jscore = (float)(2 * cols - mism); jscore = jscore * 30 * 4 / (cols); }
                    else { jscore = (float)mism; jscore = jscore / (4 * cols); }
                    const int sno = blk.first_no + c.readno, slen = blk.offs[(size_t)c.readno].size;
                    int w;
                    if (!jb.chain) w = snprintf(line, sizeof(line), "%d %d %.3f 100 0 %d %d %d 0 %d %d %d\n", sno, read_name, jscore, left_loc1 - 1, right_loc1, slen,
                                                left_loc - 1, right_loc, read_len);
                    else w = snprintf(line, sizeof(line), "%d %d %.3f 100 0 %d %d %d 1 %d %d %d\n", sno, read_name, jscore, left_loc1 - 1, right_loc1, slen,
                                      read_len - right_loc, read_len - left_loc + 1, read_len);
                    o.append(line, (size_t)w);
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; ++t) th.emplace_back(worker, t);
        worker(0);
        for (std::thread& x : th) x.join();
        for (int t = 0; t < threads; ++t) {
            FILE* f = out[(size_t)((next_file + t) % threads)];
            if (!text[(size_t)t].empty() && fwrite(text[(size_t)t].data(), 1, text[(size_t)t].size(), f) != text[(size_t)t].size()) DIE("write error");
        }
        next_file = (next_file + 1) % threads;
        t_host += now() - th0;
    };
    // extend the chunk's candidates (of query volume dq, at most cap columns per direction), then hand it to the host stage
    auto flush = [&](Chunk* C, mhip_volume* dq, int cap) {
        if (C->jobs.empty()) return;
        const double tf0 = now();
        const size_t nj = C->jobs.size();
        if (nj > C->jobs_cap) {
            if (C->dirs) mhip_host_free(C->dirs);
            if (C->woffs) mhip_host_free(C->woffs);
            C->jobs_cap = nj + nj / 4 + 64;
            MCHK(mhip_host_alloc(C->jobs_cap * 2 * 6 * sizeof(int32_t), (void**)&C->dirs));
            MCHK(mhip_host_alloc((C->jobs_cap * 2 + 1) * sizeof(uint64_t), (void**)&C->woffs));
        }
        int64_t words = 0;
        MCHK(mhip_asm_extend_run(ctx, dblk, dq, C->jobs.data(), (int)nj, cap, &words));
        if ((size_t)words > C->ops_cap) {
            if (C->ops) mhip_host_free(C->ops);
            C->ops_cap = (size_t)words + (size_t)words / 4 + 4096;
            MCHK(mhip_host_alloc(C->ops_cap * sizeof(uint32_t), (void**)&C->ops));
        }
        MCHK(mhip_asm_extend_fetch(ctx, (int)nj, C->dirs, C->woffs, C->ops));
        t_extend += now() - tf0;
        n_jobs += nj;
        Chunk* prev = &ck[cur ^ 1];
        if (prev->post.joinable()) prev->post.join();          // the host stages run one after the other (they share the output files)
        if (open_thread.joinable()) open_thread.join();
        C->post = std::thread(host_stage, C);
        cur ^= 1;
        Chunk* nxt = &ck[cur];
        if (nxt->post.joinable()) nxt->post.join();
        nxt->jobs.clear(); nxt->cand.clear(); nxt->qrid.clear();
        nxt->q = C->q; nxt->qkeep = C->qkeep;                  // (the next chunk continues in the same query block)
        t_flush += now() - tf0;
    };

    std::shared_ptr<Reads> ahead;                    // the next query block, read and packed beside the current block's device work
    std::thread ahead_thread;
    for (int bi = start; bi <= last; ++bi) {
        std::shared_ptr<Reads> qs_own;
        const Reads* qs = &blk;
        mhip_volume* dq = dblk;
        if (bi != start) {
            t0 = now();
            ahead_thread.join();
            qs_own = ahead;
            qs = qs_own.get();
            MCHK(mhip_volume_upload(ctx, qs->pac.data(), qs->offs.data(), (int)qs->offs.size(), qs->num_bases, qs->first_no, &dq));
            if (qs->has_n) MCHK(mhip_volume_set_nplane(ctx, dq, qs->npac.data()));
            t_load += now() - t0;
        }
        if (bi < last) {
            ahead = std::make_shared<Reads>();
            Reads* dst = ahead.get();
            const int nb = bi + 1;
            ahead_thread = std::thread([&, dst, nb] { load_block(block_path(nb), first_read[(size_t)nb - 1], dst); });
        }
        const int nq = (int)qs->offs.size();
        int maxlen = 16;
        for (const mhip_offset_t& o : blk.offs) maxlen = std::max(maxlen, o.size);
        for (const mhip_offset_t& o : qs->offs) maxlen = std::max(maxlen, o.size);
        const int cap = ((maxlen * 2 + 64 + 15) / 16) * 16;          // columns of one direction <= bases of both reads on that side
        const size_t dir_words = (size_t)cap / 16;
        const size_t chunk = std::max<size_t>(64, chunk_bytes / (2 * dir_words * sizeof(uint32_t)));
        ck[cur].q = qs; ck[cur].qkeep = qs_own;
        for (int rb = 0; rb < nq; rb += slab) {
            const int re = std::min(nq, rb + slab), nr = re - rb;
            t0 = now();
            MCHK(mhip_asm_seed_reads_ex(ctx, idx, dblk, dq, rb, re, tool.gate, tool.maxc, cands, counts));
            t_seed += now() - t0;
            const double tj0 = now(), tfl0 = t_flush;
            for (int r = 0; r < nr; ++r)
                for (int k = 0; k < counts[(size_t)r]; ++k) {
                    Chunk* C = &ck[cur];
                    const mhip_asm_candidate& c = cands[(size_t)r * 100 + k];
                    mhip_asm_job j;
                    const int x0 = c.loc1 - 1 - c.readstart;        // the seed 13-mer's first base inside the subject read
                    j.xid = c.readno; j.yid = rb + r; j.chain = c.chain;
                    j.lx = x0 + SEED - 1; j.ly = c.loc2 + SEED - 1; j.lnx = c.left1; j.lny = c.left2;       // :736
                    j.rx = x0; j.ry = c.loc2; j.rnx = c.right1; j.rny = c.right2;                            // :793
                    j.pad = 0;
                    C->jobs.push_back(j);
                    C->cand.push_back(c);
                    C->qrid.push_back(rb + r);
                    if (C->jobs.size() >= chunk) flush(C, dq, cap);
                }
            t_jobs += (now() - tj0) - (t_flush - tfl0);
        }
        flush(&ck[cur], dq, cap);
        if (dq != dblk) mhip_volume_free(dq);
    }
    const double td0 = now();
    if (open_thread.joinable()) open_thread.join();
    for (Chunk& c : ck) {
        if (c.post.joinable()) c.post.join();
        if (c.dirs) mhip_host_free(c.dirs);
        if (c.woffs) mhip_host_free(c.woffs);
        if (c.ops) mhip_host_free(c.ops);
    }
    mhip_host_free(cands);
    mhip_host_free(counts);
    t_drain += now() - td0;
    const double tc0 = now();
    for (FILE* f : out)
        if (fclose(f) != 0) DIE("write error");
    mhip_index_free(idx);
    mhip_volume_free(dblk);
    mhip_ctx_destroy(ctx);
    t_close = now() - tc0;
    if (times)
        fprintf(stderr, "[mecat2asmpw] %.2f s: waited for the context %.2f, blocks read + packed + uploaded %.2f, table %.2f, candidates %.2f, jobs %.2f, "
                        "extension %.2f (%zu candidates), strings + lines on %d threads %.2f (beside the device's next chunk; the device stage waited %.2f for "
                        "them), last chunk's lines %.2f, files closed + device memory freed %.2f\n",
                now() - t_begin, t_ctx, t_load, t_index, t_seed, t_jobs, t_extend, n_jobs, threads, t_host, t_flush - t_extend, t_drain, t_close);
    return 0;
}
