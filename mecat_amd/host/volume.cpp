// volume.cpp — FASTA/FASTQ reader and 2-bit volume writer/loader (host only).
//
// What must match the reference byte for byte (so mecat2cns/mecat2canu and a resumed reference run can consume wrk/):
//   * record grammar of FastaReader::read_one_seq (common/fasta_reader.cpp:6-54): '>' or '@' starts a record, a '+' line
//     ends it and swallows exactly one following line, lines starting '#' or '!' are comments, ';' ends a data line,
//     '\n', '\r' and "\r\n" all terminate lines (common/buffer_line_iterator.cpp:22-141); empty lines are skipped, except
//     inside the file's last (partial) 8 MB buffer, where the reference's line reader reports an empty line as "no more lines":
//     there a blank line ends the record (see LineReader::next);
//   * the plausibility check of data lines (fasta_reader.cpp:91-128) and the invalid-residue error (:57-82);
//   * encode table (common/defs.cpp:3-36) and the UNMASKED OR of codes > 3 into the packed byte (packed_db.h:98-101);
//   * one zero pad base after every read, volume cut when curr + rsize + 1 > MCS (split_database.cpp:240-250);
//   * file layout: int num_reads, int num_bases, int start_read_id, offset_t[num_reads], u8[(num_bases+3)/4] (:135-153).
#include "volume.h"

#include <ctype.h>

#include <algorithm>
#include <atomic>
#include <fcntl.h>
#include <functional>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <thread>
#include <unistd.h>

#define DIE(...)                                                  \
    do {                                                          \
        fprintf(stderr, "[%s, %u] ", __func__, __LINE__);         \
        fprintf(stderr, __VA_ARGS__);                             \
        fprintf(stderr, "\n");                                    \
        abort();                                                  \
    } while (0)

namespace {

struct EncodeTable {
    uint8_t t[256];
    EncodeTable() {
        memset(t, 16, sizeof(t));
        const char* letters = "-acmgrsvtwyhkdbn";
        const uint8_t vals[] = {15, 0, 1, 6, 2, 4, 9, 13, 3, 8, 5, 12, 7, 11, 10, 14};
        for (int i = 0; letters[i]; ++i) {
            t[(unsigned char)letters[i]] = vals[i];
            t[(unsigned char)toupper(letters[i])] = vals[i];
        }
    }
};
const EncodeTable kEnc;

// line reader over a whole-file buffer window: '\n', '\r', "\r\n" terminate lines
class LineReader {
public:
    explicit LineReader(const char* path) {
        f_ = fopen(path, "rb");
        if (!f_) DIE("cannot open file '%s' for reading", path);
        buf_.resize(8u << 20);
        fill();
    }
    ~LineReader() { if (f_) fclose(f_); }
    long line_number() const { return line_no_; }
    void unget() { unget_ = true; }
    // false at end of input; the line is in line().  As in the reference (buffer_line_iterator.cpp:22-73, 118-133: the last,
    // partial 8 MB buffer sets done_, and an EMPTY line then reads as "no more lines"), a blank line inside the final buffer ends
    // the record being read — so a blank line between two records is harmless, one inside a record makes the next call fail with
    // "Input doesn't start with a defline", and two in a row end the input.  In earlier buffers blank lines are just skipped.
    bool next() {
        ++line_no_;
        if (unget_) { unget_ = false; return true; }
        line_.clear();
        while (true) {
            if (cur_ == end_) { if (!fill()) break; }
            size_t p = cur_;
            while (p < end_ && buf_[p] != '\n' && buf_[p] != '\r') ++p;
            line_.append(&buf_[cur_], p - cur_);
            if (p == end_) { cur_ = p; continue; }
            const char c = buf_[p];
            cur_ = p + 1;
            if (c == '\r') {
                if (cur_ == end_) fill();
                if (cur_ < end_ && buf_[cur_] == '\n') ++cur_;
            }
            if (cur_ == end_) fill();                    // the reference refills as soon as a line ends at the buffer end
            return !(done_ && line_.empty());
        }
        return !line_.empty();                           // end of input: trailing unterminated text is a line
    }
    const std::string& line() const { return line_; }

private:
    bool fill() {
        size_t n = fread(&buf_[0], 1, buf_.size(), f_);
        cur_ = 0;
        end_ = n;
        if (n < buf_.size()) done_ = true;
        return n > 0;
    }
    FILE* f_ = nullptr;
    std::vector<char> buf_;
    size_t cur_ = 0, end_ = 0;
    std::string line_;
    long line_no_ = 0;
    bool unget_ = false, done_ = false;
};

inline bool is_nucl(unsigned char c) { return kEnc.t[c] < 16; }
inline bool is_alpha_ascii(unsigned char c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }

// fasta_reader.cpp:91-128
void check_data_line(const std::string& line, long line_no) {
    long good = 0, bad = 0, len = (long)line.size();
    for (long pos = 0; pos < len; ++pos) {
        const unsigned char c = (unsigned char)line[pos];
        if (is_alpha_ascii(c) || c == '*') ++good;
        else if (c == '-') ++good;
        else if (isspace(c) || (c >= '0' && c <= '9')) {}
        else if (c == ';') break;
        else ++bad;
    }
    if (bad >= good / 3 && (len > 3 || good == 0 || bad > good))
        DIE("FastaReader: Near line %ld, there's a line that doesn't look like plausible data, but it's not marked as defline or commnet.", line_no);
}

// returns the read length, -1 at end of input; `seq` receives the raw residue characters
long read_one_seq(LineReader& r, std::string& seq, bool* have_header) {
    seq.clear();
    size_t header_size = 0;
    bool need_defline = true;
    while (r.next()) {
        const std::string& l = r.line();
        if (l.empty()) continue;
        const int c = (unsigned char)l[0];
        if (c == '>' || c == '@') {
            if (need_defline) {
                header_size = l.size() - 1;
                if (header_size == 0) DIE("A sequence is given an empty header around line %ld.", r.line_number());
                need_defline = false;
                continue;
            }
            r.unget();
            break;
        } else if (c == '+') {
            if (!r.next()) DIE("FastaReader: quality score line is missing at around line %ld", r.line_number());
            break;
        } else if (c == '#' || c == '!') {
            continue;
        } else if (need_defline) {
            DIE("FastaReader: Input doesn't start with a defline or comment around line %ld", r.line_number());
        }
        check_data_line(l, r.line_number());
        for (size_t pos = 0; pos < l.size(); ++pos) {
            const unsigned char ch = (unsigned char)l[pos];
            if (ch == ';') break;
            if (is_nucl(ch) || ch == '-') seq.push_back((char)ch);
            else if (!isspace(ch))
                DIE("FastaReader: There are invalid residue(s) around position %d of line %ld.", (int)(pos + 1), r.line_number());
        }
    }
    if (seq.empty() && header_size > 0) DIE("FastaReader: Near line %ld, sequence data is missing.", r.line_number());
    *have_header = header_size > 0;
    if (header_size == 0 && seq.empty()) return -1;
    return (long)seq.size();
}

// the volume written last stays in memory for its first load_volume (one volume is the common case: no re-read of the file
// that was just written)
std::string g_kept_path;
HostVolume g_kept;
bool g_async_dump = false;
std::thread g_pending;           // writes the kept volume's file
std::thread g_unmapper;          // unmaps the input (volume_release_input)
const void* g_map = NULL;        // the input mapping of an asynchronous split, until it is released
size_t g_map_size = 0;

void dump_volume(const std::string& path, const HostVolume& v) {
    FILE* out = fopen(path.c_str(), "wb");
    if (!out) DIE("cannot open '%s' for writing", path.c_str());
    bool ok = fwrite(&v.num_reads, sizeof(int), 1, out) == 1 && fwrite(&v.num_bases, sizeof(int), 1, out) == 1 &&
              fwrite(&v.start_read_id, sizeof(int), 1, out) == 1;
    if (v.num_reads) ok = ok && fwrite(v.offs.data(), sizeof(mhip_offset_t), (size_t)v.num_reads, out) == (size_t)v.num_reads;
    const size_t nb = ((size_t)v.num_bases + 3) / 4;
    if (nb) ok = ok && fwrite(v.pac.data(), 1, nb, out) == nb;
    if (fclose(out) != 0 || !ok) DIE("write error!");
}

template <typename F>
void run_threads(int nt, F f) {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(f, t);
    f(0);
    for (auto& x : th) x.join();
}

struct PlainRec { size_t data; int len; int lw; };      // lw: residues per line (0: one line; -1: lines of different lengths)

// MECAT_HIP_SPLIT=gpu: the packing step on the device (mhip_volume_pack) instead of on the host threads; the hook hands out the context
// (the driver creates it on a second thread while the input is scanned)
std::function<mhip_ctx*()> g_device_ctx;

// MECAT_TRACE: seconds per stage of the threaded split on stderr
struct SplitClock {
    bool on; double t0;
    static double now() { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec + 1e-6 * t.tv_usec; }
    SplitClock() : on(getenv("MECAT_TRACE") != NULL), t0(now()) {}
    void mark(const char* what) { if (!on) return; const double t = now(); fprintf(stderr, "[trace] split: %-14s %.3f s\n", what, t - t0); t0 = t; }
};     // first byte after the header line, number of residues

// Parallel reader for plain FASTA (see volume.h).  Returns false, touching nothing, if the file is anything else.
bool split_plain_fasta(const char* reads, const char* wrk_dir, long max_volume_bases, int nt, int* out_vols, long long* out_reads,
                       long long* out_nucls) {
    const int fd = open(reads, O_RDONLY);
    if (fd < 0) return false;                       // the sequential reader reports the error
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 1 || !S_ISREG(st.st_mode)) { close(fd); return false; }
    const size_t size = (size_t)st.st_size;
    const char* txt = (const char*)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (txt == MAP_FAILED) return false;
    SplitClock clk;
    bool ok = txt[0] == '>';
    nt = std::max(1, std::min(nt, 64));
    // chunk t = records whose '>' lies in [cut[t], cut[t+1])
    std::vector<size_t> cut((size_t)nt + 1, size);
    cut[0] = 0;
    for (int t = 1; t < nt && ok; ++t) {
        size_t p = size / (size_t)nt * (size_t)t;
        const char* q = p < size ? (const char*)memmem(txt + p, size - p, "\n>", 2) : NULL;
        cut[(size_t)t] = q ? (size_t)(q - txt) + 1 : size;
    }
    std::vector<std::vector<PlainRec>> recs((size_t)nt);
    std::atomic<int> plain(ok ? 1 : 0);
    if (ok) run_threads(nt, [&](int t) {
        std::vector<PlainRec>& out = recs[(size_t)t];
        size_t p = cut[(size_t)t];
        const size_t end = cut[(size_t)t + 1];
        while (p < end && plain.load(std::memory_order_relaxed)) {
            // header line: '>' + at least one character
            const char* nl = (const char*)memchr(txt + p, '\n', size - p);
            const size_t hend = nl ? (size_t)(nl - txt) : size;
            if (txt[p] != '>' || hend - p < 2 || memchr(txt + p, '\r', hend - p)) { plain = 0; return; }
            p = hend < size ? hend + 1 : size;
            PlainRec r;
            r.data = p;
            r.lw = 0;
            long len = 0, prev_line = -1;
            while (p < size) {                       // data lines up to the next line starting with '>'
                const unsigned char c0 = (unsigned char)txt[p];
                if (c0 == '>') break;
                if (c0 == '\n') { plain = 0; return; } // an empty line: the sequential reader knows the reference's rules for those
                if (c0 == '@' || c0 == '+' || c0 == '#' || c0 == '!') { plain = 0; return; }
                size_t q = p;
                while (q < size && txt[q] != '\n') {
                    if (kEnc.t[(unsigned char)txt[q]] >= 16) { plain = 0; return; }     // blanks, digits, ';', '\r', bad residues
                    ++q;
                }
                // line widths (for the device packer): every line but the last as long as the first
                if (prev_line >= 0) {
                    if (r.lw == 0) r.lw = (int)std::min<long>(prev_line, 0x7fffffffL);
                    else if (r.lw > 0 && prev_line != r.lw) r.lw = -1;
                }
                prev_line = (long)(q - p);
                len += (long)(q - p);
                p = q < size ? q + 1 : size;
            }
            if (r.lw > 0 && prev_line > r.lw) r.lw = -1;      // (the last line may be shorter, not longer)
            if (len == 0 || len > 0x7fffffffL) { plain = 0; return; }
            r.len = (int)len;
            out.push_back(r);
        }
    });
    if (!plain.load()) { munmap((void*)txt, size); return false; }
    clk.mark("scan");

    // volume layout: the reference's loop (split_database.cpp:240-250)
    struct Vol { size_t first, count; long bases; };
    std::vector<PlainRec> all;
    for (auto& v : recs) { all.insert(all.end(), v.begin(), v.end()); std::vector<PlainRec>().swap(v); }
    std::vector<long> at(all.size());
    std::vector<Vol> vols;
    long curr = 0;
    size_t first = 0;
    long long nucls = 0;
    for (size_t i = 0; i < all.size(); ++i) {
        const long rsize = all[i].len;
        nucls += rsize;
        if (curr + rsize + 1 > max_volume_bases) { vols.push_back(Vol{first, i - first, curr}); first = i; curr = 0; }
        at[i] = curr;
        curr += rsize + 1;
    }
    if (curr > 0) vols.push_back(Vol{first, all.size() - first, curr});

    const std::string idx_name = index_file_name(wrk_dir);
    FILE* idx_file = fopen(idx_name.c_str(), "w");
    if (!idx_file) DIE("cannot open '%s' for writing", idx_name.c_str());
    int rid = 0;
    // Two volume buffers in turn: the file of volume k (535 MB at full size: 0.09 s, as long as its packing) is written by a second thread
    // while volume k + 1 is packed — 1.5 of the 3.0 s a 19-volume split (config 5) took.
    HostVolume vbuf[2];
    std::thread dumper;
    for (size_t k = 0; k < vols.size(); ++k) {
        const Vol& vo = vols[k];
        HostVolume& v = vbuf[k & 1];      // (written out two volumes ago: that writer was joined before the one now in flight was started)
        v.num_bases = (int)vo.bases;
        v.num_reads = (int)vo.count;
        v.start_read_id = rid;
        rid += v.num_reads;
        v.offs.resize(vo.count);
        const size_t pac_bytes = ((size_t)vo.bases + 3) / 4;
        v.pac.clear();
        v.pac.resize(pac_bytes);                  // uninitialised
        run_threads(nt, [&](int t) {              // zero (first touch) in parallel: the packers OR into the bytes at read boundaries
            const size_t lo = pac_bytes * (size_t)t / (size_t)nt, hi = pac_bytes * (size_t)(t + 1) / (size_t)nt;
            if (hi > lo) memset(v.pac.data() + lo, 0, hi - lo);
        });
        clk.mark("layout+zero");
        // thread t packs a contiguous range of reads holding ~1/nt of the volume's bases
        std::vector<size_t> rcut((size_t)nt + 1, vo.count);
        rcut[0] = 0;
        for (int t = 1; t < nt; ++t) {
            const long want = vo.bases / nt * t;
            rcut[(size_t)t] = (size_t)(std::lower_bound(at.begin() + (long)vo.first, at.begin() + (long)(vo.first + vo.count), want) -
                                       (at.begin() + (long)vo.first));
        }
        bool on_device = false;
        if (g_device_ctx && getenv("MECAT_HIP_SPLIT") && !strcmp(getenv("MECAT_HIP_SPLIT"), "gpu")) {
            // the packing step on the device: the file's bytes go over as they are, the packed bytes come back for the volume file
            bool regular = true;
            for (size_t i = 0; i < vo.count && regular; ++i) regular = all[vo.first + i].lw >= 0;
            mhip_ctx* dc = regular ? g_device_ctx() : NULL;
            if (dc) {
                const size_t t0 = all[vo.first].data;
                const PlainRec& lastr = all[vo.first + vo.count - 1];
                const size_t t1 = std::min(size, lastr.data + (size_t)lastr.len + (lastr.lw > 0 ? (size_t)(lastr.len - 1) / (size_t)lastr.lw : 0));
                std::vector<int64_t> sstart(vo.count);
                std::vector<int32_t> lws(vo.count);
                for (size_t i = 0; i < vo.count; ++i) {
                    const PlainRec& r = all[vo.first + i];
                    v.offs[i].offset = (int)at[vo.first + i];
                    v.offs[i].size = r.len;
                    sstart[i] = (int64_t)(r.data - t0);
                    lws[i] = r.lw;
                }
                mhip_volume* dv = NULL;
                if (mhip_volume_pack(dc, (const uint8_t*)txt + t0, (int64_t)(t1 - t0), sstart.data(), lws.data(), v.offs.data(), v.num_reads, v.num_bases,
                                     v.start_read_id, &dv, v.pac.data()) != 0)
                    DIE("mhip_volume_pack failed: %s", mhip_last_error());
                mhip_volume_free(dv);
                on_device = true;
            }
        }
        if (!on_device) run_threads(nt, [&](int t) {
            uint8_t* pac = v.pac.data();
            for (size_t i = rcut[(size_t)t]; i < rcut[(size_t)t + 1]; ++i) {
                const PlainRec& r = all[vo.first + i];
                long pos = at[vo.first + i];
                v.offs[i].offset = (int)pos;
                v.offs[i].size = r.len;
                const long last = pos + r.len - 1;
                const size_t b_first = (size_t)(pos >> 2), b_last = (size_t)(last >> 2);
                size_t bi = b_first;
                uint8_t acc = 0;
                auto put = [&]() {
                    // bytes at a read's ends may be shared with the neighbouring read (another thread's, possibly)
                    if (bi == b_first || bi == b_last) __atomic_fetch_or(&pac[bi], acc, __ATOMIC_RELAXED);
                    else pac[bi] = acc;
                };
                const char* q = txt + r.data;
                long left = r.len;
                while (left > 0) {
                    const unsigned char ch = (unsigned char)*q++;
                    if (ch == '\n') continue;
                    const size_t b = (size_t)(pos >> 2);
                    if (b != bi) { put(); bi = b; acc = 0; }
                    acc |= (uint8_t)(kEnc.t[ch] << ((~pos & 3) << 1));       // PackedDB::set_char, unmasked
                    ++pos;
                    --left;
                }
                put();
            }
        });
        clk.mark(on_device ? "pack (device)" : "pack");
        const std::string name = volume_file_name(wrk_dir, (int)k);
        fprintf(idx_file, "%s\n", name.c_str());
        if (k + 1 == vols.size()) {
            g_kept_path = name;
            g_kept = std::move(v);
            if (g_async_dump) {
                // the buffers of g_kept stay where they are when load_volume() moves them into the caller's volume
                const HostVolume* kv = &g_kept;
                const int nr = kv->num_reads, nb = kv->num_bases, sid = kv->start_read_id;
                const mhip_offset_t* offs = kv->offs.data();
                const uint8_t* pac = kv->pac.data();
                g_pending = std::thread([=]() {
                    FILE* out = fopen(name.c_str(), "wb");
                    if (!out) DIE("cannot open '%s' for writing", name.c_str());
                    bool ok = fwrite(&nr, sizeof(int), 1, out) == 1 && fwrite(&nb, sizeof(int), 1, out) == 1 && fwrite(&sid, sizeof(int), 1, out) == 1;
                    if (nr) ok = ok && fwrite(offs, sizeof(mhip_offset_t), (size_t)nr, out) == (size_t)nr;
                    const size_t bytes = ((size_t)nb + 3) / 4;
                    if (bytes) ok = ok && fwrite(pac, 1, bytes, out) == bytes;
                    if (fclose(out) != 0 || !ok) DIE("write error!");
                });
                g_map = txt;
                g_map_size = size;
            } else {
                dump_volume(name, g_kept);
            }
        } else {
            if (dumper.joinable()) dumper.join();            // one writer at a time
            const HostVolume* pv = &v;
            dumper = std::thread([name, pv]() { dump_volume(name, *pv); });
        }
        clk.mark("dump");
    }
    if (dumper.joinable()) dumper.join();
    fclose(idx_file);
    if (!g_map) munmap((void*)txt, size);
    clk.mark("unmap");
    *out_vols = (int)vols.size();
    *out_reads = (long long)all.size();
    *out_nucls = nucls;
    return true;
}

}  // namespace

std::string volume_file_name(const char* wrk_dir, int vol) {
    std::string s(wrk_dir);
    if (s.empty() || s[s.size() - 1] != '/') s += '/';
    s += "vol";
    s += std::to_string(vol);
    return s;
}

std::string index_file_name(const char* wrk_dir) {
    std::string s(wrk_dir);
    if (s.empty() || s[s.size() - 1] != '/') s += '/';
    s += "fileindex.txt";
    return s;
}

int split_raw_dataset(const char* reads, const char* wrk_dir, int num_threads) {
    struct timeval t0, t1;
    fprintf(stderr, "[%s] begins.\n", __func__);
    gettimeofday(&t0, NULL);
    // testing knob (additive, environment only): smaller volumes so the multi-volume grid can be exercised on small inputs
    long max_volume_bases = kMaxVolumeBases;
    if (const char* e = getenv("MECAT_HIP_MCS")) { long v2 = atol(e); if (v2 > 0 && v2 < kMaxVolumeBases) max_volume_bases = v2; }
    {
        const char* mode = getenv("MECAT_HIP_SPLIT");
        int pv = 0;
        long long pr = 0, pn = 0;
        if (!(mode && strcmp(mode, "seq") == 0) && split_plain_fasta(reads, wrk_dir, max_volume_bases, num_threads, &pv, &pr, &pn)) {
            gettimeofday(&t1, NULL);
            fprintf(stderr, "[%s, %u] split '%s' (%lld reads, %lld nucls) into %d volumes.\n", __func__, __LINE__, reads, pr, pn, pv);
            fprintf(stderr, "[%s] takes %.2f secs.\n", __func__, t1.tv_sec - t0.tv_sec + 1e-6 * (t1.tv_usec - t0.tv_usec));
            return pv;
        }
    }
    HostVolume v;
    int vol = 0, rid = 0;
    const std::string idx_name = index_file_name(wrk_dir);
    FILE* idx_file = fopen(idx_name.c_str(), "w");
    if (!idx_file) DIE("cannot open '%s' for writing", idx_name.c_str());
    LineReader lr(reads);
    std::string seq;
    long long num_reads = 0, num_nucls = 0;
    long curr = 0;
    auto flush = [&]() {
        v.num_bases = (int)curr;
        v.num_reads = (int)v.offs.size();
        v.start_read_id = rid;
        rid += v.num_reads;
        v.pac.resize(((size_t)curr + 3) / 4);
        const std::string name = volume_file_name(wrk_dir, vol++);
        fprintf(idx_file, "%s\n", name.c_str());
        dump_volume(name, v);
        v.offs.clear();
        v.pac.clear();
        curr = 0;
    };
    while (true) {
        bool have_header;
        const long rsize = read_one_seq(lr, seq, &have_header);
        if (rsize == -1) break;
        ++num_reads;
        num_nucls += rsize;
        if (curr + rsize + 1 > max_volume_bases) flush();
        mhip_offset_t o;
        o.offset = (int)curr;
        o.size = (int)rsize;
        v.offs.push_back(o);
        const size_t need = ((size_t)(curr + rsize + 1) + 3) / 4;
        if (v.pac.size() < need) v.pac.resize(std::max(need, v.pac.size() * 2), 0);
        for (long i = 0; i < rsize; ++i, ++curr) {
            const uint8_t c = kEnc.t[(unsigned char)seq[(size_t)i]];
            v.pac[(size_t)(curr >> 2)] |= (uint8_t)(c << ((~curr & 3) << 1));     // PackedDB::set_char, unmasked
        }
        ++curr;   // pad base
    }
    if (curr > 0) flush();
    fclose(idx_file);
    gettimeofday(&t1, NULL);
    fprintf(stderr, "[%s, %u] split '%s' (%lld reads, %lld nucls) into %d volumes.\n", __func__, __LINE__, reads, num_reads, num_nucls, vol);
    fprintf(stderr, "[%s] takes %.2f secs.\n", __func__, t1.tv_sec - t0.tv_sec + 1e-6 * (t1.tv_usec - t0.tv_usec));
    return vol;
}

bool volume_dump_in_flight() { return g_pending.joinable(); }
void volume_set_device_packer(std::function<mhip_ctx*()> get_ctx) { g_device_ctx = std::move(get_ctx); }

void* volume_big_alloc(size_t bytes) {
    const size_t huge = (size_t)2 << 20;
    void* p = NULL;
    if (bytes >= 8 * huge) {
        if (posix_memalign(&p, huge, (bytes + huge - 1) & ~(huge - 1)) != 0) throw std::bad_alloc();
        (void)madvise(p, (bytes + huge - 1) & ~(huge - 1), MADV_HUGEPAGE);
        return p;
    }
    p = malloc(std::max<size_t>(bytes, 1));
    if (!p) throw std::bad_alloc();
    return p;
}
void volume_big_free(void* p, size_t) { free(p); }

void volume_set_async_dump(bool on) {
    static bool registered = false;
    g_async_dump = on;
    if (on && !registered) { registered = true; atexit(volume_wait_pending); }
}

void volume_wait_pending() {
    // (an error exit from the writer thread itself runs the atexit handler on that thread: nothing to wait for then)
    if (g_pending.joinable() && g_pending.get_id() != std::this_thread::get_id()) g_pending.join();
    if (g_unmapper.joinable() && g_unmapper.get_id() != std::this_thread::get_id()) g_unmapper.join();
}

void volume_release_input() {
    if (!g_map || g_unmapper.joinable()) return;
    const void* m = g_map;
    const size_t n = g_map_size;
    g_map = NULL;
    g_unmapper = std::thread([m, n]() { munmap((void*)m, n); });
}

std::vector<std::string> load_volume_names(const std::string& idx_file) {
    std::vector<std::string> names;
    FILE* f = fopen(idx_file.c_str(), "r");
    if (!f) DIE("cannot open '%s'", idx_file.c_str());
    char* line = NULL;
    size_t cap = 0;
    ssize_t ls;
    while ((ls = getline(&line, &cap, f)) != -1) {
        if (ls > 0 && line[ls - 1] == '\n') --ls;
        if (ls > 0 && line[ls - 1] == '\r') --ls;
        if (ls > 0) names.emplace_back(line, (size_t)ls);
    }
    free(line);
    fclose(f);
    return names;
}

void load_volume(const std::string& path, HostVolume* v) {
    if (!g_kept_path.empty() && path == g_kept_path) {      // still in memory from split_raw_dataset
        *v = std::move(g_kept);
        g_kept_path.clear();
        return;
    }
    volume_wait_pending();
    FILE* in = fopen(path.c_str(), "rb");
    if (!in) { fprintf(stderr, "[%s, %u] failed to open file '%s'.\n", __func__, __LINE__, path.c_str()); exit(1); }
    bool ok = fread(&v->num_reads, sizeof(int), 1, in) == 1 && fread(&v->num_bases, sizeof(int), 1, in) == 1 &&
              fread(&v->start_read_id, sizeof(int), 1, in) == 1;
    if (!ok) DIE("read error!");
    v->offs.resize((size_t)v->num_reads);
    const size_t nb = ((size_t)v->num_bases + 3) / 4;
    v->pac.resize(nb);
    if (v->num_reads) ok = ok && fread(v->offs.data(), sizeof(mhip_offset_t), (size_t)v->num_reads, in) == (size_t)v->num_reads;
    if (nb) ok = ok && fread(v->pac.data(), 1, nb, in) == nb;
    fclose(in);
    if (!ok) DIE("read error!");
}
