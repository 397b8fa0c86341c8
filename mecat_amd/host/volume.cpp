// volume.cpp — FASTA/FASTQ reader and 2-bit volume writer/loader (host only).
//
// What must match the reference byte for byte (so mecat2cns/mecat2canu and a resumed reference run can consume wrk/):
//   * record grammar of FastaReader::read_one_seq (common/fasta_reader.cpp:6-54): '>' or '@' starts a record, a '+' line
//     ends it and swallows exactly one following line, lines starting '#' or '!' are comments, ';' ends a data line,
//     '\n', '\r' and "\r\n" all terminate lines (common/buffer_line_iterator.cpp:22-141), empty lines are skipped;
//   * the plausibility check of data lines (fasta_reader.cpp:91-128) and the invalid-residue error (:57-82);
//   * encode table (common/defs.cpp:3-36) and the UNMASKED OR of codes > 3 into the packed byte (packed_db.h:98-101);
//   * one zero pad base after every read, volume cut when curr + rsize + 1 > MCS (split_database.cpp:240-250);
//   * file layout: int num_reads, int num_bases, int start_read_id, offset_t[num_reads], u8[(num_bases+3)/4] (:135-153).
#include "volume.h"

#include <ctype.h>

#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

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
    // false at end of input; the line is in line()
    bool next() {
        ++line_no_;
        if (unget_) { unget_ = false; return true; }
        line_.clear();
        bool any = false;
        while (true) {
            if (cur_ == end_) { if (!fill()) break; }
            any = true;
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
            return true;
        }
        // end of input: the reference returns the trailing unterminated text as a line and stops on an empty one
        return any && !line_.empty();
    }
    const std::string& line() const { return line_; }

private:
    bool fill() {
        size_t n = fread(&buf_[0], 1, buf_.size(), f_);
        cur_ = 0;
        end_ = n;
        return n > 0;
    }
    FILE* f_ = nullptr;
    std::vector<char> buf_;
    size_t cur_ = 0, end_ = 0;
    std::string line_;
    long line_no_ = 0;
    bool unget_ = false;
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

int split_raw_dataset(const char* reads, const char* wrk_dir) {
    struct timeval t0, t1;
    fprintf(stderr, "[%s] begins.\n", __func__);
    gettimeofday(&t0, NULL);
    HostVolume v;
    int vol = 0, rid = 0;
    const std::string idx_name = index_file_name(wrk_dir);
    FILE* idx_file = fopen(idx_name.c_str(), "w");
    if (!idx_file) DIE("cannot open '%s' for writing", idx_name.c_str());
    // testing knob (additive, environment only): smaller volumes so the multi-volume grid can be exercised on small inputs
    long max_volume_bases = kMaxVolumeBases;
    if (const char* e = getenv("MECAT_HIP_MCS")) { long v2 = atol(e); if (v2 > 0 && v2 < kMaxVolumeBases) max_volume_bases = v2; }
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
