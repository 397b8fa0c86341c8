// pack.hip — residue letters -> the 2-bit volume, on the device (first half: split_raw_dataset's packing step,
// common/split_database.cpp:221-266 with PackedDB::set_char, common/packed_db.h:98-107, and the FastaReader's letter table).
//
// The host has found the records (where each read's first residue sits in the text, how long the read is, how wide its lines are) and
// laid the reads out in the volume (offs[]: one pad base of code 0 behind every read); this kernel makes the packed bytes: one thread per
// 32-bit word of the volume = 16 bases = 16 letters of the text, which are consecutive but for line ends and read boundaries.
// set_char ORs the letter's 4-bit IUPAC value into a 2-bit field WITHOUT masking it (the two upper bits land in the field of the base
// in front, inside the same byte; for the first base of a byte they are lost): reproduced here, byte by byte, so the volume equals
// the reference's bit for bit also for reads with letters other than A, C, G, T.
#include "common.h"

namespace {

// built at compile time and loaded with the code object on every device (no per-process "uploaded" flag: a context on a second GPU of
// the same process gets the same table)
struct EncTable {
    uint8_t t[256];
    constexpr EncTable() : t() {
        const char letters[] = "-acmgrsvtwyhkdbn";
        const uint8_t vals[] = {15, 0, 1, 6, 2, 4, 9, 13, 3, 8, 5, 12, 7, 11, 10, 14};
        for (int i = 0; i < 16; ++i) {
            t[(unsigned char)letters[i]] = vals[i];
            t[(unsigned char)(letters[i] >= 'a' ? letters[i] - 32 : letters[i])] = vals[i];
        }
    }
};
__constant__ EncTable c_enc_tab = EncTable();
#define c_enc c_enc_tab.t      // the letter table (fasta reader / PackedDB: "-ACMGRSVTWYHKDBN" -> 15, 0, 1, 6, 2, 4, 9, 13, 3, 8, 5, 12, 7, 11, 10, 14; others 0)

__global__ __launch_bounds__(256) void pack_letters(const uint8_t* __restrict__ text, const int64_t* __restrict__ seq_start, const int32_t* __restrict__ line_width,
                                                    const mhip_offset_t* __restrict__ offs, int num_reads, int num_bases, uint32_t* __restrict__ pac) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t j0 = w * 16;
    if (j0 >= num_bases) return;
    // the read that holds base j0 (or whose pad base it is): the last read that starts at or before j0
    int lo = 0, hi = num_reads - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int64_t)offs[mid].offset <= j0) lo = mid; else hi = mid - 1;
    }
    int r = lo;
    int64_t roff = offs[r].offset;
    int rsize = offs[r].size, lw = line_width[r];
    int64_t src = seq_start[r];
    uint32_t word = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        uint32_t byte = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t j = j0 + 4 * b + k;
            if (j < num_bases) {
                int64_t i = j - roff;
                if (i > rsize) {            // behind the pad base: the next read (reads are never empty)
                    ++r;
                    roff = offs[r].offset; rsize = offs[r].size; lw = line_width[r]; src = seq_start[r];
                    i = j - roff;
                }
                if (i < rsize) {
                    const int64_t at = src + i + (lw > 0 ? i / lw : 0);      // one line end per full line in front of residue i
                    byte |= (uint32_t)c_enc[text[at]] << (2 * (3 - k));     // unmasked, like set_char: the value's upper bits reach the field in front
                }
            }
        }
        word |= (byte & 0xffu) << (8 * b);
    }
    pac[w] = word;
}

}  // namespace

// text: the residue letters and whatever lies between them (line ends, the next header ...), host memory; seq_start[r] = index in text of
// read r's first residue, line_width[r] = residues per line of read r (all of its lines but the last hold exactly that many, each ended by
// one byte; 0: the read is one line), offs = the volume layout.  Returns the resident volume; pac_out (optional, (num_bases + 3) / 4 bytes)
// receives the packed bytes for the volume file.
int mhip_volume_pack(mhip_ctx* c, const uint8_t* text, int64_t text_bytes, const int64_t* seq_start, const int32_t* line_width, const mhip_offset_t* offs,
                     int num_reads, int num_bases, int start_read_id, mhip_volume** out, uint8_t* pac_out) {
    *out = nullptr;
    if (num_reads < 0 || num_bases < 0 || text_bytes < 0) { mhip_set_error("bad volume sizes"); return -1; }
    {   // the layout has to tile the volume: read r + 1 starts behind read r's pad base, the last pad base ends the volume
        int64_t at = 0;
        for (int r = 0; r < num_reads; ++r) {
            if (offs[r].offset != at) { mhip_set_error("read %d of the layout starts at %d, not at %lld", r, offs[r].offset, (long long)at); return -1; }
            at += (int64_t)offs[r].size + 1;
        }
        if (at != num_bases) { mhip_set_error("the layout holds %lld bases, the volume %d", (long long)at, num_bases); return -1; }
    }
    for (int r = 0; r < num_reads; ++r) {
        const int64_t lines = line_width[r] > 0 ? ((int64_t)offs[r].size - 1) / line_width[r] : 0;
        if (offs[r].size <= 0 || seq_start[r] < 0 || seq_start[r] + offs[r].size + lines > text_bytes || line_width[r] < 0) {
            mhip_set_error("read %d of the text does not lie inside it", r);
            return -1;
        }
    }
    HIPCHK(hipSetDevice(c->device));
    mhip_volume* v = nullptr;
    if (mhip_volume_upload(c, nullptr, offs, num_reads, num_bases, start_read_id, &v)) return -1;      // buffers, read table, lookup table; pac zeroed
    struct Guard { mhip_volume*& v; ~Guard() { if (v) mhip_volume_free(v); } } guard{v};
    if (num_bases == 0 || num_reads == 0) { *out = v; v = nullptr; return 0; }
    uint8_t* d_text;
    int64_t* d_start;
    int32_t* d_lw;
    if (c->scratch("pack_text", (size_t)text_bytes + 64, (void**)&d_text)) return -1;
    if (c->scratch("pack_start", sizeof(int64_t) * (size_t)num_reads, (void**)&d_start)) return -1;
    if (c->scratch("pack_lw", sizeof(int32_t) * (size_t)num_reads, (void**)&d_lw)) return -1;
    HIPCHK(hipMemcpyAsync(d_text, text, (size_t)text_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_start, seq_start, sizeof(int64_t) * (size_t)num_reads, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_lw, line_width, sizeof(int32_t) * (size_t)num_reads, hipMemcpyHostToDevice, c->stream));
    const int64_t nwords = ((int64_t)num_bases + 15) / 16;
    LAUNCH(c, "pack_letters", pack_letters, (unsigned)((nwords + 255) / 256), 256, 0, (const uint8_t*)d_text, (const int64_t*)d_start, (const int32_t*)d_lw,
           (const mhip_offset_t*)v->d_offs, num_reads, num_bases, v->d_pac);
    if (pac_out) HIPCHK(hipMemcpyAsync(pac_out, v->d_pac, ((size_t)num_bases + 3) / 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = v;
    v = nullptr;
    return 0;
}
