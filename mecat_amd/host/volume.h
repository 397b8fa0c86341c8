// volume.h — host side of the working-directory file protocol: FASTA/FASTQ -> 2-bit volumes
// (wrk/vol<k>, wrk/fileindex.txt) byte-identical to the reference's split_raw_dataset
// (common/split_database.cpp:221-266), and the loader for those files (:155-181).
#pragma once

#include <stdint.h>

#include <functional>
#include <string>
#include <memory>
#include <utility>
#include <vector>

#include "mecat_hip.h"

static const long kMaxVolumeBases = 2140000000L;   // MCS, common/split_database.h:6

// std::allocator whose value-less construct() default-initialises: vector::resize(n) does not zero-fill (the packers write every
// byte themselves; zeroing 400 MB on one thread costs 70 ms), resize(n, 0) / assign(n, 0) still do
// Large blocks (a volume's packed bytes: hundreds of MB) come 2 MB-aligned with transparent huge pages asked for: first touch costs
// microseconds per 2 MB instead of per 4 KB, and the block can be page-locked for the upload in milliseconds (mhip_host_register).
void* volume_big_alloc(size_t bytes);
void volume_big_free(void* p, size_t bytes);
template <typename T>
struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U>&) {}
    T* allocate(size_t n) { return (T*)volume_big_alloc(n * sizeof(T)); }
    void deallocate(T* p, size_t n) { volume_big_free(p, n * sizeof(T)); }
    template <typename U, typename... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
        else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};

struct HostVolume {
    int num_reads = 0;
    int num_bases = 0;         // incl. one pad base per read
    int start_read_id = 0;
    std::vector<mhip_offset_t> offs;
    std::vector<uint8_t, NoInitAlloc<uint8_t>> pac;  // (num_bases + 3) / 4 bytes; resize(n) leaves new bytes uninitialised
};

// Splits `reads` into volumes inside `wrk_dir`; returns the number of volumes.  Aborts with the reference's messages on
// malformed input (FastaReader, common/fasta_reader.cpp:6-130).  Plain FASTA ('>' records, '\n' line ends, residue
// letters only) is scanned and packed by `num_threads` threads over the mapped file; anything else (FASTQ, comments,
// '\r', blanks inside lines, invalid residues ...) goes through the sequential reader that reproduces the reference's
// grammar and error messages.  Both produce the same bytes.  MECAT_HIP_SPLIT=seq forces the sequential reader.
int split_raw_dataset(const char* reads, const char* wrk_dir, int num_threads = 1);

std::string volume_file_name(const char* wrk_dir, int vol);      // generate_vol_file_name, split_database.cpp:183-192
std::string index_file_name(const char* wrk_dir);                // generate_idx_file_name, split_database.cpp:194-200
std::vector<std::string> load_volume_names(const std::string& idx_file);   // split_database.cpp:373-392
// One-process runs: the file of the volume that stays in memory is written, and the input unmapped, on a second thread while the
// caller goes on; volume_wait_pending() returns when that is done (called before a volume file is read back, before the volume's
// memory is released and at exit).  Multi-process runs keep the write synchronous: other ranks read the file.
// MECAT_HIP_SPLIT=gpu: plain FASTA whose records have lines of one width is packed on the device (mhip_volume_pack)
// with the context the hook returns (NULL: on the host after all).  The volume files are the same bytes either way.
void volume_set_device_packer(std::function<mhip_ctx*()> get_ctx);
void volume_set_async_dump(bool on);
void volume_wait_pending();
bool volume_dump_in_flight();      // the second thread may still be writing a volume file out of the kept volume's buffers
// Unmapping a multi-GB input holds the process's mmap lock for tens of milliseconds, which every hipMalloc needs: the mapping of an
// asynchronous split is let go (on its own thread) only when the caller says the device allocations are made.  Without the call it
// goes with the process.
void volume_release_input();
void load_volume(const std::string& path, HostVolume* v);        // split_database.cpp:155-181 (exit(1) if missing)
