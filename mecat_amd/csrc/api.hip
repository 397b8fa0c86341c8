// api.hip — context, volume and bookkeeping entry points of the C ABI (include/mecat_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <thread>

#include "common.h"
#include <mutex>

static thread_local char g_err[1024] = "";

void mhip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int mhip_abi_version(void) { return MHIP_ABI_VERSION; }
const char* mhip_last_error(void) { return g_err; }

int mhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void mhip_params_default(mhip_params* p, int tech) {
    // pw_options.cpp:8-13,30-50 ; pw_impl.cpp:843-851
    p->maxc = 100;
    p->tech = tech;
    p->ddfs_cutoff = 0.25;
    if (tech == 0) { p->min_align_size = 2000; p->min_kmer_match = 4; p->min_kmer_dist = 1800; }
    else           { p->min_align_size = 500;  p->min_kmer_match = 2; p->min_kmer_dist = 400; }
}

int mhip_ctx_create(int device, void* stream, mhip_ctx** out) {
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        mhip_set_error("no HIP device available (%s); libmecat_hip has no CPU fallback", hipGetErrorString(e));
        return -1;
    }
    if (device < 0 || device >= n) { mhip_set_error("device %d out of range (have %d)", device, n); return -1; }
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        mhip_set_error("device %d is %s; libmecat_hip is built for gfx950 only", device, prop.gcnArchName);
        return -1;
    }
    mhip_ctx* c = new mhip_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; mhip_set_error("hipStreamCreate failed"); return -1; }
        c->own_stream = true;
    }
    // 8 public counters (mecat_hip.h) + 8 debug slots
    if (hipMalloc((void**)&c->d_counters, 64 * sizeof(int64_t)) != hipSuccess) { delete c; mhip_set_error("hipMalloc counters failed"); return -1; }
    (void)hipMemsetAsync(c->d_counters, 0, 64 * sizeof(int64_t), c->stream);
    *out = c;
    return 0;
}

void mhip_ctx_destroy(mhip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto& kv : c->bufs) if (kv.second.p) (void)hipFree(kv.second.p);
    if (c->d_counters) (void)hipFree(c->d_counters);
    dev_recycler_release(c->device);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int mhip_host_alloc(size_t bytes, void** out) {
    *out = nullptr;
    if (bytes == 0) bytes = 1;
    HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return 0;
}

void mhip_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

// page-locks a caller's buffer for the copies of the calls in between (a 400 MB volume out of pageable memory goes up at a tenth of the
// link's rate); pages that are touched already (huge pages best) are locked in milliseconds.  Returns 0 when the buffer is locked.
int mhip_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return -1;
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return 0;
}
void mhip_host_unregister(void* p) {
    if (p && hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
}

int mhip_ctx_reserve_index(mhip_ctx* c, int64_t bases) {
    if (bases <= 0) return 0;
    // ix_ent1 (every k-mer start of the volume) and ix_ent2 (the ping buffer of one group of coarse bins: a quarter of the
    // k-mer space; 40 % leaves room for uneven groups — a build that needs more simply reallocates it)
    const size_t nbytes[2] = {sizeof(uint64_t) * ((size_t)bases + 64), sizeof(uint64_t) * ((size_t)(bases * 2 / 5) + 64)};
    std::lock_guard<std::mutex> lk(c->bufs_mu);
    const char* names[2] = {"ix_ent1", "ix_ent2"};
    void* p[2] = {nullptr, nullptr};
    hipError_t err[2] = {hipSuccess, hipSuccess};
    bool need[2];
    for (int i = 0; i < 2; ++i) need[i] = c->bufs[names[i]].cap < nbytes[i];
    const int device = c->device;
    auto grab = [&](int i) {                                           // the two mappings proceed side by side
        if (hipSetDevice(device) != hipSuccess) { err[i] = hipErrorInvalidDevice; return; }
        err[i] = hipMalloc(&p[i], nbytes[i] + 4096);
    };
    std::thread t1;
    if (need[1]) t1 = std::thread(grab, 1);
    if (need[0]) grab(0);
    if (t1.joinable()) t1.join();
    int rc = 0;
    for (int i = 0; i < 2; ++i) {
        if (!need[i]) continue;
        if (err[i] != hipSuccess) { mhip_set_error("hipMalloc of %zu bytes for the index scratch failed: %s", nbytes[i], hipGetErrorString(err[i])); rc = -1; continue; }
        DevBuf& b = c->bufs[names[i]];
        if (b.p) (void)hipFree(b.p);
        b.p = p[i];
        b.cap = nbytes[i] + 4096;
    }
    return rc;
}

int mhip_ctx_buffer(mhip_ctx* c, const char* name, size_t bytes, void** d_ptr) {
    HIPCHK(hipSetDevice(c->device));
    if (!name || !*name) { mhip_set_error("buffer without a name"); return -1; }
    const std::string key = std::string("user_") + name;       // kept apart from the library's own scratch names
    return c->scratch(key.c_str(), bytes ? bytes : 1, d_ptr);
}

int mhip_download(mhip_ctx* c, void* host_dst, const void* d_src, size_t bytes) {
    HIPCHK(hipSetDevice(c->device));
    if (bytes == 0) return 0;
    HIPCHK(hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int mhip_ctx_mem_info(mhip_ctx* c, size_t* free_bytes, size_t* total_bytes) {
    HIPCHK(hipSetDevice(c->device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

int mhip_ctx_sync(mhip_ctx* c) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int mhip_ctx_set_profiling(mhip_ctx* c, int on) {
    if (c->drain_events()) return -1;
    c->profiling = on != 0;
    return 0;
}

int mhip_ctx_reset_stats(mhip_ctx* c) {
    if (c->drain_events()) return -1;
    c->stats.clear();
    HIPCHK(hipMemsetAsync(c->d_counters, 0, 64 * sizeof(int64_t), c->stream));
    return 0;
}

// debug slots 8..15 (8: dw blocks re-run with spilled rows)
int mhip_debug_counter(mhip_ctx* c, int slot, int64_t* out) {
    if (slot < 0 || slot >= 64) return -1;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out, c->d_counters + slot, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int mhip_ctx_kernel_stats(mhip_ctx* c, const char* name, int64_t* launches, double* total_ms) {
    if (c->drain_events()) return -1;
    auto it = c->stats.find(name);
    if (it == c->stats.end()) { *launches = 0; *total_ms = 0.0; return 0; }
    *launches = it->second.launches;
    *total_ms = it->second.ms;
    return 0;
}

int mhip_ctx_kernel_names(mhip_ctx* c, char* buf, int buflen) {
    if (c->drain_events()) return -1;
    std::string s;
    for (auto& kv : c->stats) { s += kv.first; s += "\n"; }
    snprintf(buf, (size_t)buflen, "%s", s.c_str());
    return 0;
}

int mhip_ctx_counters(mhip_ctx* c, int64_t out[8]) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out, c->d_counters, 8 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int mhip_volume_upload(mhip_ctx* c, const uint8_t* pac, const mhip_offset_t* offs, int num_reads, int num_bases,
                       int start_read_id, mhip_volume** out) {
    *out = nullptr;
    if (num_reads < 0 || num_bases < 0) { mhip_set_error("bad volume sizes"); return -1; }
    HIPCHK(hipSetDevice(c->device));
    mhip_volume* v = new mhip_volume();
    struct Guard { mhip_volume*& v; ~Guard() { if (v) mhip_volume_free(v); } } guard{v};      // every early return frees the buffers
    v->device = c->device;
    v->num_reads = num_reads;
    v->num_bases = num_bases;
    v->start_read_id = start_read_id;
    size_t nb = ((size_t)num_bases + 3) / 4;
    v->pac_bytes = ((nb + 63) / 64) * 64 + 128;     // zero padding so whole-word window loads never leave the buffer
    if (hipMalloc((void**)&v->d_pac, v->pac_bytes) != hipSuccess ||
        hipMalloc((void**)&v->d_offs, sizeof(mhip_offset_t) * (size_t)(num_reads + 1)) != hipSuccess) {
        mhip_set_error("hipMalloc failed for a %zu-byte volume", v->pac_bytes);
        return -1;
    }
    HIPCHK(hipMemsetAsync(v->d_pac, 0, v->pac_bytes, c->stream));
    if (nb && pac) HIPCHK(hipMemcpyAsync(v->d_pac, pac, nb, hipMemcpyHostToDevice, c->stream));      // (pac == NULL: mhip_volume_pack fills it)
    v->h_offs.assign(offs, offs + num_reads);
    if (num_reads) HIPCHK(hipMemcpyAsync(v->d_offs, offs, sizeof(mhip_offset_t) * (size_t)num_reads, hipMemcpyHostToDevice, c->stream));
    // position -> read without a binary search over the offsets (get_read_id_from_offset_list, common/split_database.cpp:15-35,
    // is 17 dependent loads at volume size): table entry b = the read that holds base 1024 b
    std::vector<uint32_t> blk((size_t)(num_bases >> 10) + 2, 0u);
    {
        uint32_t last = 0;
        size_t filled = 0;
        for (int r = 0; r < num_reads; ++r) {
            const size_t b0 = ((size_t)offs[r].offset + 1023) >> 10, b1 = ((size_t)offs[r].offset + (size_t)std::max(offs[r].size, 1) - 1) >> 10;
            for (; filled < b0 && filled < blk.size(); ++filled) blk[filled] = last;      // bases between reads: the read before
            for (size_t b = b0; b <= b1 && b < blk.size(); ++b) { blk[b] = (uint32_t)r; filled = b + 1; }
            last = (uint32_t)r;
        }
        for (; filled < blk.size(); ++filled) blk[filled] = last;
    }
    if (hipMalloc((void**)&v->d_blk2read, sizeof(uint32_t) * blk.size()) != hipSuccess) {
        mhip_set_error("hipMalloc failed for the read lookup table");
        return -1;
    }
    HIPCHK(hipMemcpyAsync(v->d_blk2read, blk.data(), sizeof(uint32_t) * blk.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = v;
    v = nullptr;                                     // handed over: the guard lets go
    return 0;
}

void mhip_volume_free(mhip_volume* v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->d_pac) (void)hipFree(v->d_pac);
    if (v->d_npac) (void)hipFree(v->d_npac);
    if (v->d_offs) (void)hipFree(v->d_offs);
    if (v->d_blk2read) (void)hipFree(v->d_blk2read);
    delete v;
}

// The bases that are not A, C, G or T (mecat2asmpw / mecat2trimpw only: the tools restart a k-mer at such a base and compare it as a
// character, mecat2canu/src/mecat2asmpw/mecat2asmpw.c:445, 486, 316-335): a second plane in the volume's own 2-bit layout, 3 at every such
// base (whose code in the volume itself is 0), 0 elsewhere.  nplane = (num_bases + 3) / 4 host bytes; NULL removes the plane.
int mhip_volume_set_nplane(mhip_ctx* c, mhip_volume* v, const uint8_t* nplane) {
    HIPCHK(hipSetDevice(c->device));
    if (v->d_npac) { HIPCHK(hipFree(v->d_npac)); v->d_npac = nullptr; }
    if (!nplane) return 0;
    if (hipMalloc((void**)&v->d_npac, v->pac_bytes) != hipSuccess) { mhip_set_error("hipMalloc failed for a %zu-byte N plane", v->pac_bytes); return -1; }
    HIPCHK(hipMemsetAsync(v->d_npac, 0, v->pac_bytes, c->stream));
    const size_t nb = ((size_t)v->num_bases + 3) / 4;
    if (nb) HIPCHK(hipMemcpyAsync(v->d_npac, nplane, nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int mhip_volume_num_reads(const mhip_volume* v) { return v->num_reads; }
int mhip_volume_num_bases(const mhip_volume* v) { return v->num_bases; }

}  // extern "C"

namespace {
struct Parked { int device; void* p; size_t cap; };
std::mutex g_park_mu;
std::vector<Parked> g_parked;
}  // namespace

int dev_alloc_recycled(int device, size_t bytes, void** p, size_t* cap) {
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        for (size_t i = 0; i < g_parked.size(); ++i) {
            Parked& k = g_parked[i];
            if (k.device == device && k.cap >= bytes && k.cap / 2 <= bytes) {
                *p = k.p;
                *cap = k.cap;
                g_parked.erase(g_parked.begin() + i);
                return 0;
            }
        }
    }
    size_t want = bytes + bytes / 16 + 4096;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {
        dev_recycler_release(device);            // give parked blocks back and retry at the exact size
        want = bytes;
        e = hipMalloc(p, want);
    }
    if (e != hipSuccess) {
        *p = nullptr;
        mhip_set_error("hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e));
        return -1;
    }
    *cap = want;
    return 0;
}

void dev_free_recycled(int device, void* p, size_t cap) {
    if (!p) return;
    void* drop = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        size_t n = 0, smallest = (size_t)-1;
        for (size_t i = 0; i < g_parked.size(); ++i)
            if (g_parked[i].device == device) { ++n; if (smallest == (size_t)-1 || g_parked[i].cap < g_parked[smallest].cap) smallest = i; }
        if (n >= 4) {
            if (g_parked[smallest].cap < cap) { drop = g_parked[smallest].p; g_parked[smallest] = Parked{device, p, cap}; }
            else drop = p;
        } else g_parked.push_back(Parked{device, p, cap});
    }
    if (drop) (void)hipFree(drop);
}

void dev_recycler_release(int device) {
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        for (size_t i = 0; i < g_parked.size();)
            if (g_parked[i].device == device) { drop.push_back(g_parked[i].p); g_parked.erase(g_parked.begin() + i); }
            else ++i;
    }
    for (void* q : drop) (void)hipFree(q);
}

int mhip_ctx::scratch(const char* name, size_t bytes, void** out) {
    std::lock_guard<std::mutex> lk(bufs_mu);
    DevBuf& b = bufs[name];
    if (b.cap < bytes) {
        if (b.p) {
            HIPCHK(hipStreamSynchronize(stream));
            HIPCHK(hipFree(b.p));
            b.p = nullptr;
            b.cap = 0;
        }
        size_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) {
            b.p = nullptr;
            mhip_set_error("hipMalloc of %zu bytes for scratch '%s' failed: %s", want, name, hipGetErrorString(e));
            return -1;
        }
        b.cap = want;
        b.gen += 1;
    }
    *out = b.p;
    return 0;
}

uint64_t mhip_ctx::scratch_generation(const char* name) {
    std::lock_guard<std::mutex> lk(bufs_mu);
    auto it = bufs.find(name);
    return it == bufs.end() ? 0 : it->second.gen;
}

hipEvent_t mhip_ctx::get_event() {
    if (!ev_pool.empty()) {
        hipEvent_t e = ev_pool.back();
        ev_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

int mhip_ctx::drain_events() {
    if (pending.empty()) return 0;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize(stream));
    for (auto& p : pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            KStat& s = stats[p.name];
            s.launches += 1;
            s.ms += ms;
        }
        ev_pool.push_back(p.a);
        ev_pool.push_back(p.b);
    }
    pending.clear();
    return 0;
}
