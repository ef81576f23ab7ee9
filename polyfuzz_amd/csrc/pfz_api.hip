// Context, error reporting, timers, CSR / top-n containers and the shared
// device scan of libpolyfuzz_hip.so.
#include "pfz_internal.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

namespace pfz {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    if (e == hipErrorOutOfMemory) return PFZ_ERR_NOMEM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return PFZ_ERR_NO_DEVICE;
    return PFZ_ERR_HIP;
}

// The slot / event pool of a context is touched by whoever frees a matrix or an index -- a Python finalizer on another thread while
// a ctypes call (GIL released) acquires a slot on the same context: one lock for it, as pool_free has for the device blocks.
static std::mutex g_lazy_mu;

int lazy_acquire(pfz_ctx *ctx, LazyI32 *z)
{
    std::lock_guard<std::mutex> lk(g_lazy_mu);
    if (ctx->lazy_slots.empty()) {
        int32_t *chunk = nullptr;
        PFZ_HIP(hipHostMalloc((void **)&chunk, 64 * sizeof(int32_t), hipHostMallocDefault));      // (pinned, mapped: kernels may write it)
        ctx->lazy_chunks.push_back(chunk);
        for (int i = 0; i < 64; ++i) ctx->lazy_slots.push_back(chunk + i);
    }
    if (ctx->lazy_events.empty()) {
        hipEvent_t ev;
        PFZ_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->lazy_events.push_back(ev);
    }
    z->slot = ctx->lazy_slots.back();
    ctx->lazy_slots.pop_back();
    z->ev = ctx->lazy_events.back();
    ctx->lazy_events.pop_back();
    z->pending = false;
    return PFZ_OK;
}

int lazy_mark(pfz_ctx *ctx, LazyI32 *z)
{
    PFZ_HIP(hipEventRecord(z->ev, ctx->stream));
    z->pending = true;
    return PFZ_OK;
}

int lazy_begin(pfz_ctx *ctx, LazyI32 *z, const int32_t *dev)
{
    PFZ_TRY(lazy_acquire(ctx, z));
    PFZ_HIP(hipMemcpyAsync(z->slot, dev, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    return lazy_mark(ctx, z);
}

int lazy_get(pfz_ctx *ctx, LazyI32 *z, int32_t *out)
{
    if (z->pending) {
        PFZ_HIP(hipEventSynchronize(z->ev));
        z->value = *z->slot;
        lazy_release(ctx, z);
    }
    *out = z->value;
    return PFZ_OK;
}

void lazy_release(pfz_ctx *ctx, LazyI32 *z)
{
    std::lock_guard<std::mutex> lk(g_lazy_mu);
    if (z->slot) ctx->lazy_slots.push_back(z->slot);
    if (z->ev) ctx->lazy_events.push_back(z->ev);
    z->slot = nullptr;
    z->ev = nullptr;
    z->pending = false;
}

int64_t csr_nnz(const pfz_csr *m)
{
    if (m->nnz_lazy.pending) {
        int32_t v = 0;
        if (lazy_get(m->ctx, &m->nnz_lazy, &v) != PFZ_OK) return -1;
        m->nnz = v;
    }
    return m->nnz;
}

int ensure_scratch(pfz_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return PFZ_OK;
    if (ctx->scratch) {
        PFZ_HIP(hipStreamSynchronize(ctx->stream));
        PFZ_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    size_t want = bytes + bytes / 4 + 4096;
    PFZ_HIP(hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
    return PFZ_OK;
}

// ---- staged host <-> device copies ----------------------------------------------
constexpr size_t kStageBytes = (size_t)64 << 20;    // pinned once per context
constexpr size_t kStageDirect = (size_t)256 << 20;  // larger one-off buffers (dense matrices) go to the runtime as they are

static int stage_take(pfz_ctx *ctx, size_t bytes, char **out)
{
    if (!ctx->stage) {
        PFZ_HIP(hipHostMalloc((void **)&ctx->stage, kStageBytes, hipHostMallocDefault));
        ctx->stage_bytes = kStageBytes;
        ctx->stage_off = 0;
    }
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (ctx->stage_off + need > ctx->stage_bytes) {   // wrap: everything staged so far must have left the buffer
        PFZ_HIP(hipStreamSynchronize(ctx->stream));
        ctx->stage_off = 0;
    }
    *out = ctx->stage + ctx->stage_off;
    ctx->stage_off += need;
    return PFZ_OK;
}

int ensure_side_stream(pfz_ctx *ctx)
{
    if (!ctx->stream2) {
        // HIGH priority: what runs here are short pieces beside a long kernel of the context's stream -- the download of a finished
        // row range while K3's pass 1 fills every CU (at equal priority the copy was served when pass 1 had ended: measured), the row
        // top-n of one score panel beside the GEMM of the next (K5)
        int pr_lo = 0, pr_hi = 0;
        PFZ_HIP(hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi));
        PFZ_HIP(hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, pr_hi));
        for (hipEvent_t &ev : ctx->side_events) PFZ_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    return PFZ_OK;
}

int event_flag_next(pfz_ctx *ctx, int32_t slot, int32_t **flag, int32_t *value)
{
    if (!ctx->evt_flag) {
        PFZ_HIP(hipHostMalloc((void **)&ctx->evt_flag, kEventSlots * sizeof(int32_t), hipHostMallocDefault));
        memset(ctx->evt_flag, 0, kEventSlots * sizeof(int32_t));
    }
    ctx->evt_serial = ctx->evt_serial == 0x7fffffff ? 1 : ctx->evt_serial + 1;
    ctx->evt_want[slot] = ctx->evt_serial;
    *flag = ctx->evt_flag + slot;
    *value = ctx->evt_serial;
    return PFZ_OK;
}

int copy_h2d(pfz_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0) return PFZ_OK;
    if (ctx->stage && (const char *)src_host >= ctx->stage && (const char *)src_host + bytes <= ctx->stage + ctx->stage_bytes) {
        // the source already lies in the pinned staging buffer (pfz_stage_reserve: the string packer wrote it there): no copy of a
        // copy -- the DMA reads it where it is; the region is not handed out again before the stream has been waited for (stage_take)
        PFZ_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
        return PFZ_OK;
    }
    if (bytes > kStageDirect) {
        PFZ_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
        PFZ_HIP(hipStreamSynchronize(ctx->stream));
        return PFZ_OK;
    }
    for (size_t done = 0; done < bytes;) {
        const size_t part = bytes - done < kStageBytes ? bytes - done : kStageBytes;
        char *st = nullptr;
        PFZ_TRY(stage_take(ctx, part, &st));
        memcpy(st, (const char *)src_host + done, part);
        PFZ_HIP(hipMemcpyAsync((char *)dst_dev + done, st, part, hipMemcpyHostToDevice, ctx->stream));
        done += part;
    }
    return PFZ_OK;
}

int copy_d2h(pfz_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes == 0) return PFZ_OK;
    if (bytes > kStageDirect) {
        PFZ_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        PFZ_HIP(hipStreamSynchronize(ctx->stream));
        return PFZ_OK;
    }
    for (size_t done = 0; done < bytes;) {
        const size_t part = bytes - done < kStageBytes ? bytes - done : kStageBytes;
        char *st = nullptr;
        PFZ_TRY(stage_take(ctx, part, &st));
        PFZ_HIP(hipMemcpyAsync(st, (const char *)src_dev + done, part, hipMemcpyDeviceToHost, ctx->stream));
        PFZ_HIP(hipStreamSynchronize(ctx->stream));
        memcpy((char *)dst_host + done, st, part);
        ctx->stage_off = 0;                           // the stream is idle: the whole buffer is free again
        done += part;
    }
    return PFZ_OK;
}

// ---- caching allocator -------------------------------------------------------
static std::mutex g_pool_mu;
static std::unordered_map<void *, std::pair<pfz_ctx *, size_t>> g_pool_owner;  // live block -> (ctx, class size)

static size_t size_class(size_t bytes)
{
    if (bytes < 512) return 512;
    size_t p2 = 512;
    while (p2 * 2 <= bytes) p2 *= 2;       // p2 <= bytes < 2*p2
    for (int j = 0; j <= 4; ++j) {
        const size_t c = p2 + (p2 / 4) * j;  // p2, 1.25, 1.5, 1.75, 2 x p2
        if (c >= bytes) return c;
    }
    return p2 * 2;
}

int pool_alloc_raw(pfz_ctx *ctx, void **p, size_t bytes)
{
    const size_t cls = size_class(bytes);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = ctx->pool_free_lists.find(cls);
    if (it != ctx->pool_free_lists.end() && !it->second.empty()) {
        *p = it->second.back();
        it->second.pop_back();
        ctx->pool_cached_bytes -= cls;
    } else {
        hipError_t e = hipMalloc(p, cls);
        if (e == hipErrorOutOfMemory && ctx->pool_cached_bytes > 0) {   // give the cache back and retry once
            (void)hipStreamSynchronize(ctx->stream);
            for (auto &kv : ctx->pool_free_lists) {
                for (void *q : kv.second) (void)hipFree(q);
                kv.second.clear();
            }
            ctx->pool_cached_bytes = 0;
            e = hipMalloc(p, cls);
        }
        if (e != hipSuccess) {
            *p = nullptr;
            return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
        }
    }
    g_pool_owner[*p] = {ctx, cls};
    ctx->pool_live_bytes += cls;
    return PFZ_OK;
}

void pool_free(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_owner.find(p);
    if (it == g_pool_owner.end()) {   // not ours (should not happen): hand it to the runtime
        (void)hipFree(p);
        return;
    }
    pfz_ctx *ctx = it->second.first;
    const size_t cls = it->second.second;
    g_pool_owner.erase(it);
    ctx->pool_free_lists[cls].push_back(p);
    ctx->pool_cached_bytes += cls;
    ctx->pool_live_bytes -= cls;
}

int pool_release(pfz_ctx *ctx)
{
    PFZ_HIP(hipSetDevice(ctx->device));
    PFZ_HIP(hipStreamSynchronize(ctx->stream));
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto &kv : ctx->pool_free_lists) {
        for (void *q : kv.second) (void)hipFree(q);
        kv.second.clear();
    }
    ctx->pool_cached_bytes = 0;
    return PFZ_OK;
}

ProfScope::ProfScope(pfz_ctx *c, const char *n, hipStream_t on) : ctx(c), name(n), st(on ? on : c->stream)
{
    if (!ctx->prof) return;
    // level 2: only the dominant kernels (the ones a roofline is computed for) -- two event records per launch of
    // every small kernel cost the step ~2 %
    if (ctx->prof_level == 2 && strncmp(n, "k3_cossim", 9) != 0 && strncmp(n, "k4_indel", 8) != 0 && strncmp(n, "k5_gemm", 7) != 0)
        return;
    auto take = [&]() -> hipEvent_t {
        if (!ctx->event_pool.empty()) {
            hipEvent_t ev = ctx->event_pool.back();
            ctx->event_pool.pop_back();
            return ev;
        }
        hipEvent_t ev = nullptr;
        (void)hipEventCreate(&ev);
        return ev;
    };
    b = take();
    e = take();
    (void)hipEventRecord(b, st);
}

ProfScope::~ProfScope()
{
    static const bool debug_sync = getenv("PFZ_DEBUG_SYNC") != nullptr;
    if (debug_sync) {  // developer aid: localise an asynchronous device fault to a kernel
        fprintf(stderr, "[pfz] %s ...", name);
        fflush(stderr);
        hipError_t err = hipStreamSynchronize(st);
        fprintf(stderr, " %s\n", hipGetErrorString(err));
    }
    if (!ctx->prof || !b || !e) return;
    (void)hipEventRecord(e, st);
    ProfEntry &pe = ctx->prof_entries[name];
    pe.begin.push_back(b);
    pe.end.push_back(e);
}

static void prof_fold(pfz_ctx *ctx, ProfEntry &pe)
{
    for (size_t i = 0; i < pe.begin.size(); ++i) {
        float ms = 0.f;
        (void)hipEventSynchronize(pe.end[i]);
        if (hipEventElapsedTime(&ms, pe.begin[i], pe.end[i]) == hipSuccess) {
            pe.total_ms += ms;
            pe.launches += 1;
        }
        ctx->event_pool.push_back(pe.begin[i]);
        ctx->event_pool.push_back(pe.end[i]);
    }
    pe.begin.clear();
    pe.end.clear();
}

// ---- multi-block exclusive scan (int32) -----------------------------------
// Three launches: per-tile sums, scan of the tile sums (one workgroup), tile
// re-scan with the carried-in base.  Tile = 256 threads x 16 items.
constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanTile = kScanThreads * kScanItems;

__device__ inline int32_t wave_incl_scan(int32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan across the 256 threads of a workgroup; returns the prefix of
// this thread, *total = workgroup sum.
__device__ inline int32_t block_excl_scan(int32_t v, int32_t *lds4, int32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    int32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        int32_t s = lds4[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_tile_sums(const int32_t *__restrict__ in, int64_t n,
                                                                 int32_t *__restrict__ tile_sums)
{
    __shared__ int32_t lds4[kScanThreads / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        int64_t p = base + i;
        if (p < n) s += in[p];
    }
    int32_t tot;
    (void)block_excl_scan(s, lds4, &tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_sums(int32_t *__restrict__ tile_sums, int64_t n_tiles,
                                                            int32_t *__restrict__ grand_total)
{
    __shared__ int32_t lds4[kScanThreads / 64];
    int32_t carry = 0;
    for (int64_t t0 = 0; t0 < n_tiles; t0 += kScanThreads) {
        int64_t p = t0 + threadIdx.x;
        int32_t v = (p < n_tiles) ? tile_sums[p] : 0;
        int32_t tot;
        int32_t ex = block_excl_scan(v, lds4, &tot);
        if (p < n_tiles) tile_sums[p] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_apply(int32_t *__restrict__ data, int64_t n,
                                                             const int32_t *__restrict__ tile_sums)
{
    __shared__ int32_t lds4[kScanThreads / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int32_t v[kScanItems];
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        int64_t p = base + i;
        v[i] = (p < n) ? data[p] : 0;
        s += v[i];
    }
    int32_t tot;
    int32_t ex = block_excl_scan(s, lds4, &tot) + tile_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        int64_t p = base + i;
        if (p < n) data[p] = ex;
        ex += v[i];
    }
}

// The same scan in ONE launch (decoupled look-back): a tile takes a ticket (so every tile before it is running or done),
// scans its 4096 items, publishes its sum, and its first thread walks back over the tiles before it -- adding sums until it
// meets a tile that already knows its whole prefix -- then publishes its own.  A state word is (epoch * 4 + kind) << 32 | value,
// kind 1 = the tile's sum, 2 = the sum through the tile; every call has a new epoch, so the words are never cleared, and the
// last tile puts the ticket counter back to 0.  Agent-scope loads / stores: the tiles run on different XCDs.
// (Three launches -- ~15 us on a 10 000-string list whose whole fit + transform + index + match is 0.4 ms -- were the price of
// the version above, which stays for PFZ_SCAN3=1.)
__global__ __launch_bounds__(kScanThreads) void k_scan_lookback(int32_t *__restrict__ data, int64_t n, uint64_t *__restrict__ state,
                                                                uint32_t *__restrict__ ticket, uint32_t epoch4, int32_t n_tiles,
                                                                int32_t *__restrict__ host_total)
{
    __shared__ int32_t lds4[kScanThreads / 64];
    __shared__ int32_t s_tile, s_prefix;
    if (threadIdx.x == 0) s_tile = (int32_t)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = s_tile;
    const int64_t base = (int64_t)tile * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int32_t v[kScanItems];
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const int64_t p = base + i;
        v[i] = (p < n) ? data[p] : 0;
        s += v[i];
    }
    int32_t tot;
    int32_t ex = block_excl_scan(s, lds4, &tot);
    if (threadIdx.x == 0) {
        int32_t prefix = 0;
        if (tile > 0) {
            __hip_atomic_store(&state[tile], ((uint64_t)(epoch4 + 1u) << 32) | (uint32_t)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int p = tile - 1;; --p) {
                uint64_t w;
                for (;;) {
                    w = __hip_atomic_load(&state[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t f = (uint32_t)(w >> 32);
                    if (f == epoch4 + 1u || f == epoch4 + 2u) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                prefix += (int32_t)(uint32_t)w;
                if ((uint32_t)(w >> 32) == epoch4 + 2u) break;
            }
        }
        __hip_atomic_store(&state[tile], ((uint64_t)(epoch4 + 2u) << 32) | (uint32_t)(prefix + tot), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        s_prefix = prefix;
        if (tile == n_tiles - 1) {
            data[n] = prefix + tot;      // the grand total
            // ... and straight into a pinned word of the host's when it wants it (LazyI32): no copy kernel behind the scan
            if (host_total) __hip_atomic_store(host_total, prefix + tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            *ticket = 0u;                // (every ticket has been taken: this tile holds the last one)
        }
    }
    __syncthreads();
    ex += s_prefix;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const int64_t p = base + i;
        if (p < n) data[p] = ex;
        ex += v[i];
    }
}

int exclusive_scan_i32(pfz_ctx *ctx, int32_t *data, int64_t n, LazyI32 *total)
{
    if (n <= 0) {
        PFZ_HIP(hipMemsetAsync(data, 0, sizeof(int32_t), ctx->stream));
        return total ? lazy_begin(ctx, total, data) : PFZ_OK;
    }
    const int64_t n_tiles = (n + kScanTile - 1) / kScanTile;
    static const bool three = getenv("PFZ_SCAN3") != nullptr;
    if (!three && n_tiles < (1 << 24)) {
        if ((size_t)n_tiles > ctx->scan_tiles) {
            // (grown, never shrunk; the old words may still be read by a scan in flight on this stream: stream order frees it)
            size_t cap = ctx->scan_tiles ? ctx->scan_tiles : 1024;
            while (cap < (size_t)n_tiles) cap *= 2;
            uint64_t *fresh = nullptr;
            PFZ_TRY(pool_alloc(ctx, &fresh, (cap + 1) * sizeof(uint64_t)));
            PFZ_HIP(hipMemsetAsync(fresh, 0, (cap + 1) * sizeof(uint64_t), ctx->stream));
            if (ctx->scan_state) pool_free(ctx->scan_state);
            ctx->scan_state = fresh;
            ctx->scan_tiles = cap;
            ctx->scan_epoch = 0;
        }
        if (ctx->scan_epoch >= (1u << 29)) {          // (the epoch field is 30 bits: start over on clean words)
            PFZ_HIP(hipMemsetAsync(ctx->scan_state, 0, (ctx->scan_tiles + 1) * sizeof(uint64_t), ctx->stream));
            ctx->scan_epoch = 0;
        }
        const uint32_t epoch4 = ++ctx->scan_epoch * 4u;
        if (total) PFZ_TRY(lazy_acquire(ctx, total));
        hipLaunchKernelGGL(k_scan_lookback, dim3((unsigned)n_tiles), dim3(kScanThreads), 0, ctx->stream, data, n, ctx->scan_state,
                           (uint32_t *)(ctx->scan_state + ctx->scan_tiles), epoch4, (int32_t)n_tiles, total ? total->slot : nullptr);
        PFZ_HIP(hipGetLastError());
        return total ? lazy_mark(ctx, total) : PFZ_OK;
    }
    PFZ_TRY(ensure_scratch(ctx, (size_t)n_tiles * sizeof(int32_t)));
    int32_t *tile_sums = (int32_t *)ctx->scratch;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)n_tiles), dim3(kScanThreads), 0, ctx->stream, data, n, tile_sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanThreads), 0, ctx->stream, tile_sums, n_tiles, data + n);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)n_tiles), dim3(kScanThreads), 0, ctx->stream, data, n, tile_sums);
    PFZ_HIP(hipGetLastError());
    return total ? lazy_begin(ctx, total, data + n) : PFZ_OK;
}

}  // namespace pfz

using namespace pfz;

extern "C" {

int pfz_version(void) { return PFZ_VERSION; }

const char *pfz_last_error(void) { return pfz::g_err; }

int pfz_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pfz_ctx_create(int device, pfz_ctx **out)
{
    PFZ_REQUIRE(out != nullptr, "pfz_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("pfz_ctx_create: no HIP device visible (hipGetDeviceCount -> %d, %s); "
                  "libpolyfuzz_hip has no CPU fallback", (int)e, hipGetErrorString(e));
        return PFZ_ERR_NO_DEVICE;
    }
    PFZ_REQUIRE(device >= 0 && device < n, "pfz_ctx_create: device %d out of range [0,%d)", device, n);
    PFZ_HIP(hipSetDevice(device));
    pfz_ctx *ctx = new pfz_ctx();
    ctx->device = device;
    PFZ_HIP(hipGetDeviceProperties(&ctx->prop, device));
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("pfz_ctx_create: device %d is %s; this library ships gfx950 code objects only",
                  device, ctx->prop.gcnArchName);
        delete ctx;
        return PFZ_ERR_NO_DEVICE;
    }
    PFZ_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    for (int i = 0; i < kEventSlots; ++i) PFZ_HIP(hipEventCreate(&ctx->events[i]));
    *out = ctx;
    return PFZ_OK;
}

void pfz_ctx_destroy(pfz_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->prof_entries) prof_fold(ctx, kv.second);
    for (hipEvent_t ev : ctx->event_pool) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->lazy_events) (void)hipEventDestroy(ev);
    for (int32_t *c : ctx->lazy_chunks) (void)hipHostFree(c);
    for (int i = 0; i < kEventSlots; ++i)
        if (ctx->events[i]) (void)hipEventDestroy(ctx->events[i]);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    if (ctx->stage2) (void)hipHostFree(ctx->stage2);
    if (ctx->evt_flag) (void)hipHostFree(ctx->evt_flag);
    if (ctx->mirror) (void)hipHostFree(ctx->mirror);
    for (char *p : ctx->rows_stage)
        if (p) (void)hipHostFree(p);
    for (hipEvent_t ev : ctx->rows_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->side_events)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->stream2) {
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamDestroy(ctx->stream2);
    }
    if (ctx->stream3) {
        (void)hipStreamSynchronize(ctx->stream3);
        (void)hipStreamDestroy(ctx->stream3);
    }
    if (ctx->ev3) (void)hipEventDestroy(ctx->ev3);
    for (int q = 0; q < 3; ++q) {
        if (ctx->stream3x[q]) {
            (void)hipStreamSynchronize(ctx->stream3x[q]);
            (void)hipStreamDestroy(ctx->stream3x[q]);
        }
        if (ctx->ev3x[q]) (void)hipEventDestroy(ctx->ev3x[q]);
    }
    (void)pool_release(ctx);
    {   // blocks still owned by live handles of this context: free them, the handles become inert
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (auto it = g_pool_owner.begin(); it != g_pool_owner.end();) {
            if (it->second.first == ctx) {
                (void)hipFree(it->first);
                it = g_pool_owner.erase(it);
            } else {
                ++it;
            }
        }
    }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int pfz_ctx_sync(pfz_ctx *ctx)
{
    PFZ_REQUIRE(ctx, "pfz_ctx_sync: ctx is NULL");
    PFZ_HIP(hipStreamSynchronize(ctx->stream));
    return PFZ_OK;
}

int pfz_ctx_info(pfz_ctx *ctx, char *name256, int32_t *n_cu, int64_t *hbm_bytes)
{
    PFZ_REQUIRE(ctx, "pfz_ctx_info: ctx is NULL");
    if (name256) snprintf(name256, 256, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
    return PFZ_OK;
}

int pfz_event_record(pfz_ctx *ctx, int32_t slot)
{
    PFZ_REQUIRE(ctx && slot >= 0 && slot < kEventSlots, "pfz_event_record: bad slot %d", slot);
    PFZ_HIP(hipEventRecord(ctx->events[slot], ctx->stream));
    ctx->evt_want[slot] = 0;      // (an event of the context's own stream: no word in pinned memory)
    return PFZ_OK;
}

// spin on the pinned word of an event slot (event_flag_next); the side stream running dry without the word is an error, not a hang
static int event_flag_spin(pfz_ctx *ctx, int32_t slot, const char *who)
{
    const int32_t want = ctx->evt_want[slot];
    const volatile int32_t *flag = ctx->evt_flag + slot;
    for (uint64_t spins = 0; *flag != want; ++spins) {
        __builtin_ia32_pause();
        bool dry = (spins & 0xffff) == 0xffff && ctx->stream3 && hipStreamQuery(ctx->stream3) == hipSuccess;
        for (int q = 0; q < 3 && dry; ++q) dry = !ctx->stream3x[q] || hipStreamQuery(ctx->stream3x[q]) == hipSuccess;
        if (dry && *flag != want) {
            PFZ_HIP(hipStreamSynchronize(ctx->stream3));
            for (int q = 0; q < 3; ++q)
                if (ctx->stream3x[q]) PFZ_HIP(hipStreamSynchronize(ctx->stream3x[q]));
            if (*flag != want) {
                set_error("%s: event slot %d was never announced", who, slot);
                return PFZ_ERR_HIP;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return PFZ_OK;
}

int pfz_stage_reserve(pfz_ctx *ctx, int64_t bytes, void **host)
{
    PFZ_REQUIRE(ctx && host && bytes >= 0, "pfz_stage_reserve: bad arguments");
    *host = nullptr;
    if ((size_t)bytes > kStageBytes / 2) return PFZ_OK;      // (too big for the ring: the caller takes the ordinary path)
    PFZ_HIP(hipSetDevice(ctx->device));
    char *p = nullptr;
    PFZ_TRY(stage_take(ctx, (size_t)bytes, &p));
    *host = p;
    return PFZ_OK;
}

int pfz_event_wait(pfz_ctx *ctx, int32_t slot)
{
    PFZ_REQUIRE(ctx && slot >= 0 && slot < kEventSlots, "pfz_event_wait: bad slot %d", slot);
    if (ctx->evt_want[slot] != 0 && ctx->evt_flag) return event_flag_spin(ctx, slot, "pfz_event_wait");
    PFZ_HIP(hipSetDevice(ctx->device));
    PFZ_HIP(hipEventSynchronize(ctx->events[slot]));
    return PFZ_OK;
}

int pfz_event_elapsed_ms(pfz_ctx *ctx, int32_t a, int32_t b, float *ms)
{
    PFZ_REQUIRE(ctx && ms && a >= 0 && a < kEventSlots && b >= 0 && b < kEventSlots, "pfz_event_elapsed_ms: bad args");
    PFZ_HIP(hipEventSynchronize(ctx->events[a]));
    PFZ_HIP(hipEventSynchronize(ctx->events[b]));
    PFZ_HIP(hipEventElapsedTime(ms, ctx->events[a], ctx->events[b]));
    return PFZ_OK;
}

int pfz_prof_enable(pfz_ctx *ctx, int32_t on)
{
    PFZ_REQUIRE(ctx, "pfz_prof_enable: ctx is NULL");
    ctx->prof = on != 0;
    ctx->prof_level = on;
    return PFZ_OK;
}

int pfz_prof_reset(pfz_ctx *ctx)
{
    PFZ_REQUIRE(ctx, "pfz_prof_reset: ctx is NULL");
    PFZ_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &kv : ctx->prof_entries) {
        prof_fold(ctx, kv.second);
        kv.second.total_ms = 0.0;
        kv.second.launches = 0;
    }
    return PFZ_OK;
}

int pfz_prof_get(pfz_ctx *ctx, const char *name, double *total_ms, int64_t *launches)
{
    PFZ_REQUIRE(ctx && name, "pfz_prof_get: bad args");
    auto it = ctx->prof_entries.find(name);
    if (it == ctx->prof_entries.end()) {
        if (total_ms) *total_ms = 0.0;
        if (launches) *launches = 0;
        return PFZ_OK;
    }
    prof_fold(ctx, it->second);
    if (total_ms) *total_ms = it->second.total_ms;
    if (launches) *launches = it->second.launches;
    return PFZ_OK;
}

// ---- CSR -------------------------------------------------------------------

int pfz_csr_upload(pfz_ctx *ctx, int64_t n_rows, int64_t n_cols, const int64_t *indptr,
                   const int32_t *indices, const float *data, pfz_csr **out)
{
    PFZ_REQUIRE(ctx && out && indptr, "pfz_csr_upload: NULL argument");
    PFZ_REQUIRE(n_rows >= 0 && n_cols >= 0, "pfz_csr_upload: negative shape");
    const int64_t nnz = indptr[n_rows];
    PFZ_REQUIRE(indptr[0] == 0 && nnz >= 0, "pfz_csr_upload: indptr must start at 0");
    if (nnz >= ((int64_t)1 << 31) || n_rows >= ((int64_t)1 << 31) - 1) {
        set_error("pfz_csr_upload: nnz=%lld / n_rows=%lld exceed the int32 device layout", (long long)nnz, (long long)n_rows);
        return PFZ_ERR_UNSUPPORTED;
    }
    PFZ_REQUIRE(nnz == 0 || (indices && data), "pfz_csr_upload: NULL indices/data");
    PFZ_HIP(hipSetDevice(ctx->device));
    Owner<pfz_csr, pfz_csr_free> m(new pfz_csr());
    m->ctx = ctx;
    m->n_rows = n_rows;
    m->n_cols = n_cols;
    m->nnz = nnz;
    m->nnz_cap = nnz;
    std::vector<int32_t> ip32((size_t)n_rows + 1);
    for (int64_t i = 0; i <= n_rows; ++i) {
        if (i > 0 && indptr[i] < indptr[i - 1]) {
            set_error("pfz_csr_upload: indptr not monotone at row %lld", (long long)i);
            return PFZ_ERR_INVALID;
        }
        ip32[(size_t)i] = (int32_t)indptr[i];
    }
    {   // largest row norm (the caller's matrix need not be normalised, reference _utils.py:74-82) and the
        // column range: K3 indexes its offset table with these ids, scipy does not guarantee them
        double mx = 0.0;
        for (int64_t i = 0; i < n_rows; ++i) {
            double ss = 0.0;
            for (int64_t p = indptr[i]; p < indptr[i + 1]; ++p) {
                if (indices[p] < 0 || indices[p] >= n_cols) {
                    set_error("pfz_csr_upload: row %lld has column index %d outside [0, %lld)", (long long)i,
                              indices[p], (long long)n_cols);
                    return PFZ_ERR_INVALID;
                }
                ss += (double)data[p] * (double)data[p];
            }
            if (!(ss == ss) || ss > 1e60) {
                set_error("pfz_csr_upload: row %lld has a non-finite or huge norm", (long long)i);
                return PFZ_ERR_INVALID;
            }
            if (ss > mx) mx = ss;
        }
        m->max_norm = (float)(sqrt(mx) * 1.000001);
    }
    PFZ_TRY(pool_alloc(ctx, &m->indptr, (size_t)(n_rows + 1) * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &m->indices, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &m->data, (size_t)(nnz > 0 ? nnz : 1) * sizeof(float)));
    PFZ_TRY(copy_h2d(ctx, m->indptr, ip32.data(), (size_t)(n_rows + 1) * sizeof(int32_t)));
    if (nnz > 0) {
        PFZ_TRY(copy_h2d(ctx, m->indices, indices, (size_t)nnz * sizeof(int32_t)));
        PFZ_TRY(copy_h2d(ctx, m->data, data, (size_t)nnz * sizeof(float)));
    }
    *out = m.release();
    return PFZ_OK;
}

int pfz_csr_shape(const pfz_csr *m, int64_t *n_rows, int64_t *n_cols, int64_t *nnz)
{
    PFZ_REQUIRE(m, "pfz_csr_shape: NULL matrix");
    if (n_rows) *n_rows = m->n_rows;
    if (n_cols) *n_cols = m->n_cols;
    if (nnz) {
        *nnz = csr_nnz(m);
        if (*nnz < 0) return PFZ_ERR_HIP;
    }
    return PFZ_OK;
}

int pfz_csr_download(pfz_ctx *ctx, const pfz_csr *m, int64_t *indptr, int32_t *indices, float *data)
{
    PFZ_REQUIRE(ctx && m, "pfz_csr_download: NULL argument");
    PFZ_HIP(hipSetDevice(ctx->device));
    PFZ_HIP(hipStreamSynchronize(ctx->stream));
    if (indptr) {
        std::vector<int32_t> ip32((size_t)m->n_rows + 1);
        PFZ_TRY(copy_d2h(ctx, ip32.data(), m->indptr, ip32.size() * sizeof(int32_t)));
        for (size_t i = 0; i < ip32.size(); ++i) indptr[i] = ip32[i];
    }
    const int64_t nnz = csr_nnz(m);
    if (nnz < 0) return PFZ_ERR_HIP;
    if (indices && nnz > 0) PFZ_TRY(copy_d2h(ctx, indices, m->indices, (size_t)nnz * sizeof(int32_t)));
    if (data && nnz > 0) PFZ_TRY(copy_d2h(ctx, data, m->data, (size_t)nnz * sizeof(float)));
    return PFZ_OK;
}

void pfz_csr_free(pfz_csr *m)
{
    if (!m) return;
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    if (m->nnz_lazy.pending || m->nnz_lazy.slot) {
        // the copy into the slot may still be in flight: the slot goes back to the pool only behind it
        if (m->nnz_lazy.ev) (void)hipEventSynchronize(m->nnz_lazy.ev);
        pfz::lazy_release(m->ctx, &m->nnz_lazy);
    }
    if (m->indptr) pool_free(m->indptr);
    if (m->indices) pool_free(m->indices);
    if (m->data) pool_free(m->data);
    delete m;
}

// ---- top-n result container -------------------------------------------------

int pfz_topn_alloc(pfz_ctx *ctx, int64_t n_rows, int32_t ntop, pfz_topn **out)
{
    PFZ_REQUIRE(ctx && out && n_rows >= 0 && ntop >= 1, "pfz_topn_alloc: bad args");
    PFZ_HIP(hipSetDevice(ctx->device));
    Owner<pfz_topn, pfz_topn_free> t(new pfz_topn());
    t->ctx = ctx;
    t->n_rows = n_rows;
    t->ntop = ntop;
    // one block: the indices, then (256-byte aligned) the scores -- a small result comes back in ONE device-to-host copy
    size_t n = (size_t)(n_rows > 0 ? n_rows : 1) * (size_t)ntop;
    const size_t idx_bytes = (n * sizeof(int32_t) + 255) & ~(size_t)255;
    PFZ_TRY(pool_alloc(ctx, &t->idx, idx_bytes + n * sizeof(float)));
    t->val = (float *)((char *)t->idx + idx_bytes);
    *out = t.release();
    return PFZ_OK;
}

void pfz_topn_free(pfz_topn *t)
{
    if (!t) return;
    if (t->ctx) (void)hipSetDevice(t->ctx->device);
    if (t->idx) pool_free(t->idx);      // (val lives in the same block)
    delete t;
}

int pfz_topn_clear(pfz_ctx *ctx, pfz_topn *t)
{
    PFZ_REQUIRE(ctx && t, "pfz_topn_clear: NULL argument");
    PFZ_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)t->n_rows * (size_t)t->ntop;
    if (n == 0) return PFZ_OK;
    PFZ_HIP(hipMemsetAsync(t->idx, 0xFF, n * sizeof(int32_t), ctx->stream));   // -1
    PFZ_HIP(hipMemsetAsync(t->val, 0, n * sizeof(float), ctx->stream));
    return PFZ_OK;
}

int pfz_topn_download(pfz_ctx *ctx, const pfz_topn *t, int32_t *out_idx, float *out_val)
{
    PFZ_REQUIRE(ctx && t, "pfz_topn_download: NULL argument");
    PFZ_HIP(hipSetDevice(ctx->device));
    size_t n = (size_t)t->n_rows * (size_t)t->ntop;
    if (n == 0) {
        PFZ_HIP(hipStreamSynchronize(ctx->stream));
        return PFZ_OK;
    }
    const size_t span = (size_t)((const char *)t->val - (const char *)t->idx) + n * sizeof(float);
    if (out_idx && out_val && span <= (64u << 10)) {
        // a small result (a query batch): indices and scores lie in one block -- one copy, one wait (stream order puts it
        // behind the kernels: no synchronise of its own in front)
        char *st = nullptr;
        PFZ_TRY(stage_take(ctx, span, &st));
        PFZ_HIP(hipMemcpyAsync(st, t->idx, span, hipMemcpyDeviceToHost, ctx->stream));
        PFZ_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(out_idx, st, n * sizeof(int32_t));
        memcpy(out_val, st + ((const char *)t->val - (const char *)t->idx), n * sizeof(float));
        ctx->stage_off = 0;
        return PFZ_OK;
    }
    if (out_idx) PFZ_TRY(copy_d2h(ctx, out_idx, t->idx, n * sizeof(int32_t)));
    if (out_val) PFZ_TRY(copy_d2h(ctx, out_val, t->val, n * sizeof(float)));
    return PFZ_OK;
}

int pfz_topn_download_rows_after(pfz_ctx *ctx, const pfz_topn *t, int64_t row_begin, int64_t row_end, int32_t event_slot,
                                 int32_t *out_idx, float *out_val)
{
    PFZ_REQUIRE(ctx && t && out_idx && out_val, "pfz_topn_download_rows_after: NULL argument");
    PFZ_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= t->n_rows, "pfz_topn_download_rows_after: bad row range");
    PFZ_REQUIRE(event_slot >= 0 && event_slot < kEventSlots, "pfz_topn_download_rows_after: bad event slot %d", event_slot);
    PFZ_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)(row_end - row_begin) * (size_t)t->ntop;
    if (n == 0) return PFZ_OK;
    PFZ_TRY(ensure_side_stream(ctx));
    const size_t bytes = n * (sizeof(int32_t) + sizeof(float));
    if (bytes > ctx->stage2_bytes) {
        PFZ_HIP(hipStreamSynchronize(ctx->stream2));
        if (ctx->stage2) PFZ_HIP(hipHostFree(ctx->stage2));
        ctx->stage2 = nullptr;
        ctx->stage2_bytes = 0;
        PFZ_HIP(hipHostMalloc((void **)&ctx->stage2, bytes + bytes / 4, hipHostMallocDefault));
        ctx->stage2_bytes = bytes + bytes / 4;
    }
    // the copy runs on the side stream as soon as the event has fired -- work enqueued on the context stream AFTER
    // the event (the second half of a split match) keeps running beside it
    PFZ_HIP(hipStreamWaitEvent(ctx->stream2, ctx->events[event_slot], 0));
    PFZ_HIP(hipMemcpyAsync(ctx->stage2, t->idx + row_begin * t->ntop, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream2));
    PFZ_HIP(hipMemcpyAsync(ctx->stage2 + n * sizeof(int32_t), t->val + row_begin * t->ntop, n * sizeof(float),
                           hipMemcpyDeviceToHost, ctx->stream2));
    PFZ_HIP(hipStreamSynchronize(ctx->stream2));
    memcpy(out_idx, ctx->stage2, n * sizeof(int32_t));
    memcpy(out_val, ctx->stage2 + n * sizeof(int32_t), n * sizeof(float));
    return PFZ_OK;
}

// the two copies of a pfz_topn_rows_begin job into its pinned half, and the event pfz_topn_rows_finish waits for
static int rows_issue(pfz_ctx *ctx, int32_t half)
{
    pfz_ctx::RowsJob &job = ctx->rows_job[half];
    const pfz_topn *t = job.t;
    const size_t n = ctx->rows_n[half];
    if (n) {
        PFZ_HIP(hipMemcpyAsync(ctx->rows_stage[half], t->idx + job.row_begin * t->ntop, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream2));
        PFZ_HIP(hipMemcpyAsync(ctx->rows_stage[half] + n * sizeof(int32_t), t->val + job.row_begin * t->ntop, n * sizeof(float),
                               hipMemcpyDeviceToHost, ctx->stream2));
    }
    PFZ_HIP(hipEventRecord(ctx->rows_ev[half], ctx->stream2));
    job.issued = true;
    return PFZ_OK;
}

int pfz_topn_rows_begin(pfz_ctx *ctx, const pfz_topn *t, int64_t row_begin, int64_t row_end, int32_t event_slot, int32_t half)
{
    PFZ_REQUIRE(ctx && t, "pfz_topn_rows_begin: NULL argument");
    PFZ_REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= t->n_rows, "pfz_topn_rows_begin: bad row range");
    PFZ_REQUIRE(event_slot >= 0 && event_slot < kEventSlots && (half == 0 || half == 1), "pfz_topn_rows_begin: bad event slot %d / half %d",
                event_slot, half);
    PFZ_HIP(hipSetDevice(ctx->device));
    PFZ_TRY(ensure_side_stream(ctx));
    if (!ctx->rows_ev[0])
        for (hipEvent_t &ev : ctx->rows_ev) PFZ_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const size_t n = (size_t)(row_end - row_begin) * (size_t)t->ntop;
    const size_t bytes = (n * (sizeof(int32_t) + sizeof(float)) + 255) & ~(size_t)255;
    if (bytes > ctx->rows_bytes[half]) {
        if (ctx->rows_stage[half]) {
            PFZ_HIP(hipStreamSynchronize(ctx->stream2));
            PFZ_HIP(hipHostFree(ctx->rows_stage[half]));
        }
        ctx->rows_stage[half] = nullptr;
        ctx->rows_bytes[half] = 0;
        PFZ_HIP(hipHostMalloc((void **)&ctx->rows_stage[half], bytes + bytes / 4, hipHostMallocDefault));
        ctx->rows_bytes[half] = bytes + bytes / 4;
    }
    ctx->rows_n[half] = n;
    pfz_ctx::RowsJob &job = ctx->rows_job[half];
    job.t = t;
    job.row_begin = row_begin;
    job.row_end = row_end;
    job.slot = event_slot;
    job.issued = false;
    const int32_t want = ctx->evt_want[event_slot];
    if (want == 0) {                 // an event of the context's stream: the side stream waits for it
        PFZ_HIP(hipStreamWaitEvent(ctx->stream2, ctx->events[event_slot], 0));
        return rows_issue(ctx, half);
    }
    // an event of a side stream with a word in pinned memory: if it has fired, the copies start now (beside whatever the host does
    // next); if not, pfz_topn_rows_finish polls the word and starts them then
    if (__atomic_load_n(&ctx->evt_flag[event_slot], __ATOMIC_ACQUIRE) == want) return rows_issue(ctx, half);
    return PFZ_OK;
}

int pfz_topn_rows_finish(pfz_ctx *ctx, int32_t half, const int32_t **idx, const float **val)
{
    PFZ_REQUIRE(ctx && idx && val && (half == 0 || half == 1) && ctx->rows_ev[0] && ctx->rows_job[half].t,
                "pfz_topn_rows_finish: no pfz_topn_rows_begin on half %d", half);
    PFZ_HIP(hipSetDevice(ctx->device));
    pfz_ctx::RowsJob &job = ctx->rows_job[half];
    if (!job.issued) {
        PFZ_TRY(event_flag_spin(ctx, job.slot, "pfz_topn_rows_finish"));
        PFZ_TRY(rows_issue(ctx, half));
    }
    PFZ_HIP(hipEventSynchronize(ctx->rows_ev[half]));
    *idx = (const int32_t *)ctx->rows_stage[half];
    *val = (const float *)(ctx->rows_stage[half] + ctx->rows_n[half] * sizeof(int32_t));
    return PFZ_OK;
}

int pfz_topn_upload(pfz_ctx *ctx, pfz_topn *t, const int32_t *idx, const float *val)
{
    PFZ_REQUIRE(ctx && t && idx && val, "pfz_topn_upload: NULL argument");
    PFZ_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)t->n_rows * (size_t)t->ntop;
    PFZ_TRY(copy_h2d(ctx, t->idx, idx, n * sizeof(int32_t)));
    PFZ_TRY(copy_h2d(ctx, t->val, val, n * sizeof(float)));
    return PFZ_OK;
}

int pfz_topn_device_ptrs(const pfz_topn *t, void **idx_dev, void **val_dev, int64_t *n_rows, int32_t *ntop)
{
    PFZ_REQUIRE(t, "pfz_topn_device_ptrs: NULL argument");
    if (idx_dev) *idx_dev = t->idx;
    if (val_dev) *val_dev = t->val;
    if (n_rows) *n_rows = t->n_rows;
    if (ntop) *ntop = t->ntop;
    return PFZ_OK;
}

}  // extern "C"
