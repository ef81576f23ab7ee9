// Sort of 64-bit n-gram codes for the sorted-vocabulary path of the vectoriser (k1_vectorize.hip: n-gram
// codes wider than the presence bitmap can address).  A bitonic network: tiles of 4096 keys are sorted and
// merged in LDS, compare-exchange distances of a tile or more run as one streaming pass over HBM each.
// This path handles a few million codes once per fit; (log2 n - 12)(log2 n - 11)/2 passes of 16 B per key
// are a millisecond or two and need no temporary storage.
#include "pfz_internal.h"

namespace pfz {

constexpr int kSortTile = 4096;        // keys per workgroup tile: 32 KiB of LDS
constexpr int kSortThreads = 1024;

__device__ inline void cmp_swap(uint64_t &a, uint64_t &b, bool ascending)
{
    if ((a > b) == ascending) {
        const uint64_t t = a;
        a = b;
        b = t;
    }
}

// steps j = j_hi, j_hi/2, ..., 1 of stage k on the tile held in LDS; the direction of element i is
// ascending when (global index & k) == 0
__device__ inline void tile_steps(uint64_t *s, int64_t tile_base, int64_t k, int j_hi)
{
    for (int j = j_hi; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < kSortTile / 2; t += kSortThreads) {
            const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
            const bool asc = ((tile_base + lo) & k) == 0;
            cmp_swap(s[lo], s[lo | j], asc);
        }
        __syncthreads();
    }
}

// stages k = 2 .. kSortTile: every tile becomes a sorted run (ascending / descending alternately)
__global__ __launch_bounds__(kSortThreads) void k_bitonic_tile_sort(uint64_t *__restrict__ d)
{
    __shared__ uint64_t s[kSortTile];
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
    for (int t = threadIdx.x; t < kSortTile; t += kSortThreads) s[t] = d[base + t];
    __syncthreads();
    for (int k = 2; k <= kSortTile; k <<= 1) tile_steps(s, base, k, k >> 1);
    for (int t = threadIdx.x; t < kSortTile; t += kSortThreads) d[base + t] = s[t];
}

// one step of stage k with a distance j >= kSortTile: partners live in different tiles
__global__ __launch_bounds__(256) void k_bitonic_global(uint64_t *__restrict__ d, int64_t n_half, int64_t j, int64_t k)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_half) return;
    const int64_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    uint64_t a = d[lo], b = d[lo | j];
    const uint64_t a0 = a;
    cmp_swap(a, b, (lo & k) == 0);
    if (a != a0) {
        d[lo] = a;
        d[lo | j] = b;
    }
}

// the last steps j = kSortTile/2 .. 1 of a stage k > kSortTile
__global__ __launch_bounds__(kSortThreads) void k_bitonic_tile_merge(uint64_t *__restrict__ d, int64_t k)
{
    __shared__ uint64_t s[kSortTile];
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
    for (int t = threadIdx.x; t < kSortTile; t += kSortThreads) s[t] = d[base + t];
    __syncthreads();
    tile_steps(s, base, k, kSortTile >> 1);
    for (int t = threadIdx.x; t < kSortTile; t += kSortThreads) d[base + t] = s[t];
}

__global__ __launch_bounds__(256) void k_sort_pad(const uint64_t *__restrict__ in, int64_t n, int64_t n_pad,
                                                   uint64_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_pad) out[i] = i < n ? in[i] : ~0ull;
}

int64_t sort_codes_capacity(int64_t n)
{
    int64_t p = kSortTile;
    while (p < n) p <<= 1;
    return p;
}

// out must hold sort_codes_capacity(n) keys; out[0..n) = in[0..n) ascending on return
int sort_codes_u64(pfz_ctx *ctx, const uint64_t *in, uint64_t *out, int64_t n)
{
    if (n <= 0) return PFZ_OK;
    const int64_t n_pad = sort_codes_capacity(n);
    hipLaunchKernelGGL(k_sort_pad, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, ctx->stream, in, n, n_pad, out);
    const unsigned tiles = (unsigned)(n_pad / kSortTile);
    hipLaunchKernelGGL(k_bitonic_tile_sort, dim3(tiles), dim3(kSortThreads), 0, ctx->stream, out);
    for (int64_t k = (int64_t)kSortTile << 1; k <= n_pad; k <<= 1) {
        for (int64_t j = k >> 1; j >= kSortTile; j >>= 1)
            hipLaunchKernelGGL(k_bitonic_global, dim3((unsigned)((n_pad / 2 + 255) / 256)), dim3(256), 0, ctx->stream, out,
                               n_pad / 2, j, k);
        hipLaunchKernelGGL(k_bitonic_tile_merge, dim3(tiles), dim3(kSortThreads), 0, ctx->stream, out, k);
    }
    PFZ_HIP(hipGetLastError());
    return PFZ_OK;
}

}  // namespace pfz
