// K5 -- dense cosine top-n for embedding matrices (fp32 MFMA).
//
// Replaces the dense branches of the reference's cosine_similarity operator
// (polyfuzz/models/_utils.py:74-77,95-102; called with precomputed vectors from
// Embeddings.match, _embeddings.py:127-133): cosine = normalize(A) . normalize(B)^T
// (sklearn.metrics.pairwise.cosine_similarity), then the top-n of every row,
// diagonal excluded for self-match.
//
// Plan:
//   k5_inv_norms        : 1/||row|| for both matrices (wave per row).
//   k5_gemm_panel_pipe  : S[P x n_to] = A_panel . B^T scaled by both inverse norms (matrices are stored with their
//                         width padded to a multiple of 32); 128x128x32 workgroup tiles, 4 waves x (2x2)
//                         v_mfma_f32_32x32x2_f32 -- exact fp32 products at the fp32 peak rate -- software-pipelined
//                         through two LDS buffers, and the maximum of every row over each 64-column block on the
//                         side (M).  The score panel IS written to HBM (two panels of <= 4 GiB): at d = 768 that is
//                         4 B per 1536 flops, far below the machine balance.
//   k5_row_topn         : wave per row; with M it selects the ntop-th largest block maximum and reads only the
//                         blocks that reach it, without M it streams the row (float4); threshold filter, 64-bit
//                         keys score_bits<<32 | ~col, compaction by wave-max rounds (same scheme as K3), writes
//                         (idx, score) by (score desc, col asc).
#include "pfz_internal.h"

#include <algorithm>
#include <stdlib.h>

namespace pfz {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTile = 128;   // workgroup tile (rows of A x rows of B)
constexpr int kBK = 32;      // k-depth staged per step

// normalize == 0: raw dot products are wanted, every scale factor is 1
__global__ __launch_bounds__(256) void k5_inv_norms(const float *__restrict__ x, int64_t n, int64_t d,
                                                     float *__restrict__ inv, int32_t normalize)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    if (!normalize) {
        if (lane == 0) inv[row] = 1.f;
        return;
    }
    const float *p = x + row * d;
    double ss = 0.0;
    for (int64_t k = lane; k < d; k += 64) ss += (double)p[k] * (double)p[k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) inv[row] = ss > 0.0 ? (float)(1.0 / sqrt(ss)) : 0.f;   // zero rows stay zero (sklearn normalize)
}

// Row maxima of a wave's 64 x 64 corner.  x[q] (q = 16 i + r) is the lane's maximum over its two columns of
// accumulator row-slot q; the maximum over the 32 lanes of a half-wave is wanted for all 32 slots.  Halving
// exchange: each step pairs lanes (row_mirror, row_half_mirror, quad xor 2, quad xor 1 by DPP, then lane ^ 16 by
// ds_swizzle), a lane keeps one half of its slots, hands the other half to its partner and folds in what it gets --
// 16 + 8 + 4 + 2 + 1 exchanges instead of 32 x 5, and every lane ends with ONE slot's maximum:
// slot(lane) = bit3 bit2 bit1 bit0 bit4 of the lane id (most significant first).
template <int CTRL>
__device__ inline float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ inline float block_row_max(float (&x)[32], int lane)
{
    const bool p3 = lane & 8, p2 = lane & 4, p1 = lane & 2, p0 = lane & 1, p4 = lane & 16;
    float y[16], z[8], u[4], w[2];
#pragma unroll
    for (int t = 0; t < 16; ++t) y[t] = fmaxf(p3 ? x[16 + t] : x[t], dpp_f32<0x140>(p3 ? x[t] : x[16 + t]));     // row_mirror
#pragma unroll
    for (int t = 0; t < 8; ++t) z[t] = fmaxf(p2 ? y[8 + t] : y[t], dpp_f32<0x141>(p2 ? y[t] : y[8 + t]));        // row_half_mirror
#pragma unroll
    for (int t = 0; t < 4; ++t) u[t] = fmaxf(p1 ? z[4 + t] : z[t], dpp_f32<0x4E>(p1 ? z[t] : z[4 + t]));         // quad_perm 2,3,0,1
#pragma unroll
    for (int t = 0; t < 2; ++t) w[t] = fmaxf(p0 ? u[2 + t] : u[t], dpp_f32<0xB1>(p0 ? u[t] : u[2 + t]));         // quad_perm 1,0,3,2
    const float other = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(p4 ? w[0] : w[1]), 0x401F));   // lane ^ 16
    return fmaxf(p4 ? w[1] : w[0], other);
}
__device__ inline int block_row_slot(int lane)
{
    return ((lane >> 3) & 1) << 4 | ((lane >> 2) & 1) << 3 | ((lane >> 1) & 1) << 2 | (lane & 1) << 1 | ((lane >> 4) & 1);
}

__device__ inline float f4c(const float4 &v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

// The tile program for d % 32 == 0 (embedding widths: 128 ... 768 ... 4096), software-pipelined: the operands of chunk
// c + 2 travel from HBM / L2 into registers and those of chunk c + 1 from registers into the other LDS buffer while the
// MFMAs of chunk c run out of LDS; one barrier per chunk.  Rows beyond the matrix edge are clamped to the last row
// (their products are not stored), so the loop has no edge tests.  Its shape is what the counters and what-if builds
// of its predecessor (round 2's first pipelined kernel, removed; DESIGN.md section 4 keeps the numbers) asked for:
//  * LDS rows of 36 floats: operands staged with ds_write_b128, MFMA fragments read with ds_read_b128 -- lane (r, h)
//    takes the four floats k = 8t + 4h .. + 3 of row r, and MFMA step s of sub-step t multiplies the k-pairs
//    {8t + s, 8t + 4 + s} (A and B use the same assignment, so the sum is the same dot product in another order).
//    24 LDS instructions per wave and k-chunk instead of 48; both access patterns are conflict-free
//    (row pitch 36 = 4 (9 r mod 16) banks: the 16 lanes of a b128 group hit 16 distinct bank quads).
//  * the global loads of chunk c + 2 and the LDS stores of chunk c + 1 are spread over the chunk, one operand pair
//    per group of eight MFMAs (eight loads issued back to back kept the wave out of the matrix pipe for ~300 cycles
//    per chunk: 8 % of the GEMM, with L2-hot loads just the same); addresses are a wave-uniform base + a 32-bit lane
//    offset, so a load costs no vector ALU work.
__global__ __launch_bounds__(256, 2) void k5_gemm_panel_pipe(const float *__restrict__ A, const float *__restrict__ B,
                                                              const float *__restrict__ inv_a, const float *__restrict__ inv_b,
                                                              int64_t a0, int64_t a1, int64_t n_b, int64_t d,
                                                              float *__restrict__ S, int64_t ld, int tiles_m, int tiles_n,
                                                              float *__restrict__ M, int64_t ldm)
{
    constexpr int BK = 32, LD = 36, NP = 4;
    __shared__ __attribute__((aligned(16))) float As[2][kTile * LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][kTile * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroup -> tile.  Consecutive workgroup ids go round-robin over the 8 XCDs and each XCD has its own L2, so the
    // 64 workgroups an XCD runs at a time (32 CUs x 2) get one 8 x 8 block of tiles: 8 A + 8 B tiles feed 64 tile
    // products out of that L2 (a 1-D sweep over row tiles gives every XCD 32 A tiles + 2 B tiles for the same 64).
    // The grid is a whole number of 8-block rounds; workgroups of blocks or tiles that do not exist leave at once.
    const int w = blockIdx.x, xcd = w & 7, idx = w >> 3, pos = idx & 63;
    const int g = (idx >> 6) * 8 + xcd, bm = (tiles_m + 7) >> 3;
    const int tm = (g % bm) * 8 + (pos & 7), tn = (g / bm) * 8 + (pos >> 3);
    if (tm >= tiles_m || tn >= tiles_n) return;
    const int64_t row0 = a0 + (int64_t)tm * kTile;
    const int64_t col0 = (int64_t)tn * kTile;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = tid >> 3, lk = (tid & 7) * 4;          // staging: 8 threads per tile row, 32 rows per pass
    // buffer loads: descriptor = the tile's first operand row (wave-uniform), 32-bit lane offset, scalar k offset
    const __amdgpu_buffer_rsrc_t resA = __builtin_amdgcn_make_buffer_rsrc((void *)(A + row0 * d), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t resB = __builtin_amdgcn_make_buffer_rsrc((void *)(B + col0 * d), 0, 0x7fffffff, 0x00020000);
    uint32_t offA[NP], offB[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {        // rows beyond the edge are clamped to the last row (their products are not stored)
        offA[p] = (uint32_t)((min(row0 + lr + p * 32, a1 - 1) - row0) * d + lk) * 4u;
        offB[p] = (uint32_t)((min(col0 + lr + p * 32, n_b - 1) - col0) * d + lk) * 4u;
    }
    u32x4 ra[NP], rb[NP];
    auto load = [&](int p, int k) {
        ra[p] = __builtin_amdgcn_raw_buffer_load_b128(resA, offA[p], k * 4, 0);
        rb[p] = __builtin_amdgcn_raw_buffer_load_b128(resB, offB[p], k * 4, 0);
    };
    auto stage = [&](int p, int buf) {
        *(u32x4 *)(As[buf] + (lr + p * 32) * LD + lk) = ra[p];
        *(u32x4 *)(Bs[buf] + (lr + p * 32) * LD + lk) = rb[p];
    };
#pragma unroll
    for (int p = 0; p < NP; ++p) load(p, 0);
#pragma unroll
    for (int p = 0; p < NP; ++p) stage(p, 0);
    const int dk = (int)d;
#pragma unroll
    for (int p = 0; p < NP; ++p) load(p, min(BK, dk - BK));
    __syncthreads();

    const int frag_off = (lane & 31) * LD + 4 * (lane >> 5);
    int cur = 0;
    for (int k0 = 0; k0 < dk; k0 += BK) {
        const int k2 = min(k0 + 2 * BK, dk - BK);      // (the last two chunks re-fetch the last one: no branches in the loop)
        const float *ap = As[cur] + wm * LD + frag_off, *bp = Bs[cur] + wn * LD + frag_off;
        auto frag = [&](int t, float4 (&a)[2], float4 (&b)[2]) {
            a[0] = *(const float4 *)(ap + 8 * t);
            a[1] = *(const float4 *)(ap + 32 * LD + 8 * t);
            b[0] = *(const float4 *)(bp + 8 * t);
            b[1] = *(const float4 *)(bp + 32 * LD + 8 * t);
        };
        auto mfma8 = [&](const float4 (&a)[2], const float4 (&b)[2], int s0) {
#pragma unroll
            for (int s = s0; s < s0 + 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(a[i], s), f4c(b[j], s), acc[i][j], 0, 0, 0);
        };
        // one operand pair per group of eight MFMAs: groups 0..3 store chunk c + 1 (in registers since the previous
        // chunk) to the other LDS buffer, groups 4..7 fetch chunk c + 2 into the registers just freed
        auto item = [&](int grp) {
            if (grp < 4) stage(grp, cur ^ 1);
            else load(grp - 4, k2);
        };
        float4 fa0[2], fb0[2], fa1[2], fb1[2];
        frag(0, fa0, fb0);
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            frag(t + 1, fa1, fb1);                 // fragments are read one sub-step (16 MFMAs) ahead
            __builtin_amdgcn_sched_barrier(0);
            mfma8(fa0, fb0, 0);
            item(2 * t);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(fa0, fb0, 2);
            item(2 * t + 1);
            if (t + 2 < 4) frag(t + 2, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(fa1, fb1, 0);
            item(2 * t + 2);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(fa1, fb1, 2);
            item(2 * t + 3);
        }
        __syncthreads();
        cur ^= 1;
    }

    // Epilogue.  MFMA 32x32 accumulator r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31.
    // Besides the scores, every wave leaves the maximum of each of its 64 rows over its 64 columns in M[row][col / 64]:
    // the row top-n reads those block maxima (1/64 of the panel) and then only the few blocks that can hold a winner.
    const int cb = (int)((col0 + wn) >> 6);
    if (row0 + kTile <= a1) {
        // interior tile (all but the last row tile of the last panel; ld is a whole number of tiles): the 1/|a| factors
        // come as eight float4 loads issued together, the 64 stores go out back to back from a wave-uniform base plus
        // one 32-bit lane offset.  (Row-by-row predicated code makes the compiler wait for EVERYTHING in flight,
        // the previous store included, before each element: 5.5 us per tile, 13 % of a tile's MFMA time.)
        const int uwm = __builtin_amdgcn_readfirstlane(wm), uwn = __builtin_amdgcn_readfirstlane(wn);
        const float4 *ia = (const float4 *)(inv_a + row0 + uwm) + (lane >> 5);
        float4 sa[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) sa[i][q] = ia[i * 8 + q * 2];
        float sb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = col0 + uwn + j * 32 + (lane & 31);
            sb[j] = col < n_b ? inv_b[col] : 0.f;
        }
        float *tile = S + (row0 - a0 + uwm) * ld + col0 + uwn;
        const uint32_t lane_off = (uint32_t)(4 * (lane >> 5)) * (uint32_t)ld + (uint32_t)(lane & 31);
        float x[32];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = f4c(sa[i][r / 4], r % 4);
                float *rowp = tile + (int64_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ld;
                const float v0 = acc[i][0][r] * f * sb[0], v1 = acc[i][1][r] * f * sb[1];
                rowp[lane_off] = v0;
                (rowp + 32)[lane_off] = v1;
                x[i * 16 + r] = fmaxf(v0, v1);
            }
        if (M) {
            const float m = block_row_max(x, lane);
            const int q = block_row_slot(lane), rl = (q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * (lane >> 5);
            M[(row0 - a0 + uwm + rl) * ldm + cb] = m;
        }
        return;
    }
    const float sb0 = col0 + wn + (lane & 31) < n_b ? inv_b[col0 + wn + (lane & 31)] : 0.f;
    const float sb1 = col0 + wn + 32 + (lane & 31) < n_b ? inv_b[col0 + wn + 32 + (lane & 31)] : 0.f;
    float xe[32];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const bool ok = row < a1;
            const float f = ok ? inv_a[row] : 0.f;
            const float v0 = acc[i][0][r] * f * sb0, v1 = acc[i][1][r] * f * sb1;
            if (ok) {
                float *rowp = S + (row - a0) * ld + col0 + wn + (lane & 31);
                rowp[0] = v0;
                rowp[32] = v1;
            }
            xe[i * 16 + r] = ok ? fmaxf(v0, v1) : 0.f;
        }
    }
    if (M) {
        const float m = block_row_max(xe, lane);
        const int q = block_row_slot(lane);
        const int64_t row = row0 + wm + (q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * (lane >> 5);
        if (row < a1) M[(row - a0) * ldm + cb] = m;
    }
}

__device__ inline uint64_t wave_max_u64_5(uint64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, d, 64);
        uint32_t hi = __shfl_xor((uint32_t)(v >> 32), d, 64);
        uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

// One wave per score row: top-n of the scores > thr.  kCap5: candidate keys per wave (>= ntop + 1 + 64).
// kDeep (round 6: top_n beyond the 1024 keys one pass keeps -- the reference clips top_n to the number of distinct to-strings
// only, _utils.py:54-56): a PASS of a deep top-n.  It keeps the `ntop` best keys strictly BELOW ub[row] (keys are distinct:
// score bits << 32 | ~column), writes them at columns col0 .. of the row's out_ld-wide result and leaves its last key in ub[row]
// for the pass that follows (0 when the row has run out of candidates: nothing is below 0).  The scheme of K3's deep passes.
template <int kCap5, bool kDeep = false>
__global__ __launch_bounds__(256) void k5_row_topn(const float *__restrict__ S, int64_t ld, int64_t a0, int64_t a1,
                                                    int64_t n_b, int32_t ntop, float lower_bound, int32_t exclude_diag,
                                                    int64_t diag_offset, int32_t *__restrict__ out_idx,
                                                    float *__restrict__ out_val, const float *__restrict__ M, int64_t ldm,
                                                    int32_t out_ld = 0, int32_t col0 = 0, uint64_t *__restrict__ ub = nullptr)
{
    __shared__ __attribute__((aligned(16))) uint64_t cand_all[4][kCap5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = a0 + (int64_t)blockIdx.x * 4 + wave;
    if (row >= a1) return;
    uint64_t *cand = cand_all[wave];
    const float *s = S + (row - a0) * ld;
    const int64_t self_col = exclude_diag ? row + diag_offset : -1;
    int cnt = 0;
    float thr = lower_bound;
    int want = ntop;            // how many keys a compaction keeps
    const uint64_t ubk = kDeep ? ub[row - a0] : ~0ull;

    // sorted == false (intermediate compactions) and a large top_n: select the ntop-th largest key bit by bit
    // (64 ballot steps) instead of one wave-max round per kept key -- the same scheme as K3's compact()
    auto compact = [&](bool sorted) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint64_t e[kCap5 / 64];
#pragma unroll
        for (int i = 0; i < kCap5 / 64; ++i) e[i] = lane + 64 * i < cnt ? cand[lane + 64 * i] : 0ull;
        __builtin_amdgcn_wave_barrier();
        if (!sorted && want > 16) {
            if (cnt <= want) return;
            uint64_t T = 0ull;
            for (int bit = 62; bit >= 0; --bit) {   // positive floats: bit 63 is never set
                const uint64_t c = T | (1ull << bit);
                int n = 0;
#pragma unroll
                for (int i = 0; i < kCap5 / 64; ++i) n += __popcll(__ballot(e[i] >= c));
                T = n >= want ? c : T;
            }
            int base = 0;
#pragma unroll
            for (int i = 0; i < kCap5 / 64; ++i) {
                const bool keep_it = e[i] >= T;
                const uint64_t mk = __ballot(keep_it);
                if (keep_it) cand[base + __popcll(mk & ((1ull << lane) - 1ull))] = e[i];
                base += __popcll(mk);
            }
            cnt = base;
            const float t = __uint_as_float((uint32_t)(T >> 32) - 1u);
            thr = t > thr ? t : thr;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            return;
        }
        const int keep = cnt < want ? cnt : want;
        uint64_t best = 0;
        for (int r = 0; r < keep; ++r) {
            uint64_t m = e[0];
#pragma unroll
            for (int i = 1; i < kCap5 / 64; ++i) m = e[i] > m ? e[i] : m;
            best = wave_max_u64_5(m);
#pragma unroll
            for (int i = 0; i < kCap5 / 64; ++i)
                if (e[i] == best) e[i] = 0ull;
            if (lane == 0) cand[r] = best;
        }
        cnt = keep;
        if (keep == want) {
            const float t = __uint_as_float((uint32_t)(best >> 32) - 1u);   // accept >= the want-th score
            thr = t > thr ? t : thr;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    if (M) {
        // With block maxima (M[row][b] = max of the row's scores in columns 64 b .. 64 b + 63, diagonal and padding
        // included): the ntop-th largest block maximum T is a lower bound of the ntop-th best score (ntop blocks hold
        // an element >= T each; one more when the diagonal is to be skipped), so only blocks with a maximum >= T can
        // hold a winner -- usually ntop of the n_b / 64.  Pass 1 selects T with the machinery below, pass 2 reads
        // those blocks' scores.
        const float *m = M + (row - a0) * ldm;
        const int64_t n_blocks = (n_b + 63) >> 6;
        want = ntop + (exclude_diag ? 1 : 0);
        for (int64_t b0 = 0; b0 < n_blocks; b0 += 64) {
            const int64_t b = b0 + lane;
            const float v = b < n_blocks ? m[b] : 0.f;
            const bool pred = v > thr;
            const uint64_t mk = __ballot(pred);
            if (!mk) continue;
            const int pos = cnt + __popcll(mk & ((1ull << lane) - 1ull));
            if (pred) cand[pos] = ((uint64_t)__float_as_uint(v) << 32) | (uint32_t)(~(uint32_t)b);
            cnt += __popcll(mk);
            if (cnt > kCap5 - 64) compact(false);
        }
        compact(true);              // cnt == want: thr is now just below the want-th largest block maximum
        cnt = 0;
        want = ntop;
        __builtin_amdgcn_wave_barrier();
        for (int64_t b0 = 0; b0 < n_blocks; b0 += 64) {
            const int64_t b = b0 + lane;
            uint64_t hot = __ballot(b < n_blocks && m[b] > thr);
            while (hot) {
                const int t = __builtin_ctzll(hot);
                hot &= hot - 1;
                const int64_t j = (b0 + t) * 64 + lane;
                const float v = s[j];                       // (j < ld: ld is a whole number of blocks)
                const bool pred = v > thr && j < n_b && j != self_col;
                const uint64_t mk = __ballot(pred);
                if (mk) {
                    const int pos = cnt + __popcll(mk & ((1ull << lane) - 1ull));
                    if (pred) cand[pos] = ((uint64_t)__float_as_uint(v) << 32) | (uint32_t)(~(uint32_t)j);
                    cnt += __popcll(mk);
                    if (cnt > kCap5 - 64) compact(false);
                }
            }
        }
    }
    else
    for (int64_t c0 = 0; c0 < ld; c0 += 256) {
        const int64_t c = c0 + lane * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < ld) v = *(const float4 *)(s + c);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
        if (!__ballot(mx > thr)) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t j = c + q;
            bool pred = vv[q] > thr && j < n_b && j != self_col;
            if (kDeep) pred = pred && (((uint64_t)__float_as_uint(vv[q]) << 32) | (uint32_t)(~(uint32_t)j)) < ubk;
            const uint64_t mk = __ballot(pred);
            if (mk) {
                const int pos = cnt + __popcll(mk & ((1ull << lane) - 1ull));
                if (pred) cand[pos] = ((uint64_t)__float_as_uint(vv[q]) << 32) | (uint32_t)(~(uint32_t)j);
                cnt += __popcll(mk);
                if (cnt > kCap5 - 64) compact(false);
            }
        }
    }
    compact(true);
    const int64_t o_ld = kDeep ? out_ld : ntop;
    for (int r = lane; r < ntop; r += 64) {
        const uint64_t key = r < cnt ? cand[r] : 0ull;
        out_idx[row * o_ld + col0 + r] = key ? (int32_t)(~(uint32_t)key) : -1;
        out_val[row * o_ld + col0 + r] = key ? __uint_as_float((uint32_t)(key >> 32)) : 0.f;
    }
    if (kDeep && lane == 0) ub[row - a0] = cnt >= ntop ? cand[ntop - 1] : 0ull;
}

}  // namespace pfz

using namespace pfz;

struct pfz_dense {
    pfz_ctx *ctx = nullptr;
    int64_t n = 0, dim = 0;
    int64_t ld = 0;          // dim rounded up to a multiple of 32 (the GEMM's k-chunk), the extra columns are zero
    int32_t normalize = 1;
    float *x = nullptr;      // device [n][ld] row-major
    float *inv = nullptr;    // device [n]: 1 / ||row|| (1 when normalize == 0)
};

extern "C" {

void pfz_dense_free(pfz_dense *m)
{
    if (!m) return;
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    if (m->x) pool_free(m->x);
    if (m->inv) pool_free(m->inv);
    delete m;
}

int pfz_dense_upload(pfz_ctx *ctx, const float *vec, int64_t n, int64_t dim, int32_t normalize, pfz_dense **out)
{
    PFZ_REQUIRE(ctx && out && (n == 0 || vec), "pfz_dense_upload: NULL argument");
    PFZ_REQUIRE(n >= 0 && dim >= 1, "pfz_dense_upload: bad shape %lld x %lld", (long long)n, (long long)dim);
    if (n >= ((int64_t)1 << 31) - 256) {
        set_error("pfz_dense_upload: %lld rows exceed the int32 result indices", (long long)n);
        return PFZ_ERR_UNSUPPORTED;
    }
    PFZ_HIP(hipSetDevice(ctx->device));
    Owner<pfz_dense, pfz_dense_free> m(new pfz_dense());
    m->ctx = ctx;
    m->n = n;
    m->dim = dim;
    // Widths that are not a multiple of the GEMM's 32-column k-chunk (300-d word vectors) are padded with zero columns
    // on the device: dot products and norms do not change and every width takes the pipelined tile program.
    const int64_t ld = (dim + kBK - 1) / kBK * kBK;
    m->ld = ld;
    m->normalize = normalize ? 1 : 0;
    PFZ_TRY(pool_alloc(ctx, &m->x, (size_t)(n > 0 ? n : 1) * (size_t)ld * sizeof(float)));
    PFZ_TRY(pool_alloc(ctx, &m->inv, (size_t)(n > 0 ? n : 1) * sizeof(float)));
    if (n > 0) {
        if (ld == dim)
            PFZ_TRY(copy_h2d(ctx, m->x, vec, (size_t)n * (size_t)dim * sizeof(float)));
        else {
            const int64_t rows_per = std::max<int64_t>(1, ((int64_t)32 << 20) / (ld * (int64_t)sizeof(float)));
            std::vector<float> padded((size_t)std::min(rows_per, n) * (size_t)ld, 0.f);
            for (int64_t r0 = 0; r0 < n; r0 += rows_per) {
                const int64_t rows = std::min(rows_per, n - r0);
                for (int64_t r = 0; r < rows; ++r)
                    std::copy(vec + (r0 + r) * dim, vec + (r0 + r + 1) * dim, padded.begin() + (size_t)r * (size_t)ld);
                PFZ_TRY(copy_h2d(ctx, m->x + r0 * ld, padded.data(), (size_t)rows * (size_t)ld * sizeof(float)));
            }
        }
        hipLaunchKernelGGL(k5_inv_norms, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, m->x, n, ld, m->inv, m->normalize);
        PFZ_HIP(hipGetLastError());
    }
    *out = m.release();
    return PFZ_OK;
}

int pfz_dense_shape(const pfz_dense *m, int64_t *n, int64_t *dim)
{
    PFZ_REQUIRE(m, "pfz_dense_shape: NULL matrix");
    if (n) *n = m->n;
    if (dim) *dim = m->dim;
    return PFZ_OK;
}

int pfz_dense_topn(pfz_ctx *ctx, const pfz_dense *from, const pfz_dense *to, int32_t ntop, float lower_bound,
                   int32_t exclude_diag, int64_t diag_offset, pfz_topn *out)
{
    PFZ_REQUIRE(ctx && from && to && out, "pfz_dense_topn: NULL argument");
    PFZ_REQUIRE(from->dim == to->dim, "pfz_dense_topn: from-vectors have %lld columns, to-vectors %lld", (long long)from->dim,
                (long long)to->dim);
    PFZ_REQUIRE(ntop >= 1, "pfz_dense_topn: ntop must be >= 1");
    PFZ_REQUIRE(lower_bound == lower_bound, "pfz_dense_topn: lower_bound is NaN");
    constexpr int32_t kDeepPass = 1024;      // keys one pass keeps (k5_row_topn<1152>)
    PFZ_REQUIRE(out->n_rows >= from->n && out->ntop == ntop, "pfz_dense_topn: result buffer is %lldx%d, need %lldx%d",
                (long long)out->n_rows, out->ntop, (long long)from->n, ntop);
    const int64_t n_from = from->n, n_to = to->n, dim = from->ld;        // the padded width: a multiple of 32
    if (n_from == 0) return PFZ_OK;
    PFZ_HIP(hipSetDevice(ctx->device));
    if (lower_bound < 0.f) lower_bound = 0.f;   // non-positive similarities are "no match" (_utils.py:122-123)
    struct Buf {
        void *p = nullptr;
        ~Buf() { if (p) pool_free(p); }
    } dS[2], dM[2], dU[2];
    const int64_t ld = ((n_to + 255) / 256) * 256;                      // whole float4 x 64-lane steps
    // Two score panels of <= 4 GiB.  At 500 000 to-vectors that is 2048 rows: each B tile
    // serves 16 row tiles per panel.  (16 GiB panels are 1.5 % faster per step -- B is re-read once per panel -- but
    // their first allocation costs a second, which a one-shot host call cannot afford.)  The row top-n of panel p
    // runs on a side stream while the GEMM of panel p + 1 fills the other buffer; with block maxima it is a ~0.2 ms
    // kernel per panel and the overlap no longer matters, without them (d % 32 != 0) it is a full read of the panel.
    const int64_t panel_bytes = (int64_t)4 << 30;
    int64_t panel = ld > 0 ? panel_bytes / (ld * 4) : n_from;
    if (const char *forced = getenv("PFZ_K5_PANEL_ROWS")) panel = atoll(forced);   // tests: several panels on small inputs
    panel = std::max<int64_t>(kTile, std::min<int64_t>(panel / kTile * kTile, ((n_from + kTile - 1) / kTile) * kTile));
    const int64_t n_panels = (n_from + panel - 1) / panel;
    const bool two = n_panels > 1 && !getenv("PFZ_K5_NO_OVERLAP");       // env: A/B timing
    if (two) PFZ_TRY(ensure_side_stream(ctx));
    if (ld > 0) {
        PFZ_TRY(pool_alloc(ctx, &dS[0].p, (size_t)panel * (size_t)ld * sizeof(float)));
        if (two) PFZ_TRY(pool_alloc(ctx, &dS[1].p, (size_t)panel * (size_t)ld * sizeof(float)));
        // block maxima (one float per row and 64 columns), written by the second-generation GEMM
        PFZ_TRY(pool_alloc(ctx, &dM[0].p, (size_t)panel * (size_t)(ld / 64) * sizeof(float)));
        if (two) PFZ_TRY(pool_alloc(ctx, &dM[1].p, (size_t)panel * (size_t)(ld / 64) * sizeof(float)));
    }
    if (ntop > kDeepPass) {      // a deep top-n: every row's last key of the pass before
        PFZ_TRY(pool_alloc(ctx, &dU[0].p, (size_t)panel * sizeof(uint64_t)));
        if (two) PFZ_TRY(pool_alloc(ctx, &dU[1].p, (size_t)panel * sizeof(uint64_t)));
    }
    hipEvent_t *ready = ctx->side_events, *consumed = ctx->side_events + 2;
    int64_t pi = 0;
    for (int64_t a0 = 0; a0 < n_from; a0 += panel, ++pi) {
        const int64_t a1 = std::min(n_from, a0 + panel);
        const int buf = two ? (int)(pi & 1) : 0;
        float *S = (float *)dS[buf].p;
        const float *M = nullptr;          // set when the GEMM leaves block maxima
        if (two && pi >= 2) PFZ_HIP(hipStreamWaitEvent(ctx->stream, consumed[buf], 0));   // the top-n of panel pi - 2 read this buffer
        if (ld > 0) {
            ProfScope ps(ctx, "k5_gemm_panel");
            const int tiles_m = (int)((a1 - a0 + kTile - 1) / kTile), tiles_n = (int)(ld / kTile);
            if (n_to > 0) {
                // 1-D grid of 8 x 8 tile blocks dealt round-robin to the XCDs (see the kernel)
                const dim3 grid_p((unsigned)((((tiles_m + 7) / 8) * ((tiles_n + 7) / 8) + 7) / 8 * 512));
                M = getenv("PFZ_K5_NO_BLOCK_MAX") ? nullptr : (const float *)dM[buf].p;      // A/B knob, tests
                hipLaunchKernelGGL(k5_gemm_panel_pipe, grid_p, dim3(256), 0, ctx->stream, from->x, to->x, from->inv, to->inv, a0, a1,
                                   n_to, dim, S, ld, tiles_m, tiles_n, (float *)M, ld / 64);
            }
        }
        hipStream_t ts = two ? ctx->stream2 : ctx->stream;
        if (two) {
            PFZ_HIP(hipEventRecord(ready[buf], ctx->stream));
            PFZ_HIP(hipStreamWaitEvent(ts, ready[buf], 0));
        }
        {
            ProfScope ps(ctx, "k5_row_topn", ts);
            if (ntop > kDeepPass) {
                // passes of 1024 over the same score panel, each continuing strictly below the last key of the one before (the
                // block maxima are of no use below a bound: the passes stream the rows)
                PFZ_HIP(hipMemsetAsync(dU[buf].p, 0xff, (size_t)(a1 - a0) * sizeof(uint64_t), ts));
                for (int32_t col0 = 0; col0 < ntop; col0 += kDeepPass)
                    hipLaunchKernelGGL((k5_row_topn<1152, true>), dim3((unsigned)((a1 - a0 + 3) / 4)), dim3(256), 0, ts, (const float *)S, ld,
                                       a0, a1, n_to, std::min(kDeepPass, ntop - col0), lower_bound, exclude_diag, diag_offset, out->idx,
                                       out->val, (const float *)nullptr, ld / 64, ntop, col0, (uint64_t *)dU[buf].p);
            }
            else if (ntop <= 128)
                hipLaunchKernelGGL(k5_row_topn<256>, dim3((unsigned)((a1 - a0 + 3) / 4)), dim3(256), 0, ts, (const float *)S, ld, a0,
                                   a1, n_to, ntop, lower_bound, exclude_diag, diag_offset, out->idx, out->val, M, ld / 64);
            else
                hipLaunchKernelGGL(k5_row_topn<1152>, dim3((unsigned)((a1 - a0 + 3) / 4)), dim3(256), 0, ts, (const float *)S, ld, a0,
                                   a1, n_to, ntop, lower_bound, exclude_diag, diag_offset, out->idx, out->val, M, ld / 64);
        }
        if (two) PFZ_HIP(hipEventRecord(consumed[buf], ts));
    }
    if (two) {   // later work on the context stream (downloads, the pool's reuse of the panels) is behind both top-n streams
        PFZ_HIP(hipStreamWaitEvent(ctx->stream, consumed[0], 0));
        if (pi >= 2) PFZ_HIP(hipStreamWaitEvent(ctx->stream, consumed[1], 0));
    }
    PFZ_HIP(hipGetLastError());
    return PFZ_OK;     // (the score panels go back to the pool: stream order keeps them alive until the kernels are done)
}

static int dense_topn_host(pfz_ctx *ctx, const float *from_vec, int64_t n_from, const float *to_vec, int64_t n_to,
                           int64_t dim, int32_t ntop, float lower_bound, int32_t exclude_diag, int32_t normalize,
                           int32_t *out_idx, float *out_val)
{
    PFZ_REQUIRE(ctx && out_idx && out_val, "pfz_dense_cossim_topn_host: NULL argument");
    PFZ_REQUIRE(n_from >= 0 && n_to >= 0 && dim >= 1, "pfz_dense_cossim_topn_host: bad shape");
    if (n_from == 0) return PFZ_OK;
    PFZ_REQUIRE(from_vec && (n_to == 0 || to_vec), "pfz_dense_cossim_topn_host: NULL matrix");
    const bool same = to_vec == from_vec && n_to == n_from;
    pfz_dense *A = nullptr, *B = nullptr;
    pfz_topn *res = nullptr;
    int rc = pfz_dense_upload(ctx, from_vec, n_from, dim, normalize, &A);
    if (rc == PFZ_OK && !same) rc = pfz_dense_upload(ctx, to_vec, n_to, dim, normalize, &B);
    if (rc == PFZ_OK) rc = pfz_topn_alloc(ctx, n_from, ntop, &res);
    if (rc == PFZ_OK) rc = pfz_dense_topn(ctx, A, same ? A : B, ntop, lower_bound, exclude_diag, 0, res);
    if (rc == PFZ_OK) rc = pfz_topn_download(ctx, res, out_idx, out_val);
    pfz_topn_free(res);
    pfz_dense_free(B);
    pfz_dense_free(A);
    return rc;
}

int pfz_dense_cossim_topn_host(pfz_ctx *ctx, const float *from_vec, int64_t n_from, const float *to_vec, int64_t n_to,
                               int64_t dim, int32_t ntop, float lower_bound, int32_t exclude_diag, int32_t *out_idx,
                               float *out_val)
{
    return dense_topn_host(ctx, from_vec, n_from, to_vec, n_to, dim, ntop, lower_bound, exclude_diag, 1, out_idx, out_val);
}

int pfz_dense_dot_topn_host(pfz_ctx *ctx, const float *from_vec, int64_t n_from, const float *to_vec, int64_t n_to,
                            int64_t dim, int32_t ntop, float lower_bound, int32_t exclude_diag, int32_t *out_idx,
                            float *out_val)
{
    return dense_topn_host(ctx, from_vec, n_from, to_vec, n_to, dim, ntop, lower_bound, exclude_diag, 0, out_idx, out_val);
}

}  // extern "C"
