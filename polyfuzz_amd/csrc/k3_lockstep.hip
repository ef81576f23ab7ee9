// K3, to-block-major ("lock-step") form -- the same sparse cosine top-n as k3_cossim_topn.hip (reference
// polyfuzz/models/_utils.py:82-91, 128-146: awesome_cossim_topn + diagonal + per-row top-n), bit-identical results,
// for to-sides whose inverted index is far larger than the 8 x 4 MiB of L2.
//
// Why.  The main kernel gives every wave ONE from-row and walks all to-blocks with it; the waves of a chip are then
// spread over the whole index.  With 100 000 to-rows (21 MB of index) 72 % of the posting loads still hit L2; with
// 1 000 000 (210 MB, BASELINE config 4) 7 % do -- rocprofv3 FETCH_SIZE x 2 = 344 GB leave L2 per launch for 346 GB of
// algorithmic posting bytes, 7.4 TB/s of fabric traffic: the kernel is bound by the Infinity Fabric, not by LDS.
// The bytes are dominated by the heavy n-grams ("inc", "llc", "cor"): their lists are read by a third of all from-rows.
//
// What.  The loops are swapped.  Work item = (slice of S consecutive to-blocks, chunk of 32 consecutive from-rows), handed
// out slice-major: every wave of an XCD works on the same ~1 MB of index at the same time, so a posting line is fetched
// into each XCD's L2 once per slice instead of once per use.  A from-row's top-n state (threshold, <= 32 candidate
// keys) lives in HBM between its slices: 264 B read per (row, slice), written back only when the slice pushed a
// candidate (rare once the threshold is warm).
//   * Hand-out: 8 heads (one per XCD, HW_REG_XCC_ID; head h owns the chunks c = h mod 8), each a counter on a cache line
//     of its own that persistent one-wave workgroups pull from; an exhausted head's waves steal from the others, so any
//     placement finishes.  (A first version pulled 4 rows at a time from counters sharing one line and held a barrier
//     per slice: 13 x SLOWER than the row-major kernel -- 60 000 atomics and the spinning of 2 300 waves on one line
//     per slice.)
//   * Ordering: (chunk, slice s+1) must see what (chunk, slice s) wrote.  State is stored and loaded with agent-scope
//     atomics (sc1: no stale L1 / remote-L2 lines); a wave flags a chunk as through slice s only after its stores are
//     acknowledged (s_waitcnt vmcnt(0)), and whoever pulls (chunk, s+1) waits for that flag -- normally set long ago:
//     thousands of chunks are pulled in between.  Pulls are slice-major, so whoever is waited for is a resident wave that
//     waits for nothing later than itself: no deadlock, whatever the placement.
//   * Inside a chunk the rows are taken in windows of as many consecutive rows as fit the 64 lanes (lane = one CSR entry
//     of the window: its value and the first pieces of its n-gram's lists in the slice's blocks); the next window's loads
//     and the next row's keys are in flight while a row is scattered and swept.
//   * Everything inside a (row, block) -- scatter, sweep, candidate buffer, warm start, final compaction -- is k3_core.h,
//     shared with the main kernel; integer sums make the result independent of the order of the blocks.
#include "k3_core.h"

#include <stdlib.h>

namespace pfz {

constexpr int kLsCap = 96;            // candidate keys per wave (ntop <= 32), as in the main kernel
constexpr int kLsKeep = kLsCap - 64;  // a row's state holds at most this many keys between slices (push4 compacts above)
constexpr int kLsChunkMax = 32;       // from-rows per pull at most (one atomic, one flag, one load of the rows' pointers and states)
constexpr int kLsHeads = 8;
constexpr int kLsCtrStride = 32;      // uint32 words between two heads' pull counters: a 128-byte line each

struct K3LsArgs {
    const int32_t *a_indptr;
    const int32_t *a_idx;
    const float *a_val;
    int32_t n_a;
    const int32_t *tab;
    const int2 *post;
    int32_t nb, n_pieces, ntop, thr0;
    float scale, inv_scale;
    int32_t exclude_diag;
    int64_t diag_offset;
    int32_t *out_idx;
    float *out_val;
    uint32_t *ctr;          // [8 x 32] pull counter of head h at ctr[32 h]
    uint32_t *chunk_done;   // [n_chunks] slices this chunk has been through (and whose state is visible)
    uint64_t *st_meta;      // [n_a]      lo: cnt | warmed << 8, hi: thr
    uint64_t *st_keys;      // [n_a][32]
    int32_t n_slices, n_chunks, chunk_rows;
};

__device__ inline uint64_t ld_agent(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline void st_agent(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one window of a chunk: the rows jw .. jn-1, whose CSR entries [base, end) fit the 64 lanes (lane l = entry base + l);
// a row of more than 64 n-grams is a window of its own (its tail goes through the slow loop)
template <int S> struct LsWindow {
    int jw, jn, base;
    float as;          // the lane's entry: value x fixed-point scale
    int tabv[S + 1];   // ... and the first pieces of its n-gram's lists in the slice's blocks (+ the end of the last)
};

template <int S>
__device__ inline LsWindow<S> ls_window(const K3LsArgs &a, int ip, int jw, int nrows, int lane, int b0, int n_blk)
{
    LsWindow<S> w;
    w.jw = jw;
    w.jn = jw;
    w.base = 0;
    w.as = 0.f;
#pragma unroll
    for (int i = 0; i <= S; ++i) w.tabv[i] = 0;
    if (jw >= nrows) return w;
    w.base = __builtin_amdgcn_readlane(ip, jw);
    int jn = jw + 1;
    while (jn < nrows && __builtin_amdgcn_readlane(ip, jn + 1) - w.base <= 64) ++jn;
    w.jn = jn;
    if (w.base + lane < __builtin_amdgcn_readlane(ip, jn)) {
        w.as = a.a_val[w.base + lane] * a.scale;
        const int32_t *trow = a.tab + (int64_t)a.a_idx[w.base + lane] * a.nb + b0;
#pragma unroll
        for (int i = 0; i <= S; ++i)
            if (i <= n_blk) w.tabv[i] = trow[i];          // (tab has V*nb + 2 entries: trow[n_blk] exists)
    }
    return w;
}

template <int C, int S>
__global__ __launch_bounds__(64) void k3_lockstep_kernel(const K3LsArgs a)
{
    __shared__ __attribute__((aligned(16))) struct {
        int acc[C];
        uint64_t cand[kLsCap];
    } sm;
    int *const acc = sm.acc;
    uint64_t *const cand = sm.cand;
    int *const mark = (int *)(sm.cand + kLsCap) - 64;         // scatter scratch: the tail of the candidate buffer
    if ((uint32_t)(uintptr_t)sm.acc != 0u) __builtin_trap();  // layout assumption of run_steps()
    const int lane = threadIdx.x;
    int4 *acc4 = (int4 *)acc;
    constexpr int N4 = C / 4;
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    for (int t = lane; t < C / 4; t += 64) acc4[t] = make_int4(0, 0, 0, 0);
    wave_sync();
    const char *post_bytes = (const char *)a.post;
    const int src4 = (4 * (lane & 15) + (lane >> 4)) * 4;
    const int sub8 = (lane & 15) * 8;
    const int dummy_addr = 0;      // (piece 0 is the all-zero dummy)
    const int nb = a.nb, ntop = a.ntop;

    // hwreg(HW_REG_XCC_ID = 20, offset 0, 4 bits): which XCD this wave runs on (a speed hint only: any value works)
    int head = (int)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & (kLsHeads - 1));
    int heads_tried = 0;

    for (;;) {
        // ---- pull the next chunk: (head, slice, chunk) ---------------------------------------------------------------
        int slice = 0, chunk = 0;
        bool got = false;
        while (heads_tried < kLsHeads) {
            const int count_h = head < a.n_chunks ? (a.n_chunks - head + kLsHeads - 1) / kLsHeads : 0;
            const uint32_t items_h = (uint32_t)count_h * (uint32_t)a.n_slices;
            uint32_t t = 0;
            if (lane == 0 && items_h) t = atomicAdd(&a.ctr[head * kLsCtrStride], 1u);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            if (items_h && t < items_h) {
                slice = (int)(t / (uint32_t)count_h);
                chunk = (int)(t - (uint32_t)slice * (uint32_t)count_h) * kLsHeads + head;
                got = true;
                break;
            }
            head = (head + 1) & (kLsHeads - 1);     // this head is exhausted for good: move on (steal)
            ++heads_tried;
        }
        if (!got) break;
        const int r0 = chunk * a.chunk_rows;
        const int nrows = min(a.chunk_rows, a.n_a - r0);
        const bool last_slice = slice == a.n_slices - 1;
        const int b0 = slice * S;
        const int n_blk = min(S, nb - b0);

        // ---- this chunk's rows have been through the slices before this one, and their state is visible ------------
        // (pulls are slice-major: whoever holds (chunk, slice - 1) is a resident wave that waits for nothing later)
        if (slice > 0) {
            for (;;) {
                uint32_t d = 0;
                if (lane == 0) d = __hip_atomic_load(&a.chunk_done[chunk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                if (d >= (uint32_t)slice) break;
                __builtin_amdgcn_s_sleep(32);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (compiler order; the state loads below are sc1)
        }

        // ---- the chunk's rows: row pointers in lanes 0..nrows, their state words in lanes 0..nrows-1 ---------------
        const int ip = a.a_indptr[r0 + min(lane, nrows)];
        uint64_t meta = 0ull;
        if (slice > 0 && lane < nrows) meta = ld_agent(&a.st_meta[r0 + lane]);
        uint64_t key_next = 0ull;          // the keys of the next row to be processed, loaded one row ahead
        if (slice > 0 && lane < kLsKeep) key_next = ld_agent(&a.st_keys[(int64_t)r0 * kLsKeep + lane]);

        LsWindow<S> w = ls_window<S>(a, ip, 0, nrows, lane, b0, n_blk);
        while (w.jw < nrows) {
            const LsWindow<S> wn = ls_window<S>(a, ip, w.jn, nrows, lane, b0, n_blk);     // the next window's loads, now

            for (int j = w.jw; j < w.jn; ++j) {
                const int row = r0 + j;
                const int p0 = __builtin_amdgcn_readlane(ip, j), p1 = __builtin_amdgcn_readlane(ip, j + 1);
                const int nnz = p1 - p0;
                const uint64_t keyv = key_next;
                if (slice > 0 && j + 1 < nrows && lane < kLsKeep) key_next = ld_agent(&a.st_keys[(int64_t)(row + 1) * kLsKeep + lane]);
                if (nnz == 0) {
                    if (last_slice)
                        for (int r = lane; r < ntop; r += 64) {
                            a.out_idx[(int64_t)row * ntop + r] = -1;
                            a.out_val[(int64_t)row * ntop + r] = 0.f;
                        }
                    continue;
                }
                const int lo = p0 - w.base;
                const bool in_row = lane >= lo && lane < lo + nnz;      // (lanes are < 64: a long row keeps its first 64)
                const int64_t self_col64 = (int64_t)row + a.diag_offset;
                const int self_col = (a.exclude_diag && self_col64 >= 0 && self_col64 < 0x7fffffff) ? (int)self_col64 : -1;

                TopState st;
                st.cnt = 0;
                st.thr = a.thr0;
                st.pushed = 0;
                bool warmed = false;
                if (slice > 0) {
                    const uint32_t m_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)meta, j);
                    const uint32_t m_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(meta >> 32), j);
                    st.cnt = (int)(m_lo & 0xffu);
                    warmed = (m_lo >> 8) & 1u;
                    st.thr = (int)m_hi;
                }
                const bool warmed_in = warmed;
                bool restored = slice == 0;

                for (int i = 0; i < n_blk; ++i) {
                    const int b = b0 + i;
                    int s_i = 0, e_i = 0;
#pragma unroll
                    for (int q = 0; q < S; ++q)
                        if (q == i) {
                            s_i = w.tabv[q];
                            e_i = w.tabv[q + 1];
                        }
                    const int np = in_row ? e_i - s_i : 0;
                    bool touched = __ballot(np > 0) != 0;
                    if (touched) scatter_pieces(acc, post_bytes, mark, np, s_i, w.as, lane, src4, sub8, dummy_addr);
                    for (int c0 = p0 + 64; c0 < p1; c0 += 64) {      // rows with more than 64 n-grams
                        int s2 = 0, e2 = 0;
                        float as2 = 0.f;
                        if (c0 + lane < p1) {
                            const int k = a.a_idx[c0 + lane];
                            as2 = a.a_val[c0 + lane] * a.scale;
                            s2 = a.tab[(int64_t)k * nb + b];
                            e2 = a.tab[(int64_t)k * nb + b + 1];
                        }
                        if (__ballot(e2 > s2)) {
                            touched = true;
                            scatter_pieces(acc, post_bytes, mark, e2 - s2, s2, as2, lane, src4, sub8, dummy_addr);
                        }
                    }
                    if (touched) {
                        wave_sync();
                        if (!restored) {
                            restored = true;
                            if (lane < st.cnt) cand[lane] = keyv;
                            wave_sync();
                        }
                        if (!warmed) {
                            warmed = true;
                            if (ntop <= kWarmMaxTop) {
                                const int t = warm_threshold<N4>(acc4, 0, ntop + (self_col >= 0 ? 1 : 0), lane);
                                st.thr = t > st.thr ? t : st.thr;
                            }
                        }
                        sweep_block<N4, kLsCap>(acc4, cand, st, b * C, self_col, ntop, lane, zero);
                        wave_sync();
                    }
                }

                if (last_slice) {
                    if (!restored) {
                        if (lane < st.cnt) cand[lane] = keyv;
                        wave_sync();
                    }
                    compact<kLsCap>(cand, st, ntop, lane);
                    for (int r = lane; r < ntop; r += 64) {
                        const uint64_t key = r < st.cnt ? cand[r] : 0ull;
                        a.out_idx[(int64_t)row * ntop + r] = key ? (int32_t)(~(uint32_t)key) : -1;
                        a.out_val[(int64_t)row * ntop + r] = key ? (float)(int32_t)(uint32_t)(key >> 32) * a.inv_scale : 0.f;
                    }
                } else if (st.pushed || warmed != warmed_in || slice == 0) {
                    // the state changed (or is written for the first time): keys and the word that describes them.  Both are
                    // acknowledged before the chunk is flagged as through this slice, and nobody reads them earlier.
                    if (lane < st.cnt) st_agent(&a.st_keys[(int64_t)row * kLsKeep + lane], cand[lane]);
                    if (lane == 0)
                        st_agent(&a.st_meta[row], ((uint64_t)(uint32_t)st.thr << 32) | (uint32_t)(st.cnt | ((int)warmed << 8)));
                }
                wave_sync();     // cand is reused by the next row
            }
            w = wn;
        }

        // ---- through this slice: every store of the chunk acknowledged, then the flag --------------------------------
        if (!last_slice) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&a.chunk_done[chunk], (uint32_t)(slice + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static int ls_env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

bool k3_lockstep_wanted(const pfz_ctx *ctx, const pfz_index *ix, int64_t n_rows, int32_t ntop)
{
    if (ntop > kLsKeep || ix->n_blocks < 2) return false;
    if (ix->block_cols != 2048 && ix->block_cols != 4096) return false;
    const int force = ls_env_int("PFZ_K3_LOCKSTEP", -1);    // 1: whenever possible (tests), 0: never
    if (force >= 0) return force != 0;
    // auto: the index no longer fits the L2s by a wide margin and there are enough from-rows for the slices to stay in
    // step (1M to-rows: 40 000 from-rows 10.9 ms against 15.8 row-major, 125 000: 33.3 against 46.9; 125 000 from-rows against
    // 500k / 300k / 200k to-rows: 17.2 / 10.8 / 7.7 ms against 23.7 / 12.6 / 7.9)
    return ix->n_rows > ls_env_int("PFZ_K3_LS_MIN_TO", 250000) && n_rows >= 16384;
}

int k3_lockstep_launch(pfz_ctx *ctx, const pfz_index *ix, const pfz_csr *A, int64_t row_begin, int64_t n_rows, int32_t ntop,
                       int32_t thr0, float scale, float inv_scale, int32_t exclude_diag, int64_t diag_offset, pfz_topn *out)
{
    // to-blocks per item: a slice of ~16 000 to-rows (3 - 4 MB of index) still lives in an XCD's L2 and halves the state
    // round trips of narrower ones.  125k x 1M, top-10, K3 alone, 32 rows per pull: row-major 46.9 ms; lock-step 4096-row
    // blocks S = 1 / 2 / 4: 43.9 / 41.4 / 40.3 ms, 2048-row blocks S = 2 / 4 / 8: 41.3 / 39.5 / 38.9 ms; with 8 rows per pull
    // (18 waves per CU want more chunks than 125 000 / 32) 2048-row blocks, S = 8: 33.3 ms
    int S = ls_env_int("PFZ_K3_LS_BLOCKS", ix->block_cols == 4096 ? 4 : 8);
    if (S != 1 && S != 2 && S != 4 && S != 8) S = 1;
    const int n_slices = (ix->n_blocks + S - 1) / S;
    // persistent one-wave workgroups: as many as the LDS lets a CU hold (8 KiB / 16 KiB of accumulators + 768 B)
    const int lds = ix->block_cols * 4 + kLsCap * 8;
    const int per_cu = ls_env_int("PFZ_K3_LS_WAVES", (160 * 1024) / lds);
    int64_t grid = (int64_t)ctx->prop.multiProcessorCount * (per_cu < 1 ? 1 : per_cu);
    // rows per pull: 32 when that still leaves every wave ~8 chunks per slice (the waves take chunks as they finish: the
    // slowest wave of a slice is one chunk behind), fewer for shorter from-lists
    int chunk_rows = ls_env_int("PFZ_K3_LS_CHUNK", 0);
    if (chunk_rows <= 0) {
        chunk_rows = kLsChunkMax;
        while (chunk_rows > 4 && n_rows / chunk_rows < 8 * grid) chunk_rows >>= 1;
    }
    chunk_rows = chunk_rows > kLsChunkMax ? kLsChunkMax : (chunk_rows < 1 ? 1 : chunk_rows);
    const int64_t n_chunks = (n_rows + chunk_rows - 1) / chunk_rows;
    if (n_chunks * (int64_t)n_slices / kLsHeads >= ((int64_t)1 << 32) - 1 || n_chunks >= ((int64_t)1 << 28)) {
        set_error("pfz_cossim_topn (lock-step): %lld chunks x %d slices exceed the 32-bit pull counters", (long long)n_chunks,
                  n_slices);
        return PFZ_ERR_UNSUPPORTED;
    }
    // scratch: [8 pull counters, a line each][n_chunks flags][n_rows state words][n_rows x 32 keys]
    const size_t ctr_bytes = (size_t)kLsHeads * kLsCtrStride * 4 + (((size_t)n_chunks * 4 + 255) & ~(size_t)255);
    const size_t meta_bytes = (size_t)n_rows * 8, key_bytes = (size_t)n_rows * kLsKeep * 8;
    PFZ_TRY(ensure_scratch(ctx, ctr_bytes + meta_bytes + key_bytes));
    char *base = (char *)ctx->scratch;
    PFZ_HIP(hipMemsetAsync(base, 0, ctr_bytes, ctx->stream));
    K3LsArgs a;
    a.a_indptr = A->indptr + row_begin;
    a.a_idx = A->indices;
    a.a_val = A->data;
    a.n_a = (int32_t)n_rows;
    a.tab = ix->tab;
    a.post = ix->post;
    a.nb = ix->n_blocks;
    a.n_pieces = 0;                // (not needed any more: the dummy piece is piece 0)
    a.ntop = ntop;
    a.thr0 = thr0;
    a.scale = scale;
    a.inv_scale = inv_scale;
    a.exclude_diag = exclude_diag;
    a.diag_offset = diag_offset + row_begin;
    a.out_idx = out->idx + row_begin * ntop;
    a.out_val = out->val + row_begin * ntop;
    a.ctr = (uint32_t *)base;
    a.chunk_done = (uint32_t *)base + kLsHeads * kLsCtrStride;
    a.st_meta = (uint64_t *)(base + ctr_bytes);
    a.st_keys = (uint64_t *)(base + ctr_bytes + meta_bytes);
    a.n_slices = n_slices;
    a.n_chunks = (int32_t)n_chunks;
    a.chunk_rows = chunk_rows;
    if (grid > n_chunks) grid = n_chunks;
#define PFZ_K3_LS(CC, SS) hipLaunchKernelGGL((k3_lockstep_kernel<CC, SS>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a)
    if (ix->block_cols == 4096) {
        if (S == 1) PFZ_K3_LS(4096, 1);
        else if (S == 2) PFZ_K3_LS(4096, 2);
        else if (S == 4) PFZ_K3_LS(4096, 4);
        else PFZ_K3_LS(4096, 8);
    } else {
        if (S == 1) PFZ_K3_LS(2048, 1);
        else if (S == 2) PFZ_K3_LS(2048, 2);
        else if (S == 4) PFZ_K3_LS(2048, 4);
        else PFZ_K3_LS(2048, 8);
    }
#undef PFZ_K3_LS
    PFZ_HIP(hipGetLastError());
    return PFZ_OK;
}

}  // namespace pfz
