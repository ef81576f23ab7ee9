// Internal definitions shared by the translation units of libpolyfuzz_hip.so.
// gfx950 (CDNA4, wave64) only -- no portability macros on purpose.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <atomic>

#include "polyfuzz_hip.h"

namespace pfz {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define PFZ_HIP(call)                                                      \
    do {                                                                   \
        hipError_t _e = (call);                                            \
        if (_e != hipSuccess) return pfz::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define PFZ_TRY(call)                 \
    do {                              \
        int _s = (call);              \
        if (_s != PFZ_OK) return _s;  \
    } while (0)

#define PFZ_REQUIRE(cond, ...)                    \
    do {                                          \
        if (!(cond)) {                            \
            pfz::set_error(__VA_ARGS__);          \
            return PFZ_ERR_INVALID;               \
        }                                         \
    } while (0)

constexpr int kWave = 64;
// LDS histograms with two 16-bit counters per word (index build, document frequencies): 144 KiB of the
// CU's 160 KiB, i.e. vocabularies of up to 73 728 n-grams; larger ones fall back to global atomics
constexpr int kHistWords = 36864;
constexpr int kEventSlots = 64;

// every CSR matrix gets a number of its own: "is this the matrix the index was built from?" (k3_symmetric.hip) must not be
// answered by pointers, which the caching allocator hands out again
inline uint64_t next_serial()
{
    static std::atomic<uint64_t> s{1};
    return s.fetch_add(1);
}
struct K3SymState;   // k3_symmetric.hip: the exchange buffers and the running session of a symmetric self-match

// A 32-bit scalar the device computes and the host wants LATER (a matrix' number of non-zeros, a vocabulary's size): the
// copy into a pinned slot and an event are enqueued where the value is ready on the device; the host waits for the event
// only when (if) it reads the value -- by then usually long past.  A hipStreamSynchronize at that point instead idles the
// device until the host has woken up and enqueued the next launch: 25 - 50 us per read-back, four of them in a 0.37-ms
// TF-IDF step of 10 000 x 10 000 names.
struct LazyI32 {
    hipEvent_t ev = nullptr;
    int32_t *slot = nullptr;     // pinned, owned by the context's slot pool
    bool pending = false;
    int32_t value = 0;
};

struct ProfEntry {
    std::vector<hipEvent_t> begin, end;  // recorded pairs not yet folded in
    double total_ms = 0.0;
    int64_t launches = 0;
};

}  // namespace pfz

struct pfz_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;        // side stream (K5: row top-n of one score panel beside the GEMM of the next), created on first use
    hipEvent_t side_events[4] = {};       // K5: panel ready x2, panel consumed x2
    hipStream_t stream3 = nullptr;        // K3's streamed self-match: the per-range merges beside the one pass-1 launch (k3_sym_launch_streamed)
    hipEvent_t ev3 = nullptr;             // ... "pass 1 is about to start" on the main stream
    hipStream_t stream3x[3] = {nullptr, nullptr, nullptr};      // ... more side streams: the ranges' chains (wait, merge, overflow pass) go round them
    hipEvent_t ev3x[3] = {nullptr, nullptr, nullptr};
    hipDeviceProp_t prop;
    hipEvent_t events[pfz::kEventSlots] = {};
    bool prof = false;
    int prof_level = 0;                   // 1: every profiled kernel, 2: the dominant kernels only
    std::map<std::string, pfz::ProfEntry> prof_entries;
    std::vector<hipEvent_t> event_pool;
    // reusable scratch (grown on demand, never inside a timed region after warm-up)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // pinned staging buffer of the host <-> device copies (copy_h2d / copy_d2h)
    char *stage = nullptr;
    size_t stage_bytes = 0, stage_off = 0;
    char *stage2 = nullptr;               // pinned staging of the side stream's downloads (pfz_topn_download_rows_after)
    size_t stage2_bytes = 0;
    // "event slot i has fired" as a word in pinned host memory, for events that fire on a side stream (K3's streamed self-match): the
    // host polls the word and enqueues its copy WITHOUT a stream dependency -- a copy that waits for another stream's event was
    // started a millisecond late by the runtime (measured, k3_sym_launch_streamed).  evt_want[i] = 0: slot i has no word, use the event
    int32_t *evt_flag = nullptr;          // [kEventSlots] pinned, written by the side stream's kernels (k3_sym_wait)
    int32_t evt_want[pfz::kEventSlots] = {};
    int32_t evt_serial = 0;
    struct RowsJob { const pfz_topn *t = nullptr; int64_t row_begin = 0, row_end = 0; int32_t slot = -1; bool issued = false; } rows_job[2];
    char *mirror = nullptr;               // pinned host mirror of a streamed match's result (pfz_cossim_topn_ranges)
    size_t mirror_bytes = 0;
    char *rows_stage[2] = {};             // pinned staging halves of pfz_topn_rows_begin / _finish
    size_t rows_bytes[2] = {}, rows_n[2] = {};
    hipEvent_t rows_ev[2] = {};
    // single-pass scan (exclusive_scan_i32): per-tile state words + the tile ticket, zeroed once; every call has its own epoch
    uint64_t *scan_state = nullptr;
    size_t scan_tiles = 0;
    uint32_t scan_epoch = 0;
    // LazyI32 slots: pinned words + events, recycled
    std::vector<int32_t *> lazy_slots;
    std::vector<hipEvent_t> lazy_events;
    std::vector<int32_t *> lazy_chunks;   // the hipHostMalloc'ed chunks the slots are cut from
    // caching allocator state: size class -> free blocks
    std::map<size_t, std::vector<void *>> pool_free_lists;
    size_t pool_cached_bytes = 0, pool_live_bytes = 0;
};

struct pfz_csr {
    pfz_ctx *ctx = nullptr;
    int64_t n_rows = 0, n_cols = 0;
    mutable int64_t nnz = 0;     // (device-vectorised matrices: resolved on first use, see nnz_lazy -- read it through csr_nnz())
    mutable pfz::LazyI32 nnz_lazy;
    int64_t nnz_cap = 0;         // entries indices / data have room for (>= nnz)
    int32_t *indptr = nullptr;   // [n_rows + 1], nnz < 2^31
    int32_t *indices = nullptr;  // [nnz]
    float *data = nullptr;       // [nnz]
    float max_norm = 1.f;        // upper bound of the rows' L2 norms (sizes K3's fixed-point scale)
    uint64_t serial = pfz::next_serial();   // unique per matrix; the contents never change after creation
};

// Inverted index of the to-side: for n-gram id k and to-row block b (block =
// block_cols consecutive to-rows) the postings (local_row, value), padded with
// zero-valued entries to whole 16-posting pieces, are pieces
// tab[k * n_blocks + b] .. tab[k * n_blocks + b + 1) of post; piece 0 is all zero.
struct pfz_index {
    pfz_ctx *ctx = nullptr;
    int64_t n_rows = 0, n_cols = 0;
    int32_t block_cols = 0, n_blocks = 0;
    // Pieces are numbered from 1: piece 0 is the all-zero dummy every kernel pads its rounds with.  The number of pieces and of
    // postings are known on the DEVICE when the build is enqueued; the host learns them when somebody asks (index_ready()).
    mutable int32_t n_pieces = 0;       // real pieces (after index_ready())
    mutable int64_t nnz = 0;            // postings (after index_ready())
    mutable pfz::LazyI32 pieces_lazy, nnz_lazy;
    int64_t piece_cap = 0;              // pieces `post` / `pblk` have room for (dummy included): a bound, see pfz_index_build
    int32_t *tab_base = nullptr;        // the allocation behind tab
    int32_t *tab = nullptr;  // = tab_base + 1: [n_cols * n_blocks + 1] first piece of every list; tab[n_cols * n_blocks] = pieces incl. the dummy
    int2 *post = nullptr;    // [piece_cap * 16]  .x = 4 * (to-row - b*block_cols), .y = fp32 bits
    uint16_t *pblk = nullptr; // [piece_cap]  the to-block of every piece (k3_symmetric.hip re-deals a block's rows to its accumulator slots)
    float max_norm = 1.f;    // of the indexed matrix' rows
    uint64_t src_serial = 0; // pfz_csr::serial of the matrix it was built from
    mutable pfz::K3SymState *sym = nullptr;   // k3_symmetric.hip, allocated by the first symmetric self-match on this index
};

struct pfz_topn {
    pfz_ctx *ctx = nullptr;
    int64_t n_rows = 0;
    int32_t ntop = 0;
    int32_t *idx = nullptr;
    float *val = nullptr;
};

struct pfz_strings {
    pfz_ctx *ctx = nullptr;
    int64_t n = 0, n_units = 0;
    int32_t char_width = 1;      // bytes per code unit: 1 or 4
    void *chars = nullptr;       // device [n_units]
    bool chars_in_offsets = false;   // small lists: chars points into the block of `offsets` (one upload)
    int64_t *offsets = nullptr;  // device [n+1], in code units
    int64_t max_len = 0;
    // host mirror of the offsets: string lengths for the host-side planning of K4
    std::vector<int64_t> h_off;      // n + 1
    // K4's to-side plan (alphabet, length-sorted packed groups), built on first use as a to-list
    struct pfz_indel_plan *indel_plan = nullptr;
    // K7: the list's token forms (a property of the list alone) and, built on first use as a to-list, its plan
    struct pfz_fuzz_forms *fuzz_forms = nullptr;
    struct pfz_fuzz_plan *fuzz_plan = nullptr;
    // n-gram cache of the vectoriser (see k1_vectorize.hip): per-string slot
    // ranges holding first the packed n-gram codes, then (column id, tf) pairs
    uint64_t cache_gen = 0;      // pfz_tfidf::gen the cache was made for (0 = none)
    uint64_t *slots = nullptr;   // device [n_units * R]
    size_t slots_cap = 0;        // in uint64 entries
    int32_t *row_cnt = nullptr;  // device [n + 1]: n-grams, then distinct ids per string
};

void pfz_indel_plan_free(struct pfz_indel_plan *p);   // k4_indel.hip
void pfz_fuzz_forms_free(struct pfz_fuzz_forms *f);   // k7_fuzz.hip
void pfz_fuzz_plan_free(struct pfz_fuzz_plan *p);

struct pfz_tfidf {
    pfz_ctx *ctx = nullptr;
    pfz_tfidf_params params;
    uint64_t gen = 0;            // unique id of this fit (cache tag)
    int32_t bits_per_char = 0;   // w
    int32_t code_bits = 0;       // ngram_hi * w  (<= 36)
    // alphabet: code unit -> rank+1 (0 = not in alphabet).  clean=1 uses the
    // fixed [ 0-9a-z] alphabet and no table.
    uint32_t *alpha_map = nullptr;   // device [alpha_map_len]
    int64_t alpha_map_len = 0;
    std::vector<uint32_t> alphabet;  // host: sorted code points, rank r -> alphabet[r]
    // vocabulary as a presence bitmap over codes + rank prefix per 256-bit group
    uint32_t *bitmap = nullptr;      // device [2^code_bits / 32]
    int32_t *prefix = nullptr;       // device [n_groups + 1]
    int64_t n_groups = 0;
    int64_t vocab = 0, n_docs = 0;
    // codes wider than kBitmapMaxBits: the vocabulary is the sorted array of distinct codes instead
    uint64_t *vcodes = nullptr;      // device [vocab], ascending; NULL in bitmap mode
    int32_t *df = nullptr;           // device [vocab]
    double *idf = nullptr;           // device [vocab]
};

namespace pfz {

// Skip codes of the best-choice kernels (K4 / K7, `skip_idx[from-row]`, host): -1 = no choice is left out; s >= 0 = choice s is
// (a self-match leaves out the from-string's own first occurrence, reference _distance.py:93-96); s <= -2 = every choice up
// to and including -2 - s is -- the reference RapidFuzz matcher's shared, shrinking list (_rapidfuzz.py:103-104 with
// n_jobs = 1: when row i is scored the rows 0 .. i have been removed).  One call uses one of the two forms (-1 goes with
// either): decode_skip_codes() says which -- 0: "equal", the array as it is; 1: "up to", the array rewritten to the last
// choice left out (-1: none); -1: both forms in one array (refused) -- and the kernels test choice_left_out(orig, s, up_to).
inline int decode_skip_codes(std::vector<int32_t> &v)
{
    bool eq = false, le = false;
    for (int32_t x : v) {
        eq |= x >= 0;
        le |= x <= -2;
    }
    if (eq && le) return -1;
    if (!le) return 0;
    for (int32_t &x : v) x = x <= -2 ? -2 - x : -1;
    return 1;
}
__host__ __device__ inline bool choice_left_out(int orig, int s, int up_to) { return up_to ? orig <= s : orig == s; }

// Owns a half-built object until the entry point succeeds: an error return (PFZ_TRY / PFZ_HIP /
// PFZ_REQUIRE) releases it through its own free function instead of leaking the device buffers.
template <typename T, void (*Free)(T *)> struct Owner {
    T *p;
    explicit Owner(T *q) : p(q) {}
    ~Owner() { if (p) Free(p); }
    Owner(const Owner &) = delete;
    Owner &operator=(const Owner &) = delete;
    T *operator->() const { return p; }
    T *release() { T *q = p; p = nullptr; return q; }
};

int ensure_scratch(pfz_ctx *ctx, size_t bytes);
// LazyI32: enqueue the copy of *dev (+ event) on ctx->stream; read the value (waits for the event if it has not been read yet);
// give the slot back
int lazy_begin(pfz_ctx *ctx, LazyI32 *z, const int32_t *dev);
int lazy_acquire(pfz_ctx *ctx, LazyI32 *z);      // a slot a kernel writes itself (z->slot is device-visible) ...
int lazy_mark(pfz_ctx *ctx, LazyI32 *z);         // ... and the event behind that kernel
int lazy_get(pfz_ctx *ctx, LazyI32 *z, int32_t *out);
void lazy_release(pfz_ctx *ctx, LazyI32 *z);
// a matrix' number of non-zeros (waits for the vectoriser's count if nobody has asked before); < 0: a HIP error
int64_t csr_nnz(const pfz_csr *m);
// what the host has not been told about an index yet -- its numbers of pieces and postings -- is fetched (k3_cossim_topn.hip)
int index_ready(const pfz_index *ix);
// the context's side stream (ctx->stream2) and its four events (ctx->side_events), created on first use
int ensure_side_stream(pfz_ctx *ctx);
// event slot `slot` will be announced by a word in pinned host memory too: *flag / *value = the word and what a kernel behind the
// event's work has to store there (system scope)
int event_flag_next(pfz_ctx *ctx, int32_t slot, int32_t **flag, int32_t *value);
// Host <-> device copies of caller-owned (pageable) buffers through the context's pinned staging buffer.
// Handing a pageable pointer to hipMemcpy makes the runtime register those pages with the GPU driver; when
// the caller later frees the buffer (a numpy array, a Python bytes object) the unmap evicts and restores the
// process' GPU queues -- measured here as +20..30 ms on every second TFIDF.match() call.  Both enqueue on
// ctx->stream; copy_d2h blocks until the data is in `dst`, copy_h2d returns once `src` may be reused.
int copy_h2d(pfz_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int copy_d2h(pfz_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
// profile helpers: bracket a launch when ctx->prof is on
struct ProfScope {
    pfz_ctx *ctx;
    const char *name;
    hipStream_t st;
    hipEvent_t b = nullptr, e = nullptr;
    ProfScope(pfz_ctx *c, const char *n, hipStream_t on = nullptr);   // on: the stream the launch goes to (default ctx->stream)
    ~ProfScope();
};

// Caching device allocator of a context (pfz_api.hip).  hipMalloc/hipFree are
// slow and hipFree synchronises the device, so blocks are size-classed and
// recycled; every block is only ever touched by work on the owning context's
// stream, which makes stream order the only ordering that is needed.
int pool_alloc_raw(pfz_ctx *ctx, void **p, size_t bytes);
template <typename T> inline int pool_alloc(pfz_ctx *ctx, T **p, size_t bytes)
{
    return pool_alloc_raw(ctx, (void **)p, bytes);
}
void pool_free(void *p);           // returns the block to its context's cache
int pool_release(pfz_ctx *ctx);    // hipFree every cached block (blocks)

// RCCL helpers (pfz_comm.hip); all enqueue on the communicator's context stream
int comm_rank(const pfz_comm *c);
int comm_world(const pfz_comm *c);
int comm_allgather_bytes(pfz_comm *c, const void *send, void *recv, size_t bytes_per_rank);
int comm_allreduce_sum_i32(pfz_comm *c, int32_t *buf, size_t n);
int comm_allreduce_sum_i64(pfz_comm *c, int64_t *buf, size_t n);
int comm_agree(pfz_comm *c, bool mine, bool *all);      // host-side AND over the ranks (waits)
void comm_abort(pfz_comm *c);                           // a rank failed behind an agreement: the peers must not wait for it

// exclusive scan of n int32 counters in place, total written to in[n]
// (array must have n+1 slots).  Enqueues on ctx->stream.
// total != NULL: the grand total also starts its way to the host (written by the scan itself into the LazyI32's pinned word)
int exclusive_scan_i32(pfz_ctx *ctx, int32_t *data, int64_t n, LazyI32 *total = nullptr);

// ascending sort of n-gram codes (sort_u64.hip: bitonic network, LDS tiles + streaming passes);
// `out` must hold sort_codes_capacity(n) keys
// (first best index, float64 score)[n] -> a two-column result buffer (idx[r][0] = index, the score's bits in the row's two
// value lanes): what the sharded edit-distance jobs all-gather (k7_fuzz.hip)
int best_to_topn(pfz_ctx *ctx, const int32_t *d_idx, const double *d_score, int64_t n, pfz_topn *out);

int64_t sort_codes_capacity(int64_t n);
int sort_codes_u64(pfz_ctx *ctx, const uint64_t *in, uint64_t *out, int64_t n);

}  // namespace pfz
