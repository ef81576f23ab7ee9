// Internal definitions shared by the translation units of libpolyfuzz_hip.so.
// gfx950 (CDNA4, wave64) only -- no portability macros on purpose.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>

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
constexpr int kEventSlots = 64;

struct ProfEntry {
    std::vector<hipEvent_t> begin, end;  // recorded pairs not yet folded in
    double total_ms = 0.0;
    int64_t launches = 0;
};

}  // namespace pfz

struct pfz_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    hipEvent_t events[pfz::kEventSlots] = {};
    bool prof = false;
    std::map<std::string, pfz::ProfEntry> prof_entries;
    std::vector<hipEvent_t> event_pool;
    // reusable scratch (grown on demand, never inside a timed region after warm-up)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
};

struct pfz_csr {
    pfz_ctx *ctx = nullptr;
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int32_t *indptr = nullptr;   // [n_rows + 1], nnz < 2^31
    int32_t *indices = nullptr;  // [nnz]
    float *data = nullptr;       // [nnz]
};

// Inverted index of the to-side: for n-gram id k and to-row block b (block =
// block_cols consecutive to-rows) the postings (local_row, value) live at
// post[tab[k * n_blocks + b] .. tab[k * n_blocks + b + 1]).
struct pfz_index {
    pfz_ctx *ctx = nullptr;
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int32_t block_cols = 0, n_blocks = 0;
    int32_t *tab = nullptr;  // [n_cols * n_blocks + 1]
    int2 *post = nullptr;    // [nnz]  .x = to-row - b*block_cols, .y = fp32 bits
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
    int32_t char_width = 1;
    void *chars = nullptr;     // device
    int64_t *offsets = nullptr;  // device [n+1]
    int64_t max_len = 0;
    uint32_t max_code = 0;
    // host mirrors (small; used for planning)
    std::vector<int64_t> h_offsets;
};

namespace pfz {

int ensure_scratch(pfz_ctx *ctx, size_t bytes);
// profile helpers: bracket a launch when ctx->prof is on
struct ProfScope {
    pfz_ctx *ctx;
    const char *name;
    hipEvent_t b = nullptr, e = nullptr;
    ProfScope(pfz_ctx *c, const char *n);
    ~ProfScope();
};

// exclusive scan of n int32 counters in place, total written to in[n]
// (array must have n+1 slots).  Enqueues on ctx->stream.
int exclusive_scan_i32(pfz_ctx *ctx, int32_t *data, int64_t n);

}  // namespace pfz
