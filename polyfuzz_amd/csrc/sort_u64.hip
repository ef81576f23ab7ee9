// Device radix sort of 64-bit n-gram codes (rocPRIM) for the sorted-vocabulary path of the vectoriser
// (k1_vectorize.hip: n-gram codes wider than the presence bitmap can address).  Kept in its own
// translation unit: the rocPRIM templates are the slowest thing in the build.
#include <cstring>
#include <string.h>

#include "pfz_internal.h"

#include <rocprim/rocprim.hpp>

namespace pfz {

// out[0..n) = in[0..n) sorted ascending on bits [0, end_bit); temporary storage from the context's scratch
int sort_codes_u64(pfz_ctx *ctx, const uint64_t *in, uint64_t *out, int64_t n, int end_bit)
{
    if (n <= 0) return PFZ_OK;
    size_t tmp_bytes = 0;
    PFZ_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, in, out, (size_t)n, 0u, (unsigned)end_bit, ctx->stream));
    PFZ_TRY(ensure_scratch(ctx, tmp_bytes));
    PFZ_HIP(rocprim::radix_sort_keys(ctx->scratch, tmp_bytes, in, out, (size_t)n, 0u, (unsigned)end_bit, ctx->stream));
    return PFZ_OK;
}

}  // namespace pfz
