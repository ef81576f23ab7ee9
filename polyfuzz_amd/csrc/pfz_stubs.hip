// Entry points declared in include/polyfuzz_hip.h whose kernels are not built
// yet.  They fail loudly (PFZ_ERR_UNSUPPORTED) -- there is no CPU fallback.
#include "pfz_internal.h"

#define PFZ_NOT_YET(name)                                               \
    do {                                                                \
        pfz::set_error(name ": not implemented in this build");         \
        return PFZ_ERR_UNSUPPORTED;                                     \
    } while (0)

extern "C" {

int pfz_dense_cossim_topn_host(pfz_ctx *, const float *, int64_t, const float *, int64_t, int64_t, int32_t, float, int32_t, int32_t *, float *) { PFZ_NOT_YET("pfz_dense_cossim_topn_host"); }

}  // extern "C"
