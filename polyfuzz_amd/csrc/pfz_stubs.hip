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
int pfz_comm_unique_id(uint8_t *) { PFZ_NOT_YET("pfz_comm_unique_id"); }
int pfz_comm_init(pfz_ctx *, const uint8_t *, int32_t, int32_t, pfz_comm **) { PFZ_NOT_YET("pfz_comm_init"); }
void pfz_comm_destroy(pfz_comm *) {}
int pfz_comm_allgather_topn(pfz_comm *, const pfz_topn *, pfz_topn *) { PFZ_NOT_YET("pfz_comm_allgather_topn"); }
int pfz_comm_barrier(pfz_comm *) { PFZ_NOT_YET("pfz_comm_barrier"); }

}  // extern "C"
