// Entry points declared in include/polyfuzz_hip.h whose kernels are not built
// yet.  They fail loudly (PFZ_ERR_UNSUPPORTED) -- there is no CPU fallback.
#include "pfz_internal.h"

#define PFZ_NOT_YET(name)                                               \
    do {                                                                \
        pfz::set_error(name ": not implemented in this build");         \
        return PFZ_ERR_UNSUPPORTED;                                     \
    } while (0)

extern "C" {

int pfz_strings_upload(pfz_ctx *, const void *, const int64_t *, int64_t, int32_t, pfz_strings **) { PFZ_NOT_YET("pfz_strings_upload"); }
void pfz_strings_free(pfz_strings *) {}
int pfz_tfidf_fit(pfz_ctx *, const pfz_tfidf_params *, const pfz_strings *, const pfz_strings *, pfz_tfidf **) { PFZ_NOT_YET("pfz_tfidf_fit"); }
void pfz_tfidf_free(pfz_tfidf *) {}
int pfz_tfidf_info(const pfz_tfidf *, int64_t *, int64_t *, int32_t *) { PFZ_NOT_YET("pfz_tfidf_info"); }
int pfz_tfidf_export(pfz_ctx *, const pfz_tfidf *, uint64_t *, double *, int64_t *) { PFZ_NOT_YET("pfz_tfidf_export"); }
int pfz_tfidf_import(pfz_ctx *, const pfz_tfidf_params *, int64_t, int64_t, int32_t, const uint64_t *, const double *, pfz_tfidf **) { PFZ_NOT_YET("pfz_tfidf_import"); }
int pfz_tfidf_transform(pfz_ctx *, const pfz_tfidf *, const pfz_strings *, pfz_csr **) { PFZ_NOT_YET("pfz_tfidf_transform"); }
int pfz_indel_argmax(pfz_ctx *, const pfz_strings *, const pfz_strings *, int32_t, int64_t, int64_t, int32_t *, double *) { PFZ_NOT_YET("pfz_indel_argmax"); }
int pfz_indel_matrix_host(pfz_ctx *, const pfz_strings *, const pfz_strings *, int64_t, int64_t, float *) { PFZ_NOT_YET("pfz_indel_matrix_host"); }
int pfz_dense_cossim_topn_host(pfz_ctx *, const float *, int64_t, const float *, int64_t, int64_t, int32_t, float, int32_t, int32_t *, float *) { PFZ_NOT_YET("pfz_dense_cossim_topn_host"); }
int pfz_comm_unique_id(uint8_t *) { PFZ_NOT_YET("pfz_comm_unique_id"); }
int pfz_comm_init(pfz_ctx *, const uint8_t *, int32_t, int32_t, pfz_comm **) { PFZ_NOT_YET("pfz_comm_init"); }
void pfz_comm_destroy(pfz_comm *) {}
int pfz_comm_allgather_topn(pfz_comm *, const pfz_topn *, pfz_topn *) { PFZ_NOT_YET("pfz_comm_allgather_topn"); }
int pfz_comm_barrier(pfz_comm *) { PFZ_NOT_YET("pfz_comm_barrier"); }

}  // extern "C"
