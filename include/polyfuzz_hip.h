/*
 * polyfuzz_hip.h -- C ABI of libpolyfuzz_hip.so, the MI355X (gfx950) engine
 * behind PolyFuzz's TFIDF / EditDistance matchers.
 *
 * The reference (MaartenGr/PolyFuzz v0.4.3) is pure Python and has no FFI of
 * its own; the entry points below are what a binding for its hot path would
 * bind.  Each one cites the reference interface it replaces (file:line under
 * the reference tree).  Conventions:
 *   - plain pointers and sizes only; no torch / numpy types;
 *   - every function returns 0 on success or a negative pfz_status; the
 *     message of the last failure on the calling thread is pfz_last_error();
 *   - host ("_host" / upload / download) entry points take caller-owned host
 *     buffers; every other buffer is device memory owned by an opaque handle;
 *   - all device work of a context is issued on that context's own HIP stream;
 *     calls return after enqueueing unless documented otherwise
 *     (downloads and pfz_ctx_sync block);
 *   - "no match" is idx = -1, score = 0.
 * There is NO CPU fallback anywhere behind this ABI: without a gfx950 device
 * pfz_ctx_create fails with PFZ_ERR_NO_DEVICE.
 */
#ifndef POLYFUZZ_HIP_H
#define POLYFUZZ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFZ_VERSION 100 /* 0.1.0 */

typedef enum pfz_status {
    PFZ_OK = 0,
    PFZ_ERR_INVALID = -1,     /* bad argument */
    PFZ_ERR_NO_DEVICE = -2,   /* no usable HIP device */
    PFZ_ERR_HIP = -3,         /* a HIP runtime call failed (see pfz_last_error) */
    PFZ_ERR_UNSUPPORTED = -4, /* valid request outside what the kernels cover */
    PFZ_ERR_NOMEM = -5,
    PFZ_ERR_RCCL = -6
} pfz_status;

typedef struct pfz_ctx pfz_ctx;       /* one device + one stream + scratch */
typedef struct pfz_csr pfz_csr;       /* device-resident CSR matrix (fp32 values, sorted indices) */
typedef struct pfz_index pfz_index;   /* device-resident inverted index of a to-side CSR */
typedef struct pfz_topn pfz_topn;     /* device-resident (idx int32, score fp32)[n_rows][ntop] */
typedef struct pfz_strings pfz_strings; /* device-resident packed string list */
typedef struct pfz_tfidf pfz_tfidf;   /* fitted vectoriser: vocabulary + idf */
typedef struct pfz_comm pfz_comm;     /* RCCL communicator bound to a context (one process per GPU) */

/* ---- library / context -------------------------------------------------- */
int pfz_version(void);
const char *pfz_last_error(void);
/* number of HIP devices, 0 if none (never an error) */
int pfz_device_count(void);
int pfz_ctx_create(int device, pfz_ctx **out);
void pfz_ctx_destroy(pfz_ctx *ctx);
int pfz_ctx_sync(pfz_ctx *ctx);
/* device properties of the context: name (<=255 chars), CU count, HBM bytes */
int pfz_ctx_info(pfz_ctx *ctx, char *name256, int32_t *n_cu, int64_t *hbm_bytes);

/* HIP-event timers on the context's stream (used by bench.py for the live
 * per-kernel timing the roofline is computed from).  slot in [0, 64). */
int pfz_event_record(pfz_ctx *ctx, int32_t slot);
/* blocks until both events completed */
int pfz_event_elapsed_ms(pfz_ctx *ctx, int32_t slot_begin, int32_t slot_end, float *ms);
/* `bytes` of the context's PINNED staging buffer for the caller to fill (*host; NULL when the request is too large for it): what is
 * handed to pfz_strings_upload from there is read by the DMA where it lies -- the host's string packer (the lists of reference
 * _base.py:13-16, `match(from_list, to_list)`) writes offsets and code units straight into it instead of into a buffer that is then
 * copied into it.  Valid until the next upload or download on this context. */
int pfz_stage_reserve(pfz_ctx *ctx, int64_t bytes, void **host);
/* block until event slot `slot` has fired (recorded by pfz_event_record, or by pfz_cossim_topn_ranges for its row ranges -- those
 * are also announced through a word in pinned host memory, which this call polls instead of sleeping on the runtime's event) */
int pfz_event_wait(pfz_ctx *ctx, int32_t slot);
/* per-kernel profile: when enabled, every launch of the named hot kernels is
 * bracketed by its own event pair (adds a little launch overhead: ~2 % of a 100k x 100k step).
 * on = 1: every profiled kernel; on = 2: only the dominant ones (k3_cossim_topn, k4_indel, k5_gemm_panel). */
int pfz_prof_enable(pfz_ctx *ctx, int32_t on);
int pfz_prof_reset(pfz_ctx *ctx);
/* total ms and launch count of kernel `name` since the last reset (blocks). */
int pfz_prof_get(pfz_ctx *ctx, const char *name, double *total_ms, int64_t *launches);

/* ---- CSR matrices --------------------------------------------------------
 * The layout scipy.sparse.csr_matrix uses for TfidfVectorizer output
 * (reference _tfidf.py:110-111): indptr[n_rows+1], indices[nnz] sorted per
 * row, data[nnz].  Values are converted to fp32 by the caller. */
int pfz_csr_upload(pfz_ctx *ctx, int64_t n_rows, int64_t n_cols,
                   const int64_t *indptr, const int32_t *indices, const float *data,
                   pfz_csr **out);
int pfz_csr_shape(const pfz_csr *m, int64_t *n_rows, int64_t *n_cols, int64_t *nnz);
/* blocks; any of the three output pointers may be NULL */
int pfz_csr_download(pfz_ctx *ctx, const pfz_csr *m, int64_t *indptr, int32_t *indices, float *data);
void pfz_csr_free(pfz_csr *m);

/* ---- K3: sparse cosine top-n ---------------------------------------------
 * Replaces sparse_dot_topn.awesome_cossim_topn(from, to.T, top_n+1, min_sim)
 * as called at reference polyfuzz/models/_utils.py:82, fused with the
 * diagonal removal of _utils.py:84-87 and the per-row top-n extraction of
 * _utils.py:89-91,128-146.
 *
 * pfz_index_build: device-side inverted index (n-gram id -> postings, split
 * by to-row block) of the to-side matrix; built once per fit and kept
 * resident (reference keeps self.tf_idf_to, _tfidf.py:110).  */
int pfz_index_build(pfz_ctx *ctx, const pfz_csr *to_matrix, pfz_index **out);
void pfz_index_free(pfz_index *ix);
/* bytes of postings + offset table (for the bench's byte accounting) */
int pfz_index_info(const pfz_index *ix, int64_t *n_rows, int64_t *n_cols, int64_t *nnz,
                   int64_t *block_cols, int64_t *n_blocks, int64_t *table_bytes);
/* the index stores every (n-gram, to-block) posting list padded to whole pieces of `piece_postings`
 * (16) postings = one 128-byte line each: number of pieces (index bytes = 128 * (n_pieces + 1)) */
int pfz_index_pieces(const pfz_index *ix, int64_t *n_pieces, int64_t *piece_postings);
/* how often pfz_cossim_topn[_rows] on this index took the symmetric form (a list against itself, reference
 * _tfidf.py:113-116 -> _utils.py:82-87: every unordered pair of rows scored once, csrc/k3_symmetric.hip), and how many
 * from-rows those launches covered -- the bench prices its roofline by the kernel that ran */
int pfz_index_symmetric_launches(const pfz_index *ix, int64_t *launches, int64_t *rows);
/* census of the LAST symmetric launch on this index (waits for the stream): rows taken out of the hand-over because of
 * their low own-block threshold ("magnets": they fetch their matches below their own block themselves) and rows that were
 * sent more candidates than their push slots hold and were recomputed row-major -- what the tests of those two paths assert
 * (nothing to replace in the reference: the exchange only exists because every unordered pair is scored once) */
int pfz_index_symmetric_census(const pfz_index *ix, int64_t *magnet_rows, int64_t *recomputed_rows);

int pfz_topn_alloc(pfz_ctx *ctx, int64_t n_rows, int32_t ntop, pfz_topn **out);
void pfz_topn_free(pfz_topn *t);
/* set every entry to "no match" (idx -1, score 0); enqueues */
int pfz_topn_clear(pfz_ctx *ctx, pfz_topn *t);
/* blocks; out_idx / out_val are [n_rows * ntop] row-major host buffers */
int pfz_topn_download(pfz_ctx *ctx, const pfz_topn *t, int32_t *out_idx, float *out_val);
/* download rows [row_begin, row_end) as soon as the event recorded in `event_slot` (pfz_event_record) has fired: the
 * copy goes through a side stream, so work enqueued on the context stream after the event is not waited for.  Blocks. */
int pfz_topn_download_rows_after(pfz_ctx *ctx, const pfz_topn *t, int64_t row_begin, int64_t row_end,
                                 int32_t event_slot, int32_t *out_idx, float *out_val);
/* The same in two halves, without the copy into caller buffers: _begin enqueues the side stream's wait for the event and the two
 * device-to-host copies into pinned staging half `half` (0 / 1) and returns; _finish waits for them and hands out the pinned
 * pointers (valid until the next _begin on that half).  A caller that consumes range i while range i + 1 is already on its way
 * (the frame's column fills of reference _utils.py:104-125) alternates the halves. */
int pfz_topn_rows_begin(pfz_ctx *ctx, const pfz_topn *t, int64_t row_begin, int64_t row_end, int32_t event_slot, int32_t half);
int pfz_topn_rows_finish(pfz_ctx *ctx, int32_t half, const int32_t **idx, const float **val);
/* fill a result buffer from host arrays [n_rows * ntop] (results computed elsewhere, e.g. another rank's block) */
int pfz_topn_upload(pfz_ctx *ctx, pfz_topn *t, const int32_t *idx, const float *val);
/* raw device pointers (for an RCCL all-gather issued by the caller) */
int pfz_topn_device_ptrs(const pfz_topn *t, void **idx_dev, void **val_dev, int64_t *n_rows, int32_t *ntop);

/* For every from-row i: the ntop largest cosine scores C[i][j] = sum_k
 * A[i][k] * B[j][k] with C[i][j] > lower_bound (strict, as sparse_dot_topn),
 * ordered by (score desc, j asc).  Every product is computed in fp32, scaled by
 * 2^30 / (norm bound) and truncated to int32; the sums are int32 (order-
 * independent, bit-reproducible; |error| ~ 3e-8).  exclude_diag != 0 drops j == i + diag_offset (self-match,
 * _utils.py:84-87; diag_offset = global index of from-row 0 when the from
 * side is a row shard).  lower_bound < 0 is treated as 0 (non-positive scores
 * are "no match" in the reference's output contract, _utils.py:122-123).
 * When from_matrix IS the matrix the index was built from, exclude_diag != 0 and diag_offset == 0 -- a list against
 * itself -- lists of 20 480 to 250 000 rows (ntop <= 32) take the symmetric form: C is symmetric bit for bit in this
 * arithmetic, so every unordered pair of rows is scored once and handed to both rows; same results (PFZ_K3_SYM=0 / 1:
 * never / whenever possible).
 * Limits: ntop >= 1 (above 128 a larger, slower candidate buffer; above 1024 passes of 1024, each continuing below the
 * last key of the one before -- the reference clips top_n to the number of distinct to-strings only, _utils.py:54-56); fewer than 2^25
 * 16-posting index pieces (4 GiB) in the to-side; n_cols equal on both sides.  `out` may have MORE rows than the from-matrix (a padded
 * shard buffer for the equal-sized all-gather): the extra rows are not touched.
 * Enqueues on the context stream. */
int pfz_cossim_topn(pfz_ctx *ctx, const pfz_index *to_index, const pfz_csr *from_matrix,
                    int32_t ntop, float lower_bound, int32_t exclude_diag, int64_t diag_offset,
                    pfz_topn *out);

/* The same for from-rows [row_begin, row_end) only (results into the same rows of `out`; exclude_diag still means
 * j == global from-row + diag_offset).  Lets a caller split one match into launches and consume the first part
 * (pfz_event_record + pfz_topn_download_rows_after) while the next one runs.  (Row ranges of a self-match that ascend
 * from row 0 without gaps, into the same `out`, continue one symmetric session; any other order is served row by row.) */
int pfz_cossim_topn_rows(pfz_ctx *ctx, const pfz_index *to_index, const pfz_csr *from_matrix,
                         int64_t row_begin, int64_t row_end, int32_t ntop, float lower_bound,
                         int32_t exclude_diag, int64_t diag_offset, pfz_topn *out);

/* The whole match, its results handed on in n_ranges ascending row ranges: range i = from-rows [range_ends[i-1], range_ends[i])
 * (the last one ends at the matrix' last row); event slot first_event + i fires when the rows of range i are final in `out`
 * (consume them with pfz_topn_rows_begin / _finish or pfz_topn_download_rows_after while the device works on).  What
 * `TFIDF.match` (reference _tfidf.py:68-100 -> _utils.py:82-125) enqueues for a big list, so that the frame's columns are built
 * under the device's work.  A list against itself in the symmetric form with range ends on multiples of 2048 rows runs ONE
 * pass-1 launch whose ranges are merged on a side stream as their blocks complete; anything else is a launch per range.
 * host_idx / host_val (may be NULL): where that streamed form runs, *host_idx / *host_val come back as int32 [n_rows][ntop] /
 * fp32 [n_rows][ntop] in PINNED HOST memory of the context that the device fills beside `out` -- rows of range i are there once
 * pfz_event_wait(first_event + i) has returned; valid until the next call with a mirror on this context --, NULL otherwise (then:
 * pfz_topn_rows_begin / _finish).  diag_offset is 0.  Enqueues. */
int pfz_cossim_topn_ranges(pfz_ctx *ctx, const pfz_index *to_index, const pfz_csr *from_matrix, int32_t ntop, float lower_bound,
                           int32_t exclude_diag, int32_t n_ranges, const int64_t *range_ends, int32_t first_event, pfz_topn *out,
                           const int32_t **host_idx, const float **host_val);

/* One-shot host convenience: upload both CSR matrices, build the index, run
 * pfz_cossim_topn, download.  Blocks. */
int pfz_cossim_topn_host(pfz_ctx *ctx,
                         int64_t n_from, int64_t n_to, int64_t n_cols,
                         const int64_t *from_indptr, const int32_t *from_indices, const float *from_data,
                         const int64_t *to_indptr, const int32_t *to_indices, const float *to_data,
                         int32_t ntop, float lower_bound, int32_t exclude_diag,
                         int32_t *out_idx, float *out_val);

/* ---- K1/K2: char-n-gram TF-IDF vectorisation ------------------------------
 * Replaces TfidfVectorizer(min_df=1, analyzer=TFIDF._create_ngrams)
 * .fit/.transform as used at reference _tfidf.py:102-118 (sklearn
 * feature_extraction/text.py: vocabulary = sorted distinct n-grams, tf = raw
 * count, idf = ln((1+n)/(1+df))+1, rows L2-normalised).
 *
 * Strings are passed as code units of `char_width` bytes (1: Latin-1/ASCII
 * code points, 4: UTF-32 code points), concatenated, with n+1 offsets (in
 * code units).  Cleaning (reference _tfidf.py:142-146) of ASCII input is done
 * on the device; the host must pre-clean strings that contain non-ASCII code
 * points (str.lower() is Unicode-aware) and pass clean = 0 for those. */
int pfz_strings_upload(pfz_ctx *ctx, const void *chars, const int64_t *offsets, int64_t n_strings,
                       int32_t char_width, pfz_strings **out);
void pfz_strings_free(pfz_strings *s);

typedef struct pfz_tfidf_params {
    int32_t ngram_lo, ngram_hi;   /* inclusive, reference n_gram_range */
    int32_t clean;                /* reference clean_string (ASCII fast path) */
    int32_t remove_space_ngrams;  /* reference remove_space_ngrams */
} pfz_tfidf_params;

/* Fit the vocabulary and idf on the concatenation docs_a + docs_b (either may
 * be NULL) -- reference: fit(to_list + from_list), _tfidf.py:109,114. */
int pfz_tfidf_fit(pfz_ctx *ctx, const pfz_tfidf_params *params,
                  const pfz_strings *docs_a, const pfz_strings *docs_b, pfz_tfidf **out);
void pfz_tfidf_free(pfz_tfidf *v);
/* vocabulary size, number of fitted documents, bits of the packed n-gram code */
int pfz_tfidf_info(const pfz_tfidf *v, int64_t *vocab_size, int64_t *n_docs, int32_t *code_bits);
/* blocks; ngrams[vocab * ngram_hi] = the vocabulary in column order, each
 * n-gram as ngram_hi UTF-32 code points (shorter n-grams 0-padded on the
 * right) -- independent of the device-side code packing; idf[vocab] fp64
 * (sklearn idf_), df[vocab].  Any pointer may be NULL. */
int pfz_tfidf_export(pfz_ctx *ctx, const pfz_tfidf *v, uint32_t *ngrams, double *idf, int64_t *df);
/* Re-create a fitted vectoriser from exported state (joblib load path,
 * reference polyfuzz.py:429-457).  ngrams must be sorted as exported. */
int pfz_tfidf_import(pfz_ctx *ctx, const pfz_tfidf_params *params, int64_t vocab_size, int64_t n_docs,
                     const uint32_t *ngrams, const double *idf, pfz_tfidf **out);
/* Vectorise a string list with a fitted vocabulary (out-of-vocabulary n-grams
 * are ignored, sklearn text.py:1271-1273).  Enqueues; result resident. */
int pfz_tfidf_transform(pfz_ctx *ctx, const pfz_tfidf *v, const pfz_strings *docs, pfz_csr **out);

/* ---- K4: all-pairs Indel ratio + row arg-max ------------------------------
 * Replaces the hot loop of EditDistance._calculate_edit_distance
 * (reference polyfuzz/models/_distance.py:89-102) with scorer =
 * rapidfuzz.fuzz.ratio (_distance.py:4,32): for every from-string the FIRST
 * to-string with the maximal ratio (np.argmax, _distance.py:99) and that
 * ratio as float64 = (1 - (|a|+|b|-2*LCS)/(|a|+|b|)) * 100 (100 when both
 * are empty).  skip_idx (host, one entry per from-string of the whole list, or
 * NULL) restates list.remove(from_string) of the self-match path
 * (_distance.py:93-96): skip_idx[i] is the index of the FIRST to-entry equal
 * to from-string i (-1: none) and is not a candidate for row i.  A code
 * skip_idx[i] <= -2 leaves out EVERY to-entry up to and including -2 - skip_idx[i]:
 * the RapidFuzz matcher's shared list that shrinks as its rows are processed
 * (_rapidfuzz.py:103-104 with n_jobs = 1: -2 - i for row i); a row that is left
 * without a candidate gets index -1, score 0.  One call uses one of the two
 * forms (entries >= 0, or codes <= -2; -1 goes with either): PFZ_ERR_INVALID otherwise.
 * Rows [from_begin, from_end) are scored (a row shard); out_idx / out_score
 * are host buffers of from_end - from_begin entries.  Blocks. */
int pfz_indel_argmax(pfz_ctx *ctx, const pfz_strings *from_strings, const pfz_strings *to_strings,
                     const int32_t *skip_idx, int64_t from_begin, int64_t from_end,
                     int32_t *out_idx, double *out_score);
/* pfz_indel_argmax with the result left on the device in the layout the sharded jobs all-gather: `out` must have 2
 * columns and >= from_end - from_begin rows; row r gets idx[r][0] = the first arg-max (idx[r][1] = -1) and the
 * float64 ratio's bits in its two value lanes.  Enqueues. */
int pfz_indel_argmax_dev(pfz_ctx *ctx, const pfz_strings *from_strings, const pfz_strings *to_strings,
                         const int32_t *skip_idx, int64_t from_begin, int64_t from_end, pfz_topn *out);
/* every ratio of rows [from_begin, from_end) x all to-strings as float64,
 * row-major host buffer (test / small-input entry point).  Blocks. */
int pfz_indel_matrix_host(pfz_ctx *ctx, const pfz_strings *from_strings, const pfz_strings *to_strings,
                          int64_t from_begin, int64_t from_end, double *out_matrix);
/* K4's preparation of a to-list -- alphabet (distinct code points of the to-list), to-strings sorted by
 * length into groups of 64, symbols packed per group on the device -- depends on the to-list alone: it is
 * built on the first pfz_indel_* call that uses the handle as the to-side and cached on it, so matching
 * further from-lists against the same pfz_strings (reference PolyFuzz.transform, polyfuzz.py:234-240)
 * costs no preparation.  This entry builds the plan if needed and reports it: alphabet size, groups,
 * and sum over to-strings of their padded length (x n_from x ceil(|from| / word) = the kernel's word-steps). */
int pfz_indel_plan_info(pfz_ctx *ctx, const pfz_strings *to_strings, int64_t *n_symbols, int64_t *n_groups,
                        int64_t *char_steps);

/* ---- K7: the per-pair rapidfuzz scorers -------------------------------------
 * Replaces process.extractOne(from_string, to_list, scorer=..., score_cutoff=...) of the reference's RapidFuzz
 * matcher (_rapidfuzz.py:99-113; scorer default fuzz.WRatio, _rapidfuzz.py:48) for the scorers that build a
 * different string pair per (from, to): the best choice (first maximum) of every from-string of rows
 * [from_begin, from_end) among all to-strings.
 * Everything is prepared on the device from the two resident lists: the three forms of every string -- 0: the string,
 * 1: its whitespace tokens (Python's str.split()) sorted and joined by one space (fuzz.token_sort_ratio's operand),
 * 2: its DISTINCT tokens sorted and joined (fuzz.token_set_ratio's) -- are built once per list and cached on its
 * handle; the to-side plan (alphabet of the to-list, token table, length-sorted groups of 64 with symbols / token ids /
 * character-class histograms) once per to-list, cached on ITS handle, so matching further from-lists against the same
 * pfz_strings (reference PolyFuzz.transform, polyfuzz.py:234-240) costs no preparation.
 * scorer: 0 WRatio, 1 partial_ratio, 2 token_set_ratio, 3 token_ratio, 4 partial_token_sort_ratio,
 * 5 partial_token_set_ratio, 6 partial_token_ratio (ratio / QRatio / token_sort_ratio are one string per list
 * element: pfz_indel_argmax).  skip_idx (host, one entry per from-string of the whole list, or NULL): a to-index
 * left out for from-string i (self-match: the first list element equal to it), or a code <= -2: every to-index up to
 * and including -2 - skip_idx[i] is left out (the reference's shared, shrinking list, _rapidfuzz.py:103-104; see pfz_indel_argmax).
 * out_idx[i] = -1 / out_score[i] = 0 when there is no choice; scores are rapidfuzz's 0..100 float64.
 * Strings of any length and token count are accepted: from-strings beyond 256 characters or 32 distinct tokens (and
 * to-strings beyond 32 distinct tokens) take a general -- slow -- kernel.  out_idx / out_score: host buffers of
 * from_end - from_begin entries.  Blocks. */
int pfz_fuzz_extract_one(pfz_ctx *ctx, const pfz_strings *from_strings, const pfz_strings *to_strings, int32_t scorer,
                         const int32_t *skip_idx, int64_t from_begin, int64_t from_end, int32_t *out_idx, double *out_score);
/* The same with the result left on the device, in the layout the sharded jobs all-gather ("shard
 * from-strings, replicate to-strings"): `out` must have 2 columns and >= from_end - from_begin rows; row r gets
 * idx[r][0] = the first best index (idx[r][1] = -1) and the float64 score's bits in its two value lanes.  Enqueues.
 * work_counters (host, 4 entries, or NULL): pairs whose upper bound was computed, pairs scored exactly, 64-bit
 * word-steps of the scored pairs (an estimate from their lengths), 0 -- the bench's work accounting; blocks if given. */
int pfz_fuzz_extract_one_dev(pfz_ctx *ctx, const pfz_strings *from_strings, const pfz_strings *to_strings, int32_t scorer,
                             const int32_t *skip_idx, int64_t from_begin, int64_t from_end, pfz_topn *out,
                             uint64_t *work_counters);
/* builds (once) and describes K7's cached plan of a to-list: alphabet size, groups of 64, distinct tokens over all
 * strings, strings with more than 32 distinct tokens (scored by the general kernel) */
int pfz_fuzz_plan_info(pfz_ctx *ctx, const pfz_strings *to_strings, int64_t *n_symbols, int64_t *n_groups, int64_t *n_tokens,
                       int64_t *n_general_strings);

/* ---- K5: dense cosine top-n -----------------------------------------------
 * Replaces cosine_similarity on dense embedding matrices
 * (reference _utils.py:74-77,95; Embeddings.match _embeddings.py:127-133):
 * rows are L2-normalised on the device (true cosine, as the sklearn branch),
 * fp32 MFMA product, fused per-row top-n. Host buffers; blocks. */
int pfz_dense_cossim_topn_host(pfz_ctx *ctx, const float *from_vec, int64_t n_from,
                               const float *to_vec, int64_t n_to, int64_t dim,
                               int32_t ntop, float lower_bound, int32_t exclude_diag,
                               int32_t *out_idx, float *out_val);
/* The same without the normalisation: raw dot products.  This is what the reference's
 * "sparse" back-end computes on dense input (_utils.py:74-82: the ndarrays are wrapped in
 * csr_matrix and multiplied as they are); Embeddings._embed hands it unit-norm rows
 * (_embeddings.py:136-145), user-supplied embeddings need not be. */
int pfz_dense_dot_topn_host(pfz_ctx *ctx, const float *from_vec, int64_t n_from,
                            const float *to_vec, int64_t n_to, int64_t dim,
                            int32_t ntop, float lower_bound, int32_t exclude_diag,
                            int32_t *out_idx, float *out_val);

/* The same operator with device-resident operands -- the to-side embeddings stay in HBM between
 * Embeddings.match(..., re_train=False) calls (reference polyfuzz.py:234-240) and a row shard of the
 * from-side is matched against a replicated to-side on every GPU (BASELINE config 5).  normalize != 0:
 * rows are scaled by 1/||row|| (true cosine), 0: raw dot products.  pfz_dense_topn enqueues on the
 * context stream and leaves (idx, score) in `out` (rows [0, n_from)); exclude_diag drops
 * j == i + diag_offset.  ntop >= 1 (beyond 1024 in passes of 1024 over the same score panel, each continuing below the last key of
 * the one before: the reference clips top_n to the number of distinct to-strings only, _utils.py:54-56). */
typedef struct pfz_dense pfz_dense;
int pfz_dense_upload(pfz_ctx *ctx, const float *vec, int64_t n, int64_t dim, int32_t normalize, pfz_dense **out);
int pfz_dense_shape(const pfz_dense *m, int64_t *n, int64_t *dim);
void pfz_dense_free(pfz_dense *m);
int pfz_dense_topn(pfz_ctx *ctx, const pfz_dense *from_vectors, const pfz_dense *to_vectors, int32_t ntop,
                   float lower_bound, int32_t exclude_diag, int64_t diag_offset, pfz_topn *out);

/* ---- K6: reductions on the hot path's output --------------------------------
 * precision_recall_curve (reference polyfuzz/metrics.py:12-53): for every threshold p_k
 * (ascending, n_thresholds <= 4096) count_ge[k] = #{i : sim[i] >= p_k} and sum_ge[k] = the sum of
 * those similarities (recall = count / n, average precision = sum / count on the host).  NaN
 * similarities are never >= a threshold.  Sums are accumulated in 64-bit fixed point (bit-
 * reproducible; absolute error of a mean below 1e-13 for similarities in [0, 100]).  Blocks. */
int pfz_pr_curve_host(pfz_ctx *ctx, const double *sim, int64_t n, const double *thresholds,
                      int32_t n_thresholds, int64_t *count_ge, double *sum_ge);
/* single_linkage (reference polyfuzz/linkage.py:5-53) of a self-match TOP-1 result, as
 * PolyFuzz._create_groups builds it (polyfuzz.py:468-475: model.match(strings) on unique strings):
 * row i = (From i, To = result idx[i][0], Similarity = round(score, 3)), rows with Similarity >
 * min_similarity (>= 0) walked in order with the reference's greedy, order-dependent rule -- cluster 0
 * is falsy, so the first cluster's members are re-assigned when met again.  out_cluster[i] = cluster
 * id of string i (-1: in no cluster); out_key[i] = position key of string i in the reference's dicts
 * (ascending key = insertion order; -1 where out_cluster is -1); out_info3 (may be NULL) = {first kept
 * row or -1, fixpoint rounds, clusters founded}.  `result` may hold more than one rank per row; only
 * rank 0 is read.  Blocks. */
int pfz_linkage_top1(pfz_ctx *ctx, const pfz_topn *result, double min_similarity,
                     int32_t *out_cluster, int32_t *out_key, int32_t *out_info3);

/* ---- multi-GPU (one process per GPU: RCCL over xGMI; or one process, many contexts) ------
 * The from-side is row-sharded, the to-side replicated; the only exchange is
 * the all-gather of per-shard results.  Bootstrap: rank 0 calls
 * pfz_comm_unique_id, the 128-byte id is broadcast by the host launcher
 * (torch.distributed / MPI / a file), every rank calls pfz_comm_init. */
int pfz_comm_unique_id(uint8_t id128[128]);
int pfz_comm_init(pfz_ctx *ctx, const uint8_t id128[128], int32_t rank, int32_t world, pfz_comm **out);
/* NCCL_VERSION_CODE of the rccl.h this library was compiled against, and ncclGetVersion() of the librccl the process resolved (a
 * torch wheel brings its own; the reference has no collective library at all -- joblib only, _distance.py:77).  pfz_comm_init
 * refuses (PFZ_ERR_RCCL) when the majors differ; bench.py prints both in its line. */
int pfz_rccl_versions(int32_t *header, int32_t *runtime);
void pfz_comm_destroy(pfz_comm *c);
/* The same communicator interface inside ONE process: `world` contexts -- on different GPUs or on the
 * same one -- each driven by its own host thread.  A collective is a host rendezvous of the ranks plus
 * event waits and device-to-device copies on every rank's stream (no RCCL).  Every rank must make the
 * same sequence of collective calls, from different threads.  The group outlives its communicators. */
typedef struct pfz_comm_group pfz_comm_group;
int pfz_comm_group_create(int32_t world, pfz_comm_group **out);
void pfz_comm_group_destroy(pfz_comm_group *g);
int pfz_comm_init_local(pfz_ctx *ctx, pfz_comm_group *g, int32_t rank, pfz_comm **out);
/* all-gather equal-sized shards of top-n results: `local` holds rows
 * [rank*rows_per_rank, (rank+1)*rows_per_rank); `global` has world *
 * rows_per_rank rows.  Enqueues on the context stream. */
int pfz_comm_allgather_topn(pfz_comm *c, const pfz_topn *local, pfz_topn *global);
/* The other sharding (BASELINE north_star: "row-sharded from/to lists ... all-gather of per-shard top-n candidates"):
 * every rank indexes a SHARD OF THE TO-ROWS and matches all from-rows against it; `local` = its top-n per from-row
 * with to-indices local to the shard, to_offset = global index of the shard's first to-row.  All-gathers the ranks'
 * candidates and leaves in `out` (same shape as `local`), on every rank, the top-n by (score desc, global to-index asc).
 * Needed only when the to-side does not fit one GPU; world x ntop <= 1024.  Enqueues. */
int pfz_comm_merge_to_shards(pfz_comm *c, const pfz_topn *local, int64_t to_offset, pfz_topn *out);
int pfz_comm_barrier(pfz_comm *c);
/* The self-match of a list against itself (reference _tfidf.py:113-116 -> _utils.py:82-91) cut over the communicator's GPUs in
 * K3's symmetric form (every unordered pair of rows scored once over all GPUs; csrc/k3_symmetric.hip): every
 * rank holds the whole list's matrix A and its index; rank r works on the rows r, r + world, ...; the ranks' pass-0 thresholds
 * and their per-row candidate lists (n x ntop 64-bit keys per rank) are all-gathered; `out` (n_rows x ntop) holds the FULL result
 * on every rank, bit-identical to pfz_cossim_topn(..., exclude_diag = 1) on one GPU.  PFZ_ERR_UNSUPPORTED when
 * pfz_index_symmetric_ok says no (then: row shards + pfz_comm_allgather_topn). */
int pfz_comm_cossim_topn_symmetric(pfz_comm *c, const pfz_index *ix, const pfz_csr *A, int32_t ntop, float lower_bound,
                                   pfz_topn *out);
int pfz_index_symmetric_ok(const pfz_index *ix, const pfz_csr *A, int32_t ntop, int32_t n_parts, int32_t *yes);
/* The same question asked by ALL ranks together (a collective: every rank calls it, it waits): *yes = 1 only if every rank's own
 * pfz_index_symmetric_ok says yes AND every rank could allocate the session buffers of its index.  What a rank reads from its own
 * environment or gets from its own allocator must not send it into pfz_comm_cossim_topn_symmetric while a peer goes to
 * pfz_comm_allgather_topn: ask this once per job (sizes fixed), then every rank takes the same branch of reference
 * _tfidf.py:113-116 -> _utils.py:82-91's self-match.  A rank that fails INSIDE pfz_comm_cossim_topn_symmetric afterwards aborts
 * the communicator (its peers' collectives return PFZ_ERR_RCCL instead of waiting for it). */
int pfz_comm_symmetric_ok(pfz_comm *c, const pfz_index *ix, const pfz_csr *A, int32_t ntop, int32_t *yes);
int pfz_comm_info(const pfz_comm *c, int32_t *rank, int32_t *world);
/* pfz_tfidf_fit over a corpus that is split across the ranks of `comm`:
 * `replicated` (may be NULL) is identical on every rank and counted once --
 * the to-list -- and `local_shard` is this rank's part of the from-list.  The
 * vocabulary is the union over ranks (all-gather of the code bitmaps) and df /
 * n_docs are all-reduced, so every rank ends with the SAME fitted vectoriser
 * the single-GPU fit on the concatenated lists would produce. */
int pfz_tfidf_fit_sharded(pfz_ctx *ctx, pfz_comm *comm, const pfz_tfidf_params *params,
                          const pfz_strings *replicated, const pfz_strings *local_shard, pfz_tfidf **out);

#ifdef __cplusplus
}
#endif
#endif /* POLYFUZZ_HIP_H */
