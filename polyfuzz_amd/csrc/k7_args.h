// The argument block of K7's match kernels (k7_fuzz.hip: the register / LDS kernel; k7_general.hip: the general one).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfz {

struct FuzzArgs {
    // from side: the three forms as code units + the distinct tokens (forms layout), mapped through the to-list's alphabet here
    const void *a_form[3];
    int32_t a_width;
    const int64_t *a_off;
    const int32_t *a_len1, *a_len2, *a_ntok, *a_ntok_all, *a_tok_len, *a_tok_id;
    const uint16_t *lut;
    uint32_t lut_len;
    const uint8_t *cls;
    int32_t space_rank, space_class;
    const int32_t *rows;         // from-rows of this launch
    int32_t n_rows;
    // to side (the plan): per-string records + per-slot summaries
    const uint16_t *b_sym;
    const uint8_t *b_tag;
    const int32_t *b_tok_id, *b_tok_len;
    const int4 *b_meta, *b_meta2, *b_meta3, *b_meta4;
    const uint4 *b_hist;
    const uint2 *b_pres;         // [slot] symbol presence, 64 bits (k7_core.h: fz_presence_miss): part of the bound since round 4
    const int32_t *big_slots;    // general kernel: only these to-slots (n_big > 0), else all
    int32_t n_big;
    int32_t n_groups, n_sym1, mode;
    const int32_t *skip_idx;     // [n_from] or NULL (decoded: pfz_internal.h decode_skip_codes)
    int32_t skip_up_to;          // 0: choice skip_idx[row] is left out; 1: every choice up to skip_idx[row] is
    // every launch leaves the best of its (row, part) in part_*[row_slot * n_parts_total + part0 + part]
    int32_t parts, part0, n_parts_total;
    const int32_t *row_slot;     // [n_rows] position of the row in the output range
    double *part_score;
    int32_t *part_idx;
    unsigned long long *counters;    // [0] pairs bounded, [1] pairs scored, [2] 64-bit word-steps of the scored pairs (or NULL)
    // tuning aids (PFZ_K7_ROW_STATS / PFZ_K7_EXP; never set in production): per from-row {pairs scored, clock ticks}, experiment
    unsigned long long *row_stats;
    unsigned long long *phase_ticks;         // profiling only (PFZ_K7_ROW_STATS): [24] shader-clock ticks of all waves by phase and batch statistics, then per row {2^62 - first start, last end} (100 MHz clock)
    int32_t exp;                     // 1: bound only, score nothing (results wrong)
    int32_t *next_unit;              // dynamic distribution of the (row, part) units over the workgroups
    // sweep 1 leaves every pair's bound here (one byte: the bound rounded UP in steps of 1 / 1.27, bit 7 = coarse; 0 = not a
    // candidate), [stretch][group][lane]; sweep 2 walks the bytes.  A stretch per workgroup of the launch, then spare ones:
    // a row that is handed over keeps the stretch its bytes are in, and its workgroup goes on with a spare one
    uint8_t *ub_cache;
    int32_t *region_next, *cont_region;      // spare stretches taken so far; per hand-over (at its first record): the row's stretch (-1: none)
    int32_t n_regions;                       // stretches of this launch: one per workgroup, then the spare ones
    // hand-over of heavy rows: once its seeds are scored, a wave counts the bytes that reach their best score; a row with
    // more than hand_batches batches of them (and hand_min_groups groups) is dealt to continuation units -- one record
    // {row position r, first group, group step, units | part << 8 (-1: void)} per unit, claimed together from n_cont,
    // cont_cur / cont_region at the first of them: the best score of the row so far (the units raise it as they go) and
    // the stretch with its bytes -- and the wave goes on to the next row.  The units after the last row are those shares,
    // taken by whichever wave runs out of rows first, while others are still on theirs (a wave waits for a record only as
    // long as rows are unfinished: rows_done counts them).  Exact: the shares are the same set of pairs, bounded against a
    // score that is really attained.
    int4 *cont_list;
    unsigned long long *cont_cur;
    int32_t *n_cont, *rows_done;
    int32_t cont_cap, cont_parts, cont_part0, hand_batches, hand_min_groups;
    int32_t hand_short_len, hand_short_batches;      // from-strings this short are heavy from fewer batches on (their pairs sweep windows)
};

}  // namespace pfz
