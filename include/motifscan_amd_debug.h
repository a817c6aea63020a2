/*
 * motifscan_amd_debug.h -- host-only inspection of the integer pre-filter plan of
 * libmotifscan_amd.so.  NOT part of the drop-in surface (nothing in the reference corresponds
 * to it): it lets CPU tests prove that the quantised 2-mer tables can never drop a window the
 * reference scorer (cscore.c:340-390) reports.  Needs no GPU.
 */
#ifndef MOTIFSCAN_AMD_DEBUG_H
#define MOTIFSCAN_AMD_DEBUG_H

#include "motifscan_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Build (and cache) the plan for a strand mask and an LDS budget in bytes; report its shape. */
int ms_debug_plan_dims(const ms_pwmset *pwms, int strand_mask, int64_t lds_budget, int32_t *n_fast,
                       int32_t *n_exact, int32_t *n_groups, int32_t *n_tiles);

/* Copy the plan built by the last ms_debug_plan_dims call (any pointer may be NULL):
 *   group_motifs [n_groups][8] (-1 = empty slot), group_G [n_groups] (2-mer positions),
 *   group_fb [n_groups] (field bits, 10 or 16),
 *   tables [n_groups][16 positions][16 codes][4 words]: field n (motif slot n >> 1, even n forward,
 *   odd n reverse) sits in word n & 3 at bit (n >> 2) * fb,
 *   exact_motifs [n_exact], tile_first_group [n_tiles + 1]. */
int ms_debug_plan_tables(const ms_pwmset *pwms, int32_t *group_motifs, int32_t *group_G, int32_t *group_fb,
                         uint32_t *tables, int32_t *exact_motifs, int32_t *tile_first_group);

/* The int8 / matrix-core forms of the plan (environment MS_PF_ENGINE=1 or 2 when ms_debug_plan_dims ran), decoded
 * from the operand image the kernel reads: rows [n_groups][16 fields][32 motif columns][4 bases] = what the product
 * adds for that base at that column, bias [n_groups][16] (engine 2: the row constant held in the spare k-slots;
 * engine 1: 0); field n: motif slot n >> 1, even n forward, odd n reverse; a window is a candidate for a field iff
 * bias + the sum over its columns of rows[..][column][base at window start + column] is >= 0;
 * group_kb [n_groups] = k-blocks evaluated for the group (8 columns each for engine 1, 10 for engine 2). */
int ms_debug_plan_mfma_rows(const ms_pwmset *pwms, int16_t *rows, int32_t *bias, int32_t *group_kb);

/* Free the current device's grow-only work buffers (candidate list, hit list, sort space), so a
 * test can force the "buffer too small -> grow -> run the pass again" path.  Needs a GPU. */
int ms_debug_release_scratch(void);

#ifdef __cplusplus
}
#endif
#endif
