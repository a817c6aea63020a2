/*
 * motifscan_amd_debug.h -- host-only inspection of the integer pre-filter plan of
 * libmotifscan_amd.so.  NOT part of the drop-in surface (nothing in the reference corresponds
 * to it): it lets CPU tests prove that the quantised operand rows can never drop a window the
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

/* The plan built by the last ms_debug_plan_dims call, decoded from the PHYSICAL fp6 operand image the kernel reads
 * (any pointer may be NULL).  A table group has 16 fields = the 16 result registers of a lane: with both strands scanned
 * field n = motif slot n >> 1, even n forward, odd n reverse; with one strand field n = motif slot n.
 *   group_fields [n_groups][16]  motif of the field, -1 = empty
 *   rows [n_groups][16][64 motif columns][4 bases] int16: what the product adds for that base at that column, in units
 *        of 1/8; a non-ACGT base adds nothing (its one-hot column is all zero)
 *   bias [n_groups][16]: the row's constant (stored in the last column of the row tile, which the kernel never clears)
 *   group_kb [n_groups]: k-blocks of 16 columns evaluated for the group's row tile
 *   exact_motifs [n_exact], tile_first_group [n_tiles + 1]
 * A window is a candidate for a field iff bias + the sum over its ACGT columns of rows[..][column][base] is >= 0. */
int ms_debug_plan_rows(const ms_pwmset *pwms, int32_t *group_fields, int16_t *rows, int32_t *bias, int32_t *group_kb,
                       int32_t *exact_motifs, int32_t *tile_first_group);

/* The host packer of ms_seqset_create_hostpacked alone (no device): codes [2 x ceil(n / 32)], nmask [ceil(n / 32)], blk2reg [(n + 63) / 64 + 1],
 * blkinfo [4 x that], n = offsets[n_seqs] -- the layout pack_kernel / blk2reg_kernel write on the device. */
int ms_debug_host_pack(const char *bases, const int64_t *offsets, int64_t n_seqs, uint32_t *codes, uint32_t *nmask, int32_t *blk2reg, int32_t *blkinfo);

/* The NUMA look-ups of ms_numa_bind_thread against a sysfs tree under `root` ("" = the machine's own; no device, no binding): the node of
 * PCI device `bdf` (-1 unknown), the number of CPUs of that node (0 unknown), the number of online nodes. */
int ms_debug_numa_probe(const char *root, const char *bdf, int32_t *node, int32_t *n_cpus, int32_t *n_nodes);

/* Free the current device's grow-only work buffers (candidate list, hit list, sort space), so a
 * test can force the "buffer too small -> grow -> run the pass again" path.  Needs a GPU. */
int ms_debug_release_scratch(void);

#ifdef __cplusplus
}
#endif
#endif
