/*
 * motifscan_amd.h -- C-ABI of the MI355X-native PWM scan path (libmotifscan_amd.so).
 *
 * Drop-in boundary for ONE hot path of shao-lab/MotifScan: the native scorer
 * motifscan/motif/cscore.c (module motifscan.motif.cscore) as used by
 * motifscan/scanner.py:125 (c_scan_motif) and motifscan/cli/motif.py:134 (c_score).
 * Plain C, plain pointers and sizes, no Python.h, no torch types.  Every entry point
 * returns an int status (MS_OK = 0) and never calls exit(); the message of the last
 * failure on the calling thread is ms_last_error().  All handles are re-entrant and
 * device-bound (no file-scope state like cscore.c:26-34).
 *
 * There is NO CPU fallback behind this interface: without a gfx950 device every
 * compute entry point fails with MS_ERR_RUNTIME.
 *
 * Reference interface each entry point replaces (paths relative to /root/reference):
 *
 *   ms_pwmset_create      convert_pwm + get_max_raw_score          cscore.c:36-79
 *                         (PWM list -> double[4][W], max_raw clamped at 0 per column,
 *                          cutoff defaults to 1 when no cutoffs are given, cscore.c:70-74)
 *   ms_seqset_create      convert_seq                              cscore.c:81-114
 *                         (ASCII -> base codes: A/a C/c G/g T/t, anything else "no
 *                          contribution"); here 2-bit codes + a 1-bit non-ACGT plane in HBM
 *   ms_seqset_from_device same, for ASCII that is already resident in device memory
 *   ms_genome_create /    Scanner._extract_seq -> Genome.fetch_sequence (pysam)   scanner.py:71-87,
 *   ms_seqset_from_genome   genome/__init__.py:117-135: packed genome resident in HBM, regions cut on device
 *   ms_genome_create_packed / ms_genome_packed_host / ms_pack_bases_host
 *                         Genome.__init__ -> pysam.FastaFile (genome/__init__.py:61-83): the FASTA is packed ONCE into a genome
 *                         file (motifscan_amd/genome.py) whose planes are uploaded as they are
 *   ms_scan_sweep         the same extraction + scan for the windows of a fixed-stride sweep of one chromosome
 *                         (BASELINE configs[4]); every base is scored once instead of window / stride times
 *   ms_scan_regions_once  the same extraction + scan for region lists that overlap (peaks +- window/2, random controls:
 *                         cli/scan.py:43-48,76-86, region/utils.py:89-145): the union of the regions is scored once
 *   ms_scan               scan_motif / scan_motif_thread           cscore.c:317-476
 *                         (Python name c_scan_motif; "OOOII" = pwms, cutoffs, seqs, strand,
 *                          n_threads; n_threads has no meaning on the GPU and is not taken)
 *   ms_result_*           the list-of-lists result building         cscore.c:443-471
 *                         (per PWM: [seq_idx, pos, score, strand], order = seq asc, pos asc,
 *                          '+' (1) before '-' (2)), delivered as flat arrays + offsets
 *   ms_result_region_counts   the per-motif "number of regions with >= 1 site" that
 *                         motifscan/stats.py:29-31 derives from the nested site lists
 *                         (the only quantity the multi-GPU all-reduce needs)
 *   ms_result_dedup       _deduplicate_sites / deduplicate_motif_sites  scanner.py:156-193 (device)
 *   ms_result_site_tables the per-(motif, region) count / max score of write_sites_table  io/__init__.py:23-33
 *   ms_score              motif_score / motif_score_thread         cscore.c:174-302
 *                         (Python name c_score; "OOII")
 *   ms_score_ranks        c_score + the sort / rank pick of get_score_cutoffs   motif/__init__.py:378-401
 *   ms_dedup_hits         _deduplicate_sites / deduplicate_motif_sites  scanner.py:156-193
 *                         (host-side, on the sparse hit arrays)
 *   ms_stream_*           the same scan_motifs path (scanner.py:89-132) for region lists handed over in batches: host ASCII
 *                         in -> hit arrays in pinned host memory out, with the upload of batch i+2, the packing + scan of
 *                         batch i+1 and the copy-out of batch i overlapped per device (SURVEY.md 8(d) "pack + H2D + kernel +
 *                         D2H", 8(e) "double-buffered chunks"); ms_stream_submit_span does the same for the spans of a
 *                         host-streamed window sweep (BASELINE configs[4]: cli/scan.py:43-48 over a whole genome)
 *   ms_sweep_spans        the windows of a fixed-stride sweep over all chromosomes (Scanner._extract_seq per window,
 *                         scanner.py:71-87) cut into spans of bounded size, each with the global index of its first window
 */
#ifndef MOTIFSCAN_AMD_H
#define MOTIFSCAN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MS_OK            0
#define MS_ERR_INVALID   1   /* bad argument / shape           -> ValueError   */
#define MS_ERR_NOMEM     2   /* host or device allocation      -> MemoryError  */
#define MS_ERR_RUNTIME   3   /* HIP failure, no usable device  -> RuntimeError */

#define MS_STRAND_FWD    1   /* same bit mask as cscore.c:319 */
#define MS_STRAND_REV    2
#define MS_STRAND_BOTH   3

/* ms_scan flags */
#define MS_SCAN_DEFAULT      0u
#define MS_SCAN_EXACT_ONLY   1u   /* skip the integer pre-filter: score every window in fp64 (validation) */
#define MS_SCAN_COUNTS_ONLY  2u   /* only what the enrichment statistics read of a region set (stats.py:27-31): n_hits, the per-motif site numbers
                                     (ms_result_motif_offsets) and the per-motif region counts -- the hits are counted unordered, no site array is
                                     made (the hit accessors of the result fail with MS_ERR_INVALID).  What cli/scan.py:81-89 needs of the control regions. */

typedef struct ms_pwmset ms_pwmset;
typedef struct ms_seqset ms_seqset;
typedef struct ms_result ms_result;
typedef struct ms_genome ms_genome;

/* Per-call measurements, filled by ms_scan (HIP events on the library's own stream). */
typedef struct ms_scan_stats {
    int64_t n_bases;            /* total bases in the sequence set                                   */
    int64_t n_windows;          /* sum_p sum_r max(L_r - W_p + 1, 0): exact unit count               */
    int64_t n_candidates;       /* windows x strands that passed the integer pre-filter              */
    int64_t n_hits;
    int32_t n_pwms;
    int32_t n_pwms_exact;       /* PWMs routed to the all-fp64 path (W > 63, max_raw <= 0, non-finite entries, a cutoff below the quantiser's floor) */
    int32_t n_tiles;            /* LDS tiles of pre-filter operand tables                            */
    int32_t n_passes;           /* 1, or 2 when a buffer had to grow and the scan was re-run         */
    double  ms_prefilter;       /* device time of the pre-filter kernel (dominant kernel)            */
    double  ms_exact;           /* fp64 kernels: candidate re-scoring + the motifs of the all-fp64 path */
    double  ms_sort;            /* ordering of the hit list                                          */
    double  ms_finalize;        /* coordinates, per-motif offsets, region counts                     */
    double  ms_total;           /* first launch -> last kernel done                                  */
    int64_t lds_bytes_read;     /* bytes the pre-filter reads from LDS (operand tables)              */
    int64_t hbm_bytes_algorithmic; /* codes + mask + offsets + PWMs + 16 B/hit + 8 B/PWM */
    double  pf_clock_mhz;       /* shader clock held inside the pre-filter kernel; 0 unless MS_PF_CLOCK=1 */
    int64_t mfma_ops;           /* multiply-adds x 2 the pre-filter issues on the matrix cores (one-hot zeros and width padding included) */
    int64_t mfma_ops_algorithmic; /* 2 x windows x strands x W: the adds the reference performs */
    int32_t pf_engine;          /* 3: the fp6 x fp4 one-hot product on the matrix cores, candidates parked and decoded later; 4: the same with the flags decoded in place (chosen when the previous scan of the PWM set found many hits per row tile: p >= ~5e-4) */
    int32_t reserved;
} ms_scan_stats;

const char *ms_last_error(void);
int ms_version(void);
/* Bit 0: the library's pre-filter holds the hand-written gfx950 asm blocks (the variant libmotifscan_amd_asm.so, csrc/Makefile); 0 for the
 * product library, whose pre-filter is built from compiler builtins only.  Identical results either way.  No reference counterpart. */
int ms_build_flags(void);

/* Device selection is per calling thread (like hipSetDevice).  Handles remember their device. */
int ms_device_count(int *count);
int ms_set_device(int device);
int ms_device_name(char *buf, int buflen);
/* NUMA placement of the host side (no reference counterpart: the reference has one process and no device): binds the CALLING thread to the
 * CPUs of the NUMA node the calling thread's device hangs off (sysfs numa_node of its PCI address; sched_setaffinity), so that memory the
 * thread allocates afterwards -- pinned buffers included -- is node-local.  The batch stream's three threads do this by themselves where the
 * policy says so: MS_NUMA_BIND=0 never, =1 always, unset = on machines with more than one NUMA node and more than one visible GPU.
 * force != 0 overrides the policy.  *node = the node bound to, or -1 if nothing was done. */
int ms_numa_bind_thread(int force, int *node);
/* The calling thread's device keeps freed HBM blocks for reuse (hipMalloc / hipFree stall every stream of the device).
 * out[0] requests served from the cache, out[1] requests that went to the driver, out[2] blocks returned to the driver,
 * out[3] nanoseconds spent inside the driver for [1] and [2], out[4] bytes cached now, out[5] blocks cached now.
 * No reference counterpart (numpy owns the reference's memory). */
int ms_device_pool_stats(uint64_t out[6]);

/* ---- PWM set -------------------------------------------------------------------------- */
/* values: the P matrices concatenated, each row-major [4][width] (rows A,C,G,T).
 * cutoffs: P doubles or NULL (then every cutoff is 1, cscore.c:70-74). */
int ms_pwmset_create(const double *values, const int32_t *widths, const double *cutoffs,
                     int32_t n_pwms, ms_pwmset **out);
int ms_pwmset_set_cutoffs(ms_pwmset *pwms, const double *cutoffs);
int ms_pwmset_size(const ms_pwmset *pwms, int32_t *n_pwms);
int ms_pwmset_max_raw(const ms_pwmset *pwms, double *out /* [P] */);
void ms_pwmset_free(ms_pwmset *pwms);

/* ---- sequence set ---------------------------------------------------------------------- */
/* bases: the R sequences concatenated as ASCII; offsets[R+1] (offsets[0] = 0). Copies to the
 * device and packs there.  keep_ascii != 0 keeps the ASCII resident so ms_seqset_repack can
 * re-run the extraction/packing kernel (bench.py times it inside the step). */
int ms_seqset_create(const char *bases, const int64_t *offsets, int64_t n_seqs, int keep_ascii,
                     ms_seqset **out);
/* The same set with convert_seq done by n_threads HOST threads: the 2-bit codes, the non-ACGT mask and the region hints are made in pinned
 * staging memory and copied over -- no kernel runs, so building the set never waits for CUs a running scan holds (what the batch stream's
 * upload stage does under MS_STREAM_HOST_PACK).  bases is borrowed for the duration of the call. */
int ms_seqset_create_hostpacked(const char *bases, const int64_t *offsets, int64_t n_seqs, int n_threads, ms_seqset **out);
/* d_bases: device pointer to the concatenated ASCII (borrowed for the call); offsets on host. */
int ms_seqset_from_device(const void *d_bases, const int64_t *offsets, int64_t n_seqs,
                          ms_seqset **out);
int ms_seqset_repack(ms_seqset *seqs);
int ms_seqset_size(const ms_seqset *seqs, int64_t *n_seqs, int64_t *n_bases);
void ms_seqset_free(ms_seqset *seqs);

/* ---- resident genome + on-device region extraction ------------------------------------------ */
/* Replaces the per-region Genome.fetch_sequence (pysam) calls of Scanner._extract_seq
 * (scanner.py:71-87, genome/__init__.py:117-135): the genome is packed ONCE into HBM (2-bit codes +
 * non-ACGT plane), and a region list (chromosome index, 0-based half-open [start, end) already
 * clipped to the chromosome as scanner.py:81-83 does) is cut into a sequence set on the device. */
int ms_genome_create(const char *bases, const int64_t *chrom_offsets, int32_t n_chroms, ms_genome **out);
/* The genome from its packed form -- the two planes of the HBM layout: codes[2 * ceil(n / 32)] (2 bits per base, base i of a 32-base unit at
 * bits [2i, 2i + 2) of the unit's two words; a/A 0, c/C 1, g/G 2, t/T 3, anything else 0) and nmask[ceil(n / 32)] (bit i: base i is none of
 * those), the chromosomes back to back.  "Pack the FASTA once" (SURVEY.md N3): a genome file made by motifscan_amd/genome.py is uploaded as it
 * is, 0.375 B per base, with no ASCII and no pack kernel -- what replaces Genome.__init__ / pysam.FastaFile (genome/__init__.py:61-83) on the
 * measured path.  The planes are validated (MS_ERR_INVALID for a non-ACGT base with a non-zero code, or bits past the last base). */
int ms_genome_create_packed(const uint32_t *codes, const uint32_t *nmask, const int64_t *chrom_offsets, int32_t n_chroms, ms_genome **out);
/* The two planes of a resident genome copied back to host buffers of those sizes (to write the genome file). */
int ms_genome_packed_host(const ms_genome *genome, uint32_t *codes, uint32_t *nmask);
/* convert_seq (cscore.c:81-114) into the same two planes on n_threads HOST threads -- no device is touched: the genome-file builder.
 * (Marshalling only: there is still no CPU scan path.) */
int ms_pack_bases_host(const char *bases, int64_t n_bases, int n_threads, uint32_t *codes, uint32_t *nmask);
int ms_genome_size(const ms_genome *genome, int32_t *n_chroms, int64_t *n_bases);
void ms_genome_free(ms_genome *genome);
int ms_seqset_from_genome(const ms_genome *genome, const int32_t *chrom, const int64_t *start,
                          const int64_t *end, int64_t n_regions, ms_seqset **out);

/* ---- scan (c_scan_motif) ---------------------------------------------------------------- */
int ms_scan(const ms_pwmset *pwms, const ms_seqset *seqs, int strand_mask, uint32_t flags,
            ms_result **out);
/* Window sweep: windows k = 0 .. n-1 = [begin + k*stride, begin + k*stride + window) of chromosome `chrom`, all inside
 * [begin, end) (n = (end - begin - window) / stride + 1).  The result is what ms_scan gives for those n windows as n
 * regions (seq_idx = k, pos relative to the window start, same order, same fp64 scores, region counts = windows with
 * a site) -- scanner.py:71-87 + cscore.c:336-390 per window -- but the span is scanned ONCE and every hit is handed to
 * the windows that contain all of its bases. */
int ms_scan_sweep(const ms_pwmset *pwms, const ms_genome *genome, int32_t chrom, int64_t begin, int64_t end,
                  int32_t window, int32_t stride, int strand_mask, uint32_t flags, ms_result **out);
/* Region lists that OVERLAP (peaks +- window/2 closer than the window; the random control set drawn around them:
 * cli/scan.py:43-48, 76-86): the same result as ms_seqset_from_genome + ms_scan over the n_regions regions (seq_idx =
 * index in the caller's list, any order, any overlap, empty regions allowed) -- but the union of the regions is scanned
 * ONCE and every hit is handed to each region that holds all of its bases.  stats.n_bases = bases actually scanned. */
int ms_scan_regions_once(const ms_pwmset *pwms, const ms_genome *genome, const int32_t *chrom, const int64_t *start,
                         const int64_t *end, int64_t n_regions, int strand_mask, uint32_t flags, ms_result **out);
int ms_result_num_hits(const ms_result *res, int64_t *n_hits);
int ms_result_motif_offsets(const ms_result *res, int64_t *out /* [P+1] */);
/* Copy the hit arrays to host buffers of length n_hits (any pointer may be NULL). */
int ms_result_hits(const ms_result *res, int64_t *seq_idx, int64_t *pos, double *score, int8_t *strand);
/* The same arrays in library-owned pinned host memory (one device-to-host copy at PCIe rate); the
 * pointers stay valid until the result is freed or de-duplicated. */
int ms_result_hits_host(ms_result *res, const int64_t **seq_idx, const int64_t **pos, const double **score,
                        const int8_t **strand);
int ms_result_region_counts(const ms_result *res, int64_t *out /* [P] */);
/* Device pointer (int64[P]) of the same counts, for a device-side all-reduce; valid until free. */
int ms_result_region_counts_device(const ms_result *res, void **d_counts);
int ms_result_stats(const ms_result *res, ms_scan_stats *out);
/* De-duplicate overlapping sites on the device, in place (scanner.py:156-193: per motif, region and
 * strand, greedy left to right, the lower-scoring of two sites closer than the motif width is
 * dropped, a tie keeps the earlier one).  Order and per-motif region counts are unaffected. */
int ms_result_dedup(ms_result *res, const ms_pwmset *pwms);
/* Per (motif, region): number of sites and maximum score (NaN where there is none) -- the two
 * aggregates the reference's site tables are made of (io/__init__.py:23-33).  Host buffers [P][R]. */
int ms_result_site_tables(const ms_result *res, int32_t *n_sites, double *max_score);
void ms_result_free(ms_result *res);

/* The hit arrays in COMPACT form in library-owned pinned host memory: coord[i] = seq_idx << 32 | pos << 1 | (strand - 1),
 * score[i] -- 16 bytes per hit on the host link instead of 25.  Needs seq_idx < 2^32 and pos < 2^31 (MS_ERR_INVALID
 * otherwise).  Valid until the result is freed or de-duplicated. */
int ms_result_hits_packed_host(ms_result *res, const uint64_t **coord, const double **score);
/* ... and in 12 bytes per hit: coord32[i] = seq_idx << *shift | pos << 1 | (strand - 1), score[i] -- for results whose region indices and
 * positions fit 31 bits together (a batch of 250 000 regions of 500 bp: 18 + 9), i.e. every batch of a batch stream; MS_ERR_INVALID (and the
 * 16-byte form stays available) when they do not.  Same list building (cscore.c:443-471), a quarter less on the host link. */
int ms_result_hits_packed12_host(ms_result *res, const uint32_t **coord, const double **score, int32_t *shift);
/* Which compact form a result holds after a batch stream's copy-out: *bytes_per_hit = 12, 16, or 0 (the plain arrays). */
int ms_result_packed_form(const ms_result *res, int32_t *bytes_per_hit);

/* ---- pinned host memory -------------------------------------------------------------------- */
/* Page-locked host memory for sequence input: uploads from it run at the full link rate and overlap with scans. */
int ms_host_alloc(size_t bytes, void **out);
/* The library's cache of pinned result blocks (size classes, like the device block cache): out[0] requests served from it, out[1] requests that
 * went to the driver, out[2] blocks returned to the driver, out[3] nanoseconds inside the driver.  Steady-state batches show no [1] / [2]. */
int ms_host_pool_stats(uint64_t out[4]);
void ms_host_free(void *p);

/* ---- batch streams: upload | pack + scan | copy-out overlapped ------------------------------- */
#define MS_STREAM_DEDUP       1u   /* de-duplicate every batch on the device (scanner.py:156-193) before the copy-out      */
#define MS_STREAM_NO_HITS     2u   /* counts only (control regions: stats.py:29-31): a batch is scanned with MS_SCAN_COUNTS_ONLY, a sweep span
                                      makes ONLY the per-motif window counts and the number of sites (no site array exists: the hit
                                      accessors of such a result fail with MS_ERR_INVALID; ms_result_motif_offsets then gives the running
                                      per-motif site numbers, consistent with n_hits; with MS_STREAM_DEDUP such a span is NOT de-duplicated --
                                      de-duplication never empties a window, the counts are the same)                           */
#define MS_STREAM_EXACT_ONLY  4u   /* MS_SCAN_EXACT_ONLY for every batch (validation)                                      */
#define MS_STREAM_PACKED      8u   /* copy the hits out in the compact form of ms_result_hits_packed_host                  */
#define MS_STREAM_PACKED12   32u   /* ... in the 12-byte form of ms_result_hits_packed12_host for every batch that fits it, the 16-byte form for
                                      the others (ms_result_packed_form tells which)                                        */
#define MS_STREAM_HOST_PACK  16u   /* the upload stage packs on host threads (ms_seqset_create_hostpacked): no kernel beside the scan */
typedef struct ms_stream ms_stream;
/* depth: batches that may wait between two stages (>= 1; 2 overlaps all three stages).  The stream is bound to the
 * calling thread's device (ms_set_device).  The PWM set must outlive the stream. */
int ms_stream_create(const ms_pwmset *pwms, int strand_mask, uint32_t flags, int depth, ms_stream **out);
/* Queue one batch of regions (same arguments as ms_seqset_create).  offsets is copied; bases is BORROWED until
 * ms_stream_next has returned this batch.  Fails with MS_ERR_INVALID when ms_stream_capacity batches are in flight. */
int ms_stream_submit(ms_stream *st, const char *bases, const int64_t *offsets, int64_t n_seqs);
/* The same for a batch of which only the per-motif region counts will be read (MS_STREAM_NO_HITS for this batch alone): what the
 * reference does with the control regions -- cli/scan.py:81-89 scans them, stats.py:29-31 counts the regions with >= 1 site, no
 * writer ever sees their sites -- while the input regions' batches of the same stream carry their hits out. */
int ms_stream_submit_counts_only(ms_stream *st, const char *bases, const int64_t *offsets, int64_t n_seqs);
/* Queue one span of a window sweep: n_bases of ONE chromosome starting at a window start; the result is what
 * ms_scan_sweep gives for it (seq_idx = window index inside the span: add ms_span.first_window). */
int ms_stream_submit_span(ms_stream *st, const char *bases, int64_t n_bases, int32_t window, int32_t stride);
/* A batch of regions of a genome that is resident in HBM (ms_genome_create): what ms_seqset_from_genome + ms_scan give for them,
 * with the cut of batch i + 1 out of the 2-bit genome overlapping the scan of batch i and the copy-out of batch i - 1
 * (scanner.py:71-87 + 125 per batch; the genome must outlive the batch's result).  The arrays are copied. */
int ms_stream_submit_regions(ms_stream *st, const ms_genome *genome, const int32_t *chrom, const int64_t *start, const int64_t *end,
                             int64_t n_regions);
/* The oldest batch's result (submission order), its hit arrays already in pinned host memory (ms_result_hits_host /
 * ms_result_hits_packed_host return at once).  *out = NULL when nothing is in flight.  The caller frees the result. */
int ms_stream_next(ms_stream *st, ms_result **out);
int ms_stream_in_flight(const ms_stream *st, int *n);
/* Where the stream's three stages spent their time so far, for stage k = 0 uploader, 1 scanner, 2 downloader:
 * out[4k] batches done, out[4k+1] ms working, out[4k+2] ms waiting for input, out[4k+3] ms waiting for room downstream.
 * The stage with the least waiting is the one that bounds the stream. */
int ms_stream_stats(const ms_stream *stream, double out[12]);
int ms_stream_capacity(const ms_stream *st, int *n);
void ms_stream_free(ms_stream *st);          /* drains and discards whatever is still in flight */

/* ---- sweep planning ------------------------------------------------------------------------ */
typedef struct ms_span {
    int32_t chrom;            /* chromosome index                                                       */
    int32_t reserved;
    int64_t begin, end;       /* bases [begin, end) of the chromosome; begin is a window start          */
    int64_t first_window;     /* global index (over all chromosomes, in order) of the span's first window */
    int64_t n_windows;
} ms_span;
/* Cut the sweep "windows [k*stride, k*stride + window) of every chromosome" (chromosomes shorter than a window have
 * none) into spans of at most max_span_bases bases; neighbouring spans of a chromosome overlap by window - stride bases.
 * spans may be NULL (count only); at most cap entries are written; *n_spans is the number needed. */
int ms_sweep_spans(const int64_t *chrom_len, int32_t n_chroms, int32_t window, int32_t stride, int64_t max_span_bases,
                   ms_span *spans, int64_t cap, int64_t *n_spans);

/* ---- score (c_score) -------------------------------------------------------------------- */
/* out: host, [P][R] row-major.  A sequence shorter than a PWM scores its missing bases as
 * non-ACGT (the reference reads out of bounds there, cscore.c:195-196). */
int ms_score(const ms_pwmset *pwms, const ms_seqset *seqs, int strand_mask, double *out);

/* The device half of the cutoff builder (`motifscan motif --build`: cli/motif.py:134-137 and
 * get_score_cutoffs, motif/__init__.py:378-401): c_score of all R sequences for every PWM, each
 * PWM's scores sorted in DESCENDING order, out[p][k] = score at 0-based rank ranks[k]
 * (the reference reads rank int(R * 0.1**e) - 1 for e = 2 .. min(len(str(R)), 7) - 1). */
int ms_score_ranks(const ms_pwmset *pwms, const ms_seqset *seqs, int strand_mask, const int64_t *ranks,
                   int32_t n_ranks, double *out /* [P][n_ranks] */);

/* ---- de-duplication of overlapping sites (scanner.py:156-193), host side ---------------- */
/* In: hits in ms_result order.  Out: keep[n_hits] (1 = kept).  Kept hits are already in the
 * order Scanner.scan_motifs returns them (start ascending, '+' before '-' on ties). */
int ms_dedup_hits(const int64_t *motif_offsets, int32_t n_pwms, const int32_t *widths,
                  const int64_t *seq_idx, const int64_t *pos, const double *score,
                  const int8_t *strand, uint8_t *keep);

#ifdef __cplusplus
}
#endif
#endif /* MOTIFSCAN_AMD_H */
