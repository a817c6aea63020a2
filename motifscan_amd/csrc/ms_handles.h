// ms_handles.h -- private to libmotifscan_amd: per-device state and the structs behind the opaque handles of
// include/motifscan_amd.h, shared by ms_api.hip (scan pipeline) and ms_stream.hip (batch streams, host-streamed sweeps).
#pragma once
#include <sched.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ms_kernels.h"

namespace ms {

struct Scratch {                 // grow-only work buffers of the scan pipeline
    uint64_t *cand = nullptr;     size_t cand_cap = 0;
    uint64_t *keys = nullptr;     double *vals = nullptr;   uint64_t *keys_sorted = nullptr;  size_t hit_cap = 0;
    void *sort_tmp = nullptr;     size_t sort_tmp_bytes = 0;
    unsigned int *chunk_counters = nullptr;    size_t chunk_counters_cap = 0;   // per LDS tile: the pre-filter's chunk dispenser
    unsigned long long *counters = nullptr;      // 8 words, see scan_locked
    unsigned long long *h_counters = nullptr;    // pinned
};

// Free list of result blocks (ms_result keeps its arrays in HBM until freed; a scan loop would
// otherwise pay a hipMalloc + hipFree of ~200 MB per call).
struct BlockPool {
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> free_;
    size_t bytes = 0;
    // Blocks are handed out in size classes (pool_class: eight per octave), so that batches of similar but unequal size -- the
    // chromosomes of a sweep, the batches of a region stream -- reuse each other's blocks instead of going to the driver:
    // hipMalloc maps pages for milliseconds and hipFree waits for the whole device, either stalls every stage of a stream.
    size_t max_bytes = 96ull << 30;                        // cached at most: a third of the device's memory (set in get_ctx from hipMemGetInfo)
    static constexpr size_t kMaxBlocks = 512;
    uint64_t n_hit = 0, n_miss = 0, n_driver_free = 0, ns_driver = 0;
};

// A stream in two editions: `whole` may use every CU; `part` is confined by a CU mask (hipExtStreamCreateWithCUMask).
// With MS_MEASURE=1 MS_CU_PARTITION=1 the CUs are partitioned while a batch stream is live (DeviceCtx::n_streams > 0): the
// scan's stream keeps all but 1 of every 32 CUs, the upload / copy-out streams get those.  What it is for
// (tools/ubench/cu_share_probe.hip): the pre-filter's blocks are persistent and each fills a CU, and a second kernel's
// workgroups are handed to the shader engines in order -- one full engine stalls the whole hand-out, so a pack or copy kernel
// launched beside the pre-filter ends only when the pre-filter does, even when whole CUs elsewhere are idle (248 or 240 hog
// blocks still hold a side kernel back for their whole run; 224, one free CU per engine, do not).  With the masks the 62 MB
// upload + pack takes 2.4 ms beside a scan, not 4.  It is OFF by default because it buys nothing end to end
// (profiles/r02_cu_partition_ab.log): the pre-filter runs power-limited, and copy kernels that really run beside it lower its
// clock (2236 vs 2343 MHz) on top of the 3 % of CUs they take -- 76.5 vs 75.2 ms per configs[3] pass, and the sweep's
// copy-out of every site drops to 41 GB/s on 8 CUs.  Starved copies that wait for the pre-filter cost less than concurrent ones.
struct StreamSel {
    hipStream_t whole = nullptr, part = nullptr;
    const std::atomic<int> *n_streams = nullptr;
    operator hipStream_t() const { return (part && n_streams && n_streams->load(std::memory_order_relaxed) > 0) ? part : whole; }
};

struct DeviceCtx {
    int device = -1;
    BlockPool pool;
    // stream is only used under `mu`, and n_streams only changes under `mu`: one scan sees one edition throughout.
    // stream_up / stream_down are used without `mu`: every function reads them ONCE into a local hipStream_t.
    StreamSel stream;
    StreamSel stream_up;                     // sequence upload + packing (runs beside a scan of the previous batch)
    StreamSel stream_down;                   // copy-out of hit arrays (runs beside a scan of the next batch)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev[8] = {};
    int n_cu = 0;
    int n_cu_copy = 0;                       // CUs the copy streams own while the device is partitioned (0: masks unavailable)
    std::atomic<int> n_streams{0};           // live batch streams on this device
    size_t lds_max = 0;
    bool rc_lds_set = false;                // rescore_carry_kernel's dynamic-LDS attribute raised
    size_t lds_set[128] = {};               // dynamic-LDS attribute already raised to this, per kernel variant (+64: measurement instantiation)
    Scratch sc;
    std::mutex mu;               // one scan at a time per device (shared scratch)
    // the pending-scan slots (events + pinned counter words) of batch streams that have ended, for the next stream on this device: a pass that
    // opens and drains its own stream (bench.py 'pipelined') otherwise pays 14 hipEventCreate + 2 hipHostMalloc at its start and their release --
    // hipHostFree waits for the whole device -- at its end (round 6)
    std::mutex pend_mu;
    std::vector<struct PendingScan *> pend_cache;
};

int get_ctx(int device, DeviceCtx **out);
int current_device();                 // the calling thread's device (ms_set_device)
void set_current_device(int device);  // thread-local only; no HIP call

// Give every cached block of the calling thread's device back to the driver (ms_api.hip); returns the bytes released.
size_t pool_trim_current_device();
// ms_hostpack.cpp: convert_seq and the region hints on host threads (units of 32 bases / blocks of 64 positions [u0, u1) / [b0, b1))
void host_pack_units(const uint8_t *bases, int64_t n_bases, int64_t u0, int64_t u1, uint32_t *codes, uint32_t *nmask);
void host_region_hints(const int64_t *offsets, int64_t R, int64_t b0, int64_t b1, int32_t *blk2reg, int32_t *info, bool all_far);
// ms_numa.cpp: NUMA placement of a device's host side (sysfs + sched_setaffinity; no device code)
int parse_cpulist(const char *text, cpu_set_t *set);
int numa_node_of_bdf(const char *bdf, const char *root);
int numa_cpus_of_node(int node, const char *root, cpu_set_t *set);
int numa_node_count(const char *root);
int numa_bind_calling_thread(int node);
// the policy (MS_NUMA_BIND, ms_numa.cpp) applied to the calling thread for `device`: returns the node bound to, -1 if none
int numa_bind_for_device(int device, bool force);
void pinned_pool_stats(uint64_t out[4]);
int result_fetch_region_counts(ms_result *r);
int seqset_create_upload_only(const char *bases, const int64_t *offsets, int64_t n_seqs, ms_seqset **out);
int seqset_pack_pending(const ms_seqset *s, hipStream_t st);
int seqset_create_hostpacked(const char *bases, const int64_t *offsets, int64_t n_seqs, int n_threads, void **stage_io, size_t *stage_bytes_io, ms_seqset **out);

template <typename T>
inline int dev_alloc(T **p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T));
    if (e == hipErrorOutOfMemory && pool_trim_current_device() > 0) {       // the block cache may be what fills the device: drop it, once
        (void) hipGetLastError();
        e = hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T));
    }
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? MS_ERR_NOMEM : MS_ERR_RUNTIME;
    }
    return MS_OK;
}

template <typename T>
inline void dev_free(T *&p) {
    if (p) (void) hipFree(p);
    p = nullptr;
}

int pool_alloc(DeviceCtx *c, size_t bytes, void **out, size_t *got);
void pool_free(DeviceCtx *c, void *p, size_t bytes);
void *pinned_alloc(size_t bytes, size_t *got);
void pinned_free(void *p, size_t bytes);

}  // namespace ms

// ------------------------------------------------------------------------- handles --

struct ms_pwmset {
    int32_t P = 0;
    std::vector<double> values;       // concatenated [4][W] row-major, as given
    std::vector<int64_t> val_off;     // [P+1] in doubles
    std::vector<int32_t> widths;
    std::vector<double> cutoffs;
    std::vector<double> max_raw;
    int max_width = 0;
    uint64_t cutoff_version = 1;
    // device copies (lazy)
    int device = -1;
    uint64_t dev_cutoff_version = 0;
    double2 *d_tab2 = nullptr;
    int64_t *d_tab_off = nullptr;
    int32_t *d_width = nullptr;
    int64_t tab2_entries = 0;                     // entries of d_tab2 before its closing all-zero entry
    std::vector<int64_t> tab_off_host;            // the motifs' offsets in d_tab2 (entries)
    double *d_thresh = nullptr;                   // [P][4] {max_raw, cutoff, raw_floor, 0} (DevPwm::thresh)
    std::vector<double> raw_floor_host;           // [P] (FieldMeta::floor32 is made of it)
    ms::FieldMeta *d_field_meta = nullptr;        // [table groups][16] of the plan (rescore_kernel)
    double *d_max_raw = nullptr;
    double *d_cutoff = nullptr;
    double *d_raw_floor = nullptr;
    // what the last scan with these PWMs found, for the one-sync form of the next (scan_locked): hits per window, and how far
    // above it the next count may be before the prediction counts as failed
    double pred_density = -1.0, pred_margin = 0.06;
    int pred_strand = -1;
    uint64_t pred_cutoff_version = 0;
    bool pred_exact_only = false;
    // pre-filter plan (lazy, keyed by strand mask / cutoffs / LDS budget / exact-only)
    ms::PrefilterPlan plan;
    int plan_strand = -1;
    uint64_t plan_cutoff_version = 0;
    size_t plan_lds = 0;
    bool plan_exact_only = false;
    bool plan_pair = true;                        // paired rows in the plan (always, but for MS_MEASURE=1 MS_PF_PAIR=0)
    int plan_device = -1;
    // (strand, cutoffs, exact-only) for which a set WITH motifs of >= 32 columns is known to plan without a wide tile (scan_locked)
    int narrow_strand = -1;
    uint64_t narrow_cutoff_version = 0;
    bool narrow_exact_only = false;
    uint4 *d_tables = nullptr;
    ms::TileDesc *d_tiles = nullptr;
    int32_t *d_group_fields = nullptr;            // [table groups][kGroupFields] motif of the field, -1 = empty
    int32_t *d_exact_motifs = nullptr;
    std::mutex mu;
};

struct ms_seqset {
    int device = 0;
    int64_t R = 0;
    int64_t n_bases = 0;
    std::vector<int64_t> offsets;         // host copy [R+1]
    std::vector<int64_t> len_sorted;      // DISTINCT region lengths ascending
    std::vector<int64_t> len_cnt_ge;      // [i]: number of regions with length >= len_sorted[i]   (+ trailing 0)
    std::vector<int64_t> len_sum_ge;      // [i]: their total length                              (+ trailing 0)
    void *block = nullptr;                // one pooled device block holding codes / nmask / offsets / blk2reg
    size_t block_bytes = 0;
    uint8_t *d_ascii = nullptr;           // pooled block; kept only when asked to
    size_t ascii_bytes = 0;
    uint32_t *d_codes = nullptr;
    uint32_t *d_nmask = nullptr;
    int64_t *d_offsets = nullptr;
    int32_t *d_blk2reg = nullptr;         // region of position 64*b
    int4 *d_blkinfo = nullptr;            // ... with the region's and the next two regions' starts relative to 64*b (DevSeq::blkinfo)
    hipStream_t up = nullptr;             // the upload stream this set is being built on (DeviceCtx::stream_up, read once)
    void *h_off_pin = nullptr;            // the offsets' way to the device: a pinned copy (from pageable memory the runtime copies with a KERNEL, which waits for a CU the pre-filter holds)
    size_t h_off_pin_bytes = 0;
    bool built = false;                   // construction finished (its work on `up` is done)
    bool pack_pending = false;            // a batch stream's upload-only set: ASCII and offsets are in HBM, pack_kernel / blk2reg_kernel still to run (seqset_pack_pending, on the scan stream)
};

struct ms_result {
    int device = 0;
    int32_t P = 0;
    int64_t R = 0;                                    // sequences of the scanned set
    int64_t n_hits = 0;
    bool deduped = false;
    int raw_gbits = 0, raw_pbits = 0;                 // MS_SCAN_RAW_INTERNAL: layout of the unordered hit keys left in the device scratch
    bool invalid = false;                             // a no-emit measurement run (MS_MEASURE=1 MS_PF_NOEMIT=1): stage times only, no hits
    bool counts_only = false;                         // a sweep span of a counts-only stream: per-motif window counts and the number of sites, NO site arrays
    void *block = nullptr;                            // one device block holding everything below
    size_t block_bytes = 0;
    int64_t *d_seq_idx = nullptr;
    int64_t *d_pos = nullptr;
    double *d_score = nullptr;
    int8_t *d_strand = nullptr;
    unsigned long long *d_region_counts = nullptr;   // [P]
    int64_t *d_motif_first = nullptr;                 // [P+1]: after ms_scan returns, the per-motif offsets
    std::vector<int64_t> motif_offsets;               // [P+1]
    void *coord_blk = nullptr;                        // MS_SCAN_PACK_INTERNAL: the compact coordinate words, made by the scan itself ...
    size_t coord_bytes = 0;
    uint64_t *d_coord = nullptr;                      // ... [n slots] + a "does not fit" flag word behind them
    unsigned int *d_coord_bad = nullptr;
    int coord_shift = 0;                              // 0: 8-byte words; > 0: 4-byte words seq_idx << coord_shift | pos << 1 | strand bit (MS_SCAN_PACK12_INTERNAL, when the set fits)
    int h_coord_shift = 0;                            // ... the form of the host copy
    void *h_pinned = nullptr;                         // host copy of the hit arrays (pinned), made on demand
    bool h_packed = false;                            // ... in the compact form (coord | score)
    size_t h_pinned_bytes = 0;
    int64_t h_pinned_hits = -1;
    std::vector<int64_t> h_region_counts;             // a batch stream's copy-out stage brings the per-motif region counts along (ms_result_region_counts then copies host to host)
    ms_scan_stats stats;
};

namespace ms {

// Internal scan flag: stop after the fp64 stage -- the hits stay UNORDERED in the device scratch (c->sc.keys / vals, key =
// motif << (gbits + 1) | coordinate << 1 | strand bit, coordinate = sequence << pbits | position when pbits > 0, else the
// global base position); the result holds only the count, the key layout and the stage times.  For callers that re-key the
// hits anyway (ms_scan_regions_once) and keep holding c->mu until they are done with the scratch.
#define MS_SCAN_RAW_INTERNAL 0x80000000u

// Internal scan flag: also produce the compact coordinate words (ms_result_hits_packed_host) at the end of the scan, on the scan
// stream -- a batch stream's copy-out then consists of copies only (a pack kernel launched on the copy-out stream would wait for
// the next batch's pre-filter, whose persistent blocks fill every CU).
#define MS_SCAN_PACK_INTERNAL 0x20000000u
// ... in the 4-byte form (ms_result_hits_packed12_host) when every (region, position) of the set fits 31 bits; the 8-byte form otherwise.
#define MS_SCAN_PACK12_INTERNAL 0x10000000u
// Internal scan flag: COUNTS ONLY -- the result holds n_hits, the per-motif site numbers (motif_offsets) and the per-motif region counts, NO site
// arrays (ms_result::counts_only: the hit accessors refuse it): the hits are counted where the fp64 stage left them, unordered
// (count_only_kernel), and the radix sort + finalize are skipped.  Falls back to the ordered path by itself where the bitmap form does not
// apply (global-position keys, more than 4096 motifs, an absurd P x R).
#define MS_SCAN_COUNTS_ONLY_INTERNAL 0x08000000u
// Internal scan flag: never use the predicted-size form (the exactly-sized re-run after a failed prediction).
#define MS_SCAN_NO_PREDICT_INTERNAL 0x40000000u

// A scan whose launches are all queued but whose completion has not been waited for (the predicted-size form, scan_locked with
// pend != nullptr): the owner -- a batch stream's scanner thread -- queues the NEXT batch's scan behind it before it waits, so
// the device never idles between batches.  The slot's events and pinned counter words belong to one scan at a time.
struct PendingScan {
    hipEvent_t ev[6] = {};                   // around the stages, as DeviceCtx::ev
    hipEvent_t done = nullptr;
    unsigned long long *h_counters = nullptr;   // pinned, 8 words
    int64_t *h_offsets = nullptr;               // pinned: the queued scan's per-motif offsets land here, in stream order, in front of `done`
    size_t h_offsets_cap = 0;                   // (words; grown to P + 1)
    bool offsets_queued = false;                // (false only under MS_MEASURE=1 MS_OFFSETS_BLOCKING=1: round 5's blocking fetch in scan_complete, for A/B runs)
    size_t cand_cap = 0, hit_cap = 0;           // the scratch capacities at queue time
    bool active = false;
    ms_result *raw = nullptr;
    size_t n_pred = 0;
    uint64_t cand_static = 0;
    int64_t n_bases = 0, R = 0;
    int strand_mask = 3;
    bool exact_only = false;
};
constexpr int MS_SCAN_PENDING = 1000;        // scan_locked: queued, call scan_complete
constexpr int MS_SCAN_RETRY = 1001;          // scan_complete: the prediction failed, run scan_locked(..., MS_SCAN_NO_PREDICT_INTERNAL) again
int pending_scan_init(PendingScan *p);
// a slot from the device's cache (or a new one), and back (ms_api.hip); a slot that holds an unfinished scan must not be released
PendingScan *pending_scan_acquire(DeviceCtx *c);
void pending_scan_release(DeviceCtx *c, PendingScan *p);
void pending_scan_destroy(PendingScan *p);
// waits for a pending scan; MS_OK: *out is the result; MS_SCAN_RETRY: nothing was produced.  The caller holds pwms->mu.
int scan_complete(DeviceCtx *c, ms_pwmset *pwms, PendingScan *p, ms_result **out);

// The scan pipeline proper (ms_api.hip); the caller holds c->mu and pwms->mu.  pend != nullptr: if the sizes can be predicted the
// scan is only QUEUED (returns MS_SCAN_PENDING, finish with scan_complete); otherwise it runs to the end as usual.
int scan_locked(DeviceCtx *c, ms_pwmset *pwms, const ms_seqset *seqs, int strand_mask, uint32_t flags, ms_result **out, PendingScan *pend = nullptr);
// Hand the hits of a span scan (one region) to the windows of a fixed-stride sweep, in place of *span_res (ms_api.hip);
// the caller holds c->mu and pwms->mu.
// counts_only: only the per-motif window counts and the number of sites are made (ms_result::counts_only: the hit accessors refuse it)
int sweep_handout_locked(DeviceCtx *c, ms_pwmset *pwms, ms_result *span_res, int64_t span_bases, int32_t window, int32_t stride,
                         int64_t n_windows, ms_result **out, bool counts_only = false);
int pwmset_upload(ms_pwmset *p, int device, hipStream_t st);
// One pooled device block holds everything a result owns: [counts P+1][offsets P+1][seq_idx n][pos n][score n][strand n]
size_t result_block_bytes(int32_t P, size_t n);
void result_carve(ms_result *r, void *blk, size_t n);

}  // namespace ms
