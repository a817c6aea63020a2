// ms_stream.hip -- batch streams and sweep planning (include/motifscan_amd.h, "batch streams" / "sweep planning").
//
// SURVEY.md 8(d) defines the metric over "pack + H2D + kernel + D2H of results" and 8(e) names what multi-GPU scaling
// needs: "per-GPU host threads doing pack + H2D concurrently, pinned memory, and overlap (double-buffered chunks)".  A
// stream is exactly that for one device: three host threads, one per stage, connected by bounded queues --
//     uploader     host ASCII -> HBM (upload stream) + pack kernel              ms_seqset_create
//     scanner      pre-filter -> fp64 -> order (-> sweep hand-out) (-> de-dup)   scan_locked, sweep_handout_locked
//     downloader   hit arrays -> pinned host memory (copy-out stream)           ms_result_hits_host / _packed_host
// so that batch i's copy-out, batch i+1's scan and batch i+2's upload run together.  Results leave in submission order.
// The reference has no counterpart (its Scanner holds every sequence as a Python str and makes one c_scan_motif call,
// scanner.py:71-87, 125); the batches' concatenation is what that one call returns.
#include "ms_handles.h"
#include "ms_pipeline.h"

using namespace ms;

namespace {

struct Job {
    int kind = 0;                     // 0: batch of regions (host ASCII); 1: span of a window sweep; 2: batch of regions of a resident genome
    const ms_genome *genome = nullptr;                 // kind 2 (borrowed)
    std::vector<int32_t> chrom;                        // kind 2: [n_seqs]
    std::vector<int64_t> ends;                         // kind 2: [n_seqs] (offsets holds the starts)
    const char *bases = nullptr;      // borrowed
    std::vector<int64_t> offsets;     // [n_seqs + 1]
    int64_t n_seqs = 0;
    int32_t window = 0, stride = 0;
    int64_t n_windows = 0;
    uint32_t flags = 0;               // the stream's flags | this batch's own (ms_stream_submit_counts_only adds MS_STREAM_NO_HITS)
    ms_seqset *seqs = nullptr;
    ms_result *res = nullptr;
    int rc = MS_OK;
    std::string err;
};

void fail_job(Job *j, int rc) {
    j->rc = rc;
    j->err = ms_last_error();         // the failing call left its message on THIS worker thread
}

}  // namespace

// The stream is the pipeline's Ops (ms_pipeline.h): the three stage bodies below are all that touches the device; the queues,
// the threads and the scan stage's one-scan-ahead loop live in the host-only header (and run under TSan with stub stages).
struct ms_stream {
    ms_pwmset *pwms = nullptr;
    int strand = 3;
    uint32_t flags = 0;
    int device = 0;
    int depth = 2;
    int pack_threads = 8;             // MS_STREAM_HOST_PACK: host threads of the upload stage (MS_PACK_THREADS overrides)
    void *pack_stage = nullptr;       // ... and the uploader's pinned staging block (grow-only, freed with the stream)
    size_t pack_stage_bytes = 0;
    bool upload_only = true;          // the upload stage copies, the scan stage packs (seqset_create_upload_only; MS_MEASURE=1 MS_STREAM_PACK_IN_UPLOAD=1: round 5's form, for A/B runs)
    std::unique_ptr<StagePipeline<Job, ms_stream>> pipe;

    void bind_thread() {
        set_current_device(device);
        (void) hipSetDevice(device);
        (void) numa_bind_for_device(device, false);      // the stage's thread (and the pinned blocks it allocates) on the GPU's NUMA node: ms_numa.cpp
    }

    void upload(Job *j) {
        if (j->rc != MS_OK) return;
        // (kind 2: the "upload" is the cut of the regions out of the resident 2-bit genome, on the set's own stream like a copy)
        const int rc = j->kind == 2 ? ms_seqset_from_genome(j->genome, j->chrom.data(), j->offsets.data(), j->ends.data(), j->n_seqs, &j->seqs)
                       : (flags & MS_STREAM_HOST_PACK) ? seqset_create_hostpacked(j->bases, j->offsets.data(), j->n_seqs, pack_threads, &pack_stage, &pack_stage_bytes, &j->seqs)
                       : upload_only ? seqset_create_upload_only(j->bases, j->offsets.data(), j->n_seqs, &j->seqs)
                                                       : ms_seqset_create(j->bases, j->offsets.data(), j->n_seqs, 0, &j->seqs);
        if (rc) fail_job(j, rc);
    }

    // The scan stage keeps ONE scan queued behind the one it is waiting for: a plain batch whose sizes can be predicted
    // (scan_locked with a PendingScan) is only queued; the stage then takes the next batch, queues its scan too, and only then
    // waits for the first -- the device goes from one batch's last kernel straight into the next batch's first.  Sweep spans
    // and de-duplicated batches need their result at once and run to the end as before.
    PendingScan *pend_slot[2] = {nullptr, nullptr};      // from the device's cache (pending_scan_acquire): no event / pinned allocation per stream

    bool scanner_begin() {
        DeviceCtx *c = nullptr;
        if (get_ctx(device, &c) != MS_OK) return false;
        pend_slot[0] = pending_scan_acquire(c);
        pend_slot[1] = pending_scan_acquire(c);
        return pend_slot[0] && pend_slot[1];
    }
    void scanner_end() {
        DeviceCtx *c = nullptr;
        if (get_ctx(device, &c) != MS_OK) return;
        for (PendingScan *&p : pend_slot) { pending_scan_release(c, p); p = nullptr; }
    }

    // returns true if the job's scan is pending in pend_slot[slot] (finish with scan_finish), false if the job is done (or failed)
    bool scan_start(Job *j, int slot) {
        if (j->rc != MS_OK) return false;
        const uint32_t flags = j->flags;
        DeviceCtx *c = nullptr;
        int rc = get_ctx(device, &c);
        bool pending = false;
        if (!rc) {
            std::lock_guard<std::mutex> lk_dev(c->mu);
            std::lock_guard<std::mutex> lk_pwm(pwms->mu);
            const bool simple = j->kind != 1 && !(flags & MS_STREAM_DEDUP);
            const uint32_t sf = ((flags & MS_STREAM_EXACT_ONLY) ? MS_SCAN_EXACT_ONLY : MS_SCAN_DEFAULT) |
                                ((simple && (flags & MS_STREAM_NO_HITS)) ? MS_SCAN_COUNTS_ONLY_INTERNAL : 0u) |      // nothing but the counts will be read: no ordering
                                ((simple && (flags & (MS_STREAM_PACKED | MS_STREAM_PACKED12)) && !(flags & MS_STREAM_NO_HITS)) ? ((flags & MS_STREAM_PACKED12) ? MS_SCAN_PACK12_INTERNAL : MS_SCAN_PACK_INTERNAL) : 0u);
            const bool plain = simple && slot >= 0;
            rc = scan_locked(c, pwms, j->seqs, strand, sf, &j->res, plain ? pend_slot[slot] : nullptr);
            if (rc == MS_SCAN_PENDING) { rc = MS_OK; pending = true; }
            if (!rc && !pending && j->kind == 1) {
                ms_result *r1 = j->res;
                j->res = nullptr;
                rc = sweep_handout_locked(c, pwms, r1, j->seqs->n_bases, j->window, j->stride, j->n_windows, &j->res, (flags & MS_STREAM_NO_HITS) != 0);
            }
        }
        // (a counts-only span holds no site arrays to de-duplicate, and de-duplication never empties a window: its counts stand as they are)
        if (!rc && !pending && (flags & MS_STREAM_DEDUP) && !(j->res && j->res->counts_only)) rc = ms_result_dedup(j->res, pwms);
        if (rc) fail_job(j, rc);
        if (!pending && j->seqs) { ms_seqset_free(j->seqs); j->seqs = nullptr; }
        return pending;
    }

    void scan_finish(Job *j, int slot) {
        const uint32_t flags = j->flags;
        DeviceCtx *c = nullptr;
        int rc = get_ctx(device, &c);
        if (!rc) {
            {
                std::lock_guard<std::mutex> lk_pwm(pwms->mu);
                rc = scan_complete(c, pwms, pend_slot[slot], &j->res);
            }
            if (rc == MS_SCAN_RETRY) {                       // the predicted sizes were too small: once more, exactly sized
                std::lock_guard<std::mutex> lk_dev(c->mu);
                std::lock_guard<std::mutex> lk_pwm(pwms->mu);
                const uint32_t sf = ((flags & MS_STREAM_EXACT_ONLY) ? MS_SCAN_EXACT_ONLY : MS_SCAN_DEFAULT) | MS_SCAN_NO_PREDICT_INTERNAL |
                                    ((flags & MS_STREAM_NO_HITS) ? MS_SCAN_COUNTS_ONLY_INTERNAL : 0u) |
                                    (((flags & (MS_STREAM_PACKED | MS_STREAM_PACKED12)) && !(flags & MS_STREAM_NO_HITS)) ? ((flags & MS_STREAM_PACKED12) ? MS_SCAN_PACK12_INTERNAL : MS_SCAN_PACK_INTERNAL) : 0u);
                rc = scan_locked(c, pwms, j->seqs, strand, sf, &j->res);
            }
        }
        if (rc) fail_job(j, rc);
        if (j->seqs) { ms_seqset_free(j->seqs); j->seqs = nullptr; }
    }

    void download(Job *j) {
        const uint32_t flags = j->flags;
        if (j->rc != MS_OK) return;
        { const int rc0 = result_fetch_region_counts(j->res); if (rc0) { fail_job(j, rc0); return; } }      // every batch, counts-only ones included
        if (flags & MS_STREAM_NO_HITS) return;
        // MS_STREAM_PACKED12: the 12-byte form where the scan could make it (the batch's region indices and positions fit 31 bits), and for
        // results the scan left without words (sweep spans, de-duplicated batches) if it fits them; the 16-byte form otherwise
        int rc;
        if (flags & MS_STREAM_PACKED12) {
            rc = (j->res->d_coord && !j->res->coord_shift) ? MS_ERR_INVALID : ms_result_hits_packed12_host(j->res, nullptr, nullptr, nullptr);
            if (rc == MS_ERR_INVALID) rc = ms_result_hits_packed_host(j->res, nullptr, nullptr);
        } else
            rc = (flags & MS_STREAM_PACKED) ? ms_result_hits_packed_host(j->res, nullptr, nullptr)
                                            : ms_result_hits_host(j->res, nullptr, nullptr, nullptr, nullptr);
        if (rc) fail_job(j, rc);
    }
};

static void drop_job(Job *j) {
    if (j->res) ms_result_free(j->res);
    if (j->seqs) ms_seqset_free(j->seqs);
    delete j;
}

extern "C" {

int ms_stream_create(const ms_pwmset *pwms, int strand_mask, uint32_t flags, int depth, ms_stream **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    if (!pwms) { set_error("NULL handle"); return MS_ERR_INVALID; }
    if (strand_mask < 1 || strand_mask > 3) { set_error("invalid strand mask %d (1 '+', 2 '-', 3 both)", strand_mask); return MS_ERR_INVALID; }
    if (depth < 1 || depth > 16) { set_error("depth must be in [1, 16]"); return MS_ERR_INVALID; }
    if (flags & ~(MS_STREAM_DEDUP | MS_STREAM_NO_HITS | MS_STREAM_EXACT_ONLY | MS_STREAM_PACKED | MS_STREAM_HOST_PACK | MS_STREAM_PACKED12)) { set_error("unknown stream flags 0x%x", flags); return MS_ERR_INVALID; }
    DeviceCtx *c;
    int rc = get_ctx(current_device(), &c);             // no device: fail here, loudly, not in a worker
    if (rc) return rc;
    std::unique_ptr<ms_stream> st(new (std::nothrow) ms_stream());
    if (!st) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    st->pwms = const_cast<ms_pwmset *>(pwms);
    st->strand = strand_mask;
    st->flags = flags;
    st->device = c->device;
    st->depth = depth;
    if (const char *e = getenv("MS_PACK_THREADS")) st->pack_threads = std::max(1, std::min(64, atoi(e)));
    if (const char *e = measure_env("MS_STREAM_PACK_IN_UPLOAD")) st->upload_only = !(e[0] == '1');
    try {
        st->pipe.reset(new StagePipeline<Job, ms_stream>(st.get(), depth));
        st->pipe->start();
    } catch (const std::exception &e) {
        set_error("could not start the stream's threads: %s", e.what());
        if (st->pipe) st->pipe->shutdown(drop_job);
        return MS_ERR_RUNTIME;
    }
    {
        std::lock_guard<std::mutex> lk_dev(c->mu);       // between scans: a scan sees one edition of its streams throughout
        c->n_streams.fetch_add(1);
    }
    *out = st.release();
    return MS_OK;
}

static int stream_enqueue(ms_stream *st, std::unique_ptr<Job> j, uint32_t batch_flags = 0) {
    j->flags = st->flags | batch_flags;
    if (!st->pipe->submit(j.get())) {                    // (may wait for the uploader; never for the consumer)
        set_error("%d batches are in flight: collect results with ms_stream_next first", st->pipe->capacity());
        return MS_ERR_INVALID;
    }
    j.release();
    return MS_OK;
}

static int submit_batch(ms_stream *st, const char *bases, const int64_t *offsets, int64_t n_seqs, uint32_t batch_flags);

int ms_stream_submit(ms_stream *st, const char *bases, const int64_t *offsets, int64_t n_seqs) { return submit_batch(st, bases, offsets, n_seqs, 0); }

int ms_stream_submit_counts_only(ms_stream *st, const char *bases, const int64_t *offsets, int64_t n_seqs) {
    return submit_batch(st, bases, offsets, n_seqs, MS_STREAM_NO_HITS);
}

static int submit_batch(ms_stream *st, const char *bases, const int64_t *offsets, int64_t n_seqs, uint32_t batch_flags) {
    if (!st) { set_error("NULL handle"); return MS_ERR_INVALID; }
    if (n_seqs < 0 || !offsets) { set_error("bad offsets / n_seqs"); return MS_ERR_INVALID; }
    if (offsets[n_seqs] > 0 && !bases) { set_error("bases is NULL"); return MS_ERR_INVALID; }
    std::unique_ptr<Job> j(new (std::nothrow) Job());
    if (!j) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    try { j->offsets.assign(offsets, offsets + n_seqs + 1); }
    catch (const std::bad_alloc &) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    j->bases = bases;
    j->n_seqs = n_seqs;
    return stream_enqueue(st, std::move(j), batch_flags);
}

int ms_stream_submit_regions(ms_stream *st, const ms_genome *genome, const int32_t *chrom, const int64_t *start, const int64_t *end, int64_t n_regions) {
    if (!st || !genome) { set_error("NULL handle"); return MS_ERR_INVALID; }
    if (n_regions < 0 || (n_regions > 0 && (!chrom || !start || !end))) { set_error("bad region arrays"); return MS_ERR_INVALID; }
    std::unique_ptr<Job> j(new (std::nothrow) Job());
    if (!j) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    try {
        j->chrom.assign(chrom, chrom + n_regions);
        j->offsets.assign(start, start + n_regions);
        j->ends.assign(end, end + n_regions);
    } catch (const std::bad_alloc &) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    j->kind = 2;
    j->genome = genome;
    j->n_seqs = n_regions;
    return stream_enqueue(st, std::move(j));
}

int ms_stream_submit_span(ms_stream *st, const char *bases, int64_t n_bases, int32_t window, int32_t stride) {
    if (!st) { set_error("NULL handle"); return MS_ERR_INVALID; }
    if (window < 1 || stride < 1) { set_error("window and stride must be positive"); return MS_ERR_INVALID; }
    if (n_bases < 0 || (n_bases > 0 && !bases)) { set_error("bad span"); return MS_ERR_INVALID; }
    std::unique_ptr<Job> j(new (std::nothrow) Job());
    if (!j) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    j->kind = 1;
    j->window = window;
    j->stride = stride;
    j->n_windows = n_bases >= window ? (n_bases - window) / stride + 1 : 0;
    const int64_t used = j->n_windows > 0 ? (j->n_windows - 1) * stride + window : 0;      // bases past the last whole window are not scanned
    j->offsets = {0, used};
    j->bases = bases;
    j->n_seqs = 1;
    return stream_enqueue(st, std::move(j));
}

int ms_stream_next(ms_stream *st, ms_result **out) {
    if (!st || !out) { set_error("NULL argument"); return MS_ERR_INVALID; }
    *out = nullptr;
    if (st->pipe->in_flight() == 0) return MS_OK;
    std::unique_ptr<Job> j(st->pipe->next());
    if (!j) { set_error("stream is closed"); return MS_ERR_RUNTIME; }
    if (j->rc != MS_OK) {
        set_error("%s", j->err.c_str());
        if (j->res) ms_result_free(j->res);
        return j->rc;
    }
    *out = j->res;
    return MS_OK;
}

int ms_stream_in_flight(const ms_stream *st, int *n) {
    if (!st || !n) { set_error("NULL argument"); return MS_ERR_INVALID; }
    *n = st->pipe->in_flight();
    return MS_OK;
}

int ms_stream_stats(const ms_stream *st, double out[12]) {
    if (!st || !out) { set_error("NULL argument"); return MS_ERR_INVALID; }
    for (int k = 0; k < 3; k++) {
        const StageClock &clk = st->pipe->clock(k);
        out[4 * k + 0] = (double) clk.jobs.load();
        out[4 * k + 1] = clk.work_us.load() * 1e-3;
        out[4 * k + 2] = clk.wait_in_us.load() * 1e-3;
        out[4 * k + 3] = clk.wait_out_us.load() * 1e-3;
    }
    return MS_OK;
}

int ms_stream_capacity(const ms_stream *st, int *n) {
    if (!st || !n) { set_error("NULL argument"); return MS_ERR_INVALID; }
    *n = st->pipe->capacity();
    return MS_OK;
}

void ms_stream_free(ms_stream *st) {
    if (!st) return;
    st->pipe->shutdown(drop_job);
    if (st->pack_stage) (void) hipHostFree(st->pack_stage);
    DeviceCtx *c = nullptr;
    if (get_ctx(st->device, &c) == MS_OK) {
        std::lock_guard<std::mutex> lk_dev(c->mu);
        c->n_streams.fetch_sub(1);
    }
    delete st;
}

// ---------------------------------------------------------------------- sweep planning --

int ms_sweep_spans(const int64_t *chrom_len, int32_t n_chroms, int32_t window, int32_t stride, int64_t max_span_bases,
                   ms_span *spans, int64_t cap, int64_t *n_spans) {
    if (!n_spans) { set_error("n_spans is NULL"); return MS_ERR_INVALID; }
    *n_spans = 0;
    if (n_chroms < 0 || (n_chroms > 0 && !chrom_len)) { set_error("bad chromosome lengths"); return MS_ERR_INVALID; }
    if (window < 1 || stride < 1) { set_error("window and stride must be positive"); return MS_ERR_INVALID; }
    if (max_span_bases < window) { set_error("a span must hold at least one window (%d bases)", window); return MS_ERR_INVALID; }
    if (max_span_bases > kMaxBases) max_span_bases = kMaxBases;                 // a span is one sequence set
    const int64_t per_span = (max_span_bases - window) / stride + 1;            // windows that fit one span
    int64_t first = 0, count = 0;
    for (int32_t ch = 0; ch < n_chroms; ch++) {
        const int64_t L = chrom_len[ch];
        if (L < 0) { set_error("chromosome %d has a negative length", ch); return MS_ERR_INVALID; }
        const int64_t n_w = L >= window ? (L - window) / stride + 1 : 0;
        if (n_w == 0) continue;
        const int64_t n_sp = (n_w + per_span - 1) / per_span;
        for (int64_t k = 0; k < n_sp; k++) {                                    // near-equal spans
            const int64_t k0 = n_w * k / n_sp, k1 = n_w * (k + 1) / n_sp;
            if (spans && count < cap) {
                ms_span &sp = spans[count];
                sp.chrom = ch;
                sp.reserved = 0;
                sp.begin = k0 * stride;
                sp.end = (k1 - 1) * stride + window;
                sp.first_window = first + k0;
                sp.n_windows = k1 - k0;
            }
            count++;
        }
        first += n_w;
    }
    *n_spans = count;
    return MS_OK;
}

}  // extern "C"
