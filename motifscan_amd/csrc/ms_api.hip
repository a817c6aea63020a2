// ms_api.hip -- the extern "C" boundary declared in include/motifscan_amd.h: handles, device
// memory, the scan pipeline (pre-filter -> fp64 re-score -> sort -> finalize) and its
// measurements.  One HIP stream per device owned by the library; no file-scope scan state
// (contrast cscore.c:26-34), so handles can be used from several threads / devices.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include <chrono>

#include "ms_handles.h"

namespace ms {

static thread_local std::string g_err;
static thread_local int g_device = 0;

int current_device() { return g_device; }
void set_current_device(int device) { g_device = device; }

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

// ----------------------------------------------------------------- per-device state --

static std::mutex g_ctx_mu;
static std::map<int, std::unique_ptr<DeviceCtx>> g_ctx;

int get_ctx(int device, DeviceCtx **out) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_ctx.find(device);
    if (it != g_ctx.end()) { *out = it->second.get(); MS_HIP(hipSetDevice(device)); return MS_OK; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no usable HIP device (%s); libmotifscan_amd has no CPU fallback", hipGetErrorString(e));
        return MS_ERR_RUNTIME;
    }
    if (device < 0 || device >= n) { set_error("device %d out of range (%d devices)", device, n); return MS_ERR_INVALID; }
    MS_HIP(hipSetDevice(device));
    std::unique_ptr<DeviceCtx> c(new DeviceCtx());
    c->device = device;
    hipDeviceProp_t prop;
    MS_HIP(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int lds_attr = 0;
    if (hipDeviceGetAttribute(&lds_attr, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) lds_attr = 0;
    c->lds_max = std::max<size_t>((size_t) lds_attr, prop.sharedMemPerBlock);
    if (c->lds_max < 65536) c->lds_max = 65536;
    if (c->lds_max > 163840) c->lds_max = 163840;
    MS_HIP(hipStreamCreateWithFlags(&c->stream.whole, hipStreamNonBlocking));
    {
        // MS_MEASURE=1 MS_COPY_PRIORITY=1 (A/B): the copy streams at the device's highest priority -- a copy-out that runs as a blit kernel then wins freed CUs
        // over the pre-filter's pending blocks
        int lo = 0, hi = 0;
        const bool prio = measure_env("MS_COPY_PRIORITY") && atoi(measure_env("MS_COPY_PRIORITY")) != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo;
        if (prio) {
            MS_HIP(hipStreamCreateWithPriority(&c->stream_up.whole, hipStreamNonBlocking, hi));
            MS_HIP(hipStreamCreateWithPriority(&c->stream_down.whole, hipStreamNonBlocking, hi));
        } else {
            MS_HIP(hipStreamCreateWithFlags(&c->stream_up.whole, hipStreamNonBlocking));
            MS_HIP(hipStreamCreateWithFlags(&c->stream_down.whole, hipStreamNonBlocking));
        }
    }
    for (StreamSel *s : {&c->stream, &c->stream_up, &c->stream_down}) s->n_streams = &c->n_streams;
    if (measure_env("MS_CU_PARTITION")) {
        // A/B switch (MS_MEASURE=1 MS_CU_PARTITION=1); off by default, see StreamSel.
        // CU mask bit i = CU (i / n_xcc) of XCC (i % n_xcc) (measured, tools/ubench/cumask_probe.hip: the first 8 bits select one
        // CU on each of the 8 XCCs; an XCC without a bit is left unmasked, so the copy share must cover every XCC): the first
        // n_cu / 32 bits go to the copy streams, the rest to the scan.
        const int k = std::max(1, c->n_cu / 32);
        std::vector<uint32_t> m_copy((size_t) (c->n_cu + 31) / 32, 0u), m_scan((size_t) (c->n_cu + 31) / 32, 0u);
        for (int i = 0; i < c->n_cu; i++) (i < k ? m_copy : m_scan)[(size_t) i / 32] |= 1u << (i % 32);
        bool ok = c->n_cu >= 64;
        ok = ok && hipExtStreamCreateWithCUMask(&c->stream.part, (uint32_t) m_scan.size(), m_scan.data()) == hipSuccess;
        ok = ok && hipExtStreamCreateWithCUMask(&c->stream_up.part, (uint32_t) m_copy.size(), m_copy.data()) == hipSuccess;
        ok = ok && hipExtStreamCreateWithCUMask(&c->stream_down.part, (uint32_t) m_copy.size(), m_copy.data()) == hipSuccess;
        if (ok) {
            c->n_cu_copy = k;
        } else {                                   // no masks on this device / runtime: everything stays on the whole-device streams
            (void) hipGetLastError();
            for (StreamSel *s : {&c->stream, &c->stream_up, &c->stream_down}) {
                if (s->part) (void) hipStreamDestroy(s->part);
                s->part = nullptr;
            }
        }
    }
    MS_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    MS_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    for (auto &ev : c->ev) MS_HIP(hipEventCreate(&ev));
    MS_HIP(hipMalloc(&c->sc.counters, 8 * sizeof(unsigned long long)));
    MS_HIP(hipHostMalloc(&c->sc.h_counters, 8 * sizeof(unsigned long long)));
    {
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && mem_total > 0) c->pool.max_bytes = mem_total / 3;
        else (void) hipGetLastError();
    }
    *out = c.get();
    g_ctx[device] = std::move(c);
    return MS_OK;
}

// Size class of a request: the next of {8..15} x 2^k at or above it (at most 12.5 % over), 64 KB at least.
static size_t pool_class(size_t bytes) {
    size_t b = std::max<size_t>(bytes, 1u << 16);
    int k = 63 - __builtin_clzll((unsigned long long) b);          // 2^k <= b
    const size_t step = (size_t) 1 << (k - 3);
    return (b + step - 1) & ~(step - 1);
}

static uint64_t now_ns() {
    return (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int pool_alloc(DeviceCtx *c, size_t bytes, void **out, size_t *got) {
    const size_t want = pool_class(bytes);
    {
        std::lock_guard<std::mutex> lk(c->pool.mu);
        size_t best = (size_t) -1;
        for (size_t i = 0; i < c->pool.free_.size(); i++) {
            const size_t sz = c->pool.free_[i].second;
            if (sz >= want && sz <= want + want / 2 && (best == (size_t) -1 || sz < c->pool.free_[best].second)) best = i;
        }
        if (best != (size_t) -1) {
            *out = c->pool.free_[best].first;
            *got = c->pool.free_[best].second;
            c->pool.bytes -= *got;
            c->pool.free_.erase(c->pool.free_.begin() + (long) best);
            c->pool.n_hit++;
            return MS_OK;
        }
    }
    const uint64_t t0 = now_ns();
    char *p = nullptr;
    int rc = dev_alloc(&p, want);               // (dev_alloc itself drops the cache and retries once when the device is full)
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(c->pool.mu);
        c->pool.n_miss++;
        c->pool.ns_driver += now_ns() - t0;
    }
    *out = p;
    *got = want;
    return MS_OK;
}

// Pinned host blocks are even dearer to create than device blocks (page-locking ~0.25 ms per MB, and hipHostFree waits for the whole device):
// kept for reuse in the device pool's SIZE CLASSES (round 6).  Rounds 2-5 matched a request to any free block of 1 ... 2 x its size, at most 16
// blocks: the 14 batches of a configs[3] pass ask for 14 different sizes, a small request took the block a larger one needed, the largest
// went to the driver, the list ran over and a block went back to the driver -- episodes of 80-ms passes among 48-ms ones
// (profiles/r06f_e2e_cli_probe.log).  With classes a pass's blocks come back to exactly the requests that made them.
static std::mutex g_pin_mu;
static std::vector<std::pair<void *, size_t>> g_pin_free;
static uint64_t g_pin_stats[4] = {0, 0, 0, 0};            // served from the list, went to hipHostMalloc, returned to the driver, ns inside the driver

void *pinned_alloc(size_t bytes, size_t *got) {
    const size_t want = pool_class(bytes);
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = g_pin_free.size(); i-- > 0;)
            if (g_pin_free[i].second == want) {
                void *p = g_pin_free[i].first;
                *got = want;
                g_pin_free.erase(g_pin_free.begin() + (long) i);
                g_pin_stats[0]++;
                return p;
            }
    }
    void *p = nullptr;
    const uint64_t t0 = now_ns();
    const bool ok = hipHostMalloc(&p, want) == hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        g_pin_stats[1]++;
        g_pin_stats[3] += now_ns() - t0;
    }
    if (!ok) { (void) hipGetLastError(); return nullptr; }
    *got = want;
    return p;
}

void pinned_free(void *p, size_t bytes) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        size_t total = bytes;
        for (auto &b : g_pin_free) total += b.second;
        if (g_pin_free.size() < 96 && total <= (24ull << 30)) { g_pin_free.emplace_back(p, bytes); return; }
    }
    const uint64_t t0 = now_ns();
    (void) hipHostFree(p);
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_stats[2]++;
    g_pin_stats[3] += now_ns() - t0;
}

void pinned_pool_stats(uint64_t out[4]) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (int i = 0; i < 4; i++) out[i] = g_pin_stats[i];
}

size_t pool_trim_current_device() {
    DeviceCtx *c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        auto it = g_ctx.find(g_device);
        if (it == g_ctx.end()) return 0;
        c = it->second.get();
    }
    std::vector<std::pair<void *, size_t>> victims;
    {
        std::lock_guard<std::mutex> lk(c->pool.mu);
        victims.swap(c->pool.free_);
        c->pool.bytes = 0;
        c->pool.n_driver_free += victims.size();
    }
    size_t freed = 0;
    for (auto &b : victims) { (void) hipFree(b.first); freed += b.second; }
    return freed;
}

void pool_free(DeviceCtx *c, void *p, size_t bytes) {
    if (!p) return;
    std::unique_lock<std::mutex> lk(c->pool.mu);
    if (c->pool.bytes + bytes <= c->pool.max_bytes && c->pool.free_.size() < BlockPool::kMaxBlocks) {
        c->pool.free_.emplace_back(p, bytes);
        c->pool.bytes += bytes;
        return;
    }
    // full.  By BYTES (the cache holds a third of the device): give the driver the SMALLEST cached block that is smaller than this one, or
    // this one -- the large blocks are the dear ones to make again (~100 ms per GB-sized hipMalloc), and a sweep that cycles through more
    // result blocks than fit keeps missing the cheap ones, not the dear ones (an LRU rule, tried in round 5, cost the 3 Gbp sweep with all
    // sites copied out 300 ms of driver time per pass: profiles/r05_bench_c5_3000mbp.json against r04e_).  By COUNT (kMaxBlocks, 512 since
    // round 5: 64 large blocks of a sweep once filled the list and every block of a later stream of small batches went back to the driver
    // and came from hipMalloc again, forever): the stalest block leaves -- the list is in order of return, its front is the stalest.
    std::vector<void *> victims;
    void *keep = p;
    if (c->pool.free_.size() >= BlockPool::kMaxBlocks) {
        victims.push_back(c->pool.free_.front().first);
        c->pool.bytes -= c->pool.free_.front().second;
        c->pool.free_.erase(c->pool.free_.begin());
    }
    if (c->pool.bytes + bytes > c->pool.max_bytes) {
        size_t small = (size_t) -1;
        for (size_t i = 0; i < c->pool.free_.size(); i++)
            if (c->pool.free_[i].second < bytes && (small == (size_t) -1 || c->pool.free_[i].second < c->pool.free_[small].second)) small = i;
        if (small != (size_t) -1 && c->pool.bytes - c->pool.free_[small].second + bytes <= c->pool.max_bytes) {
            victims.push_back(c->pool.free_[small].first);
            c->pool.bytes -= c->pool.free_[small].second;
            c->pool.free_.erase(c->pool.free_.begin() + (long) small);
        } else {
            victims.push_back(p);
            keep = nullptr;
        }
    }
    if (keep) {
        c->pool.free_.emplace_back(keep, bytes);
        c->pool.bytes += bytes;
    }
    c->pool.n_driver_free += victims.size();
    lk.unlock();
    const uint64_t t0 = now_ns();
    for (void *v : victims) (void) hipFree(v);
    lk.lock();
    c->pool.ns_driver += now_ns() - t0;
}

}  // namespace ms

using namespace ms;

// ------------------------------------------------------------------------- handles --

// Lay the result arrays out in one device block: [counts P+1][offsets P+1][seq_idx n][pos n][score n][strand n]
size_t ms::result_block_bytes(int32_t P, size_t n) {
    const size_t n_round = (n + 65535) & ~(size_t) 65535;
    return 8 * (2 * ((size_t) P + 1) + 3 * n_round) + n_round + 256;
}

void ms::result_carve(ms_result *r, void *blk, size_t n) {
    const size_t P1 = (size_t) r->P + 1, n_round = (n + 65535) & ~(size_t) 65535;
    char *b = static_cast<char *>(blk);
    r->d_region_counts = reinterpret_cast<unsigned long long *>(b);
    r->d_motif_first = reinterpret_cast<int64_t *>(b + 8 * P1);
    r->d_seq_idx = reinterpret_cast<int64_t *>(b + 16 * P1);
    r->d_pos = r->d_seq_idx + n_round;
    r->d_score = reinterpret_cast<double *>(r->d_pos + n_round);
    r->d_strand = reinterpret_cast<int8_t *>(r->d_score + n_round);
}

// Host loops over tens of millions of regions (whole-genome window sweeps) are split over a few threads.
template <typename F>
static void parallel_chunks(int64_t n, F fn) {                 // fn(chunk index, begin, end)
    int T = 1;
    if (n >= (1 << 20)) T = (int) std::min<int64_t>(16, std::max<int64_t>(1, (int64_t) std::thread::hardware_concurrency()));
    if (T <= 1) { fn(0, (int64_t) 0, n); return; }
    std::vector<std::thread> th;
    const int64_t step = (n + T - 1) / T;
    for (int t = 0; t < T; t++) {
        const int64_t b = std::min<int64_t>(n, t * step), e = std::min<int64_t>(n, (t + 1) * step);
        th.emplace_back([=]() { fn(t, b, e); });
    }
    for (auto &x : th) x.join();
}
static constexpr int kMaxChunks = 16;

// C-style max_raw: column maxima start at 0 (cscore.c:36-48)
static double c_max_raw(const double *m, int W) {
    double total = 0;
    for (int c = 0; c < W; c++) {
        double best = 0;
        for (int b = 0; b < 4; b++)
            if (m[(int64_t) b * W + c] > best) best = m[(int64_t) b * W + c];
        total += best;
    }
    return total;
}

static void pwmset_free_device(ms_pwmset *p) {
    if (p->device >= 0 || p->plan_device >= 0) (void) hipSetDevice(p->device >= 0 ? p->device : p->plan_device);
    dev_free(p->d_tab2); dev_free(p->d_tab_off); dev_free(p->d_width); dev_free(p->d_max_raw); dev_free(p->d_cutoff); dev_free(p->d_raw_floor); dev_free(p->d_thresh);
    dev_free(p->d_tables); dev_free(p->d_tiles); dev_free(p->d_group_fields); dev_free(p->d_exact_motifs); dev_free(p->d_field_meta);
    p->device = -1;
    p->plan_device = -1;
    p->dev_cutoff_version = 0;
}

int ms::pwmset_upload(ms_pwmset *p, int device, hipStream_t st) {
    if (p->device != device) {
        pwmset_free_device(p);
        MS_HIP(hipSetDevice(device));
        size_t total_w = 0;
        for (int32_t i = 0; i < p->P; i++) total_w += (size_t) p->widths[i];
        std::vector<double2> tab(total_w * 4 + 1);       // (+ one all-zero entry at the end: what a column that adds nothing reads, DevPwm::zero_bytes)
        tab[total_w * 4].x = 0.0;
        tab[total_w * 4].y = 0.0;
        std::vector<int64_t> off(p->P);
        size_t o = 0;
        for (int32_t i = 0; i < p->P; i++) {
            const int W = p->widths[i];
            const double *m = p->values.data() + p->val_off[i];
            off[i] = (int64_t) o;
            for (int c = 0; c < W; c++)
                for (int b = 0; b < 4; b++) {
                    double2 t;
                    t.x = m[(int64_t) b * W + c];
                    t.y = m[(int64_t) (3 - b) * W + (W - 1 - c)];       // cscore.c:351
                    tab[o + (size_t) c * 4 + b] = t;
                }
            o += (size_t) W * 4;
        }
        p->tab2_entries = (int64_t) total_w * 4;
        p->tab_off_host = off;
        int rc;
        if ((rc = dev_alloc(&p->d_tab2, tab.size()))) return rc;
        if ((rc = dev_alloc(&p->d_tab_off, (size_t) p->P))) return rc;
        if ((rc = dev_alloc(&p->d_width, (size_t) p->P))) return rc;
        if ((rc = dev_alloc(&p->d_max_raw, (size_t) p->P))) return rc;
        if ((rc = dev_alloc(&p->d_cutoff, (size_t) p->P))) return rc;
        if ((rc = dev_alloc(&p->d_raw_floor, (size_t) p->P))) return rc;
        if ((rc = dev_alloc(&p->d_thresh, (size_t) p->P * 4 + 4))) return rc;
        if (p->P > 0) {
            MS_HIP(hipMemcpy(p->d_tab2, tab.data(), tab.size() * sizeof(double2), hipMemcpyHostToDevice));
            MS_HIP(hipMemcpy(p->d_tab_off, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
            MS_HIP(hipMemcpy(p->d_width, p->widths.data(), (size_t) p->P * sizeof(int32_t), hipMemcpyHostToDevice));
            MS_HIP(hipMemcpy(p->d_max_raw, p->max_raw.data(), (size_t) p->P * sizeof(double), hipMemcpyHostToDevice));
        }
        p->device = device;
        p->dev_cutoff_version = 0;
    }
    if (p->dev_cutoff_version != p->cutoff_version) {
        if (p->P > 0) {
            MS_HIP(hipMemcpy(p->d_cutoff, p->cutoffs.data(), (size_t) p->P * sizeof(double), hipMemcpyHostToDevice));
            // raw-sum floor of the hit test (same bound as ms_plan.cpp's T, with twice its slack): a window whose
            // fp64 column sum is below it fails `sum / max_raw - cutoff >= -1e-10` for sure
            std::vector<double> fl((size_t) p->P);
            for (int32_t i = 0; i < p->P; i++) {
                const int W = p->widths[i];
                const double *m = p->values.data() + p->val_off[i];
                double abs_sum = 0;
                bool finite = std::isfinite(p->max_raw[i]) && p->max_raw[i] > 0 && std::isfinite(p->cutoffs[i]);
                for (int c = 0; c < W && finite; c++) {
                    double colmax = 0;
                    for (int b = 0; b < 4; b++) {
                        const double v = m[(int64_t) b * W + c];
                        if (!std::isfinite(v)) { finite = false; break; }
                        colmax = std::max(colmax, std::fabs(v));
                    }
                    abs_sum += colmax;
                }
                fl[(size_t) i] = finite ? (p->cutoffs[i] - 1e-10) * p->max_raw[i] - 2e-9 * (1.0 + abs_sum) : -INFINITY;
            }
            MS_HIP(hipMemcpy(p->d_raw_floor, fl.data(), (size_t) p->P * sizeof(double), hipMemcpyHostToDevice));
            std::vector<double> th((size_t) p->P * 4, 0.0);              // the hit test's three numbers side by side (rescore_kernel: one 16-byte + one 8-byte read)
            for (int32_t i = 0; i < p->P; i++) { th[4 * (size_t) i] = p->max_raw[i]; th[4 * (size_t) i + 1] = p->cutoffs[i]; th[4 * (size_t) i + 2] = fl[(size_t) i]; }
            p->raw_floor_host = fl;
            MS_HIP(hipMemcpy(p->d_thresh, th.data(), th.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        p->dev_cutoff_version = p->cutoff_version;
    }
    (void) st;
    return MS_OK;
}

static int pwmset_plan(ms_pwmset *p, int strand_mask, size_t lds_budget, bool exact_only, bool need_device,
                       int device) {
    const char *pe = measure_env("MS_PF_PAIR");                          // measurement only: "0" = no paired rows
    const bool pair_rows = !(pe && pe[0] == '0');
    const bool stale = p->plan_strand != strand_mask || p->plan_cutoff_version != p->cutoff_version ||
                       p->plan_lds != lds_budget || p->plan_exact_only != exact_only || p->plan_pair != pair_rows;
    if (stale) {
        if (exact_only) {
            p->plan = PrefilterPlan();
            p->plan.strand_mask = strand_mask;
            for (int32_t i = 0; i < p->P; i++) p->plan.exact_motifs.push_back(i);
        } else {
            int rc = build_plan(p->values.data(), p->val_off.data(), p->widths.data(), p->cutoffs.data(), p->max_raw.data(),
                                p->P, strand_mask, lds_budget, pair_rows, &p->plan);
            if (rc) return rc;
        }
        p->plan_strand = strand_mask;
        p->plan_pair = pair_rows;
        p->plan_cutoff_version = p->cutoff_version;
        p->plan_lds = lds_budget;
        p->plan_exact_only = exact_only;
        if (p->plan_device >= 0) {
            (void) hipSetDevice(p->plan_device);
            dev_free(p->d_tables); dev_free(p->d_tiles); dev_free(p->d_group_fields); dev_free(p->d_exact_motifs); dev_free(p->d_field_meta);
            p->plan_device = -1;
        }
    }
    if (need_device && p->plan_device != device) {
        MS_HIP(hipSetDevice(device));
        const PrefilterPlan &pl = p->plan;
        int rc;
        if ((rc = dev_alloc(&p->d_tables, pl.tables.size() / 4))) return rc;
        if ((rc = dev_alloc(&p->d_tiles, pl.tiles.size()))) return rc;
        if ((rc = dev_alloc(&p->d_group_fields, pl.group_fields.size()))) return rc;
        if ((rc = dev_alloc(&p->d_exact_motifs, pl.exact_motifs.size()))) return rc;
        if (!pl.tables.empty())
            MS_HIP(hipMemcpy(p->d_tables, pl.tables.data(), pl.tables.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        if (!pl.tiles.empty())
            MS_HIP(hipMemcpy(p->d_tiles, pl.tiles.data(), pl.tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
        if (!pl.group_fields.empty())
            MS_HIP(hipMemcpy(p->d_group_fields, pl.group_fields.data(), pl.group_fields.size() * sizeof(int32_t),
                             hipMemcpyHostToDevice));
        if (!pl.exact_motifs.empty())
            MS_HIP(hipMemcpy(p->d_exact_motifs, pl.exact_motifs.data(), pl.exact_motifs.size() * sizeof(int32_t),
                             hipMemcpyHostToDevice));
        {   // per field of every table group: motif, width, table offset (pwmset_upload has run: scan_locked's order)
            std::vector<FieldMeta> fmv(pl.group_fields.size());
            for (size_t i = 0; i < fmv.size(); i++) {
                const int32_t m = pl.group_fields[i];
                fmv[i].motif = m;
                fmv[i].width = m >= 0 ? p->widths[m] : 0;
                fmv[i].tab_bytes = m >= 0 && (size_t) m < p->tab_off_host.size() ? (uint32_t) ((uint64_t) p->tab_off_host[m] * sizeof(double2)) : 0u;
                float f32 = -INFINITY;                          // rounded DOWN: never above the fp64 floor
                if (m >= 0 && (size_t) m < p->raw_floor_host.size() && std::isfinite(p->raw_floor_host[m])) {
                    f32 = (float) p->raw_floor_host[m];
                    if ((double) f32 > p->raw_floor_host[m] || !std::isfinite(f32)) f32 = std::isfinite(f32) ? std::nextafterf(f32, -INFINITY) : -INFINITY;
                }
                fmv[i].floor32 = f32;
            }
            if ((rc = dev_alloc(&p->d_field_meta, fmv.size() + 1))) return rc;
            if (!fmv.empty()) MS_HIP(hipMemcpy(p->d_field_meta, fmv.data(), fmv.size() * sizeof(FieldMeta), hipMemcpyHostToDevice));
        }
        p->plan_device = device;
    }
    return MS_OK;
}

static DevPwm dev_pwm(const ms_pwmset *p) {
    DevPwm d;
    d.tab2 = p->d_tab2; d.tab_off = p->d_tab_off; d.width = p->d_width; d.max_raw = p->d_max_raw;
    d.thresh = p->d_thresh;
    d.zero_bytes = (uint32_t) ((uint64_t) p->tab2_entries * sizeof(double2));
    d.tab32 = (uint64_t) (p->tab2_entries + 1) * sizeof(double2) <= 0xFFFFFFFFull ? 1 : 0;
    d.cutoff = p->d_cutoff; d.raw_floor = p->d_raw_floor; d.P = p->P;
    return d;
}

static DevSeq dev_seq(const ms_seqset *s) {
    DevSeq d;
    d.codes = s->d_codes; d.nmask = s->d_nmask; d.offsets = s->d_offsets; d.blk2reg = s->d_blk2reg; d.blkinfo = s->d_blkinfo; d.R = s->R;
    d.n_bases = s->n_bases;
    return d;
}

// sum_r max(L_r - W + 1, 0) from the sorted lengths
static int64_t windows_for_width(const ms_seqset *s, int W) {
    const auto it = std::lower_bound(s->len_sorted.begin(), s->len_sorted.end(), (int64_t) W);
    const size_t idx = (size_t) (it - s->len_sorted.begin());
    return s->len_sum_ge[idx] - (int64_t) (W - 1) * s->len_cnt_ge[idx];
}

// =============================================================================== API ==

extern "C" {

const char *ms_last_error(void) { return g_err.c_str(); }

int ms_version(void) { return 100; }

int ms_device_count(int *count) {
    if (!count) { set_error("count is NULL"); return MS_ERR_INVALID; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return MS_ERR_RUNTIME; }
    *count = n;
    return MS_OK;
}

int ms_set_device(int device) {
    DeviceCtx *c;
    int rc = get_ctx(device, &c);
    if (rc) return rc;
    g_device = device;
    return MS_OK;
}

}  // extern "C"

// MS_NUMA_BIND policy (ms_numa.cpp) for the calling thread and `device`; the node is looked up once per device
int ms::numa_bind_for_device(int device, bool force) {
    static std::mutex mu;
    static std::map<int, int> node_of;                      // device -> node (-1 unknown)
    static int policy = -2;                                 // -2 unread, 0 never, 1 always, 2 auto
    static int n_nodes = 0, n_gpus = 0;
    int node = -1;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (policy == -2) {
            const char *e = getenv("MS_NUMA_BIND");
            policy = !e ? 2 : (e[0] == '0' ? 0 : 1);
            n_nodes = numa_node_count("");
            if (hipGetDeviceCount(&n_gpus) != hipSuccess) { n_gpus = 0; (void) hipGetLastError(); }
        }
        if (!force && (policy == 0 || (policy == 2 && !(n_nodes > 1 && n_gpus > 1)))) return -1;
        auto it = node_of.find(device);
        if (it == node_of.end()) {
            char bdf[64] = {0};
            int nd = -1;
            if (hipDeviceGetPCIBusId(bdf, (int) sizeof(bdf), device) == hipSuccess) nd = numa_node_of_bdf(bdf, "");
            else (void) hipGetLastError();
            it = node_of.emplace(device, nd).first;
        }
        node = it->second;
    }
    if (node < 0) return -1;
    return numa_bind_calling_thread(node) > 0 ? node : -1;
}

extern "C" {

// Bind the CALLING thread to the CPUs of the NUMA node the calling thread's device hangs off (memory it allocates afterwards -- pinned
// buffers included -- is placed there by first touch).  force = 0: only where the policy says so (MS_NUMA_BIND; default: multi-GPU nodes
// with more than one NUMA node); force != 0: always.  *node = the node bound to, -1 if nothing was done (no NUMA information, policy off).
int ms_numa_bind_thread(int force, int *node) {
    DeviceCtx *c;
    int rc = get_ctx(g_device, &c);
    if (rc) return rc;
    const int nd = numa_bind_for_device(c->device, force != 0);
    if (node) *node = nd;
    return MS_OK;
}

int ms_device_name(char *buf, int buflen) {
    if (!buf || buflen <= 0) { set_error("bad buffer"); return MS_ERR_INVALID; }
    DeviceCtx *c;
    int rc = get_ctx(g_device, &c);
    if (rc) return rc;
    hipDeviceProp_t prop;
    MS_HIP(hipGetDeviceProperties(&prop, g_device));
    snprintf(buf, (size_t) buflen, "%s (%s, %d CUs, %zu B LDS/block)", prop.name, prop.gcnArchName, c->n_cu, c->lds_max);
    return MS_OK;
}

// ------------------------------------------------------------------------- PWM set --

int ms_pwmset_create(const double *values, const int32_t *widths, const double *cutoffs, int32_t n_pwms,
                     ms_pwmset **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    if (n_pwms < 0 || n_pwms > kMaxMotifs) { set_error("n_pwms must be in [0, %d]", kMaxMotifs); return MS_ERR_INVALID; }
    if (n_pwms > 0 && (!values || !widths)) { set_error("values / widths is NULL"); return MS_ERR_INVALID; }
    std::unique_ptr<ms_pwmset> p(new (std::nothrow) ms_pwmset());
    if (!p) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    p->P = n_pwms;
    p->val_off.assign((size_t) n_pwms + 1, 0);
    for (int32_t i = 0; i < n_pwms; i++) {
        if (widths[i] < 1) { set_error("PWM %d has width %d (need >= 1 position per row)", i, widths[i]); return MS_ERR_INVALID; }
        p->val_off[i + 1] = p->val_off[i] + 4 * (int64_t) widths[i];
        p->max_width = std::max(p->max_width, (int) widths[i]);
    }
    try {
        p->values.assign(values, values + p->val_off[n_pwms]);
        p->widths.assign(widths, widths + n_pwms);
        p->cutoffs.assign((size_t) n_pwms, 1.0);                       // cscore.c:70-74
        if (cutoffs) p->cutoffs.assign(cutoffs, cutoffs + n_pwms);
        p->max_raw.resize((size_t) n_pwms);
    } catch (const std::bad_alloc &) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    for (int32_t i = 0; i < n_pwms; i++) p->max_raw[i] = c_max_raw(p->values.data() + p->val_off[i], widths[i]);
    *out = p.release();
    return MS_OK;
}

int ms_pwmset_set_cutoffs(ms_pwmset *p, const double *cutoffs) {
    if (!p || (!cutoffs && p->P > 0)) { set_error("NULL argument"); return MS_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->P > 0) p->cutoffs.assign(cutoffs, cutoffs + p->P);
    p->cutoff_version++;
    return MS_OK;
}

int ms_pwmset_size(const ms_pwmset *p, int32_t *n_pwms) {
    if (!p || !n_pwms) { set_error("NULL argument"); return MS_ERR_INVALID; }
    *n_pwms = p->P;
    return MS_OK;
}

int ms_pwmset_max_raw(const ms_pwmset *p, double *out) {
    if (!p || (!out && p->P > 0)) { set_error("NULL argument"); return MS_ERR_INVALID; }
    if (p->P > 0) std::memcpy(out, p->max_raw.data(), (size_t) p->P * sizeof(double));
    return MS_OK;
}

void ms_pwmset_free(ms_pwmset *p) {
    if (!p) return;
    pwmset_free_device(p);
    delete p;
}

// -------------------------------------------------------------------- sequence set --

static int seqset_common(const int64_t *offsets, int64_t n_seqs, std::unique_ptr<ms_seqset> &s,
                         std::vector<int64_t> *adopt = nullptr) {
    if (n_seqs < 0) { set_error("n_seqs < 0"); return MS_ERR_INVALID; }
    if (!offsets) { set_error("offsets is NULL"); return MS_ERR_INVALID; }
    if (offsets[0] != 0) { set_error("offsets[0] must be 0"); return MS_ERR_INVALID; }
    // one pass: monotonicity, shortest and longest sequence
    int64_t cmin[kMaxChunks], cmax[kMaxChunks], cbad[kMaxChunks];
    for (int t = 0; t < kMaxChunks; t++) { cmin[t] = INT64_MAX; cmax[t] = 0; cbad[t] = -1; }
    parallel_chunks(n_seqs, [&](int t, int64_t b, int64_t e) {
        int64_t mn = INT64_MAX, mx = 0, bad = -1;
        for (int64_t r = b; r < e; r++) {
            const int64_t L = offsets[r + 1] - offsets[r];
            if (L < 0 && bad < 0) bad = r;
            mn = std::min(mn, L);
            mx = std::max(mx, L);
        }
        cmin[t] = mn; cmax[t] = mx; cbad[t] = bad;
    });
    int64_t lmin = INT64_MAX, lmax = 0;
    for (int t = 0; t < kMaxChunks; t++) {
        if (cbad[t] >= 0) { set_error("offsets must be non-decreasing (at %lld)", (long long) cbad[t]); return MS_ERR_INVALID; }
        lmin = std::min(lmin, cmin[t]);
        lmax = std::max(lmax, cmax[t]);
    }
    s.reset(new (std::nothrow) ms_seqset());
    if (!s) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    s->device = g_device;
    s->R = n_seqs;
    s->n_bases = offsets[n_seqs];
    if (s->n_bases > kMaxBases) { set_error("a sequence set holds at most %lld bases; split the regions over several sets", (long long) kMaxBases); return MS_ERR_INVALID; }
    if (n_seqs >= (1LL << 31)) { set_error("too many sequences in one set"); return MS_ERR_INVALID; }
    try {
        if (adopt) s->offsets.swap(*adopt);
        else s->offsets.assign(offsets, offsets + n_seqs + 1);
        offsets = s->offsets.data();
        // distinct lengths with counts: direct histograms when the lengths span a small range
        // (fixed-size windows: one bin), a sort otherwise
        std::vector<std::pair<int64_t, int64_t>> runs;           // (length, count) ascending
        if (n_seqs > 0 && lmax - lmin <= (1 << 16)) {
            const size_t bins = (size_t) (lmax - lmin + 1);
            std::vector<std::vector<int64_t>> hist(kMaxChunks);
            parallel_chunks(n_seqs, [&](int t, int64_t b, int64_t e) {
                hist[(size_t) t].assign(bins, 0);
                for (int64_t r = b; r < e; r++) hist[(size_t) t][(size_t) (offsets[r + 1] - offsets[r] - lmin)]++;
            });
            for (size_t i = 0; i < bins; i++) {
                int64_t cnt = 0;
                for (int t = 0; t < kMaxChunks; t++) if (!hist[(size_t) t].empty()) cnt += hist[(size_t) t][i];
                if (cnt) runs.emplace_back(lmin + (int64_t) i, cnt);
            }
        } else if (n_seqs > 0) {
            std::vector<int64_t> lens((size_t) n_seqs);
            for (int64_t r = 0; r < n_seqs; r++) lens[(size_t) r] = offsets[r + 1] - offsets[r];
            std::sort(lens.begin(), lens.end());
            for (size_t i = 0; i < lens.size();) {
                size_t j = i;
                while (j < lens.size() && lens[j] == lens[i]) j++;
                runs.emplace_back(lens[i], (int64_t) (j - i));
                i = j;
            }
        }
        const size_t nd = runs.size();
        s->len_sorted.resize(nd);
        s->len_cnt_ge.assign(nd + 1, 0);
        s->len_sum_ge.assign(nd + 1, 0);
        for (size_t i = nd; i-- > 0;) {
            s->len_sorted[i] = runs[i].first;
            s->len_cnt_ge[i] = s->len_cnt_ge[i + 1] + runs[i].second;
            s->len_sum_ge[i] = s->len_sum_ge[i + 1] + runs[i].first * runs[i].second;
        }
    } catch (const std::bad_alloc &) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    return MS_OK;
}

static int seqset_alloc_packed(ms_seqset *s, bool pads_by_copy = false, bool defer_pads = false) {
    const size_t n_units = (size_t) ((s->n_bases + 31) / 32);
    const size_t b_codes = (2 * n_units + kPadWords) * sizeof(uint32_t);
    const size_t b_nmask = (n_units + kPadWords) * sizeof(uint32_t);
    const size_t b_off = ((size_t) s->R + 1) * sizeof(int64_t);
    const size_t b_blk = ((size_t) (s->n_bases / 64) + 2) * sizeof(int32_t);
    const size_t b_info = ((size_t) (s->n_bases / 64) + 2) * sizeof(int4);
    auto up = [](size_t x) { return (x + 255) & ~(size_t) 255; };
    DeviceCtx *c;
    int rc = get_ctx(s->device, &c);
    if (rc) return rc;
    if ((rc = pool_alloc(c, up(b_codes) + up(b_nmask) + up(b_off) + up(b_blk) + up(b_info), &s->block, &s->block_bytes))) return rc;
    char *b = static_cast<char *>(s->block);
    s->d_codes = reinterpret_cast<uint32_t *>(b);
    s->d_nmask = reinterpret_cast<uint32_t *>(b + up(b_codes));
    s->d_offsets = reinterpret_cast<int64_t *>(b + up(b_codes) + up(b_nmask));
    s->d_blk2reg = reinterpret_cast<int32_t *>(b + up(b_codes) + up(b_nmask) + up(b_off));
    s->d_blkinfo = reinterpret_cast<int4 *>(b + up(b_codes) + up(b_nmask) + up(b_off) + up(b_blk));
    // only the pad words behind the packed data need clearing: the kernels write everything else
    // sequence sets are built on the upload stream: a batch can be packed while the previous one is being scanned
    s->up = c->stream_up;                                 // read once: every step of building this set stays on one stream
    if (!pads_by_copy && !defer_pads) {                   // (a memset is a KERNEL: the host-packed form copies zero pad words along instead, the upload-only form clears them when it packs)
        MS_HIP(hipMemsetAsync(s->d_codes + 2 * n_units, 0, kPadWords * sizeof(uint32_t), s->up));
        MS_HIP(hipMemsetAsync(s->d_nmask + n_units, 0, kPadWords * sizeof(uint32_t), s->up));
    }
    if (!pads_by_copy) {                                  // (host-packed: the offsets travel in the one copy of the whole block)
        // through pinned words, so that the copy runs on the SDMA engines like the sequence's: from the pageable vector the runtime copies with a one-workgroup
        // KERNEL, which in a batch stream waits for a CU while the previous batch's pre-filter holds them all (profiles/r06z_trace_e2e_gaps_sdma.log: 3.5 ms
        // per batch on the upload stream).  MS_MEASURE=1 MS_OFFSETS_PAGEABLE=1: the old way, for A/B runs.
        const void *src = s->offsets.data();
        if (!measure_env("MS_OFFSETS_PAGEABLE") && (s->h_off_pin = pinned_alloc(b_off, &s->h_off_pin_bytes))) {
            std::memcpy(s->h_off_pin, s->offsets.data(), b_off);
            src = s->h_off_pin;
        }
        MS_HIP(hipMemcpyAsync(s->d_offsets, src, b_off, hipMemcpyHostToDevice, s->up));
    }
    return MS_OK;
}

int ms_seqset_create(const char *bases, const int64_t *offsets, int64_t n_seqs, int keep_ascii, ms_seqset **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    std::unique_ptr<ms_seqset> s;
    int rc = seqset_common(offsets, n_seqs, s);
    if (rc) return rc;
    if (s->n_bases > 0 && !bases) { set_error("bases is NULL"); return MS_ERR_INVALID; }
    DeviceCtx *c;
    if ((rc = get_ctx(s->device, &c))) return rc;
    ms_seqset *raw = s.release();
    auto fail = [&](int code) { ms_seqset_free(raw); return code; };
    if ((rc = seqset_alloc_packed(raw))) return fail(rc);
    {
        void *blk = nullptr;
        if ((rc = pool_alloc(c, (size_t) raw->n_bases + 64, &blk, &raw->ascii_bytes))) return fail(rc);
        raw->d_ascii = static_cast<uint8_t *>(blk);
    }
    if (raw->n_bases > 0) {
        hipError_t e = hipMemcpyAsync(raw->d_ascii, bases, (size_t) raw->n_bases, hipMemcpyHostToDevice, raw->up);
        if (e != hipSuccess) { set_error("H2D copy failed: %s", hipGetErrorString(e)); return fail(MS_ERR_RUNTIME); }
    }
    if ((rc = launch_pack(raw->d_ascii, raw->n_bases, raw->d_codes, raw->d_nmask, raw->up))) return fail(rc);
    if ((rc = launch_blk2reg(raw->d_offsets, raw->R, raw->n_bases, raw->d_blk2reg, raw->d_blkinfo, raw->up))) return fail(rc);
    hipError_t e = hipStreamSynchronize(raw->up);
    if (e != hipSuccess) { set_error("upload / pack failed: %s", hipGetErrorString(e)); return fail(MS_ERR_RUNTIME); }
    if (!keep_ascii) { pool_free(c, raw->d_ascii, raw->ascii_bytes); raw->d_ascii = nullptr; raw->ascii_bytes = 0; }
    raw->built = true;
    *out = raw;
    return MS_OK;
}

// convert_seq on HOST threads (ms_hostpack.cpp): the packed codes, the mask and the region hints are made in pinned staging memory by
// n_threads threads and cross the link on the copy engines -- no kernel is launched, so nothing of the set's construction waits for CUs
// a running scan holds (the batch stream's upload stage, MS_STREAM_HOST_PACK).  The set is identical to ms_seqset_create's.
int ms_seqset_create_hostpacked(const char *bases, const int64_t *offsets, int64_t n_seqs, int n_threads, ms_seqset **out) {
    return seqset_create_hostpacked(bases, offsets, n_seqs, n_threads, nullptr, nullptr, out);
}

}  // extern "C"  (an internal C++ entry follows: the stream's uploader calls it with its own staging block)

// stage / stage_bytes: the caller's grow-only pinned staging block (a batch stream's uploader keeps ONE for its life: page-locking ~90 MB
// per batch costs tens of milliseconds); nullptr: a block of the call's own.
int ms::seqset_create_hostpacked(const char *bases, const int64_t *offsets, int64_t n_seqs, int n_threads, void **stage_io, size_t *stage_bytes_io, ms_seqset **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    std::unique_ptr<ms_seqset> s;
    int rc = seqset_common(offsets, n_seqs, s);
    if (rc) return rc;
    if (s->n_bases > 0 && !bases) { set_error("bases is NULL"); return MS_ERR_INVALID; }
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    DeviceCtx *c;
    if ((rc = get_ctx(s->device, &c))) return rc;
    ms_seqset *raw = s.release();
    auto fail = [&](int code) { ms_seqset_free(raw); return code; };
    if ((rc = seqset_alloc_packed(raw, true))) return fail(rc);
    const int64_t n_units = (raw->n_bases + 31) / 32, n_blocks = (raw->n_bases + 63) / 64 + 1;
    // The staging block MIRRORS the set's device block (seqset_alloc_packed: codes + pad | mask + pad | offsets | region of every 64th
    // position | block records, 256-byte aligned), so that ONE copy moves everything -- no kernel (a memset is one) and no second copy is
    // queued on the upload stream.
    auto up256 = [](size_t x) { return (x + 255) & ~(size_t) 255; };
    const size_t o_nmask = (size_t) (reinterpret_cast<char *>(raw->d_nmask) - static_cast<char *>(raw->block));
    const size_t o_off = (size_t) (reinterpret_cast<char *>(raw->d_offsets) - static_cast<char *>(raw->block));
    const size_t o_blk = (size_t) (reinterpret_cast<char *>(raw->d_blk2reg) - static_cast<char *>(raw->block));
    const size_t o_info = (size_t) (reinterpret_cast<char *>(raw->d_blkinfo) - static_cast<char *>(raw->block));
    const size_t need = up256(o_info + (size_t) n_blocks * 16) + 256;
    size_t got = 0;
    char *stage = nullptr;
    if (stage_io) {
        if (*stage_bytes_io < need) {
            if (*stage_io) (void) hipHostFree(*stage_io);
            *stage_io = nullptr; *stage_bytes_io = 0;
            void *p = nullptr;
            if (hipHostMalloc(&p, need + need / 8) != hipSuccess) { set_error("out of pinned host memory"); return fail(MS_ERR_NOMEM); }
            *stage_io = p; *stage_bytes_io = need + need / 8;
        }
        stage = static_cast<char *>(*stage_io);
    } else {
        stage = static_cast<char *>(pinned_alloc(need, &got));
        if (!stage) { set_error("out of pinned host memory"); return fail(MS_ERR_NOMEM); }
    }
    auto release = [&]() { if (!stage_io) pinned_free(stage, got); };
    uint32_t *h_codes = reinterpret_cast<uint32_t *>(stage), *h_nmask = reinterpret_cast<uint32_t *>(stage + o_nmask);
    int32_t *h_blk = reinterpret_cast<int32_t *>(stage + o_blk);
    int32_t *h_info = reinterpret_cast<int32_t *>(stage + o_info);
    {
        const bool all_far = measure_env("MS_BLKINFO_FAR") != nullptr;
        const int64_t *off = raw->offsets.data();
        const int64_t R = raw->R, nb = raw->n_bases;
        const int T = (int) std::min<int64_t>(n_threads, std::max<int64_t>(1, n_units / 4096));
        auto work = [&](int t) {
            host_pack_units(reinterpret_cast<const uint8_t *>(bases), nb, n_units * t / T, n_units * (t + 1) / T, h_codes, h_nmask);
            host_region_hints(off, R, n_blocks * t / T, n_blocks * (t + 1) / T, h_blk, h_info, all_far);
        };
        std::vector<std::thread> th;
        try { for (int t = 1; t < T; t++) th.emplace_back(work, t); }
        catch (const std::exception &) { for (auto &x : th) x.join(); release(); set_error("could not start packing threads"); return fail(MS_ERR_RUNTIME); }
        work(0);
        for (auto &x : th) x.join();
        std::memset(h_codes + 2 * n_units, 0, kPadWords * sizeof(uint32_t));
        std::memset(h_nmask + n_units, 0, kPadWords * sizeof(uint32_t));
        std::memcpy(stage + o_off, off, ((size_t) R + 1) * sizeof(int64_t));
    }
    hipError_t e = hipMemcpyAsync(raw->block, stage, o_info + (size_t) n_blocks * 16, hipMemcpyHostToDevice, raw->up);
    if (e == hipSuccess) e = hipStreamSynchronize(raw->up);
    release();
    if (e != hipSuccess) { set_error("upload of the packed set failed: %s", hipGetErrorString(e)); return fail(MS_ERR_RUNTIME); }
    raw->built = true;
    *out = raw;
    return MS_OK;
}

extern "C" {

// the host packer alone, for CPU tests (no device): codes [2 * ceil(n / 32)], nmask [ceil(n / 32)], blk2reg [(n + 63) / 64 + 1], blkinfo [4 x that]
int ms_debug_numa_probe(const char *root, const char *bdf, int32_t *node, int32_t *n_cpus, int32_t *n_nodes) {
    if (!root || !bdf) { set_error("NULL argument"); return MS_ERR_INVALID; }
    const int nd = numa_node_of_bdf(bdf, root);
    cpu_set_t set;
    if (node) *node = nd;
    if (n_cpus) *n_cpus = numa_cpus_of_node(nd, root, &set);
    if (n_nodes) *n_nodes = numa_node_count(root);
    return MS_OK;
}

int ms_debug_host_pack(const char *bases, const int64_t *offsets, int64_t n_seqs, uint32_t *codes, uint32_t *nmask, int32_t *blk2reg, int32_t *blkinfo) {
    if (!offsets || n_seqs < 0 || !codes || !nmask || !blk2reg || !blkinfo) { set_error("NULL argument"); return MS_ERR_INVALID; }
    const int64_t n = offsets[n_seqs];
    if (n > 0 && !bases) { set_error("bases is NULL"); return MS_ERR_INVALID; }
    host_pack_units(reinterpret_cast<const uint8_t *>(bases), n, 0, (n + 31) / 32, codes, nmask);
    host_region_hints(offsets, n_seqs, 0, (n + 63) / 64 + 1, blk2reg, blkinfo, false);
    return MS_OK;
}

}  // extern "C"

// The batch stream's upload stage (round 6): ONLY the copies -- ASCII and offsets cross the link on the DMA engines, nothing here needs a CU.
// convert_seq (pack_kernel) and the region hints are made by the SCAN stage, on the scan stream, in front of the batch's pre-filter
// (seqset_pack_pending, called by scan_locked).  Why: every kernel launched on the upload stream beside a scan waits for the pre-filter's
// persistent blocks to retire -- a full CU stalls the dispatcher's in-order hand-out (ms_handles.h, StreamSel) -- so the upload stage used
// to end when the PREVIOUS batch's pre-filter did (53 ms of "work" per configs[3] pass for 18 ms of copies and 0.6 ms of kernels,
// profiles/r05d_e2e_bounds.log) and the scan stage could not queue the next batch's scan behind the running one.  On the scan stream the
// two kernels cost what they cost in the resident step (0.3 ms per 500 Mbase) and wait for nothing.
int ms::seqset_create_upload_only(const char *bases, const int64_t *offsets, int64_t n_seqs, ms_seqset **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    std::unique_ptr<ms_seqset> s;
    int rc = seqset_common(offsets, n_seqs, s);
    if (rc) return rc;
    if (s->n_bases > 0 && !bases) { set_error("bases is NULL"); return MS_ERR_INVALID; }
    DeviceCtx *c;
    if ((rc = get_ctx(s->device, &c))) return rc;
    ms_seqset *raw = s.release();
    auto fail = [&](int code) { ms_seqset_free(raw); return code; };
    if ((rc = seqset_alloc_packed(raw, false, true))) return fail(rc);
    {
        void *blk = nullptr;
        if ((rc = pool_alloc(c, (size_t) raw->n_bases + 64, &blk, &raw->ascii_bytes))) return fail(rc);
        raw->d_ascii = static_cast<uint8_t *>(blk);
    }
    hipError_t e = hipSuccess;
    if (raw->n_bases > 0) e = hipMemcpyAsync(raw->d_ascii, bases, (size_t) raw->n_bases, hipMemcpyHostToDevice, raw->up);
    if (e == hipSuccess) e = hipStreamSynchronize(raw->up);
    if (e != hipSuccess) { set_error("H2D copy failed: %s", hipGetErrorString(e)); return fail(MS_ERR_RUNTIME); }
    raw->built = true;
    raw->pack_pending = true;
    *out = raw;
    return MS_OK;
}

// ... and the rest of the set's construction, queued on `st` (the scan stream; the caller holds the device's lock)
int ms::seqset_pack_pending(const ms_seqset *s_c, hipStream_t st) {
    ms_seqset *s = const_cast<ms_seqset *>(s_c);
    if (!s->pack_pending) return MS_OK;
    const size_t n_units = (size_t) ((s->n_bases + 31) / 32);
    MS_HIP(hipMemsetAsync(s->d_codes + 2 * n_units, 0, kPadWords * sizeof(uint32_t), st));
    MS_HIP(hipMemsetAsync(s->d_nmask + n_units, 0, kPadWords * sizeof(uint32_t), st));
    int rc;
    if ((rc = launch_pack(s->d_ascii, s->n_bases, s->d_codes, s->d_nmask, st))) return rc;
    if ((rc = launch_blk2reg(s->d_offsets, s->R, s->n_bases, s->d_blk2reg, s->d_blkinfo, st))) return rc;
    s->pack_pending = false;                       // (queued: everything that reads the set follows on the same stream, or waits for it)
    return MS_OK;
}

extern "C" {

int ms_seqset_from_device(const void *d_bases, const int64_t *offsets, int64_t n_seqs, ms_seqset **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    std::unique_ptr<ms_seqset> s;
    int rc = seqset_common(offsets, n_seqs, s);
    if (rc) return rc;
    if (s->n_bases > 0 && !d_bases) { set_error("d_bases is NULL"); return MS_ERR_INVALID; }
    DeviceCtx *c;
    if ((rc = get_ctx(s->device, &c))) return rc;
    ms_seqset *raw = s.release();
    auto fail = [&](int code) { ms_seqset_free(raw); return code; };
    if ((rc = seqset_alloc_packed(raw))) return fail(rc);
    if ((rc = launch_pack(static_cast<const uint8_t *>(d_bases), raw->n_bases, raw->d_codes, raw->d_nmask, raw->up)))
        return fail(rc);
    if ((rc = launch_blk2reg(raw->d_offsets, raw->R, raw->n_bases, raw->d_blk2reg, raw->d_blkinfo, raw->up))) return fail(rc);
    hipError_t e = hipStreamSynchronize(raw->up);
    if (e != hipSuccess) { set_error("pack kernel failed: %s", hipGetErrorString(e)); return fail(MS_ERR_RUNTIME); }
    raw->built = true;
    *out = raw;
    return MS_OK;
}

int ms_seqset_repack(ms_seqset *s) {
    if (!s) { set_error("NULL argument"); return MS_ERR_INVALID; }
    if (!s->d_ascii) { set_error("sequence set was created without keep_ascii"); return MS_ERR_INVALID; }
    DeviceCtx *c;
    int rc = get_ctx(s->device, &c);
    if (rc) return rc;
    const hipStream_t up = c->stream_up;
    if ((rc = launch_pack(s->d_ascii, s->n_bases, s->d_codes, s->d_nmask, up))) return rc;
    if ((rc = launch_blk2reg(s->d_offsets, s->R, s->n_bases, s->d_blk2reg, s->d_blkinfo, up))) return rc;
    MS_HIP(hipStreamSynchronize(up));
    return MS_OK;
}

int ms_seqset_size(const ms_seqset *s, int64_t *n_seqs, int64_t *n_bases) {
    if (!s) { set_error("NULL argument"); return MS_ERR_INVALID; }
    if (n_seqs) *n_seqs = s->R;
    if (n_bases) *n_bases = s->n_bases;
    return MS_OK;
}

void ms_seqset_free(ms_seqset *s) {
    if (!s) return;
    (void) hipSetDevice(s->device);
    // a set freed on an ERROR path of its construction may still have its memsets / upload / pack kernel queued on the upload
    // stream: the blocks must not go back to the pool (another thread can take them at once) before that work is done.
    // A set that was built (built: its stream was synchronised then) needs no wait -- the stream is shared, waiting would stall
    // behind the NEXT batch's upload.
    if (!s->built && s->up) (void) hipStreamSynchronize(s->up);
    DeviceCtx *c = nullptr;
    const bool have = get_ctx(s->device, &c) == MS_OK;
    if (s->d_ascii) { if (have) pool_free(c, s->d_ascii, s->ascii_bytes); else (void) hipFree(s->d_ascii); }
    if (s->block) { if (have) pool_free(c, s->block, s->block_bytes); else (void) hipFree(s->block); }
    if (s->h_off_pin) pinned_free(s->h_off_pin, s->h_off_pin_bytes);
    delete s;
}

// ------------------------------------------------------------------- resident genome --

// A genome is a sequence set whose "sequences" are the chromosomes, kept packed in HBM
// (3 Gbp = 1.1 GB); regions are then cut out on the device with no ASCII traffic at all.
int ms_genome_create(const char *bases, const int64_t *chrom_offsets, int32_t n_chroms, ms_genome **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    ms_seqset *s = nullptr;
    int rc = ms_seqset_create(bases, chrom_offsets, n_chroms, 0, &s);
    if (rc) return rc;
    *out = reinterpret_cast<ms_genome *>(s);
    return MS_OK;
}

// convert_seq (cscore.c:81-114) into the packed layout on HOST threads, no device: what the genome-file builder runs (ms_hostpack.cpp)
int ms_pack_bases_host(const char *bases, int64_t n_bases, int n_threads, uint32_t *codes, uint32_t *nmask) {
    if (n_bases < 0 || (n_bases > 0 && (!bases || !codes || !nmask))) { set_error("NULL argument"); return MS_ERR_INVALID; }
    const int64_t n_units = (n_bases + 31) / 32;
    const int T = (int) std::max<int64_t>(1, std::min<int64_t>(std::min(n_threads < 1 ? 1 : n_threads, 64), n_units / 4096));
    auto work = [&](int t) { host_pack_units(reinterpret_cast<const uint8_t *>(bases), n_bases, n_units * t / T, n_units * (t + 1) / T, codes, nmask); };
    std::vector<std::thread> th;
    try { for (int t = 1; t < T; t++) th.emplace_back(work, t); }
    catch (const std::exception &) { for (auto &x : th) x.join(); set_error("could not start packing threads"); return MS_ERR_RUNTIME; }
    work(0);
    for (auto &x : th) x.join();
    return MS_OK;
}

// A genome from its PACKED form (a genome file made once: motifscan_amd/genome.py): the two planes cross the link as they are -- 0.375 B per
// base, no ASCII, no pack kernel; only the region hints are made on the device.  The planes are validated first (a file is outside input):
// a non-ACGT base must hold code 0 and nothing may be set past the last base -- the invariants pack_kernel guarantees and the scan kernels rely on.
int ms_genome_create_packed(const uint32_t *codes, const uint32_t *nmask, const int64_t *chrom_offsets, int32_t n_chroms, ms_genome **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    std::unique_ptr<ms_seqset> s;
    int rc = seqset_common(chrom_offsets, n_chroms, s);
    if (rc) return rc;
    const int64_t n_bases = s->n_bases, n_units = (n_bases + 31) / 32;
    if (n_bases > 0 && (!codes || !nmask)) { set_error("codes / nmask is NULL"); return MS_ERR_INVALID; }
    {
        int64_t cbad[kMaxChunks];
        for (int t = 0; t < kMaxChunks; t++) cbad[t] = -1;
        parallel_chunks(n_units, [&](int t, int64_t b, int64_t e) {
            for (int64_t u = b; u < e; u++) {
                const uint64_t cw = (uint64_t) codes[2 * u] | ((uint64_t) codes[2 * u + 1] << 32);
                uint64_t nw = nmask[u], spread = 0;                      // bit i of the mask -> bits 2i, 2i + 1
                for (int i = 0; nw; i++, nw >>= 1) if (nw & 1u) spread |= 3ULL << (2 * i);
                uint64_t tail_c = 0;
                uint32_t tail_n = 0;
                if (32 * (u + 1) > n_bases) {
                    const int valid = (int) (n_bases - 32 * u);
                    tail_c = valid >= 32 ? 0 : ~0ULL << (2 * valid);
                    tail_n = valid >= 32 ? 0 : ~0u << valid;
                }
                if ((cw & spread) || (cw & tail_c) || (nmask[u] & tail_n)) { cbad[t] = u; return; }
            }
        });
        for (int t = 0; t < kMaxChunks; t++)
            if (cbad[t] >= 0) { set_error("packed genome is corrupt at bases %lld..%lld: a non-ACGT base with a non-zero code, or bits past the end", (long long) (32 * cbad[t]), (long long) (32 * cbad[t] + 31)); return MS_ERR_INVALID; }
    }
    DeviceCtx *c;
    if ((rc = get_ctx(s->device, &c))) return rc;
    ms_seqset *raw = s.release();
    auto fail = [&](int code) { ms_seqset_free(raw); return code; };
    if ((rc = seqset_alloc_packed(raw))) return fail(rc);
    if (n_units > 0) {
        hipError_t e = hipMemcpyAsync(raw->d_codes, codes, (size_t) n_units * 8, hipMemcpyHostToDevice, raw->up);
        if (e == hipSuccess) e = hipMemcpyAsync(raw->d_nmask, nmask, (size_t) n_units * 4, hipMemcpyHostToDevice, raw->up);
        if (e != hipSuccess) { set_error("H2D copy of the packed genome failed: %s", hipGetErrorString(e)); return fail(MS_ERR_RUNTIME); }
    }
    if ((rc = launch_blk2reg(raw->d_offsets, raw->R, raw->n_bases, raw->d_blk2reg, raw->d_blkinfo, raw->up))) return fail(rc);
    hipError_t e = hipStreamSynchronize(raw->up);
    if (e != hipSuccess) { set_error("upload of the packed genome failed: %s", hipGetErrorString(e)); return fail(MS_ERR_RUNTIME); }
    raw->built = true;
    *out = reinterpret_cast<ms_genome *>(raw);
    return MS_OK;
}

// ... and back: the two planes of a resident genome on the host (to write the genome file after a genome was packed on the device)
int ms_genome_packed_host(const ms_genome *g, uint32_t *codes, uint32_t *nmask) {
    if (!g || !codes || !nmask) { set_error("NULL argument"); return MS_ERR_INVALID; }
    const ms_seqset *s = reinterpret_cast<const ms_seqset *>(g);
    const size_t n_units = (size_t) ((s->n_bases + 31) / 32);
    if (n_units == 0) return MS_OK;
    MS_HIP(hipSetDevice(s->device));
    MS_HIP(hipMemcpy(codes, s->d_codes, n_units * 8, hipMemcpyDeviceToHost));
    MS_HIP(hipMemcpy(nmask, s->d_nmask, n_units * 4, hipMemcpyDeviceToHost));
    return MS_OK;
}

int ms_genome_size(const ms_genome *g, int32_t *n_chroms, int64_t *n_bases) {
    if (!g) { set_error("NULL argument"); return MS_ERR_INVALID; }
    const ms_seqset *s = reinterpret_cast<const ms_seqset *>(g);
    if (n_chroms) *n_chroms = (int32_t) s->R;
    if (n_bases) *n_bases = s->n_bases;
    return MS_OK;
}

void ms_genome_free(ms_genome *g) { ms_seqset_free(reinterpret_cast<ms_seqset *>(g)); }

int ms_seqset_from_genome(const ms_genome *g, const int32_t *chrom, const int64_t *start, const int64_t *end,
                          int64_t n_regions, ms_seqset **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    if (!g) { set_error("NULL genome"); return MS_ERR_INVALID; }
    if (n_regions < 0 || (n_regions > 0 && (!chrom || !start || !end))) { set_error("bad region arrays"); return MS_ERR_INVALID; }
    const ms_seqset *G = reinterpret_cast<const ms_seqset *>(g);
    std::vector<int64_t> dst, src;
    try { dst.resize((size_t) n_regions + 1); src.resize((size_t) n_regions + 1); }
    catch (const std::bad_alloc &) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    // pass 1 (threads): validate, source position and length of every region, per-chunk totals
    int64_t csum[kMaxChunks], cbad[kMaxChunks];
    for (int t = 0; t < kMaxChunks; t++) { csum[t] = 0; cbad[t] = -1; }
    const int64_t n_chroms = G->R;
    const int64_t *goff = G->offsets.data();
    parallel_chunks(n_regions, [&](int t, int64_t b, int64_t e) {
        int64_t sum = 0, bad = -1;
        for (int64_t r = b; r < e; r++) {
            const int64_t ch = chrom[r];
            bool ok = ch >= 0 && ch < n_chroms;
            if (ok) {
                const int64_t clen = goff[ch + 1] - goff[ch];
                ok = start[r] >= 0 && end[r] >= start[r] && end[r] <= clen;
            }
            if (!ok) { if (bad < 0) bad = r; continue; }
            src[(size_t) r] = goff[ch] + start[r];
            dst[(size_t) r + 1] = end[r] - start[r];          // length for now
            sum += end[r] - start[r];
        }
        csum[t] = sum; cbad[t] = bad;
    });
    for (int t = 0; t < kMaxChunks; t++)
        if (cbad[t] >= 0) {
            const int64_t r = cbad[t];
            if (chrom[r] < 0 || chrom[r] >= n_chroms) set_error("region %lld: chromosome index %d out of range", (long long) r, chrom[r]);
            else set_error("region %lld: [%lld, %lld) is outside chromosome %d of length %lld", (long long) r, (long long) start[r],
                           (long long) end[r], chrom[r], (long long) (goff[chrom[r] + 1] - goff[chrom[r]]));
            return MS_ERR_INVALID;
        }
    // pass 2 (threads): lengths -> offsets, every chunk starting at the sum of the chunks before it
    int64_t cbase[kMaxChunks + 1];
    cbase[0] = 0;
    for (int t = 0; t < kMaxChunks; t++) cbase[t + 1] = cbase[t] + csum[t];
    dst[0] = 0;
    parallel_chunks(n_regions, [&](int t, int64_t b, int64_t e) {
        int64_t run = cbase[t];
        for (int64_t r = b; r < e; r++) { run += dst[(size_t) r + 1]; dst[(size_t) r + 1] = run; }
    });
    src[(size_t) n_regions] = 0;
    const int prev_dev = g_device;
    g_device = G->device;                              // the new set lives next to its genome
    std::unique_ptr<ms_seqset> s;
    int rc = seqset_common(dst.data(), n_regions, s, &dst);
    g_device = prev_dev;
    if (rc) return rc;
    DeviceCtx *c;
    if ((rc = get_ctx(G->device, &c))) return rc;
    ms_seqset *raw = s.release();
    auto fail = [&](int code) { ms_seqset_free(raw); return code; };
    if ((rc = seqset_alloc_packed(raw))) return fail(rc);
    int64_t *d_src = nullptr;
    if ((rc = dev_alloc(&d_src, (size_t) n_regions + 1))) return fail(rc);
    hipError_t he = hipMemcpyAsync(d_src, src.data(), ((size_t) n_regions + 1) * sizeof(int64_t), hipMemcpyHostToDevice, raw->up);
    if (he == hipSuccess) {
        rc = launch_extract(G->d_codes, G->d_nmask, d_src, raw->d_offsets, raw->R, raw->n_bases, raw->d_codes, raw->d_nmask, raw->up);
        if (!rc) rc = launch_blk2reg(raw->d_offsets, raw->R, raw->n_bases, raw->d_blk2reg, raw->d_blkinfo, raw->up);
        if (!rc) he = hipStreamSynchronize(raw->up);
    }
    dev_free(d_src);
    if (rc) return fail(rc);
    if (he != hipSuccess) { set_error("extraction failed: %s", hipGetErrorString(he)); return fail(MS_ERR_RUNTIME); }
    raw->built = true;
    *out = raw;
    return MS_OK;
}

// ---------------------------------------------------------------------------- scan --

static int scratch_reserve(Scratch &sc, size_t cand_cap, size_t hit_cap) {
    int rc;
    if (cand_cap > sc.cand_cap) {
        dev_free(sc.cand);
        sc.cand_cap = 0;
        if ((rc = dev_alloc(&sc.cand, cand_cap))) return rc;
        sc.cand_cap = cand_cap;
    }
    if (hit_cap > sc.hit_cap) {
        dev_free(sc.keys); dev_free(sc.vals); dev_free(sc.keys_sorted);
        sc.hit_cap = 0;
        if ((rc = dev_alloc(&sc.keys, hit_cap))) return rc;
        if ((rc = dev_alloc(&sc.vals, hit_cap))) return rc;
        if ((rc = dev_alloc(&sc.keys_sorted, hit_cap))) return rc;
        sc.hit_cap = hit_cap;
    }
    return MS_OK;
}

}  // extern "C"

namespace ms {

// The scan pipeline: pre-filter -> fp64 re-score of the candidates (+ motifs the filter cannot take) -> order -> coordinates.
// The caller holds c->mu (one scan at a time per device: shared scratch) and pwms->mu (lazily cached device copies / plan).
int pending_scan_init(PendingScan *p) {
    for (auto &e : p->ev) MS_HIP(hipEventCreate(&e));
    MS_HIP(hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
    MS_HIP(hipHostMalloc(&p->h_counters, 8 * sizeof(unsigned long long)));
    return MS_OK;
}

PendingScan *pending_scan_acquire(DeviceCtx *c) {
    {
        std::lock_guard<std::mutex> lk(c->pend_mu);
        if (!c->pend_cache.empty()) { PendingScan *p = c->pend_cache.back(); c->pend_cache.pop_back(); return p; }
    }
    PendingScan *p = new (std::nothrow) PendingScan();
    if (!p) { set_error("out of host memory"); return nullptr; }
    if (pending_scan_init(p) != MS_OK) { pending_scan_destroy(p); delete p; return nullptr; }
    return p;
}

void pending_scan_release(DeviceCtx *c, PendingScan *p) {
    if (!p) return;
    if (p->raw) { ms_result_free(p->raw); p->raw = nullptr; }
    p->active = false;
    std::lock_guard<std::mutex> lk(c->pend_mu);
    if (c->pend_cache.size() < 8) { c->pend_cache.push_back(p); return; }
    pending_scan_destroy(p);
    delete p;
}

void pending_scan_destroy(PendingScan *p) {
    for (auto &e : p->ev) if (e) (void) hipEventDestroy(e);
    if (p->done) (void) hipEventDestroy(p->done);
    if (p->h_counters) (void) hipHostFree(p->h_counters);
    if (p->h_offsets) (void) hipHostFree(p->h_offsets);
    *p = PendingScan();
}

// stage times, counts and the next scan's prediction, once a scan's kernels are known to be done
static void finish_scan(ms_result *raw, hipEvent_t *ev, ms_pwmset *pwms, int64_t n_bases, int64_t R, int strand_mask, bool exact_only,
                        unsigned long long n_cand, unsigned long long n_hits, bool with_back) {
    ms_scan_stats &stt = raw->stats;
    stt.n_candidates = (int64_t) n_cand;
    stt.n_hits = (int64_t) n_hits;
    raw->n_hits = (int64_t) n_hits;
    int64_t pwm_bytes = 0;
    for (int32_t p = 0; p < pwms->P; p++) pwm_bytes += 32LL * pwms->widths[p];
    // compulsory HBM bytes of one call
    stt.hbm_bytes_algorithmic = (n_bases + 3) / 4 + (n_bases + 7) / 8 + 8 * (R + 1) + pwm_bytes + 16 * (int64_t) n_hits + 8LL * pwms->P;
    float ms01 = 0, ms12 = 0, ms34 = 0, ms45 = 0, ms05 = 0;
    (void) hipEventElapsedTime(&ms01, ev[0], ev[1]);
    (void) hipEventElapsedTime(&ms12, ev[1], ev[2]);
    stt.ms_prefilter = ms01; stt.ms_exact = ms12; stt.ms_total = ms01 + ms12;
    if (with_back) {
        (void) hipEventElapsedTime(&ms34, ev[3], ev[4]);
        (void) hipEventElapsedTime(&ms45, ev[4], ev[5]);
        (void) hipEventElapsedTime(&ms05, ev[0], ev[5]);
        stt.ms_sort = ms34; stt.ms_finalize = ms45; stt.ms_total = ms05;
    }
    if (measure_env("MS_TRACE_BLOCKS"))                  // measurement: which device blocks a scan worked on, beside its stage times (tools/e2e_block_probe.py)
        fprintf(stderr, "MSBLK done res=%p bases=%lld pf=%.3f fp64=%.3f sort=%.3f fin=%.3f\n", raw->block, (long long) n_bases, ms01, ms12, ms34, ms45);
    if (stt.n_windows > 0 && !raw->invalid) {            // what the next scan of this set of PWMs may expect (scan_locked)
        pwms->pred_density = (double) n_hits / (double) stt.n_windows;
        pwms->pred_strand = strand_mask;
        pwms->pred_cutoff_version = pwms->cutoff_version;
        pwms->pred_exact_only = exact_only;
    }
}

int scan_complete(DeviceCtx *c, ms_pwmset *pwms, PendingScan *p, ms_result **out) {
    *out = nullptr;
    if (!p || !p->active) { set_error("no pending scan"); return MS_ERR_INVALID; }
    p->active = false;
    ms_result *raw = p->raw;
    p->raw = nullptr;
    hipError_t he = hipEventSynchronize(p->done);
    if (he != hipSuccess) { set_error("scan kernels failed: %s", hipGetErrorString(he)); ms_result_free(raw); return MS_ERR_RUNTIME; }
    const unsigned long long n_cand = p->cand_static + p->h_counters[0], n_hits = p->h_counters[1];
    if (n_cand <= p->cand_cap && n_hits <= p->hit_cap && n_hits <= p->n_pred) {
        if (p->offsets_queued) {                       // (copied in stream order, in front of `done`: scan_locked)
            const size_t n_off = raw->motif_offsets.size();
            std::memcpy(raw->motif_offsets.data(), p->h_offsets, n_off * sizeof(int64_t));
            try { raw->h_region_counts.assign(p->h_offsets + n_off, p->h_offsets + n_off + raw->P); } catch (const std::bad_alloc &) { raw->h_region_counts.clear(); }
        } else {
            he = hipMemcpy(raw->motif_offsets.data(), raw->d_motif_first, raw->motif_offsets.size() * sizeof(int64_t), hipMemcpyDeviceToHost);
            if (he != hipSuccess) { set_error("copy failed: %s", hipGetErrorString(he)); ms_result_free(raw); return MS_ERR_RUNTIME; }
        }
        pwms->pred_margin = std::max(0.04, pwms->pred_margin * 0.9);
        finish_scan(raw, p->ev, pwms, p->n_bases, p->R, p->strand_mask, p->exact_only, n_cand, n_hits, true);
        *out = raw;
        return MS_OK;
    }
    pwms->pred_margin = std::min(1.0, pwms->pred_margin * 2.0);
    pwms->pred_density = -1.0;                           // learn again through the exactly-sized form
    ms_result_free(raw);
    (void) c;
    return MS_SCAN_RETRY;
}

int scan_locked(DeviceCtx *c, ms_pwmset *pwms, const ms_seqset *seqs, int strand_mask, uint32_t flags, ms_result **out, PendingScan *pend) {
    int rc;
    const bool exact_only = (flags & MS_SCAN_EXACT_ONLY) != 0;
    // the B-operand table, the waves' sequence staging, their PfEmit and (at least) kRareCapMin parking entries follow the tables -- and the
    // waves' one-hot arrays (40 KB) in the double-pass kernels ONLY: a plan with a wide tile (a motif of 32 ... 63 columns on the pre-filter)
    // runs the single-pass kernels, which never touch that array, and keeps the room for its tables
    if ((rc = seqset_pack_pending(seqs, c->stream))) return rc;       // a batch stream's set: its pack / hint kernels run here, in front of its pre-filter
    const size_t lds_fixed_wide = kF6LutBytes + kPfStageBytes + kPfEmitBytes + kPfRareBytesMin;
    const size_t lds_fixed_narrow = lds_fixed_wide + kPfOnehotBytes;
    if ((rc = pwmset_upload(pwms, c->device, c->stream))) return rc;
    auto plan_is_wide = [&]() { for (const TileDesc &t : pwms->plan.tiles) if (t.max_nk > 2) return true; return false; };
    bool wide_layout = false;
    {
        auto budget_of = [&](size_t fixed) {
            size_t b = c->lds_max / (size_t) kPfBlocksPerCu - fixed;
            if (const char *e = measure_env("MS_PF_LDS_BUDGET")) b = std::min(b, (size_t) std::max(1, atoi(e)));   // test aid: several LDS tiles, as a very large motif set would have
            return b;
        };
        // only a motif of 32 ... kMaxFastWidth columns can make a wide tile; whether one does (it may fall to the all-fp64 path) the plan says.
        // The outcome is remembered per (strand, cutoffs, exact-only): a set whose wide motifs all left the pre-filter plans once, not twice per scan
        bool may_be_wide = false;
        for (int32_t w : pwms->widths) may_be_wide = may_be_wide || (w >= 2 * kF6Cols && w <= kMaxFastWidth);
        const bool known_narrow = pwms->narrow_strand == strand_mask && pwms->narrow_cutoff_version == pwms->cutoff_version && pwms->narrow_exact_only == exact_only;
        if (may_be_wide && !exact_only && !known_narrow) {
            if ((rc = pwmset_plan(pwms, strand_mask, budget_of(lds_fixed_wide), exact_only, false, c->device))) return rc;
            wide_layout = plan_is_wide();
            if (!wide_layout) { pwms->narrow_strand = strand_mask; pwms->narrow_cutoff_version = pwms->cutoff_version; pwms->narrow_exact_only = exact_only; }
        }
        if ((rc = pwmset_plan(pwms, strand_mask, budget_of(wide_layout ? lds_fixed_wide : lds_fixed_narrow), exact_only, true, c->device))) return rc;
        if (wide_layout != plan_is_wide()) { set_error("internal: the pre-filter plan changed its kernel family between two builds"); return MS_ERR_RUNTIME; }
    }
    const size_t lds_fixed = wide_layout ? lds_fixed_wide : lds_fixed_narrow;
    const PrefilterPlan &plan = pwms->plan;

    std::unique_ptr<ms_result> res(new (std::nothrow) ms_result());
    if (!res) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    res->device = c->device;
    res->P = pwms->P;
    res->R = seqs->R;
    res->motif_offsets.assign((size_t) pwms->P + 1, 0);
    ms_scan_stats &stt = res->stats;
    std::memset(&stt, 0, sizeof(stt));
    stt.n_bases = seqs->n_bases;
    stt.n_pwms = pwms->P;
    stt.n_pwms_exact = (int32_t) plan.exact_motifs.size();
    stt.n_tiles = (int32_t) plan.tiles.size();
    int64_t fast_windows = 0;
    for (int32_t p = 0; p < pwms->P; p++) stt.n_windows += windows_for_width(seqs, pwms->widths[p]);
    for (int32_t p : plan.fast_motifs) fast_windows += windows_for_width(seqs, pwms->widths[p]);
    const int64_t padded = ((seqs->n_bases + 63) / 64) * 64;
    {
        bool wide_tiles = false;
        for (const TileDesc &t : plan.tiles) wide_tiles = wide_tiles || t.max_nk > 2;
        stt.lds_bytes_read = plan.lds_bytes_per_position * padded / (wide_tiles ? 1 : 2);      // (the double pass reads a row tile's operand once for 128 window starts)
    }
    stt.pf_engine = 3;
    {
        int64_t cells = 0;                                                          // (window, column) pairs of one strand
        for (int32_t p : plan.fast_motifs) cells += windows_for_width(seqs, pwms->widths[p]) * pwms->widths[p];
        stt.mfma_ops_algorithmic = 2 * (strand_mask == 3 ? 2 : 1) * cells;           // one multiply-add per cell and strand
        stt.mfma_ops = padded / 32 * plan.kb_total * (2LL * 32 * 32 * 64);           // one 32x32x64 instruction per (32 windows, row tile, k-block)
    }

    ms_result *raw = res.release();
    auto fail = [&](int code) { ms_result_free(raw); return code; };
    hipError_t he = hipSuccess;

    int gbits = 1;
    while ((1LL << gbits) <= seqs->n_bases) gbits++;
    // hit coordinate = (region, position in the region) when that costs at most 2 more key bits than the global base
    // position: finalize_rp_kernel then needs no position -> region look-ups.  MS_HIT_COORD=global forces the other form.
    int rbits = 1, pbits = 0;
    {
        const int64_t max_len = seqs->len_sorted.empty() ? 0 : seqs->len_sorted.back();
        int pb = 1;
        while ((1LL << pb) < std::max<int64_t>(max_len, 1)) pb++;
        while ((1LL << rbits) < std::max<int64_t>(seqs->R, 1)) rbits++;
        const char *e = measure_env("MS_HIT_COORD");
        if (rbits + pb <= gbits + 2 && !(e && e[0] == 'g')) { pbits = pb; gbits = rbits + pb; }
    }
    int mbits = 1;
    while ((1 << mbits) < std::max(pwms->P, 1)) mbits++;

    if (pwms->P == 0 || seqs->n_bases == 0) {                 // nothing to scan: [] / [[]...]  (cscore.c:443-445)
        void *blk = nullptr;
        size_t got = 0;
        if ((rc = pool_alloc(c, result_block_bytes(pwms->P, 0), &blk, &got))) return fail(rc);
        raw->block = blk;
        raw->block_bytes = got;
        result_carve(raw, blk, 0);
        he = hipMemsetAsync(blk, 0, 16 * ((size_t) pwms->P + 1), c->stream);       // counts and offsets: all zero
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { set_error("memset failed: %s", hipGetErrorString(he)); return fail(MS_ERR_RUNTIME); }
        *out = raw;
        return MS_OK;
    }

    Scratch &sc = c->sc;
    // expected density at the CLI default p = 1e-4 is ~1.5e-4 candidates per window and strand; 4x head room
    size_t want_cand = (size_t) std::min<double>(std::max<double>(1 << 20, 6e-4 * (double) fast_windows), 3.0e9);
    size_t want_hits = want_cand;
    if (!plan.exact_motifs.empty()) want_hits = std::max<size_t>(want_hits, 1 << 22);
    // a wave reserves candidate slots in blocks (ms_kernels.hip, "candidate hand-off"): about a quarter of what it is expected to
    // need, 64 ... 2048 (its first block is its own, without an atomic); the slots a wave leaves unused in its last block are head
    // room on top
    const int64_t pf_waves_max = (int64_t) c->n_cu * kPfBlocksPerCu * (kPfThreads / 64);
    // what the previous scan of this set at these cutoffs and strands found, per (motif, window): sizes the blocks, and picks the kernel form
    const bool density_known = pwms->pred_density >= 0 && pwms->pred_strand == strand_mask && pwms->pred_cutoff_version == pwms->cutoff_version &&
                               pwms->pred_exact_only == exact_only;
    const double cand_density = density_known ? std::max(1.5e-4, 1.3 * pwms->pred_density) : 1.5e-4;
    uint32_t cand_block = 64;
    while (cand_block < 2048 && (double) cand_block * 4.0 * (double) pf_waves_max < cand_density * (double) fast_windows) cand_block *= 2;
    // the dense-candidate form of the pre-filter (ms_kernels.hip): expected hits per row tile and 64 window starts above kDenseHitsPerHalfTile
    bool pf_dense = false;
    {
        int64_t n_row_tiles = 0;
        bool wide_plan = false;
        for (const TileDesc &t : plan.tiles) {
            wide_plan = wide_plan || t.max_nk > 2;
            for (int i = 0; i < t.n_classes; i++) n_row_tiles += t.cls[i].n_row_tiles;
        }
        if (density_known && !wide_plan && n_row_tiles > 0)
            pf_dense = pwms->pred_density * 64.0 * (double) plan.fast_motifs.size() / (double) n_row_tiles > kDenseHitsPerHalfTile;
        if (const char *e = measure_env("MS_PF_DENSE")) pf_dense = !wide_plan && atoi(e) != 0;       // test aid / A-B: either form at any density
    }
    want_cand += (size_t) 2 * pf_waves_max * cand_block;          // the waves' own first blocks + the unused rest of their last ones
    want_cand = std::max(want_cand, sc.cand_cap);
    want_hits = std::max(want_hits, sc.hit_cap);

    const DevSeq S = dev_seq(seqs);
    const DevPwm Pw = dev_pwm(pwms);
    size_t lds_bytes = 0;
    for (const TileDesc &t : plan.tiles) lds_bytes = std::max(lds_bytes, (size_t) t.table_len16 * 16);
    const uint32_t lut_off16 = (uint32_t) (lds_bytes / 16);
    lds_bytes += lds_fixed;
    // the waves' candidate parking space takes what the tables leave of the block's LDS: kRareCapMin ... kRareCapMax entries per wave
    uint32_t rare_cap = (uint32_t) kRareCapMin;
    {
        const size_t per_entry = (size_t) (kPfThreads / 64) * kRareEntryWords * sizeof(uint32_t);
        const size_t avail = c->lds_max / (size_t) kPfBlocksPerCu;
        if (avail > lds_bytes) rare_cap = (uint32_t) std::min<size_t>((size_t) kRareCapMax, (size_t) kRareCapMin + (avail - lds_bytes) / per_entry);
        if (const char *e = measure_env("MS_PF_RARE_CAP")) rare_cap = (uint32_t) std::max(kRareCapMin, std::min((int) rare_cap, atoi(e)));    // test aid
        lds_bytes += (size_t) (rare_cap - (uint32_t) kRareCapMin) * per_entry;
    }
    int pf_no_emit = 0;
    if (const char *e = measure_env("MS_PF_NOEMIT")) pf_no_emit = atoi(e);
    const bool pf_clock = measure_env("MS_PF_CLOCK") && atoi(measure_env("MS_PF_CLOCK")) != 0;
    unsigned long long *d_clk = nullptr;
    int clk_blocks = 0;
    const bool pf_meas = pf_no_emit != 0 || pf_clock;              // the measurement instantiation of the kernel
    int pf_floor = 0;                                              // MS_MEASURE=1 MS_PF_FLOOR=1..4: a compile-time cut of the product kernel (ms_kernels.hip, FLOOR)
    if (const char *e = measure_env("MS_PF_FLOOR")) pf_floor = std::max(0, std::min(4, atoi(e)));
    if (pf_meas) pf_floor = 0;
    raw->invalid = pf_no_emit != 0 || pf_floor != 0;               // stage times only: the hit accessors refuse such a result

    // ---- the pre-filter's launch geometry
    const int n_tiles = (int) plan.tiles.size();
    int bpt = 1;
    int64_t pf_wave_passes = 1;
    bool counter_used = true;
    if (n_tiles > 0) {
        // While a batch stream is live and the device is partitioned (StreamSel): the scan owns n_cu - n_cu_copy CUs (the
        // units are handed out dynamically: fewer blocks just take more each)
        const int reserve = c->n_streams.load() > 0 ? c->n_cu_copy : 0;
        const int64_t pf_chunks = (S.n_bases + kPfThreads - 1) / kPfThreads;
        bpt = (int) std::max<int64_t>(1, std::min<int64_t>(pf_chunks, (c->n_cu - reserve) * kPfBlocksPerCu / n_tiles));
        if (const char *e = measure_env("MS_PF_MAX_BLOCKS")) bpt = std::max(1, std::min(bpt, atoi(e)));    // test aid: few blocks per tile, as a very large motif set would have
        // unit of the per-wave hand-out: a pass (64 window starts against a tile's k-blocks; 128 in a double pass) takes ~0.25 us per k-block and 64 windows with 16
        // waves per CU, and the launch's waves should not exceed ~47 atomics per microsecond on a tile's counter word
        const int64_t kb_tile = std::max<int64_t>(1, plan.kb_total / n_tiles);
        const double waves = (double) bpt * (kPfThreads / 64);                          // per tile
        const double waves_word = waves / std::min(kPfCounters, bpt);                   // ... and per counter word
        bool wide_plan = false;
        for (const TileDesc &t : plan.tiles) wide_plan = wide_plan || t.max_nk > 2;
        const int64_t pass_windows = wide_plan ? 64 : 128;                              // the kernels without wide classes scan double passes (ms_kernels.hip)
        const int64_t need = (int64_t) std::ceil(waves_word / (47.0 * 0.25 * (double) (pass_windows / 64) * (double) kb_tile));
        const int64_t passes_total = (S.n_bases + pass_windows - 1) / pass_windows, n_waves = (int64_t) waves;
        int64_t wp = 2;                                               // a power of two: units start on 128-position boundaries at least
        while (wp < 256 && (double) wp < 0.9 * (double) need) wp *= 2;            // the words' rate limit
        while (wp < 8 && 128 * wp <= passes_total / n_waves) wp *= 2;             // a long launch: the tail (one unit) stays below 1 % anyway, fewer atomics
        if (passes_total <= 8 * std::max<int64_t>(wp, 8) * n_waves) {
            // fewer than 8 units (of 8 passes at least) per wave: one even unit each and no atomics (a second round of a few
            // units would leave most waves idle); the kernel then never touches the counter words
            wp = std::max<int64_t>(1, (passes_total + n_waves - 1) / n_waves);
            counter_used = false;
        }
        pf_wave_passes = wp;
        if (counter_used && pf_wave_passes < 2) {                     // the kernel's hand-written atomic needs a second pass before its value is read (ms_kernels.hip)
            set_error("internal: a dynamic hand-out of single-pass units (wave_passes %lld)", (long long) pf_wave_passes);
            return MS_ERR_RUNTIME;
        }
    }

    // counters: [0] candidate record slots, [1] hits
    // pre-filter + fp64 stage of one pass, queued on the scan stream (events 0, 1, 2 around the two stages)
    hipEvent_t *ev = c->ev;                                        // stage events of this scan (a pending scan's own set, below)
    uint64_t cand_static = 0;                                      // the pre-filter waves' own first candidate blocks (set by front)
    auto front = [&](const HitOut &H) -> int {
        cand_static = 0;
        he = hipMemsetAsync(sc.counters, 0, 8 * sizeof(unsigned long long), c->stream);
        if (he != hipSuccess) { set_error("memset failed: %s", hipGetErrorString(he)); return MS_ERR_RUNTIME; }
        (void) hipEventRecord(ev[0], c->stream);
        if (const char *e = measure_env("MS_EXTRA_MEMSETS"))       // measurement: what a tiny kernel costs at this point of a batch stream
            for (int i = 0; i < atoi(e); i++) (void) hipMemsetAsync(sc.counters + 4, 0, 8, c->stream);
        if (n_tiles > 0) {
            PfArgs A;
            A.codes = S.codes; A.nmask = S.nmask; A.n_bases = S.n_bases; A.no_emit = pf_no_emit; A.skip_alln = plan.alln_can_hit ? 0 : 1;
            A.tables = pwms->d_tables; A.tiles = pwms->d_tiles; A.lut_off16 = lut_off16; A.stage_off16 = lut_off16 + (uint32_t) (kF6LutBytes / 16);
            A.emit_off16 = A.stage_off16 + (uint32_t) (kPfStageBytes / 16);
            A.onehot_off16 = A.emit_off16 + (uint32_t) (kPfEmitBytes / 16);
            A.rare_off16 = A.onehot_off16 + (uint32_t) ((wide_layout ? 0 : kPfOnehotBytes) / 16);      // (a wide plan's kernels have no one-hot array)
            A.rare_cap = rare_cap;
            A.cand = sc.cand; A.n_cand = sc.counters; A.cand_cap = sc.cand_cap; A.cand_block = cand_block;
            const size_t counter_words = (size_t) n_tiles * kPfCounters * 16;            // kPfCounters words per tile, 64 bytes apart
            if (counter_words > sc.chunk_counters_cap) {
                dev_free(sc.chunk_counters);
                sc.chunk_counters_cap = 0;
                if ((rc = dev_alloc(&sc.chunk_counters, counter_words + 16))) return rc;
                sc.chunk_counters_cap = counter_words + 16;
            }
            A.chunk_counter = sc.chunk_counters;
            A.wave_passes = (int) pf_wave_passes;
            A.use_counters = counter_used ? 1 : 0;
            if (counter_used) {
                he = hipMemsetAsync(A.chunk_counter, 0, sizeof(unsigned int) * counter_words, c->stream);
                if (he != hipSuccess) { set_error("memset failed: %s", hipGetErrorString(he)); return MS_ERR_RUNTIME; }
            }
            A.clk = nullptr;
            A.cls_clk = pf_clock && atoi(measure_env("MS_PF_CLOCK")) == 2;
            if (pf_clock) {
                clk_blocks = bpt * n_tiles;
                if (!d_clk && (rc = dev_alloc(&d_clk, (size_t) kPfClkWords * clk_blocks))) return rc;
                (void) hipMemsetAsync(d_clk, 0, sizeof(unsigned long long) * kPfClkWords * clk_blocks, c->stream);
                A.clk = d_clk;
            }
            {
                const int64_t n_chunks = (S.n_bases + kPfThreads - 1) / kPfThreads;
                cand_static = (uint64_t) std::min<int64_t>(bpt, n_chunks) * n_tiles * (kPfThreads / 64) * cand_block;
                A.cand_static = cand_static;
            }
            bool wide = false;
            for (const TileDesc &t : plan.tiles) wide = wide || t.max_nk > 2;
            const int floor_ = wide ? 0 : pf_floor;
            const bool dense = pf_dense && !wide && !pf_meas && !floor_;
            const int li = (wide ? 2 : 0) + (pf_meas ? 1 : 0) + (dense ? 4 : 0) + 8 * floor_;
            if (lds_bytes > c->lds_set[li]) {
                if ((rc = prefilter_set_lds(wide, pf_meas, dense, lds_bytes, floor_))) return rc;
                c->lds_set[li] = lds_bytes;
            }
            if ((rc = launch_prefilter(A, wide, pf_meas, dense, bpt, n_tiles, lds_bytes, c->stream, floor_))) return rc;
            stt.pf_engine = dense ? 4 : 3;
        }
        (void) hipEventRecord(ev[1], c->stream);
        if (!plan.fast_motifs.empty())                 // (few blocks for a small scan measured slower: the kernel is a chain of dependent gathers and wants every record in flight at once)
        {
            // long lists: chunks of 4096 candidates in motif order with their windows carried along (fewer cache lines per read); short ones: list order, many small blocks
            double rs_min = 2.0e6;
            if (const char *e = measure_env("MS_RESCORE_SORTED_MIN")) rs_min = atof(e);       // test aid / A-B: 0 = always, 1e30 = never
            if (1.5e-4 * (double) fast_windows >= rs_min) {
                if (!c->rc_lds_set) { if ((rc = rescore_carry_set_lds())) return rc; c->rc_lds_set = true; }
                if ((rc = launch_rescore_carry(S, Pw, sc.cand, sc.counters, cand_static, sc.cand_cap, pwms->d_field_meta, strand_mask, H, c->n_cu, c->stream))) return rc;
            } else if ((rc = launch_rescore(S, Pw, sc.cand, sc.counters, cand_static, sc.cand_cap, pwms->d_field_meta, strand_mask, H, c->n_cu * 8, c->stream))) return rc;
        }
        if (!plan.exact_motifs.empty())
            {
                int max_w = 0;
                for (int32_t m : plan.exact_motifs) max_w = std::max(max_w, (int) pwms->widths[m]);
                if ((rc = launch_exact_all(S, Pw, pwms->d_exact_motifs, (int32_t) plan.exact_motifs.size(), strand_mask, H, c->stream, max_w))) return rc;
            }
        (void) hipEventRecord(ev[2], c->stream);
        return MS_OK;
    };
    // ordering + coordinates of the first n_sort slots of the hit list into the result block (events 3, 4, 5); n_dev != nullptr:
    // only the device knows how many of them are hits (the rest are all-ones keys, which sort behind every hit)
    const int end_bit = gbits + 1 + mbits;
    // the radix passes cover the key bits above kSortLowBits, sort_fixup_kernel the rest (MS_SORT_FULL: all bits by radix passes;
    // a short hit list is ordered by launch latencies, not passes: one kernel fewer matters more there)
    const int sort_begin_large = (end_bit > 2 * kSortLowBits && !measure_env("MS_SORT_FULL")) ? kSortLowBits : 0;
    bool queue_only = false;                     // this back() belongs to a scan that is only queued (scan_complete finishes it)
    // counts only: the flag map costs ~2 x P x R bytes of traffic (clear + count), the ordering it replaces ~200 bytes per hit -- so the map is
    // taken where the set holds at least one hit per ~50 (motif, region) cells (configs[3]: one per 9; a million 50-bp regions: one per 100,
    // where the sort is the cheaper way to the same counts).  Decided once per scan, before the result block is sized.
    const bool counts_ok = (flags & MS_SCAN_COUNTS_ONLY_INTERNAL) && !(flags & MS_SCAN_RAW_INTERNAL) && count_only_supported(pwms->P, seqs->R, pbits);
    bool counts_fast = false;
    auto decide_counts = [&](size_t n_expected) { counts_fast = counts_ok && (double) pwms->P * (double) seqs->R <= 50.0 * (double) std::max<size_t>(n_expected, 1); };
    auto back = [&](size_t n_sort, const unsigned long long *n_dev) -> int {
        he = hipMemsetAsync(raw->d_region_counts, 0, 8 * ((size_t) pwms->P + 1), c->stream);
        if (he != hipSuccess) { set_error("memset failed: %s", hipGetErrorString(he)); return MS_ERR_RUNTIME; }
        (void) hipEventRecord(ev[3], c->stream);
        raw->counts_only = false;                // (a second, exactly-sized run after a failed prediction decides again)
        if (counts_fast) {
            // the bitmap + the per-motif site counters in one pooled block the result owns (a queued scan's kernels outlive this call)
            const size_t words = count_only_bitmap_words(pwms->P, seqs->R);
            if ((rc = pool_alloc(c, words * 4 + 256 + 8 * (size_t) pwms->P, &raw->coord_blk, &raw->coord_bytes))) return rc;
            uint32_t *bitmap = static_cast<uint32_t *>(raw->coord_blk);
            unsigned long long *motif_hits = reinterpret_cast<unsigned long long *>(static_cast<char *>(raw->coord_blk) + ((words * 4 + 255) & ~(size_t) 255));
            if ((rc = launch_count_only(sc.keys, (int64_t) n_sort, n_dev, gbits, pbits, seqs->R, pwms->P, bitmap, raw->d_region_counts, motif_hits, raw->d_motif_first, c->stream))) return rc;
            raw->counts_only = true;
            (void) hipEventRecord(ev[4], c->stream);
            (void) hipEventRecord(ev[5], c->stream);
            if (!queue_only) {
                he = hipMemcpyAsync(raw->motif_offsets.data(), raw->d_motif_first, raw->motif_offsets.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
                if (he != hipSuccess) { set_error("copy failed: %s", hipGetErrorString(he)); return MS_ERR_RUNTIME; }
            }
            return MS_OK;
        }
        size_t fixup_min = (size_t) 1 << 20;
        if (const char *e = measure_env("MS_SORT_FIXUP_MIN")) fixup_min = (size_t) std::max(0, atoi(e));     // test aid: the fix-up form on short lists too
        const int sort_begin = n_sort >= fixup_min ? sort_begin_large : 0;
        if (n_sort > 0) {
            size_t need = 0;
            if ((rc = sort_hit_pairs(nullptr, &need, sc.keys, sc.keys_sorted, sc.vals, raw->d_score, n_sort, sort_begin, end_bit, c->stream))) return rc;
            if (need > sc.sort_tmp_bytes) {
                if (sc.sort_tmp) (void) hipFree(sc.sort_tmp);
                sc.sort_tmp = nullptr; sc.sort_tmp_bytes = 0;
                unsigned char *tmp = nullptr;
                if ((rc = dev_alloc(&tmp, need))) return rc;
                sc.sort_tmp = tmp;
                sc.sort_tmp_bytes = need;
            }
            size_t have = sc.sort_tmp_bytes;
            if ((rc = sort_hit_pairs(sc.sort_tmp, &have, sc.keys, sc.keys_sorted, sc.vals, raw->d_score, n_sort, sort_begin, end_bit, c->stream))) return rc;
            if (sort_begin && (rc = launch_sort_fixup(sc.keys_sorted, raw->d_score, (int64_t) n_sort, n_dev, c->stream))) return rc;
        }
        (void) hipEventRecord(ev[4], c->stream);
        if ((rc = launch_finalize(sc.keys_sorted, (int64_t) n_sort, n_dev, gbits, rbits, pbits, pwms->P, S, raw->d_seq_idx, raw->d_pos,
                                  raw->d_strand, raw->d_motif_first, raw->d_region_counts, c->stream))) return rc;
        if ((flags & (MS_SCAN_PACK_INTERNAL | MS_SCAN_PACK12_INTERNAL)) && n_sort > 0) {
            const size_t n_round = (n_sort + 65535) & ~(size_t) 65535;
            if ((rc = pool_alloc(c, 8 * n_round + 256, &raw->coord_blk, &raw->coord_bytes))) return rc;
            raw->d_coord = static_cast<uint64_t *>(raw->coord_blk);
            raw->d_coord_bad = reinterpret_cast<unsigned int *>(raw->d_coord + n_round);
            he = hipMemsetAsync(raw->d_coord_bad, 0, sizeof(unsigned int), c->stream);
            if (he != hipSuccess) { set_error("memset failed: %s", hipGetErrorString(he)); return MS_ERR_RUNTIME; }
            // the 4-byte form when asked for and the set's largest region index and position fit 31 bits beside the strand bit
            raw->coord_shift = 0;
            if (flags & MS_SCAN_PACK12_INTERNAL) {
                const int64_t max_len = seqs->len_sorted.empty() ? 0 : seqs->len_sorted.back();
                int pb = 1, rb = 1;
                while ((1LL << pb) < std::max<int64_t>(max_len, 1)) pb++;
                while ((1LL << rb) < std::max<int64_t>(seqs->R, 1)) rb++;
                if (rb + pb + 1 <= 32) raw->coord_shift = pb + 1;
            }
            if ((rc = launch_pack_hits((int64_t) n_sort, n_dev, raw->d_seq_idx, raw->d_pos, raw->d_strand, raw->d_coord, raw->d_coord_bad, c->stream, raw->coord_shift))) return rc;
        }
        (void) hipEventRecord(ev[5], c->stream);
        // the complete per-motif offsets are on the device; one copy brings them to the host.  (Not for a scan that is only being
        // QUEUED: the destination is pageable memory, for which the "async" copy makes the host wait for everything queued before
        // it -- scan_complete fetches the offsets once the scan is done.)
        if (!queue_only) {
            he = hipMemcpyAsync(raw->motif_offsets.data(), raw->d_motif_first, raw->motif_offsets.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
            if (he != hipSuccess) { set_error("copy failed: %s", hipGetErrorString(he)); return MS_ERR_RUNTIME; }
        }
        return MS_OK;
    };
    auto result_block = [&](size_t n) -> int {
        void *blk = nullptr;
        size_t got = 0;
        if ((rc = pool_alloc(c, result_block_bytes(pwms->P, n), &blk, &got))) return rc;
        raw->block = blk;
        raw->block_bytes = got;
        result_carve(raw, blk, n);
        return MS_OK;
    };
    auto read_clock = [&]() {
        if (!d_clk) return;                                  // median over blocks of cycles per 10 ns tick
        std::vector<unsigned long long> h((size_t) kPfClkWords * clk_blocks);
        if (hipMemcpy(h.data(), d_clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            std::vector<double> mhz;
            for (int b = 0; b < clk_blocks; b++)
                if (h[(size_t) kPfClkWords * b + 1] > 0) mhz.push_back(100.0 * (double) h[(size_t) kPfClkWords * b] / (double) h[(size_t) kPfClkWords * b + 1]);
            if (!mhz.empty()) { std::sort(mhz.begin(), mhz.end()); stt.pf_clock_mhz = mhz[mhz.size() / 2]; }
            if (atoi(measure_env("MS_PF_CLOCK")) == 2) {              // where a wave's cycles go: per class of the first LDS tile, and outside the classes
                double tot = 0, cls[kMaxClasses] = {0, 0, 0, 0, 0, 0};
                for (int b = 0; b < clk_blocks / (int) plan.tiles.size(); b++) {
                    tot += (double) h[(size_t) kPfClkWords * b];
                    for (int i = 0; i < kMaxClasses; i++) cls[i] += (double) h[(size_t) kPfClkWords * b + 2 + i];
                }
                fprintf(stderr, "pf wave-0 cycles:");
                double in = 0;
                const TileDesc &t0 = plan.tiles[0];
                for (int i = 0; i < t0.n_classes; i++) {
                    fprintf(stderr, " [%s nk %d x %d row tiles] %.1f %%", t0.cls[i].paired ? "paired" : "plain", t0.cls[i].nk, t0.cls[i].n_row_tiles, 100.0 * cls[i] / tot);
                    in += cls[i];
                }
                fprintf(stderr, " [outside the classes] %.1f %%\n", 100.0 * (tot - in) / tot);
            }
        }
        dev_free(d_clk);
    };
    auto finish = [&](unsigned long long n_cand, unsigned long long n_hits, bool with_back) {
        finish_scan(raw, ev, pwms, seqs->n_bases, seqs->R, strand_mask, exact_only, n_cand, n_hits, with_back);
    };

    // ---- one-sync form: sizes PREDICTED from the previous scan of these PWMs.  The hit density of a motif set at its cutoffs is a
    // property of the set (p-value x windows x strands): consecutive scans -- the batches of a stream, the input and control sets of
    // a run, a benchmark's steps -- repeat it within a fraction of a percent.  The result block, the sort and the coordinate kernel
    // are sized for the predicted count plus a margin, every launch is queued at once (the fp64 stage's real count stays on the
    // device: the unused slots are filled with all-ones keys that sort last, finalize reads the count there), and ONE
    // synchronisation at the very end validates the prediction.  A wrong prediction (count above the margin, or a scratch buffer
    // too small) costs a second, exactly-sized run below and doubles the margin of the next scans.
    const bool predicted = pwms->pred_density >= 0 && pwms->pred_strand == strand_mask && pwms->pred_cutoff_version == pwms->cutoff_version &&
                           pwms->pred_exact_only == exact_only && !(flags & (MS_SCAN_RAW_INTERNAL | MS_SCAN_NO_PREDICT_INTERNAL)) && !pf_meas &&
                           !measure_env("MS_NO_PREDICT");
    if (predicted) {
        if (pend) ev = pend->ev;
        const double mu = pwms->pred_density * (double) stt.n_windows;
        const size_t n_pred = (size_t) std::min<double>(mu * (1.0 + pwms->pred_margin) + 6.0 * std::sqrt(mu + 1.0) + 256.0, 3.0e9);
        want_hits = std::max(want_hits, n_pred);
        if ((rc = scratch_reserve(sc, want_cand, want_hits))) return fail(rc);
        decide_counts(n_pred);
        if ((rc = result_block(counts_fast ? 1 : n_pred))) return fail(rc);
        stt.n_passes = 1;
        HitOut H;
        H.keys = sc.keys; H.vals = sc.vals; H.n_hits = sc.counters + 1; H.cap = sc.hit_cap; H.gbits = gbits; H.pbits = pbits;
        if (measure_env("MS_TRACE_BLOCKS"))
            fprintf(stderr, "MSBLK launch res=%p (%zu bytes) codes=%p nmask=%p blkinfo=%p offsets=%p cand=%p keys=%p\n", raw->block, raw->block_bytes, (const void *) S.codes, (const void *) S.nmask,
                    (const void *) S.blkinfo, (const void *) S.offsets, (void *) sc.cand, (void *) sc.keys);
        if ((rc = front(H))) return fail(rc);
        if ((rc = launch_fill_tail(sc.keys, sc.counters + 1, n_pred, c->stream))) return fail(rc);
        queue_only = pend != nullptr;
        rc = back(n_pred, sc.counters + 1);
        queue_only = false;                      // (the exactly-sized form below always runs to the end, with or without a PendingScan slot:
        if (rc) return fail(rc);                 //  the first batch of a stream, whose PWM set has no prediction yet, came back with all-zero offsets)
        if (pend) {
            // queued, not waited for: the owner queues its next scan behind this one first (everything is in order on one stream:
            // the next scan's kernels only touch the shared scratch after this scan's are done; a scratch buffer that has to grow
            // is freed by hipFree, which waits for the device)
            // the per-motif offsets too, in stream order, into the slot's own pinned words: scan_complete used to fetch them with a blocking hipMemcpy
            // into pageable memory once the scan was done -- one more synchronous driver call per batch on the scan stage's thread (same pass time
            // either way, profiles/r06z_offsets_ab.log; kept because nothing on that thread should block that need not).  The per-motif region counts
            // ride along: the stream's copy-out stage brings them to the host with every batch, and its own small copy is one more blit kernel that
            // has to find a CU beside the running pre-filter
            const size_t n_off = raw->motif_offsets.size(), n_words = n_off + (size_t) pwms->P;
            pend->offsets_queued = !measure_env("MS_OFFSETS_BLOCKING");
            if (pend->offsets_queued && pend->h_offsets_cap < n_words) {
                if (pend->h_offsets) (void) hipHostFree(pend->h_offsets);
                pend->h_offsets = nullptr; pend->h_offsets_cap = 0;
                he = hipHostMalloc(&pend->h_offsets, (n_words + 64) * sizeof(int64_t));
                if (he != hipSuccess) { set_error("out of pinned host memory"); return fail(MS_ERR_NOMEM); }
                pend->h_offsets_cap = n_words + 64;
            }
            he = pend->offsets_queued ? hipMemcpyAsync(pend->h_offsets, raw->d_motif_first, n_off * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream) : hipSuccess;
            if (he == hipSuccess && pend->offsets_queued && pwms->P > 0)
                he = hipMemcpyAsync(pend->h_offsets + n_off, raw->d_region_counts, (size_t) pwms->P * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(pend->h_counters, sc.counters, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipEventRecord(pend->done, c->stream);
            if (he != hipSuccess) { set_error("scan kernels failed: %s", hipGetErrorString(he)); return fail(MS_ERR_RUNTIME); }
            pend->cand_cap = sc.cand_cap;
            pend->hit_cap = sc.hit_cap;
            pend->active = true;
            pend->raw = raw;
            pend->n_pred = n_pred;
            pend->cand_static = cand_static;
            pend->n_bases = seqs->n_bases;
            pend->R = seqs->R;
            pend->strand_mask = strand_mask;
            pend->exact_only = exact_only;
            *out = nullptr;
            return MS_SCAN_PENDING;
        }
        he = hipMemcpyAsync(sc.h_counters, sc.counters, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { set_error("scan kernels failed: %s", hipGetErrorString(he)); return fail(MS_ERR_RUNTIME); }
        const unsigned long long n_cand = cand_static + sc.h_counters[0], n_hits = sc.h_counters[1];
        if (n_cand <= sc.cand_cap && n_hits <= sc.hit_cap && n_hits <= n_pred) {
            pwms->pred_margin = std::max(0.04, pwms->pred_margin * 0.9);
            read_clock();
            finish(n_cand, n_hits, true);
            *out = raw;
            return MS_OK;
        }
        // the prediction failed: give the blocks back and run again with exact sizes
        pool_free(c, raw->block, raw->block_bytes);
        raw->block = nullptr;
        raw->block_bytes = 0;
        if (raw->coord_blk) { pool_free(c, raw->coord_blk, raw->coord_bytes); raw->coord_blk = nullptr; raw->d_coord = nullptr; raw->d_coord_bad = nullptr; }
        pwms->pred_margin = std::min(1.0, pwms->pred_margin * 2.0);
        stt.n_passes = 1;
        want_cand = std::max<size_t>(sc.cand_cap, (size_t) (n_cand + n_cand / 16 + 1024));
        const unsigned long long hit_need = n_cand > sc.cand_cap ? std::max<unsigned long long>(n_hits, 2 * n_cand) : n_hits;
        want_hits = std::max<size_t>(sc.hit_cap, (size_t) (hit_need + hit_need / 16 + 1024));
    }

    unsigned long long n_cand = 0, n_hits = 0;
    for (int pass = 1;; pass++) {
        if ((rc = scratch_reserve(sc, want_cand, want_hits))) return fail(rc);
        stt.n_passes += 1;
        HitOut H;
        H.keys = sc.keys; H.vals = sc.vals; H.n_hits = sc.counters + 1; H.cap = sc.hit_cap; H.gbits = gbits; H.pbits = pbits;
        if ((rc = front(H))) return fail(rc);
        he = hipMemcpyAsync(sc.h_counters, sc.counters, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { set_error("scan kernels failed: %s", hipGetErrorString(he)); return fail(MS_ERR_RUNTIME); }
        n_cand = cand_static + sc.h_counters[0];
        n_hits = sc.h_counters[1];
        if (n_cand <= sc.cand_cap && n_hits <= sc.hit_cap) break;
        if (pass >= 8) { set_error("scan buffers kept overflowing (%llu candidates, %llu hits)", n_cand, n_hits); return fail(MS_ERR_RUNTIME); }
        // a buffer was too small: the counters hold the exact need (a truncated candidate list
        // under-reports hits, so leave head room there) -- grow and run the pass again
        want_cand = std::max<size_t>(sc.cand_cap, (size_t) (n_cand + n_cand / 16 + 1024));
        const unsigned long long hit_need = n_cand > sc.cand_cap ? std::max<unsigned long long>(n_hits, 2 * n_cand) : n_hits;
        want_hits = std::max<size_t>(sc.hit_cap, (size_t) (hit_need + hit_need / 16 + 1024));
    }
    read_clock();

    if (flags & MS_SCAN_RAW_INTERNAL) {                      // the caller takes the unordered hits from the scratch
        finish(n_cand, n_hits, false);
        raw->raw_gbits = gbits;
        raw->raw_pbits = pbits;
        *out = raw;
        return MS_OK;
    }

    // one pooled block for everything the result owns
    decide_counts((size_t) n_hits);
    if ((rc = result_block(counts_fast ? 1 : (size_t) n_hits))) return fail(rc);
    if ((rc = back((size_t) n_hits, nullptr))) return fail(rc);
    he = hipStreamSynchronize(c->stream);
    if (he != hipSuccess) { set_error("finalize failed: %s", hipGetErrorString(he)); return fail(MS_ERR_RUNTIME); }
    finish(n_cand, n_hits, true);
    *out = raw;
    return MS_OK;
}

}  // namespace ms

extern "C" {

int ms_scan(const ms_pwmset *pwms_c, const ms_seqset *seqs, int strand_mask, uint32_t flags, ms_result **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    if (!pwms_c || !seqs) { set_error("NULL handle"); return MS_ERR_INVALID; }
    if (strand_mask < 1 || strand_mask > 3) { set_error("invalid strand mask %d (1 '+', 2 '-', 3 both)", strand_mask); return MS_ERR_INVALID; }
    if (flags & ~(uint32_t) (MS_SCAN_EXACT_ONLY | MS_SCAN_COUNTS_ONLY)) { set_error("unknown scan flags 0x%x", flags); return MS_ERR_INVALID; }
    ms_pwmset *pwms = const_cast<ms_pwmset *>(pwms_c);       // lazily cached device copies / plan
    DeviceCtx *c;
    int rc = get_ctx(seqs->device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk_dev(c->mu);
    std::lock_guard<std::mutex> lk_pwm(pwms->mu);
    const uint32_t internal = (flags & MS_SCAN_EXACT_ONLY) | ((flags & MS_SCAN_COUNTS_ONLY) ? MS_SCAN_COUNTS_ONLY_INTERNAL : 0u);
    return scan_locked(c, pwms, seqs, strand_mask, internal, out);
}

int ms_result_num_hits(const ms_result *r, int64_t *n_hits) {
    if (!r || !n_hits) { set_error("NULL argument"); return MS_ERR_INVALID; }
    *n_hits = r->n_hits;
    return MS_OK;
}

int ms_result_motif_offsets(const ms_result *r, int64_t *out) {
    if (!r || !out) { set_error("NULL argument"); return MS_ERR_INVALID; }
    std::memcpy(out, r->motif_offsets.data(), r->motif_offsets.size() * sizeof(int64_t));
    return MS_OK;
}

int ms_result_hits(const ms_result *r, int64_t *seq_idx, int64_t *pos, double *score, int8_t *strand) {
    if (!r) { set_error("NULL argument"); return MS_ERR_INVALID; }
    if (r->invalid) { set_error("result of a no-emit measurement run (MS_MEASURE=1 MS_PF_NOEMIT=1) holds no hits"); return MS_ERR_INVALID; }
    if (r->counts_only) { set_error("a counts-only result (MS_SCAN_COUNTS_ONLY, a counts-only batch or sweep span of a stream) holds the per-motif region counts and site numbers, no site arrays"); return MS_ERR_INVALID; }
    if (r->n_hits == 0) return MS_OK;
    MS_HIP(hipSetDevice(r->device));
    const size_t n = (size_t) r->n_hits;
    if (seq_idx) MS_HIP(hipMemcpy(seq_idx, r->d_seq_idx, n * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (pos) MS_HIP(hipMemcpy(pos, r->d_pos, n * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (score) MS_HIP(hipMemcpy(score, r->d_score, n * sizeof(double), hipMemcpyDeviceToHost));
    if (strand) MS_HIP(hipMemcpy(strand, r->d_strand, n * sizeof(int8_t), hipMemcpyDeviceToHost));
    return MS_OK;
}

// Hit arrays in library-owned PINNED host memory (one D2H copy at PCIe rate instead of four copies
// into pageable buffers).  The pointers stay valid until the result is freed or de-duplicated.
// The copy runs on the device's copy-out stream, beside whatever scan is running.
int ms_result_hits_host(ms_result *r, const int64_t **seq_idx, const int64_t **pos, const double **score,
                        const int8_t **strand) {
    if (!r) { set_error("NULL argument"); return MS_ERR_INVALID; }
    if (r->invalid) { set_error("result of a no-emit measurement run (MS_MEASURE=1 MS_PF_NOEMIT=1) holds no hits"); return MS_ERR_INVALID; }
    if (r->counts_only) { set_error("a counts-only result (MS_SCAN_COUNTS_ONLY, a counts-only batch or sweep span of a stream) holds the per-motif region counts and site numbers, no site arrays"); return MS_ERR_INVALID; }
    const size_t n = (size_t) r->n_hits;
    const size_t n_round = (n + 65535) & ~(size_t) 65535;
    const size_t bytes = 25 * n_round + 64;
    if (r->h_pinned_hits != r->n_hits || !r->h_pinned || r->h_packed) {
        if (r->h_pinned && r->h_pinned_bytes < bytes) { pinned_free(r->h_pinned, r->h_pinned_bytes); r->h_pinned = nullptr; }
        if (!r->h_pinned) {
            r->h_pinned = pinned_alloc(bytes, &r->h_pinned_bytes);
            if (!r->h_pinned) { set_error("pinned host allocation of %zu bytes failed", bytes); return MS_ERR_NOMEM; }
        }
        if (n > 0) {
            DeviceCtx *c;
            int rc = get_ctx(r->device, &c);
            if (rc) return rc;
            // the device block may have been carved for more hits than there are (a predicted-size scan sizes it by a bound):
            // four copies, packed n_round elements apart on the host
            char *hb = static_cast<char *>(r->h_pinned);
            const hipStream_t down = c->stream_down;
            MS_HIP(hipMemcpyAsync(hb, r->d_seq_idx, 8 * n, hipMemcpyDeviceToHost, down));
            MS_HIP(hipMemcpyAsync(hb + 8 * n_round, r->d_pos, 8 * n, hipMemcpyDeviceToHost, down));
            MS_HIP(hipMemcpyAsync(hb + 16 * n_round, r->d_score, 8 * n, hipMemcpyDeviceToHost, down));
            MS_HIP(hipMemcpyAsync(hb + 24 * n_round, r->d_strand, n, hipMemcpyDeviceToHost, down));
            MS_HIP(hipStreamSynchronize(down));
        }
        r->h_pinned_hits = r->n_hits;
        r->h_packed = false;
    }
    char *b = static_cast<char *>(r->h_pinned);
    if (seq_idx) *seq_idx = reinterpret_cast<const int64_t *>(b);
    if (pos) *pos = reinterpret_cast<const int64_t *>(b + 8 * n_round);
    if (score) *score = reinterpret_cast<const double *>(b + 16 * n_round);
    if (strand) *strand = reinterpret_cast<const int8_t *>(b + 24 * n_round);
    return MS_OK;
}

// The same in 16 bytes per hit: coord = seq_idx << 32 | pos << 1 | (strand - 1), score.  The device packs the three
// coordinate arrays into one word per hit first, so only 16 of the 25 bytes cross the host link.
int ms_result_hits_packed_host(ms_result *r, const uint64_t **coord, const double **score) {
    if (!r) { set_error("NULL argument"); return MS_ERR_INVALID; }
    if (r->invalid) { set_error("result of a no-emit measurement run (MS_MEASURE=1 MS_PF_NOEMIT=1) holds no hits"); return MS_ERR_INVALID; }
    if (r->counts_only) { set_error("a counts-only result (MS_SCAN_COUNTS_ONLY, a counts-only batch or sweep span of a stream) holds the per-motif region counts and site numbers, no site arrays"); return MS_ERR_INVALID; }
    if (r->d_coord && r->coord_shift) { set_error("this result holds the 12-byte compact form (MS_STREAM_PACKED12): read it with ms_result_hits_packed12_host"); return MS_ERR_INVALID; }
    const size_t n = (size_t) r->n_hits;
    const size_t n_round = (n + 65535) & ~(size_t) 65535;
    const size_t bytes = 16 * n_round + 64;
    if (r->h_pinned_hits != r->n_hits || !r->h_pinned || !r->h_packed || r->h_coord_shift) {
        if (r->h_pinned && r->h_pinned_bytes < bytes) { pinned_free(r->h_pinned, r->h_pinned_bytes); r->h_pinned = nullptr; }
        if (!r->h_pinned) {
            r->h_pinned = pinned_alloc(bytes, &r->h_pinned_bytes);
            if (!r->h_pinned) { set_error("pinned host allocation of %zu bytes failed", bytes); return MS_ERR_NOMEM; }
        }
        if (n > 0 && r->d_coord) {                      // the scan made the coordinate words already: copies only
            DeviceCtx *c;
            int rc = get_ctx(r->device, &c);
            if (rc) return rc;
            unsigned int bad = 0;
            const hipStream_t down = c->stream_down;
            char *hb = static_cast<char *>(r->h_pinned);
            hipError_t he = hipMemcpyAsync(hb, r->d_coord, 8 * n, hipMemcpyDeviceToHost, down);
            if (he == hipSuccess) he = hipMemcpyAsync(hb + 8 * n_round, r->d_score, 8 * n, hipMemcpyDeviceToHost, down);
            if (he == hipSuccess) he = hipMemcpyAsync(&bad, r->d_coord_bad, sizeof(unsigned int), hipMemcpyDeviceToHost, down);
            const hipError_t hs = hipStreamSynchronize(down);
            if (he != hipSuccess || hs != hipSuccess) { set_error("compact copy-out failed: %s", hipGetErrorString(he != hipSuccess ? he : hs)); return MS_ERR_RUNTIME; }
            if (bad) { set_error("a hit does not fit the compact form (needs seq_idx < 2^32 and pos < 2^31): use ms_result_hits_host"); return MS_ERR_INVALID; }
        } else if (n > 0) {
            DeviceCtx *c;
            int rc = get_ctx(r->device, &c);
            if (rc) return rc;
            void *blk = nullptr;
            size_t got = 0;
            if ((rc = pool_alloc(c, 8 * n_round + 256, &blk, &got))) return rc;
            uint64_t *d_coord = static_cast<uint64_t *>(blk);
            unsigned int *d_bad = reinterpret_cast<unsigned int *>(d_coord + n_round);
            unsigned int bad = 0;
            const hipStream_t down = c->stream_down;
            hipError_t he = hipMemsetAsync(d_bad, 0, sizeof(unsigned int), down);
            if (he == hipSuccess) rc = launch_pack_hits((int64_t) n, nullptr, r->d_seq_idx, r->d_pos, r->d_strand, d_coord, d_bad, down);
            char *hb = static_cast<char *>(r->h_pinned);
            if (he == hipSuccess && !rc) he = hipMemcpyAsync(hb, d_coord, 8 * n, hipMemcpyDeviceToHost, down);
            if (he == hipSuccess && !rc) he = hipMemcpyAsync(hb + 8 * n_round, r->d_score, 8 * n, hipMemcpyDeviceToHost, down);
            if (he == hipSuccess && !rc) he = hipMemcpyAsync(&bad, d_bad, sizeof(unsigned int), hipMemcpyDeviceToHost, down);
            const hipError_t hs = hipStreamSynchronize(down);
            pool_free(c, blk, got);
            if (rc) return rc;
            if (he != hipSuccess || hs != hipSuccess) { set_error("compact copy-out failed: %s", hipGetErrorString(he != hipSuccess ? he : hs)); return MS_ERR_RUNTIME; }
            if (bad) { set_error("a hit does not fit the compact form (needs seq_idx < 2^32 and pos < 2^31): use ms_result_hits_host"); return MS_ERR_INVALID; }
        }
        r->h_pinned_hits = r->n_hits;
        r->h_packed = true;
        r->h_coord_shift = 0;
    }
    char *b = static_cast<char *>(r->h_pinned);
    if (coord) *coord = reinterpret_cast<const uint64_t *>(b);
    if (score) *score = reinterpret_cast<const double *>(b + 8 * n_round);
    return MS_OK;
}

// The 12-byte form: coord32[i] = seq_idx << shift | pos << 1 | (strand - 1) and score[i].  A batch stream made with MS_STREAM_PACKED12 has
// produced the words during the scan when the batch fits (copies only here); a result that holds none is packed now if it fits.
int ms_result_hits_packed12_host(ms_result *r, const uint32_t **coord, const double **score, int32_t *shift_out) {
    if (!r) { set_error("NULL argument"); return MS_ERR_INVALID; }
    if (r->invalid) { set_error("result of a no-emit measurement run (MS_MEASURE=1 MS_PF_NOEMIT=1) holds no hits"); return MS_ERR_INVALID; }
    if (r->counts_only) { set_error("a counts-only result (MS_SCAN_COUNTS_ONLY, a counts-only batch or sweep span of a stream) holds the per-motif region counts and site numbers, no site arrays"); return MS_ERR_INVALID; }
    if (r->d_coord && !r->coord_shift) { set_error("this result holds the 16-byte compact form (its batch did not fit 31 bits of region index and position): read it with ms_result_hits_packed_host"); return MS_ERR_INVALID; }
    const size_t n = (size_t) r->n_hits;
    const size_t n_round = (n + 65535) & ~(size_t) 65535;
    const size_t bytes = 12 * n_round + 64;
    if (r->h_pinned_hits != r->n_hits || !r->h_pinned || !r->h_packed || !r->h_coord_shift) {
        if (r->h_pinned && r->h_pinned_bytes < bytes) { pinned_free(r->h_pinned, r->h_pinned_bytes); r->h_pinned = nullptr; }
        if (!r->h_pinned) {
            r->h_pinned = pinned_alloc(bytes, &r->h_pinned_bytes);
            if (!r->h_pinned) { set_error("pinned host allocation of %zu bytes failed", bytes); return MS_ERR_NOMEM; }
        }
        int shift = r->coord_shift;
        DeviceCtx *c = nullptr;
        int rc = n > 0 ? get_ctx(r->device, &c) : MS_OK;
        if (rc) return rc;
        void *blk = nullptr;
        size_t got = 0;
        const uint64_t *d_words = r->d_coord;
        const unsigned int *d_bad = r->d_coord_bad;
        if (n > 0 && !r->d_coord) {                     // no words yet: the shift from the hits themselves (largest region index / position)
            const hipStream_t down = c->stream_down;
            int64_t last_seq = 0;
            // the hits of a motif are ordered by region: the largest region index is the maximum over the motifs' last hits -- P small
            // copies would do; R is known to the result, so the bound is R - 1 whatever the hits are
            last_seq = std::max<int64_t>(r->R - 1, 0);
            int rb = 1;
            while ((1LL << rb) <= last_seq) rb++;
            if (rb >= 31) { set_error("the result's region indices need %d bits: the 12-byte compact form does not fit (use ms_result_hits_packed_host)", rb); return MS_ERR_INVALID; }
            shift = 32 - rb;                            // positions get every remaining bit; the kernel flags a position that does not fit
            if ((rc = pool_alloc(c, 4 * n_round + 256, &blk, &got))) return rc;
            unsigned int *bad_w = reinterpret_cast<unsigned int *>(static_cast<char *>(blk) + 4 * n_round);
            hipError_t he = hipMemsetAsync(bad_w, 0, sizeof(unsigned int), down);
            if (he == hipSuccess) rc = launch_pack_hits((int64_t) n, nullptr, r->d_seq_idx, r->d_pos, r->d_strand, static_cast<uint64_t *>(blk), bad_w, down, shift);
            if (he != hipSuccess || rc) { pool_free(c, blk, got); if (!rc) { set_error("memset failed: %s", hipGetErrorString(he)); rc = MS_ERR_RUNTIME; } return rc; }
            d_words = static_cast<const uint64_t *>(blk);
            d_bad = bad_w;
        }
        if (n > 0) {
            unsigned int bad = 0;
            const hipStream_t down = c->stream_down;
            char *hb = static_cast<char *>(r->h_pinned);
            hipError_t he = hipMemcpyAsync(hb, d_words, 4 * n, hipMemcpyDeviceToHost, down);
            if (he == hipSuccess) he = hipMemcpyAsync(hb + 4 * n_round, r->d_score, 8 * n, hipMemcpyDeviceToHost, down);
            if (he == hipSuccess) he = hipMemcpyAsync(&bad, d_bad, sizeof(unsigned int), hipMemcpyDeviceToHost, down);
            const hipError_t hs = hipStreamSynchronize(down);
            if (blk) pool_free(c, blk, got);
            if (he != hipSuccess || hs != hipSuccess) { set_error("compact copy-out failed: %s", hipGetErrorString(he != hipSuccess ? he : hs)); return MS_ERR_RUNTIME; }
            if (bad) { set_error("a hit does not fit the 12-byte compact form (region index and position beside the strand bit need more than 32 bits): use ms_result_hits_packed_host"); return MS_ERR_INVALID; }
        }
        if (n == 0 && shift == 0) shift = 1;
        r->h_pinned_hits = r->n_hits;
        r->h_packed = true;
        r->h_coord_shift = shift;
    }
    char *b = static_cast<char *>(r->h_pinned);
    if (coord) *coord = reinterpret_cast<const uint32_t *>(b);
    if (score) *score = reinterpret_cast<const double *>(b + 4 * n_round);
    if (shift_out) *shift_out = r->h_coord_shift;
    return MS_OK;
}

// which compact form a batch stream left in a result: 16 (coord64 | score), 12 (coord32 | score), 0 (none: the plain arrays)
int ms_result_packed_form(const ms_result *r, int32_t *bytes_per_hit) {
    if (!r || !bytes_per_hit) { set_error("NULL argument"); return MS_ERR_INVALID; }
    *bytes_per_hit = r->h_packed && r->h_pinned_hits == r->n_hits ? (r->h_coord_shift ? 12 : 16) : (r->d_coord ? (r->coord_shift ? 12 : 16) : 0);
    return MS_OK;
}

int ms_device_pool_stats(uint64_t out[6]) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    DeviceCtx *c;
    int rc = get_ctx(g_device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->pool.mu);
    out[0] = c->pool.n_hit; out[1] = c->pool.n_miss; out[2] = c->pool.n_driver_free; out[3] = c->pool.ns_driver;
    out[4] = c->pool.bytes; out[5] = c->pool.free_.size();
    return MS_OK;
}

// the pinned-block cache (process-wide): out[0] requests served from the cache, out[1] requests that went to hipHostMalloc, out[2] blocks returned
// to the driver, out[3] nanoseconds inside the driver for [1] and [2].  Steady-state batches should show no [1] / [2].
int ms_host_pool_stats(uint64_t out[4]) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    pinned_pool_stats(out);
    return MS_OK;
}

int ms_host_alloc(size_t bytes, void **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    DeviceCtx *c;
    int rc = get_ctx(g_device, &c);                       // pinned memory needs a live HIP runtime: fails loudly without a device
    if (rc) return rc;
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) { set_error("hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e)); return MS_ERR_NOMEM; }
    *out = p;
    return MS_OK;
}

void ms_host_free(void *p) { if (p) (void) hipHostFree(p); }

int ms_result_region_counts(const ms_result *r, int64_t *out) {
    if (!r || (!out && r->P > 0)) { set_error("NULL argument"); return MS_ERR_INVALID; }
    if (r->P == 0) return MS_OK;
    if (r->h_region_counts.size() == (size_t) r->P) { std::memcpy(out, r->h_region_counts.data(), (size_t) r->P * sizeof(int64_t)); return MS_OK; }
    MS_HIP(hipSetDevice(r->device));
    MS_HIP(hipMemcpy(out, r->d_region_counts, (size_t) r->P * sizeof(int64_t), hipMemcpyDeviceToHost));
    return MS_OK;
}

}  // extern "C"

// The counts onto the host on the copy-out stream (a batch stream's third stage: the consumer's ms_result_region_counts then never issues a
// blocking device copy of its own -- 14 of those per configs[3] pass, each queued behind the copy engine's hit arrays, cost the reference
// CLI's job 5-7 ms per pass in round 6's first measurements)
int ms::result_fetch_region_counts(ms_result *r) {
    if (!r || r->P == 0) return MS_OK;
    DeviceCtx *c;
    int rc = get_ctx(r->device, &c);
    if (rc) return rc;
    if (r->h_region_counts.size() == (size_t) r->P) return MS_OK;      // (a queued scan brought them along: scan_complete)
    try { r->h_region_counts.assign((size_t) r->P, 0); } catch (const std::bad_alloc &) { set_error("out of host memory"); return MS_ERR_NOMEM; }
    const hipStream_t down = c->stream_down;
    hipError_t he = hipMemcpyAsync(r->h_region_counts.data(), r->d_region_counts, (size_t) r->P * sizeof(int64_t), hipMemcpyDeviceToHost, down);
    if (he == hipSuccess) he = hipStreamSynchronize(down);
    if (he != hipSuccess) { r->h_region_counts.clear(); set_error("copy of the region counts failed: %s", hipGetErrorString(he)); return MS_ERR_RUNTIME; }
    return MS_OK;
}

extern "C" {

int ms_result_region_counts_device(const ms_result *r, void **d_counts) {
    if (!r || !d_counts) { set_error("NULL argument"); return MS_ERR_INVALID; }
    *d_counts = r->d_region_counts;
    return MS_OK;
}

int ms_result_stats(const ms_result *r, ms_scan_stats *out) {
    if (!r || !out) { set_error("NULL argument"); return MS_ERR_INVALID; }
    *out = r->stats;
    return MS_OK;
}

// scanner.py:156-193 on the device, in place: afterwards the result holds only the kept sites (same
// order), with new per-motif offsets.  The per-motif region counts do not change (a region never
// loses its last site).
int ms_result_dedup(ms_result *r, const ms_pwmset *pwms_c) {
    if (!r || !pwms_c) { set_error("NULL handle"); return MS_ERR_INVALID; }
    ms_pwmset *pwms = const_cast<ms_pwmset *>(pwms_c);
    if (pwms->P != r->P) { set_error("result and PWM set disagree on the number of PWMs"); return MS_ERR_INVALID; }
    if (r->counts_only) { set_error("a counts-only result (MS_SCAN_COUNTS_ONLY, a counts-only batch or sweep span of a stream) holds the per-motif region counts and site numbers, no site arrays"); return MS_ERR_INVALID; }
    if (r->deduped || r->n_hits == 0) { r->deduped = true; return MS_OK; }
    if (r->coord_blk) {                                  // coordinate words of the hits before de-duplication: void
        DeviceCtx *c0;
        if (get_ctx(r->device, &c0) == MS_OK) pool_free(c0, r->coord_blk, r->coord_bytes); else (void) hipFree(r->coord_blk);
        r->coord_blk = nullptr; r->d_coord = nullptr; r->d_coord_bad = nullptr; r->coord_shift = 0;
    }
    DeviceCtx *c;
    int rc = get_ctx(r->device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk_dev(c->mu);
    std::lock_guard<std::mutex> lk_pwm(pwms->mu);
    if ((rc = pwmset_upload(pwms, c->device, c->stream))) return rc;
    const size_t n = (size_t) r->n_hits;
    // keep flags, destinations and the prefix sum's work space in ONE pooled block: hipMalloc / hipFree stall every stage of a
    // batch stream (hipFree waits for the whole device), and this runs per batch under MS_STREAM_DEDUP
    void *wblk = nullptr, *blk = nullptr;
    size_t tmp_bytes = 0, got = 0, wgot = 0;
    auto cleanup = [&]() { if (wblk) pool_free(c, wblk, wgot); wblk = nullptr; };
    const size_t n8 = (n + 31) & ~(size_t) 31;
    if ((rc = exclusive_sum_u32(nullptr, &tmp_bytes, nullptr, nullptr, n, c->stream))) return rc;
    if ((rc = pool_alloc(c, 12 * n8 + tmp_bytes + 256, &wblk, &wgot))) return rc;
    uint64_t *d_dst = static_cast<uint64_t *>(wblk);
    uint32_t *d_keep = reinterpret_cast<uint32_t *>(d_dst + n8);
    void *d_tmp = d_keep + n8;
    rc = launch_dedup((int64_t) n, r->d_motif_first, r->P, pwms->d_width, r->d_seq_idx, r->d_pos, r->d_score, r->d_strand,
                      d_keep, c->stream);
    if (!rc) rc = exclusive_sum_u32(d_tmp, &tmp_bytes, d_keep, d_dst, n, c->stream);
    uint64_t last_dst = 0;
    uint32_t last_keep = 0;
    if (!rc) {
        hipError_t he = hipMemcpyAsync(&last_dst, d_dst + (n - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(&last_keep, d_keep + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { set_error("de-dup kernels failed: %s", hipGetErrorString(he)); rc = MS_ERR_RUNTIME; }
    }
    if (rc) { cleanup(); return rc; }
    const size_t n_kept = (size_t) last_dst + last_keep;
    if ((rc = pool_alloc(c, result_block_bytes(r->P, n_kept), &blk, &got))) { cleanup(); return rc; }
    ms_result nr;                                   // carve the new block with the same layout
    nr.P = r->P;
    result_carve(&nr, blk, n_kept);
    const size_t P1 = (size_t) r->P + 1;
    hipError_t he = hipMemcpyAsync(nr.d_region_counts, r->d_region_counts, 8 * P1, hipMemcpyDeviceToDevice, c->stream);
    rc = launch_compact_hits((int64_t) n, d_keep, d_dst, r->d_seq_idx, r->d_pos, r->d_score, r->d_strand, nr.d_seq_idx,
                             nr.d_pos, nr.d_score, nr.d_strand, r->d_motif_first, r->P, nr.d_motif_first, c->stream);
    std::vector<int64_t> off(P1);
    if (!rc && he == hipSuccess) he = hipMemcpyAsync(off.data(), nr.d_motif_first, 8 * P1, hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    cleanup();
    if (rc || he != hipSuccess) {
        pool_free(c, blk, got);
        if (!rc) { set_error("compaction failed: %s", hipGetErrorString(he)); rc = MS_ERR_RUNTIME; }
        return rc;
    }
    pool_free(c, r->block, r->block_bytes);
    r->block = blk; r->block_bytes = got;
    result_carve(r, blk, n_kept);
    r->n_hits = (int64_t) n_kept;
    r->motif_offsets = off;
    r->deduped = true;
    r->h_pinned_hits = -1;                            // host copy (if any) is stale
    return MS_OK;
}

// io/__init__.py:23-33: what the site tables need per (motif, region): number of sites, max score
// (NaN where the writer prints 'NA').  Host buffers [P][R].
int ms_result_site_tables(const ms_result *r, int32_t *n_sites, double *max_score) {
    if (!r) { set_error("NULL handle"); return MS_ERR_INVALID; }
    const size_t cells = (size_t) r->P * (size_t) r->R;
    if (cells == 0) return MS_OK;
    if (r->counts_only) { set_error("a counts-only result (MS_SCAN_COUNTS_ONLY, a counts-only batch or sweep span of a stream) holds the per-motif region counts and site numbers, no site arrays"); return MS_ERR_INVALID; }
    if (!n_sites || !max_score) { set_error("NULL output"); return MS_ERR_INVALID; }
    DeviceCtx *c;
    int rc = get_ctx(r->device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk_dev(c->mu);
    int32_t *d_n = nullptr;
    double *d_m = nullptr;
    if ((rc = dev_alloc(&d_n, cells)) || (rc = dev_alloc(&d_m, cells))) { dev_free(d_n); dev_free(d_m); return rc; }
    rc = launch_site_tables(r->n_hits, r->d_motif_first, r->P, r->R, r->d_seq_idx, r->d_score, d_n, d_m, c->stream);
    hipError_t he = hipSuccess;
    if (!rc) he = hipMemcpyAsync(n_sites, d_n, cells * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
    if (!rc && he == hipSuccess) he = hipMemcpyAsync(max_score, d_m, cells * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    dev_free(d_n); dev_free(d_m);
    if (rc) return rc;
    if (he != hipSuccess) { set_error("site table kernels failed: %s", hipGetErrorString(he)); return MS_ERR_RUNTIME; }
    return MS_OK;
}

void ms_result_free(ms_result *r) {
    if (!r) return;
    (void) hipSetDevice(r->device);
    if (r->h_pinned) pinned_free(r->h_pinned, r->h_pinned_bytes);
    DeviceCtx *c = nullptr;
    const bool have = (r->block || r->coord_blk) && get_ctx(r->device, &c) == MS_OK;
    if (r->block) { if (have) pool_free(c, r->block, r->block_bytes); else (void) hipFree(r->block); }
    if (r->coord_blk) { if (have) pool_free(c, r->coord_blk, r->coord_bytes); else (void) hipFree(r->coord_blk); }
    delete r;
}

// --------------------------------------------------------------------- window sweep --
//
// configs[4]-style sweep (N3): the windows [begin + k*stride, begin + k*stride + window), k = 0..n_windows-1, of one
// chromosome, with the result the reference gives when every window is a region of its own (scanner.py:71-87 cuts them,
// cscore.c:336-390 scans each) -- but every base is scored ONCE: the span is scanned as one region and each hit is
// handed to all windows that contain it whole (window / stride of them), written straight to its place in the
// reference's order (motif, window, position, strand) -- sweep_scatter_kernel, no second sort.

}  // extern "C"

namespace ms {

// r1: the scan of the span as ONE region.  Consumes r1 (also on failure).
int sweep_handout_locked(DeviceCtx *c, ms_pwmset *pwms, ms_result *r1, int64_t span_bases, int32_t window, int32_t stride,
                         int64_t n_windows, ms_result **out, bool counts_only) {
    int rc;
    auto fail = [&](int code) { ms_result_free(r1); return code; };
    // the device copies of the widths may have moved since the scan if the set was used on another device in between
    if ((rc = pwmset_upload(pwms, c->device, c->stream))) return fail(rc);
    const size_t n1 = (size_t) r1->n_hits;

    std::unique_ptr<ms_result> res(new (std::nothrow) ms_result());
    if (!res) { set_error("out of host memory"); return fail(MS_ERR_NOMEM); }
    res->device = r1->device;
    res->P = pwms->P;
    res->R = n_windows;
    res->stats = r1->stats;
    res->invalid = r1->invalid;
    res->motif_offsets.assign((size_t) pwms->P + 1, 0);
    ms_result *raw = res.release();
    auto fail2 = [&](int code) { ms_result_free(raw); return fail(code); };

    // counts, destinations and the prefix sum's work space in ONE pooled block (no hipMalloc / hipFree per span: ADVICE r2)
    uint32_t *d_cnt = nullptr;
    uint64_t *d_dst = nullptr;
    void *wblk = nullptr;
    size_t wgot = 0;
    auto cleanup = [&]() { if (wblk) pool_free(c, wblk, wgot); wblk = nullptr; };
    uint64_t total = 0;
    hipError_t he = hipSuccess;
    (void) hipEventRecord(c->ev[0], c->stream);
    if (counts_only) {
        // what a counts-only sweep reads of a span: per motif the windows with >= 1 site, and the number of sites -- one pass over the
        // span's hit positions, nothing handed out (sweep_countonly_kernel)
        void *blk = nullptr;
        size_t got = 0;
        if ((size_t) pwms->P > 65536) { set_error("internal: counts-only hand-out with more than 65536 motifs"); return fail2(MS_ERR_RUNTIME); }
        if ((rc = pool_alloc(c, result_block_bytes(pwms->P, 1), &blk, &got))) return fail2(rc);        // (room for >= 65536 words behind the counts)
        raw->block = blk;
        raw->block_bytes = got;
        result_carve(raw, blk, 1);
        raw->counts_only = true;
        const size_t P1 = (size_t) pwms->P + 1;
        unsigned long long *d_sites = reinterpret_cast<unsigned long long *>(raw->d_seq_idx);           // per-motif sites, summed on the host
        he = hipMemsetAsync(raw->d_region_counts, 0, 8 * P1, c->stream);
        if (he == hipSuccess) he = hipMemsetAsync(raw->d_motif_first, 0, 8 * P1, c->stream);
        if (he == hipSuccess) he = hipMemsetAsync(d_sites, 0, 8 * P1, c->stream);
        if (he != hipSuccess) { set_error("memset failed: %s", hipGetErrorString(he)); return fail2(MS_ERR_RUNTIME); }
        rc = launch_sweep_countonly((int64_t) n1, r1->d_motif_first, r1->P, pwms->d_width, r1->d_pos, window, stride, n_windows,
                                    raw->d_region_counts, d_sites, c->stream);
        if (rc) return fail2(rc);
        (void) hipEventRecord(c->ev[1], c->stream);
        std::vector<unsigned long long> per_motif((size_t) pwms->P, 0ULL);
        if (pwms->P > 0) he = hipMemcpyAsync(per_motif.data(), d_sites, 8 * (size_t) pwms->P, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) { set_error("sweep count failed: %s", hipGetErrorString(he)); return fail2(MS_ERR_RUNTIME); }
        unsigned long long n_sites = 0;
        for (size_t p = 0; p < per_motif.size(); p++) {      // the offsets a caller slices by: consistent with n_hits although no site array exists
            n_sites += per_motif[p];
            raw->motif_offsets[p + 1] = (int64_t) n_sites;
        }
        total = n_sites;
        raw->n_hits = (int64_t) total;
        float ms01 = 0;
        (void) hipEventElapsedTime(&ms01, c->ev[0], c->ev[1]);
        ms_scan_stats &stt = raw->stats;
        stt.ms_finalize += ms01;
        stt.ms_total += ms01;
        stt.n_hits = (int64_t) total;
        stt.n_bases = span_bases;
        stt.n_windows = 0;
        for (int32_t p = 0; p < pwms->P; p++) stt.n_windows += n_windows * std::max<int64_t>(window - pwms->widths[p] + 1, 0);
        ms_result_free(r1);
        *out = raw;
        return MS_OK;
    }
    if (n1 > 0) {
        size_t tmp_bytes = 0;
        const size_t n8 = (n1 + 31) & ~(size_t) 31;
        if ((rc = exclusive_sum_u32(nullptr, &tmp_bytes, nullptr, nullptr, n1, c->stream))) return fail2(rc);
        if ((rc = pool_alloc(c, 12 * n8 + tmp_bytes + 256, &wblk, &wgot))) return fail2(rc);
        d_dst = static_cast<uint64_t *>(wblk);
        d_cnt = reinterpret_cast<uint32_t *>(d_dst + n8);
        void *d_tmp = d_cnt + n8;
        rc = launch_sweep_count((int64_t) n1, r1->d_motif_first, r1->P, pwms->d_width, r1->d_pos, window, stride, n_windows,
                                d_cnt, c->stream);
        if (!rc) rc = exclusive_sum_u32(d_tmp, &tmp_bytes, d_cnt, d_dst, n1, c->stream);
        uint32_t last_cnt = 0;
        uint64_t last_dst = 0;
        if (!rc) {
            he = hipMemcpyAsync(&last_cnt, d_cnt + (n1 - 1), 4, hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(&last_dst, d_dst + (n1 - 1), 8, hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            if (he != hipSuccess) { set_error("sweep count failed: %s", hipGetErrorString(he)); rc = MS_ERR_RUNTIME; }
        }
        if (rc) { cleanup(); return fail2(rc); }
        total = last_dst + last_cnt;                          // 64-bit prefix sums: > 2^32 sites per call are fine
    }
    raw->n_hits = (int64_t) total;
    {
        void *blk = nullptr;
        size_t got = 0;
        if ((rc = pool_alloc(c, result_block_bytes(pwms->P, (size_t) total), &blk, &got))) { cleanup(); return fail2(rc); }
        raw->block = blk;
        raw->block_bytes = got;
        result_carve(raw, blk, (size_t) total);
        const size_t P1 = (size_t) pwms->P + 1;
        he = hipMemsetAsync(raw->d_region_counts, 0, 8 * P1, c->stream);
        if (he == hipSuccess) he = hipMemsetAsync(raw->d_motif_first, 0xFF, 8 * P1, c->stream);
        if (he != hipSuccess) { cleanup(); set_error("memset failed: %s", hipGetErrorString(he)); return fail2(MS_ERR_RUNTIME); }
    }
    (void) hipEventRecord(c->ev[1], c->stream);
    rc = launch_sweep_scatter((int64_t) n1, r1->d_motif_first, r1->P, pwms->d_width, r1->d_pos, r1->d_score, r1->d_strand, d_dst,
                              window, stride, n_windows, (int64_t) total, raw->d_seq_idx, raw->d_pos, raw->d_score, raw->d_strand,
                              raw->d_motif_first, raw->d_region_counts, c->stream);
    if (rc) { cleanup(); return fail2(rc); }
    (void) hipEventRecord(c->ev[2], c->stream);
    he = hipMemcpyAsync(raw->motif_offsets.data(), raw->d_motif_first, raw->motif_offsets.size() * sizeof(int64_t),
                        hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    cleanup();
    if (he != hipSuccess) { set_error("sweep hand-out failed: %s", hipGetErrorString(he)); return fail2(MS_ERR_RUNTIME); }

    // statistics: the span scan's stage times plus the hand-out (counts + prefix sum + scatter, booked under ms_finalize)
    float ms01 = 0, ms12 = 0;
    (void) hipEventElapsedTime(&ms01, c->ev[0], c->ev[1]);
    (void) hipEventElapsedTime(&ms12, c->ev[1], c->ev[2]);
    ms_scan_stats &stt = raw->stats;
    stt.ms_finalize += ms01 + ms12;
    stt.ms_total += ms01 + ms12;
    stt.n_hits = (int64_t) total;
    stt.n_bases = span_bases;                                   // bases scanned (each once)
    stt.n_windows = 0;                                          // unit count of the sweep as the reference sees it
    for (int32_t p = 0; p < pwms->P; p++) stt.n_windows += n_windows * std::max<int64_t>(window - pwms->widths[p] + 1, 0);
    stt.hbm_bytes_algorithmic += 16 * ((int64_t) total - (int64_t) n1);
    ms_result_free(r1);
    *out = raw;
    return MS_OK;
}

}  // namespace ms

extern "C" {

int ms_scan_sweep(const ms_pwmset *pwms_c, const ms_genome *g, int32_t chrom, int64_t begin, int64_t end, int32_t window,
                  int32_t stride, int strand_mask, uint32_t flags, ms_result **out) {
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    *out = nullptr;
    if (!pwms_c || !g) { set_error("NULL handle"); return MS_ERR_INVALID; }
    if (strand_mask < 1 || strand_mask > 3) { set_error("invalid strand mask %d (1 '+', 2 '-', 3 both)", strand_mask); return MS_ERR_INVALID; }
    if (flags & ~(uint32_t) MS_SCAN_EXACT_ONLY) { set_error("unknown scan flags 0x%x", flags); return MS_ERR_INVALID; }
    if (window < 1 || stride < 1) { set_error("window and stride must be positive"); return MS_ERR_INVALID; }
    if (begin < 0 || end < begin) { set_error("bad span [%lld, %lld)", (long long) begin, (long long) end); return MS_ERR_INVALID; }
    const int64_t n_windows = end - begin >= window ? (end - begin - window) / stride + 1 : 0;
    const int64_t span_end = n_windows > 0 ? begin + (n_windows - 1) * stride + window : begin;
    ms_pwmset *pwms = const_cast<ms_pwmset *>(pwms_c);
    ms_seqset *span = nullptr;
    int rc = ms_seqset_from_genome(g, &chrom, &begin, &span_end, 1, &span);       // validates chrom / coordinates
    if (rc) return rc;
    DeviceCtx *c;
    if ((rc = get_ctx(span->device, &c))) { ms_seqset_free(span); return rc; }
    // both locks are held across the span scan AND the hand-out: nothing can move the PWM set's device copies in between
    std::lock_guard<std::mutex> lk_dev(c->mu);
    std::lock_guard<std::mutex> lk_pwm(pwms->mu);
    ms_result *r1 = nullptr;
    rc = scan_locked(c, pwms, span, strand_mask, flags, &r1);
    const int64_t span_bases = span->n_bases;
    ms_seqset_free(span);
    if (rc) return rc;
    return sweep_handout_locked(c, pwms, r1, span_bases, window, stride, n_windows, out);
}

// --------------------------------------------------------------------------- score --

int ms_score(const ms_pwmset *pwms_c, const ms_seqset *seqs, int strand_mask, double *out) {
    if (!pwms_c || !seqs) { set_error("NULL handle"); return MS_ERR_INVALID; }
    if (strand_mask < 1 || strand_mask > 3) { set_error("invalid strand mask %d", strand_mask); return MS_ERR_INVALID; }
    ms_pwmset *pwms = const_cast<ms_pwmset *>(pwms_c);
    if (pwms->P == 0 || seqs->R == 0) return MS_OK;
    if (!out) { set_error("out is NULL"); return MS_ERR_INVALID; }
    DeviceCtx *c;
    int rc = get_ctx(seqs->device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk_dev(c->mu);
    std::lock_guard<std::mutex> lk_pwm(pwms->mu);
    if ((rc = pwmset_upload(pwms, c->device, c->stream))) return rc;
    double *d_out = nullptr;
    const size_t n = (size_t) pwms->P * (size_t) seqs->R;
    if ((rc = dev_alloc(&d_out, n))) return rc;
    rc = launch_score(dev_seq(seqs), dev_pwm(pwms), strand_mask, d_out, c->stream);
    hipError_t he = hipSuccess;
    if (!rc) he = hipMemcpyAsync(out, d_out, n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (!rc && he == hipSuccess) he = hipStreamSynchronize(c->stream);
    dev_free(d_out);
    if (rc) return rc;
    if (he != hipSuccess) { set_error("score kernel failed: %s", hipGetErrorString(he)); return MS_ERR_RUNTIME; }
    return MS_OK;
}

// The cutoff builder's device half (cli/motif.py:134-137, motif/__init__.py:378-401): score R
// sampled sequences with every PWM (c_score), sort each PWM's scores in descending order and read
// the scores at the requested 0-based ranks (the reference takes rank int(n * 0.1**e) - 1).
int ms_score_ranks(const ms_pwmset *pwms_c, const ms_seqset *seqs, int strand_mask, const int64_t *ranks,
                   int32_t n_ranks, double *out) {
    if (!pwms_c || !seqs) { set_error("NULL handle"); return MS_ERR_INVALID; }
    if (strand_mask < 1 || strand_mask > 3) { set_error("invalid strand mask %d", strand_mask); return MS_ERR_INVALID; }
    if (n_ranks < 0 || (n_ranks > 0 && (!ranks || !out))) { set_error("bad ranks / out"); return MS_ERR_INVALID; }
    ms_pwmset *pwms = const_cast<ms_pwmset *>(pwms_c);
    if (pwms->P == 0 || n_ranks == 0) return MS_OK;
    if (seqs->R == 0) { set_error("no sequences to rank"); return MS_ERR_INVALID; }
    DeviceCtx *c;
    int rc = get_ctx(seqs->device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk_dev(c->mu);
    std::lock_guard<std::mutex> lk_pwm(pwms->mu);
    if ((rc = pwmset_upload(pwms, c->device, c->stream))) return rc;
    const size_t R = (size_t) seqs->R;
    const int32_t batch = (int32_t) std::max<size_t>(1, std::min<size_t>((size_t) pwms->P, ((size_t) 1 << 27) / R));
    double *d_scores = nullptr, *d_sorted = nullptr, *d_out = nullptr;
    int64_t *d_ranks = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    auto cleanup = [&]() { dev_free(d_scores); dev_free(d_sorted); dev_free(d_out); dev_free(d_ranks); if (d_tmp) (void) hipFree(d_tmp); };
    if ((rc = dev_alloc(&d_scores, (size_t) batch * R)) || (rc = dev_alloc(&d_sorted, R)) ||
        (rc = dev_alloc(&d_out, (size_t) pwms->P * (size_t) n_ranks)) || (rc = dev_alloc(&d_ranks, (size_t) n_ranks))) { cleanup(); return rc; }
    hipError_t he = hipMemcpyAsync(d_ranks, ranks, (size_t) n_ranks * sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess && (rc = sort_doubles_desc(nullptr, &tmp_bytes, d_scores, d_sorted, R, c->stream)) == MS_OK) {
        he = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1);
        if (he != hipSuccess) { set_error("hipMalloc (sort) failed: %s", hipGetErrorString(he)); rc = MS_ERR_NOMEM; }
    }
    const DevSeq S = dev_seq(seqs);
    for (int32_t p0 = 0; rc == MS_OK && he == hipSuccess && p0 < pwms->P; p0 += batch) {
        const int32_t n = std::min(batch, pwms->P - p0);
        DevPwm sub = dev_pwm(pwms);
        sub.tab_off += p0; sub.width += p0; sub.max_raw += p0; sub.cutoff += p0; sub.raw_floor += p0; sub.P = n;
        rc = launch_score(S, sub, strand_mask, d_scores, c->stream);
        for (int32_t i = 0; rc == MS_OK && i < n; i++) {
            size_t tb = tmp_bytes;
            rc = sort_doubles_desc(d_tmp, &tb, d_scores + (size_t) i * R, d_sorted, R, c->stream);
            if (rc == MS_OK) rc = launch_gather_ranks(d_sorted, (int64_t) R, d_ranks, n_ranks, d_out + (size_t) (p0 + i) * n_ranks, c->stream);
        }
    }
    if (rc == MS_OK && he == hipSuccess)
        he = hipMemcpyAsync(out, d_out, (size_t) pwms->P * (size_t) n_ranks * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (rc == MS_OK && he == hipSuccess) he = hipStreamSynchronize(c->stream);
    else (void) hipStreamSynchronize(c->stream);
    cleanup();
    if (rc) return rc;
    if (he != hipSuccess) { set_error("score/rank kernels failed: %s", hipGetErrorString(he)); return MS_ERR_RUNTIME; }
    return MS_OK;
}

// --------------------------------------------------------------------------- dedup --

// scanner.py:156-193 on the flat hit arrays.  Within one (motif, region) the reference splits
// the sites by strand, runs the greedy pass on each strand, and re-merges by a stable sort on
// start with '+' first -- which is the order the hits already have, so de-dup is a filter.
int ms_dedup_hits(const int64_t *motif_offsets, int32_t n_pwms, const int32_t *widths, const int64_t *seq_idx,
                  const int64_t *pos, const double *score, const int8_t *strand, uint8_t *keep) {
    if (n_pwms < 0 || !motif_offsets) { set_error("bad arguments"); return MS_ERR_INVALID; }
    const int64_t n = motif_offsets[n_pwms];
    if (n > 0 && (!widths || !seq_idx || !pos || !score || !strand || !keep)) { set_error("NULL array"); return MS_ERR_INVALID; }
    for (int64_t i = 0; i < n; i++) keep[i] = 1;
    for (int32_t p = 0; p < n_pwms; p++) {
        const int64_t W = widths[p];
        int64_t a = motif_offsets[p];
        const int64_t end = motif_offsets[p + 1];
        while (a < end) {
            int64_t b = a;
            while (b < end && seq_idx[b] == seq_idx[a]) b++;
            for (int8_t sd = 1; sd <= 2; sd++) {
                int64_t cur = -1;                        // index of the site currently kept (scanner.py:160)
                for (int64_t i = a; i < b; i++) {
                    if (strand[i] != sd) continue;
                    if (cur < 0) { cur = i; continue; }
                    if (pos[i] - pos[cur] < W) {
                        if (score[cur] >= score[i]) keep[i] = 0;           // tie keeps the earlier site
                        else { keep[cur] = 0; cur = i; }
                    } else {
                        cur = i;
                    }
                }
            }
            a = b;
        }
    }
    return MS_OK;
}

// ------------------------------------------------------------------ test inspection --
// Host-only views of the pre-filter plan, so CPU tests can prove the quantiser never drops a
// window the reference reports (tests/test_prefilter_plan.py).  Not part of the drop-in surface.

int ms_debug_plan_dims(const ms_pwmset *pwms_c, int strand_mask, int64_t lds_budget, int32_t *n_fast,
                       int32_t *n_exact, int32_t *n_groups, int32_t *n_tiles) {
    if (!pwms_c) { set_error("NULL handle"); return MS_ERR_INVALID; }
    if (strand_mask < 1 || strand_mask > 3) { set_error("invalid strand mask %d", strand_mask); return MS_ERR_INVALID; }
    ms_pwmset *pwms = const_cast<ms_pwmset *>(pwms_c);
    std::lock_guard<std::mutex> lk(pwms->mu);
    int rc = pwmset_plan(pwms, strand_mask, (size_t) lds_budget, false, false, -1);
    if (rc) return rc;
    if (n_fast) *n_fast = (int32_t) pwms->plan.fast_motifs.size();
    if (n_exact) *n_exact = (int32_t) pwms->plan.exact_motifs.size();
    if (n_groups) *n_groups = (int32_t) pwms->plan.group_kb.size();
    if (n_tiles) *n_tiles = (int32_t) pwms->plan.tiles.size();
    return MS_OK;
}

// The plan built by the last ms_debug_plan_dims call, decoded from the PHYSICAL operand image the kernel reads (any pointer
// may be NULL): group_fields [n_groups][16] motif of the field (-1 = empty), rows [n_groups][16 fields][64 columns][4 bases]
// int16 = what the product adds for that base at that column, units of 1/8 (the bias column reads 0 here), bias [n_groups][16]
// = the entry of the field's last column (MS_ERR_RUNTIME if its four bases disagree), group_kb [n_groups] matrix instructions
// per row tile, group_cols [n_groups] columns of the group's fields incl. the bias column (16 per instruction, paired rows: 8),
// group_paired [n_groups] 0 = plain row, 1 / 2 = field X / Y of a paired row, exact_motifs [n_exact], tile_first_group [n_tiles + 1].
int ms_debug_plan_rows(const ms_pwmset *pwms_c, int32_t *group_fields, int16_t *rows, int32_t *bias, int32_t *group_kb,
                       int32_t *group_cols, int32_t *group_paired, int32_t *exact_motifs, int32_t *tile_first_group) {
    if (!pwms_c) { set_error("NULL handle"); return MS_ERR_INVALID; }
    ms_pwmset *pwms = const_cast<ms_pwmset *>(pwms_c);
    std::lock_guard<std::mutex> lk(pwms->mu);
    const PrefilterPlan &pl = pwms->plan;
    if (pwms->plan_strand < 0) { set_error("call ms_debug_plan_dims first"); return MS_ERR_INVALID; }
    const size_t nq = pl.group_kb.size();
    if (group_fields && nq) std::memcpy(group_fields, pl.group_fields.data(), pl.group_fields.size() * sizeof(int32_t));
    if (exact_motifs && !pl.exact_motifs.empty())
        std::memcpy(exact_motifs, pl.exact_motifs.data(), pl.exact_motifs.size() * sizeof(int32_t));
    if (tile_first_group) {
        for (size_t t = 0; t < pl.tiles.size(); t++) tile_first_group[t] = pl.tiles[t].first_group;
        tile_first_group[pl.tiles.size()] = (int32_t) nq;
    }
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(pl.tables.data());
    for (size_t q = 0; q < nq; q++) {
        const GroupInfo &gi = pl.group_info[q];
        const int n_cols = pl.group_cols[q];
        const uint8_t *tab = bytes + gi.tab_off;
        // column c of the field: plain rows -- column c % 16 of k-block c / 16; paired rows -- column c % 8 of half-block c / 8 in k-half `sel`
        auto entry = [&](int row, int c, int b) {
            return gi.paired ? f6_value(f6_get(tab, gi.nk, c / kPairCols, row, kPairCols * gi.sel + c % kPairCols, b))
                             : f6_value(f6_get(tab, gi.nk, c / kF6Cols, row, c % kF6Cols, b));
        };
        if (group_kb) group_kb[q] = gi.nk;
        if (group_cols) group_cols[q] = n_cols;
        if (group_paired) group_paired[q] = gi.paired ? 1 + gi.sel : 0;
        for (int f = 0; f < kGroupFields; f++) {
            const int row = mfma_row_of(gi.h, f);
            int b0 = entry(row, n_cols - 1, 0);
            if (gi.paired) {                                    // what the kernel's constant B slots make of the four entries, less the field offset
                b0 = -kPairOffset;
                for (int b = 0; b < 4; b++) b0 += kPairBiasW[b] * entry(row, n_cols - 1, b);
            } else {
                for (int b = 1; b < 4; b++)
                    if (entry(row, n_cols - 1, b) != b0) {
                        set_error("bias column of group %zu field %d differs between bases", q, f);
                        return MS_ERR_RUNTIME;
                    }
            }
            if (bias) bias[q * kGroupFields + f] = b0;
            if (rows)
                for (int c = 0; c < kF6Cols * kF6MaxKb; c++)
                    for (int b = 0; b < 4; b++)
                        rows[((q * kGroupFields + f) * (kF6Cols * kF6MaxKb) + c) * 4 + b] = (int16_t) (c < n_cols - 1 ? entry(row, c, b) : 0);       // units of 1/8
        }
    }
    return MS_OK;
}

// Free the calling thread's device work buffers (they are grow-only otherwise); lets a test
// exercise the "buffer too small -> grow -> second pass" path deterministically.
int ms_debug_release_scratch(void) {
    DeviceCtx *c;
    int rc = get_ctx(g_device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    Scratch &sc = c->sc;
    dev_free(sc.cand); dev_free(sc.keys); dev_free(sc.vals); dev_free(sc.keys_sorted); dev_free(sc.chunk_counters);
    sc.chunk_counters_cap = 0;
    if (sc.sort_tmp) (void) hipFree(sc.sort_tmp);
    sc.sort_tmp = nullptr;
    sc.cand_cap = sc.hit_cap = sc.sort_tmp_bytes = 0;
    return MS_OK;
}

}  // extern "C"
