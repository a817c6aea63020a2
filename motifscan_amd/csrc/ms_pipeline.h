// ms_pipeline.h -- the host-thread machinery of a batch stream (ms_stream.hip), free of any device code so that the SAME
// source runs under ThreadSanitizer with stub stage functions (tests/sanitize/stream_tsan.cpp, `make -C motifscan_amd/csrc
// sanitize`; never on the GPU box).  The reference has nothing like it: its scanner is one call under the GIL with its
// state in file-scope globals (cscore.c:26-34).
//
//     submit -> [q_in] -> uploader -> [q_up] -> scanner -> [q_scan] -> downloader -> [q_done] -> next
//
// Three threads, one per stage, bounded queues between them, results in submission order.  The scan stage keeps ONE job
// queued behind the one it waits for (Ops::scan_start may answer "pending": finish it with Ops::scan_finish after the
// NEXT job has been started), so the device goes from one batch's last kernel straight into the next batch's first.
//
// Ops (the stream) provides:   void bind_thread();                        per-thread set-up (device binding)
//                              void upload(J *), void download(J *);      stages 0 and 2
//                              bool scanner_begin(); void scanner_end();  true: pending scans are available (two slots)
//                              bool scan_start(J *, int slot);            slot < 0: run to the end; true = pending in `slot`
//                              void scan_finish(J *, int slot);
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

namespace ms {

template <class J>
class JobQueue {
public:
    explicit JobQueue(size_t cap) : cap_(cap) {}
    void push(J *j) {
        std::unique_lock<std::mutex> lk(mu_);
        not_full_.wait(lk, [&] { return q_.size() < cap_; });
        q_.push_back(j);
        not_empty_.notify_one();
    }
    // true: *out is a job, or nullptr when the queue is closed and drained; false: nothing there right now
    bool try_pop(J **out) {
        std::lock_guard<std::mutex> lk(mu_);
        if (q_.empty()) {
            if (!closed_) return false;
            *out = nullptr;
            return true;
        }
        *out = q_.front();
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    J *pop() {                        // nullptr = closed and drained
        std::unique_lock<std::mutex> lk(mu_);
        not_empty_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return nullptr;
        J *j = q_.front();
        q_.pop_front();
        not_full_.notify_one();
        return j;
    }
    void close() {
        std::lock_guard<std::mutex> lk(mu_);
        closed_ = true;
        not_empty_.notify_all();
    }

private:
    std::mutex mu_;
    std::condition_variable not_full_, not_empty_;
    std::deque<J *> q_;
    size_t cap_;
    bool closed_ = false;
};

inline double pipeline_now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Where a stage's thread spends its time: working, waiting for input, waiting for room downstream
struct StageClock {
    std::atomic<uint64_t> work_us{0}, wait_in_us{0}, wait_out_us{0}, jobs{0};
    static void add(std::atomic<uint64_t> &a, double s) { a.fetch_add((uint64_t) (s * 1e6)); }
};

template <class J, class Ops>
class StagePipeline {
public:
    StagePipeline(Ops *ops, int depth)
        : ops_(ops), capacity_(4 * depth + 3),               // three bounded queues + one job inside each stage + done results
          q_in_((size_t) depth), q_up_((size_t) depth), q_scan_((size_t) depth),
          q_done_((size_t) capacity_ + 1) {}                 // q_done never blocks: in_flight <= capacity

    int capacity() const { return capacity_; }
    int in_flight() const { return in_flight_.load(); }
    const StageClock &clock(int k) const { return clk_[k]; }

    void start() {                                           // may throw std::system_error; then call shutdown()
        th_up_ = std::thread([this] { uploader(); });
        n_started_++;
        th_scan_ = std::thread([this] { scanner(); });
        n_started_++;
        th_down_ = std::thread([this] { downloader(); });
        n_started_++;
    }

    // false: `capacity` jobs are in flight (collect results first); the job is then still the caller's
    bool submit(J *j) {
        if (in_flight_.load() >= capacity_) return false;
        in_flight_.fetch_add(1);
        q_in_.push(j);                                       // may wait for the uploader; never for the consumer
        return true;
    }

    // The oldest job, done (the caller owns it again); nullptr when nothing is in flight or the pipeline is shut down
    J *next() {
        if (in_flight_.load() == 0) return nullptr;
        J *j = q_done_.pop();
        if (j) in_flight_.fetch_sub(1);
        return j;
    }

    // Close the input, let the stages finish what they hold, join; `drop` gets every finished job nobody collected
    template <class F>
    void shutdown(F &&drop) {
        q_in_.close();
        if (th_up_.joinable()) th_up_.join();
        if (n_started_ < 2) q_up_.close();                   // a stage that never started closes nothing downstream
        if (th_scan_.joinable()) th_scan_.join();
        if (n_started_ < 3) q_scan_.close();
        if (th_down_.joinable()) th_down_.join();
        if (n_started_ < 3) q_done_.close();
        while (J *j = q_done_.pop()) drop(j);                // closed by the downloader: drains, then nullptr
    }

private:
    // one stage: pop -> work -> push, each leg timed
    template <class F>
    void run_stage(int k, JobQueue<J> &in, JobQueue<J> &out, F &&work) {
        ops_->bind_thread();
        for (;;) {
            const double t0 = pipeline_now_s();
            J *j = in.pop();
            const double t1 = pipeline_now_s();
            if (!j) break;
            work(j);
            const double t2 = pipeline_now_s();
            out.push(j);
            const double t3 = pipeline_now_s();
            StageClock::add(clk_[k].wait_in_us, t1 - t0);
            StageClock::add(clk_[k].work_us, t2 - t1);
            StageClock::add(clk_[k].wait_out_us, t3 - t2);
            clk_[k].jobs.fetch_add(1);
        }
        out.close();
    }

    void uploader() {
        run_stage(0, q_in_, q_up_, [this](J *j) { ops_->upload(j); });
    }

    void downloader() {
        run_stage(2, q_scan_, q_done_, [this](J *j) { ops_->download(j); });
    }

    void scanner() {
        ops_->bind_thread();
        const bool pend_ok = ops_->scanner_begin();
        JobQueue<J> &in = q_up_, &out = q_scan_;
        J *waiting = nullptr;                                // its scan is queued on the device, not yet waited for
        int wslot = 0;
        bool drained = false;
        auto span = [&](std::atomic<uint64_t> &acc, auto &&fn) { const double t0 = pipeline_now_s(); fn(); StageClock::add(acc, pipeline_now_s() - t0); };
        while (!drained || waiting) {
            J *j = nullptr;
            if (!drained) {
                bool have = true;
                span(clk_[1].wait_in_us, [&] {
                    if (waiting) have = in.try_pop(&j);      // a scan is in flight: take the next batch only if it is already there
                    else j = in.pop();
                });
                if (have && !j) drained = true;
            }
            bool j_pending = false;
            if (j) span(clk_[1].work_us, [&] { j_pending = ops_->scan_start(j, pend_ok ? (wslot ^ 1) : -1); });   // queued BEHIND the waiting scan (or run to the end)
            if (waiting) {
                span(clk_[1].work_us, [&] { ops_->scan_finish(waiting, wslot); });
                span(clk_[1].wait_out_us, [&] { out.push(waiting); });
                clk_[1].jobs.fetch_add(1);
                waiting = nullptr;
            }
            if (j) {
                if (j_pending) { waiting = j; wslot ^= 1; }
                else {
                    span(clk_[1].wait_out_us, [&] { out.push(j); });
                    clk_[1].jobs.fetch_add(1);
                }
            }
        }
        out.close();
        ops_->scanner_end();
    }

    Ops *ops_;
    int capacity_;
    int n_started_ = 0;                                      // written by start()'s thread only, before any shutdown()
    std::atomic<int> in_flight_{0};
    JobQueue<J> q_in_, q_up_, q_scan_, q_done_;
    std::thread th_up_, th_scan_, th_down_;
    StageClock clk_[3];
};

}  // namespace ms
