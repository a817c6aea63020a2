// tests/sanitize/stream_tsan.cpp -- the batch stream's host-thread machinery (motifscan_amd/csrc/ms_pipeline.h: the SAME
// header libmotifscan_amd.so is built from) under ThreadSanitizer, with stub stage functions in place of the device work.
// CPU build container only (`make -C motifscan_amd/csrc sanitize`; tests/test_sanitizers.py runs it).
//
// What the stubs model: uploads / scans / copy-outs of random duration, scans that are "pending" (queued behind the previous
// one, finished after the NEXT has been started -- the two-slot protocol of ms_stream::scan_start / scan_finish) or run to the
// end, failing jobs, a consumer on a second thread, statistics read while the stages run, and a stream freed with jobs in flight.
// Checked besides the races TSan itself reports: results leave in submission order, every job passes every stage exactly once,
// a pending job's slot is never reused before it is finished, nothing is lost at shutdown.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../motifscan_amd/csrc/ms_pipeline.h"

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
    } while (0)

namespace {

struct Job {
    int id = 0;
    int stages = 0;            // bit 0 upload, 1 scan started, 2 scan finished (or run to the end), 3 download
    bool want_pending = false;
    int rc = 0;
    std::vector<int> payload;  // written by one stage, read by the next: a missing happens-before edge would be a reported race
};

struct StubOps {
    int max_us;
    bool pend_ok;
    std::atomic<int> bound{0}, begun{0}, ended{0};
    int slot_owner[2] = {-1, -1};       // scanner thread only
    explicit StubOps(int us, bool pend) : max_us(us), pend_ok(pend) {}

    static void nap(int max_us, unsigned seed) {
        if (max_us <= 0) return;
        std::this_thread::sleep_for(std::chrono::microseconds(seed % (unsigned) max_us));
    }
    void bind_thread() { bound.fetch_add(1); }
    void upload(Job *j) {
        CHECK(j->stages == 0);
        nap(max_us, 7u * (unsigned) j->id + 1);
        j->payload.assign(16, j->id);
        j->stages |= 1;
    }
    bool scanner_begin() { begun.fetch_add(1); return pend_ok; }
    void scanner_end() { ended.fetch_add(1); CHECK(slot_owner[0] == -1 && slot_owner[1] == -1); }
    bool scan_start(Job *j, int slot) {
        CHECK(j->stages == 1 && (int) j->payload.size() == 16 && j->payload[3] == j->id);
        CHECK(pend_ok ? (slot == 0 || slot == 1) : slot < 0);
        nap(max_us, 13u * (unsigned) j->id + 5);
        j->stages |= 2;
        if (j->id % 11 == 10) { j->rc = 3; j->stages |= 4; return false; }      // a failing batch runs to the end at once
        if (slot >= 0 && j->want_pending) {
            CHECK(slot_owner[slot] == -1);                                       // never a slot that still holds a pending scan
            slot_owner[slot] = j->id;
            return true;
        }
        j->payload.push_back(-j->id);
        j->stages |= 4;
        return false;
    }
    void scan_finish(Job *j, int slot) {
        CHECK(j->stages == 3 && slot_owner[slot] == j->id);
        slot_owner[slot] = -1;
        nap(max_us, 3u * (unsigned) j->id + 2);
        j->payload.push_back(-j->id);
        j->stages |= 4;
    }
    void download(Job *j) {
        CHECK(j->stages == 7);
        CHECK(j->rc != 0 || ((int) j->payload.size() == 17 && j->payload.back() == -j->id));
        nap(max_us, 5u * (unsigned) j->id + 3);
        j->stages |= 8;
    }
};

using Pipe = ms::StagePipeline<Job, StubOps>;

// producer and consumer on ONE thread, the way _lib.scan_stream drives a stream
void run_single_thread(int n_jobs, int depth, int max_us, bool pend_ok, unsigned seed) {
    StubOps ops(max_us, pend_ok);
    Pipe pipe(&ops, depth);
    pipe.start();
    std::mt19937 rng(seed);
    int next_id = 0, next_out = 0;
    auto collect = [&] {
        Job *j = pipe.next();
        CHECK(j && j->id == next_out && j->stages == 15);
        CHECK((j->rc != 0) == (j->id % 11 == 10));
        next_out++;
        delete j;
    };
    while (next_id < n_jobs) {
        while (pipe.in_flight() >= pipe.capacity()) collect();
        Job *j = new Job();
        j->id = next_id++;
        j->want_pending = rng() % 4 != 0;
        CHECK(pipe.submit(j));
        if (rng() % 3 == 0 && pipe.in_flight() > 0) collect();
    }
    // over capacity: refused, the job stays ours
    while (pipe.in_flight() < pipe.capacity()) { Job *j = new Job(); j->id = next_id++; j->want_pending = true; CHECK(pipe.submit(j)); }
    { Job extra; extra.id = -1; CHECK(!pipe.submit(&extra)); }
    while (pipe.in_flight() > 0) collect();
    CHECK(pipe.next() == nullptr && next_out == next_id);
    pipe.shutdown([](Job *j) { delete j; });
    uint64_t jobs = 0;                                   // (a stage counts a job after handing it on: exact only once the threads are joined)
    for (int k = 0; k < 3; k++) jobs += pipe.clock(k).jobs.load();
    CHECK(jobs == 3u * (uint64_t) next_id);
    CHECK(ops.bound.load() == 3 && ops.begun.load() == 1 && ops.ended.load() == 1);
}

// producer, consumer and a statistics reader on three threads; the stream is freed with work still in flight
void run_three_threads(int n_jobs, int depth, int max_us, int leave_in_flight) {
    StubOps ops(max_us, true);
    Pipe pipe(&ops, depth);
    pipe.start();
    std::atomic<int> submitted{0}, collected{0};
    std::atomic<bool> stop{false};
    const int to_collect = n_jobs - leave_in_flight;
    std::thread producer([&] {
        for (int i = 0; i < n_jobs;) {
            Job *j = new Job();
            j->id = i;
            j->want_pending = i % 3 != 1;
            if (pipe.submit(j)) { i++; submitted.store(i); }
            else { delete j; std::this_thread::yield(); }
        }
    });
    std::thread consumer([&] {
        int want = 0;
        while (want < to_collect) {
            Job *j = pipe.next();
            if (!j) { std::this_thread::yield(); continue; }
            CHECK(j->id == want && j->stages == 15);
            want++;
            collected.store(want);
            delete j;
        }
    });
    std::thread reader([&] {
        uint64_t seen = 0;
        while (!stop.load()) {
            uint64_t s = 0;
            for (int k = 0; k < 3; k++) s += pipe.clock(k).jobs.load() + pipe.clock(k).work_us.load() + pipe.clock(k).wait_in_us.load();
            CHECK(s >= seen || true);
            seen = s;
            (void) pipe.in_flight();
            std::this_thread::yield();
        }
    });
    producer.join();
    consumer.join();
    stop.store(true);
    reader.join();
    int dropped = 0;
    pipe.shutdown([&](Job *j) { CHECK(j->stages == 15); dropped++; delete j; });
    CHECK(dropped == leave_in_flight && submitted.load() == n_jobs && collected.load() == to_collect);
}

// queues alone: many producers and consumers, close() while consumers wait
void run_queue_storm() {
    ms::JobQueue<Job> q(3);
    std::atomic<int> got{0};
    std::vector<std::thread> th;
    for (int c = 0; c < 3; c++)
        th.emplace_back([&] { while (Job *j = q.pop()) { got.fetch_add(1); delete j; } });
    for (int p = 0; p < 3; p++)
        th.emplace_back([&, p] { for (int i = 0; i < 200; i++) { Job *j = new Job(); j->id = p * 1000 + i; q.push(j); } });
    for (int k = 3; k < 6; k++) th[(size_t) k].join();
    q.close();
    for (int k = 0; k < 3; k++) th[(size_t) k].join();
    CHECK(got.load() == 600);
    Job *j = reinterpret_cast<Job *>(1);
    CHECK(q.try_pop(&j) && j == nullptr);               // closed and drained
}

}  // namespace

int main() {
    for (unsigned seed = 1; seed <= 6; seed++) {
        run_single_thread(120, 1 + (int) (seed % 3), 60, true, seed);
        run_single_thread(60, 2, 0, seed % 2 == 0, 100 + seed);
    }
    run_single_thread(0, 2, 10, true, 9);                // a stream nobody submits to
    run_three_threads(300, 2, 40, 0);
    run_three_threads(150, 3, 25, 5);
    run_three_threads(40, 1, 0, 3);
    run_queue_storm();
    std::printf("stream_tsan: ok\n");
    return 0;
}
