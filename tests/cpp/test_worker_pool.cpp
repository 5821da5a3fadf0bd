// Stress of csrc/worker_pool.hpp (the long-lived helper threads of the staging copies and of the construction windows):
// every task index runs exactly once per section, sections of different sizes follow each other, sections started from
// inside a section and sections of concurrent callers share the helper threads (nobody deadlocks, nobody is silently
// serialised behind a long section), asynchronous jobs run while a long section occupies the pool.
#include <atomic>
#include <chrono>
#include <set>
#include <mutex>
#include <cstdio>
#include <stdexcept>
#include <thread>
#include <vector>

#include "worker_pool.hpp"

using hnswgpu::WorkerPool;

int main() {
    WorkerPool& pool = WorkerPool::instance();
    // 1. many sections, every index exactly once, results visible to the caller afterwards
    for (int round = 0; round < 3000; ++round) {
        const unsigned n = 1u + (unsigned)(round * 7 % 61);
        std::vector<int> hit(n, 0);
        std::atomic<unsigned> sum{0};
        pool.run(n, 1u + (unsigned)(round % 16), [&](unsigned t) {
            hit[t] += 1;
            sum.fetch_add(t + 1, std::memory_order_relaxed);
        });
        unsigned want = 0;
        for (unsigned t = 0; t < n; ++t) {
            if (hit[t] != 1) { std::printf("round %d: task %u ran %d times\n", round, t, hit[t]); return 1; }
            want += t + 1;
        }
        if (sum.load() != want) { std::printf("round %d: sum %u != %u\n", round, sum.load(), want); return 1; }
    }
    // 2. a section inside a section (runs inline on the worker that asked)
    std::atomic<unsigned> inner{0};
    pool.run(8, 8, [&](unsigned) { pool.run(5, 4, [&](unsigned) { inner.fetch_add(1); }); });
    if (inner.load() != 40) { std::printf("nested: %u != 40\n", inner.load()); return 1; }
    // 3. concurrent callers: whoever finds the pool busy does its own work
    std::atomic<unsigned> total{0};
    std::vector<std::thread> callers;
    for (int c = 0; c < 6; ++c)
        callers.emplace_back([&]() {
            for (int r = 0; r < 400; ++r) pool.run(9, 4, [&](unsigned) { total.fetch_add(1, std::memory_order_relaxed); });
        });
    for (auto& t : callers) t.join();
    if (total.load() != 6u * 400u * 9u) { std::printf("concurrent: %u\n", total.load()); return 1; }
    // 4. a task that throws: the other tasks of the section still run, the exception arrives on the caller, the pool lives on
    for (int round = 0; round < 200; ++round) {
        std::atomic<unsigned> ran{0};
        bool caught = false;
        try {
            pool.run(24, 8, [&](unsigned t) {
                ran.fetch_add(1, std::memory_order_relaxed);
                if (t == (unsigned)(round % 24)) throw std::runtime_error("task failed");
            });
        } catch (const std::runtime_error&) {
            caught = true;
        }
        if (!caught || ran.load() != 24u) { std::printf("throwing task: caught %d, %u of 24 ran\n", (int)caught, ran.load()); return 1; }
    }
    // 5. a long section does not serialise a second caller's section: while one caller keeps 4 threads busy for a long time,
    // another caller's section must still be served by more than one thread (the old pool ran it on the
    // caller alone, silently)
    if (std::thread::hardware_concurrency() >= 4) {
        std::atomic<bool> stop{false};
        std::thread long_caller([&]() {
            pool.run(4, 4, [&](unsigned) { while (!stop.load(std::memory_order_relaxed)) std::this_thread::yield(); });
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        std::mutex m;
        std::set<std::thread::id> who;
        pool.run(64, 4, [&](unsigned) {
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            std::lock_guard<std::mutex> g(m);
            who.insert(std::this_thread::get_id());
        });
        stop.store(true);
        long_caller.join();
        if (who.size() < 2) { std::printf("a second section next to a long one ran on %zu thread(s)\n", who.size()); return 1; }
    }
    // 6. asynchronous jobs (the tickets of the C ABI): they run, wait() returns after them, also while sections keep the pool busy
    {
        std::atomic<unsigned> jobs_done{0};
        std::atomic<bool> stop{false};
        std::thread busy([&]() { pool.run(8, 8, [&](unsigned) { while (!stop.load(std::memory_order_relaxed)) std::this_thread::yield(); }); });
        std::vector<std::shared_ptr<WorkerPool::Job>> js;
        for (int i = 0; i < 32; ++i) js.push_back(pool.submit([&]() { jobs_done.fetch_add(1); }));
        for (auto& j : js) j->wait();
        if (jobs_done.load() != 32u) { std::printf("jobs: %u of 32 ran\n", jobs_done.load()); return 1; }
        stop.store(true);
        busy.join();
    }
    // 6b. two jobs issued back to back do not share one thread (with ONE idle thread both went to the front of the
    // queue and ran one after the other, newest first): each waits for the other to have started -- that only ends if they overlap
    for (int round = 0; round < 50; ++round) {
        std::atomic<int> started{0};
        std::atomic<bool> gave_up{false};
        auto body = [&]() {
            started.fetch_add(1);
            const auto t0 = std::chrono::steady_clock::now();
            while (started.load() < 2) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { gave_up.store(true); break; }
                std::this_thread::yield();
            }
        };
        auto j1 = pool.submit(body);
        auto j2 = pool.submit(body);
        j1->wait();
        j2->wait();
        if (gave_up.load()) { std::printf("two jobs issued back to back ran one after the other (round %d)\n", round); return 1; }
        if (round % 5 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(2));  // (threads asleep again)
    }
    // 6c. a burst of jobs from one thread: every one runs exactly once, wait() returns after it
    {
        std::mutex m;
        std::vector<int> ran(16, 0);
        std::vector<std::shared_ptr<WorkerPool::Job>> js;
        for (int i = 0; i < 16; ++i) js.push_back(pool.submit([&, i]() { std::lock_guard<std::mutex> g(m); ran[(size_t)i] += 1; }));
        for (auto& j : js) j->wait();
        for (int i = 0; i < 16; ++i)
            if (ran[(size_t)i] != 1) { std::printf("burst of jobs: job %d ran %d times\n", i, ran[(size_t)i]); return 1; }
    }
    // 7. a pool that has grown large (a build on every core) and small sections after it, at gaps shorter and longer than the
    // helpers' lingering time: the section wakes only the sleepers the lingering threads leave work for -- every task still runs
    // exactly once, helpers still take part (not always: a section never waits for one), and a section of 8 short tasks is not
    // held up by hundreds of threads waking for nothing
    {
        std::atomic<unsigned> grown{0};
        pool.run(96, 96, [&](unsigned) { grown.fetch_add(1); std::this_thread::sleep_for(std::chrono::milliseconds(2)); });
        if (grown.load() != 96u) { std::printf("grow: %u of 96 ran\n", grown.load()); return 1; }
        std::this_thread::sleep_for(std::chrono::milliseconds(5));  // everyone asleep
        unsigned helped = 0;
        for (int round = 0; round < 1500; ++round) {
            std::vector<int> hit(8, 0);
            std::mutex m;
            std::set<std::thread::id> who;
            pool.run(8, 8, [&](unsigned t) {
                hit[t] += 1;
                std::this_thread::sleep_for(std::chrono::microseconds(30));
                std::lock_guard<std::mutex> g(m);
                who.insert(std::this_thread::get_id());
            });
            for (unsigned t = 0; t < 8; ++t)
                if (hit[t] != 1) { std::printf("large pool, round %d: task %u ran %d times\n", round, t, hit[t]); return 1; }
            if (who.size() > 1) ++helped;
            if (round % 3 == 0) std::this_thread::sleep_for(std::chrono::microseconds(round % 7 == 0 ? 600 : 50));
        }
        if (std::thread::hardware_concurrency() >= 4 && helped < 750) { std::printf("large pool: helpers took part in %u of 1500 sections\n", helped); return 1; }
    }
    std::atomic<unsigned> after{0};
    pool.run(16, 8, [&](unsigned) { after.fetch_add(1); });
    if (after.load() != 16u) { std::printf("after a throwing section: %u != 16\n", after.load()); return 1; }
    std::printf("worker pool OK\n");
    return 0;
}
