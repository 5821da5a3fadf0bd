// worker_pool.hpp -- long-lived host threads for the parallel sections of the library (staging copies of a host-buffer
// search, the host side of a construction window, a whole host build) and for its asynchronous calls (tickets).  Creating
// std::threads per section costs more than some of the sections themselves (three sections per window, ~100 windows per
// build; two per search call).
//
// Sections of different callers share the threads: every section queues "invitations" for helpers, idle threads take them in
// arrival order, and the caller always works on its own section -- so a section never waits for a thread (it is at worst run
// by its caller alone while the helpers are busy elsewhere, and helpers that come free join it late), sections may be nested,
// and a long section (a 48 s host build) does not turn every other one into a serial loop.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace hnswgpu {

class WorkerPool {
public:
    // the process-wide pool (never destroyed: its threads may outlive static destructors)
    static WorkerPool& instance() {
        static WorkerPool* p = new WorkerPool();
        return *p;
    }
    // the most helper threads a section can get (the caller comes on top).  A section gets the thread count it asks for, also
    // beyond the hardware threads of the box (oversubscription is the caller's choice: the race tests rely on it); threads
    // are created on demand and then kept.
    unsigned max_helpers() const { return cap_; }

    // Runs fn(0) .. fn(n_tasks - 1), the caller taking part, on at most max_threads threads (capped at max_helpers() + 1);
    // returns when all are done.  A task that throws does not take the process down on a helper thread: the section is still
    // run to its end (the remaining tasks included) and the first exception is rethrown here, on the caller.
    void run(unsigned n_tasks, unsigned max_threads, const std::function<void(unsigned)>& fn) {
        if (n_tasks == 0) return;
        if (n_tasks == 1 || max_threads <= 1) {
            for (unsigned t = 0; t < n_tasks; ++t) fn(t);
            return;
        }
        Section s;
        s.fn = &fn;
        s.n = n_tasks;
        const unsigned helpers = std::min(std::min(n_tasks, max_threads) - 1u, cap_);
        {
            std::lock_guard<std::mutex> g(mu_);
            for (unsigned i = 0; i < helpers; ++i) queue_.push_back(Item{&s, nullptr});
            pending_.fetch_add(helpers, std::memory_order_relaxed);
            s.invited = helpers;
            grow_locked(helpers);
        }
        // Wake as many sleepers as there are invitations the lingering (awake) threads will not take -- not all of them: after a
        // build on every core the pool holds hundreds of threads, and a notify_all per 1.3 ms search call had each of them wake,
        // queue for the mutex and go back to sleep (~130 us per call, measured through the reference's FFI symbol).  A lingerer
        // that times out at this very moment leaves its invitation to be withdrawn below: a section never waits for a helper.
        const unsigned awake = lingering_.load(std::memory_order_relaxed);
        for (unsigned i = awake; i < helpers; ++i) cv_.notify_one();
        work(s);
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> g(mu_);
            // invitations nobody took are withdrawn; helpers inside the section are waited for
            for (auto it = queue_.begin(); it != queue_.end();) {
                if (it->section == &s) { it = queue_.erase(it); --s.invited; pending_.fetch_sub(1, std::memory_order_relaxed); }
                else ++it;
            }
            s.done_cv.wait(g, [&]() { return s.invited == 0; });
            err = s.err;
        }
        if (err) std::rethrow_exception(err);
    }

    // An asynchronous one-off job (a ticket of the C ABI): runs on a pool thread of its own -- one is created when the idle ones are
    // spoken for -- so a job queues neither behind a long section nor behind another job; when no thread can be had at all it runs
    // on the caller.  wait() blocks until it has run; a job that throws is the
    // caller's bug (jobs catch their own errors).
    class Job {
    public:
        void wait() {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [this]() { return done_; });
        }
    private:
        friend class WorkerPool;
        std::function<void()> fn_;
        std::mutex mu_;
        std::condition_variable cv_;
        bool done_ = false;
    };
    std::shared_ptr<Job> submit(std::function<void()> fn) {
        std::shared_ptr<Job> j(new Job());
        j->fn_ = std::move(fn);
        bool queued = false;
        {
            std::lock_guard<std::mutex> g(mu_);
            // A thread per waiting job: the idle (or just created) threads may all be spoken for by jobs queued a moment ago --
            // two tickets issued back to back must not run one after the other on the one idle thread.
            bool have_thread = idle_ + spawning_ > queued_jobs_;
            if (!have_thread && n_threads_ < HARD_CAP) have_thread = spawn_locked();
            if (have_thread || idle_ + spawning_ > 0) {  // (no thread of its own but some thread will come by: it waits its turn)
                // ahead of the invitations (a job has no caller working on it), behind the jobs already waiting (first in, first out)
                queue_.insert(queue_.begin() + (std::ptrdiff_t)queued_jobs_, Item{nullptr, j});
                ++queued_jobs_;
                pending_.fetch_add(1, std::memory_order_relaxed);
                queued = true;
            }
        }
        if (queued) {
            cv_.notify_one();
        } else {  // no thread to be had and none idle: run it here rather than never
            run_job(*j);
        }
        return j;
    }

private:
    struct Section {
        const std::function<void(unsigned)>* fn = nullptr;
        unsigned n = 0;
        std::atomic<unsigned> next{0};
        unsigned invited = 0;            // invitations queued or taken and not yet finished (guarded by the pool's mu_)
        std::condition_variable done_cv;
        std::exception_ptr err;          // first exception (guarded by mu_)
    };
    struct Item {
        Section* section;                // an invitation to help with a section, or
        std::shared_ptr<Job> job;        // an asynchronous job
    };
    static constexpr unsigned HARD_CAP = 1024;  // threads this pool will ever create

    WorkerPool() : cap_(255u) {
        // HNSWGPU_POOL_SPIN_US: how long an idle pool thread stays awake before it sleeps (default 200; 0 = sleep at once)
        if (const char* e = std::getenv("HNSWGPU_POOL_SPIN_US")) linger_us_ = std::max(0, std::atoi(e));
    }
    void work(Section& s) {
        for (;;) {
            const unsigned t = s.next.fetch_add(1, std::memory_order_relaxed);
            if (t >= s.n) break;
            try {
                (*s.fn)(t);
            } catch (...) {
                std::lock_guard<std::mutex> g(mu_);
                if (!s.err) s.err = std::current_exception();
            }
        }
    }
    static void run_job(Job& j) {
        try {
            j.fn_();
        } catch (...) {
        }
        {
            std::lock_guard<std::mutex> g(j.mu_);
            j.done_ = true;
        }
        j.cv_.notify_all();
    }
    // threads for `wanted` queued invitations beyond what the idle ones can take (sections never create more than cap_
    // threads in all; jobs may go beyond)
    void grow_locked(unsigned wanted) {
        while (idle_ + spawning_ < wanted && n_threads_ < cap_) {
            if (!spawn_locked()) break;
        }
    }
    bool spawn_locked() {
        try {
            std::thread([this]() { loop(); }).detach();
        } catch (...) {
            return false;  // no more threads to be had: the ones there are (and the callers) do the work
        }
        ++n_threads_;
        ++spawning_;
        return true;
    }
    void loop() {
        std::unique_lock<std::mutex> g(mu_);
        --spawning_;
        for (;;) {
            ++idle_;
            if (queue_.empty() && linger_us_ > 0) {
                // stay awake for a moment: the next section of a caller that issues call after call (a batch every ~1.3 ms)
                // then finds its helpers running instead of paying ~50 us per wake-up
                lingering_.fetch_add(1, std::memory_order_relaxed);
                g.unlock();
                const auto t0 = std::chrono::steady_clock::now();
                unsigned n = 0;
                while (pending_.load(std::memory_order_relaxed) == 0) {
#if defined(__x86_64__) || defined(__i386__)
                    __builtin_ia32_pause();
#endif
                    if ((++n & 63u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(linger_us_)) break;
                }
                g.lock();
                lingering_.fetch_sub(1, std::memory_order_relaxed);
            }
            cv_.wait(g, [this]() { return !queue_.empty(); });
            --idle_;
            Item it = std::move(queue_.front());
            queue_.pop_front();
            if (it.job) --queued_jobs_;  // (jobs sit in front of the invitations)
            pending_.fetch_sub(1, std::memory_order_relaxed);
            g.unlock();
            if (it.job) {
                run_job(*it.job);
                it.job.reset();
                g.lock();
            } else {
                Section* s = it.section;
                work(*s);
                g.lock();
                if (--s->invited == 0) s->done_cv.notify_all();  // (the section lives until its caller has seen invited == 0)
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Item> queue_;
    unsigned n_threads_ = 0, idle_ = 0, spawning_ = 0;
    unsigned queued_jobs_ = 0;          // jobs at the front of queue_ (in arrival order), not yet taken by a thread
    std::atomic<unsigned> pending_{0};  // items in queue_ (read without the lock by lingering threads)
    std::atomic<unsigned> lingering_{0};  // idle threads that are awake, polling pending_
    int linger_us_ = 200;
    const unsigned cap_;
};

}  // namespace hnswgpu
