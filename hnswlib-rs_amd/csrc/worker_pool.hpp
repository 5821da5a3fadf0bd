// worker_pool.hpp -- a few long-lived host threads for the short parallel sections of the library (staging copies of a
// host-buffer search, the host side of a construction window).  Creating std::threads per section costs more than some of the
// sections themselves (three sections per window, ~100 windows per build; two per search call).
#pragma once
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace hnswgpu {

class WorkerPool {
public:
    // the process-wide pool (never destroyed: its threads may outlive static destructors)
    static WorkerPool& instance() {
        static WorkerPool* p = new WorkerPool();
        return *p;
    }
    // Runs fn(0) .. fn(n_tasks - 1), the caller taking part, on at most max_threads threads; returns when all are done.
    // A pool that is busy with another caller's section (or a section started from inside one) runs the tasks on the caller.
    // A task that throws does not take the process down on a helper thread: the section is still run to its end (the
    // remaining tasks included) and the first exception is rethrown here, on the caller.
    void run(unsigned n_tasks, unsigned max_threads, const std::function<void(unsigned)>& fn) {
        if (n_tasks == 0) return;
        std::unique_lock<std::mutex> busy(run_mu_, std::try_to_lock);
        if (n_tasks == 1 || max_threads <= 1 || !busy.owns_lock()) {
            for (unsigned t = 0; t < n_tasks; ++t) fn(t);
            return;
        }
        const unsigned helpers = std::min(std::min(n_tasks, max_threads) - 1u, limit_);
        {
            std::lock_guard<std::mutex> g(mu_);
            while (threads_.size() < helpers) {
                try {
                    threads_.emplace_back([this]() { loop(); });
                    threads_.back().detach();
                } catch (...) {
                    break;  // no more threads to be had: the ones there are (and the caller) do the work
                }
            }
            job_fn_ = &fn;
            job_n_ = n_tasks;
            job_next_.store(0, std::memory_order_relaxed);
            job_open_ = std::min<unsigned>(helpers, (unsigned)threads_.size());  // helpers that may still join this section
            job_left_ = 0;                                                       // helpers inside it
            ++generation_;
        }
        cv_.notify_all();
        work(fn, n_tasks);
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> g(mu_);
            job_open_ = 0;  // late wakers find the section closed
            done_cv_.wait(g, [this]() { return job_left_ == 0; });
            job_fn_ = nullptr;
            err = err_;
            err_ = nullptr;
        }
        if (err) std::rethrow_exception(err);
    }

private:
    WorkerPool() : limit_(std::max(1u, std::min(63u, std::thread::hardware_concurrency()) ) - 1u) {}
    void work(const std::function<void(unsigned)>& fn, unsigned n) {
        for (;;) {
            const unsigned t = job_next_.fetch_add(1, std::memory_order_relaxed);
            if (t >= n) break;
            try {
                fn(t);
            } catch (...) {
                std::lock_guard<std::mutex> g(mu_);
                if (!err_) err_ = std::current_exception();
            }
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned)>* fn = nullptr;
            unsigned n = 0;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&]() { return generation_ != seen; });
                seen = generation_;
                if (job_open_ == 0) continue;  // the section is full or already over
                --job_open_;
                ++job_left_;
                fn = job_fn_;
                n = job_n_;
            }
            work(*fn, n);
            {
                std::lock_guard<std::mutex> g(mu_);
                --job_left_;
            }
            done_cv_.notify_all();
        }
    }
    std::mutex run_mu_;   // one section at a time
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> threads_;
    const std::function<void(unsigned)>* job_fn_ = nullptr;
    unsigned job_n_ = 0, job_open_ = 0, job_left_ = 0;
    std::atomic<unsigned> job_next_{0};
    uint64_t generation_ = 0;
    std::exception_ptr err_;  // first exception of the running section (guarded by mu_)
    const unsigned limit_;
};

}  // namespace hnswgpu
