// zl_pool.h -- a small persistent host thread pool for the host tails of the device work (the window Horner of an MSM is a few hundred
// group operations: worth spreading, not worth a thread spawn per call).  One pool per process (zl_pool_get, defined in zl_capi.hip).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

class zl_pool {
public:
    explicit zl_pool(unsigned workers) {
        for (unsigned i = 0; i < workers; i++) th_.emplace_back([this]() { loop(); });
    }
    ~zl_pool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // f(0) .. f(n - 1), each exactly once, on the workers and the caller; returns when all are done
    // (Several callers at once -- the completion threads of a proof's MSMs finish side by side -- each post their own job: the workers take tasks from
    // whichever posted job still has some.  Round 6: there was ONE job slot, so a second caller's job displaced the first one's and each caller ran most of
    // its own loop alone.)
    void parallel_for(size_t n, const std::function<void(size_t)>& f) {
        if (n == 0) return;
        if (n == 1 || th_.empty()) { for (size_t i = 0; i < n; i++) f(i); return; }
        auto job = std::make_shared<Job>();
        job->fn = &f;
        job->n = n;
        job->left.store(n, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(m_);
            jobs_.push_back(job);
        }
        cv_.notify_all();
        work(*job);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&]() { return job->left.load(std::memory_order_acquire) == 0; });
        for (size_t i = 0; i < jobs_.size(); i++)
            if (jobs_[i] == job) { jobs_.erase(jobs_.begin() + (long)i); break; }
    }

private:
    struct Job {  // one parallel_for; a worker that arrives late finds next >= n and leaves it alone
        const std::function<void(size_t)>* fn = nullptr;
        size_t n = 0;
        std::atomic<size_t> next{0}, left{0};
    };
    void work(Job& j) {
        for (;;) {
            const size_t i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j.n) return;
            (*j.fn)(i);
            if (j.left.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_);
                done_.notify_all();
            }
        }
    }
    void loop() {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() {
                    if (stop_) return true;
                    for (auto& x : jobs_)
                        if (x->next.load(std::memory_order_relaxed) < x->n) { j = x; return true; }
                    return false;
                });
                if (stop_) return;
            }
            work(*j);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<std::shared_ptr<Job>> jobs_;  // posted and not yet completed
    bool stop_ = false;
};
zl_pool& zl_pool_get();

// One persistent thread with a one-slot mailbox: for host work that makes HIP calls beside the caller (a fresh thread's first HIP call costs
// ~0.1 ms of per-thread runtime setup -- once per worker here instead of once per proof) and for issuing the launches of independent small
// jobs side by side (a 237-point MSM is ~25 launches = ~80 us of host time; four of them one after the other were the longest item of a
// small proof).  run() returns at once (after the previous task of this worker has finished); wait() blocks until the worker is idle.
class zl_worker {
public:
    zl_worker() : th_([this]() { loop(); }) {}
    ~zl_worker() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void run(std::function<void()> f) {
        std::unique_lock<std::mutex> lk(m_);
        idle_.wait(lk, [&]() { return !busy_; });
        task_ = std::move(f);
        busy_ = true;
        lk.unlock();
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        idle_.wait(lk, [&]() { return !busy_; });
    }

private:
    void loop() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return stop_ || busy_; });
                if (stop_ && !busy_) return;
                f = std::move(task_);
            }
            f();
            {
                std::lock_guard<std::mutex> lk(m_);
                busy_ = false;
            }
            idle_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, idle_;
    std::function<void()> task_;
    bool busy_ = false, stop_ = false;
    std::thread th_;  // last: the thread starts with every other member constructed
};
