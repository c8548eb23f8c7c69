// host_pool.hpp -- a small persistent pool of host worker threads (the factor work of minpack.cpp, the state machines of
// the lock-step engine in batchsolve.cpp).  The reference starts and joins std::threads on every residual call
// (shooting.cpp:1152-1157); here threads are started once per pool and handed batches of tasks.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace socp {

// Started once, handed a batch of tasks per call of run() (an atomic counter deals them out, so a thread that finishes early
// takes the next task).  Between batches a worker waits actively for at most ~40 us and then
// parks on a condition variable (the process may run under a CPU quota: no unbounded spinning).  Starting and joining 15
// threads at each of the 26 panels of n = 832 cost as much as the arithmetic.
class Pool {
public:
    explicit Pool(int threads) : n_(std::max(1, threads))
    {
        for (int w = 1; w < n_; w++) th_.emplace_back([this]() { worker(); });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_.store(true); }
        cv_start_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    int size() const { return n_; }
    // The tasks may throw (the segment workers of the host shooting path run USER model code: ComputeTraj, the boundary
    // functions, SwitchingStateFunction): the first exception of a batch -- whichever thread raised it -- is kept, the batch is
    // drained (no task starts after it; the workers are waited for, the job pointer is cleared) and the exception is rethrown on
    // the CALLING thread, where a throw from the serial path would have surfaced (a throw on a worker called
    // std::terminate, a throw on the caller unwound while workers still used the stack-local job).
    template <class Body>
    void run(int tasks, Body &&body)
    {
        if (tasks <= 0) return;
        if (n_ == 1 || tasks == 1) { for (int t = 0; t < tasks; t++) body(t); return; }
        std::function<void(int)> fn = body;
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn; tasks_ = tasks; next_.store(0); busy_.store(n_ - 1); failed_.store(false); error_ = nullptr; gen_.fetch_add(1);
        }
        cv_start_.notify_all();
        take_tasks(fn, tasks);
        if (!spin_until([this]() { return busy_.load() == 0; })) {
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [this]() { return busy_.load() == 0; });
        }
        std::exception_ptr err;
        {
            std::lock_guard<std::mutex> lk(m_);     // the last worker has left its critical section
            job_ = nullptr;
            err = error_;
            error_ = nullptr;
        }
        if (err) std::rethrow_exception(err);
    }

private:
    // this thread's share of a batch; an exception ends the batch for everybody (the counter is pushed past the end)
    void take_tasks(const std::function<void(int)> &fn, int tasks)
    {
        for (int t; !failed_.load(std::memory_order_relaxed) && (t = next_.fetch_add(1)) < tasks;) {
            try {
                fn(t);
            } catch (...) {
                std::lock_guard<std::mutex> lk(m_);
                if (!error_) error_ = std::current_exception();
                failed_.store(true);
                next_.store(tasks);
            }
        }
    }
    // A batch is tens of microseconds of work per thread and the next one follows at once, while waking a parked thread
    // costs 50-100 us: wait actively for a SHORT, bounded time (about 40 us), then park on the condition variable.
    template <class Pred>
    static bool spin_until(Pred &&done)
    {
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        for (int it = 0;; it++) {
            if (done()) return true;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(40)) return false;
        }
    }
    void worker()
    {
        int seen = 0;
        for (;;) {
            if (!spin_until([&]() { return stop_.load() || gen_.load() != seen; })) {
                std::unique_lock<std::mutex> lk(m_);
                cv_start_.wait(lk, [&]() { return stop_.load() || gen_.load() != seen; });
            }
            if (stop_.load()) return;
            const std::function<void(int)> *fn;
            int tasks;
            {
                std::lock_guard<std::mutex> lk(m_);
                seen = gen_.load(); fn = job_; tasks = tasks_;
            }
            if (fn) take_tasks(*fn, tasks);
            bool last;
            {
                std::lock_guard<std::mutex> lk(m_);
                last = busy_.fetch_sub(1) == 1;
            }
            if (last) cv_done_.notify_one();
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_start_, cv_done_;
    const std::function<void(int)> *job_ = nullptr;
    int tasks_ = 0;
    std::atomic<int> next_{0}, busy_{0}, gen_{0};
    std::atomic<bool> stop_{false}, failed_{false};
    std::exception_ptr error_;                 // first exception of the batch in hand (guarded by m_)
};

}  // namespace socp
