/* vslam_pool.h -- private: the small worker pool of a context (host-side staging copies, host quadtree fallback).
 * No HIP dependency, so tests/cpp/pool_stress.cpp can build it with -fsanitize=thread. */
#ifndef VSLAM_POOL_H
#define VSLAM_POOL_H
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

/* Every parallel_for owns a Job (index counter, total, function, completion count).  A worker takes a
 * shared_ptr to the job it was woken for under the lock and claims indices from THAT job only, so a worker
 * that is late leaving job k can never claim an index of job k+1 (the first version kept next/total/fn in the
 * pool itself and had exactly that race). */
class WorkerPool {
    struct Job {
        std::atomic<int> next{0};
        int total = 0;
        std::atomic<int> done{0};
        const std::function<void(int)>* fn = nullptr;
    };

public:
    explicit WorkerPool(int n) : stop_(false), gen_(0) {
        for (int i = 0; i < n; i++) th_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (th_.empty() || n == 1) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        auto job = std::make_shared<Job>();
        job->total = n;
        job->fn = &fn;
        {
            std::lock_guard<std::mutex> l(m_);
            job_ = job;
            gen_++;
        }
        cv_.notify_all();
        run(*job);
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [&] { return job->done.load() == n; });
        if (job_ == job) job_.reset(); /* late wakers find no job; `fn` is not touched after this point */
    }

private:
    void run(Job& j) {
        for (;;) {
            const int i = j.next.fetch_add(1);
            if (i >= j.total) break;
            (*j.fn)(i); /* an index below total was claimed: parallel_for cannot have returned yet */
            if (j.done.fetch_add(1) + 1 == j.total) {
                std::lock_guard<std::mutex> l(m_);
                done_.notify_all();
            }
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                j = job_;
            }
            if (j) run(*j);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    bool stop_;
    unsigned long gen_;
    std::shared_ptr<Job> job_;
};

#endif
