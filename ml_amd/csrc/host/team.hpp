// A small persistent thread team for the K independent per-component factorizations of the host-side closing arithmetic
// (host/em_math.cpp) -- what `#pragma omp parallel for` did until round 5. The OpenMP runtime was dropped because its first use
// costs a process 60 - 160 ms on a many-core GPU host (topology discovery of the runtime's affinity layer), even when the `if`
// clause keeps the region serial: that was most of the first `EM::fit` of a process at the reference's own benchmark sizes
// (profiles/r05_first_call.txt). Here nothing happens until a region is actually worth threads; then the workers of the CALLING
// thread's team are started once and woken by a generation counter (10 - 20 us per region, like an OpenMP team).
//
// One team per calling thread (thread_local): the shards of a device group drive their contexts from threads of their own, and each
// gets its share of the cores (host::set_host_ranks), as with OpenMP. Plain C++17, no HIP: the protocol is stress-tested under
// -fsanitize=thread (tests/cpp/team_stress.cpp).
#pragma once
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace mlhip {
namespace host {

class Team {
public:
    Team() = default;
    Team(const Team&) = delete;
    Team& operator=(const Team&) = delete;
    ~Team()
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
            ++generation_;
        }
        wake_.notify_all();
        for (std::thread& t : workers_) t.join();
    }

    /// The calling thread's team.
    static Team& mine()
    {
        thread_local Team team;
        return team;
    }

    /// fn(index) for index = 0 .. count - 1, dealt in contiguous static chunks to `threads` participants (the caller is one of them);
    /// returns when all are done. The first exception thrown by any participant is rethrown here. threads <= 1: a plain loop.
    void for_each(int count, int threads, const std::function<void(int)>& fn)
    {
        if (threads > count) threads = count;
        if (threads <= 1) {
            for (int i = 0; i < count; ++i) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lock(m_);
            while ((int)workers_.size() < threads - 1) {
                const int index = (int)workers_.size() + 1;                      // participant 0 is the caller
                workers_.emplace_back([this, index, seen = generation_] { work(index, seen); });
            }
            fn_ = &fn;
            count_ = count;
            participants_ = threads;
            pending_ = threads - 1;
            failure_ = nullptr;
            ++generation_;
        }
        wake_.notify_all();
        run_share(0, threads, count, fn);
        std::unique_lock<std::mutex> lock(m_);
        done_.wait(lock, [this] { return pending_ == 0; });
        fn_ = nullptr;
        if (failure_) {
            std::exception_ptr e = failure_;
            failure_ = nullptr;
            std::rethrow_exception(e);
        }
    }

private:
    void run_share(int index, int participants, int count, const std::function<void(int)>& fn)
    {
        const int lo = (int)((long long)count * index / participants), hi = (int)((long long)count * (index + 1) / participants);
        try {
            for (int i = lo; i < hi; ++i) fn(i);
        } catch (...) {
            std::lock_guard<std::mutex> lock(m_);
            if (!failure_) failure_ = std::current_exception();
        }
    }

    void work(int index, unsigned long long seen)
    {
        for (;;) {
            const std::function<void(int)>* fn = nullptr;
            int participants = 0, count = 0;
            {
                std::unique_lock<std::mutex> lock(m_);
                wake_.wait(lock, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                if (index >= participants_) continue;                            // (a smaller region than the team: not this worker's)
                fn = fn_; participants = participants_; count = count_;
            }
            run_share(index, participants, count, *fn);
            {
                std::lock_guard<std::mutex> lock(m_);
                --pending_;
            }
            done_.notify_one();
        }
    }

    std::mutex m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* fn_ = nullptr;
    int count_ = 0, participants_ = 0, pending_ = 0;
    unsigned long long generation_ = 0;
    bool stop_ = false;
    std::exception_ptr failure_;
};

}  // namespace host
}  // namespace mlhip
