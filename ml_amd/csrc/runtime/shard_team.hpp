// The host threads of a device group (runtime/group.cpp) and the rules they meet by, WITHOUT any HIP: a barrier that a failing
// shard can tear down, and a team of one persistent thread per shard that runs a task on every shard, reports the failure that
// happened FIRST and makes every shard pass through a recovery step before the next task. Kept apart from the GPU code so that the
// protocol can be stress-tested on a CPU under -fsanitize=thread / address (tests/cpp/team_stress.cpp, tests/test_sanitizers.py);
// the reference is single-threaded: these threads are this library's own invention.
#pragma once
#include <atomic>
#include <climits>
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

namespace mlhip_rt {

struct GroupAborted : std::runtime_error {
    GroupAborted() : std::runtime_error("another shard of the device group failed") {}
};

/// Barrier of the shard threads that can be torn down: a shard that fails (an exception on its way out of the task) aborts it, and
/// the shards waiting in it -- or arriving later -- fail too instead of waiting for ever. Arrivals spin briefly before they sleep:
/// the barrier sits inside every all-reduce of an iteration that may take tens of microseconds.
class Barrier {
public:
    void reset(int n) { n_ = n; arrived_.store(0); aborted_.store(false); }
    void abort()
    {
        aborted_.store(true);
        std::lock_guard<std::mutex> lock(m_);
        cv_.notify_all();
    }
    bool aborted() const { return aborted_.load(); }
    void wait()
    {
        if (aborted_.load()) throw GroupAborted();
        const uint64_t gen = generation_.load();
        if (arrived_.fetch_add(1) + 1 == n_) {
            arrived_.store(0);
            {
                std::lock_guard<std::mutex> lock(m_);
                generation_.fetch_add(1);
            }
            cv_.notify_all();
            return;
        }
        for (int spin = 0; spin < 4000; ++spin) {
            if (generation_.load() != gen) return;
            if (aborted_.load()) throw GroupAborted();
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
        std::unique_lock<std::mutex> lock(m_);
        cv_.wait(lock, [&] { return generation_.load() != gen || aborted_.load(); });
        if (generation_.load() == gen) throw GroupAborted();
    }

private:
    int n_ = 1;
    std::atomic<int> arrived_{0};
    std::atomic<uint64_t> generation_{0};
    std::atomic<bool> aborted_{false};
    std::mutex m_;
    std::condition_variable cv_;
};

/// One persistent thread per shard.
///   run(f):   f(shard) on every shard's thread; returns when all are done. If any shard threw, the exception that was thrown FIRST
///             is rethrown in the caller's thread (the others are its consequences: shards torn out of `barrier`), and the team is
///             `dirty`: the next run() first calls recover(shard) on every shard's thread (twice bracketed by the barrier, see
///             ShardTeam::run) before f.
///   on_abort: called ONCE per failed task, from the thread of the shard that failed first, right after the barrier was aborted --
///             where the group cancels work that the other shards may be blocked in OUTSIDE the barrier (a collective that waits
///             for the failed shard: runtime/group.cpp aborts the RCCL communicators here).
class ShardTeam {
public:
    Barrier barrier;

    ShardTeam() = default;
    ShardTeam(const ShardTeam&) = delete;
    ShardTeam& operator=(const ShardTeam&) = delete;
    ~ShardTeam() { stop(); }

    void start(int n, std::function<void(int)> recover, std::function<void(int)> on_abort)
    {
        n_ = n;
        recover_ = std::move(recover);
        on_abort_ = std::move(on_abort);
        errors_.assign((size_t)n, nullptr);
        error_order_.assign((size_t)n, INT_MAX);
        for (int s = 0; s < n; ++s) workers_.emplace_back([this, s] { loop(s); });
    }

    void stop()
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            quit_ = true;
        }
        cv_work_.notify_all();
        for (std::thread& t : workers_)
            if (t.joinable()) t.join();
        workers_.clear();
    }

    int size() const { return n_; }
    bool dirty() const { return dirty_; }

    void run(const std::function<void(int)>& f)
    {
        const bool recover = dirty_;
        dirty_ = false;
        const std::function<void(int)> task = [&](int s) {
            if (recover) {
                // every shard first brings ITS side to rest (drains its stream), all meet, then every shard resets what the
                // others may have been reading (its slots of the in-process all-reduce), all meet again: nobody starts the new
                // task against a half-reset neighbour
                if (recover_) recover_(s);
            }
            f(s);
        };
        {
            std::lock_guard<std::mutex> lock(m_);
            task_ = &task;
            errors_.assign((size_t)n_, nullptr);
            error_order_.assign((size_t)n_, INT_MAX);
            error_seq_ = 0;
            pending_ = n_;
            barrier.reset(n_);
            ++generation_;
        }
        cv_work_.notify_all();
        {
            std::unique_lock<std::mutex> lock(m_);
            cv_done_.wait(lock, [&] { return pending_ == 0; });
            task_ = nullptr;
        }
        int first = -1;
        for (int s = 0; s < n_; ++s)
            if (errors_[(size_t)s] && (first < 0 || error_order_[(size_t)s] < error_order_[(size_t)first])) first = s;
        if (first >= 0) {
            dirty_ = true;
            std::rethrow_exception(errors_[(size_t)first]);
        }
    }

private:
    void loop(int s)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* task = nullptr;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_work_.wait(lock, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_;
                task = task_;
            }
            std::exception_ptr err;
            try {
                (*task)(s);
            } catch (...) {
                err = std::current_exception();
                bool first;
                {
                    std::lock_guard<std::mutex> lock(m_);      // (its place in the order of failures BEFORE the others are torn out)
                    first = error_seq_ == 0;
                    errors_[(size_t)s] = err;
                    error_order_[(size_t)s] = error_seq_++;
                }
                barrier.abort();
                if (first && on_abort_) {
                    try { on_abort_(s); } catch (...) {}
                }
            }
            {
                std::lock_guard<std::mutex> lock(m_);
                if (--pending_ == 0) cv_done_.notify_all();
            }
        }
    }

    int n_ = 0;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    uint64_t generation_ = 0;
    int pending_ = 0;
    bool quit_ = false;
    const std::function<void(int)>* task_ = nullptr;
    std::vector<std::exception_ptr> errors_;
    std::vector<int> error_order_;
    int error_seq_ = 0;
    bool dirty_ = false;
    std::function<void(int)> recover_, on_abort_;
};

/// The in-process all-reduce of a device group whose shards share a process (runtime/group.cpp, kDirect): every shard copies its
/// buffer into its own slot, waits ON ITS STREAM for the other shards' slots and sums all slots in shard order -- bit-identical on
/// every shard. Two slot generations alternate, so that a shard may start filling the slots of all-reduce i + 1 while others still
/// read those of all-reduce i. This class is the PROTOCOL -- who waits for what, when a slot may be overwritten, how the slots grow
/// and how a failed task is cleaned up; everything that touches a device goes through `Ops` (HIP streams and events in group.cpp,
/// plain memory with ordering checks in tests/cpp/team_stress.cpp):
///   sync_stream(r)               -- shard r's stream is idle
///   reserve_slots(r, doubles)    -- shard r's two slots hold at least that many doubles (contents undefined)
///   release_slots(r)             -- ... are given back
///   wait_consumed(r, p, q)       -- r's stream waits until shard q has finished reading generation p (its last `mark_consumed`)
///   publish(r, p, buf, count)    -- r's stream: buf -> slot (p, r), then mark it ready
///   wait_ready(r, p, q)          -- r's stream waits until slot (p, q) is ready
///   sum(r, p, buf, count)        -- r's stream: buf = slot(p, 0) + slot(p, 1) + ... in shard order
///   mark_consumed(r, p)          -- r's stream: r has finished reading generation p
/// Every shard calls allreduce() with the same count, in the same order, from its own thread.
template <class Ops> class SlotAllreduce {
public:
    void init(int n)
    {
        n_ = n;
        capacity_.assign((size_t)n, 0);
        sequence_.assign((size_t)n, 0);
    }
    bool active() const { return n_ > 0; }

    void allreduce(Ops& ops, Barrier& barrier, int r, double* buf, size_t count)
    {
        const size_t R = (size_t)r;
        if (count > capacity_[R]) {
            // all shards arrive here together (same count): nobody may still read the slots that are about to be replaced
            ops.sync_stream(r);
            barrier.wait();
            size_t cap = count > 2 * capacity_[R] ? count : 2 * capacity_[R];
            if (cap < 4096) cap = 4096;
            ops.reserve_slots(r, cap);
            capacity_[R] = cap;
            sequence_[R] = 0;
        }
        const int p = (int)(sequence_[R] & 1);
        // slot generation p was last read by all-reduce (sequence - 2): those reads were enqueued before the barrier of all-reduce
        // (sequence - 1), which this thread has passed
        if (sequence_[R] >= 2)
            for (int q = 0; q < n_; ++q) ops.wait_consumed(r, p, q);
        ops.publish(r, p, buf, count);
        barrier.wait();                                      // every shard has published its slot (its `ready` mark is recorded)
        for (int q = 0; q < n_; ++q)
            if (q != r) ops.wait_ready(r, p, q);
        ops.sum(r, p, buf, count);
        ops.mark_consumed(r, p);
        ++sequence_[R];
    }

    /// Recovery after a failed task, on every shard's thread: drain the own stream, all meet (no stream reads anybody's slot any
    /// more), drop the own slots -- capacity 0, so that the next all-reduce takes the growth path on EVERY shard together, whatever
    /// an interrupted growth left behind -- all meet again.
    void recover(Ops& ops, Barrier& barrier, int r)
    {
        ops.sync_stream(r);
        barrier.wait();
        if (n_ > 0) {
            ops.release_slots(r);
            capacity_[(size_t)r] = 0;
            sequence_[(size_t)r] = 0;
        }
        barrier.wait();
    }

    size_t capacity(int r) const { return capacity_[(size_t)r]; }
    uint64_t sequence(int r) const { return sequence_[(size_t)r]; }

private:
    int n_ = 0;
    std::vector<size_t> capacity_;       // doubles per slot (the same on every shard between tasks)
    std::vector<uint64_t> sequence_;     // all-reduces since the slots were (re)allocated, per shard
};

}  // namespace mlhip_rt
