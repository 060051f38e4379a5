// Stress of the host-thread protocols this library ADDS to the (single-threaded) reference, on the CPU, meant to run under
// -fsanitize=thread and -fsanitize=address,undefined (tests/test_sanitizers.py):
//   * host/team.hpp         -- the per-thread team of the host-side factorizations (what `#pragma omp parallel for` did);
//   * runtime/shard_team.hpp -- Barrier, ShardTeam (one thread per shard of a device group, first-failure reporting, recovery) and
//                              SlotAllreduce, the in-process all-reduce protocol of the shards, here with plain-memory `Ops` that
//                              CHECK the ordering the HIP events provide on a GPU (a slot is read only when its owner has published
//                              it for this all-reduce, overwritten only when every reader of its previous contents is done).
// Scenarios: normal runs, a shard failing at every barrier position (alone, while the others are inside the same all-reduce; also
// in the middle of a slot growth), recover-then-reuse, 64 shards, teams used from several caller threads at once.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "host/team.hpp"
#include "runtime/shard_team.hpp"

using mlhip::host::Team;
using namespace mlhip_rt;

#define CHECK(cond)                                                                     \
    do {                                                                                \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
    } while (0)

// ---- host/team.hpp ------------------------------------------------------------------------------------------------------------
static void team_tests()
{
    for (int threads : {1, 2, 3, 8}) {
        for (int count : {0, 1, 5, 64, 1000}) {
            std::vector<int> hits((size_t)count, 0);
            Team::mine().for_each(count, threads, [&](int i) { hits[(size_t)i] += 1; });
            for (int h : hits) CHECK(h == 1);
        }
    }
    // many short regions back to back (the generation counter), sizes that change (workers beyond a region's size sit it out)
    long total = 0;
    for (int rep = 0; rep < 3000; ++rep) {
        const int count = 1 + rep % 13, threads = 1 + rep % 5;
        std::vector<long> part((size_t)count, 0);
        Team::mine().for_each(count, threads, [&](int i) { part[(size_t)i] = i + 1; });
        total += std::accumulate(part.begin(), part.end(), 0L);
    }
    CHECK(total > 0);
    // an exception in one participant comes out in the caller; the team works afterwards
    bool thrown = false;
    try {
        Team::mine().for_each(40, 4, [&](int i) { if (i == 27) throw std::runtime_error("boom"); });
    } catch (const std::runtime_error& e) {
        thrown = std::string(e.what()) == "boom";
    }
    CHECK(thrown);
    std::atomic<int> n{0};
    Team::mine().for_each(40, 4, [&](int) { n.fetch_add(1); });
    CHECK(n.load() == 40);
    // one team per calling thread: the shards of a device group call the host math from threads of their own
    std::vector<std::thread> callers;
    std::atomic<long> grand{0};
    for (int t = 0; t < 6; ++t)
        callers.emplace_back([&, t] {
            for (int rep = 0; rep < 200; ++rep) {
                std::vector<int> v(32, 0);
                Team::mine().for_each(32, 1 + (t + rep) % 4, [&](int i) { v[(size_t)i] = i; });
                grand.fetch_add(std::accumulate(v.begin(), v.end(), 0L));
            }
        });
    for (auto& c : callers) c.join();
    CHECK(grand.load() == 6L * 200 * (31 * 32 / 2));
}

// ---- SlotAllreduce with checking mock operations ------------------------------------------------------------------------------
struct Injected : std::runtime_error { using std::runtime_error::runtime_error; };

struct MockOps {
    int n;
    std::vector<std::vector<double>> slot[2];        // plain memory: a protocol error is a data race (tsan) or a wrong tag below
    std::vector<uint64_t> published[2], consumed[2]; // per slot: number of the all-reduce that last published it / finished reading
    std::vector<uint64_t> round;                     // per shard: all-reduces since the slots were (re)allocated
    std::atomic<int> fail_reserve_shard{-1};         // reserve_slots of this shard throws once (a growth interrupted half-way)
    explicit MockOps(int n_) : n(n_), round((size_t)n_, 0)
    {
        for (int p = 0; p < 2; ++p) {
            slot[p].resize((size_t)n);
            published[p].assign((size_t)n, 0);
            consumed[p].assign((size_t)n, 0);
        }
    }
    void sync_stream(int) {}
    void reserve_slots(int r, size_t doubles)
    {
        if (r == fail_reserve_shard.load()) { fail_reserve_shard.store(-1); throw Injected("growth failed on one shard"); }
        for (int p = 0; p < 2; ++p) { slot[p][(size_t)r].assign(doubles, -1.0); published[p][(size_t)r] = consumed[p][(size_t)r] = 0; }
        round[(size_t)r] = 0;
    }
    void release_slots(int r)
    {
        for (int p = 0; p < 2; ++p) { slot[p][(size_t)r].clear(); slot[p][(size_t)r].shrink_to_fit(); published[p][(size_t)r] = consumed[p][(size_t)r] = 0; }
        round[(size_t)r] = 0;
    }
    void wait_consumed(int r, int p, int q)
    {
        // shard r is about to overwrite ITS slot of generation p for all-reduce round[r] + 1: every reader q of the previous
        // contents (all-reduce round[r] - 1) must be done
        CHECK(consumed[p][(size_t)q] >= round[(size_t)r] - 1);
    }
    void publish(int r, int p, const double* buf, size_t count)
    {
        CHECK(slot[p][(size_t)r].size() >= count);
        std::copy(buf, buf + count, slot[p][(size_t)r].begin());
        published[p][(size_t)r] = round[(size_t)r] + 1;
    }
    void wait_ready(int r, int p, int q) { CHECK(published[p][(size_t)q] == round[(size_t)r] + 1); }
    void sum(int r, int p, double* buf, size_t count)
    {
        for (size_t i = 0; i < count; ++i) {
            double s = 0;
            for (int q = 0; q < n; ++q) s += slot[p][(size_t)q][i];
            buf[i] = s;
        }
        (void)r;
    }
    void mark_consumed(int r, int p) { consumed[p][(size_t)r] = ++round[(size_t)r]; }
};

/// where a shard is made to fail inside one all-reduce sequence
enum FailAt { kNever, kBeforeFirst, kBetween, kInGrowth };

static void slot_tests(int n, FailAt fail_at, int failing_shard)
{
    ShardTeam team;
    SlotAllreduce<MockOps> exchange;
    MockOps ops(n);
    exchange.init(n);
    std::atomic<int> aborts{0}, recoveries{0};
    team.start(n, [&](int s) { recoveries.fetch_add(1); exchange.recover(ops, team.barrier, s); }, [&](int) { aborts.fetch_add(1); });

    const auto rounds = [&](int s, int first, int last, bool inject) {
        std::vector<double> buf;
        for (int k = first; k < last; ++k) {
            const size_t count = 16 + (size_t)(k % 7) * 700 + (k > 20 ? 9000 : 0);      // grows: the slots are re-allocated on the way
            buf.assign(count, 0.0);
            for (size_t i = 0; i < count; ++i) buf[i] = (double)(s + 1) * (double)(k + 1) + (double)i;
            if (inject && s == failing_shard && ((fail_at == kBeforeFirst && k == first) || (fail_at == kBetween && k == first + 5)))
                throw Injected("one shard failed alone");
            exchange.allreduce(ops, team.barrier, s, buf.data(), count);
            const double shards = (double)n * (n + 1) / 2;
            for (size_t i = 0; i < count; i += 97) CHECK(buf[i] == shards * (double)(k + 1) + (double)n * (double)i);
        }
    };

    if (fail_at == kNever) {
        team.run([&](int s) { rounds(s, 0, 40, false); });
        CHECK(aborts.load() == 0 && recoveries.load() == 0);
    } else {
        if (fail_at == kInGrowth) ops.fail_reserve_shard = failing_shard;
        bool caught = false;
        try {
            team.run([&](int s) { rounds(s, 0, 40, true); });
        } catch (const Injected&) {
            caught = true;                                    // the FIRST failure, not the GroupAborted of the shards torn out
        }
        CHECK(caught && team.dirty() && aborts.load() == 1);
        team.run([&](int s) { rounds(s, 3, 30, false); });    // recover (once per shard), then the protocol from scratch
        CHECK(recoveries.load() == n && !team.dirty());
        for (int s = 1; s < n; ++s) CHECK(exchange.capacity(s) == exchange.capacity(0) && exchange.sequence(s) == exchange.sequence(0));
        team.run([&](int s) { rounds(s, 0, 12, false); });    // ... and again without a recovery in front
        CHECK(recoveries.load() == n);
    }
    team.stop();
}

static void barrier_abort_while_sleeping()
{
    // shards parked in the barrier's condition variable (not spinning any more) are woken by an abort
    ShardTeam team;
    team.start(4, nullptr, nullptr);
    bool caught = false;
    try {
        team.run([&](int s) {
            if (s == 3) {
                std::this_thread::sleep_for(std::chrono::milliseconds(30));
                throw Injected("late failure");
            }
            team.barrier.wait();
        });
    } catch (const Injected&) {
        caught = true;
    }
    CHECK(caught);
    team.run([&](int) { team.barrier.wait(); });
    team.stop();
}

int main()
{
    team_tests();
    barrier_abort_while_sleeping();
    for (int n : {2, 3, 8}) {
        slot_tests(n, kNever, 0);
        for (FailAt f : {kBeforeFirst, kBetween, kInGrowth})
            for (int shard : {0, n / 2, n - 1}) slot_tests(n, f, shard);
    }
    slot_tests(64, kNever, 0);                                // the largest group the library admits (kGroupMaxShards)
    slot_tests(64, kBetween, 17);
    slot_tests(64, kInGrowth, 63);
    std::printf("team stress ok\n");
    return 0;
}
