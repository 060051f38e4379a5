#include "DeviceInit.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <iterator>
#include <limits>
#include <random>
#include <typeinfo>
#include <vector>

#include "ML/Device.hpp"
#include "mlhip.h"

namespace ml {
namespace Clustering {
namespace detail {

namespace {

/// Index range [0, n) as a forward range without storage, for std::sample (selection sampling walks it once).
struct IndexIterator {
    using iterator_category = std::forward_iterator_tag;
    using value_type = Index;
    using difference_type = std::ptrdiff_t;
    using pointer = const Index*;
    using reference = Index;
    Index i = 0;
    Index operator*() const { return i; }
    IndexIterator& operator++() { ++i; return *this; }
    IndexIterator operator++(int) { IndexIterator t = *this; ++i; return t; }
    bool operator==(const IndexIterator& o) const { return i == o.i; }
    bool operator!=(const IndexIterator& o) const { return i != o.i; }
};

/// Row-sharded job: where this rank's rows sit in the whole (rank-ordered) sample.
struct Shard {
    int world = 1, rank = 0;
    Index lo = 0, hi = 0, n_global = 0;
};

Shard locate_shard(mlhip_ctx* ctx, Index n_local)
{
    Shard s;
    device::check(mlhip_ctx_world(ctx, &s.world, &s.rank));
    std::vector<double> counts(static_cast<std::size_t>(s.world), 0.0);
    counts[static_cast<std::size_t>(s.rank)] = static_cast<double>(n_local);
    device::check(mlhip_ctx_allreduce(ctx, counts.data(), counts.size()));
    for (int r = 0; r < s.world; ++r) {
        if (r == s.rank) s.lo = s.n_global;
        s.n_global += static_cast<Index>(counts[static_cast<std::size_t>(r)]);
    }
    s.hi = s.lo + n_local;
    return s;
}

/// Every rank ends with rank `owner`'s values (sum with zeros elsewhere; exact up to the sign of a zero).
void broadcast_from(mlhip_ctx* ctx, int owner, int rank, double* v, std::size_t count)
{
    if (rank != owner) std::fill_n(v, count, 0.0);
    device::check(mlhip_ctx_allreduce(ctx, v, count));
}

/// Sums a d x K block across ranks (through a contiguous copy: a MatrixRef may be strided).
void allreduce_matrix(mlhip_ctx* ctx, MatrixRef m)
{
    const Index d = m.rows(), K = m.cols();
    std::vector<double> flat(static_cast<std::size_t>(d) * static_cast<std::size_t>(K));
    for (Index k = 0; k < K; ++k) std::copy_n(m.col(k), d, flat.data() + static_cast<std::size_t>(k) * d);
    device::check(mlhip_ctx_allreduce(ctx, flat.data(), flat.size()));
    for (Index k = 0; k < K; ++k) std::copy_n(flat.data() + static_cast<std::size_t>(k) * d, d, m.col(k));
}

/// Forgy on the whole sample (ML/Clustering.cpp:16-25): every rank performs the same selection sampling over the global
/// index range with its own (identically seeded) engine, so all agree on the K rows; the owners contribute them.
void forgy_sharded(const Shard& sh, ConstMatrixRef data, std::default_random_engine& prng, unsigned int K, MatrixRef centroids,
                   mlhip_ctx* ctx)
{
    std::vector<Index> chosen;
    std::sample(IndexIterator{0}, IndexIterator{sh.n_global}, std::back_inserter(chosen), K, prng);
    centroids.setZero();
    for (unsigned int k = 0; k < K && k < chosen.size(); ++k)
        if (chosen[k] >= sh.lo && chosen[k] < sh.hi) std::copy_n(data.col(chosen[k] - sh.lo), data.rows(), centroids.col(k));
    allreduce_matrix(ctx, centroids);
}

/// The stable partition of this rank's rows by their cluster draw: order[offsets[k] .. offsets[k+1]) = local rows drawn for
/// cluster k, ascending (a counting sort of `mine`).
void partition_rows(const std::vector<unsigned int>& mine, unsigned int K, std::vector<uint32_t>& order, std::vector<uint32_t>& offsets)
{
    offsets.assign(static_cast<std::size_t>(K) + 1, 0);
    for (unsigned int k : mine) ++offsets[k + 1];
    for (unsigned int k = 0; k < K; ++k) offsets[k + 1] += offsets[k];
    std::vector<uint32_t> next(offsets.begin(), offsets.end() - 1);
    order.resize(mine.size());
    for (std::size_t i = 0; i < mine.size(); ++i) order[next[mine[i]]++] = static_cast<uint32_t>(i);
}

/// RandomPartition on the whole sample (ML/Clustering.cpp:27-37): the per-row cluster draws are a function of the engine
/// alone, so every rank replays all of them and keeps its own rows'; the running means are order dependent, so the
/// ranks update the shared state one after the other, in row order. The O(N d) running means themselves -- K d independent
/// sequential chains -- run on the device-resident rows when there are any (mlhip_random_partition_means: bit-identical
/// to the host loop), on the host otherwise; a single rank is the world-of-one case of the same code.
void random_partition_sharded(const Shard& sh, ConstMatrixRef data, std::default_random_engine& prng, unsigned int K,
                              MatrixRef centroids, mlhip_ctx* ctx, mlhip_data* device_data)
{
    const Index d = data.rows(), n_local = data.cols();
    std::uniform_int_distribution<unsigned int> pick(0, K - 1);
    std::vector<unsigned int> mine(static_cast<std::size_t>(n_local));
    for (Index i = 0; i < sh.n_global; ++i) {
        const unsigned int k = pick(prng);
        if (i >= sh.lo && i < sh.hi) mine[static_cast<std::size_t>(i - sh.lo)] = k;
    }
    std::vector<uint32_t> order, offsets;
    if (device_data) partition_rows(mine, K, order, offsets);
    std::vector<double> state(static_cast<std::size_t>(d) * K + K, 0.0);   // [centroids d x K | sizes K]
    for (int r = 0; r < sh.world; ++r) {
        if (r == sh.rank) {
            double* sizes = state.data() + static_cast<std::size_t>(d) * K;
            if (device_data) {
                device::check(mlhip_random_partition_means(ctx, device_data, K, order.data(), offsets.data(), state.data(), sizes));
            } else {
                for (Index i = 0; i < n_local; ++i) {
                    const unsigned int k = mine[static_cast<std::size_t>(i)];
                    const double count = (sizes[k] += 1.0);
                    double* c = state.data() + static_cast<std::size_t>(d) * k;
                    const double* x = data.col(i);
                    for (Index j = 0; j < d; ++j) c[j] += (x[j] - c[j]) / count;
                }
            }
        }
        if (sh.world > 1) broadcast_from(ctx, r, sh.rank, state.data(), state.size());
    }
    for (unsigned int k = 0; k < K; ++k) std::copy_n(state.data() + static_cast<std::size_t>(d) * k, d, centroids.col(k));
}

/// K-means++ on the whole sample (ML/Clustering.cpp:39-59) with the draw of std::discrete_distribution reproduced over the
/// rank-ordered rows: the weight sum and the cumulative probabilities are sequential floating-point sums, so the ranks
/// take turns in row order, each continuing from its predecessor's carry (K rounds x world short hops); the distance
/// passes run on every rank's device block at once.
void kpp_sharded(const Shard& sh, ConstMatrixRef data, std::default_random_engine& prng, unsigned int K, MatrixRef centroids,
                 mlhip_ctx* ctx, mlhip_data* device_data)
{
    const Index d = data.rows();
    const std::size_t count = static_cast<std::size_t>(data.cols());
    // Large samples: the draws on the devices (mlhip_kpp_draw: certified index, see the single-rank route below); the ranks'
    // weights stay on their devices -- the host copy is only allocated and fetched when a draw has to be settled by the sequential
    // evaluation (mlhip_kpp_weights; ADVICE r3: no 8 N byte host pass per fit otherwise).
    const bool device_draw = sh.n_global >= 32768;
    std::vector<double> weights(device_draw ? 0 : count, 1.0), latest;
    std::vector<double> pick(static_cast<std::size_t>(d));
    // The sequential evaluation over the rank-ordered rows: sum = ((0 + w_0) + w_1) + ..., then the cumulative probabilities up to
    // the drawn one, the ranks taking turns.
    auto sequential_pick = [&](double p) {
        double sum = 0.0;
        for (int r = 0; r < sh.world; ++r) {
            if (r == sh.rank)
                for (std::size_t i = 0; i < count; ++i) sum += weights[i];
            broadcast_from(ctx, r, sh.rank, &sum, 1);
        }
        double carry[2] = {0.0, 0.0};              // [cumulative probability so far, found flag]
        for (int r = 0; r < sh.world; ++r) {
            if (r == sh.rank && carry[1] == 0.0) {
                double cumulative = carry[0];
                for (std::size_t i = 0; i < count; ++i) {
                    if (sh.lo + static_cast<Index>(i) == sh.n_global - 1) {   // the last probability is forced to 1
                        carry[1] = 1.0;
                        std::copy_n(data.col(static_cast<Index>(i)), d, pick.data());
                        break;
                    }
                    cumulative += weights[i] / sum;
                    if (cumulative >= p) {
                        carry[1] = 1.0;
                        std::copy_n(data.col(static_cast<Index>(i)), d, pick.data());
                        break;
                    }
                }
                carry[0] = cumulative;
            }
            broadcast_from(ctx, r, sh.rank, carry, 2);
        }
    };
    for (unsigned int chosen = 0; chosen < K; ++chosen) {
        if (device_draw) {
            std::fill(pick.begin(), pick.end(), 0.0);
            const double p = std::generate_canonical<double, std::numeric_limits<double>::digits>(prng);
            Index global = sh.n_global - 1;
            bool settled = true;
            if (chosen == 0) {
                // all weights 1: sum = N exactly, p_i = fl(1 / N), cp_i = the sequential sum of i + 1 of them (every rank runs it)
                const double q = 1.0 / static_cast<double>(sh.n_global);
                double cumulative = 0.0;
                for (Index i = 0; i + 1 < sh.n_global; ++i) {
                    cumulative += q;
                    if (cumulative >= p) { global = i; break; }
                }
            } else {
                uint64_t index = 0;
                int certain = 0;
                device::check(mlhip_kpp_draw(ctx, device_data, centroids.col(chosen - 1), chosen == 1 ? 1 : 0, p,
                                             static_cast<uint64_t>(sh.lo), &index, &certain, nullptr));
                if (certain) {
                    global = static_cast<Index>(index);
                } else {                                                // (every rank got the same verdict)
                    weights.resize(count);
                    device::check(mlhip_kpp_weights(ctx, device_data, weights.data()));
                    sequential_pick(p);
                    settled = false;
                }
            }
            if (settled && global >= sh.lo && global < sh.hi) std::copy_n(data.col(global - sh.lo), d, pick.data());
            device::check(mlhip_ctx_allreduce(ctx, pick.data(), pick.size()));   // only the owner's copy is non-zero
            std::copy_n(pick.data(), d, centroids.col(chosen));
            continue;
        }
        if (chosen > 0) {
            std::vector<double>& target = chosen == 1 ? weights : latest;
            target.resize(count);
            device::check(mlhip_min_squared_distances(ctx, device_data, 1, centroids.col(chosen - 1), target.data()));
            if (chosen > 1)
                for (std::size_t i = 0; i < count; ++i) weights[i] = std::min(weights[i], latest[i]);
        }
        std::fill(pick.begin(), pick.end(), 0.0);
        if (sh.n_global >= 2) {
            const double p = std::generate_canonical<double, std::numeric_limits<double>::digits>(prng);
            sequential_pick(p);
        } else if (sh.lo == 0 && count > 0) {
            std::copy_n(data.col(0), d, pick.data());   // fewer than two weights: index 0, no draw (bits/random.tcc)
        }
        device::check(mlhip_ctx_allreduce(ctx, pick.data(), pick.size()));   // only the owner's copy is non-zero
        std::copy_n(pick.data(), d, centroids.col(chosen));
    }
}

}  // namespace

void locate_rows(mlhip_ctx* ctx, Index n_local, Index& first_row, Index& n_global)
{
    const Shard sh = locate_shard(ctx, n_local);
    first_row = sh.lo;
    n_global = sh.n_global;
}

void sum_across_ranks(mlhip_ctx* ctx, MatrixRef m) { allreduce_matrix(ctx, m); }

void init_centroids(const CentroidsInitialiser& initialiser, ConstMatrixRef data, std::default_random_engine& prng,
                    const unsigned int number_components, MatrixRef centroids, mlhip_ctx* ctx, mlhip_data* device_data)
{
    int world = 1, rank = 0;
    if (ctx) device::check(mlhip_ctx_world(ctx, &world, &rank));
    if (world > 1) {
        // Row-sharded job: the library initialisers reproduce what a single process would draw on the whole sample
        // (same seed => same centroids, whatever the number of ranks); anything else runs on rank 0's shard and is
        // handed to the others.
        const Shard sh = locate_shard(ctx, data.cols());
        if (typeid(initialiser) == typeid(Forgy)) {
            forgy_sharded(sh, data, prng, number_components, centroids, ctx);
        } else if (typeid(initialiser) == typeid(RandomPartition)) {
            random_partition_sharded(sh, data, prng, number_components, centroids, ctx, device_data);
        } else if (typeid(initialiser) == typeid(KPP) && device_data) {
            kpp_sharded(sh, data, prng, number_components, centroids, ctx, device_data);
        } else {
            initialiser.init(data, prng, number_components, centroids);
            if (rank != 0) centroids.setZero();
            allreduce_matrix(ctx, centroids);
        }
        return;
    }
    if (ctx && device_data && typeid(initialiser) == typeid(RandomPartition)) {
        Shard sh;                                   // the whole sample on one rank
        sh.hi = sh.n_global = data.cols();
        random_partition_sharded(sh, data, prng, number_components, centroids, ctx, device_data);
        return;
    }
    if (!ctx || !device_data || typeid(initialiser) != typeid(KPP)) {
        initialiser.init(data, prng, number_components, centroids);
        return;
    }
    // K-means++ (ML/Clustering.cpp:39-59) with the distance passes on the device. The draw follows what
    // std::discrete_distribution (libstdc++ bits/random.tcc, _M_initialize + operator()) computes for these weights -- the
    // sequential sum, p_i = w_i / sum, sequential cumulative sums, first index whose cumulative probability is >= the
    // canonical uniform draw, last one forced to 1 -- without materialising the two N-long probability vectors: the
    // cumulative scan stops at the drawn index.
    const Index n = data.cols(), d = data.rows();
    const std::size_t count = static_cast<std::size_t>(n);
    // Large samples: the draw itself on the device (mlhip_kpp_draw) -- the running-minimum weights never leave it, and the index
    // is certified against the reference's sequential sums by an error bound; only a draw that falls between two rows closer
    // than that bound (about 8 N^2 2^-53 of them) comes back for the sequential evaluation below. The N-long host passes per
    // centroid were 50x the K-means iterations they prepare at N = 1M, K = 64.
    const bool device_draw = count >= 32768;
    constexpr std::size_t kHostDistanceRows = 4096;
    std::vector<double> weights(device_draw ? 0 : count, 1.0), latest;
    for (unsigned int chosen = 0; chosen < number_components; ++chosen) {
        if (device_draw) {
            std::size_t pick = count - 1;
            const double p = std::generate_canonical<double, std::numeric_limits<double>::digits>(prng);
            if (chosen == 0) {
                // all weights 1: sum = N exactly, p_i = fl(1 / N), cp_i = the sequential sum of i + 1 of them
                const double q = 1.0 / static_cast<double>(count);
                double cumulative = 0.0;
                for (std::size_t i = 0; i + 1 < count; ++i) {
                    cumulative += q;
                    if (cumulative >= p) { pick = i; break; }
                }
            } else {
                uint64_t index = 0;
                int certain = 0;
                device::check(mlhip_kpp_draw(ctx, device_data, centroids.col(chosen - 1), chosen == 1 ? 1 : 0, p, 0, &index, &certain,
                                             nullptr));
                if (certain) {
                    pick = static_cast<std::size_t>(index);
                } else {
                    weights.resize(count);                             // (only now: the weights stay on the device otherwise)
                    device::check(mlhip_kpp_weights(ctx, device_data, weights.data()));
                    double sum = 0.0;
                    for (std::size_t i = 0; i < count; ++i) sum += weights[i];
                    double cumulative = 0.0;
                    for (std::size_t i = 0; i + 1 < count; ++i) {
                        cumulative += weights[i] / sum;
                        if (cumulative >= p) { pick = i; break; }
                    }
                }
            }
            std::copy_n(data.col(static_cast<Index>(pick)), d, centroids.col(chosen));
            continue;
        }
        double sum = 0.0;
        if (chosen == 0) {
            sum = static_cast<double>(count);     // == the sequential sum of `count` ones (exact below 2^53)
        } else {
            // squared distance of every sample to the centroid chosen last; weights = running minimum
            std::vector<double>& target = chosen == 1 ? weights : latest;
            target.resize(count);
            if (count <= kHostDistanceRows) {
                // a few thousand rows: the distances here, from the caller's own array, with the kernels' arithmetic (ascending-j chain
                // s = fma(x_j - c_j, x_j - c_j, s): the same bits, so the same draws) -- a launch, an N-long download and a wait per
                // centroid cost ~50 us, more than the whole step loop of such a fit
                const double* c = centroids.col(chosen - 1);
                for (std::size_t i = 0; i < count; ++i) {
                    const double* x = data.col(static_cast<Index>(i));
                    double s = 0.0;
                    for (Index j = 0; j < d; ++j) {
                        const double t = x[j] - c[j];
                        s = std::fma(t, t, s);
                    }
                    target[i] = s;
                }
            } else {
                device::check(mlhip_min_squared_distances(ctx, device_data, 1, centroids.col(chosen - 1), target.data()));
            }
            if (chosen == 1) {
                for (std::size_t i = 0; i < count; ++i) sum += weights[i];
            } else {
                for (std::size_t i = 0; i < count; ++i) {
                    const double w = std::min(weights[i], latest[i]);
                    weights[i] = w;
                    sum += w;
                }
            }
        }
        std::size_t pick = 0;
        if (count >= 2) {
            const double p = std::generate_canonical<double, std::numeric_limits<double>::digits>(prng);
            double cumulative = 0.0;
            pick = count - 1;
            for (std::size_t i = 0; i + 1 < count; ++i) {
                cumulative += weights[i] / sum;
                if (cumulative >= p) { pick = i; break; }
            }
        }
        std::copy_n(data.col(static_cast<Index>(pick)), d, centroids.col(chosen));
    }
}

}  // namespace detail
}  // namespace Clustering
}  // namespace ml
