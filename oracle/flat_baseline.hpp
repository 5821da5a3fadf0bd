// flat_baseline.hpp -- the "optimised CPU" baseline SURVEY.md 8(d) asks for next to the faithful-cost port:
// the SAME search (one-hop-per-layer descent, ef-bounded search_layer with two binary heaps, src/hnsw.rs:1487-1580,
// :922-1064) on FLAT arrays -- contiguous vectors, CSR neighbour lists, an epoch-stamped visited array per thread, no
// Arc / RwLock / per-edge allocation -- with 8-lane SIMD-order distances (the crate's `simdeez_f` build order) and
// software prefetch of the rows about to be evaluated.  It is what a CPU implementation free of the reference's data
// model could do on the host cores; it is NOT the reference's cost structure.
//
// TEST INFRASTRUCTURE / bench.py's cpu_baseline leg only (timing and recall sanity).  Its sums differ from the scalar
// reference order in the last bits, so it is never used for a parity check.
#pragma once
#include "pinning.hpp"
#include <atomic>
#include <cstdint>
#include <cstring>
#include <queue>
#include <thread>
#include <vector>

#include "hnsw_oracle.hpp"

namespace oracle {

struct FlatBaseline {
    DistKind dist;
    size_t n = 0, d = 0, stride = 0;
    std::vector<float> vec;                    // [n][stride], rows padded to 16 floats (64-byte lines)
    std::vector<uint64_t> origin;              // [n]
    std::vector<uint32_t> ptr0, ids0;          // CSR of the search layer
    std::vector<std::vector<uint32_t>> ptr_up, ids_up;  // CSR per upper layer (index = layer - 1)
    uint32_t entry = 0;
    int entry_level = 0;

    explicit FlatBaseline(const Hnsw& h) : dist(h.dist), d(h.data_dimension) {
        stride = (d + 15) / 16 * 16;
        std::vector<size_t> offset(NB_LAYER_MAX + 1, 0);
        for (size_t l = 0; l < NB_LAYER_MAX; ++l) offset[l + 1] = offset[l] + h.get_layer_nb_point(l);
        n = offset[NB_LAYER_MAX];
        auto flat = [&](const PointId& p) { return (uint32_t)(offset[p.layer] + (size_t)p.rank); };
        vec.assign(n * stride, 0.f);
        origin.resize(n);
        size_t search_layer = 0;
        while (search_layer < NB_LAYER_MAX && h.get_layer_nb_point(search_layer) == 0) ++search_layer;
        ptr0.assign(n + 1, 0);
        ptr_up.assign(NB_LAYER_MAX - 1, std::vector<uint32_t>(n + 1, 0));
        ids_up.assign(NB_LAYER_MAX - 1, {});
        for (size_t l = 0; l < h.points_by_layer.size(); ++l)
            for (const auto& p : h.points_by_layer[l]) {
                const uint32_t f = flat(p->p_id);
                std::memcpy(vec.data() + (size_t)f * stride, p->v.data(), d * sizeof(float));
                origin[f] = p->origin_id;
            }
        // CSR in flat-id order
        std::vector<const Point*> by_flat(n, nullptr);
        for (size_t l = 0; l < h.points_by_layer.size(); ++l)
            for (const auto& p : h.points_by_layer[l]) by_flat[flat(p->p_id)] = p.get();
        for (size_t f = 0; f < n; ++f) {
            const Point* p = by_flat[f];
            ptr0[f] = (uint32_t)ids0.size();
            for (const PWO& e : p->neighbours[search_layer]) ids0.push_back(flat(e->point_ref->p_id));
            for (size_t l = 1; l < NB_LAYER_MAX; ++l) {
                ptr_up[l - 1][f] = (uint32_t)ids_up[l - 1].size();
                for (const PWO& e : p->neighbours[l]) ids_up[l - 1].push_back(flat(e->point_ref->p_id));
            }
        }
        ptr0[n] = (uint32_t)ids0.size();
        for (size_t l = 1; l < NB_LAYER_MAX; ++l) ptr_up[l - 1][n] = (uint32_t)ids_up[l - 1].size();
        if (h.entry_point) {
            entry = flat(h.entry_point->p_id);
            entry_level = h.entry_point->p_id.layer;
        }
    }

    float eval(const float* q, uint32_t id) const { return dist_simd8(dist, q, vec.data() + (size_t)id * stride, d); }

    struct Scratch {
        std::vector<uint32_t> stamp;
        uint32_t epoch = 0;
    };
    using Ent = std::pair<float, uint32_t>;

    // returns the number of answers written (ascending distance)
    size_t search(const float* q, size_t k, size_t ef_arg, Scratch& s, uint64_t* out_ids, float* out_dists) const {
        if (n == 0) return 0;
        if (s.stamp.size() < n) s.stamp.assign(n, 0);
        if (++s.epoch == 0) { std::fill(s.stamp.begin(), s.stamp.end(), 0); s.epoch = 1; }
        uint32_t pivot = entry;
        float dcur = eval(q, pivot);
        for (int layer = entry_level; layer >= 1; --layer) {
            const auto& P = ptr_up[layer - 1];
            const auto& I = ids_up[layer - 1];
            uint32_t best = pivot;
            for (uint32_t j = P[pivot]; j < P[pivot + 1]; ++j) {
                const float t = eval(q, I[j]);
                if (t < dcur) { dcur = t; best = I[j]; }
            }
            pivot = best;
        }
        const size_t ef = std::max(ef_arg, k);
        std::priority_queue<Ent> R;                                            // max-heap on distance
        std::priority_queue<Ent, std::vector<Ent>, std::greater<Ent>> C;       // min-heap on distance
        s.stamp[pivot] = s.epoch;
        R.push({dcur, pivot});
        C.push({dcur, pivot});
        while (!C.empty()) {
            const Ent c = C.top();
            C.pop();
            if (c.first > R.top().first) break;
            const uint32_t b = ptr0[c.second], e = ptr0[c.second + 1];
            for (uint32_t j = b; j < e; ++j) __builtin_prefetch(vec.data() + (size_t)ids0[j] * stride);
            for (uint32_t j = b; j < e; ++j) {
                const uint32_t id = ids0[j];
                if (s.stamp[id] == s.epoch) continue;
                s.stamp[id] = s.epoch;
                const float t = eval(q, id);
                if (t < R.top().first || R.size() < ef) {
                    C.push({t, id});
                    R.push({t, id});
                    if (R.size() > ef) R.pop();
                }
            }
        }
        std::vector<Ent> all;
        all.reserve(R.size());
        while (!R.empty()) { all.push_back(R.top()); R.pop(); }
        const size_t cnt = std::min(k, all.size());
        for (size_t i = 0; i < cnt; ++i) {
            const Ent& e = all[all.size() - 1 - i];
            out_ids[i] = origin[e.second];
            out_dists[i] = e.first;
        }
        return cnt;
    }

    void parallel_search(const float* queries, size_t nq, size_t k, size_t ef, int nthreads, uint64_t* out_ids, float* out_dists,
                         uint32_t* out_counts) const {
        if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
        if (nthreads < 1) nthreads = 1;
        std::atomic<size_t> next{0};
        auto worker = [&](int t) {
            oracle_pin::Pin on_cpu(t);
            Scratch s;
            for (;;) {
                const size_t i0 = next.fetch_add(16);
                if (i0 >= nq) break;
                for (size_t i = i0; i < std::min(nq, i0 + 16); ++i)
                    out_counts[i] = (uint32_t)search(queries + i * d, k, ef, s, out_ids + i * k, out_dists + i * k);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
        worker(0);
        for (auto& t : th) t.join();
    }
};

}  // namespace oracle
