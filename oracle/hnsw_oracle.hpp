// hnsw_oracle.hpp -- CPU ORACLE for the batched-search hot path of hnsw_rs 0.3.4.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
// and bench.py's `cpu_baseline` leg may build, load or call it.  The shipped library
// (hnswlib-rs_amd/csrc -> libhnsw_mi355x.so) never links or falls back to it.
//
// PARITY STATUS: "parity unpinned" by the reference.  The reference is Rust (no
// toolchain in this image), its tests hold no golden vectors (every data set comes
// from an unseeded rand::rng()), and the distance arithmetic lives in
// the un-vendored third-party crate `anndists` (Cargo.toml:90, requirement "0.1", no
// Cargo.lock).  This file is therefore a from-scratch, line-by-line restatement of the
// cited functions; what pins it is listed in DESIGN.md ("Oracle pinning").
//
// Deliberately literal: one heap allocation per vector, shared_ptr (= Arc) per edge,
// hash-map visited set, binary heaps of boxed entries, scalar left-to-right f32 sums --
// the reference's cost structure, so the same code doubles as the "cpu-faithful" baseline
// of BASELINE.md section 2.
//
// Reference citations are relative to /root/reference.
#pragma once
#include "pinning.hpp"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "ref_logf.hpp"

namespace oracle {

constexpr uint8_t NB_LAYER_MAX = 16;  // src/hnsw.rs:42

// ---------------------------------------------------------------------------------------
// Distances.  Third-party crate `anndists` 0.1 (NOT under /root/reference).  Published
// algorithm restated from the crate's dist/distances.rs, default feature set (scalar; the
// crate's `simdeez_f` / `stdsimd` features reorder the sums).  Call sites in the reference:
// src/hnsw.rs:952, :1026, :1112, :1146, :1359, :1374, :1506, :1518.
// Must be compiled with -ffp-contract=off and without -ffast-math (Rust never contracts).
// ---------------------------------------------------------------------------------------
enum DistKind : int { DIST_L2 = 0, DIST_COSINE = 1, DIST_DOT = 2, DIST_L1 = 3, DIST_HELLINGER = 4, DIST_JEFFREYS = 5, DIST_JENSENSHANNON = 6 };

// DistL2 on f32: norm = sum_i (a_i-b_i)*(a_i-b_i) accumulated left to right in f32
// (Iterator::sum), then sqrt.  A true metric (not squared).
inline float dist_l2(const float* a, const float* b, size_t d) {
    float norm = 0.f;
    for (size_t i = 0; i < d; ++i) {
        float t = a[i] - b[i];
        norm = norm + t * t;
    }
    return std::sqrt(norm);
}
// DistL1 on f32: sum_i |a_i-b_i| left to right in f32.
inline float dist_l1(const float* a, const float* b, size_t d) {
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) s = s + std::fabs(a[i] - b[i]);
    return s;
}
// DistCosine on f32: the three products are formed in f32, widened to f64 and accumulated
// in three f64 sums (fold, left to right); if both norms > 0:
// 1 - dot/sqrt(na*nb) in f64, clamped at 0, cast to f32; else 0.
inline float dist_cosine(const float* a, const float* b, size_t d) {
    double s0 = 0., s1 = 0., s2 = 0.;
    for (size_t i = 0; i < d; ++i) {
        float ab = a[i] * b[i], aa = a[i] * a[i], bb = b[i] * b[i];
        s0 = s0 + (double)ab;
        s1 = s1 + (double)aa;
        s2 = s2 + (double)bb;
    }
    if (s1 > 0. && s2 > 0.) {
        double du = 1. - s0 / std::sqrt(s1 * s2);
        if (!(du >= -0.00002)) throw std::runtime_error("DistCosine: assert dist_unchecked >= -2e-5");
        return (float)std::max(du, 0.);
    }
    return 0.f;
}
// DistDot on f32 (inputs are expected L2-normalised): 1 - sum_i a_i*b_i (f32, left to
// right), asserted >= ~0 and clamped at 0.
inline float dist_dot(const float* a, const float* b, size_t d) {
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) s = s + a[i] * b[i];
    float dot = 1.f - s;
    return std::max(dot, 0.f);
}
// The three distances between probability vectors of the crate's f32 FFI (src/libext.rs:334-345, :491-513), recalled like
// the others from anndists 0.1 src/dist/distances.rs (oracle/PIN.md):
// DistHellinger: sum_i (sqrt(a_i) * sqrt(b_i)) left to right in f32, then sqrt(max(1 - sum, 0)).
inline float dist_hellinger(const float* a, const float* b, size_t d) {
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) s = s + std::sqrt(a[i]) * std::sqrt(b[i]);
    if (!(1.f - s >= -0.000001f)) throw std::runtime_error("DistHellinger: assert 1 - dist >= -1e-6");
    return std::sqrt(std::max(1.f - s, 0.f));
}
// DistJeffreys: sum_i (a_i - b_i) * ln(max(a_i, M_MIN) / max(b_i, M_MIN)), M_MIN = 1e-30, f32 left to right.
inline float dist_jeffreys(const float* a, const float* b, size_t d) {
    const float M_MIN = 1.0e-30f;
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) s = s + (a[i] - b[i]) * ref_logf(std::max(a[i], M_MIN) / std::max(b[i], M_MIN));
    return s;
}
// DistJensenShannon: for each i, mean = 0.5 * (a_i + b_i); if a_i > 0: dist += a_i * ln(a_i / mean); if b_i > 0:
// dist += b_i * ln(b_i / mean) (f32, in that order); result sqrt(0.5 * dist).
inline float dist_jensenshannon(const float* a, const float* b, size_t d) {
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) {
        const float mean_ab = 0.5f * (a[i] + b[i]);
        if (a[i] > 0.f) s = s + a[i] * ref_logf(a[i] / mean_ab);
        if (b[i] > 0.f) s = s + b[i] * ref_logf(b[i] / mean_ab);
    }
    return std::sqrt(0.5f * s);
}
inline float dist_eval(DistKind k, const float* a, const float* b, size_t d) {
    switch (k) {
        case DIST_HELLINGER: return dist_hellinger(a, b, d);
        case DIST_JEFFREYS: return dist_jeffreys(a, b, d);
        case DIST_JENSENSHANNON: return dist_jensenshannon(a, b, d);
        case DIST_L2: return dist_l2(a, b, d);
        case DIST_COSINE: return dist_cosine(a, b, d);
        case DIST_DOT: return dist_dot(a, b, d);
        case DIST_L1: return dist_l1(a, b, d);
    }
    return NAN;
}
// The crate's `simdeez_f` feature (the "SIMD CPU path" of the reference's benchmarks) sums 8 f32 lanes
// vertically and adds them horizontally at the end, then the scalar tail.  Restated with GCC vector extensions
// (AVX2 when the host has it).  The sums differ from the scalar build in the last bits, so this variant is
// used ONLY to time a SIMD CPU baseline (bench.py); every parity check runs the scalar functions above.
typedef float v8f_t __attribute__((vector_size(32), aligned(4)));
__attribute__((target_clones("avx2,fma", "default"))) inline float dist_simd8(DistKind k, const float* a, const float* b, size_t d) {
    if (k >= DIST_HELLINGER) return dist_eval(k, a, b, d);  // (no SIMD-order variant of the probability distances)
    v8f_t acc = {0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc, acc2 = acc;
    size_t i = 0;
    for (; i + 8 <= d; i += 8) {
        v8f_t x, y;
        std::memcpy(&x, a + i, 32);
        std::memcpy(&y, b + i, 32);
        if (k == DIST_L2) { const v8f_t t = x - y; acc = acc + t * t; }
        else if (k == DIST_L1) { const v8f_t t = x - y; acc = acc + (t < 0 ? -t : t); }
        else { acc = acc + x * y; if (k == DIST_COSINE) { acc1 = acc1 + x * x; acc2 = acc2 + y * y; } }
    }
    float s = 0.f, s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < 8; ++j) { s += acc[j]; s1 += acc1[j]; s2 += acc2[j]; }
    for (; i < d; ++i) {
        if (k == DIST_L2) { const float t = a[i] - b[i]; s += t * t; }
        else if (k == DIST_L1) s += std::fabs(a[i] - b[i]);
        else { s += a[i] * b[i]; if (k == DIST_COSINE) { s1 += a[i] * a[i]; s2 += b[i] * b[i]; } }
    }
    switch (k) {
        case DIST_L2: return std::sqrt(s);
        case DIST_L1: return s;
        case DIST_DOT: return std::max(1.f - s, 0.f);
        default: return s1 > 0.f && s2 > 0.f ? std::max(1.f - s / std::sqrt(s1 * s2), 0.f) : 0.f;
    }
}
// anndists::dist::distances::l2_normalize: divide by sqrt(sum x^2) (f32).
inline void l2_normalize(float* v, size_t d) {
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) s = s + v[i] * v[i];
    float n = std::sqrt(s);
    if (n > 0.f)
        for (size_t i = 0; i < d; ++i) v[i] = v[i] / n;
}
inline const char* dist_type_name(DistKind k) {
    switch (k) {  // std::any::type_name::<D>() as dumped (src/hnsw.rs:839-841)
        case DIST_L2: return "anndists::dist::distances::DistL2";
        case DIST_COSINE: return "anndists::dist::distances::DistCosine";
        case DIST_DOT: return "anndists::dist::distances::DistDot";
        case DIST_L1: return "anndists::dist::distances::DistL1";
        case DIST_HELLINGER: return "anndists::dist::distances::DistHellinger";
        case DIST_JEFFREYS: return "anndists::dist::distances::DistJeffreys";
        case DIST_JENSENSHANNON: return "anndists::dist::distances::DistJensenShannon";
    }
    return "";
}

// ---------------------------------------------------------------------------------------
// Data model (src/hnsw.rs:46, :98-107, :164-173, :265-306)
// ---------------------------------------------------------------------------------------
struct PointId {
    uint8_t layer = 0;
    int32_t rank = 0;
    bool operator==(const PointId& o) const { return layer == o.layer && rank == o.rank; }
    bool operator!=(const PointId& o) const { return !(*this == o); }
};
struct PointIdHash {
    size_t operator()(const PointId& p) const {
        return std::hash<uint64_t>()(((uint64_t)p.layer << 32) | (uint32_t)p.rank);
    }
};
struct Neighbour {  // #[repr(C)] src/hnsw.rs:98-107
    size_t d_id = 0;
    float distance = 0.f;
    PointId p_id;
};

struct PointWithOrder;
struct Point {
    std::vector<float> v;  // own heap allocation, like PointData::V(Vec<T>)
    size_t origin_id;
    PointId p_id;
    // neighbours[l] for l in 0..16 (src/hnsw.rs:177-181)
    std::vector<std::vector<std::shared_ptr<PointWithOrder>>> neighbours;
    Point(const float* data, size_t d, size_t oid, PointId pid)
        : v(data, data + d), origin_id(oid), p_id(pid), neighbours(NB_LAYER_MAX) {}
};
struct PointWithOrder {
    std::shared_ptr<Point> point_ref;
    float dist_to_ref;
    PointWithOrder(const std::shared_ptr<Point>& p, float d) : point_ref(p), dist_to_ref(d) {}
};
using PWO = std::shared_ptr<PointWithOrder>;

// Ord for PointWithOrder: by dist_to_ref only; NaN panics (src/hnsw.rs:283-297).
inline void check_nan(float a, float b) {
    if (std::isnan(a) || std::isnan(b)) throw std::runtime_error("got a NaN in a distance");
}
inline bool pwo_le(const PWO& a, const PWO& b) { check_nan(a->dist_to_ref, b->dist_to_ref); return a->dist_to_ref <= b->dist_to_ref; }
inline bool pwo_lt(const PWO& a, const PWO& b) { check_nan(a->dist_to_ref, b->dist_to_ref); return a->dist_to_ref < b->dist_to_ref; }
inline bool pwo_ge(const PWO& a, const PWO& b) { check_nan(a->dist_to_ref, b->dist_to_ref); return a->dist_to_ref >= b->dist_to_ref; }

// ---------------------------------------------------------------------------------------
// std::collections::BinaryHeap<Arc<PointWithOrder>> restated (Rust std, alloc::collections::
// binary_heap: push/sift_up, pop/sift_down_to_bottom, into_sorted_vec/sift_down_range).
// A max-heap on dist_to_ref; tie behaviour follows the std algorithm literally
// (SURVEY.md Appendix C).  Used at src/hnsw.rs:940, :958-967, :971-973, :1035-1053, :1544.
// ---------------------------------------------------------------------------------------
class RustBinaryHeap {
public:
    std::vector<PWO> data;
    size_t len() const { return data.size(); }
    bool is_empty() const { return data.empty(); }
    const PWO* peek() const { return data.empty() ? nullptr : &data[0]; }
    void push(PWO item) {
        size_t old_len = data.size();
        data.push_back(std::move(item));
        sift_up(0, old_len);
    }
    bool pop(PWO& out) {
        if (data.empty()) return false;
        PWO item = std::move(data.back());
        data.pop_back();
        if (!data.empty()) {
            std::swap(item, data[0]);
            sift_down_to_bottom(0);
        }
        out = std::move(item);
        return true;
    }
    void clear() { data.clear(); }
    // BinaryHeap::retain (std, >= 1.70, recalled): Vec::retain keeps the survivors in order, remembering the index of
    // the first element removed; the heap property is then restored for the tail only (rebuild_tail).
    template <class F>
    void retain(F keep) {
        size_t rebuild_from = data.size(), i = 0, w = 0;
        for (size_t r = 0; r < data.size(); ++r, ++i) {
            if (keep(data[r])) {
                if (w != r) data[w] = std::move(data[r]);
                ++w;
            } else if (i < rebuild_from) {
                rebuild_from = i;
            }
        }
        data.resize(w);
        rebuild_tail(rebuild_from);
    }
    std::vector<PWO> into_sorted_vec() {
        size_t end = data.size();
        while (end > 1) {
            end -= 1;
            std::swap(data[0], data[end]);
            sift_down_range(0, end);
        }
        return std::move(data);
    }
private:
    // while hole > start: if element <= parent break; move parent down
    size_t sift_up(size_t start, size_t pos) {
        PWO elt = std::move(data[pos]);
        while (pos > start) {
            size_t parent = (pos - 1) / 2;
            if (pwo_le(elt, data[parent])) break;
            data[pos] = std::move(data[parent]);
            pos = parent;
        }
        data[pos] = std::move(elt);
        return pos;
    }
    void sift_down_range(size_t pos, size_t end) {
        PWO elt = std::move(data[pos]);
        size_t child = 2 * pos + 1;
        size_t lim = end >= 2 ? end - 2 : 0;  // end.saturating_sub(2)
        while (child <= lim && end >= 2) {
            // NB: with end < 2 Rust's `child <= end.saturating_sub(2)` is `1 <= 0` = false
            child += pwo_le(data[child], data[child + 1]) ? 1 : 0;
            if (pwo_ge(elt, data[child])) {
                data[pos] = std::move(elt);
                return;
            }
            data[pos] = std::move(data[child]);
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1 && pwo_lt(elt, data[child])) {
            data[pos] = std::move(data[child]);
            pos = child;
        }
        data[pos] = std::move(elt);
    }
    // rebuild_tail(start): elements [0, start) still form a heap.  Either heapify everything (rebuild) or sift the
    // tail elements up one by one, by std's cost estimate.
    void rebuild_tail(size_t start) {
        if (start >= data.size()) return;
        const size_t len = data.size(), tail_len = len - start;
        auto log2_fast = [](size_t x) { size_t l = 0; while (x >>= 1) ++l; return l; };
        bool better_to_rebuild;
        if (start < tail_len) better_to_rebuild = true;
        else if (len <= 2048) better_to_rebuild = 2 * len < tail_len * log2_fast(start);
        else better_to_rebuild = 2 * len < tail_len * 11;
        if (better_to_rebuild) {
            size_t n = len / 2;        // rebuild(): sift_down every internal node, last first
            while (n > 0) {
                n -= 1;
                sift_down_range(n, len);
            }
        } else {
            for (size_t i = start; i < len; ++i) sift_up(0, i);
        }
    }
    void sift_down_to_bottom(size_t pos) {
        size_t end = data.size();
        size_t start = pos;
        PWO elt = std::move(data[pos]);
        size_t child = 2 * pos + 1;
        size_t lim = end >= 2 ? end - 2 : 0;
        while (child <= lim && end >= 2) {
            child += pwo_le(data[child], data[child + 1]) ? 1 : 0;
            data[pos] = std::move(data[child]);
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) {
            data[pos] = std::move(data[child]);
            pos = child;
        }
        data[pos] = std::move(elt);
        sift_up(start, pos);
    }
};

// ---------------------------------------------------------------------------------------
// Level generator (src/hnsw.rs:317-386).  The reference draws from rand::StdRng seeded from
// Xoshiro256++(397); that stream cannot be reproduced here (rand crate absent).  Only the
// LAW is restated: level = floor(-ln(U) * scale), U ~ U[0,1), scale = level_scale_factor/ln(M);
// a level >= maxlevel is redrawn uniformly in [0, maxlevel).  U comes from a documented
// SplitMix64 stream seeded with 397 -- the product builder implements the same stream so
// serially built graphs can be compared edge for edge.
// ---------------------------------------------------------------------------------------
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double next_f64() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};
struct LayerGenerator {
    SplitMix64 rng{397};
    double scale;
    size_t maxlevel;
    LayerGenerator(size_t max_nb_connection, double scale_factor, size_t maxlevel_)
        : scale(scale_factor / std::log((double)max_nb_connection)), maxlevel(maxlevel_) {}
    size_t generate() {  // src/hnsw.rs:363-374
        double xsi = rng.next_f64();
        double level = -std::log(xsi) * scale;
        double fl = std::floor(level);
        size_t ulevel = (fl >= (double)maxlevel || !(fl == fl)) ? maxlevel : (size_t)fl;
        if (ulevel >= maxlevel) ulevel = (size_t)(rng.next() % (uint64_t)maxlevel);
        return ulevel;
    }
};

// Work counters: they define the "algorithmic bytes" of SURVEY.md section 8(d).
struct Counters {
    uint64_t n_dist = 0;      // distance evaluations
    uint64_t n_expand = 0;    // candidates expanded (neighbour lists read)
    uint64_t n_ids_read = 0;  // neighbour ids read from lists
    void add(const Counters& o) { n_dist += o.n_dist; n_expand += o.n_expand; n_ids_read += o.n_ids_read; }
};

// ---------------------------------------------------------------------------------------
// Hnsw (src/hnsw.rs:739-763) + PointIndexation (src/hnsw.rs:395-408)
// ---------------------------------------------------------------------------------------
class Hnsw {
public:
    size_t ef_construction;
    size_t max_nb_connection;
    bool extend_candidates = false;
    bool keep_pruned = false;
    size_t max_layer;  // min(16, arg) src/hnsw.rs:778
    size_t data_dimension = 0;
    DistKind dist;
    double level_scale_factor = 1.0;
    // PointIndexation
    std::vector<std::vector<std::shared_ptr<Point>>> points_by_layer;
    LayerGenerator layer_g;
    size_t nb_point = 0;
    std::shared_ptr<Point> entry_point;

    Hnsw(size_t max_nb_conn, size_t /*max_elements*/, size_t max_layer_arg, size_t ef_c, DistKind dk)
        : ef_construction(ef_c), max_nb_connection(max_nb_conn),
          max_layer(std::min<size_t>(NB_LAYER_MAX, max_layer_arg)), dist(dk),
          points_by_layer(std::min<size_t>(NB_LAYER_MAX, max_layer_arg)),
          layer_g(max_nb_conn, 1.0, std::min<size_t>(NB_LAYER_MAX, max_layer_arg)) {
        if (max_nb_conn > 256) throw std::runtime_error("error max_nb_connection must be less equal than 256");
    }
    ~Hnsw() {  // break Arc cycles like Drop for PointIndexation (src/hnsw.rs:413-449)
        for (auto& layer : points_by_layer)
            for (auto& p : layer) p->neighbours.clear();
    }
    Hnsw(const Hnsw&) = delete;
    Hnsw& operator=(const Hnsw&) = delete;

    void modify_level_scale(double f) {  // src/hnsw.rs:876-905
        f = std::min(1.0, std::max(0.2, f));
        level_scale_factor *= f;
        layer_g.scale *= f;
    }
    size_t get_layer_nb_point(size_t layer) const {  // src/hnsw.rs:565-572
        return layer < points_by_layer.size() ? points_by_layer[layer].size() : 0;
    }
    uint8_t get_max_level_observed() const { return entry_point ? entry_point->p_id.layer : 0; }

    bool simd_order = false;  // timing-only variant: the crate's SIMD summation order (dist_simd8)
    float eval(const float* a, const float* b, Counters* c) const {
        if (c) c->n_dist++;
        if (simd_order) return dist_simd8(dist, a, b, data_dimension);
        return dist_eval(dist, a, b, data_dimension);
    }

    // FilterT for Vec<usize> (src/filter.rs:11-15): binary search in a sorted vector of allowed origin ids
    using Filter = std::vector<size_t>;
    static bool hnsw_filter(const Filter& f, size_t id) { return std::binary_search(f.begin(), f.end(), id); }

    // ---- search_layer (src/hnsw.rs:922-1064); filter == nullptr is the unfiltered branch --
    RustBinaryHeap search_layer(const float* point, std::shared_ptr<Point> entry, size_t ef,
                                uint8_t layer, Counters* cnt, const Filter* filter = nullptr) const {
        RustBinaryHeap return_points;                                      // :940
        if (points_by_layer[layer].empty()) return return_points;          // :942-946
        if (entry->p_id.rank < 0) return return_points;                    // :947-950
        float dist_to_entry_point = eval(point, entry->v.data(), cnt);     // :952
        std::unordered_map<PointId, std::shared_ptr<Point>, PointIdHash> visited;  // :955
        visited.emplace(entry->p_id, entry);                               // :956
        RustBinaryHeap candidate_points;                                   // :958
        candidate_points.push(std::make_shared<PointWithOrder>(entry, -dist_to_entry_point));
        return_points.push(std::make_shared<PointWithOrder>(entry, dist_to_entry_point));
        while (!candidate_points.is_empty()) {                             // :969
            PWO c;
            candidate_points.pop(c);                                       // :971
            if (!return_points.peek())                                     // :973 `peek().unwrap()`: with a filter that emptied
                throw std::runtime_error("search_layer: return_points is empty (the reference panics here)");
            const PWO& f = *return_points.peek();
            if (!(f->dist_to_ref >= 0.f)) throw std::runtime_error("assert f.dist >= 0");
            if (!(c->dist_to_ref <= 0.f)) throw std::runtime_error("assert c.dist <= 0");
            if (-(c->dist_to_ref) > f->dist_to_ref) {                      // :981
                if (!filter) return return_points;                         // :992-993
                if (return_points.len() >= ef)                             // :994-1000: drop what the filter refuses,
                    return_points.retain([&](const PWO& p) { return hnsw_filter(*filter, p->point_ref->origin_id); });
            }                                                              // ... and carry on (no return with a filter)
            const auto& neighbours_c_l = c->point_ref->neighbours[layer];  // :1006
            if (cnt) { cnt->n_expand++; cnt->n_ids_read += neighbours_c_l.size(); }
            for (const PWO& e : neighbours_c_l) {                          // :1013
                const PointId epid = e->point_ref->p_id;
                if (visited.find(epid) == visited.end()) {                 // :1016
                    visited.emplace(epid, e->point_ref);                   // :1017
                    const PWO* f_opt = return_points.peek();               // :1019
                    if (!f_opt) return return_points;                      // :1020-1024
                    float e_dist_to_p = eval(point, e->point_ref->v.data(), cnt);  // :1026
                    float f_dist_to_p = (*f_opt)->dist_to_ref;
                    if (e_dist_to_p < f_dist_to_p || return_points.len() < ef) {   // :1028
                        auto e_prime = std::make_shared<PointWithOrder>(e->point_ref, e_dist_to_p);
                        candidate_points.push(std::make_shared<PointWithOrder>(e->point_ref, -e_dist_to_p));
                        if (!filter) {
                            return_points.push(e_prime);                   // :1038
                        } else if (hnsw_filter(*filter, e_prime->point_ref->origin_id)) {  // :1040-1049
                            if (return_points.len() == 1) {
                                const size_t only_id = (*return_points.peek())->point_ref->origin_id;
                                if (!hnsw_filter(*filter, only_id)) return_points.clear();
                            }
                            return_points.push(e_prime);
                        }
                        if (return_points.len() > ef) {                    // :1051-1053
                            PWO dropped;
                            return_points.pop(dropped);
                        }
                    }
                }
            }
        }
        return return_points;                                              // :1063
    }

    // ---- search_filter (src/hnsw.rs:1487-1580); filter == nullptr is Hnsw::search ---------
    std::vector<Neighbour> search(const float* data, size_t knbn, size_t ef_arg, Counters* cnt = nullptr,
                                  const Filter* filter = nullptr) const {
        if (!entry_point) return {};                                       // :1498-1503
        float dist_to_entry = eval(data, entry_point->v.data(), cnt);      // :1506
        std::shared_ptr<Point> pivot = entry_point;
        std::shared_ptr<Point> new_pivot;
        for (int layer = entry_point->p_id.layer; layer >= 1; --layer) {   // :1511
            bool has_changed = false;
            const auto& neighbours = pivot->neighbours[layer];             // :1515
            if (cnt) { cnt->n_expand++; cnt->n_ids_read += neighbours.size(); }
            for (const PWO& n : neighbours) {
                float tmp_dist = eval(data, n->point_ref->v.data(), cnt);  // :1518
                if (tmp_dist < dist_to_entry) {                            // :1519
                    new_pivot = n->point_ref;
                    has_changed = true;
                    dist_to_entry = tmp_dist;
                }
            }
            if (has_changed) pivot = new_pivot;                            // :1526-1528
        }
        size_t ef = std::max(ef_arg, knbn);                                // :1531
        uint8_t l = 0;                                                     // :1534-1540
        while (get_layer_nb_point(l) == 0) l++;
        RustBinaryHeap heap = search_layer(data, pivot, ef, l, cnt, filter);  // :1542
        std::vector<PWO> neighbours = heap.into_sorted_vec();              // :1544
        size_t last = std::min(std::min(knbn, ef), neighbours.size());     // :1547
        std::vector<Neighbour> out;
        out.reserve(last);
        for (size_t i = 0; i < last; ++i) {                                // :1549-1578
            if (filter && !hnsw_filter(*filter, neighbours[i]->point_ref->origin_id)) continue;  // filter_map :1551-1563
            out.push_back(Neighbour{neighbours[i]->point_ref->origin_id, neighbours[i]->dist_to_ref,
                                    neighbours[i]->point_ref->p_id});
        }
        return out;
    }

    // ---- parallel_search (src/hnsw.rs:1612-1635): per-item search, answers in input order.
    // Rayon's pool is restated as nthreads workers pulling request indices from an atomic
    // counter (work-stealing has no observable effect on results).
    std::vector<std::vector<Neighbour>> parallel_search(const std::vector<std::vector<float>>& datas,
                                                        size_t knbn, size_t ef, int nthreads,
                                                        Counters* total = nullptr) const {
        size_t nq = datas.size();
        std::vector<std::vector<Neighbour>> answers(nq);
        if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
        if (nthreads < 1) nthreads = 1;
        std::atomic<size_t> next{0};
        std::vector<Counters> cnts(nthreads);
        auto worker = [&](int t) {
            oracle_pin::Pin on_cpu(t);  // (orc_set_thread_pinning: worker t on the t-th CPU, node by node; off by default)
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= nq) break;
                answers[i] = search(datas[i].data(), knbn, ef, total ? &cnts[t] : nullptr);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
        worker(0);
        for (auto& x : th) x.join();
        if (total) for (auto& c : cnts) total->add(c);
        return answers;
    }

    // ---- construction: insert_slice (src/hnsw.rs:1077-1215), serial ----------------------
    void insert(const float* data, size_t d, size_t origin_id) {
        if (data_dimension == 0) data_dimension = d;
        if (d != data_dimension) throw std::runtime_error("insert: dimension mismatch");
        // generate_new_point (src/hnsw.rs:503-531)
        size_t level = layer_g.generate();
        PointId p_id{(uint8_t)level, (int32_t)points_by_layer[level].size()};
        auto new_point = std::make_shared<Point>(data, d, origin_id, p_id);
        points_by_layer[level].push_back(new_point);
        nb_point += 1;
        size_t point_rank = nb_point;
        // :1089-1109
        std::shared_ptr<Point> enter_point_copy = entry_point;
        uint8_t max_level_observed = 0;
        if (enter_point_copy) {
            if (point_rank == 1) return;
            max_level_observed = enter_point_copy->p_id.layer;
        }
        if (!enter_point_copy) {
            check_entry_point(new_point);
            return;
        }
        float dist_to_entry = eval(data, enter_point_copy->v.data(), nullptr);  // :1110-1112
        for (int l = max_level_observed; l >= (int)level + 1; --l) {            // :1114
            RustBinaryHeap sorted_points = search_layer(data, enter_point_copy, 1, (uint8_t)l, nullptr);
            if (sorted_points.len() > 1) throw std::runtime_error("in insert : search_layer returned > 1 points");
            PWO ep;
            if (sorted_points.pop(ep)) {                                        // :1138
                if (new_point->neighbours[l].size() < (size_t)(uint8_t)max_nb_connection)  // get_max_nb_connection() is u8
                    new_point->neighbours[l].push_back(ep);                     // :1140-1144
                float tmp_dist = eval(data, ep->point_ref->v.data(), nullptr);  // :1146
                if (tmp_dist < dist_to_entry) {
                    enter_point_copy = ep->point_ref;
                    dist_to_entry = tmp_dist;
                }
            }
        }
        for (int l = (int)level; l >= 0; --l) {                                 // :1158
            size_t ef = ef_construction;
            RustBinaryHeap sorted_points = search_layer(data, enter_point_copy, ef, (uint8_t)l, nullptr);
            sorted_points = from_positive_to_negative(sorted_points);           // :1173
            if (!sorted_points.is_empty()) {
                size_t nb_conn;
                bool extend_c;
                if (l == 0) { nb_conn = 2 * max_nb_connection; extend_c = extend_candidates; }
                else { nb_conn = max_nb_connection; extend_c = false; }
                std::vector<PWO> neighbours;
                select_neighbours(data, sorted_points, nb_conn, extend_c, (uint8_t)l, keep_pruned, neighbours);
                sort_unstable(neighbours);                                      // :1195
                new_point->neighbours[l] = neighbours;                          // :1197
                if (!neighbours.empty()) enter_point_copy = neighbours[0]->point_ref;  // :1201-1203
            }
        }
        reverse_update_neighborhood_simple(new_point);                          // :1210
        check_entry_point(new_point);                                           // :1212
    }

    // neighbour list of flat point (for dumps and graph comparison)
    static void sort_unstable(std::vector<PWO>& v) {
        // Rust sort_unstable by Ord (distance only); for tie-free lists any sort agrees.
        std::stable_sort(v.begin(), v.end(), [](const PWO& a, const PWO& b) { return pwo_lt(a, b); });
    }

private:
    void check_entry_point(const std::shared_ptr<Point>& new_point) {  // src/hnsw.rs:534-557
        if (entry_point) {
            if (new_point->p_id.layer > entry_point->p_id.layer) entry_point = new_point;
        } else {
            entry_point = new_point;
        }
    }
    // src/hnsw.rs:1664-1681: iterate the heap's backing array in order, push negated.
    static RustBinaryHeap from_positive_to_negative(RustBinaryHeap& positive_heap) {
        RustBinaryHeap negative_heap;
        for (const PWO& p : positive_heap.data) {
            if (!(p->dist_to_ref >= 0.f)) throw std::runtime_error("assert p.dist_to_ref >= 0");
            negative_heap.push(std::make_shared<PointWithOrder>(p->point_ref, -p->dist_to_ref));
        }
        return negative_heap;
    }
    // src/hnsw.rs:1241-1289
    void reverse_update_neighborhood_simple(const std::shared_ptr<Point>& new_point) {
        int level = new_point->p_id.layer;
        for (int l = level; l >= 0; --l) {
            // the reference holds a read lock on new_point.neighbours while iterating; the list
            // is not modified during the loop (q != new_point), so iterating in place is identical
            for (const PWO& q : new_point->neighbours[l]) {
                if (new_point->p_id != q->point_ref->p_id) {
                    Point& q_point = *q->point_ref;
                    auto n_to_add = std::make_shared<PointWithOrder>(new_point, q->dist_to_ref);
                    size_t l_n = n_to_add->point_ref->p_id.layer;  // the NEW point's level, not l (:1257)
                    auto& lst = q_point.neighbours[l_n];
                    bool already = false;
                    for (const PWO& old : lst)
                        if (old->point_ref->p_id == new_point->p_id) { already = true; break; }
                    if (already) continue;
                    lst.push_back(n_to_add);
                    size_t nbn_at_l = lst.size();
                    size_t threshold_shrinking = l_n > 0 ? max_nb_connection : 2 * max_nb_connection;
                    bool shrink = nbn_at_l > threshold_shrinking;
                    sort_unstable(lst);
                    if (shrink) lst.pop_back();
                }
            }
        }
    }
    // src/hnsw.rs:1299-1421 (Navarro heuristic).  `candidates` holds negated distances.
    void select_neighbours(const float* data, RustBinaryHeap& candidates, size_t nb_neighbours_asked,
                           bool extend_candidates_asked, uint8_t layer, bool keep_pruned_,
                           std::vector<PWO>& neighbours_vec) {
        neighbours_vec.clear();
        bool extend = false;
        if (candidates.len() <= nb_neighbours_asked) {
            if (!extend_candidates_asked) {
                PWO p;
                while (candidates.pop(p))
                    neighbours_vec.push_back(std::make_shared<PointWithOrder>(p->point_ref, -p->dist_to_ref));
                return;
            }
            extend = true;
        }
        if (extend) {
            // HashMap iteration order in the reference is arbitrary; the new candidates are all
            // pushed into a heap, so with tie-free distances the order is unobservable.
            std::unordered_map<PointId, std::shared_ptr<Point>, PointIdHash> candidates_set, new_set;
            for (const PWO& c : candidates.data) candidates_set.emplace(c->point_ref->p_id, c->point_ref);
            for (auto& kv : candidates_set)
                for (const PWO& q : kv.second->neighbours[layer])
                    if (!candidates_set.count(q->point_ref->p_id) && !new_set.count(q->point_ref->p_id))
                        new_set.emplace(q->point_ref->p_id, q->point_ref);
            for (auto& kv : new_set) {
                float dist_topoint = eval(data, kv.second->v.data(), nullptr);
                candidates.push(std::make_shared<PointWithOrder>(kv.second, -dist_topoint));
            }
        }
        RustBinaryHeap discarded_points;
        while (!candidates.is_empty() && neighbours_vec.size() < nb_neighbours_asked) {
            PWO e_p;
            candidates.pop(e_p);
            bool e_to_insert = true;
            const float* e_point_v = e_p->point_ref->v.data();
            if (!(e_p->dist_to_ref <= 0.f)) throw std::runtime_error("assert e_p.dist_to_ref <= 0");
            for (const PWO& dn : neighbours_vec) {  // any(|d| eval(e, d) <= -e_p.dist)  (:1373-1375)
                if (eval(e_point_v, dn->point_ref->v.data(), nullptr) <= -e_p->dist_to_ref) {
                    e_to_insert = false;
                    break;
                }
            }
            if (e_to_insert) {
                neighbours_vec.push_back(std::make_shared<PointWithOrder>(e_p->point_ref, -e_p->dist_to_ref));
            } else if (keep_pruned_) {
                discarded_points.push(std::make_shared<PointWithOrder>(e_p->point_ref, e_p->dist_to_ref));
            }
        }
        if (keep_pruned_) {
            while (!discarded_points.is_empty() && neighbours_vec.size() < nb_neighbours_asked) {
                PWO best;
                discarded_points.pop(best);
                neighbours_vec.push_back(std::make_shared<PointWithOrder>(best->point_ref, -best->dist_to_ref));
            }
        }
    }
};

}  // namespace oracle
