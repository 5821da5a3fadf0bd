// oracle_capi.cpp -- ctypes-friendly C entry points of the CPU ORACLE (liboracle_hnsw.so).
//
// TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and the
// `cpu_baseline` leg of bench.py.  Nothing under hnswlib-rs_amd/ links or dlopens this.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstring>
#include <string>
#include "hnsw_oracle.hpp"
#include "flat_baseline.hpp"
#include "hnswio_oracle.hpp"

using namespace oracle;

static thread_local std::string g_err;
#define ORC_TRY try {
#define ORC_CATCH(ret)                                \
    }                                                 \
    catch (const std::exception& e) {                 \
        g_err = e.what();                             \
        return ret;                                   \
    }

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

void* orc_new(size_t max_nb_conn, size_t max_elements, size_t max_layer, size_t ef_c, int dist) {
    ORC_TRY
    return new Hnsw(max_nb_conn, max_elements, max_layer, ef_c, (DistKind)dist);
    ORC_CATCH(nullptr)
}
void orc_free(void* h) { delete static_cast<Hnsw*>(h); }
void orc_modify_level_scale(void* h, double f) { static_cast<Hnsw*>(h)->modify_level_scale(f); }
void orc_set_extend_candidates(void* h, int flag) { static_cast<Hnsw*>(h)->extend_candidates = flag != 0; }
void orc_set_keeping_pruned(void* h, int flag) { static_cast<Hnsw*>(h)->keep_pruned = flag != 0; }
size_t orc_get_nb_point(void* h) { return static_cast<Hnsw*>(h)->nb_point; }
size_t orc_get_layer_nb_point(void* h, size_t l) { return static_cast<Hnsw*>(h)->get_layer_nb_point(l); }
int orc_get_max_level_observed(void* h) { return static_cast<Hnsw*>(h)->get_max_level_observed(); }
size_t orc_get_dimension(void* h) { return static_cast<Hnsw*>(h)->data_dimension; }

// Hnsw::insert for n points in order (serial; src/hnsw.rs:1069-1071).
int orc_insert_batch(void* hv, const float* data, size_t n, size_t d, const size_t* ids) {
    ORC_TRY
    Hnsw* h = static_cast<Hnsw*>(hv);
    for (size_t i = 0; i < n; ++i) h->insert(data + i * d, d, ids ? ids[i] : i);
    return 0;
    ORC_CATCH(-1)
}

// Hnsw::search (src/hnsw.rs:1597).  Outputs hold up to k entries; *count = number returned.
int orc_search(void* hv, const float* q, size_t d, size_t k, size_t ef, uint64_t* out_ids, float* out_dists,
               uint8_t* out_layer, int32_t* out_rank, uint32_t* count) {
    ORC_TRY
    Hnsw* h = static_cast<Hnsw*>(hv);
    if (h->data_dimension && d != h->data_dimension) throw std::runtime_error("search: dimension mismatch");
    auto r = h->search(q, k, ef);
    for (size_t i = 0; i < r.size(); ++i) {
        out_ids[i] = r[i].d_id;
        out_dists[i] = r[i].distance;
        if (out_layer) out_layer[i] = r[i].p_id.layer;
        if (out_rank) out_rank[i] = r[i].p_id.rank;
    }
    *count = (uint32_t)r.size();
    return 0;
    ORC_CATCH(-1)
}

// Hnsw::parallel_search (src/hnsw.rs:1612-1635).  `queries` is nq x d row-major; each row is
// first copied into its own heap vector (the reference takes &[Vec<T>]).  Only the
// parallel_search call itself is timed (*elapsed_s), like
// examples/ann-sift1m-128-euclidean.rs:148-164.  counters (may be null) = {n_dist, n_expand,
// n_ids_read} summed over all queries.
// placement of the worker threads of every parallel_search of this library: 0 = wherever the scheduler puts them (default),
// 1 = worker t pinned to the t-th logical CPU, NUMA node by NUMA node (pinning.hpp)
void orc_set_thread_pinning(int mode) { oracle_pin::mode().store(mode); }
int orc_pinning_cpu(int t) {
    const std::vector<int>& o = oracle_pin::cpu_order();
    return o.empty() ? -1 : o[(size_t)t % o.size()];
}

int orc_parallel_search(void* hv, const float* queries, size_t nq, size_t d, size_t k, size_t ef, int nthreads,
                        uint64_t* out_ids, float* out_dists, uint8_t* out_layer, int32_t* out_rank,
                        uint32_t* out_counts, uint64_t* counters, double* elapsed_s) {
    ORC_TRY
    Hnsw* h = static_cast<Hnsw*>(hv);
    if (h->data_dimension && d != h->data_dimension) throw std::runtime_error("search: dimension mismatch");
    std::vector<std::vector<float>> datas(nq);
    for (size_t i = 0; i < nq; ++i) datas[i].assign(queries + i * d, queries + (i + 1) * d);
    Counters total;
    auto t0 = std::chrono::steady_clock::now();
    auto ans = h->parallel_search(datas, k, ef, nthreads, counters ? &total : nullptr);
    auto t1 = std::chrono::steady_clock::now();
    if (elapsed_s) *elapsed_s = std::chrono::duration<double>(t1 - t0).count();
    for (size_t i = 0; i < nq; ++i) {
        const auto& r = ans[i];
        for (size_t j = 0; j < r.size(); ++j) {
            out_ids[i * k + j] = r[j].d_id;
            out_dists[i * k + j] = r[j].distance;
            if (out_layer) out_layer[i * k + j] = r[j].p_id.layer;
            if (out_rank) out_rank[i * k + j] = r[j].p_id.rank;
        }
        out_counts[i] = (uint32_t)r.size();
    }
    if (counters) {
        counters[0] = total.n_dist;
        counters[1] = total.n_expand;
        counters[2] = total.n_ids_read;
    }
    return 0;
    ORC_CATCH(-1)
}

// Hnsw::search_filter(data, knbn, ef, Some(&Vec<usize>)) (src/hnsw.rs:1487): `allowed` = sorted origin ids
int orc_search_filter(void* hv, const float* q, size_t d, size_t k, size_t ef, const uint64_t* allowed, size_t n_allowed,
                      uint64_t* out_ids, float* out_dists, uint8_t* out_layer, int32_t* out_rank, uint32_t* out_count) {
    ORC_TRY
    Hnsw* h = static_cast<Hnsw*>(hv);
    if (h->data_dimension && d != h->data_dimension) throw std::runtime_error("search: dimension mismatch");
    Hnsw::Filter f(allowed, allowed + n_allowed);
    if (!std::is_sorted(f.begin(), f.end())) throw std::runtime_error("search_filter: the id vector must be sorted");
    auto r = h->search(q, k, ef, nullptr, &f);
    for (size_t j = 0; j < r.size(); ++j) {
        out_ids[j] = r[j].d_id;
        out_dists[j] = r[j].distance;
        if (out_layer) out_layer[j] = r[j].p_id.layer;
        if (out_rank) out_rank[j] = r[j].p_id.rank;
    }
    *out_count = (uint32_t)r.size();
    return 0;
    ORC_CATCH(-1)
}
// The same for a batch (what a caller of the reference does with Rayon around Hnsw::search_filter: one filter, many queries,
// tests/filtertest.rs:155-271), on nthreads workers pulling query indices from a counter; answers in input order.
// out_status[i] = 1 where the reference panics on query i (src/hnsw.rs:973: peek().unwrap() on an emptied heap), count 0.
// counters (may be null): {n_dist, n_expand, n_ids_read} summed over the batch.
int orc_parallel_search_filter(void* hv, const float* queries, size_t nq, size_t d, size_t k, size_t ef, const uint64_t* allowed,
                               size_t n_allowed, int nthreads, uint64_t* out_ids, float* out_dists, uint8_t* out_layer,
                               int32_t* out_rank, uint32_t* out_counts, uint8_t* out_status, uint64_t* counters, double* elapsed_s) {
    ORC_TRY
    Hnsw* h = static_cast<Hnsw*>(hv);
    if (h->data_dimension && d != h->data_dimension) throw std::runtime_error("search: dimension mismatch");
    Hnsw::Filter f(allowed, allowed + n_allowed);
    if (!std::is_sorted(f.begin(), f.end())) throw std::runtime_error("search_filter: the id vector must be sorted");
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    std::atomic<size_t> next{0};
    std::vector<Counters> cnts((size_t)nthreads);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int t) {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= nq) break;
            std::vector<Neighbour> r;
            uint8_t st = 0;
            try {
                r = h->search(queries + i * d, k, ef, counters ? &cnts[(size_t)t] : nullptr, &f);
            } catch (const std::runtime_error&) {
                st = 1;  // (the only throw a well-formed search can meet: the reference's panic)
            }
            for (size_t j = 0; j < k; ++j) {
                const bool have = j < r.size();
                out_ids[i * k + j] = have ? r[j].d_id : 0;
                out_dists[i * k + j] = have ? r[j].distance : 0.f;
                if (out_layer) out_layer[i * k + j] = have ? r[j].p_id.layer : 0;
                if (out_rank) out_rank[i * k + j] = have ? r[j].p_id.rank : 0;
            }
            out_counts[i] = (uint32_t)r.size();
            if (out_status) out_status[i] = st;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& x : th) x.join();
    auto t1 = std::chrono::steady_clock::now();
    if (elapsed_s) *elapsed_s = std::chrono::duration<double>(t1 - t0).count();
    if (counters) {
        Counters total;
        for (auto& c : cnts) total.add(c);
        counters[0] = total.n_dist;
        counters[1] = total.n_expand;
        counters[2] = total.n_ids_read;
    }
    return 0;
    ORC_CATCH(-1)
}
// The optimised flat-array CPU baseline (flat_baseline.hpp; timing only, never a parity check): built from a loaded index.
void* orc_flat_new(void* hv) {
    ORC_TRY
    oracle_pin::InterleavedAllocations spread;  // (one thread builds the arrays every search thread of both sockets will read)
    return new FlatBaseline(*static_cast<Hnsw*>(hv));
    ORC_CATCH(nullptr)
}
void orc_flat_free(void* fv) { delete static_cast<FlatBaseline*>(fv); }
int orc_flat_parallel_search(void* fv, const float* queries, size_t nq, size_t d, size_t k, size_t ef, int nthreads,
                             uint64_t* out_ids, float* out_dists, uint32_t* out_counts, double* elapsed_s) {
    ORC_TRY
    FlatBaseline* f = static_cast<FlatBaseline*>(fv);
    if (d != f->d) throw std::runtime_error("flat baseline: dimension mismatch");
    auto t0 = std::chrono::steady_clock::now();
    f->parallel_search(queries, nq, k, ef, nthreads, out_ids, out_dists, out_counts);
    auto t1 = std::chrono::steady_clock::now();
    if (elapsed_s) *elapsed_s = std::chrono::duration<double>(t1 - t0).count();
    return 0;
    ORC_CATCH(-1)
}
// timing-only switch: distances in the crate's SIMD summation order (see dist_simd8); never used by a parity check
int orc_set_simd_order(void* hv, int on) {
    ORC_TRY
    static_cast<Hnsw*>(hv)->simd_order = on != 0;
    return 0;
    ORC_CATCH(-1)
}
// AnnT::file_dump (src/api.rs:70-93) without the unique-name logic: overwrites.
int orc_file_dump(void* hv, const char* dir, const char* basename) {
    ORC_TRY
    file_dump(*static_cast<Hnsw*>(hv), dir, basename);
    return 1;
    ORC_CATCH(-1)
}
// HnswIo::new(dir, basename).load_hnsw::<f32, D>()
void* orc_load(const char* dir, const char* basename, int dist) {
    ORC_TRY
    oracle_pin::InterleavedAllocations spread;  // (likewise: the points and their lists are allocated by this one thread)
    return load_hnsw(dir, basename, (DistKind)dist).release();
    ORC_CATCH(nullptr)
}

// Distance<f32>::eval of the restated anndists metrics, for arithmetic tests.
float orc_dist(int kind, const float* a, const float* b, size_t d) {
    ORC_TRY
    return dist_eval((DistKind)kind, a, b, d);
    ORC_CATCH(NAN)
}
// out[q][r] = eval(queries[q], rows[r]) (row-major matrices of dimension d), for the device's distance-routine sweep
void orc_dist_matrix(int kind, const float* queries, size_t nq, const float* rows, size_t n, size_t d, float* out) {
    for (size_t q = 0; q < nq; ++q)
        for (size_t r = 0; r < n; ++r) out[q * n + r] = dist_eval((DistKind)kind, queries + q * d, rows + r * d, d);
}
// the same matrix in the crate's SIMD summation order (dist_simd8): the checker of the device's opt-in SIMD-order arithmetic
void orc_dist_matrix_simd8(int kind, const float* queries, size_t nq, const float* rows, size_t n, size_t d, float* out) {
    for (size_t q = 0; q < nq; ++q)
        for (size_t r = 0; r < n; ++r) out[q * n + r] = dist_simd8((DistKind)kind, queries + q * d, rows + r * d, d);
}
void orc_l2_normalize(float* v, size_t d) { l2_normalize(v, d); }
// f32::ln restated (ref_logf.hpp) and the host libm's logf, for the exhaustive comparison of the two
float orc_ref_logf(float x) { return ref_logf(x); }
uint64_t orc_ref_logf_mismatches(uint32_t first_bits, uint32_t last_bits, uint32_t step) {
    uint64_t bad = 0;
    for (uint64_t u = first_bits; u <= last_bits; u += step) {
        float x, a, b;
        const uint32_t w = (uint32_t)u;
        std::memcpy(&x, &w, 4);
        a = ref_logf(x);
        b = std::log(x);
        if (std::memcmp(&a, &b, 4) != 0 && !(a != a && b != b)) ++bad;
    }
    return bad;
}

// Level generator stream (for checking the product builder draws the same levels).
void orc_levels(size_t max_nb_conn, double scale_factor, size_t maxlevel, size_t n, uint8_t* out) {
    LayerGenerator g(max_nb_conn, scale_factor, maxlevel);
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)g.generate();
}

// Raw BinaryHeap restatement driver for unit tests: push all values, then either pop `npop`
// times (mode 0: out = popped values) or into_sorted_vec (mode 1).  `tags` travel with the
// values so tie order is observable.
int orc_heap_exercise(const float* vals, const int32_t* tags, size_t n, int mode, size_t npop, float* out_vals,
                      int32_t* out_tags) {
    ORC_TRY
    RustBinaryHeap hp;
    float dummy[1] = {0.f};
    for (size_t i = 0; i < n; ++i) {
        auto p = std::make_shared<Point>(dummy, 1, (size_t)tags[i], PointId{0, tags[i]});
        hp.push(std::make_shared<PointWithOrder>(p, vals[i]));
    }
    if (mode == 0) {
        for (size_t i = 0; i < npop; ++i) {
            PWO x;
            if (!hp.pop(x)) return (int)i;
            out_vals[i] = x->dist_to_ref;
            out_tags[i] = x->point_ref->p_id.rank;
        }
        return (int)npop;
    }
    auto v = hp.into_sorted_vec();
    for (size_t i = 0; i < v.size(); ++i) {
        out_vals[i] = v[i]->dist_to_ref;
        out_tags[i] = v[i]->point_ref->p_id.rank;
    }
    return (int)v.size();
    ORC_CATCH(-1)
}

// BinaryHeap::retain driver for unit tests: push all values, retain the entries whose keep[i] != 0 (i = push index),
// then into_sorted_vec.
int orc_heap_retain(const float* vals, const int32_t* tags, const uint8_t* keep, size_t n, float* out_vals, int32_t* out_tags) {
    ORC_TRY
    RustBinaryHeap hp;
    float dummy[1] = {0.f};
    for (size_t i = 0; i < n; ++i) {
        auto p = std::make_shared<Point>(dummy, 1, i, PointId{0, tags[i]});   // origin_id = push index
        hp.push(std::make_shared<PointWithOrder>(p, vals[i]));
    }
    hp.retain([&](const PWO& e) { return keep[e->point_ref->origin_id] != 0; });
    auto v = hp.into_sorted_vec();
    for (size_t i = 0; i < v.size(); ++i) {
        out_vals[i] = v[i]->dist_to_ref;
        out_tags[i] = v[i]->point_ref->p_id.rank;
    }
    return (int)v.size();
    ORC_CATCH(-1)
}

// Scripted BinaryHeap driver: step i pushes (vals[i], tags[i]) when is_pop[i] == 0, else pops.  out_* receive
// the popped entries in order (returns their number through *npopped); sorted_* receive into_sorted_vec of what
// is left (returns its length).  Used by the second Python transcription of std's BinaryHeap (tests/test_oracle.py,
// tests/test_second_opinion.py) and by the pin probes (tests/golden/make_pin_probes.py) on interleaved pushes and pops with ties.
int orc_heap_script(const float* vals, const int32_t* tags, const uint8_t* is_pop, size_t n, float* out_vals,
                    int32_t* out_tags, size_t* npopped, float* sorted_vals, int32_t* sorted_tags) {
    ORC_TRY
    RustBinaryHeap hp;
    float dummy[1] = {0.f};
    size_t np = 0;
    for (size_t i = 0; i < n; ++i) {
        if (is_pop[i]) {
            PWO x;
            if (!hp.pop(x)) continue;
            out_vals[np] = x->dist_to_ref;
            out_tags[np] = x->point_ref->p_id.rank;
            ++np;
        } else {
            auto p = std::make_shared<Point>(dummy, 1, (size_t)tags[i], PointId{0, tags[i]});
            hp.push(std::make_shared<PointWithOrder>(p, vals[i]));
        }
    }
    *npopped = np;
    auto v = hp.into_sorted_vec();
    for (size_t i = 0; i < v.size(); ++i) {
        sorted_vals[i] = v[i]->dist_to_ref;
        sorted_tags[i] = v[i]->point_ref->p_id.rank;
    }
    return (int)v.size();
    ORC_CATCH(-1)
}

}  // extern "C"
