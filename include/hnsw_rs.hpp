// hnsw_rs.hpp -- header-only C++ mirror of the hnsw_rs interface for the search path, on top of the C ABI of
// libhnsw_mi355x.so (include/hnsw_mi355x.h).  The reference is Rust; this image has no Rust toolchain, so
// the host side above the C ABI is written in C++ with the crate's names and argument meaning:
//
//   hnsw_rs::Neighbour / PointId                      src/hnsw.rs:46, :98-107
//   hnsw_rs::DistL2 / DistCosine / DistDot / DistL1   anndists distance type names (src/hnswio.rs:473-490)
//   hnsw_rs::Hnsw<T, D>::search / parallel_search     src/hnsw.rs:1597, :1612
//   hnsw_rs::AnnT<T> (search_neighbours, parallel_search_neighbours, file_dump)   src/api.rs:13-38
//   hnsw_rs::HnswIo(dir, basename).load_hnsw<T, D>()  src/hnswio.rs:317, :431
//
// Errors: anyhow::Result becomes hnsw_rs::Error (carries the C-ABI status); nothing aborts the process.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hnsw_mi355x.h"

namespace hnsw_rs {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
    if (rc != HNSWGPU_OK) throw Error(rc, hnswgpu_last_error());
}

struct PointId {  // PointId(pub u8, pub i32)
    uint8_t layer;
    int32_t rank;
    bool operator==(const PointId& o) const { return layer == o.layer && rank == o.rank; }
};
struct Neighbour {  // #[repr(C)] { d_id: usize, distance: f32, p_id: PointId }
    size_t d_id;
    float distance;
    PointId p_id;
};

struct DistL2 { static constexpr int code = HNSWGPU_DIST_L2; };
struct DistCosine { static constexpr int code = HNSWGPU_DIST_COSINE; };
struct DistDot { static constexpr int code = HNSWGPU_DIST_DOT; };
struct DistL1 { static constexpr int code = HNSWGPU_DIST_L1; };

// trait AnnT { type Val; ... }
template <class T>
struct AnnT {
    virtual ~AnnT() = default;
    virtual std::vector<Neighbour> search_neighbours(const std::vector<T>& data, size_t knbn, size_t ef_s) const = 0;
    virtual std::vector<std::vector<Neighbour>> parallel_search_neighbours(const std::vector<std::vector<T>>& data,
                                                                           size_t knbn, size_t ef_s) const = 0;
    virtual std::string file_dump(const std::string& path, const std::string& file_basename) const = 0;
};

template <class T, class D>
class Hnsw;

// Hnsw<f32, D>: owns a hnswgpu_index (flat host graph + HBM replica).
template <class D>
class Hnsw<float, D> : public AnnT<float> {
public:
    // Hnsw::new(max_nb_connection, max_elements, max_layer, ef_construction, D{})
    Hnsw(size_t max_nb_connection, size_t /*max_elements*/, size_t max_layer, size_t ef_construction, D = D{}) {
        params_.max_nb_connection = max_nb_connection;
        params_.ef_construction = ef_construction;
        params_.max_layer = max_layer;
        params_.dist = D::code;
        params_.level_scale_factor = 1.0;
        if (max_nb_connection > 256) throw Error(HNSWGPU_ERR_ARG, "error max_nb_connection must be less equal than 256");
    }
    explicit Hnsw(hnswgpu_index* loaded) : idx_(loaded) {}
    Hnsw(Hnsw&& o) noexcept : idx_(o.idx_), params_(o.params_) { o.idx_ = nullptr; }
    Hnsw(const Hnsw&) = delete;
    Hnsw& operator=(const Hnsw&) = delete;
    ~Hnsw() override { hnswgpu_free_index(idx_); }

    void set_extend_candidates(bool f) { params_.extend_candidates = f; }
    void set_keeping_pruned(bool f) { params_.keep_pruned = f; }
    void modify_level_scale(double f) { params_.level_scale_factor = f < 0.2 ? 0.2 : (f > 1.0 ? 1.0 : f); }

    // parallel_insert(&[(&Vec<T>, usize)]): data row-major n x d, ids may be empty (0..n-1)
    void parallel_insert(const std::vector<float>& data, size_t d, const std::vector<uint64_t>& ids = {}, int nthreads = 0) {
        if (idx_) throw Error(HNSWGPU_ERR_ARG, "this binding builds an index in one parallel_insert call");
        params_.nthreads = nthreads;
        check(hnswgpu_build(data.data(), data.size() / d, d, ids.empty() ? nullptr : ids.data(), &params_, &idx_));
    }
    size_t get_nb_point() const { return idx_ ? hnswgpu_nb_point(idx_) : 0; }
    uint8_t get_max_level_observed() const { return idx_ ? (uint8_t)hnswgpu_max_level_observed(idx_) : 0; }

    // replication into the HBM of one device (one process per GPU)
    void upload(int device = 0) { check(hnswgpu_upload(idx_, device)); }
    void set_strict_ties(bool on) { check(hnswgpu_set_strict_ties(idx_, on)); }
    // distances of the following searches summed like the crate's simdeez_f build (true) or its default build (false)
    void set_simd_order_arithmetic(bool on) { check(hnswgpu_set_arithmetic(idx_, on ? HNSWGPU_ARITH_SIMD8 : HNSWGPU_ARITH_SCALAR)); }

    // search(&[T], knbn, ef) -> Vec<Neighbour>
    std::vector<Neighbour> search(const std::vector<float>& data, size_t knbn, size_t ef) const {
        return flat_search(data.data(), 1, data.size(), knbn, ef)[0];
    }
    // parallel_search(&[Vec<T>], knbn, ef) -> Vec<Vec<Neighbour>>, answers in input order
    std::vector<std::vector<Neighbour>> parallel_search(const std::vector<std::vector<float>>& datas, size_t knbn, size_t ef) const {
        if (datas.empty()) return {};
        const size_t d = datas[0].size();
        std::vector<float> flat;
        flat.reserve(datas.size() * d);
        for (const auto& v : datas) flat.insert(flat.end(), v.begin(), v.end());
        return flat_search(flat.data(), datas.size(), d, knbn, ef);
    }
    // AnnT
    std::vector<Neighbour> search_neighbours(const std::vector<float>& data, size_t knbn, size_t ef_s) const override {
        return search(data, knbn, ef_s);
    }
    std::vector<std::vector<Neighbour>> parallel_search_neighbours(const std::vector<std::vector<float>>& data, size_t knbn,
                                                                   size_t ef_s) const override {
        return parallel_search(data, knbn, ef_s);
    }
    std::string file_dump(const std::string& path, const std::string& file_basename) const override {
        if (!idx_) throw Error(HNSWGPU_ERR_EMPTY, "entry point not initialized");
        check(hnswgpu_file_dump(idx_, path.c_str(), file_basename.c_str()));
        return file_basename;
    }
    hnswgpu_index* handle() const { return idx_; }

private:
    std::vector<std::vector<Neighbour>> flat_search(const float* q, size_t nq, size_t d, size_t knbn, size_t ef) const {
        std::vector<std::vector<Neighbour>> out(nq);
        if (!idx_) return out;  // empty index => empty answers (src/hnsw.rs:1498-1503)
        std::vector<uint64_t> ids(nq * knbn);
        std::vector<float> dists(nq * knbn);
        std::vector<uint8_t> layers(nq * knbn);
        std::vector<int32_t> ranks(nq * knbn);
        std::vector<uint32_t> counts(nq);
        check(hnswgpu_search_batch(idx_, q, nq, d, knbn, ef, ids.data(), dists.data(), layers.data(), ranks.data(), counts.data()));
        for (size_t i = 0; i < nq; ++i)
            for (uint32_t j = 0; j < counts[i]; ++j) {
                const size_t o = i * knbn + j;
                out[i].push_back(Neighbour{(size_t)ids[o], dists[o], PointId{layers[o], ranks[o]}});
            }
        return out;
    }
    hnswgpu_index* idx_ = nullptr;
    hnswgpu_build_params params_{};
};

// HnswIo::new(directory, basename)
class HnswIo {
public:
    HnswIo(std::string directory, std::string basename) : dir_(std::move(directory)), basename_(std::move(basename)) {}
    const std::string& get_basename() const { return basename_; }
    // load_hnsw::<f32, D>(): the dump's distance must match D by its short name (src/hnswio.rs:473-490)
    template <class T, class D>
    Hnsw<T, D> load_hnsw() const {
        hnswgpu_index* idx = nullptr;
        check(hnswgpu_load_dump(dir_.c_str(), basename_.c_str(), D::code, &idx));
        return Hnsw<T, D>(idx);
    }

private:
    std::string dir_, basename_;
};

}  // namespace hnsw_rs
