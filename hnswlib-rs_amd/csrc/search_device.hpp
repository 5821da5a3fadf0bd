// search_device.hpp -- HBM replica of a FlatIndex and the batched-search driver (host API of
// search_device.hip).  Everything here needs a gfx950 device; there is no CPU fallback.
#pragma once
#include <cstdint>
#include <string>
#include "flat_index.hpp"

namespace hnswgpu {

// Plain-pointer view handed to the kernels by value.
struct DeviceIndexView {
    const float* vec;          // [n][row_stride] f32, rows zero-padded to 128-byte lines
    const uint32_t* nbr0;      // [n][deg_stride] flat ids of the SEARCH layer's lists, padded with NO_POINT
    const uint32_t* up_ptr;    // [n_up_layers][n+1] CSR offsets of the lists at layers >= 1
    const uint32_t* up_ids;    // concatenated lists of layers 1..n_up_layers (list order = file order)
    const uint64_t* origin_id; // [n]
    uint32_t n;
    uint32_t d;
    uint32_t row_stride;       // floats per row, multiple of 32
    uint32_t deg_stride;       // ids per nbr0 row, multiple of 16
    uint32_t n_up_layers;      // highest layer index holding any non-empty list
    uint32_t entry;            // flat id of the entry point
    uint32_t entry_level;      // its layer
    uint32_t search_layer;     // lowest non-empty layer (src/hnsw.rs:1534-1540)
    uint32_t layer_offset[NB_LAYER_MAX + 1];
};

class DeviceIndex {
public:
    DeviceIndex() = default;
    ~DeviceIndex();
    DeviceIndex(const DeviceIndex&) = delete;
    DeviceIndex& operator=(const DeviceIndex&) = delete;

    int upload(const FlatIndex& x, int device, std::string& err);
    bool ready() const { return ready_; }
    int device() const { return device_; }
    int dist() const { return dist_; }
    const DeviceIndexView& view() const { return v_; }
    uint64_t bytes() const { return bytes_; }

    // Hnsw::parallel_search on device-resident buffers.  d_queries: nq x d row-major.
    int search_device(const float* d_queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef, uint64_t* d_out_ids,
                      float* d_out_dists, uint8_t* d_out_layer, int32_t* d_out_rank, uint32_t* d_out_counts,
                      uint32_t* d_stats, void* stream, std::string& err);
    // same with host buffers (H2D + kernel + D2H)
    int search_host(const float* queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef, uint64_t* out_ids,
                    float* out_dists, uint8_t* out_layer, int32_t* out_rank, uint32_t* out_counts, std::string& err);

    // strict ties: re-run tie-affected queries with a literal emulation of the reference's binary heaps
    void set_strict_ties(bool on) { strict_ties_ = on; }
    bool strict_ties() const { return strict_ties_; }
    uint32_t last_ties() const { return last_ties_; }
    double last_kernel_ms() const { return last_ms_; }
    double last_main_kernel_ms() const { return last_main_ms_; }
    uint32_t last_launches() const { return last_launches_; }

private:
    int ensure_workspace(uint64_t nq, uint64_t k, std::string& err);
    void release();
    DeviceIndexView v_{};
    bool ready_ = false;
    int device_ = -1;
    int dist_ = DIST_L2;
    int num_cu_ = 0;
    uint64_t bytes_ = 0;
    // device allocations owned by this object
    void* d_vec_ = nullptr;
    void* d_nbr0_ = nullptr;
    void* d_up_ptr_ = nullptr;
    void* d_up_ids_ = nullptr;
    void* d_origin_ = nullptr;
    void* d_nrm2_ = nullptr;      // HNSW_COSINE_GROUPS builds: per-point squared norms (f64)
    // per-call workspace (grown on demand)
    void* d_qpad_ = nullptr;      uint64_t qpad_cap_ = 0;     // padded queries
    void* d_ctrl_ = nullptr;                                   // work counter, overflow counter
    void* h_ctrl_ = nullptr;                                   // pinned host copy of the counters (read back once per launch)
    void* d_retry_[2] = {nullptr, nullptr}; uint64_t retry_cap_ = 0;
    void* d_stats_ = nullptr;     uint64_t stats_cap_ = 0;
    void* d_bitmap_ = nullptr;    uint64_t bitmap_cap_ = 0;
    void* d_hostio_[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // for search_host
    uint64_t hostio_cap_q_ = 0, hostio_cap_k_ = 0, hostio_cap_n_ = 0;
    void* ev_start_ = nullptr;
    void* ev_stop_ = nullptr;
    void* ev_mid_ = nullptr;
    void* d_tie_ = nullptr;       uint64_t tie_cap_ = 0;      // queries flagged with an exact distance tie
    void* d_predist_ = nullptr;   void* d_order_ = nullptr;   // batch scheduling (estimate pass)
    uint64_t sched_cap_ = 0;
    void* ev_ks_ = nullptr;       void* ev_ke_ = nullptr;     // around the first launch of the search kernel
    void* d_heaps_ = nullptr;     uint64_t heaps_cap_ = 0;    // scratch of the exact replay
    void* d_cand_ = nullptr;      uint64_t strict_cap_ = 0;   // in-launch literal heaps: candidate_points beyond LDS
    uint64_t adapt_exact_ef_ = 0; bool adapt_exact_first_ = false;  // previous batch: did most queries meet a tie?
    bool strict_ties_ = true;
    uint32_t last_ties_ = 0;
    uint64_t adapt_ef_ = 0;       // visited-table sizing learned from previous batches with this ef
    uint32_t adapt_tbits_ = 0;
    double last_ms_ = 0.0;
    double last_main_ms_ = 0.0;
    uint32_t last_launches_ = 0;
};

int device_count();
// Distance<f32>::eval on the device for n pairs, same arithmetic as the search kernel.
int eval_distances_device(int dist, const float* a, const float* b, uint64_t n, uint64_t d, float* out, std::string& err);

}  // namespace hnswgpu
