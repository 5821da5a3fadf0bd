// search_device.hpp -- HBM replica of a FlatIndex and the batched-search driver (host API of
// search_device.hip).  Everything here needs a gfx950 device; there is no CPU fallback.
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "builder.hpp"
#include "flat_index.hpp"

namespace hnswgpu {

constexpr int ARITH_SCALAR = 0, ARITH_SIMD8 = 1;  // DeviceIndex::set_arithmetic

// The HNSWGPU_* tuning / test hooks.  The environment is read ONCE per process (at the first use) -- not on every launch --
// and again only when the caller says it changed it (hnswgpu_reload_env; the tests do).  -1 / false: not set.
struct Knobs {
    int hash_bits = -1;          // HNSWGPU_HASH_BITS: initial visited-table size (6..14)
    bool no_sched = false;       // HNSWGPU_NO_SCHED: searches in input order
    bool no_pair_descent = false;  // HNSWGPU_NO_PAIR_DESCENT: one query per wavefront in the descent kernel whatever the index
    bool no_inkernel = false;    // HNSWGPU_NO_INKERNEL: strict ties resolved by the literal kernel only
    int strict_wg_per_cu = -1;   // HNSWGPU_STRICT_WG_PER_CU
    int cand_lds = -1;           // HNSWGPU_CAND_LDS
    int waves_per_cu = -1;       // HNSWGPU_WAVES_PER_CU
    int exact_first = -1;        // HNSWGPU_EXACT_FIRST (test hook)
    bool trace_launch = false;   // HNSWGPU_TRACE_LAUNCH
    bool trace_host = false;     // HNSWGPU_TRACE_HOST
    int host_threads = -1;       // HNSWGPU_HOST_THREADS
    int host_chunks = -1;        // HNSWGPU_HOST_CHUNKS
    bool ffi_unpack = false;     // HNSWGPU_FFI_UNPACK
    int pair_search = -1;        // HNSWGPU_PAIR_SEARCH: 1 / 0 = two queries per wavefront as the first pass of a batch where the index allows it / never
                                 // (unset: strict DistCosine / DistDot batches of >= 40 000 queries on rows of one 128-byte line)
    int pair_tbits_delta = 0;    // HNSWGPU_PAIR_TBITS_DELTA: the pair kernel's visited tables, in powers of two relative to the one-query kernels'
    int pair_wg_per_cu = -1;     // HNSWGPU_PAIR_WG_PER_CU: cap on its resident workgroups per CU
};
const Knobs& knobs();
void reload_knobs();

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

// What a search call reports about itself (the timings are HIP events on the launch stream).
struct CallInfo {
    double ms = 0.0;         // all kernels of the call
    double main_ms = 0.0;    // first launch of the search kernel alone
    uint32_t launches = 0;
    uint32_t ties = 0;       // queries whose answer depended on the reference's heap order (resolved or flagged)
    uint32_t pair_retries = 0;  // queries the two-per-wavefront first pass handed back to the one-query kernels
    uint32_t panics = 0;     // filtered search: queries on which the reference panics (src/hnsw.rs:973)
};

// One replica of an index in the HBM of one device.  The replica is immutable after upload(); every search call takes
// a private workspace (scratch buffers, events, counters) from a pool, so concurrent calls on one replica are legal --
// like the reference's `&self` search (SURVEY.md 8b "Threading").
// Where answer j of query q is written, in bytes from the three base addresses a search is given: ids at (q k + j) id_stride,
// distances at (q k + j) dist_stride, counts at q count_stride.  The default is three dense arrays; {16, 16, 16} addresses the
// records of the reference's FFI in place (Neighbour_api {usize id; f32 d}, Neighbourhood_api {i64 nbgh; ptr}: src/libext.rs:58-87).
struct OutLayout {
    uint32_t id_stride = 8, dist_stride = 4, count_stride = 4;
};
// Host memory every device of the process can address (mapped, portable, page-locked): *dev receives the address the kernels
// use.  nullptr when the runtime refuses (no device, limits): the caller falls back to ordinary memory and an unpacking pass.
void* pinned_alloc(size_t bytes, void** dev);
void pinned_free(void* p);

class DeviceIndex {
public:
    DeviceIndex();
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
    // d_allowed != nullptr: Hnsw::search_filter with the sorted id vector d_allowed[0..n_allowed) (device memory) for
    // every query of the batch (src/hnsw.rs:1487-1580, src/filter.rs:11-15).
    // feed (may be null): d_queries is mapped pinned host memory that is still being filled -- fill(ctx, lo, hi) makes rows
    // [lo, hi) valid; the call then gathers and launches the descent kernel chunk by chunk, so that the device reads chunk i
    // across PCIe while the host gathers chunk i + 1 (the host-buffer entry points).
    struct RowFeed {
        void (*fill)(void* ctx, uint64_t lo, uint64_t hi);
        void* ctx;
        uint64_t chunk_rows;  // rows per chunk (the feeder's choice: its gather tasks are cut along these)
    };
    // `layout`: where the answers go relative to d_out_ids / d_out_dists / d_out_counts (OutLayout; default: dense arrays)
    int search_device(const float* d_queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef, uint64_t* d_out_ids,
                      float* d_out_dists, uint8_t* d_out_layer, int32_t* d_out_rank, uint32_t* d_out_counts,
                      uint32_t* d_stats, void* stream, const uint64_t* d_allowed, uint64_t n_allowed, CallInfo* info,
                      std::string& err, const RowFeed* feed = nullptr, OutLayout layout = OutLayout{});
    // same with host buffers (H2D + kernels + D2H); out_status (may be null): per query, 1 = the reference panics
    int search_host(const float* queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef, uint64_t* out_ids,
                    float* out_dists, uint8_t* out_layer, int32_t* out_rank, uint32_t* out_counts,
                    const uint64_t* allowed, uint64_t n_allowed, bool filtered, uint8_t* out_status, CallInfo* info,
                    std::string& err);
    // What search_host is made of, for callers with other shapes of input / output (the reference's FFI: an array of row
    // pointers in, per-query vectors out).  Queries come from `queries` (nq x d) or, when that is null, from rows[0..nq).
    // ONE section of the worker pool lasts the whole call: its threads gather the rows into mapped pinned memory (the descent
    // kernel reads them across PCIe, chunk by chunk, while the next chunk is gathered), stay awake while the device searches,
    // and unpack the answers the kernels wrote into a pinned arena -- a pool thread takes ~50 us to wake up, which a call of
    // 1.2 ms cannot afford twice.  sink.begin (may be null) runs before anything else, on the caller's thread (allocate);
    // sink.rows unpacks the answers of queries [lo, hi) and is called concurrently on disjoint ranges; the HostAnswers are
    // valid only during the call.
    struct HostAnswers {
        const uint64_t* ids;      // [nq][k]
        const float* dists;       // [nq][k]
        const uint8_t* layer;     // [nq][k]
        const int32_t* rank;      // [nq][k]
        const uint32_t* counts;   // [nq]
        const uint8_t* status;    // [nq] filtered search: 1 = the reference panics on this query (else nullptr)
    };
    // A sink whose own memory the device can address (pinned_alloc) may take the answers IN PLACE: `direct` (may be null; called
    // after begin) names the allocation and, inside it, the first id / distance / count slot (host addresses) with their strides
    // and returns true -- the kernels then write there (the device's view of the allocation is looked up per call, on the
    // index's device), rows() is never called and nothing is unpacked (no layer / rank / status).
    struct DirectOut {
        void* allocation;   // what pinned_alloc returned
        void* ids;
        void* dists;
        void* counts;
        OutLayout layout;
    };
    struct AnswerSink {
        bool (*begin)(void* ctx, uint64_t nq, uint64_t k);  // false: out of memory
        void (*rows)(void* ctx, const HostAnswers& a, uint64_t lo, uint64_t hi);
        void* ctx;
        bool (*direct)(void* ctx, DirectOut* out) = nullptr;
    };
    int search_host_staged(const float* queries, const float* const* rows, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef,
                           const uint64_t* allowed, uint64_t n_allowed, bool filtered, bool want_status, const AnswerSink& sink,
                           CallInfo* info, std::string& err);

    // The arithmetic of Distance::eval: ARITH_SCALAR (default) sums like the crate's default build, left to right, bit for
    // bit; ARITH_SIMD8 sums DistL2 / DistCosine / DistDot / DistL1 in the order of its simdeez_f / stdsimd builds (8 vertical
    // accumulators, Cargo.toml:104-111) -- the order behind the reference's published numbers; other distances stay scalar.
    void set_arithmetic(int a) { arith_.store(a == ARITH_SIMD8 ? ARITH_SIMD8 : ARITH_SCALAR); }
    int arithmetic() const { return arith_.load(); }
    // strict ties: decisions that depend on the reference's heap order are resolved with literal heaps
    void set_strict_ties(bool on) { strict_ties_.store(on); }
    bool strict_ties() const { return strict_ties_.load(); }
    CallInfo last_call() const;

private:
    struct Workspace;
    class Lease;
    Workspace* acquire(std::string& err);
    void release_ws(Workspace* w);
    void release();
    int run_exact(Workspace& w, const float* d_qpad, const uint32_t* d_qlist, uint32_t nq, uint64_t k, uint64_t ef,
                  const uint32_t* d_allow, uint64_t* d_out_ids, float* d_out_dists, uint8_t* d_out_layer,
                  int32_t* d_out_rank, uint32_t* d_out_counts, uint32_t* stats, void* stream, uint32_t* panics,
                  std::string& err, OutLayout layout);

    DeviceIndexView v_{};
    bool ready_ = false;
    int device_ = -1;
    int dist_ = DIST_L2;
    int num_cu_ = 0;
    uint64_t bytes_ = 0;
    // the replica (read-only after upload)
    void* d_vec_ = nullptr;
    void* d_nbr0_ = nullptr;
    void* d_up_ptr_ = nullptr;
    void* d_up_ids_ = nullptr;
    void* d_origin_ = nullptr;
    void* d_nrm2_ = nullptr;      // DistCosine: per-point squared norms (f64)
    // per-call workspaces
    std::mutex pool_mu_;
    std::vector<std::unique_ptr<Workspace>> all_ws_;
    std::vector<Workspace*> free_ws_;
    // what previous batches taught us (visited-table sizing per ef), and the last call's report
    mutable std::mutex meta_mu_;
    uint64_t adapt_ef_ = 0;
    uint32_t adapt_tbits_ = 0;
    CallInfo last_{};
    std::atomic<bool> strict_ties_{true};
    std::atomic<int> arith_{0};
    int kernel_metric() const;            // dist_, or its SIMD-order kernel variant
    std::atomic<int> descend_per_cu_[2] = {{0}, {0}};  // resident workgroups per CU of the descent kernel (asked once; [1]: two queries per wavefront)
    uint32_t up_deg_max_ = 0;  // longest list above the search layer
};

int device_count();
// The gather of a sharded search without any collective library: shard s's answers (nq_shard[s] x k, resident on
// devices[s]) are copied behind one another -- input order -- into the arrays of `root_device` with hipMemcpyPeerAsync on
// `root_stream` (xGMI between the GPUs of a node, a device-to-device copy when a shard lives on the root), then the stream is
// waited for.  layer / rank arrays and their per-shard entries may be null.
int gather_sharded_answers(const int* devices, int n_shards, const uint64_t* nq_shard, uint64_t k, const uint64_t* const* d_ids,
                           const float* const* d_dists, const uint8_t* const* d_layer, const int32_t* const* d_rank,
                           const uint32_t* const* d_counts, int root_device, uint64_t* root_ids, float* root_dists,
                           uint8_t* root_layer, int32_t* root_rank, uint32_t* root_counts, void* root_stream, std::string& err);
// the device side of GPU-assisted construction (builder.hpp) on HIP device `device`
std::unique_ptr<BuildSearchBackend> make_device_build_backend(int device);
// Distance<f32>::eval on the device through the search kernel's own distance routine: out[q][r] = dist(queries[q],
// rows[r]), rows evaluated in batches of `nf` (1..64) -- the lane-group branches the search takes for nf neighbours.
// pairs: nq == n and out[i] = dist(queries[i], rows[i]) (nf ignored).
int eval_distance_matrix_device(int dist, const float* queries, uint64_t nq, const float* rows, uint64_t n, uint64_t d,
                                uint32_t nf, bool pairs, float* out, std::string& err, int arithmetic = 0);

// The lane lab (search_kernels.hpp, lane_lab.inc): runs a script of wave-level operations of the search kernels on `device`
// (host buffers in, host buffer out; out[0] = words produced).  Test entry.
int lane_lab_device(int device, uint32_t mode, uint32_t p0, uint32_t p1, uint32_t p2, const uint32_t* ops, uint32_t n_ops,
                    const uint32_t* lanes, uint32_t n_lane_sets, uint32_t* out, uint32_t out_words, std::string& err);

}  // namespace hnswgpu
