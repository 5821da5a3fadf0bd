// builder.hpp -- host-side graph construction on flat storage.
//
// Restates Hnsw::insert_slice / parallel_insert (src/hnsw.rs:1077-1238), select_neighbours
// (:1299-1421) and reverse_update_neighborhood_simple (:1241-1289) on arrays indexed by
// insertion order instead of the reference's Arc<Point> web.  The GPU search path reads the
// result through FlatIndex; construction itself runs on the host cores (worker threads pull
// points from an atomic counter, per-point spin locks stand in for the per-point RwLocks).
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "flat_index.hpp"

namespace hnswgpu {

struct BuildParams {
    uint64_t max_nb_connection = 16;
    uint64_t ef_construction = 200;
    uint64_t max_layer = 16;
    int dist = DIST_L2;
    double level_scale_factor = 1.0;
    bool extend_candidates = false;
    bool keep_pruned = false;
    int nthreads = 0;            // 1 = serial (deterministic), 0 = hardware_concurrency
    bool fast_arithmetic = false;  // false: reference-order scalar sums
    int gpu_device = -1;         // >= 0: GPU-assisted construction (insert_batch_gpu) on that device
    uint64_t gpu_window = 0;     // 0 = default window cap
};

struct Edge {
    uint32_t id;  // builder id = insertion order
    float dist;
};

// What the device returns for one window of points (GPU-assisted construction): the results of the searches of
// insert_slice (src/hnsw.rs:1114-1197) against the snapshot the device holds.
struct WindowSearchResults {
    std::vector<uint32_t> slot0;    // [count] slot of point i's layer-0 search; layer l -> slot0[i] + l
    std::vector<uint32_t> out_ids;  // [slots][ef_c] candidates, ascending distance (builder ids)
    std::vector<float> out_d;
    std::vector<uint32_t> out_n;    // [slots]
    std::vector<uint32_t> hit_ids;  // [count][NB_LAYER_MAX] ef = 1 hit per layer above the point's level (NO_POINT: none)
    std::vector<float> hit_d;
    // select_neighbours done on the device as well (WindowSelect::on_device): then out_ids / out_d stay empty and every slot
    // comes back already selected, in selection order
    bool selected = false;
    uint32_t sel_stride = 0;
    std::vector<uint32_t> sel_ids;  // [slots][sel_stride]
    std::vector<float> sel_d;
    std::vector<uint32_t> sel_n;    // [slots]
};
// what the device needs to run select_neighbours (src/hnsw.rs:1299-1421) itself for the slots of a window
struct WindowSelect {
    bool on_device = false;
    uint32_t nb_layer0 = 0;   // neighbours asked at layer 0 (2 M) ...
    uint32_t nb_upper = 0;    // ... and above (M)
    bool keep_pruned = false;
};
// The device side of GPU-assisted construction (implemented in search_device.hip; builder.cpp stays free of HIP).
class BuildSearchBackend {
public:
    virtual ~BuildSearchBackend() = default;
    // can this backend serve a build with these parameters at all (a device is visible, the ordinal exists, ef_construction is
    // supported)?  Called BEFORE any point is accepted, so that these failures leave the index unchanged.
    virtual int check(uint64_t ef_construction, std::string& err) = 0;
    // all vectors of the build (builder order, n x d) and every point's level; empty lists everywhere
    virtual int begin(const float* const* chunks, uint64_t chunk_rows, uint64_t n, uint64_t d, const uint8_t* levels, int dist,
                      uint64_t max_nb_connection, uint64_t ef_construction, unsigned top_layer, uint64_t max_window,
                      std::string& err) = 0;
    // replace lists of the snapshot: records of rec_words() u32 = {node, layer, ids..., NO_POINT padding}.  The records are
    // packed straight into patch_buffer(n_records) -- pinned host memory of the backend, valid until the next call -- and
    // patch(n_records) sends them
    virtual uint32_t rec_words() const = 0;
    virtual uint32_t* patch_buffer(uint64_t n_records, std::string& err) = 0;
    virtual int patch(uint64_t n_records, std::string& err) = 0;
    // layer_mask: bit l set when some inserted point has level exactly l (search_layer returns nothing on other layers)
    virtual int search_window(uint32_t first, uint32_t count, uint32_t entry, uint32_t entry_level, uint32_t layer_mask,
                              const WindowSelect& select, WindowSearchResults& out, std::string& err) = 0;
};

class EdgeList;  // a neighbour list lock-free readers may copy while it is written (builder.cpp)

class GraphBuilder {
public:
    explicit GraphBuilder(const BuildParams& p);
    // Continue a reloaded index (HnswIo::load_hnsw gives a fully insertable Hnsw: load_point_indexation rebuilds the
    // layer generator from the dumped level scale, src/hnswio.rs:720-737, :1119-1178).  Builder ids = the dump's flat
    // ids; M, ef_construction, max_layer, the distance and the ABSOLUTE level scale come from the description; the
    // level stream restarts (a reloaded reference index also starts a fresh generator).
    GraphBuilder(const FlatIndex& loaded, bool fast_arithmetic);
    ~GraphBuilder();
    // insert n points (row-major n x d).  ids == nullptr: origin ids continue from nb_point().
    int insert_batch(const float* data, uint64_t n, uint64_t d, const uint64_t* ids, int nthreads, std::string& err);
    // The same with the searches of every insertion done on the device, window by window against a frozen snapshot
    // (points of one window do not see each other; windows grow with the graph: max(256, inserted / 8) up to max_window),
    // select_neighbours + list updates + reverse updates on the host cores.  window == 1 reproduces the serial insertion.
    int insert_batch_gpu(const float* data, uint64_t n, uint64_t d, const uint64_t* ids, int nthreads, BuildSearchBackend& dev,
                         uint64_t max_window, std::string& err);
    uint64_t nb_point() const { return n_; }
    uint64_t dimension() const { return d_; }
    const BuildParams& params() const { return p_; }
    // set by insert_batch_gpu when a device failure made the host builder finish the batch (the call still succeeded)
    const std::string& last_warning() const { return warning_; }
    // flatten into dump order
    void finalize(FlatIndex& out) const;

private:
    struct Node;
    struct Tls;
    static constexpr uint64_t CHUNK = 1u << 16;
    Node& node(uint32_t id) const;
    const float* vec(uint32_t id) const { return vecs_[id >> 16].get() + (uint64_t)(id & (CHUNK - 1)) * d_; }
    float eval(const float* a, const float* b) const;
    size_t draw_level();
    void insert_one(uint32_t id, Tls& t);
    int append_points(const float* data, uint64_t n, uint64_t d, const uint64_t* ids, std::string& err);
    void apply_window_point(uint32_t id, uint32_t wi, uint32_t frozen_entry, unsigned frozen_entry_level, uint32_t layer_mask,
                            const WindowSearchResults& r, uint64_t ef_c, Tls& t, std::vector<uint32_t>& dirty);
    void search_layer(const float* q, uint32_t entry, size_t ef, unsigned layer, Tls& t, std::vector<Edge>& out_sorted);
    void select_neighbours(const float* q, std::vector<Edge>& cands_sorted, size_t nb_asked, bool extend_asked,
                           unsigned layer, Tls& t, std::vector<Edge>& out);
    void reverse_update(uint32_t id, Tls& t);
    void read_list(uint32_t id, unsigned layer, std::vector<Edge>& out) const;
    EdgeList& wlist(Node& nd, unsigned layer) const;

    BuildParams p_;
    uint64_t n_ = 0, d_ = 0;
    unsigned max_layer_;
    unsigned dumped_nb_layer_ = 0;  // a reloaded index: the nb_layer of its dump (the level generator then uses NB_LAYER_MAX)
    double scale_;
    uint64_t rng_state_ = 397;
    mutable std::vector<std::unique_ptr<Node[]>> chunks_;
    std::vector<std::unique_ptr<float[]>> vecs_;
    std::array<std::atomic<uint64_t>, NB_LAYER_MAX> layer_inserted_{};  // points_by_layer[l].len() as seen by searches
    std::array<uint64_t, NB_LAYER_MAX> layer_rank_next_{};              // rank allocator (input order)
    std::mutex entry_mutex_;
    std::atomic<int64_t> entry_{-1};
    std::atomic<int> entry_level_{-1};
    std::string warning_;  // insert_batch_gpu: why the host builder finished the batch (empty: it did not)
};

// convenience: Hnsw::new + parallel_insert of a whole data set, flattened
int build_index(const float* data, uint64_t n, uint64_t d, const uint64_t* ids, const BuildParams& p, FlatIndex& out,
                std::string& err);

}  // namespace hnswgpu
