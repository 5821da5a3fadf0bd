// search_kernels.hpp -- plain argument structs and launch entry points shared by the host driver
// (search_device.hip) and the per-metric kernel translation units (search_kernels_tu.hip, compiled per metric and part).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include "search_device.hpp"

namespace hnswgpu {

constexpr uint32_t EXPANDED = 0x80000000u;  // flag bit on a result entry whose neighbour list was read (=> n < 2^31)
constexpr uint32_t EMPTY_SLOT = 0xFFFFFFFFu;
// visited-set representations (see visit_*)
constexpr int TABLE_LDS_CELL16 = 0;
constexpr int TABLE_LDS_CELL32 = 1;
constexpr int TABLE_GLOBAL_BITMAP = 2;

// kernel "metric" ids beyond the Dist ids of flat_index.hpp: the same distance summed in the crate's simdeez_f order
// (hnswgpu_set_arithmetic; search_kernels.inc, group_dist_simd8).  Only the search-path kernels exist for them.
constexpr int KM_SIMD8_FIRST = 7;
constexpr int KM_L2_SIMD8 = 7, KM_COSINE_SIMD8 = 8, KM_DOT_SIMD8 = 9, KM_L1_SIMD8 = 10;
constexpr int KM_COUNT = 11;
inline int simd8_kernel_metric(int dist) {  // -1: the metric has no SIMD-order variant (the probability distances)
    return dist == DIST_L2 ? KM_L2_SIMD8 : dist == DIST_COSINE ? KM_COSINE_SIMD8 : dist == DIST_DOT ? KM_DOT_SIMD8 : dist == DIST_L1 ? KM_L1_SIMD8 : -1;
}

constexpr uint32_t IDS_BYTES = 64 * 4;  // LDS: the compacted ids of one batch of neighbours

typedef unsigned long long hent_t;  // heap / log entry: {key f32 (high), id u32 (low)}

// the greedy descent of one query (hnsw_descend_kernel): where its search_layer starts
struct PreDescent {
    uint32_t pivot;      // entry point of the search layer (flat id)
    uint32_t dcur_bits;  // its distance to the query (f32 bit pattern; >= 0, so the patterns order like the values)
    uint32_t n_dist;     // distance evaluations of the descent (ids read = n_dist - 1)
    uint32_t n_expand;   // lists scanned
};
struct DescendArgs {
    const float* src;    // [nq][d] the caller's queries, unpadded (device memory, or pinned host memory read across PCIe)
    float* qpad;         // [nq][row_stride] zero padded copy for the search kernels
    PreDescent* pre;     // [nq]
    uint32_t nq;
    uint32_t tile_bytes;
    const double* nrm2;  // as in SearchArgs
    uint32_t* ctrl;      // the call's counters, zeroed by workgroup 0 (ctrl_words of them, <= 64)
    uint32_t ctrl_words;
    uint32_t pair;       // two queries per wavefront (hnsw_descend_pair_kernel: lists above the search layer of <= 16 ids, scalar arithmetic)
};

struct SearchArgs {
    const float* queries;   // [nq][row_stride], zero padded
    const uint32_t* qlist;  // optional: indices of the queries to run (scheduling order / retry pass), else nullptr
    uint32_t nq;            // number of work items
    uint32_t k;
    uint32_t ef;            // already max(ef_arg, k)
    uint32_t tbits;         // visited table = 1 << tbits cells
    uint32_t tile_bytes;    // LDS bytes in front of the id buffer: the staged query row (+ its squared norm for DistCosine)
    uint32_t idbits;        // ceil(log2(n))
    uint32_t restbits;      // CELL16: bits of the mixed id kept in the cell = idbits - (tbits - 3) (8-cell buckets)
    uint32_t* work_counter; // persistent-grid work queue head
    uint32_t* overflow_count;
    uint32_t* retry_out;    // queries whose visited table overflowed
    uint32_t* bitmap;       // [bitmap_blocks][bitmap_words]: per-workgroup visited bitmaps in HBM
    uint32_t bitmap_words;
    uint32_t bitmap_blocks; // workgroups with blockIdx.x < bitmap_blocks own a slice
    uint32_t* tie_list;     // queries whose answer depends on the reference's heap order and was not resolved in the launch
                            // (count at overflow_count + 3)
    hent_t* cand_scratch;   // strict ties: [gridDim.x][cand_cap] literal candidate heap beyond its LDS part
    uint32_t cand_cap;
    uint32_t cand_lds;      // strict ties: entries of the literal candidate heap kept in LDS (behind merge_list's buffer)
    uint32_t merge_entries; // LDS entries of merge_list's scatter buffer behind the visited table (64 S + 64, or 0: not used)
    uint32_t exact_first;   // strict ties, test hook: literal candidate heap from the first pop on
    uint32_t pair_ties_to_retry;  // hnsw_search_pair_kernel: a query that met equal distances goes to the retry list (strict calls), not to tie_list
    hent_t* oplog;          // strict ties: [gridDim.x][oplog_cap] per-workgroup log of heap operations
    uint32_t oplog_cap;
    const double* nrm2;     // DistCosine: [n] squared norm of every point (f64 left-to-right sum of f32 squares); nullptr: the
                            // norm sits in the last 8 bytes of every row's padding (norm_fits_row)
    uint64_t* out_ids;
    float* out_dists;
    uint8_t* out_layer;
    int32_t* out_rank;
    uint32_t* out_counts;
    const PreDescent* pre;  // [nq_total] what hnsw_descend_kernel left for every query: entry point of the search layer, its distance
    uint32_t* stats;        // [nq_total][8] = n_dist, n_expand, n_ids_read, status, t_start, t_end (10 ns ticks), bitmap_used,
                            // flags | (lists scanned by the descent << 8) | (n_dist of the descent << 16)
    // where answer j of query q goes: out_ids + (q k + j) id_stride bytes, out_dists + (q k + j) dist_stride, out_counts + q
    // count_stride.  8 / 4 / 4: three dense arrays.  16 / 16 / 16: the records the reference's FFI hands out -- out_ids is then
    // the `id` field of the first Neighbour_api {usize id; f32 d}, out_dists its `d` field, out_counts the low word of the
    // first Neighbourhood_api {i64 nbgh; ptr} (src/libext.rs:58-87) -- written in place, no unpacking pass on the host
    uint32_t id_stride, dist_stride, count_stride;
};

struct ExactArgs {
    hent_t* heaps;          // [gridDim.x][heap_stride]: return_points (ef + 2 entries) then candidate_points
    uint64_t heap_stride;   // entries per workgroup
    uint32_t cand_cap;      // capacity of candidate_points
    uint32_t r_lds_cap;     // entries of return_points kept in LDS
    uint32_t cand_lds;      // entries of candidate_points kept in LDS
    const uint32_t* allow;  // filtered search: one bit per flat id (nullptr = Hnsw::search, no filter)
};

// the snapshot's neighbour lists during construction: one fixed-stride array per layer (builder ids, EMPTY padded)
struct BuildLists {
    uint32_t* lists[NB_LAYER_MAX];   // [n][stride[l]]; nullptr above the highest layer of the build
    uint32_t stride[NB_LAYER_MAX];
};
struct BuildArgs {
    const float* vec;         // [n][row_stride] every vector of the build, builder order (= insertion order), zero padded
    uint32_t row_stride;
    const uint32_t* lists[NB_LAYER_MAX];
    uint32_t stride[NB_LAYER_MAX];
    const uint8_t* level;     // [n] level of every point
    const uint32_t* slot0;    // [count] output slot of the window's i-th point at layer 0 (layer l: slot0[i] + l)
    uint32_t first, count;    // the window: builder ids first .. first + count - 1
    uint32_t entry, entry_level;  // the frozen entry point
    uint32_t layer_mask;      // bit l: some inserted point has level exactly l (points_by_layer[l] is not empty)
    uint32_t ef_c;
    uint32_t tbits, idbits, restbits, tile_bytes;
    uint32_t* bitmap;
    uint32_t bitmap_words, bitmap_blocks;
    uint32_t* work_counter;
    uint32_t* fail_count;
    uint32_t* out_ids;        // [slots][ef_c] candidates of an ef_construction search, ascending distance
    float* out_d;
    uint32_t* out_n;          // [slots]
    uint32_t* hit_ids;        // [count][NB_LAYER_MAX] ef = 1 result of the layers above the point's level (EMPTY_SLOT: none)
    float* hit_d;
    const double* nrm2;       // DistCosine: as in SearchArgs (nullptr: in the rows)
};

// select_neighbours on the device (hnsw_build_select_kernel): one wavefront per (window point, layer) slot
struct SelectArgs {
    const float* vec;         // as BuildArgs
    uint32_t row_stride;
    uint32_t tile_bytes;
    const double* nrm2;
    uint32_t* cand_ids;       // [slots][ef_c] what the searches of the window found (ascending distance); scratch afterwards
    const float* cand_d;
    const uint32_t* cand_n;   // [slots]
    uint32_t ef_c;
    const uint16_t* slot_nb;  // [slots] neighbours asked for the slot: 2 M at layer 0, M above (src/hnsw.rs:1170-1176)
    uint32_t n_slots;
    uint32_t sel_stride;      // entries per slot in sel_ids / sel_d (>= every slot_nb)
    uint32_t keep_pruned;
    uint32_t* sel_ids;        // [slots][sel_stride] the selected neighbours, in selection order
    float* sel_d;
    uint32_t* sel_n;          // [slots]
    uint32_t* work_counter;
};

// The lane lab (lane_lab.inc, hnswgpu_lane_lab): a script of wave-level operations run by one wavefront, for the tests
constexpr uint32_t LAB_PUSH = 1, LAB_POP = 2, LAB_PUSH_LANES = 3, LAB_INSERT = 4, LAB_MERGE = 5, LAB_BATCH = 6, LAB_VISIT = 7;
struct LaneLabArgs {
    const uint32_t* ops;    // [n_ops][4] = {op, a, b, c}
    uint32_t n_ops;
    const uint32_t* lanes;  // [n_sets][64][2] = {f32 bits, id}: the lane vectors (de / idc) of LAB_PUSH_LANES / LAB_MERGE / LAB_BATCH
    uint32_t mode;          // 0 memory heap, 1 register heap, 2 result set, 3 visited table
    uint32_t p0, p1, p2;    // mode 0: LDS entries, pop variant; 1: slots per lane; 2: slots per lane, ef; 3: tbits, idbits, restbits
    hent_t* scratch;        // mode 0: the heap's global slice
    uint32_t scratch_cap;
    uint32_t* out;          // out[0] = words produced (itself included), then the results in script order, then the final state
    uint32_t out_cap;
};
hipError_t launch_lane_lab(hipStream_t stream, size_t lds, const LaneLabArgs& a);

// Events bound to ONE kernel launch (hipExtLaunchKernelGGL): they take the kernel's own start / end time -- what rocprofv3's kernel
// trace reports -- instead of the time of a marker packet in front of / behind it.  (The idle 6 us between the order kernel and
// the search kernel are not the markers': they stay, bound or not -- tools/GPU_CALLS.md round 6, call 21.)  Null members: no event.
struct LaunchEvents {
    hipEvent_t start = nullptr;
    hipEvent_t stop = nullptr;
};

// Three translation units per metric instantiate the kernels (search_kernels_tu.hip with -DHNSW_THIS_METRIC / -DHNSW_PART:
// strict search kernels, lean search kernels, everything else) -- keeps the build parallel and the objects small.
struct KernelSet {
    // search kernel: S in {1,2,4,16} result slots per lane, visited-table kind, strict (decisions that depend on the
    // reference's heap order are resolved with literal heaps inside the launch) or lean (such queries are only flagged)
    hipError_t (*launch_search)(int slots, int table, bool strict, uint32_t grid, size_t lds, hipStream_t stream,
                                const DeviceIndexView& ix, const SearchArgs& a, LaunchEvents ev);
    hipError_t (*occupancy)(int slots, int table, bool strict, size_t lds, int* per_cu);
    // two queries per wavefront (hnsw_search_pair_kernel: ef <= 128, lists of <= 64 ids, 16-bit-cell tables, scalar arithmetic)
    hipError_t (*launch_pair)(uint32_t grid, size_t lds, hipStream_t stream, const DeviceIndexView& ix, const SearchArgs& a, LaunchEvents ev);
    hipError_t (*pair_occupancy)(size_t lds, int* per_cu);
    hipError_t (*launch_exact)(int ns, uint32_t grid, size_t lds, hipStream_t stream, const DeviceIndexView& ix,
                               const SearchArgs& a, const ExactArgs& x);
    hipError_t (*exact_occupancy)(int ns, size_t lds, int* per_cu);
    // first kernel of a call: queries padded, greedy descent of every query (pre[]); then, batch scheduling, the queries in
    // descending order of the descent's distance
    hipError_t (*launch_descend)(uint32_t grid, hipStream_t stream, const DeviceIndexView& ix, const DescendArgs& a, LaunchEvents ev);
    hipError_t (*descend_occupancy)(size_t lds, bool pair, int* per_cu);
    hipError_t (*launch_order)(hipStream_t stream, const PreDescent* pre, uint32_t n, uint32_t* order);
    // arithmetic tests: out[q][r] = dist(queries[q], rows[r]) through batch_dist, rows in batches of nf; or, pairs:
    // out[q] = dist(queries[q], rows[q])
    hipError_t (*launch_eval_matrix)(hipStream_t stream, const float* queries, uint32_t nq, const float* rows, uint32_t n_rows,
                                     const double* nrm2, float* out, uint32_t row_stride, uint32_t d, uint32_t nf, bool pairs);
    // construction: the searches of insert_slice for a window of points (hnsw_build_search_kernel)
    hipError_t (*launch_build_search)(int slots, uint32_t grid, size_t lds, hipStream_t stream, const BuildArgs& a);
    hipError_t (*build_occupancy)(int slots, size_t lds, int* per_cu);
    // construction: select_neighbours for every slot of a window (hnsw_build_select_kernel)
    hipError_t (*launch_build_select)(uint32_t grid, size_t lds, hipStream_t stream, const SelectArgs& a);
};
// merge_list (the accept rule for a whole neighbour list at once) is used up to this many result slots per lane; the launch
// gets 64 S + 64 LDS entries for its scatter
#ifndef HNSW_MERGE_SMAX
#define HNSW_MERGE_SMAX 4
#endif
#ifndef HNSW_MERGE_LEAN_SMAX
#define HNSW_MERGE_LEAN_SMAX 2
#endif
// LDS in front of the id buffer: the query row; DistCosine keeps the query's squared norm (f64) behind it
inline uint32_t tile_bytes_for(int metric, uint32_t row_stride) {
    return ((row_stride * 4u + 15u) & ~15u) + (metric == DIST_COSINE || metric == KM_COSINE_SIMD8 ? 16u : 0u);
}
// LDS of a hnsw_search_pair_kernel workgroup: two query rows, two id buffers, two visited tables of 2^tb 16-bit cells, two mirrors of
// the result array (ef entries), the candidates and the cr list of a round
inline size_t pair_lds_bytes(uint32_t tile_bytes, uint32_t tb, uint32_t ef) {
    return 2u * (size_t)tile_bytes + 2u * IDS_BYTES + 2u * ((size_t)2 << tb) + 2u * (size_t)((ef + 2u) & ~1u) * sizeof(hent_t) + 2u * 16u * sizeof(hent_t) + 2u * 16u * 4u;
}
// kernels_for<METRIC>(): defined in part 2 of that metric's translation units (search_kernels_tu.hip)
template <int METRIC> const KernelSet& kernels_for();
template <> const KernelSet& kernels_for<DIST_L2>();
template <> const KernelSet& kernels_for<DIST_COSINE>();
template <> const KernelSet& kernels_for<DIST_DOT>();
template <> const KernelSet& kernels_for<DIST_L1>();
template <> const KernelSet& kernels_for<DIST_HELLINGER>();
template <> const KernelSet& kernels_for<DIST_JEFFREYS>();
template <> const KernelSet& kernels_for<DIST_JENSENSHANNON>();
template <> const KernelSet& kernels_for<KM_L2_SIMD8>();
template <> const KernelSet& kernels_for<KM_COSINE_SIMD8>();
template <> const KernelSet& kernels_for<KM_DOT_SIMD8>();
template <> const KernelSet& kernels_for<KM_L1_SIMD8>();
// metric-independent helpers (instantiated once, in part 2 of the L2 units)
hipError_t launch_allow_bitmap(hipStream_t stream, const uint64_t* origin_id, uint32_t n, const uint64_t* ids, uint64_t m, uint32_t* allow);
// DistCosine: every point's squared norm (the crate's arithmetic) into out[n] -- or, out == nullptr, into the last 8 bytes of
// each row's padding (norm_fits_row)
hipError_t launch_row_sq_norms(hipStream_t stream, float* vec, double* out, uint32_t n, uint32_t d, uint32_t row_stride);
// a DistCosine row keeps its f64 squared norm inside its own 128-byte-padded row when the padding has two free floats
inline bool norm_fits_row(int metric, uint32_t d, uint32_t row_stride) { return metric == DIST_COSINE && row_stride >= d + 2u; }
hipError_t launch_scatter_lists(hipStream_t stream, const uint32_t* upd, uint32_t n_upd, uint32_t rec_words, const BuildLists& lists);

}  // namespace hnswgpu
