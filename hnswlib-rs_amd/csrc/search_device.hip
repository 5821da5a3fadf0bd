// search_device.hip -- CDNA4 (gfx950) kernels for the batched-search hot path of hnsw_rs and
// the host driver that launches them.  Written for MI355X only: wave64, LDS, 8 XCDs.
//
// Reference path (file:line under /root/reference):
//   Hnsw::parallel_search          src/hnsw.rs:1612-1635   -> one wavefront per query, persistent grid
//   Hnsw::search_filter(None)      src/hnsw.rs:1487-1580   -> descent prologue + result epilogue
//   Hnsw::search_layer             src/hnsw.rs:922-1064    -> expansion loop (visited set in LDS,
//                                                             ef-bounded result/candidate set in VGPRs)
//   Distance<f32>::eval            anndists 0.1            -> dist_row<METRIC>: one LANE per neighbour,
//                                                             summed left-to-right exactly like the
//                                                             crate's scalar build (bit-identical)
//
// Arithmetic contract: every distance is accumulated in the reference's order (sequential over
// the vector index, no FMA contraction), so ids AND f32 distances equal the CPU oracle bit for
// bit on tie-free inputs.  Build with -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "hnswio.hpp"
#include "search_device.hpp"

#pragma clang fp contract(off)

namespace hnswgpu {

namespace {

constexpr uint32_t EXPANDED = 0x80000000u;  // flag bit on a result entry whose neighbour list was read
constexpr uint32_t EMPTY_SLOT = 0xFFFFFFFFu;
constexpr int TABLE_LDS_HASH = 0;
constexpr int TABLE_GLOBAL_BITMAP = 1;

struct SearchArgs {
    const float* queries;   // [nq][row_stride], zero padded
    const uint32_t* qlist;  // optional: indices of the queries to run (retry pass), else nullptr
    uint32_t nq;            // number of work items
    uint32_t k;
    uint32_t ef;            // already max(ef_arg, k)
    uint32_t hash_bits;     // LDS table = 1 << hash_bits slots
    uint32_t* work_counter; // persistent-grid work queue head
    uint32_t* overflow_count;
    uint32_t* retry_out;    // queries whose visited table overflowed
    uint32_t* bitmap;       // TABLE_GLOBAL_BITMAP: [gridDim.x][bitmap_words]
    uint32_t bitmap_words;
    uint64_t* out_ids;
    float* out_dists;
    uint8_t* out_layer;
    int32_t* out_rank;
    uint32_t* out_counts;
    uint32_t* stats;        // [nq_total][4] = n_dist, n_expand, n_ids_read, status
};

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint32_t popc64(unsigned long long m) { return (uint32_t)__popcll(m); }
__device__ __forceinline__ int ctz64(unsigned long long m) { return __ffsll((long long)m) - 1; }

// ---------------------------------------------------------------------------------------
// Distance<f32>::eval, one lane per (query, row) pair.  `q` is wave-uniform, `row` per lane.
// Rows and queries are zero padded to row_stride, and x + 0 == x, so running over the padding
// leaves every sum bit-identical to the d-term sum of the reference.
// ---------------------------------------------------------------------------------------
template <int METRIC>
__device__ __forceinline__ float dist_row(const float4* __restrict__ q, const float4* __restrict__ row, uint32_t nchunk) {
    if constexpr (METRIC == DIST_L2) {
        float acc = 0.f;
        for (uint32_t c = 0; c < nchunk; ++c) {
            float4 r = row[c], s = q[c];
            float t;
            t = s.x - r.x; acc = acc + t * t;
            t = s.y - r.y; acc = acc + t * t;
            t = s.z - r.z; acc = acc + t * t;
            t = s.w - r.w; acc = acc + t * t;
        }
        return __builtin_sqrtf(acc);
    } else if constexpr (METRIC == DIST_L1) {
        float acc = 0.f;
        for (uint32_t c = 0; c < nchunk; ++c) {
            float4 r = row[c], s = q[c];
            acc = acc + fabsf(s.x - r.x);
            acc = acc + fabsf(s.y - r.y);
            acc = acc + fabsf(s.z - r.z);
            acc = acc + fabsf(s.w - r.w);
        }
        return acc;
    } else if constexpr (METRIC == DIST_DOT) {
        float acc = 0.f;
        for (uint32_t c = 0; c < nchunk; ++c) {
            float4 r = row[c], s = q[c];
            acc = acc + s.x * r.x;
            acc = acc + s.y * r.y;
            acc = acc + s.z * r.z;
            acc = acc + s.w * r.w;
        }
        return fmaxf(1.f - acc, 0.f);
    } else {  // DIST_COSINE: f32 products widened to f64, three f64 running sums
        double s0 = 0., s1 = 0., s2 = 0.;
        for (uint32_t c = 0; c < nchunk; ++c) {
            float4 r = row[c], s = q[c];
            s0 = s0 + (double)(s.x * r.x); s1 = s1 + (double)(s.x * s.x); s2 = s2 + (double)(r.x * r.x);
            s0 = s0 + (double)(s.y * r.y); s1 = s1 + (double)(s.y * s.y); s2 = s2 + (double)(r.y * r.y);
            s0 = s0 + (double)(s.z * r.z); s1 = s1 + (double)(s.z * s.z); s2 = s2 + (double)(r.z * r.z);
            s0 = s0 + (double)(s.w * r.w); s1 = s1 + (double)(s.w * s.w); s2 = s2 + (double)(r.w * r.w);
        }
        if (s1 > 0. && s2 > 0.) {
            double du = 1. - s0 / __builtin_sqrt(s1 * s2);
            return (float)fmax(du, 0.);
        }
        return 0.f;
    }
}

// ---------------------------------------------------------------------------------------
// Visited set (reference: hashbrown::HashMap<PointId, Arc<Point>>, src/hnsw.rs:955-956, :1016-1017).
//   TABLE_LDS_HASH     : open-addressing table of flat ids in LDS, one table per wavefront.
//   TABLE_GLOBAL_BITMAP: one bit per point in a per-workgroup slice of HBM (exact, cannot overflow);
//                        the fallback when a query visits more points than the LDS table holds.
// Both return true when `id` was NOT yet visited and mark it.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool visit_lds(uint32_t* tab, uint32_t bits, uint32_t id) {
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t h = (id * 0x9E3779B1u) >> (32 - bits);
    for (;;) {
        uint32_t old = atomicCAS(&tab[h], EMPTY_SLOT, id);
        if (old == EMPTY_SLOT) return true;
        if (old == id) return false;
        h = (h + 1) & mask;
    }
}
__device__ __forceinline__ bool visit_bitmap(uint32_t* bm, uint32_t id) {
    uint32_t bit = 1u << (id & 31);
    uint32_t old = atomicOr(&bm[id >> 5], bit);
    return (old & bit) == 0;
}

// ---------------------------------------------------------------------------------------
// Result set R (reference: return_points, a max-heap capped at ef, plus candidate_points).
// Kept as ONE array sorted ascending by distance, entry j in VGPR slot j/64 of lane j%64, with an
// EXPANDED flag: the candidates of the reference are exactly the not-yet-expanded members of R
// (an entry evicted from R can only terminate the loop when popped: SURVEY.md section 3.1).
// Insertion keeps arrival order among equal distances.
// ---------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void r_insert(float (&rd)[S], uint32_t (&ri)[S], uint32_t& len, uint32_t ef, float xd,
                                         uint32_t xi, int lane) {
    uint32_t pos = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        uint32_t j = (uint32_t)s * 64u + (uint32_t)lane;
        pos += popc64(__ballot(j < len && rd[s] <= xd));
    }
#pragma unroll
    for (int s = S - 1; s >= 0; --s) {
        float pd = __shfl_up(rd[s], 1);
        uint32_t pi = (uint32_t)__shfl_up((int)ri[s], 1);
        if (s > 0) {
            float wd = readlane_f(rd[s - 1], 63);
            uint32_t wi = readlane_u(ri[s - 1], 63);
            if (lane == 0) { pd = wd; pi = wi; }
        }
        uint32_t j = (uint32_t)s * 64u + (uint32_t)lane;
        if (j > pos) { rd[s] = pd; ri[s] = pi; }
        else if (j == pos) { rd[s] = xd; ri[s] = xi; }
    }
    len = len + 1 > ef ? ef : len + 1;  // the entry pushed past ef-1 is the evicted worst (src/hnsw.rs:1051-1053)
}
template <int S>
__device__ __forceinline__ float r_worst(const float (&rd)[S], uint32_t len) {
    float w = 0.f;
    const uint32_t j = len - 1;
#pragma unroll
    for (int s = 0; s < S; ++s)
        if ((j >> 6) == (uint32_t)s) w = readlane_f(rd[s], (int)(j & 63));
    return w;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}

// ---------------------------------------------------------------------------------------
// The search kernel: one wavefront (= one 64-thread workgroup) per query, persistent grid pulling
// query indices from a global counter.
// ---------------------------------------------------------------------------------------
template <int METRIC, int S, int TABLE>
__global__ __launch_bounds__(64) void hnsw_search_kernel(DeviceIndexView ix, SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_table[];
    const int lane = (int)threadIdx.x;
    const uint32_t nchunk = ix.row_stride >> 2;
    const uint32_t table_slots = 1u << a.hash_bits;
    const uint32_t table_limit = table_slots - (table_slots >> 2);  // stop inserting at 75 % load
    uint32_t* bitmap = TABLE == TABLE_GLOBAL_BITMAP ? a.bitmap + (size_t)blockIdx.x * a.bitmap_words : nullptr;

    for (;;) {
        uint32_t wi = 0;
        if (lane == 0) wi = atomicAdd(a.work_counter, 1u);
        wi = readlane_u(wi, 0);
        if (wi >= a.nq) break;
        const uint32_t q = a.qlist ? a.qlist[wi] : wi;
        const float4* qv = reinterpret_cast<const float4*>(a.queries + (size_t)q * ix.row_stride);

        // ---- reset the visited set
        if constexpr (TABLE == TABLE_LDS_HASH) {
            for (uint32_t i = (uint32_t)lane; i < table_slots; i += 64) lds_table[i] = EMPTY_SLOT;
        } else {
            for (uint32_t i = (uint32_t)lane; i < a.bitmap_words; i += 64) bitmap[i] = 0u;
        }
        __syncthreads();

        uint32_t n_dist = 0, n_expand = 0, n_ids = 0, status = 0;

        // ---- greedy descent: ONE scan of the pivot's list per layer (src/hnsw.rs:1506-1529)
        uint32_t pivot = ix.entry;
        float dcur = dist_row<METRIC>(qv, reinterpret_cast<const float4*>(ix.vec + (size_t)pivot * ix.row_stride), nchunk);
        n_dist += 1;
        for (int layer = (int)ix.entry_level; layer >= 1; --layer) {
            uint32_t b = 0, e = 0;
            if ((uint32_t)layer <= ix.n_up_layers) {
                const uint32_t* ptr = ix.up_ptr + (size_t)(layer - 1) * ((size_t)ix.n + 1);
                b = ptr[pivot];
                e = ptr[pivot + 1];
            }
            n_expand += 1;
            n_ids += e - b;
            float best = INFINITY;
            uint32_t best_id = pivot;
            for (uint32_t base = b; base < e; base += 64) {
                const uint32_t j = base + (uint32_t)lane;
                const bool valid = j < e;
                const uint32_t id = valid ? ix.up_ids[j] : 0u;
                float dl = INFINITY;
                if (valid) dl = dist_row<METRIC>(qv, reinterpret_cast<const float4*>(ix.vec + (size_t)id * ix.row_stride), nchunk);
                n_dist += popc64(__ballot(valid));
                const float m = wave_min(dl);
                const unsigned long long eq = __ballot(valid && dl == m);
                if (eq != 0ull && m < best) {  // strict '<': the first index wins ties (:1519)
                    best = m;
                    best_id = readlane_u(id, ctz64(eq));
                }
            }
            if (best < dcur) {  // pivot replaced once per layer, only if strictly better (:1519-1528)
                dcur = best;
                pivot = best_id;
            }
        }

        // ---- search_layer at the lowest non-empty layer (src/hnsw.rs:1542, :922-1064)
        float rd[S];
        uint32_t ri[S];
#pragma unroll
        for (int s = 0; s < S; ++s) { rd[s] = 0.f; ri[s] = 0u; }
        uint32_t len = 1;
        if (lane == 0) { rd[0] = dcur; ri[0] = pivot; }  // dist_to_entry_point == eval(q, pivot) (:952)
        uint32_t n_visited = 1;
        if (lane == 0) {
            if constexpr (TABLE == TABLE_LDS_HASH) visit_lds(lds_table, a.hash_bits, pivot);
            else visit_bitmap(bitmap, pivot);
        }
        __syncthreads();

        for (;;) {
            // c = nearest unexpanded member of R (candidate_points.pop(), :971)
            int cs = -1, cl = 0;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const uint32_t j = (uint32_t)s * 64u + (uint32_t)lane;
                const unsigned long long m = __ballot(j < len && (ri[s] & EXPANDED) == 0u);
                if (cs < 0 && m != 0ull) { cs = s; cl = ctz64(m); }
            }
            if (cs < 0) break;  // every remaining candidate is farther than R's worst (:981-993)
            uint32_t c = 0;
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (s == cs) {
                    c = readlane_u(ri[s], cl);
                    if (lane == cl) ri[s] |= EXPANDED;
                }
            n_expand += 1;
            const uint32_t* nrow = ix.nbr0 + (size_t)c * ix.deg_stride;
            for (uint32_t base = 0; base < ix.deg_stride; base += 64) {
                const uint32_t j = base + (uint32_t)lane;
                const uint32_t id = j < ix.deg_stride ? nrow[j] : EMPTY_SLOT;
                const bool valid = id != EMPTY_SLOT;
                const unsigned long long vm = __ballot(valid);
                if (vm == 0ull) break;  // lists are padded at the end only
                n_ids += popc64(vm);
                if constexpr (TABLE == TABLE_LDS_HASH) {
                    if (n_visited + 64 > table_limit) { status = 1; break; }
                }
                bool fresh = false;
                if (valid) {
                    if constexpr (TABLE == TABLE_LDS_HASH) fresh = visit_lds(lds_table, a.hash_bits, id);
                    else fresh = visit_bitmap(bitmap, id);
                }
                const unsigned long long fm = __ballot(fresh);
                n_visited += popc64(fm);
                n_dist += popc64(fm);
                float de = INFINITY;
                if (fresh) de = dist_row<METRIC>(qv, reinterpret_cast<const float4*>(ix.vec + (size_t)id * ix.row_stride), nchunk);
                // accept rule applied sequentially in list order (:1028-1053)
                float worst = r_worst<S>(rd, len);
                unsigned long long cand = __ballot(fresh && (len < a.ef || de < worst));
                while (cand != 0ull) {
                    const int jl = ctz64(cand);
                    cand &= cand - 1ull;
                    const float xd = readlane_f(de, jl);
                    if (xd < worst || len < a.ef) {
                        const uint32_t xi = readlane_u(id, jl);
                        r_insert<S>(rd, ri, len, a.ef, xd, xi, lane);
                        worst = r_worst<S>(rd, len);
                    }
                }
            }
            if (status != 0) break;
        }

        // ---- into_sorted_vec + truncate to min(knbn, ef, len) (:1544-1547, :1567-1578)
        if (status == 0) {
            const uint32_t cnt = len < a.k ? len : a.k;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const uint32_t j = (uint32_t)s * 64u + (uint32_t)lane;
                if (j < a.k) {
                    const size_t o = (size_t)q * a.k + j;
                    if (j < cnt) {
                        const uint32_t flat = ri[s] & ~EXPANDED;
                        uint32_t l = 0;
                        while (l + 1 < NB_LAYER_MAX && flat >= ix.layer_offset[l + 1]) ++l;
                        a.out_ids[o] = ix.origin_id[flat];
                        a.out_dists[o] = rd[s];
                        if (a.out_layer) a.out_layer[o] = (uint8_t)l;
                        if (a.out_rank) a.out_rank[o] = (int32_t)(flat - ix.layer_offset[l]);
                    } else {
                        a.out_ids[o] = 0ull;
                        a.out_dists[o] = 0.f;
                        if (a.out_layer) a.out_layer[o] = 0;
                        if (a.out_rank) a.out_rank[o] = 0;
                    }
                }
            }
            if (lane == 0) a.out_counts[q] = cnt;
        } else if (lane == 0) {
            const uint32_t slot = atomicAdd(a.overflow_count, 1u);
            a.retry_out[slot] = q;
        }
        if (lane == 0) {
            uint32_t* st = a.stats + (size_t)q * 4;
            st[0] = n_dist; st[1] = n_expand; st[2] = n_ids; st[3] = status;
        }
        __syncthreads();
    }
}

// queries [nq][d] -> [nq][row_stride] zero padded
__global__ void pad_queries_kernel(const float* __restrict__ src, float* __restrict__ dst, uint32_t nq, uint32_t d,
                                   uint32_t row_stride) {
    const size_t total = (size_t)nq * row_stride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / row_stride), c = (uint32_t)(i % row_stride);
        dst[i] = c < d ? src[(size_t)r * d + c] : 0.f;
    }
}

template <int METRIC>
__global__ void eval_pairs_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                  uint32_t n, uint32_t row_stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = dist_row<METRIC>(reinterpret_cast<const float4*>(a + (size_t)i * row_stride),
                              reinterpret_cast<const float4*>(b + (size_t)i * row_stride), row_stride >> 2);
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
            return ERR_DEVICE;                                                                 \
        }                                                                                      \
    } while (0)

using KernelFn = void (*)(DeviceIndexView, SearchArgs);

template <int METRIC, int TABLE>
KernelFn pick_slots(int slots) {
    switch (slots) {
        case 1: return hnsw_search_kernel<METRIC, 1, TABLE>;
        case 2: return hnsw_search_kernel<METRIC, 2, TABLE>;
        case 4: return hnsw_search_kernel<METRIC, 4, TABLE>;
        case 8: return hnsw_search_kernel<METRIC, 8, TABLE>;
        default: return hnsw_search_kernel<METRIC, 16, TABLE>;
    }
}
template <int TABLE>
KernelFn pick_metric(int metric, int slots) {
    switch (metric) {
        case DIST_L2: return pick_slots<DIST_L2, TABLE>(slots);
        case DIST_COSINE: return pick_slots<DIST_COSINE, TABLE>(slots);
        case DIST_DOT: return pick_slots<DIST_DOT, TABLE>(slots);
        default: return pick_slots<DIST_L1, TABLE>(slots);
    }
}
KernelFn pick_kernel(int metric, int slots, int table) {
    return table == TABLE_LDS_HASH ? pick_metric<TABLE_LDS_HASH>(metric, slots) : pick_metric<TABLE_GLOBAL_BITMAP>(metric, slots);
}

uint32_t ceil_log2(uint64_t x) {
    uint32_t b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}

}  // namespace

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

DeviceIndex::~DeviceIndex() { release(); }

void DeviceIndex::release() {
    if (device_ >= 0) (void)hipSetDevice(device_);
    void** ptrs[] = {&d_vec_, &d_nbr0_, &d_up_ptr_, &d_up_ids_, &d_origin_, &d_qpad_, &d_ctrl_, &d_retry_[0], &d_retry_[1],
                     &d_stats_, &d_bitmap_, &d_hostio_[0], &d_hostio_[1], &d_hostio_[2], &d_hostio_[3], &d_hostio_[4]};
    for (void** p : ptrs)
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (ev_start_) { (void)hipEventDestroy((hipEvent_t)ev_start_); ev_start_ = nullptr; }
    if (ev_stop_) { (void)hipEventDestroy((hipEvent_t)ev_stop_); ev_stop_ = nullptr; }
    ready_ = false;
}

int DeviceIndex::upload(const FlatIndex& x, int device, std::string& err) {
    if (x.n == 0 || x.entry_flat == NO_POINT) { err = "cannot upload an empty index"; return ERR_EMPTY; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { err = "no HIP device visible (a gfx950 GPU is required; there is no CPU fallback)"; return ERR_DEVICE; }
    if (device < 0 || device >= ndev) { err = "bad device ordinal"; return ERR_ARG; }
    release();
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    num_cu_ = prop.multiProcessorCount;
    device_ = device;
    dist_ = x.dist;

    const uint64_t n = x.n, d = x.dimension;
    DeviceIndexView v{};
    v.n = (uint32_t)n;
    v.d = (uint32_t)d;
    v.row_stride = (uint32_t)((d + 31) / 32 * 32);  // 128-byte lines
    v.entry = x.entry_flat;
    v.entry_level = x.layer_of(x.entry_flat);
    v.search_layer = x.layer_to_search();
    for (unsigned l = 0; l <= NB_LAYER_MAX; ++l) v.layer_offset[l] = (uint32_t)x.layer_offset[l];

    // vectors, padded rows
    {
        std::vector<float> pad((size_t)n * v.row_stride, 0.f);
        for (uint64_t f = 0; f < n; ++f) std::memcpy(pad.data() + f * v.row_stride, x.vectors.data() + f * d, d * sizeof(float));
        HIP_TRY(hipMalloc(&d_vec_, pad.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(d_vec_, pad.data(), pad.size() * sizeof(float), hipMemcpyHostToDevice));
        bytes_ += pad.size() * sizeof(float);
    }
    // search-layer lists, fixed stride ("padded CSR": row_ptr is implicit, one aligned row per point)
    {
        const unsigned sl = v.search_layer;
        uint64_t maxdeg = 1;
        for (uint64_t f = 0; f < n; ++f)
            maxdeg = std::max<uint64_t>(maxdeg, x.nbr_ptr[f * NB_LAYER_MAX + sl + 1] - x.nbr_ptr[f * NB_LAYER_MAX + sl]);
        v.deg_stride = (uint32_t)((maxdeg + 15) / 16 * 16);
        std::vector<uint32_t> ell((size_t)n * v.deg_stride, EMPTY_SLOT);
        for (uint64_t f = 0; f < n; ++f) {
            uint64_t b = x.nbr_ptr[f * NB_LAYER_MAX + sl], e = x.nbr_ptr[f * NB_LAYER_MAX + sl + 1];
            std::memcpy(ell.data() + f * v.deg_stride, x.nbr_flat.data() + b, (e - b) * sizeof(uint32_t));
        }
        HIP_TRY(hipMalloc(&d_nbr0_, ell.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(d_nbr0_, ell.data(), ell.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        bytes_ += ell.size() * sizeof(uint32_t);
    }
    // upper layers (>= 1): CSR per layer over all flat ids (lists may exist above a point's own level)
    {
        unsigned top = 0;
        for (uint64_t f = 0; f < n; ++f)
            for (unsigned l = NB_LAYER_MAX - 1; l > top; --l)
                if (x.nbr_ptr[f * NB_LAYER_MAX + l + 1] > x.nbr_ptr[f * NB_LAYER_MAX + l]) { top = l; break; }
        v.n_up_layers = top;
        std::vector<uint32_t> ptr((size_t)std::max(1u, top) * (n + 1), 0u);
        std::vector<uint32_t> ids;
        for (unsigned l = 1; l <= top; ++l) {
            uint32_t* p = ptr.data() + (size_t)(l - 1) * (n + 1);
            for (uint64_t f = 0; f < n; ++f) {
                p[f] = (uint32_t)ids.size();
                uint64_t b = x.nbr_ptr[f * NB_LAYER_MAX + l], e = x.nbr_ptr[f * NB_LAYER_MAX + l + 1];
                ids.insert(ids.end(), x.nbr_flat.begin() + b, x.nbr_flat.begin() + e);
            }
            p[n] = (uint32_t)ids.size();
        }
        if (ids.empty()) ids.push_back(0);
        HIP_TRY(hipMalloc(&d_up_ptr_, ptr.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(d_up_ptr_, ptr.data(), ptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&d_up_ids_, ids.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(d_up_ids_, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        bytes_ += (ptr.size() + ids.size()) * sizeof(uint32_t);
    }
    HIP_TRY(hipMalloc(&d_origin_, n * sizeof(uint64_t)));
    HIP_TRY(hipMemcpy(d_origin_, x.origin_id.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice));
    bytes_ += n * sizeof(uint64_t);
    HIP_TRY(hipMalloc(&d_ctrl_, 64));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    ev_start_ = e0;
    ev_stop_ = e1;

    v.vec = static_cast<const float*>(d_vec_);
    v.nbr0 = static_cast<const uint32_t*>(d_nbr0_);
    v.up_ptr = static_cast<const uint32_t*>(d_up_ptr_);
    v.up_ids = static_cast<const uint32_t*>(d_up_ids_);
    v.origin_id = static_cast<const uint64_t*>(d_origin_);
    v_ = v;
    ready_ = true;
    return OK;
}

int DeviceIndex::ensure_workspace(uint64_t nq, uint64_t /*k*/, std::string& err) {
    const uint64_t qpad_need = nq * v_.row_stride * sizeof(float);
    if (qpad_need > qpad_cap_) {
        if (d_qpad_) (void)hipFree(d_qpad_);
        d_qpad_ = nullptr;
        HIP_TRY(hipMalloc(&d_qpad_, qpad_need));
        qpad_cap_ = qpad_need;
    }
    if (nq > retry_cap_) {
        for (int i = 0; i < 2; ++i) {
            if (d_retry_[i]) (void)hipFree(d_retry_[i]);
            d_retry_[i] = nullptr;
            HIP_TRY(hipMalloc(&d_retry_[i], nq * sizeof(uint32_t)));
        }
        retry_cap_ = nq;
    }
    if (nq * 4 * sizeof(uint32_t) > stats_cap_) {
        if (d_stats_) (void)hipFree(d_stats_);
        d_stats_ = nullptr;
        HIP_TRY(hipMalloc(&d_stats_, nq * 4 * sizeof(uint32_t)));
        stats_cap_ = nq * 4 * sizeof(uint32_t);
    }
    return OK;
}

int DeviceIndex::search_device(const float* d_queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef_arg,
                               uint64_t* d_out_ids, float* d_out_dists, uint8_t* d_out_layer, int32_t* d_out_rank,
                               uint32_t* d_out_counts, uint32_t* d_stats, void* stream_v, std::string& err) {
    if (!ready_) { err = "index is not resident on a device: call hnswgpu_upload first"; return ERR_DEVICE; }
    if (d != v_.d) { err = "query dimension differs from the index dimension"; return ERR_ARG; }
    if (nq == 0) { last_ms_ = 0; last_launches_ = 0; return OK; }
    if (!d_queries || !d_out_ids || !d_out_dists || !d_out_counts) { err = "null buffer"; return ERR_ARG; }
    if (k == 0) { err = "knbn must be > 0"; return ERR_ARG; }
    const uint64_t ef = std::max(ef_arg, k);  // src/hnsw.rs:1531
    if (ef > 1024) { err = "ef (= max(ef, knbn)) above 1024 is not supported by the register-resident result set"; return ERR_ARG; }
    if (nq > 0xFFFFFFF0ull) { err = "too many queries in one batch"; return ERR_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    HIP_TRY(hipSetDevice(device_));
    int rc = ensure_workspace(nq, k, err);
    if (rc != OK) return rc;
    uint32_t* stats = d_stats ? d_stats : static_cast<uint32_t*>(d_stats_);

    int slots = 1;
    while ((uint64_t)slots * 64 < ef) slots *= 2;

    HIP_TRY(hipEventRecord((hipEvent_t)ev_start_, stream));
    // pad queries to the row stride (tiny, stays on the launch stream)
    {
        const uint64_t total = nq * v_.row_stride;
        const int blocks = (int)std::min<uint64_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(pad_queries_kernel, dim3(blocks), dim3(256), 0, stream, d_queries,
                           static_cast<float*>(d_qpad_), (uint32_t)nq, v_.d, v_.row_stride);
    }

    // first guess for the LDS table: 2x the expected number of visited points, in [2^10, 2^14] slots
    const uint64_t expect = ef * std::min<uint64_t>(v_.deg_stride, 64) + 64;
    uint32_t bits = std::min<uint32_t>(14u, std::max<uint32_t>(10u, ceil_log2(expect * 2)));
    int table = TABLE_LDS_HASH;

    uint32_t launches = 0;
    uint32_t work = (uint32_t)nq;
    const uint32_t* qlist = nullptr;
    int pingpong = 0;
    for (;;) {
        KernelFn fn = pick_kernel(dist_, slots, table);
        size_t lds = table == TABLE_LDS_HASH ? ((size_t)4 << bits) : 0;
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, lds));
        if (per_cu < 1) per_cu = 1;
        uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)per_cu * (uint64_t)num_cu_, work);
        SearchArgs a{};
        a.queries = static_cast<const float*>(d_qpad_);
        a.qlist = qlist;
        a.nq = work;
        a.k = (uint32_t)k;
        a.ef = (uint32_t)ef;
        a.hash_bits = bits;
        a.work_counter = static_cast<uint32_t*>(d_ctrl_);
        a.overflow_count = static_cast<uint32_t*>(d_ctrl_) + 1;
        a.retry_out = static_cast<uint32_t*>(d_retry_[pingpong]);
        a.out_ids = d_out_ids;
        a.out_dists = d_out_dists;
        a.out_layer = d_out_layer;
        a.out_rank = d_out_rank;
        a.out_counts = d_out_counts;
        a.stats = stats;
        if (table == TABLE_GLOBAL_BITMAP) {
            a.bitmap_words = (v_.n + 31) / 32;
            grid = std::min<uint32_t>(grid, (uint32_t)num_cu_ * 4);
            const uint64_t need = (uint64_t)grid * a.bitmap_words * sizeof(uint32_t);
            if (need > bitmap_cap_) {
                if (d_bitmap_) (void)hipFree(d_bitmap_);
                d_bitmap_ = nullptr;
                HIP_TRY(hipMalloc(&d_bitmap_, need));
                bitmap_cap_ = need;
            }
            a.bitmap = static_cast<uint32_t*>(d_bitmap_);
        }
        HIP_TRY(hipMemsetAsync(d_ctrl_, 0, 8, stream));
        hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds, stream, v_, a);
        HIP_TRY(hipGetLastError());
        ++launches;
        uint32_t ctrl[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(ctrl, d_ctrl_, 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (ctrl[1] == 0) break;
        // some queries visited more points than the table holds: rerun only those, bigger table
        work = ctrl[1];
        qlist = static_cast<const uint32_t*>(d_retry_[pingpong]);
        pingpong ^= 1;
        if (table == TABLE_LDS_HASH && bits < 14) bits = 14;
        else if (table == TABLE_LDS_HASH) table = TABLE_GLOBAL_BITMAP;
        else { err = "internal error: bitmap visited set reported an overflow"; return ERR_DEVICE; }
    }
    HIP_TRY(hipEventRecord((hipEvent_t)ev_stop_, stream));
    HIP_TRY(hipEventSynchronize((hipEvent_t)ev_stop_));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, (hipEvent_t)ev_start_, (hipEvent_t)ev_stop_));
    last_ms_ = ms;
    last_launches_ = launches;
    return OK;
}

int DeviceIndex::search_host(const float* queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef, uint64_t* out_ids,
                             float* out_dists, uint8_t* out_layer, int32_t* out_rank, uint32_t* out_counts,
                             std::string& err) {
    if (!ready_) { err = "index is not resident on a device: call hnswgpu_upload first"; return ERR_DEVICE; }
    if (nq == 0) return OK;
    if (!queries || !out_ids || !out_dists || !out_counts) { err = "null buffer"; return ERR_ARG; }
    if (d != v_.d) { err = "query dimension differs from the index dimension"; return ERR_ARG; }
    HIP_TRY(hipSetDevice(device_));
    if (nq * d > hostio_cap_q_ || nq * k > hostio_cap_k_ || nq > hostio_cap_n_) {
        for (auto& p : d_hostio_) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
        hostio_cap_q_ = hostio_cap_k_ = hostio_cap_n_ = 0;
        HIP_TRY(hipMalloc(&d_hostio_[0], nq * d * sizeof(float)));
        HIP_TRY(hipMalloc(&d_hostio_[1], nq * k * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&d_hostio_[2], nq * k * sizeof(float)));
        HIP_TRY(hipMalloc(&d_hostio_[3], nq * k * (sizeof(int32_t) + 1)));
        HIP_TRY(hipMalloc(&d_hostio_[4], nq * sizeof(uint32_t)));
        hostio_cap_q_ = nq * d;
        hostio_cap_k_ = nq * k;
        hostio_cap_n_ = nq;
    }
    float* dq = static_cast<float*>(d_hostio_[0]);
    uint64_t* dids = static_cast<uint64_t*>(d_hostio_[1]);
    float* ddist = static_cast<float*>(d_hostio_[2]);
    int32_t* drank = static_cast<int32_t*>(d_hostio_[3]);
    uint8_t* dlayer = reinterpret_cast<uint8_t*>(drank + nq * k);
    uint32_t* dcnt = static_cast<uint32_t*>(d_hostio_[4]);
    HIP_TRY(hipMemcpy(dq, queries, nq * d * sizeof(float), hipMemcpyHostToDevice));
    int rc = search_device(dq, nq, d, k, ef, dids, ddist, dlayer, drank, dcnt, nullptr, nullptr, err);
    if (rc != OK) return rc;
    HIP_TRY(hipMemcpy(out_ids, dids, nq * k * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_dists, ddist, nq * k * sizeof(float), hipMemcpyDeviceToHost));
    if (out_layer) HIP_TRY(hipMemcpy(out_layer, dlayer, nq * k, hipMemcpyDeviceToHost));
    if (out_rank) HIP_TRY(hipMemcpy(out_rank, drank, nq * k * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_counts, dcnt, nq * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return OK;
}

int eval_distances_device(int dist, const float* a, const float* b, uint64_t n, uint64_t d, float* out, std::string& err) {
    if (n == 0) return OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { err = "no HIP device visible"; return ERR_DEVICE; }
    const uint32_t rs = (uint32_t)((d + 31) / 32 * 32);
    std::vector<float> pa((size_t)n * rs, 0.f), pb((size_t)n * rs, 0.f);
    for (uint64_t i = 0; i < n; ++i) {
        std::memcpy(pa.data() + i * rs, a + i * d, d * sizeof(float));
        std::memcpy(pb.data() + i * rs, b + i * d, d * sizeof(float));
    }
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&da, pa.size() * sizeof(float)));
    HIP_TRY(hipMalloc(&db, pb.size() * sizeof(float)));
    HIP_TRY(hipMalloc(&dout, n * sizeof(float)));
    HIP_TRY(hipMemcpy(da, pa.data(), pa.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, pb.data(), pb.size() * sizeof(float), hipMemcpyHostToDevice));
    const int blocks = (int)((n + 63) / 64);
    switch (dist) {
        case DIST_L2: hipLaunchKernelGGL(eval_pairs_kernel<DIST_L2>, dim3(blocks), dim3(64), 0, 0, da, db, dout, (uint32_t)n, rs); break;
        case DIST_COSINE: hipLaunchKernelGGL(eval_pairs_kernel<DIST_COSINE>, dim3(blocks), dim3(64), 0, 0, da, db, dout, (uint32_t)n, rs); break;
        case DIST_DOT: hipLaunchKernelGGL(eval_pairs_kernel<DIST_DOT>, dim3(blocks), dim3(64), 0, 0, da, db, dout, (uint32_t)n, rs); break;
        default: hipLaunchKernelGGL(eval_pairs_kernel<DIST_L1>, dim3(blocks), dim3(64), 0, 0, da, db, dout, (uint32_t)n, rs); break;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    return OK;
}

}  // namespace hnswgpu
