// search_device.hip -- host driver of the CDNA4 (gfx950) kernels for the batched-search hot path of hnsw_rs
// (the device code is in search_kernels.inc, instantiated per metric).  Written for MI355X only: wave64, LDS.
//
// Reference path (file:line under /root/reference):
//   Hnsw::parallel_search          src/hnsw.rs:1612-1635   -> one wavefront per query, persistent grid; batches of
//                                                             >= 256 queries are searched longest-first
//                                                             (hnsw_estimate_kernel + order_desc_kernel)
//   Hnsw::search_filter(None)      src/hnsw.rs:1487-1580   -> descent prologue + result epilogue
//   Hnsw::search_layer             src/hnsw.rs:922-1064    -> expansion loop (visited set in LDS, ef-bounded
//                                                             result/candidate set in VGPRs; literal BinaryHeaps
//                                                             for the queries that meet an exact f32 tie)
//   Distance<f32>::eval            anndists 0.1            -> batch_dist<METRIC>: rows read by groups of lanes, each
//                                                             distance summed left to right exactly like the
//                                                             crate's scalar build (bit-identical)
//
// Arithmetic contract: every distance is accumulated in the reference's order (sequential over the vector index, no
// FMA contraction), so ids AND f32 distances equal the CPU oracle bit for bit.  Build with -ffp-contract=off
// -fhip-fp32-correctly-rounded-divide-sqrt.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hnswio.hpp"
#include "search_device.hpp"
#include "search_kernels.hpp"

namespace hnswgpu {

namespace {

#if defined(HNSW_COSINE_GROUPS) && HNSW_COSINE_GROUPS
// DistCosine's third sum for every point, once: f32 squares widened to f64, summed left to right (padding adds 0.0)
__global__ void row_sq_norms_kernel(const float* __restrict__ vec, double* __restrict__ out, uint32_t n, uint32_t row_stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = vec + (size_t)i * row_stride;
    double s2 = 0.;
    for (uint32_t c = 0; c < row_stride; ++c) s2 = s2 + (double)(r[c] * r[c]);
    out[i] = s2;
}
#endif

// queries [nq][d] -> [nq][row_stride] zero padded
__global__ void pad_queries_kernel(const float* __restrict__ src, float* __restrict__ dst, uint32_t nq, uint32_t d,
                                   uint32_t row_stride) {
    const size_t total = (size_t)nq * row_stride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / row_stride), c = (uint32_t)(i % row_stride);
        dst[i] = c < d ? src[(size_t)r * d + c] : 0.f;
    }
}


#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
            return ERR_DEVICE;                                                                 \
        }                                                                                      \
    } while (0)

// A search call ends with one read-back of the counters.  The kernels of this path run for milliseconds, and a blocking
// wait costs ~0.1 ms of wake-up latency per call: poll first, block only when the work is long.
hipError_t wait_stream(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return hipStreamSynchronize(s);
    }
}
hipError_t wait_event(hipEvent_t ev) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return hipEventSynchronize(ev);
    }
}

uint32_t ceil_log2(uint64_t x) {
    uint32_t b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}


const KernelSet& kernel_set(int metric) {
    switch (metric) {
        case DIST_L2: return kernels_l2();
        case DIST_COSINE: return kernels_cosine();
        case DIST_DOT: return kernels_dot();
        default: return kernels_l1();
    }
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
    void** ptrs[] = {&d_vec_, &d_nbr0_, &d_up_ptr_, &d_up_ids_, &d_origin_, &d_nrm2_, &d_qpad_, &d_ctrl_, &d_retry_[0], &d_retry_[1],
                     &d_stats_, &d_bitmap_, &d_tie_, &d_heaps_, &d_cand_, &d_predist_, &d_order_, &d_hostio_[0], &d_hostio_[1], &d_hostio_[2], &d_hostio_[3], &d_hostio_[4]};
    for (void** p : ptrs)
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (h_ctrl_) { (void)hipHostFree(h_ctrl_); h_ctrl_ = nullptr; }
    if (ev_start_) { (void)hipEventDestroy((hipEvent_t)ev_start_); ev_start_ = nullptr; }
    if (ev_stop_) { (void)hipEventDestroy((hipEvent_t)ev_stop_); ev_stop_ = nullptr; }
    if (ev_mid_) { (void)hipEventDestroy((hipEvent_t)ev_mid_); ev_mid_ = nullptr; }
    if (ev_ks_) { (void)hipEventDestroy((hipEvent_t)ev_ks_); ev_ks_ = nullptr; }
    if (ev_ke_) { (void)hipEventDestroy((hipEvent_t)ev_ke_); ev_ke_ = nullptr; }
    ready_ = false;
}

int DeviceIndex::upload(const FlatIndex& x, int device, std::string& err) {
    if (const char* e = std::getenv("HNSWGPU_STRICT_TIES")) strict_ties_ = std::atoi(e) != 0;
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
#if defined(HNSW_COSINE_GROUPS) && HNSW_COSINE_GROUPS
    if (x.dist == DIST_COSINE) {
        HIP_TRY(hipMalloc(&d_nrm2_, n * sizeof(double)));
        hipLaunchKernelGGL(row_sq_norms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, static_cast<const float*>(d_vec_),
                           static_cast<double*>(d_nrm2_), (uint32_t)n, v.row_stride);
        HIP_TRY(hipDeviceSynchronize());
        bytes_ += n * sizeof(double);
    }
#endif
    HIP_TRY(hipMalloc(&d_ctrl_, 64));
    HIP_TRY(hipHostMalloc(&h_ctrl_, 64, hipHostMallocDefault));
    hipEvent_t e0, e1, e2;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventCreate(&e2));
    ev_start_ = e0;
    ev_stop_ = e1;
    ev_mid_ = e2;
    hipEvent_t e3, e4;
    HIP_TRY(hipEventCreate(&e3));
    HIP_TRY(hipEventCreate(&e4));
    ev_ks_ = e3;
    ev_ke_ = e4;

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
    if (nq > tie_cap_) {
        if (d_tie_) (void)hipFree(d_tie_);
        d_tie_ = nullptr;
        HIP_TRY(hipMalloc(&d_tie_, nq * sizeof(uint32_t)));
        tie_cap_ = nq;
    }
    if (nq > sched_cap_) {
        for (void** p : {&d_predist_, &d_order_}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        sched_cap_ = 0;
        HIP_TRY(hipMalloc(&d_predist_, nq * sizeof(float)));
        HIP_TRY(hipMalloc(&d_order_, nq * sizeof(uint32_t)));
        sched_cap_ = nq;
    }
    if (nq > retry_cap_) {
        for (int i = 0; i < 2; ++i) {
            if (d_retry_[i]) (void)hipFree(d_retry_[i]);
            d_retry_[i] = nullptr;
            HIP_TRY(hipMalloc(&d_retry_[i], nq * sizeof(uint32_t)));
        }
        retry_cap_ = nq;
    }
    if (nq * 8 * sizeof(uint32_t) > stats_cap_) {
        if (d_stats_) (void)hipFree(d_stats_);
        d_stats_ = nullptr;
        HIP_TRY(hipMalloc(&d_stats_, nq * 8 * sizeof(uint32_t)));
        stats_cap_ = nq * 8 * sizeof(uint32_t);
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
    if (slots == 8) slots = 16;  // kernels are instantiated for 1, 2, 4 and 16 result slots per lane

    HIP_TRY(hipEventRecord((hipEvent_t)ev_start_, stream));
    // pad queries to the row stride (tiny, stays on the launch stream)
    {
        const uint64_t total = nq * v_.row_stride;
        const int blocks = (int)std::min<uint64_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(pad_queries_kernel, dim3(blocks), dim3(256), 0, stream, d_queries,
                           static_cast<float*>(d_qpad_), (uint32_t)nq, v_.d, v_.row_stride);
    }

    // Visited-set sizing.  LDS per wavefront is what bounds occupancy, so the table is sized for the
    // typical query (ef x degree cells, ~2.4x the median number of visited points, measured); the few
    // per cent of queries that outgrow it start over on the HBM bitmap inside the same launch.
    const uint32_t idbits = std::max<uint32_t>(1u, ceil_log2(v_.n));
    const uint64_t expect = ef * std::min<uint64_t>(v_.deg_stride, 64);
    uint32_t tbits = std::min<uint32_t>(14u, std::max<uint32_t>(8u, ceil_log2(expect)));
    // ... then follows what the previous batches with the same ef measured (adapt_* below)
    if (adapt_ef_ == ef && adapt_tbits_ != 0) tbits = adapt_tbits_;
    bool env_forced = false;
    if (const char* e = std::getenv("HNSWGPU_HASH_BITS")) {  // tuning / test hook: initial table size
        int b = std::atoi(e);
        if (b >= 6 && b <= 14) { tbits = (uint32_t)b; env_forced = true; }
    }
    const uint32_t tbits_first = tbits;
    const uint32_t tile_bytes = tile_bytes_for(dist_, v_.row_stride);
    const size_t lds_fixed = tile_bytes + IDS_BYTES;
    int table = TABLE_LDS_CELL16;
    bool grown = false;

    // Batch scheduling: the searches run in descending order of the (estimated) distance to the layer-0 entry point,
    // long searches first (DESIGN.md "scheduling").  Small batches skip it (one launch, lowest latency).
    const bool scheduled = nq >= 256 && !std::getenv("HNSWGPU_NO_SCHED");
    if (scheduled) {
        SearchArgs da{};
        da.queries = static_cast<const float*>(d_qpad_);
        da.nq = (uint32_t)nq;
        da.pre_dist = static_cast<float*>(d_predist_);
        const uint32_t dgrid = (uint32_t)std::min<uint64_t>(nq, (uint64_t)num_cu_ * 24u);
        HIP_TRY(kernel_set(dist_).launch_estimate(dgrid, stream, v_, da));
        HIP_TRY(kernel_set(dist_).launch_order(stream, static_cast<const float*>(d_predist_), (uint32_t)nq, static_cast<uint32_t*>(d_order_)));
    }
    uint32_t launches = 0;
    uint32_t work = (uint32_t)nq;
    uint32_t n_ties = 0, n_converted = 0;
    const uint32_t* qlist = scheduled ? static_cast<const uint32_t*>(d_order_) : nullptr;
    int pingpong = 0;
    SearchArgs last_args{};
    for (;;) {
        SearchArgs a{};
        size_t lds = lds_fixed;
        if (table != TABLE_GLOBAL_BITMAP) {
            uint32_t tb = std::min(tbits, idbits);  // a table with one cell per possible id never probes
            if (idbits - tb <= 11u) {
                table = TABLE_LDS_CELL16;
                a.restbits = idbits - tb;
                lds += (size_t)2 << tb;
            } else {
                table = TABLE_LDS_CELL32;
                lds += (size_t)4 << tb;
            }
            a.tbits = tb;
        }
        a.tile_bytes = tile_bytes;
#if defined(HNSW_COSINE_GROUPS) && HNSW_COSINE_GROUPS
        a.nrm2 = static_cast<const double*>(d_nrm2_);
#endif
        a.idbits = idbits;
        const KernelSet& ks = kernel_set(dist_);
        const bool strict_kernel = strict_ties_ && table != TABLE_GLOBAL_BITMAP && !std::getenv("HNSWGPU_NO_INKERNEL");
        if (strict_kernel) {  // top levels of candidate_points for the queries that are answered by the literal heaps
            a.cand_lds = 512;
            lds += (size_t)a.cand_lds * sizeof(hent_t);
        }
        int per_cu = 0;
        HIP_TRY(ks.occupancy(slots, table, strict_kernel, lds, &per_cu));
        if (per_cu < 1) per_cu = 1;
        uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)per_cu * (uint64_t)num_cu_, work);
        a.queries = static_cast<const float*>(d_qpad_);
        a.qlist = qlist;
        a.nq = work;
        a.k = (uint32_t)k;
        a.ef = (uint32_t)ef;
        a.work_counter = static_cast<uint32_t*>(d_ctrl_);
        a.overflow_count = static_cast<uint32_t*>(d_ctrl_) + 1;
        a.retry_out = static_cast<uint32_t*>(d_retry_[pingpong]);
        a.out_ids = d_out_ids;
        a.out_dists = d_out_dists;
        a.out_layer = d_out_layer;
        a.out_rank = d_out_rank;
        a.out_counts = d_out_counts;
        a.stats = stats;
        {
            // HBM bitmaps for the in-launch fallback: one slice per workgroup, within a 4 GiB budget
            a.bitmap_words = (v_.n + 31) / 32;
            const uint64_t slice = (uint64_t)a.bitmap_words * sizeof(uint32_t);
            uint64_t blocks = std::min<uint64_t>(grid, std::max<uint64_t>(1, (4ull << 30) / slice));
            if (table == TABLE_GLOBAL_BITMAP) grid = (uint32_t)blocks;  // every workgroup needs one
            const uint64_t need = blocks * slice;
            if (need > bitmap_cap_) {
                if (d_bitmap_) (void)hipFree(d_bitmap_);
                d_bitmap_ = nullptr;
                bitmap_cap_ = 0;
                HIP_TRY(hipMalloc(&d_bitmap_, need));
                bitmap_cap_ = need;
            }
            a.bitmap = static_cast<uint32_t*>(d_bitmap_);
            a.bitmap_blocks = (uint32_t)blocks;
        }
        a.tie_list = strict_ties_ ? static_cast<uint32_t*>(d_tie_) : nullptr;
        if (strict_kernel) {
            // per-workgroup scratch for the part of candidate_points that does not fit in LDS
            const uint32_t cap = 8192;
            const uint64_t need = (uint64_t)grid * cap * sizeof(hent_t);
            if (need > strict_cap_) {
                if (d_cand_) (void)hipFree(d_cand_);
                d_cand_ = nullptr;
                strict_cap_ = 0;
                HIP_TRY(hipMalloc(&d_cand_, need));
                strict_cap_ = need;
            }
            a.cand_scratch = static_cast<hent_t*>(d_cand_);
            a.cand_cap = cap;
#if (defined(HNSW_STRICT_RESUME) && HNSW_STRICT_RESUME) || (defined(HNSW_EXACT_VALUE_R) && HNSW_EXACT_VALUE_R)
            {   // experiment: heap-operation log of the first attempt (same size as the candidate scratch)
                static void* d_oplog = nullptr;
                static uint64_t oplog_bytes = 0;
                if (need > oplog_bytes) {
                    if (d_oplog) (void)hipFree(d_oplog);
                    d_oplog = nullptr;
                    oplog_bytes = 0;
                    HIP_TRY(hipMalloc(&d_oplog, need));
                    oplog_bytes = need;
                }
                a.oplog = static_cast<hent_t*>(d_oplog);
                a.oplog_cap = cap;
            }
#endif
            // most queries of the previous batch met a tie (integer-valued data does that): skip the first attempt
            a.exact_first = (adapt_exact_ef_ == ef && adapt_exact_first_) ? 1u : 0u;
            if (const char* e = std::getenv("HNSWGPU_EXACT_FIRST")) a.exact_first = std::atoi(e) != 0 ? 1u : 0u;
        }
        HIP_TRY(hipMemsetAsync(d_ctrl_, 0, launches == 0 ? 32 : 16, stream));  // the tie list spans relaunches
        if (launches == 0) HIP_TRY(hipEventRecord((hipEvent_t)ev_ks_, stream));
        HIP_TRY(ks.launch_search(slots, table, strict_kernel, grid, lds, stream, v_, a));
        if (launches == 0) HIP_TRY(hipEventRecord((hipEvent_t)ev_ke_, stream));
        last_args = a;
        ++launches;
        volatile uint32_t* ctrl = static_cast<volatile uint32_t*>(h_ctrl_);  // pinned: a true asynchronous copy
        HIP_TRY(hipMemcpyAsync(h_ctrl_, d_ctrl_, 24, hipMemcpyDeviceToHost, stream));
        HIP_TRY(wait_stream(stream));
        n_ties = ctrl[4];        // flagged for the replay kernel (cumulative over relaunches)
        n_converted = ctrl[5];   // needed the literal heaps inside the launch
        if (launches == 1 && strict_kernel && nq >= 256) {
            adapt_exact_ef_ = ef;
            adapt_exact_first_ = (uint64_t)n_converted * 2 > nq;
        }
        if (launches == 1 && table != TABLE_GLOBAL_BITMAP && !env_forced && nq >= 256) {
            // Table sizing feedback for the next batch: grow when more than ~1 query in 8 had to move to the
            // HBM bitmap, shrink when a half-size table would have overflowed for fewer than 1 in 32.
            uint32_t next = tbits_first;
            if ((uint64_t)ctrl[2] * 8 > nq && tbits_first < 14u) next = tbits_first + 1;
            else if ((uint64_t)ctrl[3] * 32 < nq && tbits_first > 8u) next = tbits_first - 1;
            adapt_ef_ = ef;
            adapt_tbits_ = next;
        }
        if (ctrl[1] == 0) break;
        // some queries visited more points than the table holds: rerun only those
        work = ctrl[1];
        qlist = static_cast<const uint32_t*>(d_retry_[pingpong]);
        pingpong ^= 1;
        if (table == TABLE_GLOBAL_BITMAP) { err = "internal error: bitmap visited set reported an overflow"; return ERR_DEVICE; }
        if (!grown && tbits < 14u) {
            tbits = std::min<uint32_t>(14u, tbits + 2u);
            grown = true;
        } else {
            table = TABLE_GLOBAL_BITMAP;
        }
    }
    last_ties_ = n_ties + n_converted;
    HIP_TRY(hipEventRecord((hipEvent_t)ev_mid_, stream));  // end of the search kernel proper (before any exact replay)
    if (strict_ties_ && n_ties > 0) {
        // Exact replay of the tie-affected queries with literal binary heaps (hnsw_search_exact_kernel).
        const uint64_t bm_slice = (uint64_t)last_args.bitmap_words * sizeof(uint32_t);
        const uint64_t cand_cap = v_.n;                                   // every point is accepted at most once
        const uint64_t heap_stride = ef + 2 + cand_cap;
        const uint64_t per_block = bm_slice + heap_stride * sizeof(hent_t);
        uint32_t grid = (uint32_t)std::min<uint64_t>(n_ties, std::max<uint64_t>(1, (2ull << 30) / per_block));
        grid = std::min<uint32_t>(grid, (uint32_t)num_cu_ * 4u);
        if ((uint64_t)grid * bm_slice > bitmap_cap_) {
            if (d_bitmap_) (void)hipFree(d_bitmap_);
            d_bitmap_ = nullptr;
            bitmap_cap_ = 0;
            HIP_TRY(hipMalloc(&d_bitmap_, (uint64_t)grid * bm_slice));
            bitmap_cap_ = (uint64_t)grid * bm_slice;
        }
        if ((uint64_t)grid * heap_stride * sizeof(hent_t) > heaps_cap_) {
            if (d_heaps_) (void)hipFree(d_heaps_);
            d_heaps_ = nullptr;
            heaps_cap_ = 0;
            HIP_TRY(hipMalloc(&d_heaps_, (uint64_t)grid * heap_stride * sizeof(hent_t)));
            heaps_cap_ = (uint64_t)grid * heap_stride * sizeof(hent_t);
        }
        SearchArgs a = last_args;
        a.qlist = static_cast<const uint32_t*>(d_tie_);
        a.nq = n_ties;
        a.bitmap = static_cast<uint32_t*>(d_bitmap_);
        a.tie_list = nullptr;
        ExactArgs x{};
        x.heaps = static_cast<hent_t*>(d_heaps_);
        x.heap_stride = heap_stride;
        x.cand_cap = (uint32_t)cand_cap;
        // heaps' top levels in LDS: up to ~56 KiB per workgroup (the exact replay is latency-, not occupancy-bound)
        const uint64_t lds_budget = 56 * 1024 - (tile_bytes + IDS_BYTES);
        x.r_lds_cap = (uint32_t)std::min<uint64_t>(ef + 2, lds_budget / 2 / sizeof(hent_t));
        x.cand_lds = (uint32_t)std::min<uint64_t>(cand_cap, (lds_budget - (uint64_t)x.r_lds_cap * sizeof(hent_t)) / sizeof(hent_t));
        HIP_TRY(hipMemsetAsync(d_ctrl_, 0, 8, stream));
        const size_t lds = tile_bytes + IDS_BYTES + ((size_t)x.r_lds_cap + x.cand_lds) * sizeof(hent_t);
        const int ns = ef <= 64 ? 1 : ef <= 128 ? 2 : 0;  // return_points in VGPRs when it fits (push+pop fused when full)
        HIP_TRY(kernel_set(dist_).launch_exact(ns, grid, lds, stream, v_, a, x));
        ++launches;
        volatile uint32_t* ctrl2 = static_cast<volatile uint32_t*>(h_ctrl_);
        HIP_TRY(hipMemcpyAsync(h_ctrl_, d_ctrl_, 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(wait_stream(stream));
        if (ctrl2[1] != 0) { err = "internal error: candidate heap overflow in the exact replay"; return ERR_DEVICE; }
    }
    HIP_TRY(hipEventRecord((hipEvent_t)ev_stop_, stream));
    HIP_TRY(wait_event((hipEvent_t)ev_stop_));
    float ms = 0.f, ms_main = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, (hipEvent_t)ev_start_, (hipEvent_t)ev_stop_));
    HIP_TRY(hipEventElapsedTime(&ms_main, (hipEvent_t)ev_ks_, (hipEvent_t)ev_ke_));  // first launch of the search kernel alone
    last_ms_ = ms;
    last_main_ms_ = ms_main;
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
    HIP_TRY(kernel_set(dist).launch_eval_pairs((uint32_t)blocks, da, db, dout, (uint32_t)n, rs));
    HIP_TRY(hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    return OK;
}

}  // namespace hnswgpu
