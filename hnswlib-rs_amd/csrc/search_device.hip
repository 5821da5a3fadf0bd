// search_device.hip -- host driver of the CDNA4 (gfx950) kernels for the batched-search hot path of hnsw_rs
// (the device code is in search_kernels.inc, instantiated per metric).  Written for MI355X only: wave64, LDS.
//
// Reference path (file:line under /root/reference):
//   Hnsw::parallel_search          src/hnsw.rs:1612-1635   -> one wavefront per query, persistent grid; batches of
//                                                             >= 256 queries are searched longest-first
//                                                             (order_desc_kernel on the descent's distances)
//   Hnsw::search_filter(None)      src/hnsw.rs:1487-1580   -> hnsw_descend_kernel (padding + greedy descent of every
//                                                             query, first kernel of a call) + result epilogue
//   Hnsw::search_layer             src/hnsw.rs:922-1064    -> expansion loop (visited set in LDS, ef-bounded
//                                                             result/candidate set in VGPRs; literal BinaryHeaps
//                                                             where equal f32 distances make the reference's
//                                                             decisions depend on heap order)
//   Hnsw::search_filter(Some(&Vec<usize>))                 -> hnsw_search_exact_kernel with an allow bitmap
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
#include <thread>
#include <vector>

#include "hnswio.hpp"
#include "search_device.hpp"
#include "search_kernels.hpp"
#include "worker_pool.hpp"

namespace hnswgpu {

namespace {
Knobs read_knobs() {
    Knobs k;
    auto num = [](const char* name, int unset) { const char* e = std::getenv(name); return e ? std::atoi(e) : unset; };
    auto flag = [](const char* name) { return std::getenv(name) != nullptr; };
    k.hash_bits = num("HNSWGPU_HASH_BITS", -1);
    k.no_sched = flag("HNSWGPU_NO_SCHED");
    k.no_pair_descent = flag("HNSWGPU_NO_PAIR_DESCENT");
    k.no_inkernel = flag("HNSWGPU_NO_INKERNEL");
    k.strict_wg_per_cu = num("HNSWGPU_STRICT_WG_PER_CU", -1);
    k.cand_lds = num("HNSWGPU_CAND_LDS", -1);
    k.waves_per_cu = num("HNSWGPU_WAVES_PER_CU", -1);
    k.exact_first = num("HNSWGPU_EXACT_FIRST", -1);
    k.trace_launch = flag("HNSWGPU_TRACE_LAUNCH");
    k.trace_host = flag("HNSWGPU_TRACE_HOST");
    k.host_threads = num("HNSWGPU_HOST_THREADS", -1);
    k.host_chunks = num("HNSWGPU_HOST_CHUNKS", -1);
    k.ffi_unpack = flag("HNSWGPU_FFI_UNPACK");
    k.pair_search = num("HNSWGPU_PAIR_SEARCH", -1);
    k.pair_tbits_delta = num("HNSWGPU_PAIR_TBITS_DELTA", 0);
    k.pair_wg_per_cu = num("HNSWGPU_PAIR_WG_PER_CU", -1);
    return k;
}
std::atomic<const Knobs*> g_knobs{nullptr};
}  // namespace
const Knobs& knobs() {
    const Knobs* k = g_knobs.load(std::memory_order_acquire);
    if (k == nullptr) {
        const Knobs* fresh = new Knobs(read_knobs());
        if (g_knobs.compare_exchange_strong(k, fresh, std::memory_order_acq_rel)) k = fresh;
        else delete fresh;
    }
    return *k;
}
void reload_knobs() {  // (the previous set stays allocated: a call in flight may still be reading it; a few dozen bytes per reload)
    g_knobs.store(new Knobs(read_knobs()), std::memory_order_release);
}

namespace {

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

// Every entry point works on the replica's device and leaves the CALLER's current HIP device as it found it (a host
// thread shared with PyTorch, a Rust or a Julia runtime keeps allocating and launching where it did before the call).
class DeviceGuard {
public:
    explicit DeviceGuard(int device) {
        if (device < 0) return;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur == device) return;
        status_ = hipSetDevice(device);
        if (status_ == hipSuccess && cur >= 0) prev_ = cur;
    }
    ~DeviceGuard() { if (prev_ >= 0) (void)hipSetDevice(prev_); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
    hipError_t status() const { return status_; }
private:
    int prev_ = -1;
    hipError_t status_ = hipSuccess;
};

constexpr uint64_t PAIR_SEARCH_AUTO_MIN_QUERIES = 40000;  // hnsw_search_pair_kernel as the first pass of a strict DistCosine / DistDot batch (rows of one 128-byte line) of at least this many queries
constexpr int STRICT_WG_PER_CU = 16;  // resident workgroups per CU of a strict launch: four waves per SIMD (see search_device)

uint32_t ceil_log2(uint64_t x) {
    uint32_t b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}

const KernelSet& kernel_set(int metric) {
    switch (metric) {
        case DIST_L2: return kernels_for<DIST_L2>();
        case DIST_COSINE: return kernels_for<DIST_COSINE>();
        case DIST_DOT: return kernels_for<DIST_DOT>();
        case DIST_HELLINGER: return kernels_for<DIST_HELLINGER>();
        case DIST_JEFFREYS: return kernels_for<DIST_JEFFREYS>();
        case DIST_JENSENSHANNON: return kernels_for<DIST_JENSENSHANNON>();
        case KM_L2_SIMD8: return kernels_for<KM_L2_SIMD8>();
        case KM_COSINE_SIMD8: return kernels_for<KM_COSINE_SIMD8>();
        case KM_DOT_SIMD8: return kernels_for<KM_DOT_SIMD8>();
        case KM_L1_SIMD8: return kernels_for<KM_L1_SIMD8>();
        default: return kernels_for<DIST_L1>();
    }
}

// device buffer grown on demand; a failed allocation leaves it empty (pointer AND capacity), never half valid
struct DevBuf {
    void* p = nullptr;
    uint64_t cap = 0;
    hipError_t ensure(uint64_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = bytes;
        return hipSuccess;
    }
    void free() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};

}  // namespace

// everything one search call writes: taken from the replica's pool for the duration of the call
// pinned host memory grown on demand (staging of the host-buffer entry points)
// (mapped: the kernels of a host-buffer search read the queries from it and write the answers into it across PCIe --
// nothing is staged in HBM; dev = the same memory as the device addresses it)
struct PinnedBuf {
    void* p = nullptr;
    void* dev = nullptr;
    uint64_t cap = 0;
    uint64_t need = 0;       // what the last call asked for
    unsigned small_calls = 0;  // consecutive calls that did not need the size this buffer has grown to (see trim)
    hipError_t ensure(uint64_t bytes) {
        need = bytes;
        if (bytes <= cap) return hipSuccess;
        free();
        const uint64_t want = std::max<uint64_t>(bytes, 1u << 16);
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocMapped | hipHostMallocPortable);
        if (e != hipSuccess) { p = nullptr; return e; }
        e = hipHostGetDevicePointer(&dev, p, 0);
        if (e != hipSuccess) { (void)hipHostFree(p); p = nullptr; dev = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    // Staging memory of an unusually large batch is given back -- but not by the caller whose batches ARE that large: only after
    // eight calls in a row that did not need it (freeing and pinning hundreds of MB again on every call costs tens of ms and a
    // device synchronisation each time).
    void trim(uint64_t keep_bytes) {
        const uint64_t asked = need;
        need = 0;  // (a call that never touches this buffer -- the device-resident path shares the pool -- counts as a small one)
        if (cap <= keep_bytes) return;
        if (asked > keep_bytes) { small_calls = 0; return; }
        if (++small_calls >= 8u) { free(); small_calls = 0; }
    }
    void free() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        dev = nullptr;
        cap = 0;
    }
};

void* pinned_alloc(size_t bytes, void** dev) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipHostGetDevicePointer(dev, p, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(p); return nullptr; }
    return p;
}
void pinned_free(void* p) {
    if (p) (void)hipHostFree(p);
}

struct DeviceIndex::Workspace {
    DevBuf qpad, tie, pre, order, retry[2], stats, bitmap, heaps, cand, oplog, allow, allowed_ids;
    PinnedBuf pin_in, pin_out;
    void* d_ctrl = nullptr;   // work counter + counters
    void* h_ctrl = nullptr;   // pinned host copy (read back once per launch)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_ks = nullptr, ev_ke = nullptr;
    hipStream_t own_stream = nullptr;  // for the host-buffer entry points
    int init(std::string& err) {
        HIP_TRY(hipMalloc(&d_ctrl, 64));
        HIP_TRY(hipHostMalloc(&h_ctrl, 64, hipHostMallocDefault));
        HIP_TRY(hipEventCreate(&ev_start));
        HIP_TRY(hipEventCreate(&ev_stop));
        HIP_TRY(hipEventCreate(&ev_ks));
        HIP_TRY(hipEventCreate(&ev_ke));
        HIP_TRY(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
        return OK;
    }
    ~Workspace() {
        for (DevBuf* b : {&qpad, &tie, &pre, &order, &retry[0], &retry[1], &stats, &bitmap, &heaps, &cand, &oplog, &allow, &allowed_ids})
            b->free();
        pin_in.free();
        pin_out.free();
        if (d_ctrl) (void)hipFree(d_ctrl);
        if (h_ctrl) (void)hipHostFree(h_ctrl);
        for (hipEvent_t e : {ev_start, ev_stop, ev_ks, ev_ke})
            if (e) (void)hipEventDestroy(e);
        if (own_stream) (void)hipStreamDestroy(own_stream);
    }
};

class DeviceIndex::Lease {
public:
    Lease(DeviceIndex* o, Workspace* w) : o_(o), w_(w) {}
    ~Lease() { if (w_) o_->release_ws(w_); }
    Lease(const Lease&) = delete;
    Lease& operator=(const Lease&) = delete;
    Workspace* get() const { return w_; }
private:
    DeviceIndex* o_;
    Workspace* w_;
};

DeviceIndex::Workspace* DeviceIndex::acquire(std::string& err) {
    {
        std::lock_guard<std::mutex> g(pool_mu_);
        if (!free_ws_.empty()) {
            Workspace* w = free_ws_.back();
            free_ws_.pop_back();
            return w;
        }
    }
    std::unique_ptr<Workspace> w(new Workspace());
    if (w->init(err) != OK) return nullptr;
    std::lock_guard<std::mutex> g(pool_mu_);
    all_ws_.push_back(std::move(w));
    return all_ws_.back().get();
}
void DeviceIndex::release_ws(Workspace* w) {
    // pinned staging memory only ever grew: a single 100 000 x 784 batch left hundreds of MB pinned on every pooled workspace
    // (PinnedBuf::trim: given back after eight calls that did not need it)
    w->pin_in.trim(64ull << 20);
    w->pin_out.trim(64ull << 20);
    std::lock_guard<std::mutex> g(pool_mu_);
    free_ws_.push_back(w);
}

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lane_lab_device(int device, uint32_t mode, uint32_t p0, uint32_t p1, uint32_t p2, const uint32_t* ops, uint32_t n_ops,
                    const uint32_t* lanes, uint32_t n_lane_sets, uint32_t* out, uint32_t out_words, std::string& err) {
    if (mode > 3u || out_words < 1u) { err = "bad lane-lab mode"; return ERR_ARG; }
    if (mode == 0u && p0 > 1024u) { err = "lane lab: at most 1024 heap entries in LDS"; return ERR_ARG; }
    if ((mode == 1u || mode == 2u) && p0 != 1u && p0 != 2u && p0 != 4u) { err = "lane lab: 1, 2 or 4 slots per lane"; return ERR_ARG; }
    if (mode == 2u && (p1 == 0u || p1 > 64u * p0)) { err = "lane lab: ef beyond the result set"; return ERR_ARG; }
    if (mode == 3u && (p0 < 4u || p0 > 12u || p2 > 13u || p1 > 31u || p1 + 3u < p0 || p1 - (p0 - 3u) != p2)) { err = "lane lab: table geometry"; return ERR_ARG; }
    // the script is checked here, before anything reaches the device: an op that names a lane set the caller did not pass would
    // read device memory out of bounds (LAB_PUSH_LANES / LAB_MERGE / LAB_BATCH take the set's index in their last word)
    if (n_ops != 0u && ops == nullptr) { err = "lane lab: null script"; return ERR_ARG; }
    if (n_lane_sets != 0u && lanes == nullptr) { err = "lane lab: null lane sets"; return ERR_ARG; }
    for (uint32_t i = 0; i < n_ops; ++i) {
        const uint32_t op = ops[4u * i], c = ops[4u * i + 3u];
        if ((op == LAB_PUSH_LANES || op == LAB_MERGE || op == LAB_BATCH) && c >= n_lane_sets) {
            err = "lane lab: op " + std::to_string(i) + " names lane set " + std::to_string(c) + " of " + std::to_string(n_lane_sets);
            return ERR_ARG;
        }
    }
    DeviceGuard on_device(device);
    HIP_TRY(on_device.status());
    const uint32_t scratch_cap = 1u << 16;
    DevBuf d_ops, d_lanes, d_out, d_scratch;
    struct Free { DevBuf* b[4]; ~Free() { for (DevBuf* x : b) x->free(); } } fr{{&d_ops, &d_lanes, &d_out, &d_scratch}};
    HIP_TRY(d_ops.ensure(std::max<uint64_t>(16, (uint64_t)n_ops * 16)));
    HIP_TRY(d_lanes.ensure(std::max<uint64_t>(512, (uint64_t)n_lane_sets * 512)));
    HIP_TRY(d_out.ensure((uint64_t)out_words * 4));
    HIP_TRY(d_scratch.ensure((uint64_t)scratch_cap * sizeof(hent_t)));
    if (n_ops) HIP_TRY(hipMemcpy(d_ops.p, ops, (size_t)n_ops * 16, hipMemcpyHostToDevice));
    if (n_lane_sets) HIP_TRY(hipMemcpy(d_lanes.p, lanes, (size_t)n_lane_sets * 512, hipMemcpyHostToDevice));
    else HIP_TRY(hipMemset(d_lanes.p, 0, 512));
    HIP_TRY(hipMemset(d_out.p, 0, (size_t)out_words * 4));
    LaneLabArgs a{};
    a.ops = d_ops.as<uint32_t>();
    a.n_ops = n_ops;
    a.lanes = d_lanes.as<uint32_t>();
    a.mode = mode; a.p0 = p0; a.p1 = p1; a.p2 = p2;
    a.scratch = d_scratch.as<hent_t>();
    a.scratch_cap = scratch_cap;
    a.out = d_out.as<uint32_t>();
    a.out_cap = out_words;
    HIP_TRY(launch_lane_lab(nullptr, 16384, a));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)out_words * 4, hipMemcpyDeviceToHost));
    return OK;
}

int gather_sharded_answers(const int* devices, int n_shards, const uint64_t* nq_shard, uint64_t k, const uint64_t* const* d_ids,
                           const float* const* d_dists, const uint8_t* const* d_layer, const int32_t* const* d_rank,
                           const uint32_t* const* d_counts, int root_device, uint64_t* root_ids, float* root_dists,
                           uint8_t* root_layer, int32_t* root_rank, uint32_t* root_counts, void* root_stream, std::string& err) {
    DeviceGuard on_root(root_device);
    HIP_TRY(on_root.status());
    hipStream_t stream = static_cast<hipStream_t>(root_stream);
    // sizes first: nothing is copied when a product does not fit (the offsets below are row * k * 8 bytes at most)
    uint64_t total = 0;
    for (int s = 0; s < n_shards; ++s) {
        if (nq_shard[s] > UINT64_MAX - total) { err = "gather: the shards' query counts overflow"; return ERR_ARG; }
        total += nq_shard[s];
    }
    if (k != 0 && total > (UINT64_MAX / sizeof(uint64_t)) / k) { err = "gather: nq * k * 8 bytes overflows"; return ERR_ARG; }
    // an error part-way leaves earlier copies in flight on the caller's stream: they are waited for before the caller may free
    // or reuse the root arrays (whatever the copy that failed returned is what is reported)
    auto fail = [&](hipError_t e) {
        (void)hipStreamSynchronize(stream);
        err = std::string("HIP: ") + hipGetErrorString(e);
        return ERR_DEVICE;
    };
    uint64_t row = 0;  // first query of shard s in input order
    for (int s = 0; s < n_shards; ++s) {
        const uint64_t cnt = nq_shard[s];
        if (cnt == 0) continue;
        const int src = devices[s];
        hipError_t e = hipMemcpyPeerAsync(root_ids + row * k, root_device, d_ids[s], src, cnt * k * sizeof(uint64_t), stream);
        if (e == hipSuccess) e = hipMemcpyPeerAsync(root_dists + row * k, root_device, d_dists[s], src, cnt * k * sizeof(float), stream);
        if (e == hipSuccess && root_layer && d_layer && d_layer[s])
            e = hipMemcpyPeerAsync(root_layer + row * k, root_device, d_layer[s], src, cnt * k * sizeof(uint8_t), stream);
        if (e == hipSuccess && root_rank && d_rank && d_rank[s])
            e = hipMemcpyPeerAsync(root_rank + row * k, root_device, d_rank[s], src, cnt * k * sizeof(int32_t), stream);
        if (e == hipSuccess) e = hipMemcpyPeerAsync(root_counts + row, root_device, d_counts[s], src, cnt * sizeof(uint32_t), stream);
        if (e != hipSuccess) return fail(e);
        row += cnt;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return OK;
}

DeviceIndex::DeviceIndex() = default;
DeviceIndex::~DeviceIndex() { release(); }

void DeviceIndex::release() {
    DeviceGuard on_device(device_);
    {
        std::lock_guard<std::mutex> g(pool_mu_);
        free_ws_.clear();
        all_ws_.clear();
    }
    void** ptrs[] = {&d_vec_, &d_nbr0_, &d_up_ptr_, &d_up_ids_, &d_origin_, &d_nrm2_};
    for (void** p : ptrs)
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    ready_ = false;
}

int DeviceIndex::kernel_metric() const {
    if (arith_.load() == ARITH_SIMD8) {
        const int km = simd8_kernel_metric(dist_);
        if (km >= 0) return km;
    }
    return dist_;
}

CallInfo DeviceIndex::last_call() const {
    std::lock_guard<std::mutex> g(meta_mu_);
    return last_;
}

int DeviceIndex::upload(const FlatIndex& x, int device, std::string& err) {
    if (const char* e = std::getenv("HNSWGPU_STRICT_TIES")) strict_ties_.store(std::atoi(e) != 0);
    if (x.n == 0 || x.entry_flat == NO_POINT) { err = "cannot upload an empty index"; return ERR_EMPTY; }
    // bit 31 of a flat id is the EXPANDED flag of the result set
    if (x.n >= 0x80000000ull) { err = "index too large for the device path (2^31 points or more)"; return ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { err = "no HIP device visible (a gfx950 GPU is required; there is no CPU fallback)"; return ERR_DEVICE; }
    if (device < 0 || device >= ndev) { err = "bad device ordinal"; return ERR_ARG; }
    release();
    DeviceGuard on_device(device);
    HIP_TRY(on_device.status());
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    num_cu_ = prop.multiProcessorCount;
    device_ = device;
    dist_ = x.dist;
    bytes_ = 0;
    descend_per_cu_[0].store(0);
    descend_per_cu_[1].store(0);

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
        uint64_t up_deg_max = 0;
        for (unsigned l = 1; l <= top; ++l) {
            uint32_t* p = ptr.data() + (size_t)(l - 1) * (n + 1);
            for (uint64_t f = 0; f < n; ++f) {
                p[f] = (uint32_t)ids.size();
                uint64_t b = x.nbr_ptr[f * NB_LAYER_MAX + l], e = x.nbr_ptr[f * NB_LAYER_MAX + l + 1];
                ids.insert(ids.end(), x.nbr_flat.begin() + b, x.nbr_flat.begin() + e);
                up_deg_max = std::max<uint64_t>(up_deg_max, e - b);
            }
            p[n] = (uint32_t)ids.size();
        }
        if (ids.empty()) ids.push_back(0);
        up_deg_max_ = (uint32_t)std::min<uint64_t>(up_deg_max, 0xFFFFFFFFull);
        HIP_TRY(hipMalloc(&d_up_ptr_, ptr.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(d_up_ptr_, ptr.data(), ptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&d_up_ids_, ids.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(d_up_ids_, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        bytes_ += (ptr.size() + ids.size()) * sizeof(uint32_t);
    }
    HIP_TRY(hipMalloc(&d_origin_, n * sizeof(uint64_t)));
    HIP_TRY(hipMemcpy(d_origin_, x.origin_id.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice));
    bytes_ += n * sizeof(uint64_t);
    if (x.dist == DIST_COSINE) {  // DistCosine's per-point sum of squares, once: inside the row when its padding has room
        if (!norm_fits_row(x.dist, v.d, v.row_stride)) {
            HIP_TRY(hipMalloc(&d_nrm2_, n * sizeof(double)));
            bytes_ += n * sizeof(double);
        }
        HIP_TRY(launch_row_sq_norms(nullptr, static_cast<float*>(d_vec_), static_cast<double*>(d_nrm2_), (uint32_t)n, v.d, v.row_stride));
        HIP_TRY(hipDeviceSynchronize());
    }

    v.vec = static_cast<const float*>(d_vec_);
    v.nbr0 = static_cast<const uint32_t*>(d_nbr0_);
    v.up_ptr = static_cast<const uint32_t*>(d_up_ptr_);
    v.up_ids = static_cast<const uint32_t*>(d_up_ids_);
    v.origin_id = static_cast<const uint64_t*>(d_origin_);
    v_ = v;
    {
        std::lock_guard<std::mutex> g(meta_mu_);
        adapt_ef_ = 0;
        adapt_tbits_ = 0;
        last_ = CallInfo{};
    }
    ready_ = true;
    return OK;
}

// Literal search (hnsw_search_exact_kernel) of the queries in d_qlist (nullptr: all nq), optionally filtered.
// Resident workgroups are what this kernel lives on (a query is one long chain of round trips), so a workgroup gets the LDS
// that lets the register limit decide (~10 KB: the top ~1 000 entries of the candidate heap) and 1 MB of candidate scratch; a
// query whose candidate heap outgrows that is listed by the kernel and searched again by a second, narrow launch with room
// for every point.
int DeviceIndex::run_exact(Workspace& w, const float* d_qpad, const uint32_t* d_qlist, uint32_t nq, uint64_t k, uint64_t ef,
                           const uint32_t* d_allow, uint64_t* d_out_ids, float* d_out_dists, uint8_t* d_out_layer,
                           int32_t* d_out_rank, uint32_t* d_out_counts, uint32_t* stats, void* stream_v, uint32_t* panics,
                           std::string& err, OutLayout layout) {
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const uint32_t tile_bytes = tile_bytes_for(kernel_metric(), v_.row_stride);
    const uint32_t bitmap_words = (v_.n + 31) / 32;
    const uint64_t bm_slice = (uint64_t)bitmap_words * sizeof(uint32_t);
    const int ns = ef <= 64 ? 1 : ef <= 128 ? 2 : 0;  // return_points in VGPRs when it fits (push+pop fused when full)
    const KernelSet& ks = kernel_set(kernel_metric());
    uint32_t panics_total = 0;
    uint32_t work = nq;
    const uint32_t* qlist = d_qlist;
    for (int pass = 0; pass < 2 && work > 0; ++pass) {
        // pass 0: many workgroups with a bounded candidate heap; pass 1: the queries that outgrew it, every point has room
        const uint64_t cand_cap = pass == 0 ? std::min<uint64_t>(v_.n, 1ull << 17) : v_.n;
        const uint64_t heap_stride = ef + 2 + cand_cap;
        const uint64_t per_block = bm_slice + heap_stride * sizeof(hent_t);
        ExactArgs x{};
        // LDS: [query][ids][return_points: ef + 2 entries, or what fits][top of candidate_points]
        const uint64_t lds_fixed = tile_bytes + IDS_BYTES;
        const uint64_t lds_budget = std::max<uint64_t>(10 * 1024, lds_fixed + 4096) - lds_fixed;
        x.r_lds_cap = (uint32_t)std::min<uint64_t>(ef + 2, lds_budget / 2 / sizeof(hent_t));
        x.cand_lds = (uint32_t)std::min<uint64_t>(cand_cap, (lds_budget - (uint64_t)x.r_lds_cap * sizeof(hent_t)) / sizeof(hent_t));
        const size_t lds = lds_fixed + ((size_t)x.r_lds_cap + x.cand_lds) * sizeof(hent_t);
        int per_cu = 0;
        HIP_TRY(ks.exact_occupancy(ns, lds, &per_cu));
        per_cu = std::max(1, per_cu);
        uint32_t grid = (uint32_t)std::min<uint64_t>(work, std::max<uint64_t>(1, (16ull << 30) / per_block));
        grid = std::min<uint32_t>(grid, (uint32_t)num_cu_ * (uint32_t)per_cu);
        HIP_TRY(w.bitmap.ensure((uint64_t)grid * bm_slice));
        HIP_TRY(w.heaps.ensure((uint64_t)grid * heap_stride * sizeof(hent_t)));
        HIP_TRY(w.retry[0].ensure((uint64_t)nq * sizeof(uint32_t)));
        if (knobs().trace_launch)
            std::fprintf(stderr, "[hnswgpu launch] literal kernel, pass %d: %u queries on %u workgroups (%d per CU), %zu bytes of LDS each "
                         "(candidate heap: %u entries in LDS, %llu in all)\n", pass, work, grid, per_cu, lds, x.cand_lds, (unsigned long long)cand_cap);
        SearchArgs a{};
        a.queries = d_qpad;
        a.qlist = qlist;
        a.nq = work;
        a.k = (uint32_t)k;
        a.ef = (uint32_t)ef;
        a.tile_bytes = tile_bytes;
        a.work_counter = static_cast<uint32_t*>(w.d_ctrl);
        a.overflow_count = static_cast<uint32_t*>(w.d_ctrl) + 1;
        a.retry_out = pass == 0 ? w.retry[0].as<uint32_t>() : nullptr;
        a.bitmap = w.bitmap.as<uint32_t>();
        a.bitmap_words = bitmap_words;
        a.bitmap_blocks = grid;
        a.nrm2 = static_cast<const double*>(d_nrm2_);
        a.out_ids = d_out_ids;
        a.out_dists = d_out_dists;
        a.out_layer = d_out_layer;
        a.out_rank = d_out_rank;
        a.out_counts = d_out_counts;
        a.id_stride = layout.id_stride;
        a.dist_stride = layout.dist_stride;
        a.count_stride = layout.count_stride;
        a.stats = stats;
        a.pre = w.pre.as<PreDescent>();
        x.heaps = w.heaps.as<hent_t>();
        x.heap_stride = heap_stride;
        x.cand_cap = (uint32_t)std::min<uint64_t>(cand_cap, 0xFFFFFFFFull);
        x.allow = d_allow;
        HIP_TRY(hipMemsetAsync(w.d_ctrl, 0, 16, stream));
        HIP_TRY(ks.launch_exact(ns, grid, lds, stream, v_, a, x));
        volatile uint32_t* ctrl = static_cast<volatile uint32_t*>(w.h_ctrl);
        HIP_TRY(hipMemcpyAsync(w.h_ctrl, w.d_ctrl, 16, hipMemcpyDeviceToHost, stream));
        HIP_TRY(wait_stream(stream));
        if (ctrl[1] != 0) { err = "internal error in the literal search kernel (a refused point inside return_points, or a candidate heap beyond every point)"; return ERR_DEVICE; }
        panics_total += ctrl[2];
        work = ctrl[3];
        qlist = w.retry[0].as<uint32_t>();
    }
    if (panics) *panics = panics_total;
    return OK;
}

int DeviceIndex::search_device(const float* d_queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef_arg,
                               uint64_t* d_out_ids, float* d_out_dists, uint8_t* d_out_layer, int32_t* d_out_rank,
                               uint32_t* d_out_counts, uint32_t* d_stats, void* stream_v, const uint64_t* d_allowed,
                               uint64_t n_allowed, CallInfo* info_out, std::string& err, const RowFeed* feed, OutLayout layout) {
    if (!ready_) { err = "index is not resident on a device: call hnswgpu_upload first"; return ERR_DEVICE; }
    if (d != v_.d) { err = "query dimension differs from the index dimension"; return ERR_ARG; }
    CallInfo info{};
    auto publish = [&]() {
        if (info_out) *info_out = info;
        std::lock_guard<std::mutex> g(meta_mu_);
        last_ = info;
    };
    if (nq == 0) { publish(); return OK; }
    if (!d_queries || !d_out_ids || !d_out_dists || !d_out_counts) { err = "null buffer"; return ERR_ARG; }
    if (k == 0) { err = "knbn must be > 0"; return ERR_ARG; }
    const uint64_t ef = std::max(ef_arg, k);  // src/hnsw.rs:1531
    if (ef > 0x7FFFFFF0ull) { err = "ef too large"; return ERR_ARG; }
    if (nq > 0xFFFFFFF0ull) { err = "too many queries in one batch"; return ERR_ARG; }
    const bool filtered = d_allowed != nullptr || n_allowed != 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    DeviceGuard on_device(device_);
    HIP_TRY(on_device.status());
    Lease lease(this, acquire(err));
    if (!lease.get()) return ERR_DEVICE;
    Workspace& w = *lease.get();
    // an error return may leave kernels of this call running on the stream: they are waited for BEFORE the workspace goes
    // back to the pool (declared after the lease, destroyed before it)
    struct DrainOnError {
        hipStream_t s;
        bool done = false;
        ~DrainOnError() { if (!done) (void)hipStreamSynchronize(s); }
    } drain{stream};

    HIP_TRY(w.qpad.ensure(nq * v_.row_stride * sizeof(float)));
    if (!d_stats) HIP_TRY(w.stats.ensure(nq * 8 * sizeof(uint32_t)));
    uint32_t* stats = d_stats ? d_stats : w.stats.as<uint32_t>();
    const bool strict_ties = strict_ties_.load();

    HIP_TRY(w.pre.ensure(nq * sizeof(PreDescent)));
    const uint32_t tile_bytes = tile_bytes_for(kernel_metric(), v_.row_stride);
    // (the call's events ride on its kernels: ev_start = start of the first descent launch, ev_ks / ev_ke = the first search launch)
    // first kernel of the call: rows padded to the row stride, the greedy descent of every query (pre[]), counters zeroed
    {
        const KernelSet& ks = kernel_set(kernel_metric());
        // two queries per wavefront where every list above the search layer fits a half's 16 row slots (hnsw_descend_pair_kernel)
        const bool pair = up_deg_max_ <= 16u && kernel_metric() < KM_SIMD8_FIRST && !knobs().no_pair_descent;
        const size_t descend_lds = (pair ? 2u : 1u) * (size_t)tile_bytes + IDS_BYTES;
        int per_cu = descend_per_cu_[pair].load();
        if (per_cu <= 0) {
            HIP_TRY(ks.descend_occupancy(descend_lds, pair, &per_cu));
            per_cu = std::max(1, per_cu);
            descend_per_cu_[pair].store(per_cu);
        }
        // one launch -- or, when the rows are still being gathered into pinned memory, one per chunk: the device reads chunk i
        // across PCIe while the host fills chunk i + 1 (a few chunks: every launch costs a few microseconds of stream time)
        const uint64_t chunk = feed ? std::max<uint64_t>(1, feed->chunk_rows) : nq;
        for (uint64_t lo = 0; lo < nq; lo += chunk) {
            const uint64_t hi = std::min(nq, lo + chunk);
            if (feed) feed->fill(feed->ctx, lo, hi);
            DescendArgs da{};
            da.src = d_queries + lo * v_.d;
            da.qpad = w.qpad.as<float>() + lo * v_.row_stride;
            da.pre = w.pre.as<PreDescent>() + lo;
            da.nq = (uint32_t)(hi - lo);
            da.tile_bytes = tile_bytes;
            da.nrm2 = static_cast<const double*>(d_nrm2_);
            da.ctrl = lo == 0 ? static_cast<uint32_t*>(w.d_ctrl) : nullptr;
            da.ctrl_words = 16;
            da.pair = pair ? 1u : 0u;
            const uint64_t waves = pair ? (hi - lo + 1) / 2 : hi - lo;
            HIP_TRY(ks.launch_descend((uint32_t)std::min<uint64_t>(waves, (uint64_t)per_cu * (uint64_t)num_cu_), stream, v_, da,
                                      LaunchEvents{lo == 0 ? w.ev_start : nullptr, nullptr}));
        }
    }

    // ---- filtered search, and ef beyond the register-resident result set (64 x 16 entries): literal heaps in memory
    if (filtered || ef > 1024) {
        const uint32_t* d_allow = nullptr;
        if (filtered) {
            HIP_TRY(w.allow.ensure((uint64_t)((v_.n + 31) / 32) * sizeof(uint32_t)));
            HIP_TRY(launch_allow_bitmap(stream, v_.origin_id, v_.n, d_allowed, n_allowed, w.allow.as<uint32_t>()));
            d_allow = w.allow.as<uint32_t>();
        }
        HIP_TRY(hipEventRecord(w.ev_ks, stream));
        int rc = run_exact(w, w.qpad.as<float>(), nullptr, (uint32_t)nq, k, ef, d_allow, d_out_ids, d_out_dists, d_out_layer, d_out_rank,
                           d_out_counts, stats, stream, &info.panics, err, layout);
        if (rc != OK) return rc;
        HIP_TRY(hipEventRecord(w.ev_stop, stream));
        HIP_TRY(wait_event(w.ev_stop));
        float ms = 0.f, ms_main = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, w.ev_start, w.ev_stop));
        HIP_TRY(hipEventElapsedTime(&ms_main, w.ev_ks, w.ev_stop));
        info.ms = ms;
        info.main_ms = ms_main;
        info.launches = 1;
        publish();
        drain.done = true;
        return OK;
    }

    HIP_TRY(w.tie.ensure(nq * sizeof(uint32_t)));
    HIP_TRY(w.retry[0].ensure(nq * sizeof(uint32_t)));
    HIP_TRY(w.retry[1].ensure(nq * sizeof(uint32_t)));

    int slots = 1;
    while ((uint64_t)slots * 64 < ef) slots *= 2;
    if (slots == 8) slots = 16;  // kernels are instantiated for 1, 2, 4 and 16 result slots per lane
    // Lists of more than 64 ids (M > 32: none of BASELINE's configs) need a loop over the batches of a list; only the
    // 16-slot kernels carry it -- in the others the single batch is a compile-time fact, worth 1.5-4 % to every search.
    if (v_.deg_stride > 64u) slots = 16;

    // Visited-set sizing.  LDS per wavefront is what bounds occupancy, so the table is sized for the
    // typical query (ef x degree cells, ~2.4x the median number of visited points, measured); the few
    // per cent of queries that outgrow it move to the HBM bitmap inside the same launch.
    const uint32_t idbits = std::max<uint32_t>(1u, ceil_log2(v_.n));
    const uint64_t expect = ef * std::min<uint64_t>(v_.deg_stride, 64);
    uint32_t tbits = std::min<uint32_t>(14u, std::max<uint32_t>(8u, ceil_log2(expect)));
    {   // ... then follows what the previous batches with the same ef measured
        std::lock_guard<std::mutex> g(meta_mu_);
        if (adapt_ef_ == ef && adapt_tbits_ != 0) tbits = adapt_tbits_;
    }
    bool env_forced = false;
    const Knobs& kn = knobs();  // (the environment was read once: search_device.hpp)
    if (kn.hash_bits >= 6 && kn.hash_bits <= 14) { tbits = (uint32_t)kn.hash_bits; env_forced = true; }  // tuning / test hook: initial table size
    const uint32_t tbits_first = tbits;
    const size_t lds_fixed = tile_bytes + IDS_BYTES;
    int table = TABLE_LDS_CELL16;
    bool grown = false;

    // Batch scheduling: the searches run in descending order of the distance to the layer-0 entry point (the descent's
    // result), long searches first (DESIGN.md "scheduling").  Small batches skip it (one launch less, lowest latency).
    const bool scheduled = nq >= 256 && !kn.no_sched;
    if (scheduled) {
        HIP_TRY(w.order.ensure(nq * sizeof(uint32_t)));
        HIP_TRY(kernel_set(kernel_metric()).launch_order(stream, w.pre.as<PreDescent>(), (uint32_t)nq, w.order.as<uint32_t>()));
    }
    uint32_t launches = 0, stop_recorded_after = ~0u;
    uint32_t work = (uint32_t)nq;
    uint32_t n_flagged = 0, n_literal = 0;
    const uint32_t* qlist = scheduled ? w.order.as<uint32_t>() : nullptr;
    int pingpong = 0;
    bool all_done = false;
    // ---- first pass with two queries per wavefront (hnsw_search_pair_kernel, search_pair.inc) where the index and the call allow it:
    // ef <= 128 (4 result slots x 32 lanes), lists of <= 64 ids, 16-bit-cell tables, scalar arithmetic.  It answers every query that
    // meets none of the three places where equal distances make the reference's answer depend on its heaps' order (DESIGN.md section
    // 6); those come back on the retry list and go through the one-query kernels below (strict calls; lean calls flag them like the
    // lean kernel does).  A visited set that outgrows its LDS table moves to an HBM bitmap slice inside the launch.
    // Default (no HNSWGPU_PAIR_SEARCH): where it was measured to pay -- strict calls of tens of thousands of queries on short rows with
    // DistCosine or DistDot (config 3 at 100 000 per call: 9.75 M against 7.93 M queries/s, config 3': 10.30 M against 9.45 M; both
    // cross over at ~30 000 per call; at 10 000 the second launch for the tie queries costs more than the pass gains, rows of several
    // 128-byte lines gain nothing: profiles/r06_pair_search/README.md).
    const bool pair_auto = (kernel_metric() == DIST_COSINE || kernel_metric() == DIST_DOT) && v_.row_stride <= 32u && strict_ties &&
                           nq >= PAIR_SEARCH_AUTO_MIN_QUERIES;
    if (kn.pair_search > 0 || (kn.pair_search < 0 && pair_auto)) {
        const int tbp = (int)tbits + kn.pair_tbits_delta;
        const uint32_t tb = (uint32_t)std::max(6, std::min<int>(tbp, (int)std::min(14u, idbits + 3u)));
        const bool ok = ef <= 128 && v_.deg_stride <= 64u && kernel_metric() < KM_SIMD8_FIRST && nq >= 512 && idbits >= tb - 3u && idbits - (tb - 3u) <= 13u &&
                        idbits - (tb - 3u) >= 1u;
        if (ok) {
            const KernelSet& ks = kernel_set(kernel_metric());
            SearchArgs a{};
            a.tbits = tb;
            a.restbits = idbits - (tb - 3u);
            a.idbits = idbits;
            a.tile_bytes = tile_bytes;
            a.nrm2 = static_cast<const double*>(d_nrm2_);
            const size_t lds = pair_lds_bytes(tile_bytes, tb, (uint32_t)ef);
            int per_cu = 0;
            HIP_TRY(ks.pair_occupancy(lds, &per_cu));
            if (per_cu >= 1) {
                if (kn.pair_wg_per_cu > 0) per_cu = std::min(per_cu, kn.pair_wg_per_cu);
                const uint64_t npairs = (nq + 1) / 2;
                const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)per_cu * (uint64_t)num_cu_, npairs);
                if (kn.trace_launch)
                    std::fprintf(stderr, "[hnswgpu launch] pair pass: %u queries, %d workgroups per CU, %zu bytes of LDS each, tables 2^%u cells\n",
                                 work, per_cu, lds, tb);
                a.queries = w.qpad.as<float>();
                a.qlist = qlist;
                a.nq = work;
                a.k = (uint32_t)k;
                a.ef = (uint32_t)ef;
                a.work_counter = static_cast<uint32_t*>(w.d_ctrl);
                a.overflow_count = static_cast<uint32_t*>(w.d_ctrl) + 1;
                a.retry_out = w.retry[pingpong].as<uint32_t>();
                a.out_ids = d_out_ids;
                a.out_dists = d_out_dists;
                a.out_layer = d_out_layer;
                a.out_rank = d_out_rank;
                a.out_counts = d_out_counts;
                a.id_stride = layout.id_stride;
                a.dist_stride = layout.dist_stride;
                a.count_stride = layout.count_stride;
                a.stats = stats;
                a.pre = w.pre.as<PreDescent>();
                a.tie_list = w.tie.as<uint32_t>();
                a.pair_ties_to_retry = strict_ties ? 1u : 0u;
                {   // HBM bitmaps for the halves whose table fills up: two slices per workgroup, within the 4 GiB budget
                    a.bitmap_words = (v_.n + 31) / 32;
                    const uint64_t slice = (uint64_t)a.bitmap_words * sizeof(uint32_t);
                    const uint64_t blocks = std::min<uint64_t>(2ull * grid, std::max<uint64_t>(2, (4ull << 30) / slice));
                    HIP_TRY(w.bitmap.ensure(blocks * slice));
                    a.bitmap = w.bitmap.as<uint32_t>();
                    a.bitmap_blocks = (uint32_t)blocks;
                }
                HIP_TRY(ks.launch_pair(grid, lds, stream, v_, a, LaunchEvents{w.ev_ks, w.ev_ke}));
                ++launches;
                volatile uint32_t* ctrl = static_cast<volatile uint32_t*>(w.h_ctrl);
                HIP_TRY(hipMemcpyAsync(w.h_ctrl, w.d_ctrl, 24, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipEventRecord(w.ev_stop, stream));
                stop_recorded_after = launches;
                HIP_TRY(wait_stream(stream));
                n_flagged = ctrl[4];
                info.pair_retries = ctrl[1];
                if (table != TABLE_GLOBAL_BITMAP && !env_forced) {  // table sizing feedback, as below
                    uint32_t next = tbits_first;
                    if ((uint64_t)ctrl[2] * 8 > nq && tbits_first < 14u) next = tbits_first + 1;
                    else if ((uint64_t)ctrl[3] * 32 < nq && tbits_first > 8u) next = tbits_first - 1;
                    std::lock_guard<std::mutex> g(meta_mu_);
                    adapt_ef_ = ef;
                    adapt_tbits_ = next;
                }
                if (ctrl[1] == 0) {
                    all_done = true;
                } else {
                    work = ctrl[1];
                    qlist = w.retry[pingpong].as<uint32_t>();
                    pingpong ^= 1;
                }
            }
        }
    }
    while (!all_done) {
        SearchArgs a{};
        size_t lds = lds_fixed;
        if (table != TABLE_GLOBAL_BITMAP) {
            // buckets of 8 cells: the id's top (tb - 3) bits select the bucket, the cell keeps the other restbits bits
            // next to a 2-bit bucket displacement and the valid bit (16-bit cells: restbits <= 13)
            uint32_t tb = std::max(3u, std::min(tbits, idbits + 3u));
            if (idbits - (tb - 3u) <= 13u) {
                table = TABLE_LDS_CELL16;
                a.restbits = idbits - (tb - 3u);
                lds += (size_t)2 << tb;
            } else {
                table = TABLE_LDS_CELL32;
                lds += (size_t)4 << tb;
            }
            a.tbits = tb;
        }
        a.tile_bytes = tile_bytes;
        a.nrm2 = static_cast<const double*>(d_nrm2_);
        a.idbits = idbits;
        const KernelSet& ks = kernel_set(kernel_metric());
        const bool strict_kernel = strict_ties && table != TABLE_GLOBAL_BITMAP && !kn.no_inkernel;
        if (slots <= (strict_kernel ? HNSW_MERGE_SMAX : HNSW_MERGE_LEAN_SMAX)) {  // merge_list's scatter buffer
            a.merge_entries = (uint32_t)slots * 64u + 64u;
            lds += (size_t)a.merge_entries * sizeof(hent_t);
        }
        int per_cu = 0;
        int strict_cap = STRICT_WG_PER_CU;
        if (kn.strict_wg_per_cu > 0) strict_cap = kn.strict_wg_per_cu;  // tuning hook (reported by HNSWGPU_TRACE_LAUNCH)
        if (strict_kernel) {
            // top levels of the (lazy) literal candidate heap, for the few pops that need it: 512 entries when that costs no
            // resident wave (the strict kernel sits at 4 waves per SIMD by its registers, which leaves ~10 KB of LDS per
            // wave), else 256 -- a replay touches the heap ~800 times per query, every level out of LDS is an L2 round trip
            a.cand_lds = 256;
            if (kn.cand_lds >= 0) {
                a.cand_lds = (uint32_t)std::min(4096, kn.cand_lds);  // tuning hook
            } else {
                int occ256 = 0, occ512 = 0;
                HIP_TRY(ks.occupancy(slots, table, true, lds + 256 * sizeof(hent_t), &occ256));
                HIP_TRY(ks.occupancy(slots, table, true, lds + 512 * sizeof(hent_t), &occ512));
                if (std::min(occ512, strict_cap) >= std::min(occ256, strict_cap)) a.cand_lds = 512;
            }
            lds += (size_t)a.cand_lds * sizeof(hent_t);
        }
        HIP_TRY(ks.occupancy(slots, table, strict_kernel, lds, &per_cu));
        if (per_cu < 1) per_cu = 1;
        // A strict launch ends with its longest search -- normally one that replayed its heap-operation log -- and a fifth wave
        // per SIMD slows every expansion of it: with the descent out of the kernel the strict kernel needs 89 VGPRs and would
        // fit 20 workgroups per CU; measured (config 2, one box) 7.63 M queries/s at 20, 8.49 M at 16.  The lean kernel gains
        // from its 20 (9.7 M) and keeps them.
        if (strict_kernel) per_cu = std::min(per_cu, strict_cap);
        if (kn.waves_per_cu > 0) per_cu = std::max(1, std::min(per_cu, kn.waves_per_cu));  // tuning hook
        uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)per_cu * (uint64_t)num_cu_, work);
        if (kn.trace_launch)  // diagnostics: what bounds the resident workgroups of this launch
            std::fprintf(stderr, "[hnswgpu launch] %u queries, %d workgroups per CU (strict cap %d), %zu bytes of LDS each (literal heap: %u entries), table 2^%u cells, strict %d, work list %s\n",
                         work, per_cu, strict_cap, lds, a.cand_lds, a.tbits, (int)strict_kernel,
                         qlist ? "sorted" : "input order");
        a.queries = w.qpad.as<float>();
        a.qlist = qlist;
        a.nq = work;
        a.k = (uint32_t)k;
        a.ef = (uint32_t)ef;
        a.work_counter = static_cast<uint32_t*>(w.d_ctrl);
        a.overflow_count = static_cast<uint32_t*>(w.d_ctrl) + 1;
        a.retry_out = w.retry[pingpong].as<uint32_t>();
        a.out_ids = d_out_ids;
        a.out_dists = d_out_dists;
        a.out_layer = d_out_layer;
        a.out_rank = d_out_rank;
        a.out_counts = d_out_counts;
        a.id_stride = layout.id_stride;
        a.dist_stride = layout.dist_stride;
        a.count_stride = layout.count_stride;
        a.stats = stats;
        a.pre = w.pre.as<PreDescent>();
        {
            // HBM bitmaps for the in-launch fallback: one slice per workgroup, within a 4 GiB budget
            a.bitmap_words = (v_.n + 31) / 32;
            const uint64_t slice = (uint64_t)a.bitmap_words * sizeof(uint32_t);
            uint64_t blocks = std::min<uint64_t>(grid, std::max<uint64_t>(1, (4ull << 30) / slice));
            if (table == TABLE_GLOBAL_BITMAP) grid = (uint32_t)blocks;  // every workgroup needs one
            HIP_TRY(w.bitmap.ensure(blocks * slice));
            a.bitmap = w.bitmap.as<uint32_t>();
            a.bitmap_blocks = (uint32_t)blocks;
        }
        a.tie_list = w.tie.as<uint32_t>();
        if (strict_kernel) {
            // per-workgroup scratch: the heap-operation log, and the part of the literal candidate heap beyond LDS
            const uint32_t cap = 4096;
            const uint64_t need = (uint64_t)grid * cap * sizeof(hent_t);
            HIP_TRY(w.cand.ensure(need));
            HIP_TRY(w.oplog.ensure(need));
            a.cand_scratch = w.cand.as<hent_t>();
            a.cand_cap = cap;
            a.oplog = w.oplog.as<hent_t>();
            a.oplog_cap = cap;
            if (kn.exact_first >= 0) a.exact_first = kn.exact_first != 0 ? 1u : 0u;  // test hook
        }
        // (the first launch finds the counters zeroed by the descent kernel; the flagged list spans relaunches)
        if (launches != 0) HIP_TRY(hipMemsetAsync(w.d_ctrl, 0, 16, stream));
        HIP_TRY(ks.launch_search(slots, table, strict_kernel, grid, lds, stream, v_, a,
                                 launches == 0 ? LaunchEvents{w.ev_ks, w.ev_ke} : LaunchEvents{}));
        ++launches;
        volatile uint32_t* ctrl = static_cast<volatile uint32_t*>(w.h_ctrl);  // pinned: a true asynchronous copy
        HIP_TRY(hipMemcpyAsync(w.h_ctrl, w.d_ctrl, 24, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(w.ev_stop, stream));  // the end of the call unless another launch follows (the usual case: one wait)
        stop_recorded_after = launches;
        HIP_TRY(wait_stream(stream));
        n_flagged = ctrl[4];     // not resolved in the launch (cumulative over relaunches)
        n_literal += ctrl[5];    // resolved with the literal heaps inside the launch
        if (launches == 1 && table != TABLE_GLOBAL_BITMAP && !env_forced && nq >= 256) {
            // Table sizing feedback for the next batch: grow when more than ~1 query in 8 had to move to the
            // HBM bitmap, shrink when a half-size table would have overflowed for fewer than 1 in 32.
            uint32_t next = tbits_first;
            if ((uint64_t)ctrl[2] * 8 > nq && tbits_first < 14u) next = tbits_first + 1;
            else if ((uint64_t)ctrl[3] * 32 < nq && tbits_first > 8u) next = tbits_first - 1;
            std::lock_guard<std::mutex> g(meta_mu_);
            adapt_ef_ = ef;
            adapt_tbits_ = next;
        }
        if (ctrl[1] == 0) break;
        // some queries visited more points than the table holds and had no bitmap slice: rerun only those
        work = ctrl[1];
        qlist = w.retry[pingpong].as<uint32_t>();
        pingpong ^= 1;
        if (table == TABLE_GLOBAL_BITMAP) { err = "internal error: bitmap visited set reported an overflow"; return ERR_DEVICE; }
        if (!grown && tbits < 14u) {
            tbits = std::min<uint32_t>(14u, tbits + 2u);
            grown = true;
        } else {
            table = TABLE_GLOBAL_BITMAP;
        }
    }
    info.ties = n_flagged + n_literal;
    if (strict_ties && n_flagged > 0) {
        // the flagged queries again, with both heaps literal from the first operation on
        int rc = run_exact(w, w.qpad.as<float>(), w.tie.as<uint32_t>(), n_flagged, k, ef, nullptr, d_out_ids, d_out_dists, d_out_layer,
                           d_out_rank, d_out_counts, stats, stream, nullptr, err, layout);
        if (rc != OK) return rc;
        ++launches;
    }
    if (stop_recorded_after != launches) {  // the literal kernel ran after the last recorded end
        HIP_TRY(hipEventRecord(w.ev_stop, stream));
        HIP_TRY(wait_event(w.ev_stop));
    }
    float ms = 0.f, ms_main = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, w.ev_start, w.ev_stop));
    HIP_TRY(hipEventElapsedTime(&ms_main, w.ev_ks, w.ev_ke));  // first launch of the search kernel alone
    info.ms = ms;
    info.main_ms = ms_main;
    info.launches = launches;
    publish();
    drain.done = true;
    return OK;
}

namespace {
// What the threads of a host-buffer call share (search_host_staged): gather tasks cut along the chunks the descent kernel is
// launched on, the hand-over of the answers, unpack tasks.
struct HostCall {
    // gather: task t copies rows [t * task_rows, ...) of its chunk; chunk c is complete when chunk_left[c] reaches 0
    const float* queries;
    const float* const* rows;
    float* hq;
    uint64_t d, nq, chunk_rows, task_rows, tasks_per_chunk, n_chunks;
    std::atomic<uint64_t> next_gather{0};
    std::vector<std::atomic<uint32_t>> chunk_left;
    // answers: 0 = the device is still searching, 2 = all of them in the arena, 3 = the call failed; then unpack tasks.
    // (Unpacking answer by answer WHILE the device searches -- the kernels flagging every finished query in mapped host memory
    // behind a system-scope fence, the helper threads polling -- was built and measured in round 4: the call went from 1.40 to
    // 2.03 ms, the fence being an L2 write-back per query.  Removed.)
    std::atomic<int> phase{0};
    std::atomic<uint64_t> next_unpack{0};
    uint64_t unpack_rows = 256;
    DeviceIndex::HostAnswers answers{};
    const uint32_t* stats = nullptr;   // want_status: [nq][8], status word -> flags[]
    uint8_t* flags = nullptr;
    const DeviceIndex::AnswerSink* sink = nullptr;
    double us_gather = 0., us_unpack = 0.;  // (the caller's share, for HNSWGPU_TRACE_HOST)

    HostCall(uint64_t n_chunks_) : chunk_left(n_chunks_) {}
    bool gather_one() {  // one task, if any is left
        const uint64_t t = next_gather.fetch_add(1, std::memory_order_relaxed);
        if (t >= n_chunks * tasks_per_chunk) return false;
        const uint64_t c = t / tasks_per_chunk, lo = c * chunk_rows + (t % tasks_per_chunk) * task_rows;
        const uint64_t hi = std::min({nq, (c + 1) * chunk_rows, lo + task_rows});
        const uint64_t row_bytes = d * sizeof(float);
        if (lo < hi) {
            if (queries) std::memcpy(hq + lo * d, queries + lo * d, (hi - lo) * row_bytes);
            else for (uint64_t i = lo; i < hi; ++i) std::memcpy(hq + i * d, rows[i], row_bytes);
        }
        chunk_left[c].fetch_sub(1, std::memory_order_release);
        return true;
    }
    void unpack_all() {
        for (;;) {
            const uint64_t lo = next_unpack.fetch_add(unpack_rows, std::memory_order_relaxed);
            if (lo >= nq) break;
            const uint64_t hi = std::min(nq, lo + unpack_rows);
            if (flags) for (uint64_t i = lo; i < hi; ++i) flags[i] = stats[i * 8 + 3] == 6u ? 1 : 0;
            sink->rows(sink->ctx, answers, lo, hi);
        }
    }
    // Busy waiting for as long as the search of a usual batch lasts (4 ms by the clock -- a count of `pause` instructions is 1.2 ms
    // on one CPU and 3 ms on another, and a helper that dozes off just before the answers arrive makes its section end 50-150 us
    // late: measured, the call went from 1.25 to 1.40 ms whenever the search took a little longer), then polite polling.
    // The busy wait is for the lone caller that issues call after call: with several host-buffer calls in flight (concurrent
    // callers on one or more handles) the helpers of all of them would spin on cores the callers' own gathers need -- they then
    // spin for 200 us only and nap.
    static std::atomic<int>& calls_in_flight() {
        static std::atomic<int> n{0};
        return n;
    }
    struct InFlight {
        InFlight() { calls_in_flight().fetch_add(1, std::memory_order_relaxed); }
        ~InFlight() { calls_in_flight().fetch_sub(1, std::memory_order_relaxed); }
    };
    struct Spin {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        unsigned n = 0;
        bool napping = false;
    };
    static void relax(Spin& w) {
        if (!w.napping) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
            if ((++w.n & 255u) == 0u) {
                const auto budget = calls_in_flight().load(std::memory_order_relaxed) <= 1 ? std::chrono::microseconds(4000) : std::chrono::microseconds(200);
                if (std::chrono::steady_clock::now() - w.t0 > budget) w.napping = true;
            }
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
};
}  // namespace

int DeviceIndex::search_host_staged(const float* queries, const float* const* rows, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef,
                                    const uint64_t* allowed, uint64_t n_allowed, bool filtered, bool want_status, const AnswerSink& sink,
                                    CallInfo* info, std::string& err) {
    if (!ready_) { err = "index is not resident on a device: call hnswgpu_upload first"; return ERR_DEVICE; }
    if (nq == 0) { if (info) *info = CallInfo{}; return OK; }
    HostCall::InFlight in_flight;  // (how long this call's helpers may spin depends on how many calls there are: HostCall::relax)
    if ((!queries && !rows) || !sink.rows) { err = "null buffer"; return ERR_ARG; }
    if (d != v_.d) { err = "query dimension differs from the index dimension"; return ERR_ARG; }
    if (k == 0) { err = "knbn must be > 0"; return ERR_ARG; }
    if (filtered && n_allowed && !allowed) { err = "null filter"; return ERR_ARG; }
    DeviceGuard on_device(device_);
    HIP_TRY(on_device.status());
    // the staging buffers live in their own workspace: search_device takes a second one for its scratch
    Lease lease(this, acquire(err));
    if (!lease.get()) return ERR_DEVICE;
    Workspace& w = *lease.get();
    hipStream_t stream = w.own_stream;
    // an error return may leave work of this call in flight on the staging buffers: it is waited for BEFORE the workspace
    // goes back to the pool (declared after the lease, destroyed before it)
    struct DrainStaging {
        hipStream_t s;
        bool done = false;
        ~DrainStaging() { if (!done) (void)hipStreamSynchronize(s); }
    } drain{stream};
    // Nothing is staged in HBM: the queries are gathered into MAPPED pinned memory, which the descent kernel reads across PCIe
    // while it pads them (the H2D copy disappears into a pass that runs anyway), and the search kernels write the answers --
    // ids | dists | rank | layer | counts, 1.7 MB for 10 000 x 10 -- straight into a pinned arena the sink reads.  (Rounds 2-3:
    // gather -> H2D copy -> pad kernel ... -> D2H copy, every step waiting for the one before.)
    const bool trace = knobs().trace_host;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    if (sink.begin && !sink.begin(sink.ctx, nq, k)) { err = "out of memory"; return ERR_ARG; }
    const double us_begin = since(t_begin);
    DirectOut direct{};
    const bool in_place = !want_status && sink.direct && sink.direct(sink.ctx, &direct);
    unsigned char* direct_dev = nullptr;  // the sink's allocation as this device addresses it
    if (in_place) {
        void* dp = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&dp, direct.allocation, 0));
        direct_dev = static_cast<unsigned char*>(dp);
    }
    const double us_view = since(t_begin) - us_begin;
    auto in_sink = [&](void* host) { return direct_dev + (static_cast<unsigned char*>(host) - static_cast<unsigned char*>(direct.allocation)); };
    const uint64_t q_bytes = nq * d * sizeof(float);
    const uint64_t o_ids = 0, o_dists = o_ids + nq * k * sizeof(uint64_t), o_rank = o_dists + nq * k * sizeof(float),
                   o_layer = o_rank + nq * k * sizeof(int32_t), o_cnt = (o_layer + nq * k + 7) & ~7ull,
                   o_ans_end = o_cnt + nq * sizeof(uint32_t), o_stat = (o_ans_end + 7) & ~7ull,
                   o_end = o_stat + (want_status ? nq * 8 * sizeof(uint32_t) + nq : 0);
    HIP_TRY(w.pin_in.ensure(q_bytes));
    HIP_TRY(w.pin_out.ensure(o_end));
    unsigned char* ho = static_cast<unsigned char*>(w.pin_out.p);
    unsigned char* dout = static_cast<unsigned char*>(w.pin_out.dev);  // the same arena as the device addresses it
    const uint64_t* dallowed = nullptr;
    if (filtered) {
        HIP_TRY(w.allowed_ids.ensure(std::max<uint64_t>(1, n_allowed) * sizeof(uint64_t)));
        if (n_allowed) HIP_TRY(hipMemcpyAsync(w.allowed_ids.p, allowed, n_allowed * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        dallowed = w.allowed_ids.as<uint64_t>();
    }
    // how many threads, how the work is cut: ~64 KB per gather task, a few chunks (every chunk is a launch of the descent kernel)
    const uint64_t total_bytes = q_bytes + o_ans_end;
    // (measured, tools/host_call_sweep.py: 2 chunks beat 4 and 1 -- every chunk is a launch of the descent kernel, whose reads across
    // PCIe are what the front of the call waits for, not the gather; 4 to 8 threads are equal, more are slower)
    uint64_t max_threads = 8, max_chunks = 2;
    if (knobs().host_threads > 0) max_threads = (uint64_t)knobs().host_threads;  // tuning hooks
    if (knobs().host_chunks > 0) max_chunks = (uint64_t)knobs().host_chunks;
    const unsigned nt = total_bytes < (256u << 10) || max_threads == 1 ? 1u
                        : (unsigned)std::min<uint64_t>(max_threads, std::max<uint64_t>(2, total_bytes / (256u << 10)));
    const uint64_t n_chunks = nt == 1 ? 1 : std::min<uint64_t>(max_chunks, std::max<uint64_t>(1, nq / 1024));
    HostCall hc(n_chunks);
    hc.queries = queries; hc.rows = rows; hc.hq = static_cast<float*>(w.pin_in.p);
    hc.d = d; hc.nq = nq; hc.n_chunks = n_chunks;
    hc.chunk_rows = (nq + n_chunks - 1) / n_chunks;
    hc.task_rows = std::max<uint64_t>(1, std::min<uint64_t>(hc.chunk_rows, (64u << 10) / std::max<uint64_t>(1, d * sizeof(float))));
    hc.tasks_per_chunk = (hc.chunk_rows + hc.task_rows - 1) / hc.task_rows;
    for (auto& c : hc.chunk_left) c.store((uint32_t)hc.tasks_per_chunk, std::memory_order_relaxed);
    hc.sink = &sink;
    hc.answers.ids = reinterpret_cast<const uint64_t*>(ho + o_ids);
    hc.answers.dists = reinterpret_cast<const float*>(ho + o_dists);
    hc.answers.rank = reinterpret_cast<const int32_t*>(ho + o_rank);
    hc.answers.layer = ho + o_layer;
    hc.answers.counts = reinterpret_cast<const uint32_t*>(ho + o_cnt);
    if (want_status) {
        hc.stats = reinterpret_cast<const uint32_t*>(ho + o_stat);
        hc.flags = ho + o_stat + nq * 8 * sizeof(uint32_t);
        hc.answers.status = hc.flags;
    }
    // the caller's side of the gather, called from inside search_device in front of every launch of the descent kernel:
    // rows [lo, hi) -- one chunk -- must be in place; the caller takes gather tasks itself until they are
    RowFeed feed{[](void* ctx, uint64_t lo, uint64_t) {
                     HostCall& h = *static_cast<HostCall*>(ctx);
                     const auto t0 = std::chrono::steady_clock::now();
                     const uint64_t c = lo / h.chunk_rows;
                     HostCall::Spin spin;
                     while (h.chunk_left[c].load(std::memory_order_acquire) != 0)
                         if (!h.gather_one()) HostCall::relax(spin);
                     h.us_gather += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                 },
                 &hc, hc.chunk_rows};
    int rc = OK;
    double us_search = 0.;
    const std::thread::id caller = std::this_thread::get_id();
    std::atomic<bool> main_taken{false};
    auto participant = [&](unsigned) {
        if (std::this_thread::get_id() == caller && !main_taken.exchange(true)) {
            // the caller: the device side of the call (its launches wait for the chunks), then its share of the unpacking
            const auto t_search = std::chrono::steady_clock::now();
            if (in_place)
                rc = search_device(static_cast<const float*>(w.pin_in.dev), nq, d, k, ef, reinterpret_cast<uint64_t*>(in_sink(direct.ids)),
                                   reinterpret_cast<float*>(in_sink(direct.dists)), nullptr, nullptr, reinterpret_cast<uint32_t*>(in_sink(direct.counts)),
                                   nullptr, stream, dallowed, filtered ? n_allowed : 0, info, err, &feed, direct.layout);
            else
                rc = search_device(static_cast<const float*>(w.pin_in.dev), nq, d, k, ef, reinterpret_cast<uint64_t*>(dout + o_ids),
                                   reinterpret_cast<float*>(dout + o_dists), dout + o_layer, reinterpret_cast<int32_t*>(dout + o_rank),
                                   reinterpret_cast<uint32_t*>(dout + o_cnt), want_status ? reinterpret_cast<uint32_t*>(dout + o_stat) : nullptr,
                                   stream, dallowed, filtered ? n_allowed : 0, info, err, &feed);
            us_search = since(t_search);
            // (search_device returns with the stream idle: the answers are in the arena -- or the call failed, and whatever
            // it left in flight is waited for by `drain`; the gather is then finished by nobody, which is fine)
            hc.phase.store(rc == OK && !in_place ? 2 : 3, std::memory_order_release);  // (in place: nothing to unpack)
            if (rc == OK && !in_place) {
                const auto t0 = std::chrono::steady_clock::now();
                hc.unpack_all();
                hc.us_unpack = since(t0);
            }
            return;
        }
        // a helper: gather tasks while there are any, stay awake while the device searches (also when there is nothing to unpack:
        // the next call of a caller that issues call after call then finds it running), unpack
        while (hc.phase.load(std::memory_order_relaxed) == 0 && hc.gather_one()) {}
        HostCall::Spin spin;
        int ph;
        while ((ph = hc.phase.load(std::memory_order_acquire)) == 0) HostCall::relax(spin);
        if (ph == 2) hc.unpack_all();
    };
    const double us_setup = since(t_begin) - us_begin - us_view;
    if (nt == 1) participant(0);
    else WorkerPool::instance().run(nt, nt, participant);
    if (rc != OK) return rc;
    drain.done = true;
    if (trace)
        std::fprintf(stderr, "[hnswgpu host call] %llu queries on %u threads: %.0f us in all; the caller waited %.0f us for gathered chunks, "
                     "search %.0f us (descent launches included), its share of the unpacking %.0f us; before the pool section: sink.begin %.0f us, "
                     "the sink's memory as the device sees it %.0f us, staging buffers and tasks %.0f us\n",
                     (unsigned long long)nq, nt, since(t_begin), hc.us_gather, us_search, hc.us_unpack, us_begin, us_view, us_setup);
    return OK;
}

int DeviceIndex::search_host(const float* queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef, uint64_t* out_ids,
                             float* out_dists, uint8_t* out_layer, int32_t* out_rank, uint32_t* out_counts,
                             const uint64_t* allowed, uint64_t n_allowed, bool filtered, uint8_t* out_status,
                             CallInfo* info, std::string& err) {
    if (nq != 0 && (!queries || !out_ids || !out_dists || !out_counts)) { err = "null buffer"; return ERR_ARG; }
    struct Out {
        uint64_t k;
        uint64_t* ids; float* dists; uint8_t* layer; int32_t* rank; uint32_t* counts; uint8_t* status;
    } o{k, out_ids, out_dists, out_layer, out_rank, out_counts, out_status};
    AnswerSink sink{nullptr,
                    [](void* ctx, const HostAnswers& a, uint64_t b, uint64_t e) {
                        Out& o = *static_cast<Out*>(ctx);
                        std::memcpy(o.ids + b * o.k, a.ids + b * o.k, (e - b) * o.k * sizeof(uint64_t));
                        std::memcpy(o.dists + b * o.k, a.dists + b * o.k, (e - b) * o.k * sizeof(float));
                        if (o.layer) std::memcpy(o.layer + b * o.k, a.layer + b * o.k, (e - b) * o.k);
                        if (o.rank) std::memcpy(o.rank + b * o.k, a.rank + b * o.k, (e - b) * o.k * sizeof(int32_t));
                        std::memcpy(o.counts + b, a.counts + b, (e - b) * sizeof(uint32_t));
                        if (o.status && a.status) std::memcpy(o.status + b, a.status + b, e - b);
                    },
                    &o};
    return search_host_staged(queries, nullptr, nq, d, k, ef, allowed, n_allowed, filtered, out_status != nullptr, sink, info, err);
}


// ---------------------------------------------------------------------------------------------------------------------
// GPU-assisted construction: the device holds every vector of the build and a snapshot of the neighbour lists (one
// fixed-stride array per layer, builder ids); search_window runs hnsw_build_search_kernel for a window of new points,
// patch() brings the snapshot up to date with what the host did with the results (builder.cpp).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
class DeviceBuildBackend : public BuildSearchBackend {
public:
    explicit DeviceBuildBackend(int device) : device_(device) {}
    ~DeviceBuildBackend() override {
        DeviceGuard on_device(device_);
        for (DevBuf* b : {&vec_, &level_, &nrm2_, &slot0_, &out_ids_, &out_d_, &out_n_, &hit_ids_, &hit_d_, &bitmap_, &upd_, &slot_nb_, &sel_ids_, &sel_d_, &sel_n_}) b->free();
        for (auto& b : lists_) b.free();
        upd_host_.free();
        if (d_ctrl_) (void)hipFree(d_ctrl_);
    }
    int check(uint64_t ef_construction, std::string& err) override {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { err = "no HIP device visible (GPU-assisted construction needs a gfx950 GPU)"; return ERR_DEVICE; }
        if (device_ < 0 || device_ >= ndev) { err = "bad device ordinal"; return ERR_ARG; }
        if (ef_construction > 1024) { err = "GPU-assisted construction supports ef_construction up to 1024"; return ERR_ARG; }
        return OK;
    }
    int begin(const float* const* chunks, uint64_t chunk_rows, uint64_t n, uint64_t d, const uint8_t* levels, int dist,
              uint64_t max_nb_connection, uint64_t ef_construction, unsigned top_layer, uint64_t max_window, std::string& err) override {
        const int crc = check(ef_construction, err);
        if (crc != OK) return crc;
        DeviceGuard on_device(device_);
        HIP_TRY(on_device.status());
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device_));
        num_cu_ = prop.multiProcessorCount;
        n_ = n;
        dist_ = dist;
        ef_c_ = (uint32_t)ef_construction;
        row_stride_ = (uint32_t)((d + 31) / 32 * 32);
        max_window_ = (uint32_t)std::max<uint64_t>(1, max_window);
        top_layer_ = std::min<unsigned>(top_layer, NB_LAYER_MAX - 1);
        // vectors, padded rows
        HIP_TRY(vec_.ensure(n * row_stride_ * sizeof(float)));
        HIP_TRY(hipMemset(vec_.p, 0, n * row_stride_ * sizeof(float)));
        for (uint64_t r0 = 0; r0 < n; r0 += chunk_rows) {
            const uint64_t rows = std::min<uint64_t>(chunk_rows, n - r0);
            HIP_TRY(hipMemcpy2D(vec_.as<float>() + r0 * row_stride_, row_stride_ * sizeof(float), chunks[r0 / chunk_rows], d * sizeof(float),
                                d * sizeof(float), rows, hipMemcpyHostToDevice));
        }
        HIP_TRY(level_.ensure(n));
        HIP_TRY(hipMemcpy(level_.p, levels, n, hipMemcpyHostToDevice));
        if (dist == DIST_COSINE) {
            if (!norm_fits_row(dist, (uint32_t)d, row_stride_)) HIP_TRY(nrm2_.ensure(n * sizeof(double)));
            HIP_TRY(launch_row_sq_norms(nullptr, vec_.as<float>(), nrm2_.as<double>(), (uint32_t)n, (uint32_t)d, row_stride_));
        }
        // neighbour lists: layer 0 holds up to 2M ids, the others up to M
        bl_ = BuildLists{};
        max_stride_ = 0;
        for (unsigned l = 0; l <= top_layer_; ++l) {
            const uint32_t stride = (uint32_t)(((l == 0 ? 2 : 1) * max_nb_connection + 15) / 16 * 16);
            HIP_TRY(lists_[l].ensure(n * stride * sizeof(uint32_t)));
            HIP_TRY(hipMemset(lists_[l].p, 0xFF, n * stride * sizeof(uint32_t)));
            bl_.lists[l] = lists_[l].as<uint32_t>();
            bl_.stride[l] = stride;
            max_stride_ = std::max(max_stride_, stride);
        }
        HIP_TRY(hipMalloc(&d_ctrl_, 64));
        HIP_TRY(hipDeviceSynchronize());
        return OK;
    }
    uint32_t rec_words() const override { return 2u + max_stride_; }
    uint32_t* patch_buffer(uint64_t n_records, std::string& err) override {
        DeviceGuard on_device(device_);
        if (on_device.status() != hipSuccess) { err = std::string("selecting the build device: ") + hipGetErrorString(on_device.status()); return nullptr; }
        const hipError_t e = upd_host_.ensure(std::max<uint64_t>(1, n_records) * rec_words() * sizeof(uint32_t));
        if (e != hipSuccess) { err = std::string("pinned buffer for the list updates: ") + hipGetErrorString(e); return nullptr; }
        return static_cast<uint32_t*>(upd_host_.p);
    }
    int patch(uint64_t n_records, std::string& err) override {
        if (n_records == 0) return OK;
        DeviceGuard on_device(device_);
        HIP_TRY(on_device.status());
        const uint32_t rw = rec_words();
        const uint32_t* records = static_cast<const uint32_t*>(upd_host_.p);
        if (!records || upd_host_.cap < n_records * rw * sizeof(uint32_t)) { err = "internal error: list updates were not packed into patch_buffer"; return ERR_ARG; }
        if (n_records > 0xFFFFFFFFull) { err = "too many list updates"; return ERR_ARG; }
        for (uint64_t u = 0; u < n_records; ++u)
            if (records[u * rw + 1] > top_layer_ || records[u * rw] >= n_) { err = "internal error: list update outside the snapshot"; return ERR_ARG; }
        HIP_TRY(upd_.ensure(n_records * rw * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(upd_.p, records, n_records * rw * sizeof(uint32_t), hipMemcpyHostToDevice, nullptr));  // pinned: a DMA at link speed
        HIP_TRY(launch_scatter_lists(nullptr, upd_.as<uint32_t>(), (uint32_t)n_records, rw, bl_));
        HIP_TRY(hipDeviceSynchronize());
        return OK;
    }
    int search_window(uint32_t first, uint32_t count, uint32_t entry, uint32_t entry_level, uint32_t layer_mask,
                      const WindowSelect& select, WindowSearchResults& out, std::string& err) override {
        DeviceGuard on_device(device_);
        HIP_TRY(on_device.status());
        if (entry_level > top_layer_) { err = "internal error: entry point above the snapshot's layers"; return ERR_ARG; }
        // output slots: one per (point, layer <= min(level, entry level))
        levels_h_.resize(count);
        HIP_TRY(hipMemcpy(levels_h_.data(), level_.as<uint8_t>() + first, count, hipMemcpyDeviceToHost));
        out.slot0.resize(count);
        uint64_t slots = 0;
        for (uint32_t i = 0; i < count; ++i) {
            out.slot0[i] = (uint32_t)slots;
            slots += std::min<uint32_t>(levels_h_[i], entry_level) + 1u;
        }
        HIP_TRY(slot0_.ensure(count * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(slot0_.p, out.slot0.data(), count * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(out_ids_.ensure(slots * ef_c_ * sizeof(uint32_t)));
        HIP_TRY(out_d_.ensure(slots * ef_c_ * sizeof(float)));
        HIP_TRY(out_n_.ensure(slots * sizeof(uint32_t)));
        HIP_TRY(hit_ids_.ensure((uint64_t)count * NB_LAYER_MAX * sizeof(uint32_t)));
        HIP_TRY(hit_d_.ensure((uint64_t)count * NB_LAYER_MAX * sizeof(float)));
        HIP_TRY(hipMemset(hit_ids_.p, 0xFF, (uint64_t)count * NB_LAYER_MAX * sizeof(uint32_t)));

        BuildArgs a{};
        a.vec = vec_.as<float>();
        a.row_stride = row_stride_;
        for (unsigned l = 0; l < NB_LAYER_MAX; ++l) { a.lists[l] = bl_.lists[l]; a.stride[l] = bl_.stride[l]; }
        a.level = level_.as<uint8_t>();
        a.slot0 = slot0_.as<uint32_t>();
        a.first = first;
        a.count = count;
        a.entry = entry;
        a.entry_level = entry_level;
        a.layer_mask = layer_mask;
        a.ef_c = ef_c_;
        int slots_per_lane = 1;
        while ((uint32_t)slots_per_lane * 64u < ef_c_) slots_per_lane *= 2;
        if (slots_per_lane == 8) slots_per_lane = 16;
        // visited table: ef_c x degree cells (the construction search visits about that many points), 16-bit cells
        const uint32_t idbits = std::max<uint32_t>(1u, ceil_log2(n_));
        uint32_t tbits = std::min<uint32_t>(14u, std::max<uint32_t>(8u, ceil_log2((uint64_t)ef_c_ * std::min<uint32_t>(bl_.stride[0], 64u))));
        tbits = std::max(3u, std::min(tbits, idbits + 3u));
        while (idbits - (tbits - 3u) > 13u && tbits < 16u) ++tbits;  // (16-bit cells keep at most 13 id bits)
        if (idbits - (tbits - 3u) > 13u) { err = "GPU-assisted construction: index too large for the 16-bit visited cells"; return ERR_ARG; }
        a.tbits = tbits;
        a.idbits = idbits;
        a.restbits = idbits - (tbits - 3u);
        a.tile_bytes = tile_bytes_for(dist_, row_stride_);
        const size_t lds = a.tile_bytes + IDS_BYTES + ((size_t)2 << tbits);
        const KernelSet& ks = kernel_set(dist_);
        int per_cu = 0;
        HIP_TRY(ks.build_occupancy(slots_per_lane, lds, &per_cu));
        if (per_cu < 1) per_cu = 1;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)per_cu * (uint64_t)num_cu_, count);
        a.bitmap_words = (uint32_t)((n_ + 31) / 32);
        HIP_TRY(bitmap_.ensure((uint64_t)grid * a.bitmap_words * sizeof(uint32_t)));
        a.bitmap = bitmap_.as<uint32_t>();
        a.bitmap_blocks = grid;
        a.work_counter = static_cast<uint32_t*>(d_ctrl_);
        a.fail_count = static_cast<uint32_t*>(d_ctrl_) + 1;
        a.out_ids = out_ids_.as<uint32_t>();
        a.out_d = out_d_.as<float>();
        a.out_n = out_n_.as<uint32_t>();
        a.hit_ids = hit_ids_.as<uint32_t>();
        a.hit_d = hit_d_.as<float>();
        a.nrm2 = nrm2_.as<double>();
        HIP_TRY(hipMemset(d_ctrl_, 0, 8));
        HIP_TRY(ks.launch_build_search(slots_per_lane, grid, lds, nullptr, a));
        uint32_t ctrl[2] = {0, 0};
        HIP_TRY(hipMemcpy(ctrl, d_ctrl_, 8, hipMemcpyDeviceToHost));
        if (ctrl[1] != 0) { err = "internal error: visited set overflow in the construction search"; return ERR_DEVICE; }
        out.hit_ids.resize((size_t)count * NB_LAYER_MAX);
        out.hit_d.resize((size_t)count * NB_LAYER_MAX);
        HIP_TRY(hipMemcpy(out.hit_ids.data(), hit_ids_.p, (size_t)count * NB_LAYER_MAX * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out.hit_d.data(), hit_d_.p, (size_t)count * NB_LAYER_MAX * sizeof(float), hipMemcpyDeviceToHost));
        out.selected = false;
        out.sel_stride = 0;
        const uint32_t sel_stride = std::max(select.nb_layer0, select.nb_upper);
        if (select.on_device && slots > 0 && sel_stride > 0 && sel_stride <= 0xFFFFu) {
            // select_neighbours for every slot, on the candidates the searches just left in HBM (hnsw_build_select_kernel): what
            // comes back over PCIe is the selected lists (<= 2 M entries per slot) instead of ef_construction candidates
            slot_nb_h_.assign(slots, 0);
            for (uint32_t i = 0; i < count; ++i) {
                const uint32_t top = std::min<uint32_t>(levels_h_[i], entry_level);
                for (uint32_t l = 0; l <= top; ++l) slot_nb_h_[out.slot0[i] + l] = (uint16_t)(l == 0 ? select.nb_layer0 : select.nb_upper);
            }
            HIP_TRY(slot_nb_.ensure(slots * sizeof(uint16_t)));
            HIP_TRY(hipMemcpy(slot_nb_.p, slot_nb_h_.data(), slots * sizeof(uint16_t), hipMemcpyHostToDevice));
            HIP_TRY(sel_ids_.ensure(slots * sel_stride * sizeof(uint32_t)));
            HIP_TRY(sel_d_.ensure(slots * sel_stride * sizeof(float)));
            HIP_TRY(sel_n_.ensure(slots * sizeof(uint32_t)));
            SelectArgs sa{};
            sa.vec = vec_.as<float>();
            sa.row_stride = row_stride_;
            sa.tile_bytes = a.tile_bytes;
            sa.nrm2 = nrm2_.as<double>();
            sa.cand_ids = out_ids_.as<uint32_t>();
            sa.cand_d = out_d_.as<float>();
            sa.cand_n = out_n_.as<uint32_t>();
            sa.ef_c = ef_c_;
            sa.slot_nb = slot_nb_.as<uint16_t>();
            sa.n_slots = (uint32_t)slots;
            sa.sel_stride = sel_stride;
            sa.keep_pruned = select.keep_pruned ? 1u : 0u;
            sa.sel_ids = sel_ids_.as<uint32_t>();
            sa.sel_d = sel_d_.as<float>();
            sa.sel_n = sel_n_.as<uint32_t>();
            sa.work_counter = static_cast<uint32_t*>(d_ctrl_);
            HIP_TRY(hipMemset(d_ctrl_, 0, 8));
            const size_t sel_lds = a.tile_bytes + IDS_BYTES + (size_t)sel_stride * 8u;
            const uint32_t sgrid = (uint32_t)std::min<uint64_t>((uint64_t)num_cu_ * 24u, slots);
            HIP_TRY(ks.launch_build_select(sgrid, sel_lds, nullptr, sa));
            // (a kernel that faults is reported by the synchronising copies below at the latest; asked here so that the
            // message names the kernel, and before sel_n is trusted)
            HIP_TRY(hipStreamSynchronize(nullptr));
            out.sel_ids.resize(slots * sel_stride);
            out.sel_d.resize(slots * sel_stride);
            out.sel_n.resize(slots);
            HIP_TRY(hipMemcpy(out.sel_ids.data(), sel_ids_.p, slots * sel_stride * sizeof(uint32_t), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(out.sel_d.data(), sel_d_.p, slots * sel_stride * sizeof(float), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(out.sel_n.data(), sel_n_.p, slots * sizeof(uint32_t), hipMemcpyDeviceToHost));
            out.out_ids.clear();
            out.out_d.clear();
            out.out_n.clear();
            out.selected = true;
            out.sel_stride = sel_stride;
            for (uint64_t sidx = 0; sidx < slots; ++sidx)  // a selection longer than its slot: never read past it
                if (out.sel_n[sidx] > sel_stride) { err = "internal error: select_neighbours on the device returned more entries than a slot holds"; return ERR_DEVICE; }
            return OK;
        }
        out.out_ids.resize(slots * ef_c_);
        out.out_d.resize(slots * ef_c_);
        out.out_n.resize(slots);
        HIP_TRY(hipMemcpy(out.out_ids.data(), out_ids_.p, slots * ef_c_ * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out.out_d.data(), out_d_.p, slots * ef_c_ * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(out.out_n.data(), out_n_.p, slots * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return OK;
    }

private:
    int device_;
    int num_cu_ = 0;
    int dist_ = DIST_L2;
    uint64_t n_ = 0;
    uint32_t ef_c_ = 0, row_stride_ = 0, max_window_ = 1, max_stride_ = 0;
    unsigned top_layer_ = 0;
    DevBuf vec_, level_, nrm2_, slot0_, out_ids_, out_d_, out_n_, hit_ids_, hit_d_, bitmap_, upd_, slot_nb_, sel_ids_, sel_d_, sel_n_;
    std::vector<uint16_t> slot_nb_h_;
    PinnedBuf upd_host_;
    DevBuf lists_[NB_LAYER_MAX];
    BuildLists bl_{};
    void* d_ctrl_ = nullptr;
    std::vector<uint8_t> levels_h_;
};
}  // namespace

std::unique_ptr<BuildSearchBackend> make_device_build_backend(int device) {
    return std::unique_ptr<BuildSearchBackend>(new DeviceBuildBackend(device));
}

int eval_distance_matrix_device(int dist, const float* queries, uint64_t nq, const float* rows, uint64_t n, uint64_t d,
                                uint32_t nf, bool pairs, float* out, std::string& err, int arithmetic) {
    if (nq == 0 || n == 0) return OK;
    int km = dist;
    if (arithmetic == ARITH_SIMD8) {
        km = simd8_kernel_metric(dist);
        if (km < 0) { err = "this distance has no SIMD-order variant"; return ERR_ARG; }
    }
    if (!pairs && (nf < 1 || nf > 64)) { err = "nf must be in 1..64"; return ERR_ARG; }
    if (pairs && nq != n) { err = "pair mode needs as many queries as rows"; return ERR_ARG; }
    if (!pairs && nq > 65535) { err = "too many queries for one launch"; return ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { err = "no HIP device visible"; return ERR_DEVICE; }
    const uint32_t rs = (uint32_t)((d + 31) / 32 * 32);
    std::vector<float> pq((size_t)nq * rs, 0.f), pr((size_t)n * rs, 0.f);
    for (uint64_t i = 0; i < nq; ++i) std::memcpy(pq.data() + i * rs, queries + i * d, d * sizeof(float));
    for (uint64_t i = 0; i < n; ++i) std::memcpy(pr.data() + i * rs, rows + i * d, d * sizeof(float));
    DevBuf dq, dr, dn, dout;
    struct Free { DevBuf *a, *b, *c, *e; ~Free() { a->free(); b->free(); c->free(); e->free(); } } guard{&dq, &dr, &dn, &dout};
    const uint64_t n_out = pairs ? n : nq * n;
    HIP_TRY(dq.ensure(pq.size() * sizeof(float)));
    HIP_TRY(dr.ensure(pr.size() * sizeof(float)));
    HIP_TRY(dout.ensure(n_out * sizeof(float)));
    HIP_TRY(hipMemcpy(dq.p, pq.data(), pq.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dr.p, pr.data(), pr.size() * sizeof(float), hipMemcpyHostToDevice));
    if (dist == DIST_COSINE) {  // the layout the search uses for this dimension: norm in the row, or beside it
        if (!norm_fits_row(dist, (uint32_t)d, rs)) HIP_TRY(dn.ensure(n * sizeof(double)));
        HIP_TRY(launch_row_sq_norms(nullptr, dr.as<float>(), dn.as<double>(), (uint32_t)n, (uint32_t)d, rs));
    }
    if (pairs) {
        for (uint64_t q0 = 0; q0 < n; q0 += 32768) {  // gridDim.y is limited to 65535
            const uint32_t cnt = (uint32_t)std::min<uint64_t>(32768, n - q0);
            HIP_TRY(kernel_set(km).launch_eval_matrix(nullptr, dq.as<float>() + q0 * rs, cnt, dr.as<float>() + q0 * rs, cnt,
                                                      dn.p ? dn.as<double>() + q0 : nullptr, dout.as<float>() + q0, rs, (uint32_t)d, 1, true));
        }
    } else {
        HIP_TRY(kernel_set(km).launch_eval_matrix(nullptr, dq.as<float>(), (uint32_t)nq, dr.as<float>(), (uint32_t)n, dn.as<double>(),
                                                  dout.as<float>(), rs, (uint32_t)d, nf, false));
    }
    HIP_TRY(hipMemcpy(out, dout.p, n_out * sizeof(float), hipMemcpyDeviceToHost));
    return OK;
}

}  // namespace hnswgpu
