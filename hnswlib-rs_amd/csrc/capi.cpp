// capi.cpp -- the C ABI declared in include/hnsw_mi355x.h: the thin hnswgpu_* entry points and the
// name/layout-compatible replacements of the reference's own f32 FFI (src/libext.rs).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hnsw_mi355x.h"
#include "builder.hpp"
#include "datamap.hpp"
#include "flat_index.hpp"
#include "hnswio.hpp"
#include "search_device.hpp"
#include "worker_pool.hpp"

using namespace hnswgpu;

static thread_local std::string g_last_error;
static int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
// worker threads of one call: whatever was spawned is joined when this goes out of scope -- also when a later spawn throws,
// so that no joinable std::thread is ever destroyed (that would be std::terminate, past every guard)
struct JoinAll {
    std::vector<std::thread> th;
    template <class F> void spawn(F&& f) { th.emplace_back(std::forward<F>(f)); }
    ~JoinAll() {
        for (auto& t : th)
            if (t.joinable()) t.join();
    }
};
// a call that SUCCEEDED with something worth telling (hnswgpu_last_error() returns it until the next failure or note)
static void note(const std::string& msg) { g_last_error = msg; }
// No C++ exception may cross the C ABI (the host may be Rust, Julia or C): every entry point runs inside this guard.
#define CAPI_GUARD_BEGIN try {
#define CAPI_GUARD_END(ret)                                                             \
    } catch (const std::bad_alloc&) {                                                   \
        fail(HNSWGPU_ERR_ARG, "out of memory");                                         \
        return ret;                                                                     \
    } catch (const std::exception& e) {                                                 \
        fail(HNSWGPU_ERR_FORMAT, std::string("internal error: ") + e.what());           \
        return ret;                                                                     \
    } catch (...) {                                                                     \
        fail(HNSWGPU_ERR_FORMAT, "internal error");                                     \
        return ret;                                                                     \
    }

struct hnswgpu_index {
    // Searches take this lock shared (they are `&self` in the reference: concurrent calls on one handle are legal);
    // whatever changes the graph or the set of replicas (insert, upload, dump of a stale flat view) takes it exclusive.
    std::shared_mutex mu;
    std::unique_ptr<FlatIndex> flat;        // dump-order view; rebuilt from `builder` when stale
    std::unique_ptr<GraphBuilder> builder;  // construction state (created lazily for reloaded indexes)
    bool flat_stale = false;
    std::map<int, std::unique_ptr<DeviceIndex>> replicas;  // HBM replicas by HIP device ordinal
    int primary = -1;                       // device of the single-GPU entry points
    bool dev_stale = true;
    int strict_ties = -1;  // -1: library default (env HNSWGPU_STRICT_TIES, else on)
    int arithmetic = 0;    // HNSWGPU_ARITH_*
    BuildParams params;

    const FlatIndex* get_flat() {  // exclusive lock held (or the view is known to be fresh)
        if (builder && (flat_stale || !flat)) {
            flat.reset(new FlatIndex());
            builder->finalize(*flat);
            flat_stale = false;
            dev_stale = true;
        }
        return flat.get();
    }
    bool fresh() const { return flat && !flat_stale; }
    DeviceIndex* replica(int device) const {
        auto it = replicas.find(device);
        return it == replicas.end() || !it->second->ready() ? nullptr : it->second.get();
    }
};

static int default_device() {
    const char* e = std::getenv("HNSWGPU_DEVICE");
    return e ? std::atoi(e) : 0;
}

// make sure an HBM replica on `device` (< 0: the primary / default one) reflects the host graph.  Exclusive lock held.
static int ensure_device(hnswgpu_index* idx, int device) {
    const FlatIndex* f = idx->get_flat();
    if (!f || f->n == 0) return fail(HNSWGPU_ERR_EMPTY, "index is empty");
    if (idx->dev_stale) {  // the graph changed: every replica is out of date
        idx->replicas.clear();
        idx->dev_stale = false;
    }
    if (device < 0) device = idx->primary >= 0 ? idx->primary : default_device();
    if (!idx->replica(device)) {
        std::unique_ptr<DeviceIndex> dev(new DeviceIndex());
        std::string err;
        int rc = dev->upload(*f, device, err);
        if (rc != OK) return fail(rc, err);
        if (idx->strict_ties >= 0) dev->set_strict_ties(idx->strict_ties != 0);
        dev->set_arithmetic(idx->arithmetic);
        idx->replicas[device] = std::move(dev);
    }
    if (idx->primary < 0 || !idx->replica(idx->primary)) idx->primary = device;
    return HNSWGPU_OK;
}
// the replica a search on the primary device uses; takes the exclusive lock only when something has to be (re)built
static int primary_replica(hnswgpu_index* idx, std::shared_lock<std::shared_mutex>& sl, DeviceIndex** out) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (idx->fresh() && !idx->dev_stale && idx->primary >= 0 && idx->replica(idx->primary)) {
            *out = idx->replica(idx->primary);
            return HNSWGPU_OK;
        }
        sl.unlock();
        {
            std::unique_lock<std::shared_mutex> xl(idx->mu);
            int rc = ensure_device(idx, -1);
            if (rc != HNSWGPU_OK) { xl.unlock(); sl.lock(); return rc; }
        }
        sl.lock();
    }
    return fail(HNSWGPU_ERR_DEVICE, "index changed while a search was starting");
}

extern "C" {

const char* hnswgpu_last_error(void) { return g_last_error.c_str(); }

int hnswgpu_load_dump(const char* dir, const char* basename, int dist, hnswgpu_index** out) {
    CAPI_GUARD_BEGIN
    if (!dir || !basename || !out) return fail(HNSWGPU_ERR_ARG, "null argument");
    *out = nullptr;
    std::unique_ptr<hnswgpu_index> h(new hnswgpu_index());
    h->flat.reset(new FlatIndex());
    std::string err;
    int rc = load_dump(dir, basename, dist, *h->flat, err);
    if (rc != OK) return fail(rc, err);
    *out = h.release();
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_FORMAT)
}

int hnswgpu_file_dump(const hnswgpu_index* cidx, const char* dir, const char* basename) {
    CAPI_GUARD_BEGIN
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || !dir || !basename) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::unique_lock<std::shared_mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    if (!f) return fail(HNSWGPU_ERR_EMPTY, "entry point not initialized");
    std::string err;
    int rc = write_dump(*f, dir, basename, err);
    if (rc != OK) return fail(rc, err);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_IO)
}

void hnswgpu_free_index(hnswgpu_index* idx) { delete idx; }

static void fill_descr(hnswgpu_description* o, uint32_t ver, uint8_t mode, uint8_t m, uint8_t nbl, double ls, uint64_t ef,
                       uint64_t nbp, uint64_t dim, const std::string& dn, const std::string& tn) {
    std::memset(o, 0, sizeof(*o));
    o->format_version = ver;
    o->dumpmode = mode;
    o->max_nb_connection = m;
    o->nb_layer = nbl;
    o->level_scale = ls;
    o->ef_construction = ef;
    o->nb_point = nbp;
    o->dimension = dim;
    std::strncpy(o->distname, dn.c_str(), sizeof(o->distname) - 1);
    std::strncpy(o->t_name, tn.c_str(), sizeof(o->t_name) - 1);
}

int hnswgpu_load_description(const char* graph_file_path, hnswgpu_description* out) {
    CAPI_GUARD_BEGIN
    if (!graph_file_path || !out) return fail(HNSWGPU_ERR_ARG, "null argument");
    DumpDescription d;
    std::string err;
    int rc = load_description_file(graph_file_path, d, err);
    if (rc != OK) return fail(rc, err);
    fill_descr(out, d.format_version, d.dumpmode, d.max_nb_connection, d.nb_layer, d.level_scale, d.ef, d.nb_point,
               d.dimension, d.distname, d.t_name);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_FORMAT)
}

int hnswgpu_get_description(const hnswgpu_index* cidx, hnswgpu_description* out) {
    CAPI_GUARD_BEGIN
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || !out) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::unique_lock<std::shared_mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    if (!f) return fail(HNSWGPU_ERR_EMPTY, "index is empty");
    fill_descr(out, f->format_version, f->dumpmode, (uint8_t)f->max_nb_connection, f->nb_layer, f->level_scale,
               f->ef_construction, f->n, f->dimension, f->distname.empty() ? dist_type_name(f->dist) : f->distname, f->t_name);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_FORMAT)
}

static BuildParams to_params(const hnswgpu_build_params* p) {
    BuildParams b;
    b.max_nb_connection = p->max_nb_connection;
    b.ef_construction = p->ef_construction;
    b.max_layer = p->max_layer ? p->max_layer : 16;
    b.dist = p->dist;
    b.level_scale_factor = p->level_scale_factor > 0 ? p->level_scale_factor : 1.0;
    b.extend_candidates = p->extend_candidates != 0;
    b.keep_pruned = p->keep_pruned != 0;
    b.nthreads = p->nthreads;
    b.fast_arithmetic = p->fast_arithmetic != 0;
    b.gpu_device = p->gpu_assist ? p->gpu_device : -1;
    b.gpu_window = p->gpu_window;
    return b;
}

int hnswgpu_build(const float* data, uint64_t n, uint64_t d, const uint64_t* ids, const hnswgpu_build_params* params,
                  hnswgpu_index** out) {
    CAPI_GUARD_BEGIN
    if (!params || !out || (n && !data)) return fail(HNSWGPU_ERR_ARG, "null argument");
    *out = nullptr;
    if (params->dist < 0 || params->dist >= DIST_COUNT) return fail(HNSWGPU_ERR_DISTANCE, "unknown distance");
    if (params->max_nb_connection < 2 || params->max_nb_connection > 256)
        return fail(HNSWGPU_ERR_ARG, "error max_nb_connection must be less equal than 256");  // src/hnsw.rs:784-787
    std::unique_ptr<hnswgpu_index> h(new hnswgpu_index());
    h->params = to_params(params);
    h->builder.reset(new GraphBuilder(h->params));
    std::string err;
    int rc;
    if (h->params.gpu_device >= 0) {
        std::unique_ptr<BuildSearchBackend> dev = make_device_build_backend(h->params.gpu_device);
        rc = h->builder->insert_batch_gpu(data, n, d, ids, h->params.nthreads, *dev, h->params.gpu_window, err);
        if (rc == OK && !h->builder->last_warning().empty()) note(h->builder->last_warning());
    } else {
        rc = h->builder->insert_batch(data, n, d, ids, h->params.nthreads, err);
    }
    if (rc != OK) return fail(rc, err);
    h->flat_stale = true;
    *out = h.release();
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_ARG)
}

// Hnsw::insert / parallel_insert on ANY handle, reloaded ones included (HnswIo::load_hnsw returns a fully insertable
// Hnsw).  Exclusive lock held by the caller.
static int insert_points(hnswgpu_index* idx, const float* data, uint64_t n, uint64_t d, const uint64_t* ids, int nthreads,
                         int gpu_device = -1, uint64_t gpu_window = 0) {
    if (!idx->builder) {
        if (!idx->flat) return fail(HNSWGPU_ERR_EMPTY, "handle holds no index");
        idx->builder.reset(new GraphBuilder(*idx->flat, false));  // continue the reloaded graph (builder.hpp)
        idx->params = idx->builder->params();
    }
    std::string err;
    int rc;
    if (gpu_device >= 0) {
        std::unique_ptr<BuildSearchBackend> dev = make_device_build_backend(gpu_device);
        rc = idx->builder->insert_batch_gpu(data, n, d, ids, nthreads, *dev, gpu_window, err);
        if (rc == OK && !idx->builder->last_warning().empty()) note(idx->builder->last_warning());
    } else {
        rc = idx->builder->insert_batch(data, n, d, ids, nthreads, err);
    }
    if (rc != OK) return fail(rc, err);
    idx->flat_stale = true;
    idx->dev_stale = true;
    return HNSWGPU_OK;
}

int hnswgpu_insert(hnswgpu_index* idx, const float* data, uint64_t n, uint64_t d, const uint64_t* ids, int nthreads) {
    CAPI_GUARD_BEGIN
    if (!idx || (n && !data)) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::unique_lock<std::shared_mutex> g(idx->mu);
    return insert_points(idx, data, n, d, ids, nthreads);
    CAPI_GUARD_END(HNSWGPU_ERR_ARG)
}

int hnswgpu_insert_gpu(hnswgpu_index* idx, const float* data, uint64_t n, uint64_t d, const uint64_t* ids, int nthreads,
                       int gpu_device, uint64_t gpu_window) {
    CAPI_GUARD_BEGIN
    if (!idx || (n && !data)) return fail(HNSWGPU_ERR_ARG, "null argument");
    if (gpu_device < 0) return fail(HNSWGPU_ERR_ARG, "gpu_device must name a HIP device");
    std::unique_lock<std::shared_mutex> g(idx->mu);
    return insert_points(idx, data, n, d, ids, nthreads, gpu_device, gpu_window);
    CAPI_GUARD_END(HNSWGPU_ERR_ARG)
}

uint64_t hnswgpu_nb_point(const hnswgpu_index* cidx) {
    CAPI_GUARD_BEGIN
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return 0;
    std::shared_lock<std::shared_mutex> g(idx->mu);
    if (idx->builder) return idx->builder->nb_point();
    return idx->flat ? idx->flat->n : 0;
    CAPI_GUARD_END(0)
}
uint64_t hnswgpu_dimension(const hnswgpu_index* cidx) {
    CAPI_GUARD_BEGIN
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return 0;
    std::shared_lock<std::shared_mutex> g(idx->mu);
    if (idx->builder) return idx->builder->dimension();
    return idx->flat ? idx->flat->dimension : 0;
    CAPI_GUARD_END(0)
}
int hnswgpu_dist(const hnswgpu_index* cidx) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return -1;
    std::shared_lock<std::shared_mutex> g(idx->mu);
    if (idx->builder) return idx->params.dist;
    return idx->flat ? idx->flat->dist : -1;
}
uint64_t hnswgpu_layer_nb_point(const hnswgpu_index* cidx, unsigned layer) {
    CAPI_GUARD_BEGIN
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || layer >= NB_LAYER_MAX) return 0;
    std::unique_lock<std::shared_mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    return f ? f->layer_count(layer) : 0;
    CAPI_GUARD_END(0)
}
int hnswgpu_max_level_observed(const hnswgpu_index* cidx) {
    CAPI_GUARD_BEGIN
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return 0;
    std::unique_lock<std::shared_mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    if (!f || f->entry_flat == NO_POINT) return 0;
    return (int)f->layer_of(f->entry_flat);
    CAPI_GUARD_END(0)
}
int hnswgpu_entry_point(const hnswgpu_index* cidx, uint64_t* origin_id, uint8_t* layer, int32_t* rank) {
    CAPI_GUARD_BEGIN
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::unique_lock<std::shared_mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    if (!f || f->entry_flat == NO_POINT) return fail(HNSWGPU_ERR_EMPTY, "index is empty");
    if (origin_id) *origin_id = f->origin_id[f->entry_flat];
    if (layer) *layer = (uint8_t)f->layer_of(f->entry_flat);
    if (rank) *rank = f->rank_of(f->entry_flat);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_ARG)
}
int64_t hnswgpu_neighbours(const hnswgpu_index* cidx, unsigned layer, int32_t rank, unsigned l, uint64_t cap,
                           uint64_t* origin_ids, uint8_t* layers, int32_t* ranks, float* dists) {
    CAPI_GUARD_BEGIN
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || layer >= NB_LAYER_MAX || l >= NB_LAYER_MAX || rank < 0) { fail(HNSWGPU_ERR_ARG, "bad argument"); return -1; }
    std::unique_lock<std::shared_mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    if (!f || (uint64_t)rank >= f->layer_count(layer)) { fail(HNSWGPU_ERR_ARG, "no such point"); return -1; }
    uint64_t flat = f->layer_offset[layer] + (uint64_t)rank;
    uint64_t b = f->nbr_ptr[flat * NB_LAYER_MAX + l], e = f->nbr_ptr[flat * NB_LAYER_MAX + l + 1];
    for (uint64_t j = b; j < e && j - b < cap; ++j) {
        uint32_t nf = f->nbr_flat[j];
        if (origin_ids) origin_ids[j - b] = f->origin_id[nf];
        if (layers) layers[j - b] = (uint8_t)f->layer_of(nf);
        if (ranks) ranks[j - b] = f->rank_of(nf);
        if (dists) dists[j - b] = f->nbr_dist[j];
    }
    return (int64_t)(e - b);
    CAPI_GUARD_END(-1)
}

int hnswgpu_device_count(void) { return device_count(); }

int hnswgpu_upload(hnswgpu_index* idx, int device) {
    CAPI_GUARD_BEGIN
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::unique_lock<std::shared_mutex> g(idx->mu);
    int rc = ensure_device(idx, device);
    if (rc == HNSWGPU_OK && device >= 0) idx->primary = device;  // the last explicit upload names the primary device
    return rc;
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}

// shared implementation of the host-buffer searches on the primary replica
static int search_host_common(const hnswgpu_index* cidx, const float* queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef,
                              const uint64_t* allowed, uint64_t n_allowed, bool filtered, uint64_t* out_ids, float* out_dists,
                              uint8_t* out_layer, int32_t* out_rank, uint32_t* out_counts, uint8_t* out_status, uint32_t* panics) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::shared_lock<std::shared_mutex> sl(idx->mu);
    const bool empty = idx->builder ? idx->builder->nb_point() == 0 : (!idx->flat || idx->flat->n == 0);
    if (empty) {  // empty index => every answer is empty (src/hnsw.rs:1498-1503)
        if (out_counts) std::memset(out_counts, 0, nq * sizeof(uint32_t));
        if (out_status) std::memset(out_status, 0, nq);
        if (panics) *panics = 0;
        return HNSWGPU_OK;
    }
    DeviceIndex* dev = nullptr;
    int rc = primary_replica(idx, sl, &dev);
    if (rc != HNSWGPU_OK) return rc;
    std::string err;
    CallInfo info;
    rc = dev->search_host(queries, nq, d, k, ef, out_ids, out_dists, out_layer, out_rank, out_counts, allowed, n_allowed, filtered,
                          out_status, &info, err);
    if (rc != OK) return fail(rc, err);
    if (panics) *panics = info.panics;
    return HNSWGPU_OK;
}

int hnswgpu_search_batch(const hnswgpu_index* cidx, const float* queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef,
                         uint64_t* out_ids, float* out_dists, uint8_t* out_layer, int32_t* out_rank, uint32_t* out_counts) {
    CAPI_GUARD_BEGIN
    return search_host_common(cidx, queries, nq, d, k, ef, nullptr, 0, false, out_ids, out_dists, out_layer, out_rank, out_counts,
                              nullptr, nullptr);
    CAPI_GUARD_END(HNSWGPU_ERR_ARG)
}

int hnswgpu_search_batch_filtered(const hnswgpu_index* cidx, const float* queries, uint64_t nq, uint64_t d, uint64_t k,
                                  uint64_t ef, const uint64_t* allowed_ids, uint64_t n_allowed, uint64_t* out_ids,
                                  float* out_dists, uint8_t* out_layer, int32_t* out_rank, uint32_t* out_counts,
                                  uint8_t* out_status) {
    CAPI_GUARD_BEGIN
    if (n_allowed && !allowed_ids) return fail(HNSWGPU_ERR_ARG, "null filter");
    for (uint64_t i = 1; i < n_allowed; ++i)  // `impl FilterT for Vec<usize>` is a binary search: the vector must be sorted
        if (allowed_ids[i - 1] > allowed_ids[i]) return fail(HNSWGPU_ERR_ARG, "the id vector of a filter must be sorted ascending");
    uint32_t panics = 0;
    int rc = search_host_common(cidx, queries, nq, d, k, ef, allowed_ids, n_allowed, true, out_ids, out_dists, out_layer, out_rank,
                                out_counts, out_status, &panics);
    if (rc != HNSWGPU_OK) return rc;
    if (panics != 0 && !out_status)
        return fail(HNSWGPU_ERR_REF_PANIC, "the reference panics on " + std::to_string(panics) +
                    " of these queries (return_points.peek().unwrap() on a heap the filter emptied, src/hnsw.rs:973); "
                    "pass out_status to learn which");
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_ARG)
}

// Replicas on every device of devices[0..n): the missing ones are uploaded AT THE SAME TIME, one host thread per device
// (an upload is a host-side re-layout plus ~0.65 GB over PCIe for BASELINE config 2: eight of them one after the other
// was the first-call cost of an 8-GPU host).  Exclusive lock held.
static int ensure_devices(hnswgpu_index* idx, const int* devices, int n) {
    const FlatIndex* f = idx->get_flat();
    if (!f || f->n == 0) return fail(HNSWGPU_ERR_EMPTY, "index is empty");
    if (idx->dev_stale) {  // the graph changed: every replica is out of date
        idx->replicas.clear();
        idx->dev_stale = false;
    }
    std::vector<int> missing;
    for (int s = 0; s < n; ++s)
        if (!idx->replica(devices[s]) && std::find(missing.begin(), missing.end(), devices[s]) == missing.end()) missing.push_back(devices[s]);
    if (!missing.empty()) {
        std::vector<std::unique_ptr<DeviceIndex>> fresh(missing.size());
        std::vector<int> rcs(missing.size(), OK);
        std::vector<std::string> errs(missing.size());
        auto upload_one = [&](size_t i) {
            try {
                fresh[i].reset(new DeviceIndex());
                rcs[i] = fresh[i]->upload(*f, missing[i], errs[i]);
            } catch (const std::exception& e) {
                rcs[i] = ERR_DEVICE;
                errs[i] = e.what();
            }
        };
        {
            JoinAll th;
            for (size_t i = 1; i < missing.size(); ++i) th.spawn([&, i]() { upload_one(i); });
            upload_one(0);
        }
        for (size_t i = 0; i < missing.size(); ++i)
            if (rcs[i] != OK) return fail(rcs[i], "device " + std::to_string(missing[i]) + ": " + errs[i]);
        for (size_t i = 0; i < missing.size(); ++i) {
            if (idx->strict_ties >= 0) fresh[i]->set_strict_ties(idx->strict_ties != 0);
            fresh[i]->set_arithmetic(idx->arithmetic);
            idx->replicas[missing[i]] = std::move(fresh[i]);
        }
    }
    if (n > 0 && (idx->primary < 0 || !idx->replica(idx->primary))) idx->primary = devices[0];
    return HNSWGPU_OK;
}

// shared front of the sharded entry points: argument checks, replicas, then the shared lock for the searches.
// Returns HNSWGPU_OK with *empty = true when the index holds no point (the answer is "no neighbours").
static int sharded_prepare(hnswgpu_index* idx, const int* devices, int n_shards, std::shared_lock<std::shared_mutex>& sl,
                           std::vector<DeviceIndex*>& reps, bool* empty) {
    *empty = false;
    {   // replicas on every device named (exclusive: uploads change the handle)
        std::unique_lock<std::shared_mutex> xl(idx->mu);
        if (idx->builder ? idx->builder->nb_point() == 0 : (!idx->flat || idx->flat->n == 0)) { *empty = true; return HNSWGPU_OK; }
        int rc = ensure_devices(idx, devices, n_shards);
        if (rc != HNSWGPU_OK) return rc;
    }
    sl = std::shared_lock<std::shared_mutex>(idx->mu);
    if (!idx->fresh() || idx->dev_stale) return fail(HNSWGPU_ERR_DEVICE, "index changed while a search was starting");
    reps.assign((size_t)n_shards, nullptr);
    for (int s = 0; s < n_shards; ++s) {  // every replica resolved BEFORE the first thread exists
        reps[(size_t)s] = idx->replica(devices[s]);
        if (!reps[(size_t)s]) return fail(HNSWGPU_ERR_DEVICE, "replica missing");
    }
    return HNSWGPU_OK;
}

// Hnsw::parallel_search with the batch sharded over several GPUs of this process: graph replicated (one replica per
// distinct device, uploaded on first use, all at once), contiguous balanced blocks of queries, one host thread per shard,
// every shard copies its answers straight into the caller's arrays -- the gather.  No collective: the shards are independent.
int hnswgpu_search_batch_sharded(const hnswgpu_index* cidx, const int* devices, int n_shards, const float* queries, uint64_t nq,
                                 uint64_t d, uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dists, uint8_t* out_layer,
                                 int32_t* out_rank, uint32_t* out_counts) {
    CAPI_GUARD_BEGIN
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || !devices || n_shards <= 0) return fail(HNSWGPU_ERR_ARG, "bad argument");
    if (nq && (!queries || !out_ids || !out_dists || !out_counts)) return fail(HNSWGPU_ERR_ARG, "null buffer");
    std::shared_lock<std::shared_mutex> sl;
    std::vector<DeviceIndex*> reps;
    bool empty = false;
    int prc = sharded_prepare(idx, devices, n_shards, sl, reps, &empty);
    if (prc != HNSWGPU_OK) return prc;
    if (empty) {
        if (out_counts) std::memset(out_counts, 0, nq * sizeof(uint32_t));
        return HNSWGPU_OK;
    }
    std::vector<int> rcs((size_t)n_shards, OK);
    std::vector<std::string> errs((size_t)n_shards);
    {
        JoinAll th;  // joins whatever was spawned, also when a spawn throws
        const uint64_t base = nq / (uint64_t)n_shards, rem = nq % (uint64_t)n_shards;  // shard s: base + (s < rem) queries
        uint64_t start = 0;
        for (int s = 0; s < n_shards; ++s) {
            const uint64_t cnt = base + ((uint64_t)s < rem ? 1 : 0);
            DeviceIndex* dev = reps[(size_t)s];
            const uint64_t s0 = start;
            start += cnt;
            if (cnt == 0) continue;
            th.spawn([=, &rcs, &errs]() {
                try {
                    rcs[(size_t)s] = dev->search_host(queries + s0 * d, cnt, d, k, ef, out_ids + s0 * k, out_dists + s0 * k,
                                                      out_layer ? out_layer + s0 * k : nullptr, out_rank ? out_rank + s0 * k : nullptr,
                                                      out_counts + s0, nullptr, 0, false, nullptr, nullptr, errs[(size_t)s]);
                } catch (const std::exception& e) {
                    rcs[(size_t)s] = ERR_DEVICE;
                    errs[(size_t)s] = e.what();
                }
            });
        }
    }
    for (int s = 0; s < n_shards; ++s)
        if (rcs[(size_t)s] != OK) return fail(rcs[(size_t)s], "shard " + std::to_string(s) + ": " + errs[(size_t)s]);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}

int hnswgpu_search_batch_sharded_device(const hnswgpu_index* cidx, const int* devices, int n_shards, const float* const* d_queries,
                                        const uint64_t* nq_shard, uint64_t d, uint64_t k, uint64_t ef, uint64_t* const* d_out_ids,
                                        float* const* d_out_dists, uint8_t* const* d_out_layer, int32_t* const* d_out_rank,
                                        uint32_t* const* d_out_counts, void* const* streams) {
    CAPI_GUARD_BEGIN
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || !devices || n_shards <= 0 || !nq_shard || !d_queries || !d_out_ids || !d_out_dists || !d_out_counts)
        return fail(HNSWGPU_ERR_ARG, "bad argument");
    for (int s = 0; s < n_shards; ++s)
        if (nq_shard[s] && (!d_queries[s] || !d_out_ids[s] || !d_out_dists[s] || !d_out_counts[s])) return fail(HNSWGPU_ERR_ARG, "null buffer");
    std::shared_lock<std::shared_mutex> sl;
    std::vector<DeviceIndex*> reps;
    bool empty = false;
    int prc = sharded_prepare(idx, devices, n_shards, sl, reps, &empty);
    if (prc != HNSWGPU_OK) return prc;
    if (empty) return fail(HNSWGPU_ERR_EMPTY, "index is empty");  // (device-resident counters cannot be zeroed from here without a device)
    std::vector<int> rcs((size_t)n_shards, OK);
    std::vector<std::string> errs((size_t)n_shards);
    {
        JoinAll th;
        for (int s = 0; s < n_shards; ++s) {
            if (nq_shard[s] == 0) continue;
            DeviceIndex* dev = reps[(size_t)s];
            th.spawn([=, &rcs, &errs]() {
                try {
                    rcs[(size_t)s] = dev->search_device(d_queries[s], nq_shard[s], d, k, ef, d_out_ids[s], d_out_dists[s],
                                                        d_out_layer ? d_out_layer[s] : nullptr, d_out_rank ? d_out_rank[s] : nullptr,
                                                        d_out_counts[s], nullptr, streams ? streams[s] : nullptr, nullptr, 0, nullptr,
                                                        errs[(size_t)s]);
                } catch (const std::exception& e) {
                    rcs[(size_t)s] = ERR_DEVICE;
                    errs[(size_t)s] = e.what();
                }
            });
        }
    }
    for (int s = 0; s < n_shards; ++s)
        if (rcs[(size_t)s] != OK) return fail(rcs[(size_t)s], "shard " + std::to_string(s) + ": " + errs[(size_t)s]);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}

int hnswgpu_lane_lab(int device, uint32_t mode, uint32_t p0, uint32_t p1, uint32_t p2, const uint32_t* ops, uint32_t n_ops,
                     const uint32_t* lanes, uint32_t n_lane_sets, uint32_t* out, uint32_t out_words) {
    CAPI_GUARD_BEGIN
    if ((n_ops && !ops) || (n_lane_sets && !lanes) || !out) return fail(HNSWGPU_ERR_ARG, "null buffer");
    std::string err;
    int rc = lane_lab_device(device, mode, p0, p1, p2, ops, n_ops, lanes, n_lane_sets, out, out_words, err);
    if (rc != OK) return fail(rc, err);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}

int hnswgpu_gather_sharded_answers(const int* devices, int n_shards, const uint64_t* nq_shard, uint64_t k,
                                   const uint64_t* const* d_ids, const float* const* d_dists, const uint8_t* const* d_layer,
                                   const int32_t* const* d_rank, const uint32_t* const* d_counts, int root_device,
                                   uint64_t* root_ids, float* root_dists, uint8_t* root_layer, int32_t* root_rank,
                                   uint32_t* root_counts, void* root_stream) {
    CAPI_GUARD_BEGIN
    if (!devices || n_shards <= 0 || !nq_shard || !d_ids || !d_dists || !d_counts || k == 0) return fail(HNSWGPU_ERR_ARG, "bad argument");
    uint64_t total = 0;
    for (int s = 0; s < n_shards; ++s) {
        if (nq_shard[s] && (!d_ids[s] || !d_dists[s] || !d_counts[s])) return fail(HNSWGPU_ERR_ARG, "null buffer");
        total += nq_shard[s];
    }
    if (total && (!root_ids || !root_dists || !root_counts)) return fail(HNSWGPU_ERR_ARG, "null buffer");
    std::string err;
    int rc = hnswgpu::gather_sharded_answers(devices, n_shards, nq_shard, k, d_ids, d_dists, d_layer, d_rank, d_counts, root_device,
                                             root_ids, root_dists, root_layer, root_rank, root_counts, root_stream, err);
    if (rc != OK) return fail(rc, err);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}

static int search_device_common(const hnswgpu_index* cidx, const float* d_queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef,
                                const uint64_t* d_allowed, uint64_t n_allowed, uint64_t* d_out_ids, float* d_out_dists,
                                uint8_t* d_out_layer, int32_t* d_out_rank, uint32_t* d_out_counts, uint32_t* d_stats, void* stream,
                                uint32_t* panics) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::shared_lock<std::shared_mutex> sl(idx->mu);
    DeviceIndex* dev = idx->primary >= 0 ? idx->replica(idx->primary) : nullptr;
    if (!dev || idx->dev_stale || idx->flat_stale)
        return fail(HNSWGPU_ERR_DEVICE, "index is not resident on a device: call hnswgpu_upload first");
    std::string err;
    CallInfo info;
    int rc = dev->search_device(d_queries, nq, d, k, ef, d_out_ids, d_out_dists, d_out_layer, d_out_rank, d_out_counts, d_stats,
                                stream, d_allowed, n_allowed, &info, err);
    if (rc != OK) return fail(rc, err);
    if (panics) *panics = info.panics;
    return HNSWGPU_OK;
}

int hnswgpu_search_batch_device(const hnswgpu_index* cidx, const float* d_queries, uint64_t nq, uint64_t d, uint64_t k,
                                uint64_t ef, uint64_t* d_out_ids, float* d_out_dists, uint8_t* d_out_layer,
                                int32_t* d_out_rank, uint32_t* d_out_counts, uint32_t* d_stats, void* stream) {
    CAPI_GUARD_BEGIN
    return search_device_common(cidx, d_queries, nq, d, k, ef, nullptr, 0, d_out_ids, d_out_dists, d_out_layer, d_out_rank,
                                d_out_counts, d_stats, stream, nullptr);
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}

// begin / end: the blocking call as an asynchronous job of the library's worker pool (a long-lived thread: creating one per
// ticket cost tens of microseconds on a ~1 ms call), so the launch path is the one every other entry point uses; searches on
// one handle may run concurrently, each with a private workspace
struct hnswgpu_ticket {
    std::shared_ptr<WorkerPool::Job> job;
    int status = HNSWGPU_OK;
    std::string error;
};
int hnswgpu_search_batch_device_begin(const hnswgpu_index* cidx, const float* d_queries, uint64_t nq, uint64_t d, uint64_t k,
                                      uint64_t ef, uint64_t* d_out_ids, float* d_out_dists, uint8_t* d_out_layer,
                                      int32_t* d_out_rank, uint32_t* d_out_counts, uint32_t* d_stats, void* stream,
                                      hnswgpu_ticket** ticket) {
    CAPI_GUARD_BEGIN
    if (!ticket) return fail(HNSWGPU_ERR_ARG, "null ticket");
    *ticket = nullptr;
    if (!cidx) return fail(HNSWGPU_ERR_ARG, "null index");
    std::unique_ptr<hnswgpu_ticket> t(new hnswgpu_ticket());
    hnswgpu_ticket* raw = t.get();
    raw->job = WorkerPool::instance().submit([=]() {
        raw->status = hnswgpu_search_batch_device(cidx, d_queries, nq, d, k, ef, d_out_ids, d_out_dists, d_out_layer, d_out_rank,
                                                  d_out_counts, d_stats, stream);
        if (raw->status != HNSWGPU_OK) raw->error = hnswgpu_last_error();  // this thread's message, handed to the ticket
    });
    *ticket = t.release();
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}
int hnswgpu_search_batch_end(hnswgpu_ticket* ticket) {
    CAPI_GUARD_BEGIN
    if (!ticket) return fail(HNSWGPU_ERR_ARG, "null ticket");
    std::unique_ptr<hnswgpu_ticket> t(ticket);
    if (t->job) t->job->wait();
    if (t->status != HNSWGPU_OK) return fail(t->status, t->error);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}

int hnswgpu_search_batch_filtered_device(const hnswgpu_index* cidx, const float* d_queries, uint64_t nq, uint64_t d, uint64_t k,
                                         uint64_t ef, const uint64_t* d_allowed_ids, uint64_t n_allowed, uint64_t* d_out_ids,
                                         float* d_out_dists, uint8_t* d_out_layer, int32_t* d_out_rank, uint32_t* d_out_counts,
                                         uint32_t* d_stats, void* stream, uint32_t* n_panics) {
    CAPI_GUARD_BEGIN
    if (!d_allowed_ids && n_allowed) return fail(HNSWGPU_ERR_ARG, "null filter");
    uint32_t panics = 0;
    // an empty filter still is a filter (every point is refused); a null pointer would read as "no filter" below, so
    // hand over a non-null pointer that is never dereferenced (n_allowed == 0)
    const uint64_t* ids = d_allowed_ids ? d_allowed_ids : reinterpret_cast<const uint64_t*>(d_queries);
    int rc = search_device_common(cidx, d_queries, nq, d, k, ef, ids, n_allowed, d_out_ids, d_out_dists, d_out_layer, d_out_rank,
                                  d_out_counts, d_stats, stream, &panics);
    if (n_panics) *n_panics = panics;
    return rc;
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}

static DeviceIndex* any_replica(hnswgpu_index* idx) {
    if (idx->primary >= 0 && idx->replica(idx->primary)) return idx->replica(idx->primary);
    return nullptr;
}
int hnswgpu_last_kernel_ms(const hnswgpu_index* cidx, double* ms, uint32_t* launches) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::shared_lock<std::shared_mutex> g(idx->mu);
    DeviceIndex* dev = any_replica(idx);
    if (!dev) return fail(HNSWGPU_ERR_DEVICE, "index is not resident on a device");
    const CallInfo c = dev->last_call();
    if (ms) *ms = c.ms;
    if (launches) *launches = c.launches;
    return HNSWGPU_OK;
}
int hnswgpu_last_search_kernel_ms(const hnswgpu_index* cidx, double* ms) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || !ms) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::shared_lock<std::shared_mutex> g(idx->mu);
    DeviceIndex* dev = any_replica(idx);
    if (!dev) return fail(HNSWGPU_ERR_DEVICE, "index is not resident on a device");
    *ms = dev->last_call().main_ms;
    return HNSWGPU_OK;
}
int hnswgpu_set_arithmetic(hnswgpu_index* idx, int arithmetic) {
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    if (arithmetic != HNSWGPU_ARITH_SCALAR && arithmetic != HNSWGPU_ARITH_SIMD8) return fail(HNSWGPU_ERR_ARG, "unknown arithmetic");
    std::unique_lock<std::shared_mutex> g(idx->mu);
    idx->arithmetic = arithmetic;
    for (auto& kv : idx->replicas) kv.second->set_arithmetic(arithmetic);
    return HNSWGPU_OK;
}
int hnswgpu_set_strict_ties(hnswgpu_index* idx, int on) {
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::unique_lock<std::shared_mutex> g(idx->mu);
    idx->strict_ties = on != 0;
    for (auto& kv : idx->replicas) kv.second->set_strict_ties(idx->strict_ties != 0);
    return HNSWGPU_OK;
}
int hnswgpu_reload_env(void) {
    hnswgpu::reload_knobs();
    return HNSWGPU_OK;
}
int hnswgpu_last_tie_count(const hnswgpu_index* cidx, uint32_t* ties) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || !ties) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::shared_lock<std::shared_mutex> g(idx->mu);
    DeviceIndex* dev = any_replica(idx);
    if (!dev) return fail(HNSWGPU_ERR_DEVICE, "index is not resident on a device");
    *ties = dev->last_call().ties;
    return HNSWGPU_OK;
}

int hnswgpu_eval_distances(int dist, const float* a, const float* b, uint64_t n, uint64_t d, float* out) {
    CAPI_GUARD_BEGIN
    if (!a || !b || !out || dist < 0 || dist >= DIST_COUNT) return fail(HNSWGPU_ERR_ARG, "bad argument");
    std::string err;
    int rc = eval_distance_matrix_device(dist, a, n, b, n, d, 1, true, out, err);
    if (rc != OK) return fail(rc, err);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}
int hnswgpu_eval_distance_matrix(int dist, const float* queries, uint64_t nq, const float* rows, uint64_t n, uint64_t d,
                                 uint32_t batch, float* out) {
    CAPI_GUARD_BEGIN
    if (!queries || !rows || !out || dist < 0 || dist >= DIST_COUNT) return fail(HNSWGPU_ERR_ARG, "bad argument");
    std::string err;
    int rc = eval_distance_matrix_device(dist, queries, nq, rows, n, d, batch, false, out, err);
    if (rc != OK) return fail(rc, err);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}

int hnswgpu_eval_distance_matrix_arith(int dist, int arithmetic, const float* queries, uint64_t nq, const float* rows, uint64_t n,
                                       uint64_t d, uint32_t batch, float* out) {
    CAPI_GUARD_BEGIN
    if (!queries || !rows || !out || dist < 0 || dist >= DIST_COUNT) return fail(HNSWGPU_ERR_ARG, "bad argument");
    std::string err;
    int rc = eval_distance_matrix_device(dist, queries, nq, rows, n, d, batch, false, out, err, arithmetic);
    if (rc != OK) return fail(rc, err);
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_DEVICE)
}

// ---- DataMap (src/datamap.rs): the vectors of a dump by DataId, memory-mapped, without loading the graph
struct hnswgpu_datamap {
    DataMap m;
};
int hnswgpu_datamap_open(const char* dir, const char* basename, hnswgpu_datamap** out) {
    CAPI_GUARD_BEGIN
    if (!dir || !basename || !out) return fail(HNSWGPU_ERR_ARG, "null argument");
    *out = nullptr;
    std::unique_ptr<hnswgpu_datamap> h(new hnswgpu_datamap());
    std::string err;
    int rc = h->m.open(dir, basename, err);
    if (rc != OK) return fail(rc, err);
    *out = h.release();
    return HNSWGPU_OK;
    CAPI_GUARD_END(HNSWGPU_ERR_IO)
}
void hnswgpu_datamap_close(hnswgpu_datamap* m) { delete m; }
const float* hnswgpu_datamap_get_data(const hnswgpu_datamap* m, uint64_t data_id) { return m ? m->m.get_data(data_id) : nullptr; }
uint64_t hnswgpu_datamap_nb_data(const hnswgpu_datamap* m) { return m ? m->m.nb_data() : 0; }
uint64_t hnswgpu_datamap_dimension(const hnswgpu_datamap* m) { return m ? m->m.dimension() : 0; }
const char* hnswgpu_datamap_distname(const hnswgpu_datamap* m) { return m ? m->m.distname().c_str() : ""; }
const char* hnswgpu_datamap_typename(const hnswgpu_datamap* m) { return m ? m->m.type_name().c_str() : ""; }
uint64_t hnswgpu_datamap_ids(const hnswgpu_datamap* m, uint64_t* out, uint64_t cap) {
    if (!m) return 0;
    const auto& ids = m->m.ids_in_file_order();
    for (uint64_t i = 0; i < ids.size() && i < cap && out; ++i) out[i] = ids[i];
    return ids.size();
}

// =========================================================================================
// Reference-compatible f32 symbols (src/libext.rs)
// =========================================================================================
struct HnswIo {
    std::string dir;
    std::string basename;
};
struct HnswApif32 {
    hnswgpu_index* idx = nullptr;
};

const HnswIo* get_hnswio(uint64_t flen, const uint8_t* name) {  // directory is always "." (src/libext.rs:31)
    CAPI_GUARD_BEGIN
    if (!name) return nullptr;
    HnswIo* io = new HnswIo();
    io->dir = ".";
    io->basename.assign(reinterpret_cast<const char*>(name), (size_t)flen);
    return io;
    CAPI_GUARD_END(nullptr)
}
void hnswgpu_free_hnswio(const HnswIo* p) { delete p; }

static const HnswApif32* load_with(HnswIo* io, int dist) {
    CAPI_GUARD_BEGIN
    if (!io) return nullptr;
    hnswgpu_index* idx = nullptr;
    if (hnswgpu_load_dump(io->dir.c_str(), io->basename.c_str(), dist, &idx) != HNSWGPU_OK) return nullptr;  // null on failure (:298-301)
    HnswApif32* api = new HnswApif32();
    api->idx = idx;
    return api;
    CAPI_GUARD_END(nullptr)
}
const HnswApif32* load_hnswdump_f32_DistL1(HnswIo* io) { return load_with(io, HNSWGPU_DIST_L1); }
const HnswApif32* load_hnswdump_f32_DistL2(HnswIo* io) { return load_with(io, HNSWGPU_DIST_L2); }
const HnswApif32* load_hnswdump_f32_DistCosine(HnswIo* io) { return load_with(io, HNSWGPU_DIST_COSINE); }
const HnswApif32* load_hnswdump_f32_DistDot(HnswIo* io) { return load_with(io, HNSWGPU_DIST_DOT); }
const HnswApif32* load_hnswdump_f32_DistJensenShannon(HnswIo* io) { return load_with(io, HNSWGPU_DIST_JENSENSHANNON); }  // :334-339
const HnswApif32* load_hnswdump_f32_DistJeffreys(HnswIo* io) { return load_with(io, HNSWGPU_DIST_JEFFREYS); }            // :340-345

static const HnswApif32* new_api(size_t max_nb_conn, size_t ef_const, size_t namelen, const uint8_t* cdistname,
                                 size_t max_elements, size_t max_layer, bool allow_cosine) {
    CAPI_GUARD_BEGIN
    (void)max_elements;
    if (!cdistname) return nullptr;
    std::string dname(reinterpret_cast<const char*>(cdistname), namelen);
    int dist = dist_from_short_name(dname);
    // init_hnsw_f32 has no "DistCosine" arm in the reference (src/libext.rs:468-523); keep that quirk
    if (dist < 0 || (dist == DIST_COSINE && !allow_cosine)) {
        fail(HNSWGPU_ERR_DISTANCE, "init_hnsw_f32 received unknow distance " + dname);
        return nullptr;
    }
    if (max_nb_conn > 256 || max_nb_conn < 2) {
        fail(HNSWGPU_ERR_ARG, "error max_nb_connection must be less equal than 256");
        return nullptr;
    }
    hnswgpu_index* idx = new hnswgpu_index();
    idx->params.max_nb_connection = max_nb_conn;
    idx->params.ef_construction = ef_const;
    idx->params.max_layer = max_layer;
    idx->params.dist = dist;
    idx->params.nthreads = 0;
    idx->builder.reset(new GraphBuilder(idx->params));
    HnswApif32* api = new HnswApif32();
    api->idx = idx;
    return api;
    CAPI_GUARD_END(nullptr)
}
const HnswApif32* init_hnsw_f32(size_t max_nb_conn, size_t ef_const, size_t namelen, const uint8_t* cdistname) {
    return new_api(max_nb_conn, ef_const, namelen, cdistname, 10000, 16, false);  // Hnsw::new(M, 10000, 16, ef_c, D) (:475)
}
const HnswApif32* new_hnsw_f32(size_t max_nb_conn, size_t ef_const, size_t namelen, const uint8_t* cdistname,
                               size_t max_elements, size_t max_layer) {
    return new_api(max_nb_conn, ef_const, namelen, cdistname, max_elements, max_layer, false);
}
const HnswApif32* init_hnsw_ptrdist_f32(size_t, size_t, hnsw_dist_fn_f32) {
    // src/libext.rs:643-655 builds Hnsw<f32, DistCFFI<f32>> around a host function pointer: nothing the device can call
    fail(HNSWGPU_ERR_DISTANCE, "init_hnsw_ptrdist_f32: a host distance callback cannot run on the device and there is no CPU "
                               "search path; use a named distance (DistL2, DistL1, DistCosine, DistDot, DistHellinger, "
                               "DistJeffreys, DistJensenShannon)");
    return nullptr;
}

// The reference's insert_f32 / parallel_insert_f32 return nothing (:661-723).  They work on every handle, reloaded ones
// included; a failure (dimension mismatch, no device, bad ordinal, ...) leaves the index unchanged and is readable through
// hnswgpu_last_error().
void insert_f32(HnswApif32* api, size_t len, const float* data, size_t id) {
    CAPI_GUARD_BEGIN
    if (!api || !api->idx || !data) { fail(HNSWGPU_ERR_ARG, "insert_f32: null argument"); return; }
    hnswgpu_index* idx = api->idx;
    std::unique_lock<std::shared_mutex> g(idx->mu);
    uint64_t id64 = id;
    (void)insert_points(idx, data, 1, len, &id64, 1);
    CAPI_GUARD_END()
}
void parallel_insert_f32(HnswApif32* api, size_t nb_vec, size_t vec_len, const float** datas, const size_t* ids) {
    CAPI_GUARD_BEGIN
    if (!api || !api->idx || !datas || !ids) { fail(HNSWGPU_ERR_ARG, "parallel_insert_f32: null argument"); return; }
    hnswgpu_index* idx = api->idx;
    std::unique_lock<std::shared_mutex> g(idx->mu);
    std::vector<float> flat(nb_vec * vec_len);  // inputs are copied, like the reference (:700-712)
    std::vector<uint64_t> id64(nb_vec);
    for (size_t i = 0; i < nb_vec; ++i) {
        std::memcpy(flat.data() + i * vec_len, datas[i], vec_len * sizeof(float));
        id64[i] = ids[i];
    }
    (void)insert_points(idx, flat.data(), nb_vec, vec_len, id64.data(), 0);
    CAPI_GUARD_END()
}

static Neighbour_api* make_row(const uint64_t* ids, const float* dists, uint32_t cnt) {
    Neighbour_api* row = cnt ? static_cast<Neighbour_api*>(std::malloc(cnt * sizeof(Neighbour_api))) : nullptr;
    for (uint32_t j = 0; j < cnt; ++j) {
        row[j].id = (size_t)ids[j];
        row[j].d = dists[j];
    }
    return row;
}

const Neighbourhood_api* search_neighbours_f32(const HnswApif32* api, size_t len, const float* data, size_t knbn,
                                               size_t ef_search) {
    CAPI_GUARD_BEGIN
    if (!api || !api->idx || !data || knbn == 0) return nullptr;
    std::vector<uint64_t> ids(knbn);
    std::vector<float> dists(knbn);
    uint32_t cnt = 0;
    if (hnswgpu_search_batch(api->idx, data, 1, len, knbn, ef_search, ids.data(), dists.data(), nullptr, nullptr, &cnt) != HNSWGPU_OK)
        return nullptr;
    Neighbourhood_api* ans = static_cast<Neighbourhood_api*>(std::malloc(sizeof(Neighbourhood_api)));
    ans->nbgh = cnt;
    ans->neighbours = make_row(ids.data(), dists.data(), cnt);
    return ans;
    CAPI_GUARD_END(nullptr)
}

// The answer of parallel_search_neighbours_f32 is ONE allocation: a header, the Vec_api, its nb_vec Neighbourhood_api and every
// Neighbour_api row behind them (the reference leaks 1 + nb_vec vectors per call, src/libext.rs:236-253; a caller that frees
// hands the Vec_api pointer to hnswgpu_free_neighbourhood_vec, which finds the header in front of it).
namespace {
constexpr uint64_t SLAB_MAGIC = 0x486E7377536C6162ull;  // "HnswSlab"
// A slab is ordinary memory, or -- the usual case -- page-locked memory the device addresses (hnswgpu::pinned_alloc): the search
// kernels then write ids, distances and counts straight into the Neighbour_api / Neighbourhood_api records and the call has
// no unpacking pass.  The header remembers which, and the shape (nq, k) whose row pointers the records hold.
struct SlabHeader {
    uint64_t magic;
    uint64_t bytes;
    uint64_t dev;      // page-locked: the address of this header as the allocating device sees it (nonzero); 0: ordinary memory
    uint64_t nq, k;    // the shape the Neighbourhood_api records were last laid out for (0, 0: not yet)
    uint64_t reserved;
};
static_assert(sizeof(SlabHeader) % 16 == 0, "the records behind the header stay 16-byte aligned");
struct FfiAnswer {
    size_t nq, k;
    Vec_api_Neighbourhood* out;
    Neighbourhood_api* lists;
    Neighbour_api* rows;
    SlabHeader* slab;
};
// A caller that frees its answers (hnswgpu_free_neighbourhood_vec) and asks again gets the same memory back: a 10 000 x 10
// answer is a 1.9 MB allocation, which malloc serves with mmap / munmap and the kernel with ~470 fresh page faults per call --
// a tenth of a millisecond on a 1.2 ms search -- and page-locking costs more than that.  A few freed slabs (at most 64 MB) are
// kept for the next call of the same size; at most 256 MB of page-locked slabs are out at any time (answers a caller never
// frees -- the reference leaks them, src/libext.rs:236-253 -- are then served from ordinary memory, with an unpacking pass).
class SlabCache {
public:
    SlabHeader* take(size_t bytes, bool pinned_ok) {
        {
            std::lock_guard<std::mutex> g(mu_);
            for (size_t i = 0; i < n_; ++i)
                if (slab_[i]->bytes == bytes && (pinned_ok || slab_[i]->dev == 0)) {
                    SlabHeader* h = slab_[i];
                    slab_[i] = slab_[--n_];
                    held_ -= bytes;
                    h->magic = SLAB_MAGIC;
                    return h;
                }
        }
        SlabHeader* h = nullptr;
        if (pinned_ok) {
            if (pinned_out_.fetch_add(bytes) + bytes <= PINNED_LIMIT) {
                void* dev = nullptr;
                h = static_cast<SlabHeader*>(hnswgpu::pinned_alloc(bytes, &dev));
                if (h) {
                    *h = SlabHeader{SLAB_MAGIC, bytes, (uint64_t)(uintptr_t)dev, 0, 0, 0};
                    return h;
                }
            }
            pinned_out_.fetch_sub(bytes);  // refused by the runtime, or over the limit: ordinary memory
        }
        h = static_cast<SlabHeader*>(std::malloc(bytes));
        if (h) *h = SlabHeader{SLAB_MAGIC, bytes, 0, 0, 0, 0};
        return h;
    }
    void give(SlabHeader* h) {
        const size_t bytes = (size_t)h->bytes;
        h->magic = 0;  // (a second free of the same answer is then at least not taken for a slab)
        {
            std::lock_guard<std::mutex> g(mu_);
            if (n_ < KEEP && held_ + bytes <= (64ull << 20)) {
                slab_[n_++] = h;
                held_ += bytes;
                return;
            }
        }
        if (h->dev != 0) {
            hnswgpu::pinned_free(h);
            pinned_out_.fetch_sub(bytes);
        } else {
            std::free(h);
        }
    }
private:
    static constexpr size_t KEEP = 4;
    static constexpr uint64_t PINNED_LIMIT = 256ull << 20;
    std::mutex mu_;
    SlabHeader* slab_[KEEP] = {};
    size_t n_ = 0, held_ = 0;
    std::atomic<uint64_t> pinned_out_{0};
};
SlabCache& slab_cache() {
    static SlabCache* c = new SlabCache();  // (never destroyed: answers may be freed during static destruction)
    return *c;
}
}  // namespace

const Vec_api_Neighbourhood* parallel_search_neighbours_f32(const HnswApif32* api, size_t nb_vec, int64_t vec_len,
                                                            const float** data, size_t knbn, size_t ef_search) {
    CAPI_GUARD_BEGIN
    if (!api || !api->idx || !data || knbn == 0 || vec_len <= 0) return nullptr;
    hnswgpu_index* idx = api->idx;
    for (size_t i = 0; i < nb_vec; ++i)
        if (!data[i]) { fail(HNSWGPU_ERR_ARG, "parallel_search_neighbours_f32: null row pointer"); return nullptr; }
    FfiAnswer ans{nb_vec, knbn, nullptr, nullptr, nullptr, nullptr};
    // the row pointers are gathered straight into pinned staging memory (the reference copies them into Vec<Vec<f32>>,
    // :218-226); the slab is taken before the search starts.  A page-locked slab is written by the search kernels themselves
    // (sink.direct); an ordinary one is filled out of the pinned answer arena by the threads of the call's pool section
    // (HNSWGPU_FFI_UNPACK=1 forces that path: test hook)
    DeviceIndex::AnswerSink sink{
        [](void* ctx, uint64_t nq, uint64_t k) -> bool {
            FfiAnswer& f = *static_cast<FfiAnswer*>(ctx);
            const size_t bytes = sizeof(SlabHeader) + sizeof(Vec_api_Neighbourhood) + nq * sizeof(Neighbourhood_api) + nq * k * sizeof(Neighbour_api);
            const bool unpack_only = hnswgpu::knobs().ffi_unpack;
            SlabHeader* h = slab_cache().take(bytes, !unpack_only);
            if (!h) return false;
            unsigned char* slab = reinterpret_cast<unsigned char*>(h);
            Vec_api_Neighbourhood* v = reinterpret_cast<Vec_api_Neighbourhood*>(slab + sizeof(SlabHeader));
            f.lists = reinterpret_cast<Neighbourhood_api*>(v + 1);
            f.rows = reinterpret_cast<Neighbour_api*>(f.lists + nq);
            v->len = (int64_t)nq;
            v->ptr = f.lists;
            f.out = v;
            f.slab = h;
            if (h->dev != 0 && (h->nq != nq || h->k != k)) {
                // the records the kernels do not write: every list's row pointer, the high word of its count, the rows' padding
                std::memset(f.lists, 0, bytes - sizeof(SlabHeader) - sizeof(Vec_api_Neighbourhood));
                for (size_t i = 0; i < nq; ++i) f.lists[i].neighbours = f.rows + i * k;
                h->nq = nq;
                h->k = k;
            }
            return true;
        },
        [](void* ctx, const DeviceIndex::HostAnswers& a, uint64_t lo, uint64_t hi) {
            FfiAnswer& f = *static_cast<FfiAnswer*>(ctx);
            for (size_t i = lo; i < hi; ++i) {
                const uint32_t c = a.counts[i];
                Neighbour_api* r = f.rows + i * f.k;
                for (uint32_t j = 0; j < c; ++j) {
                    r[j].id = (size_t)a.ids[i * f.k + j];
                    r[j].d = a.dists[i * f.k + j];
                }
                f.lists[i].nbgh = (int64_t)c;
                f.lists[i].neighbours = r;
            }
        },
        &ans,
        [](void* ctx, DeviceIndex::DirectOut* out) -> bool {
            FfiAnswer& f = *static_cast<FfiAnswer*>(ctx);
            if (!f.slab || f.slab->dev == 0) return false;
            static_assert(sizeof(Neighbour_api) == 16 && sizeof(Neighbourhood_api) == 16 && sizeof(size_t) == 8, "the in-place layout");
            out->allocation = f.slab;
            out->ids = &f.rows[0].id;
            out->dists = &f.rows[0].d;
            out->counts = &f.lists[0].nbgh;
            out->layout = hnswgpu::OutLayout{16, 16, 16};
            return true;
        }};
    std::string err;
    int rc;
    {
        std::shared_lock<std::shared_mutex> sl(idx->mu);
        const bool empty = idx->builder ? idx->builder->nb_point() == 0 : (!idx->flat || idx->flat->n == 0);
        if (empty || nb_vec == 0) {  // empty index => every answer is empty (src/hnsw.rs:1498-1503)
            std::vector<uint32_t> zero(std::max<size_t>(1, nb_vec), 0u);
            DeviceIndex::HostAnswers a{};
            a.counts = zero.data();
            if (!sink.begin(&ans, nb_vec, knbn)) { fail(HNSWGPU_ERR_ARG, "out of memory"); return nullptr; }
            sink.rows(&ans, a, 0, nb_vec);
            return ans.out;
        }
        DeviceIndex* dev = nullptr;
        rc = primary_replica(idx, sl, &dev);
        if (rc != HNSWGPU_OK) return nullptr;
        rc = dev->search_host_staged(nullptr, data, nb_vec, (uint64_t)vec_len, knbn, ef_search, nullptr, 0, false, false, sink, nullptr, err);
    }
    if (rc != OK) {
        if (ans.out) hnswgpu_free_neighbourhood_vec(ans.out);  // (taken before the search started)
        fail(rc, err);
        return nullptr;
    }
    if (!ans.out) fail(HNSWGPU_ERR_ARG, "out of memory");
    return ans.out;
    CAPI_GUARD_END(nullptr)
}

void hnswgpu_free_neighbourhood(const Neighbourhood_api* p) {
    if (!p) return;
    std::free(const_cast<Neighbour_api*>(p->neighbours));
    std::free(const_cast<Neighbourhood_api*>(p));
}
// Every Vec_api this library hands out is the head of ONE slab (header | Vec_api | Neighbourhood_api[] | Neighbour_api[]): the
// rows are interior pointers and must never be freed one by one; the whole answer is released here, and only here.
void hnswgpu_free_neighbourhood_vec(const Vec_api_Neighbourhood* p) {
    if (!p) return;
    SlabHeader* h = reinterpret_cast<SlabHeader*>(reinterpret_cast<unsigned char*>(const_cast<Vec_api_Neighbourhood*>(p)) - sizeof(SlabHeader));
    if (h->magic != SLAB_MAGIC) return;  // not an answer of this library, or freed already
    slab_cache().give(h);
}

int64_t file_dump_f32(const HnswApif32* api, size_t namelen, const uint8_t* filename) {
    CAPI_GUARD_BEGIN
    if (!api || !api->idx || !filename) return -1;
    std::string base(reinterpret_cast<const char*>(filename), namelen);
    return hnswgpu_file_dump(api->idx, ".", base.c_str()) == HNSWGPU_OK ? 1 : -1;  // 1 / -1 (:269-272)
    CAPI_GUARD_END(-1)
}

void drop_hnsw_f32(const HnswApif32* p) {
    if (!p) return;
    hnswgpu_free_index(p->idx);
    delete p;
}

const DescriptionFFI* load_hnsw_description(size_t flen, const uint8_t* name) {
    CAPI_GUARD_BEGIN
    if (!name) return nullptr;
    std::string path(reinterpret_cast<const char*>(name), flen);
    DumpDescription d;
    std::string err;
    if (load_description_file(path, d, err) != OK) {
        fail(HNSWGPU_ERR_IO, err);
        return nullptr;
    }
    DescriptionFFI* f = static_cast<DescriptionFFI*>(std::calloc(1, sizeof(DescriptionFFI)));
    f->dumpmode = 1;  // the reference always reports 1 and never fills nb_point (src/libext.rs:1198-1206)
    f->max_nb_connection = d.max_nb_connection;
    f->nb_layer = d.nb_layer;
    f->ef = d.ef;
    f->nb_point = 0;
    f->data_dimension = d.dimension;
    char* dn = static_cast<char*>(std::malloc(d.distname.size() + 1));
    std::memcpy(dn, d.distname.c_str(), d.distname.size() + 1);
    f->distname_len = d.distname.size();
    f->distname = reinterpret_cast<const uint8_t*>(dn);
    char* tn = static_cast<char*>(std::malloc(d.t_name.size() + 1));
    std::memcpy(tn, d.t_name.c_str(), d.t_name.size() + 1);
    f->t_name_len = d.t_name.size();
    f->t_name = reinterpret_cast<const uint8_t*>(tn);
    return f;
    CAPI_GUARD_END(nullptr)
}
void hnswgpu_free_description(const DescriptionFFI* p) {
    if (!p) return;
    std::free(const_cast<uint8_t*>(p->distname));
    std::free(const_cast<uint8_t*>(p->t_name));
    std::free(const_cast<DescriptionFFI*>(p));
}

void init_rust_log(void) {}

hnswgpu_index* hnswgpu_from_api(const HnswApif32* p) { return p ? p->idx : nullptr; }

}  // extern "C"
