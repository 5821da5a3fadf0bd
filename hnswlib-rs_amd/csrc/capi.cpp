// capi.cpp -- the C ABI declared in include/hnsw_mi355x.h: the thin hnswgpu_* entry points and the
// name/layout-compatible replacements of the reference's own f32 FFI (src/libext.rs).
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hnsw_mi355x.h"
#include "builder.hpp"
#include "flat_index.hpp"
#include "hnswio.hpp"
#include "search_device.hpp"

using namespace hnswgpu;

static thread_local std::string g_last_error;
static int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

struct hnswgpu_index {
    std::mutex mu;                          // one search / mutation at a time per handle
    std::unique_ptr<FlatIndex> flat;        // dump-order view; rebuilt from `builder` when stale
    std::unique_ptr<GraphBuilder> builder;  // present for indexes created by hnswgpu_build / init_hnsw_f32
    bool flat_stale = false;
    std::unique_ptr<DeviceIndex> dev;
    bool dev_stale = true;
    int strict_ties = -1;  // -1: library default (env HNSWGPU_STRICT_TIES, else on)
    BuildParams params;

    const FlatIndex* get_flat() {
        if (builder && (flat_stale || !flat)) {
            flat.reset(new FlatIndex());
            builder->finalize(*flat);
            flat_stale = false;
            dev_stale = true;
        }
        return flat.get();
    }
};

static int default_device() {
    const char* e = std::getenv("HNSWGPU_DEVICE");
    return e ? std::atoi(e) : 0;
}

// make sure the HBM replica reflects the host graph (lazy for the reference-style entry points)
static int ensure_device(hnswgpu_index* idx, int device) {
    const FlatIndex* f = idx->get_flat();
    if (!f || f->n == 0) return fail(HNSWGPU_ERR_EMPTY, "index is empty");
    if (idx->dev && idx->dev->ready() && !idx->dev_stale && (device < 0 || device == idx->dev->device())) return HNSWGPU_OK;
    if (device < 0) device = idx->dev && idx->dev->ready() ? idx->dev->device() : default_device();
    idx->dev.reset(new DeviceIndex());
    std::string err;
    int rc = idx->dev->upload(*f, device, err);
    if (rc != OK) {
        idx->dev.reset();
        return fail(rc, err);
    }
    idx->dev_stale = false;
    if (idx->strict_ties >= 0) idx->dev->set_strict_ties(idx->strict_ties != 0);
    return HNSWGPU_OK;
}

extern "C" {

const char* hnswgpu_last_error(void) { return g_last_error.c_str(); }

int hnswgpu_load_dump(const char* dir, const char* basename, int dist, hnswgpu_index** out) {
    if (!dir || !basename || !out) return fail(HNSWGPU_ERR_ARG, "null argument");
    *out = nullptr;
    std::unique_ptr<hnswgpu_index> h(new hnswgpu_index());
    h->flat.reset(new FlatIndex());
    std::string err;
    int rc = load_dump(dir, basename, dist, *h->flat, err);
    if (rc != OK) return fail(rc, err);
    *out = h.release();
    return HNSWGPU_OK;
}

int hnswgpu_file_dump(const hnswgpu_index* cidx, const char* dir, const char* basename) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || !dir || !basename) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    if (!f) return fail(HNSWGPU_ERR_EMPTY, "entry point not initialized");
    std::string err;
    int rc = write_dump(*f, dir, basename, err);
    if (rc != OK) return fail(rc, err);
    return HNSWGPU_OK;
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
    if (!graph_file_path || !out) return fail(HNSWGPU_ERR_ARG, "null argument");
    DumpDescription d;
    std::string err;
    int rc = load_description_file(graph_file_path, d, err);
    if (rc != OK) return fail(rc, err);
    fill_descr(out, d.format_version, d.dumpmode, d.max_nb_connection, d.nb_layer, d.level_scale, d.ef, d.nb_point,
               d.dimension, d.distname, d.t_name);
    return HNSWGPU_OK;
}

int hnswgpu_get_description(const hnswgpu_index* cidx, hnswgpu_description* out) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || !out) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    if (!f) return fail(HNSWGPU_ERR_EMPTY, "index is empty");
    fill_descr(out, f->format_version, f->dumpmode, (uint8_t)f->max_nb_connection, f->nb_layer, f->level_scale,
               f->ef_construction, f->n, f->dimension, f->distname.empty() ? dist_type_name(f->dist) : f->distname, f->t_name);
    return HNSWGPU_OK;
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
    return b;
}

int hnswgpu_build(const float* data, uint64_t n, uint64_t d, const uint64_t* ids, const hnswgpu_build_params* params,
                  hnswgpu_index** out) {
    if (!params || !out || (n && !data)) return fail(HNSWGPU_ERR_ARG, "null argument");
    *out = nullptr;
    if (params->dist < 0 || params->dist > 3) return fail(HNSWGPU_ERR_DISTANCE, "unknown distance");
    if (params->max_nb_connection < 2 || params->max_nb_connection > 256)
        return fail(HNSWGPU_ERR_ARG, "error max_nb_connection must be less equal than 256");  // src/hnsw.rs:784-787
    std::unique_ptr<hnswgpu_index> h(new hnswgpu_index());
    h->params = to_params(params);
    h->builder.reset(new GraphBuilder(h->params));
    std::string err;
    int rc = h->builder->insert_batch(data, n, d, ids, h->params.nthreads, err);
    if (rc != OK) return fail(rc, err);
    h->flat_stale = true;
    *out = h.release();
    return HNSWGPU_OK;
}

uint64_t hnswgpu_nb_point(const hnswgpu_index* cidx) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return 0;
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->builder) return idx->builder->nb_point();
    return idx->flat ? idx->flat->n : 0;
}
uint64_t hnswgpu_dimension(const hnswgpu_index* cidx) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return 0;
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->builder) return idx->builder->dimension();
    return idx->flat ? idx->flat->dimension : 0;
}
int hnswgpu_dist(const hnswgpu_index* cidx) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return -1;
    if (idx->builder) return idx->params.dist;
    return idx->flat ? idx->flat->dist : -1;
}
uint64_t hnswgpu_layer_nb_point(const hnswgpu_index* cidx, unsigned layer) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || layer >= NB_LAYER_MAX) return 0;
    std::lock_guard<std::mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    return f ? f->layer_count(layer) : 0;
}
int hnswgpu_max_level_observed(const hnswgpu_index* cidx) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return 0;
    std::lock_guard<std::mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    if (!f || f->entry_flat == NO_POINT) return 0;
    return (int)f->layer_of(f->entry_flat);
}
int hnswgpu_entry_point(const hnswgpu_index* cidx, uint64_t* origin_id, uint8_t* layer, int32_t* rank) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    if (!f || f->entry_flat == NO_POINT) return fail(HNSWGPU_ERR_EMPTY, "index is empty");
    if (origin_id) *origin_id = f->origin_id[f->entry_flat];
    if (layer) *layer = (uint8_t)f->layer_of(f->entry_flat);
    if (rank) *rank = f->rank_of(f->entry_flat);
    return HNSWGPU_OK;
}
int64_t hnswgpu_neighbours(const hnswgpu_index* cidx, unsigned layer, int32_t rank, unsigned l, uint64_t cap,
                           uint64_t* origin_ids, uint8_t* layers, int32_t* ranks, float* dists) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || layer >= NB_LAYER_MAX || l >= NB_LAYER_MAX || rank < 0) { fail(HNSWGPU_ERR_ARG, "bad argument"); return -1; }
    std::lock_guard<std::mutex> g(idx->mu);
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
}

int hnswgpu_device_count(void) { return device_count(); }

int hnswgpu_upload(hnswgpu_index* idx, int device) {
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    return ensure_device(idx, device);
}

int hnswgpu_search_batch(const hnswgpu_index* cidx, const float* queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef,
                         uint64_t* out_ids, float* out_dists, uint8_t* out_layer, int32_t* out_rank, uint32_t* out_counts) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    const FlatIndex* f = idx->get_flat();
    if (!f || f->n == 0) {  // empty index => every answer is empty (src/hnsw.rs:1498-1503)
        if (out_counts) std::memset(out_counts, 0, nq * sizeof(uint32_t));
        return HNSWGPU_OK;
    }
    int rc = ensure_device(idx, -1);
    if (rc != HNSWGPU_OK) return rc;
    std::string err;
    rc = idx->dev->search_host(queries, nq, d, k, ef, out_ids, out_dists, out_layer, out_rank, out_counts, err);
    if (rc != OK) return fail(rc, err);
    return HNSWGPU_OK;
}

int hnswgpu_search_batch_device(const hnswgpu_index* cidx, const float* d_queries, uint64_t nq, uint64_t d, uint64_t k,
                                uint64_t ef, uint64_t* d_out_ids, float* d_out_dists, uint8_t* d_out_layer,
                                int32_t* d_out_rank, uint32_t* d_out_counts, uint32_t* d_stats, void* stream) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (!idx->dev || !idx->dev->ready() || idx->dev_stale || idx->flat_stale)
        return fail(HNSWGPU_ERR_DEVICE, "index is not resident on a device: call hnswgpu_upload first");
    std::string err;
    int rc = idx->dev->search_device(d_queries, nq, d, k, ef, d_out_ids, d_out_dists, d_out_layer, d_out_rank, d_out_counts,
                                     d_stats, stream, err);
    if (rc != OK) return fail(rc, err);
    return HNSWGPU_OK;
}

int hnswgpu_last_kernel_ms(const hnswgpu_index* cidx, double* ms, uint32_t* launches) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || !idx->dev) return fail(HNSWGPU_ERR_DEVICE, "index is not resident on a device");
    if (ms) *ms = idx->dev->last_kernel_ms();
    if (launches) *launches = idx->dev->last_launches();
    return HNSWGPU_OK;
}

int hnswgpu_last_search_kernel_ms(const hnswgpu_index* cidx, double* ms) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || !idx->dev || !ms) return fail(HNSWGPU_ERR_DEVICE, "index is not resident on a device");
    *ms = idx->dev->last_main_kernel_ms();
    return HNSWGPU_OK;
}
int hnswgpu_set_strict_ties(hnswgpu_index* idx, int on) {
    if (!idx) return fail(HNSWGPU_ERR_ARG, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    idx->strict_ties = on != 0;
    if (idx->dev) idx->dev->set_strict_ties(idx->strict_ties);
    return HNSWGPU_OK;
}
int hnswgpu_last_tie_count(const hnswgpu_index* cidx, uint32_t* ties) {
    hnswgpu_index* idx = const_cast<hnswgpu_index*>(cidx);
    if (!idx || !idx->dev || !ties) return fail(HNSWGPU_ERR_DEVICE, "index is not resident on a device");
    *ties = idx->dev->last_ties();
    return HNSWGPU_OK;
}

int hnswgpu_eval_distances(int dist, const float* a, const float* b, uint64_t n, uint64_t d, float* out) {
    if (!a || !b || !out || dist < 0 || dist > 3) return fail(HNSWGPU_ERR_ARG, "bad argument");
    std::string err;
    int rc = eval_distances_device(dist, a, b, n, d, out, err);
    if (rc != OK) return fail(rc, err);
    return HNSWGPU_OK;
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
    if (!name) return nullptr;
    HnswIo* io = new HnswIo();
    io->dir = ".";
    io->basename.assign(reinterpret_cast<const char*>(name), (size_t)flen);
    return io;
}
void hnswgpu_free_hnswio(const HnswIo* p) { delete p; }

static const HnswApif32* load_with(HnswIo* io, int dist) {
    if (!io) return nullptr;
    hnswgpu_index* idx = nullptr;
    if (hnswgpu_load_dump(io->dir.c_str(), io->basename.c_str(), dist, &idx) != HNSWGPU_OK) return nullptr;  // null on failure (:298-301)
    HnswApif32* api = new HnswApif32();
    api->idx = idx;
    return api;
}
const HnswApif32* load_hnswdump_f32_DistL1(HnswIo* io) { return load_with(io, HNSWGPU_DIST_L1); }
const HnswApif32* load_hnswdump_f32_DistL2(HnswIo* io) { return load_with(io, HNSWGPU_DIST_L2); }
const HnswApif32* load_hnswdump_f32_DistCosine(HnswIo* io) { return load_with(io, HNSWGPU_DIST_COSINE); }
const HnswApif32* load_hnswdump_f32_DistDot(HnswIo* io) { return load_with(io, HNSWGPU_DIST_DOT); }

static const HnswApif32* new_api(size_t max_nb_conn, size_t ef_const, size_t namelen, const uint8_t* cdistname,
                                 size_t max_elements, size_t max_layer, bool allow_cosine) {
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
}
const HnswApif32* init_hnsw_f32(size_t max_nb_conn, size_t ef_const, size_t namelen, const uint8_t* cdistname) {
    return new_api(max_nb_conn, ef_const, namelen, cdistname, 10000, 16, false);  // Hnsw::new(M, 10000, 16, ef_c, D) (:475)
}
const HnswApif32* new_hnsw_f32(size_t max_nb_conn, size_t ef_const, size_t namelen, const uint8_t* cdistname,
                               size_t max_elements, size_t max_layer) {
    return new_api(max_nb_conn, ef_const, namelen, cdistname, max_elements, max_layer, false);
}

void insert_f32(HnswApif32* api, size_t len, const float* data, size_t id) {
    if (!api || !api->idx || !api->idx->builder || !data) return;
    hnswgpu_index* idx = api->idx;
    std::lock_guard<std::mutex> g(idx->mu);
    uint64_t id64 = id;
    std::string err;
    if (idx->builder->insert_batch(data, 1, len, &id64, 1, err) != OK) { fail(HNSWGPU_ERR_ARG, err); return; }
    idx->flat_stale = true;
    idx->dev_stale = true;
}
void parallel_insert_f32(HnswApif32* api, size_t nb_vec, size_t vec_len, const float** datas, const size_t* ids) {
    if (!api || !api->idx || !api->idx->builder || !datas || !ids) return;
    hnswgpu_index* idx = api->idx;
    std::lock_guard<std::mutex> g(idx->mu);
    std::vector<float> flat(nb_vec * vec_len);  // inputs are copied, like the reference (:700-712)
    std::vector<uint64_t> id64(nb_vec);
    for (size_t i = 0; i < nb_vec; ++i) {
        std::memcpy(flat.data() + i * vec_len, datas[i], vec_len * sizeof(float));
        id64[i] = ids[i];
    }
    std::string err;
    if (idx->builder->insert_batch(flat.data(), nb_vec, vec_len, id64.data(), 0, err) != OK) { fail(HNSWGPU_ERR_ARG, err); return; }
    idx->flat_stale = true;
    idx->dev_stale = true;
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
}

const Vec_api_Neighbourhood* parallel_search_neighbours_f32(const HnswApif32* api, size_t nb_vec, int64_t vec_len,
                                                            const float** data, size_t knbn, size_t ef_search) {
    if (!api || !api->idx || !data || knbn == 0 || vec_len <= 0) return nullptr;
    // array-of-pointers input is copied into one matrix (the reference copies into Vec<Vec<f32>>, :218-226)
    std::vector<float> q((size_t)nb_vec * (size_t)vec_len);
    for (size_t i = 0; i < nb_vec; ++i) std::memcpy(q.data() + i * (size_t)vec_len, data[i], (size_t)vec_len * sizeof(float));
    std::vector<uint64_t> ids(nb_vec * knbn);
    std::vector<float> dists(nb_vec * knbn);
    std::vector<uint32_t> cnt(nb_vec);
    if (hnswgpu_search_batch(api->idx, q.data(), nb_vec, (uint64_t)vec_len, knbn, ef_search, ids.data(), dists.data(), nullptr,
                             nullptr, cnt.data()) != HNSWGPU_OK)
        return nullptr;
    Neighbourhood_api* lists = static_cast<Neighbourhood_api*>(std::malloc(std::max<size_t>(1, nb_vec) * sizeof(Neighbourhood_api)));
    for (size_t i = 0; i < nb_vec; ++i) {
        lists[i].nbgh = cnt[i];
        lists[i].neighbours = make_row(ids.data() + i * knbn, dists.data() + i * knbn, cnt[i]);
    }
    Vec_api_Neighbourhood* ans = static_cast<Vec_api_Neighbourhood*>(std::malloc(sizeof(Vec_api_Neighbourhood)));
    ans->len = (int64_t)nb_vec;
    ans->ptr = lists;
    return ans;
}

void hnswgpu_free_neighbourhood(const Neighbourhood_api* p) {
    if (!p) return;
    std::free(const_cast<Neighbour_api*>(p->neighbours));
    std::free(const_cast<Neighbourhood_api*>(p));
}
void hnswgpu_free_neighbourhood_vec(const Vec_api_Neighbourhood* p) {
    if (!p) return;
    for (int64_t i = 0; i < p->len; ++i) std::free(const_cast<Neighbour_api*>(p->ptr[i].neighbours));
    std::free(const_cast<Neighbourhood_api*>(p->ptr));
    std::free(const_cast<Vec_api_Neighbourhood*>(p));
}

int64_t file_dump_f32(const HnswApif32* api, size_t namelen, const uint8_t* filename) {
    if (!api || !api->idx || !filename) return -1;
    std::string base(reinterpret_cast<const char*>(filename), namelen);
    return hnswgpu_file_dump(api->idx, ".", base.c_str()) == HNSWGPU_OK ? 1 : -1;  // 1 / -1 (:269-272)
}

void drop_hnsw_f32(const HnswApif32* p) {
    if (!p) return;
    hnswgpu_free_index(p->idx);
    delete p;
}

const DescriptionFFI* load_hnsw_description(size_t flen, const uint8_t* name) {
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
