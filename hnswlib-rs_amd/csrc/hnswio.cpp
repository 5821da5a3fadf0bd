// hnswio.cpp -- reader / writer of the hnsw_rs two-file dump ("hnswio") into/from FlatIndex.
//
//   <base>.hnsw.graph : Description (src/hnswio.rs:878-919, read :937-1042)
//                       PointIndexation: nb_layer, per layer {MAGICLAYER, count, point records
//                       (:1063-1097, read :1221-1289)}, entry point (:1303-1340)
//   <base>.hnsw.data  : {MAGICDATAP, dimension} then per point
//                       {MAGICDATAP, origin_id u64, byte_len u64, raw f32[d]} (:1099-1112, :1382-1383); format v2 (read only,
//                       :1157-1158): the payload is bincode's Vec<f32> = u64 count + the elements
// Native-endian, packed, usize = 8 bytes.  Only DumpMode::Full exists
// in practice (src/api.rs:81) and only Full can be reloaded (src/hnswio.rs:1237-1243).
//
// Unlike the reference (which rebuilds an Arc web through a HashMap<PointId, ...>, :642-737),
// the files are mmapped and parsed in one forward pass straight into flat arrays.
#include "hnswio.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>

namespace hnswgpu {

static constexpr uint32_t MAGICPOINT = 0x000a678f;    // src/hnswio.rs:47
static constexpr uint32_t MAGICDESCR_2 = 0x002a677f;  // :49
static constexpr uint32_t MAGICDESCR_3 = 0x002a6771;  // :56
static constexpr uint32_t MAGICDESCR_4 = 0x002a6779;  // :60
static constexpr uint32_t MAGICLAYER = 0x000a676f;    // :63
static constexpr uint32_t MAGICDATAP = 0xa67f0000;    // :65

namespace {

struct MappedFile {
    const uint8_t* p = nullptr;
    size_t size = 0;
    int fd = -1;
    bool open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        size = (size_t)st.st_size;
        if (size == 0) { p = nullptr; return true; }
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = static_cast<const uint8_t*>(m);
        madvise(m, size, MADV_SEQUENTIAL);
        return true;
    }
    ~MappedFile() {
        if (p) munmap(const_cast<uint8_t*>(p), size);
        if (fd >= 0) ::close(fd);
    }
};

struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    template <class T>
    T get() {
        T v{};
        if ((size_t)(end - p) < sizeof(T)) { ok = false; p = end; return v; }
        std::memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    const uint8_t* bytes(size_t n) {
        if ((size_t)(end - p) < n) { ok = false; p = end; return nullptr; }
        const uint8_t* r = p;
        p += n;
        return r;
    }
};

int parse_description(Cursor& c, DumpDescription& d, std::string& err) {
    uint32_t magic = c.get<uint32_t>();
    if (!c.ok) { err = "truncated description"; return ERR_FORMAT; }
    if (magic == MAGICDESCR_2) d.format_version = 2;
    else if (magic == MAGICDESCR_3) d.format_version = 3;
    else if (magic == MAGICDESCR_4) d.format_version = 4;
    else { err = "bad magic at descr beginning"; return ERR_FORMAT; }
    d.dumpmode = c.get<uint8_t>();
    d.max_nb_connection = c.get<uint8_t>();
    d.level_scale = 1.0;
    if (d.format_version == 4) d.level_scale = c.get<double>();
    d.nb_layer = c.get<uint8_t>();
    d.ef = c.get<uint64_t>();
    d.nb_point = c.get<uint64_t>();
    d.dimension = c.get<uint64_t>();
    uint64_t len = c.get<uint64_t>();
    if (!c.ok) { err = "truncated description"; return ERR_FORMAT; }
    if (len > 256) { err = "bad length for distance name"; return ERR_FORMAT; }
    const uint8_t* s = c.bytes(len);
    if (!s && len) { err = "truncated description"; return ERR_FORMAT; }
    d.distname.assign(reinterpret_cast<const char*>(s), len);
    len = c.get<uint64_t>();
    if (!c.ok) { err = "truncated description"; return ERR_FORMAT; }
    if (len > 256) { err = "bad lenght for T name"; return ERR_FORMAT; }
    s = c.bytes(len);
    if (!s && len) { err = "truncated description"; return ERR_FORMAT; }
    d.t_name.assign(reinterpret_cast<const char*>(s), len);
    return OK;
}

}  // namespace

int load_description_file(const std::string& graph_path, DumpDescription& d, std::string& err) {
    MappedFile f;
    if (!f.open(graph_path)) { err = "could not open file " + graph_path; return ERR_IO; }
    Cursor c{f.p, f.p + f.size};
    return parse_description(c, d, err);
}

int load_dump(const std::string& dir, const std::string& basename, int asked_dist, FlatIndex& out, std::string& err) {
    const std::string gpath = dir + "/" + basename + ".hnsw.graph";
    const std::string dpath = dir + "/" + basename + ".hnsw.data";
    MappedFile gf, df;
    if (!gf.open(gpath)) { err = "HnswIo::init : could not open file " + gpath; return ERR_IO; }
    if (!df.open(dpath)) { err = "HnswIo::init : could not open file " + dpath; return ERR_IO; }
    Cursor g{gf.p, gf.p + gf.size};
    Cursor dt{df.p, df.p + df.size};

    DumpDescription descr;
    int rc = parse_description(g, descr, err);
    if (rc != OK) return rc;
    // data header (src/hnswio.rs:450-466)
    if (dt.get<uint32_t>() != MAGICDATAP || !dt.ok) { err = "magic not equal to MAGICDATAP in load_point"; return ERR_FORMAT; }
    if (dt.get<uint64_t>() != descr.dimension) { err = "data dimension incoherent"; return ERR_FORMAT; }
    // distance short-name rule (src/hnswio.rs:473-490)
    const std::string dumped_short = short_name(descr.distname);
    int file_dist = dist_from_short_name(dumped_short);
    if (asked_dist >= 0) {
        if (short_name(dist_type_name(asked_dist)) != dumped_short) {
            err = "error in distances : dumped distance is : " + descr.distname + " asked distance in loading is : " +
                  dist_type_name(asked_dist);
            return ERR_DISTANCE;
        }
        file_dist = asked_dist;
    } else if (file_dist < 0) {
        err = "dump uses a distance this library does not implement: " + descr.distname;
        return ERR_DISTANCE;
    }
    // element type (src/hnswio.rs:629-638: the reference panics)
    if (descr.t_name != "f32") { err = "typename in description (" + descr.t_name + ") is not f32"; return ERR_TYPE; }
    if (descr.dumpmode != 1) { err = "only DumpMode::Full dumps can be reloaded"; return ERR_FORMAT; }

    out = FlatIndex();
    out.format_version = descr.format_version;
    out.dumpmode = descr.dumpmode;
    out.max_nb_connection = descr.max_nb_connection;
    // Reference quirk, kept: the dump stores the generator's ABSOLUTE scale (get_level_scale(),
    // src/hnswio.rs:1365-1371), and the reload hands it to LayerGenerator::new_with_scale as a FACTOR of 1/ln(M)
    // (src/hnswio.rs:773-777, src/hnsw.rs:339-352).  A reloaded index therefore draws the levels of further points with,
    // and dumps again, level_scale / ln(M).
    out.level_scale = descr.level_scale / std::log((double)std::max<uint64_t>(2, descr.max_nb_connection));
    out.nb_layer = descr.nb_layer;
    out.ef_construction = descr.ef;
    out.dimension = descr.dimension;
    out.distname = descr.distname;
    out.t_name = descr.t_name;
    out.dist = file_dist;
    out.extend_candidates = true;  // reloaded Hnsw (src/hnswio.rs:510)

    const uint64_t d = descr.dimension;
    uint8_t nb_layer = g.get<uint8_t>();
    if (!g.ok) { err = "truncated graph file"; return ERR_FORMAT; }
    if (nb_layer > NB_LAYER_MAX) { err = "inconsistent number of layers"; return ERR_FORMAT; }
    // The header fields are untrusted (a truncated or corrupt file must give ERR_FORMAT, not a length_error thrown
    // across the C ABI): reserves are bounded by what the two mapped files can actually hold -- a point record is at
    // least 17 + 8 * nb_layer bytes of graph file and 20 + 4 d bytes of data file, an edge is 17 bytes.
    if (d == 0 || d > (df.size / sizeof(float))) { err = "data dimension incoherent with the data file size"; return ERR_FORMAT; }
    const uint64_t n_hint = std::min<uint64_t>({descr.nb_point, (uint64_t)gf.size / (17u + 8u * (uint64_t)descr.nb_layer),
                                                (uint64_t)df.size / (20u + 4u * d)});
    const uint64_t edge_hint = std::min<uint64_t>((uint64_t)gf.size / 17u,
                                                  n_hint * 2 * std::max<uint64_t>(1, descr.max_nb_connection));
    out.origin_id.reserve(n_hint);
    out.vectors.reserve(n_hint * d);
    out.nbr_ptr.reserve(n_hint * NB_LAYER_MAX + 1);
    out.nbr_ptr.push_back(0);
    // neighbours are first kept as packed (layer<<32 | rank): a neighbour may live in a layer that
    // comes later in the file, so flat ids are resolved once all layer counts are known.
    std::vector<uint64_t> nbr_packed;
    nbr_packed.reserve(edge_hint);
    out.nbr_dist.reserve(edge_hint);
    std::vector<uint64_t> nbr_origin;  // kept for the coherence check only
    nbr_origin.reserve(edge_hint);

    uint64_t n = 0;
    for (unsigned l = 0; l < nb_layer; ++l) {
        if (g.get<uint32_t>() != MAGICLAYER || !g.ok) { err = "bad magic at layer beginning"; return ERR_FORMAT; }
        uint64_t nbpoints = g.get<uint64_t>();
        out.layer_offset[l] = n;
        for (uint64_t r = 0; r < nbpoints; ++r) {
            if (g.get<uint32_t>() != MAGICPOINT || !g.ok) { err = "bad magic at point beginning"; return ERR_FORMAT; }
            uint64_t origin = g.get<uint64_t>();
            uint8_t pl = g.get<uint8_t>();
            int32_t pr = g.get<int32_t>();
            if (!g.ok) { err = "truncated graph file"; return ERR_FORMAT; }
            if (pl != l || pr < 0 || (uint64_t)pr != r) {  // asserted at reload, src/hnswio.rs:703-708
                err = "point id incoherent with its position in the dump";
                return ERR_FORMAT;
            }
            for (unsigned ll = 0; ll < NB_LAYER_MAX; ++ll) {
                if (ll < descr.nb_layer) {
                    uint64_t nbn = g.get<uint64_t>();
                    if (!g.ok || nbn > (uint64_t)(g.end - g.p) / 17) { err = "truncated graph file"; return ERR_FORMAT; }
                    for (uint64_t j = 0; j < nbn; ++j) {
                        uint64_t nid = g.get<uint64_t>();
                        uint8_t nl = g.get<uint8_t>();
                        int32_t nr = g.get<int32_t>();
                        float nd = g.get<float>();
                        if (nl >= NB_LAYER_MAX || nr < 0) { err = "bad neighbour point id"; return ERR_FORMAT; }
                        nbr_packed.push_back(((uint64_t)nl << 32) | (uint32_t)nr);
                        nbr_origin.push_back(nid);
                        out.nbr_dist.push_back(nd);
                    }
                }
                out.nbr_ptr.push_back(nbr_packed.size());
            }
            // data record
            if (dt.get<uint32_t>() != MAGICDATAP || !dt.ok) { err = "magic not equal to MAGICDATAP in load_point"; return ERR_FORMAT; }
            if (dt.get<uint64_t>() != origin) { err = "origin_id incoherent between graph and data"; return ERR_FORMAT; }
            uint64_t slen = dt.get<uint64_t>();
            const uint8_t* raw = dt.bytes(slen);
            if (descr.format_version == 2) {
                // v2 dumps (magic 0x002a677f) hold the vector bincode-encoded (src/hnswio.rs:1157-1158: bincode::deserialize of a
                // Vec<T>; bincode 1 defaults: a u64 little-endian element count, then the elements, little endian)
                uint64_t cnt = 0;
                if (!dt.ok || slen < 8) { err = "truncated data file"; return ERR_FORMAT; }
                std::memcpy(&cnt, raw, 8);
                if (cnt != d) { err = "bincode-encoded vector (dump format v2) does not have the dimension of the description"; return ERR_FORMAT; }
                raw += 8;
                slen -= 8;
            }
            if (!dt.ok || slen / sizeof(float) < d) { err = "truncated data file"; return ERR_FORMAT; }
            out.origin_id.push_back(origin);
            size_t off = out.vectors.size();
            out.vectors.resize(off + d);
            std::memcpy(out.vectors.data() + off, raw, d * sizeof(float));
            ++n;
        }
    }
    if (!g.ok) { err = "truncated graph file"; return ERR_FORMAT; }
    for (unsigned l = nb_layer; l <= NB_LAYER_MAX; ++l) out.layer_offset[l] = n;
    out.n = n;
    if (n >= NO_POINT) { err = "index too large for 32-bit flat ids"; return ERR_FORMAT; }

    // resolve (layer, rank) -> flat, check coherence of origin ids
    out.nbr_flat.resize(nbr_packed.size());
    for (size_t e = 0; e < nbr_packed.size(); ++e) {
        unsigned nl = (unsigned)(nbr_packed[e] >> 32);
        uint64_t nr = (uint32_t)nbr_packed[e];
        if (nr >= out.layer_count(nl)) { err = "neighbour refers to a point that is not in the dump"; return ERR_FORMAT; }
        uint64_t flat = out.layer_offset[nl] + nr;
        if (out.origin_id[flat] != nbr_origin[e]) { err = "neighbour origin id incoherent with its point id"; return ERR_FORMAT; }
        out.nbr_flat[e] = (uint32_t)flat;
    }
    // reload re-sorts every list by stored distance (src/hnswio.rs:731).  A no-op for files written
    // by the crate (lists are kept ascending: src/hnsw.rs:1195, :1280); done only when needed.
    {
        std::vector<uint32_t> perm;
        std::vector<uint32_t> tf;
        std::vector<float> td;
        for (size_t li = 0; li + 1 < out.nbr_ptr.size(); ++li) {
            uint64_t b = out.nbr_ptr[li], e = out.nbr_ptr[li + 1];
            if (e - b < 2) continue;
            if (std::is_sorted(out.nbr_dist.begin() + b, out.nbr_dist.begin() + e)) continue;
            perm.resize(e - b);
            std::iota(perm.begin(), perm.end(), 0u);
            std::stable_sort(perm.begin(), perm.end(),
                             [&](uint32_t x, uint32_t y) { return out.nbr_dist[b + x] < out.nbr_dist[b + y]; });
            tf.assign(out.nbr_flat.begin() + b, out.nbr_flat.begin() + e);
            td.assign(out.nbr_dist.begin() + b, out.nbr_dist.begin() + e);
            for (size_t i = 0; i < perm.size(); ++i) {
                out.nbr_flat[b + i] = tf[perm[i]];
                out.nbr_dist[b + i] = td[perm[i]];
            }
        }
    }
    // entry point trailer (src/hnswio.rs:746-762)
    if (n > 0) {
        uint64_t ep_origin = g.get<uint64_t>();
        uint8_t ep_layer = g.get<uint8_t>();
        int32_t ep_rank = g.get<int32_t>();
        if (!g.ok) { err = "truncated graph file (entry point)"; return ERR_FORMAT; }
        if (ep_layer >= NB_LAYER_MAX || ep_rank < 0 || (uint64_t)ep_rank >= out.layer_count(ep_layer)) {
            err = "entry point is not in the dump";
            return ERR_FORMAT;
        }
        out.entry_flat = (uint32_t)(out.layer_offset[ep_layer] + (uint64_t)ep_rank);
        if (out.origin_id[out.entry_flat] != ep_origin) { err = "entry point origin id incoherent"; return ERR_FORMAT; }
    } else {
        // the reference cannot dump an empty index ("entry point not initialized", :1323-1325)
        err = "empty dump: no entry point";
        return ERR_FORMAT;
    }
    return OK;
}

namespace {
struct BufWriter {
    FILE* f = nullptr;
    bool ok = true;
    explicit BufWriter(const std::string& path) {
        f = std::fopen(path.c_str(), "wb");
        if (f) std::setvbuf(f, nullptr, _IOFBF, 1 << 22);
        else ok = false;
    }
    ~BufWriter() { if (f) std::fclose(f); }
    template <class T>
    void put(const T& v) { if (ok && std::fwrite(&v, sizeof(T), 1, f) != 1) ok = false; }
    void bytes(const void* p, size_t n) { if (ok && n && std::fwrite(p, 1, n, f) != n) ok = false; }
    bool close() {
        if (f) { if (std::fclose(f) != 0) ok = false; f = nullptr; }
        return ok;
    }
};
}  // namespace

int write_dump(const FlatIndex& x, const std::string& dir, const std::string& basename, std::string& err) {
    if (x.n == 0 || x.entry_flat == NO_POINT) { err = "entry point not initialized"; return ERR_EMPTY; }
    if (x.nb_layer != NB_LAYER_MAX) { err = "dump of Description, nb_layer != NB_MAX_LAYER"; return ERR_ARG; }  // src/hnswio.rs:893-896
    BufWriter g(dir + "/" + basename + ".hnsw.graph");
    BufWriter dt(dir + "/" + basename + ".hnsw.data");
    if (!g.ok || !dt.ok) { err = "could not create dump files in " + dir; return ERR_IO; }
    g.put<uint32_t>(MAGICDESCR_4);
    g.put<uint8_t>(1);
    g.put<uint8_t>((uint8_t)x.max_nb_connection);
    g.put<double>(x.level_scale);
    g.put<uint8_t>(x.nb_layer);
    g.put<uint64_t>(x.ef_construction);
    g.put<uint64_t>(x.n);
    g.put<uint64_t>(x.dimension);
    std::string dn = x.distname.empty() ? std::string(dist_type_name(x.dist)) : x.distname;
    g.put<uint64_t>(dn.size());
    g.bytes(dn.data(), dn.size());
    g.put<uint64_t>(x.t_name.size());
    g.bytes(x.t_name.data(), x.t_name.size());
    dt.put<uint32_t>(MAGICDATAP);
    dt.put<uint64_t>(x.dimension);

    // p_id of every flat id (needed for each edge)
    std::vector<uint8_t> lay(x.n);
    for (unsigned l = 0; l < NB_LAYER_MAX; ++l)
        for (uint64_t f = x.layer_offset[l]; f < x.layer_offset[l + 1]; ++f) lay[f] = (uint8_t)l;

    g.put<uint8_t>((uint8_t)NB_LAYER_MAX);  // points_by_layer.len() (src/hnswio.rs:1308-1309)
    const uint64_t d = x.dimension;
    for (unsigned l = 0; l < NB_LAYER_MAX; ++l) {
        g.put<uint32_t>(MAGICLAYER);
        g.put<uint64_t>(x.layer_count(l));
        for (uint64_t f = x.layer_offset[l]; f < x.layer_offset[l + 1]; ++f) {
            g.put<uint32_t>(MAGICPOINT);
            g.put<uint64_t>(x.origin_id[f]);
            g.put<uint8_t>((uint8_t)l);
            g.put<int32_t>((int32_t)(f - x.layer_offset[l]));
            for (unsigned ll = 0; ll < NB_LAYER_MAX; ++ll) {
                uint64_t b = x.nbr_ptr[f * NB_LAYER_MAX + ll], e = x.nbr_ptr[f * NB_LAYER_MAX + ll + 1];
                g.put<uint64_t>(e - b);
                for (uint64_t j = b; j < e; ++j) {
                    uint32_t nf = x.nbr_flat[j];
                    g.put<uint64_t>(x.origin_id[nf]);
                    g.put<uint8_t>(lay[nf]);
                    g.put<int32_t>((int32_t)(nf - x.layer_offset[lay[nf]]));
                    g.put<float>(x.nbr_dist[j]);
                }
            }
            dt.put<uint32_t>(MAGICDATAP);
            dt.put<uint64_t>(x.origin_id[f]);
            dt.put<uint64_t>(d * sizeof(float));
            dt.bytes(x.vectors.data() + f * d, d * sizeof(float));
        }
    }
    g.put<uint64_t>(x.origin_id[x.entry_flat]);
    g.put<uint8_t>(lay[x.entry_flat]);
    g.put<int32_t>((int32_t)(x.entry_flat - x.layer_offset[lay[x.entry_flat]]));
    bool ok1 = g.close(), ok2 = dt.close();
    if (!ok1 || !ok2) { err = "write error while dumping"; return ERR_IO; }
    return OK;
}

}  // namespace hnswgpu
