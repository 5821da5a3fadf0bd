// hnswio_oracle.hpp -- ORACLE-side restatement of the hnswio two-file dump format
// (src/hnswio.rs) on top of the pointer-web model of hnsw_oracle.hpp.
//
// TEST INFRASTRUCTURE ONLY (see hnsw_oracle.hpp header).  Independent of the product's
// reader/writer (hnswlib-rs_amd/csrc/hnswio.cpp): tests cross-check the two byte for byte.
//
// Format, native-endian, packed:
//   <base>.hnsw.graph : Description (src/hnswio.rs:878-919) + PointIndexation (:1303-1340)
//   <base>.hnsw.data  : header (:1382-1383) + per-point records (:1099-1112)
#pragma once
#include <fstream>
#include "hnsw_oracle.hpp"

namespace oracle {

constexpr uint32_t MAGICPOINT = 0x000a678f;    // src/hnswio.rs:47
constexpr uint32_t MAGICDESCR_2 = 0x002a677f;  // :49
constexpr uint32_t MAGICDESCR_3 = 0x002a6771;  // :56
constexpr uint32_t MAGICDESCR_4 = 0x002a6779;  // :60
constexpr uint32_t MAGICLAYER = 0x000a676f;    // :63
constexpr uint32_t MAGICDATAP = 0xa67f0000;    // :65

struct Description {  // src/hnswio.rs:846-867
    size_t format_version = 0;
    uint8_t dumpmode = 0;
    uint8_t max_nb_connection = 0;
    double level_scale = 1.0;
    uint8_t nb_layer = 0;
    size_t ef = 0;
    size_t nb_point = 0;
    size_t dimension = 0;
    std::string distname;
    std::string t_name;
};

template <class T>
inline void put(std::ostream& o, const T& v) { o.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <class T>
inline T get(std::istream& i) {
    T v;
    i.read(reinterpret_cast<char*>(&v), sizeof(T));
    if (!i) throw std::runtime_error("hnswio: unexpected end of file");
    return v;
}

// Hnsw::dump + Description::dump + PointIndexation::dump + dump_point
// (src/hnswio.rs:1355-1387, :878-919, :1303-1340, :1063-1115), DumpMode::Full.
inline void file_dump(const Hnsw& h, const std::string& dir, const std::string& basename) {
    std::ofstream g(dir + "/" + basename + ".hnsw.graph", std::ios::binary);
    std::ofstream dt(dir + "/" + basename + ".hnsw.data", std::ios::binary);
    if (!g || !dt) throw std::runtime_error("file_dump: cannot open output files");
    // Description
    put<uint32_t>(g, MAGICDESCR_4);
    put<uint8_t>(g, 1);                                  // Full
    put<uint8_t>(g, (uint8_t)h.max_nb_connection);       // as u8 (src/hnsw.rs:823-825)
    put<double>(g, h.layer_g.scale);                     // get_level_scale(): the absolute scale
    put<uint8_t>(g, (uint8_t)h.max_layer);
    if (h.max_layer != NB_LAYER_MAX) throw std::runtime_error("dump of Description, nb_layer != NB_MAX_LAYER");
    put<uint64_t>(g, h.ef_construction);
    put<uint64_t>(g, h.nb_point);
    size_t datadim = h.entry_point ? h.entry_point->v.size() : 0;  // get_data_dimension()
    put<uint64_t>(g, datadim);
    std::string dn = dist_type_name(h.dist);
    put<uint64_t>(g, dn.size());
    g.write(dn.data(), dn.size());
    std::string tn = "f32";
    put<uint64_t>(g, tn.size());
    g.write(tn.data(), tn.size());
    // data header
    put<uint32_t>(dt, MAGICDATAP);
    put<uint64_t>(dt, datadim);
    // PointIndexation
    put<uint8_t>(g, (uint8_t)h.points_by_layer.size());
    for (size_t i = 0; i < h.points_by_layer.size(); ++i) {
        put<uint32_t>(g, MAGICLAYER);
        put<uint64_t>(g, h.points_by_layer[i].size());
        for (size_t j = 0; j < h.points_by_layer[i].size(); ++j) {
            const Point& p = *h.points_by_layer[i][j];
            put<uint32_t>(g, MAGICPOINT);
            put<uint64_t>(g, p.origin_id);
            put<uint8_t>(g, p.p_id.layer);
            put<int32_t>(g, p.p_id.rank);
            for (size_t l = 0; l < NB_LAYER_MAX; ++l) {  // get_neighborhood_id: 16 lists
                put<uint64_t>(g, p.neighbours[l].size());
                for (const PWO& n : p.neighbours[l]) {
                    put<uint64_t>(g, n->point_ref->origin_id);
                    put<uint8_t>(g, n->point_ref->p_id.layer);
                    put<int32_t>(g, n->point_ref->p_id.rank);
                    put<float>(g, n->dist_to_ref);
                }
            }
            put<uint32_t>(dt, MAGICDATAP);
            put<uint64_t>(dt, (uint64_t)p.origin_id);
            put<uint64_t>(dt, (uint64_t)(p.v.size() * sizeof(float)));
            dt.write(reinterpret_cast<const char*>(p.v.data()), p.v.size() * sizeof(float));
        }
    }
    if (!h.entry_point) throw std::runtime_error("entry point not initialized");
    put<uint64_t>(g, h.entry_point->origin_id);
    put<uint8_t>(g, h.entry_point->p_id.layer);
    put<int32_t>(g, h.entry_point->p_id.rank);
    if (!g || !dt) throw std::runtime_error("file_dump: write error");
}

// load_description (src/hnswio.rs:937-1042)
inline Description load_description(std::istream& in) {
    Description d;
    uint32_t magic = get<uint32_t>(in);
    if (magic == MAGICDESCR_2) d.format_version = 2;
    else if (magic == MAGICDESCR_3) d.format_version = 3;
    else if (magic == MAGICDESCR_4) d.format_version = 4;
    else throw std::runtime_error("bad magic at descr beginning");
    d.dumpmode = get<uint8_t>(in);
    d.max_nb_connection = get<uint8_t>(in);
    if (d.format_version == 4) d.level_scale = get<double>(in);
    d.nb_layer = get<uint8_t>(in);
    d.ef = get<uint64_t>(in);
    d.nb_point = get<uint64_t>(in);
    d.dimension = get<uint64_t>(in);
    uint64_t len = get<uint64_t>(in);
    if (len > 256) throw std::runtime_error("bad length for distance name");
    d.distname.resize(len);
    in.read(&d.distname[0], len);
    len = get<uint64_t>(in);
    if (len > 256) throw std::runtime_error("bad lenght for T name");
    d.t_name.resize(len);
    in.read(&d.t_name[0], len);
    if (!in) throw std::runtime_error("hnswio: truncated description");
    return d;
}

inline std::string short_name(const std::string& s) {  // rsplit_terminator("::")[0]
    size_t p = s.rfind("::");
    return p == std::string::npos ? s : s.substr(p + 2);
}

// HnswIo::load_hnsw::<f32, D> (src/hnswio.rs:431-524, :615-784, :1221-1289, :1119-1178)
inline std::unique_ptr<Hnsw> load_hnsw(const std::string& dir, const std::string& basename, DistKind asked) {
    std::ifstream g(dir + "/" + basename + ".hnsw.graph", std::ios::binary);
    std::ifstream dt(dir + "/" + basename + ".hnsw.data", std::ios::binary);
    if (!g || !dt) throw std::runtime_error("could not reload HNSW structure");
    Description descr = load_description(g);
    if (get<uint32_t>(dt) != MAGICDATAP) throw std::runtime_error("magic not equal to MAGICDATAP in load_point");
    if (get<uint64_t>(dt) != descr.dimension) throw std::runtime_error("data dimension incoherent");
    if (short_name(dist_type_name(asked)) != short_name(descr.distname))
        throw std::runtime_error("error in distances : dumped distance is : " + descr.distname);
    if (descr.t_name != "f32") throw std::runtime_error("incohrent size of T in description");
    if (descr.format_version == 2) throw std::runtime_error("format v2 (bincode) not supported by the oracle");
    if (descr.dumpmode != 1) throw std::runtime_error("only DumpMode::Full can be reloaded");

    auto h = std::unique_ptr<Hnsw>(new Hnsw(descr.max_nb_connection, descr.nb_point, descr.nb_layer, descr.ef, asked));
    h->extend_candidates = true;  // :510
    h->keep_pruned = false;
    // load_point_indexation builds LayerGenerator::new_with_scale(M, descr.level_scale, NB_LAYER_MAX) (:773-777): the
    // dumped ABSOLUTE scale is used as a FACTOR of 1/ln(M) (src/hnsw.rs:339-352) -- a quirk, restated as it is: points
    // inserted after a reload draw their levels with scale = level_scale / ln(M).
    h->layer_g = LayerGenerator(descr.max_nb_connection, descr.level_scale, NB_LAYER_MAX);
    h->level_scale_factor = descr.level_scale;
    h->data_dimension = descr.dimension;
    uint8_t nb_layer = get<uint8_t>(g);
    if (nb_layer > NB_LAYER_MAX) throw std::runtime_error("inconsistent number of layErrers");
    h->points_by_layer.assign(NB_LAYER_MAX, {});
    struct RawN { size_t d_id; PointId p_id; float distance; };
    std::vector<std::vector<std::vector<std::vector<RawN>>>> raw(nb_layer);  // [layer][rank][l][j]
    size_t nb_loaded = 0;
    for (size_t l = 0; l < nb_layer; ++l) {
        if (get<uint32_t>(g) != MAGICLAYER) throw std::runtime_error("bad magic at layer beginning");
        size_t nbpoints = get<uint64_t>(g);
        raw[l].resize(nbpoints);
        for (size_t r = 0; r < nbpoints; ++r) {
            if (get<uint32_t>(g) != MAGICPOINT) throw std::runtime_error("bad magic at point beginning");
            size_t origin_id = get<uint64_t>(g);
            PointId p_id;
            p_id.layer = get<uint8_t>(g);
            p_id.rank = get<int32_t>(g);
            raw[l][r].resize(NB_LAYER_MAX);
            for (size_t ll = 0; ll < descr.nb_layer; ++ll) {
                size_t nbn = get<uint64_t>(g);
                raw[l][r][ll].reserve(nbn);
                for (size_t j = 0; j < nbn; ++j) {
                    RawN n;
                    n.d_id = get<uint64_t>(g);
                    n.p_id.layer = get<uint8_t>(g);
                    n.p_id.rank = get<int32_t>(g);
                    n.distance = get<float>(g);
                    raw[l][r][ll].push_back(n);
                }
            }
            if (get<uint32_t>(dt) != MAGICDATAP) throw std::runtime_error("magic not equal to MAGICDATAP in load_point");
            if (get<uint64_t>(dt) != origin_id) throw std::runtime_error("origin_id incoherent between graph and data");
            uint64_t slen = get<uint64_t>(dt);
            std::vector<char> buf(slen);
            dt.read(buf.data(), slen);
            if (!dt) throw std::runtime_error("hnswio: truncated data file");
            if (l != p_id.layer || (int64_t)r != p_id.rank) throw std::runtime_error("p_id incoherent with position in file");
            h->points_by_layer[l].push_back(
                std::make_shared<Point>(reinterpret_cast<const float*>(buf.data()), descr.dimension, origin_id, p_id));
            nb_loaded++;
        }
    }
    for (size_t l = 0; l < nb_layer; ++l)
        for (size_t r = 0; r < raw[l].size(); ++r) {
            Point& p = *h->points_by_layer[l][r];
            for (size_t ll = 0; ll < NB_LAYER_MAX; ++ll) {
                for (const RawN& n : raw[l][r][ll]) {
                    const auto& n_point = h->points_by_layer.at(n.p_id.layer).at(n.p_id.rank);
                    p.neighbours[ll].push_back(std::make_shared<PointWithOrder>(n_point, n.distance));
                }
                Hnsw::sort_unstable(p.neighbours[ll]);  // :731
            }
        }
    size_t ep_origin = get<uint64_t>(g);
    uint8_t ep_layer = get<uint8_t>(g);
    int32_t ep_rank = get<int32_t>(g);
    h->entry_point = h->points_by_layer.at(ep_layer).at(ep_rank);
    if (h->entry_point->origin_id != ep_origin) throw std::runtime_error("entry point origin id incoherent");
    h->nb_point = nb_loaded;
    return h;
}

}  // namespace oracle
