// flat_index.hpp -- host-side flat model of an hnsw_rs index.
//
// The reference keeps the graph as a web of Arc<Point> / Arc<PointWithOrder> behind per-point
// RwLocks (src/hnsw.rs:164-173, :265-271, :395-408).  Search only ever needs, per point: the
// vector, origin_id, p_id and the ids+order of each neighbour list.
// Here every point gets a dense id
//      flat = layer_offset[p_id.layer] + p_id.rank
// (the order points appear in a dump, src/hnswio.rs:1311-1320) and everything is an array
// indexed by flat id.  This is what gets replicated into HBM (device_index.hpp).
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace hnswgpu {

constexpr unsigned NB_LAYER_MAX = 16;  // src/hnsw.rs:42
constexpr uint32_t NO_POINT = 0xFFFFFFFFu;

enum Dist : int { DIST_L2 = 0, DIST_COSINE = 1, DIST_DOT = 2, DIST_L1 = 3, DIST_HELLINGER = 4, DIST_JEFFREYS = 5, DIST_JENSENSHANNON = 6 };
constexpr int DIST_COUNT = 7;

inline const char* dist_type_name(int d) {  // type_name::<D>() (src/hnsw.rs:839-841)
    switch (d) {
        case DIST_L2: return "anndists::dist::distances::DistL2";
        case DIST_COSINE: return "anndists::dist::distances::DistCosine";
        case DIST_DOT: return "anndists::dist::distances::DistDot";
        case DIST_L1: return "anndists::dist::distances::DistL1";
        case DIST_HELLINGER: return "anndists::dist::distances::DistHellinger";
        case DIST_JEFFREYS: return "anndists::dist::distances::DistJeffreys";
        case DIST_JENSENSHANNON: return "anndists::dist::distances::DistJensenShannon";
    }
    return "";
}
inline std::string short_name(const std::string& s) {  // rsplit_terminator("::")[0], src/hnswio.rs:474-478
    size_t p = s.rfind("::");
    return p == std::string::npos ? s : s.substr(p + 2);
}
inline int dist_from_short_name(const std::string& s) {
    if (s == "DistL2") return DIST_L2;
    if (s == "DistCosine") return DIST_COSINE;
    if (s == "DistDot") return DIST_DOT;
    if (s == "DistL1") return DIST_L1;
    if (s == "DistHellinger") return DIST_HELLINGER;
    if (s == "DistJeffreys") return DIST_JEFFREYS;
    if (s == "DistJensenShannon") return DIST_JENSENSHANNON;
    return -1;
}

struct FlatIndex {
    // --- Description (src/hnswio.rs:846-867)
    uint32_t format_version = 4;
    uint8_t dumpmode = 1;
    uint64_t max_nb_connection = 0;
    double level_scale = 1.0;  // absolute scale of the level law (get_level_scale())
    uint8_t nb_layer = NB_LAYER_MAX;  // Hnsw::max_layer
    uint64_t ef_construction = 0;
    uint64_t dimension = 0;
    std::string distname;
    std::string t_name = "f32";
    int dist = DIST_L2;
    bool extend_candidates = false, keep_pruned = false;  // builder flags (not dumped)

    // --- points, in dump order
    uint64_t n = 0;
    std::array<uint64_t, NB_LAYER_MAX + 1> layer_offset{};  // prefix sums of per-layer counts
    std::vector<uint64_t> origin_id;                         // [n]
    std::vector<float> vectors;                              // [n * dimension], row-major
    // --- neighbour lists: list (flat, l) = nbr_*[nbr_ptr[flat*16+l] .. nbr_ptr[flat*16+l+1])
    std::vector<uint64_t> nbr_ptr;    // [n*16 + 1]
    std::vector<uint32_t> nbr_flat;   // flat id of the neighbour
    std::vector<float> nbr_dist;      // stored edge distance (not read by search)
    // --- entry point
    uint32_t entry_flat = NO_POINT;

    uint64_t layer_count(unsigned l) const { return layer_offset[l + 1] - layer_offset[l]; }
    unsigned layer_of(uint32_t flat) const {
        unsigned l = 0;
        while (l + 1 < NB_LAYER_MAX && flat >= layer_offset[l + 1]) ++l;
        return l;
    }
    int32_t rank_of(uint32_t flat) const { return (int32_t)(flat - layer_offset[layer_of(flat)]); }
    // lowest non-empty layer (src/hnsw.rs:1534-1540)
    unsigned layer_to_search() const {
        unsigned l = 0;
        while (l < NB_LAYER_MAX && layer_count(l) == 0) ++l;
        return l;
    }
};

}  // namespace hnswgpu
