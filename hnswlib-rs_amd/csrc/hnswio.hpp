// hnswio.hpp -- hnsw_rs dump format <-> FlatIndex (see hnswio.cpp).
#pragma once
#include <string>
#include "flat_index.hpp"

namespace hnswgpu {

// numeric values match HNSWGPU_* in include/hnsw_mi355x.h
enum Status : int { OK = 0, ERR_ARG = 1, ERR_IO = 2, ERR_FORMAT = 3, ERR_DISTANCE = 4, ERR_TYPE = 5, ERR_DEVICE = 6, ERR_EMPTY = 7 };

struct DumpDescription {  // src/hnswio.rs:846-867
    uint32_t format_version = 0;
    uint8_t dumpmode = 0;
    uint8_t max_nb_connection = 0;
    double level_scale = 1.0;
    uint8_t nb_layer = 0;
    uint64_t ef = 0;
    uint64_t nb_point = 0;
    uint64_t dimension = 0;
    std::string distname;
    std::string t_name;
};

// load_description on the graph file (src/hnswio.rs:937-1042)
int load_description_file(const std::string& graph_path, DumpDescription& d, std::string& err);
// HnswIo::load_hnsw::<f32, D> (src/hnswio.rs:431-524).  asked_dist < 0: accept the dump's distance.
int load_dump(const std::string& dir, const std::string& basename, int asked_dist, FlatIndex& out, std::string& err);
// Hnsw::dump in DumpMode::Full (src/hnswio.rs:1355-1387)
int write_dump(const FlatIndex& x, const std::string& dir, const std::string& basename, std::string& err);

}  // namespace hnswgpu
