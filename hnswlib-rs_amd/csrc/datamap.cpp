// datamap.cpp -- see datamap.hpp.  Data file layout (src/hnswio.rs:1099-1112, :1382-1383):
//   {MAGICDATAP u32, dimension u64} then per point {MAGICDATAP u32, origin_id u64, byte_len u64, raw f32[dimension]}
#include "datamap.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>

#include "hnswio.hpp"

namespace hnswgpu {

static constexpr uint32_t MAGICDATAP = 0xa67f0000;  // src/hnswio.rs:65

DataMap::~DataMap() {
    if (map_) munmap(const_cast<uint8_t*>(map_), size_);
    if (fd_ >= 0) ::close(fd_);
}

int DataMap::open(const std::string& dir, const std::string& basename, std::string& err) {
    DumpDescription d;
    int rc = load_description_file(dir + "/" + basename + ".hnsw.graph", d, err);  // the reference exits the process here (:54-57)
    if (rc != OK) return rc;
    if (d.format_version <= 2) { err = "data mapping is only possible for dumps with the version > 0.1.19 of this crate"; return ERR_FORMAT; }
    if (d.t_name != "f32") { err = "type error: description has typename " + d.t_name + ", this library maps f32"; return ERR_TYPE; }
    distname_ = d.distname;
    t_name_ = d.t_name;
    dimension_ = d.dimension;
    const std::string dpath = dir + "/" + basename + ".hnsw.data";
    fd_ = ::open(dpath.c_str(), O_RDONLY);
    if (fd_ < 0) { err = "could not open file : " + dpath; return ERR_IO; }
    struct stat st;
    if (fstat(fd_, &st) != 0) { err = "could not stat file : " + dpath; return ERR_IO; }
    size_ = (size_t)st.st_size;
    if (size_ < 12) { err = "truncated data file"; return ERR_FORMAT; }
    void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) { err = "could not memory map : " + dpath; return ERR_IO; }
    map_ = static_cast<const uint8_t*>(m);
    size_t at = 0;
    uint32_t magic;
    std::memcpy(&magic, map_ + at, 4);
    at += 4;
    if (magic != MAGICDATAP) { err = "magic not equal to MAGICDATAP in mmap"; return ERR_FORMAT; }
    uint64_t dim;
    std::memcpy(&dim, map_ + at, 8);
    at += 8;
    if (dim != dimension_) { err = "description and data do not agree on dimension"; return ERR_FORMAT; }
    if (dim == 0 || dim > (size_ / sizeof(float))) { err = "data dimension incoherent with the data file size"; return ERR_FORMAT; }
    // every record: MAGICDATAP, DataId, byte length, dimension * 4 bytes (:156-160)
    const size_t record = 4 + 8 + 8 + (size_t)dim * sizeof(float);
    const size_t nb_record = (size_ - at) / record;
    addr_.reserve(nb_record);
    order_.reserve(nb_record);
    for (size_t i = 0; i < nb_record; ++i) {
        if (size_ - at < 20) { err = "truncated data file"; return ERR_FORMAT; }
        std::memcpy(&magic, map_ + at, 4);
        at += 4;
        if (magic != MAGICDATAP) { err = "magic not equal to MAGICDATAP in mmap"; return ERR_FORMAT; }
        uint64_t id, len;
        std::memcpy(&id, map_ + at, 8);
        at += 8;
        const size_t here = at;  // where the byte length sits: what the reference keeps in its map (:183-185)
        std::memcpy(&len, map_ + at, 8);
        at += 8;
        if (len > size_ - at || len / sizeof(float) < dim) { err = "truncated data file"; return ERR_FORMAT; }
        at += (size_t)len;
        if (addr_.emplace(id, here).second) order_.push_back(id);  // IndexMap::insert: a repeated id keeps its first rank,
        else addr_[id] = here;                                      // and takes the last address
    }
    return OK;
}

const float* DataMap::get_data(uint64_t data_id) const {
    auto it = addr_.find(data_id);
    if (it == addr_.end()) return nullptr;
    return reinterpret_cast<const float*>(map_ + it->second + 8);
}

}  // namespace hnswgpu
