// datamap.hpp -- DataMap (src/datamap.rs:24-319): the vectors of a dump, memory-mapped and addressed by DataId, without
// loading the graph.  Row f2 of the scope table ("DataMap-style lazy host access"): a host convenience next to the path,
// so that an index built or served by this library can be consumed like the crate's.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace hnswgpu {

class DataMap {
public:
    DataMap() = default;
    ~DataMap();
    DataMap(const DataMap&) = delete;
    DataMap& operator=(const DataMap&) = delete;
    // DataMap::from_hnswdump::<f32>(dir, file_name) (src/datamap.rs:44-231)
    int open(const std::string& dir, const std::string& basename, std::string& err);
    // get_data::<f32>(dataid) (:276-297): pointer to `dimension` floats inside the mapping, nullptr for an unknown id
    const float* get_data(uint64_t data_id) const;
    uint64_t nb_data() const { return order_.size(); }                 // get_nb_data (:316)
    uint64_t dimension() const { return dimension_; }
    const std::string& distname() const { return distname_; }          // get_distname (:311)
    const std::string& type_name() const { return t_name_; }           // get_data_typename (:306)
    const std::vector<uint64_t>& ids_in_file_order() const { return order_; }  // get_dataid_iter (:301)

private:
    const uint8_t* map_ = nullptr;
    size_t size_ = 0;
    int fd_ = -1;
    uint64_t dimension_ = 0;
    std::string distname_, t_name_;
    std::unordered_map<uint64_t, size_t> addr_;  // DataId -> offset of the record's byte length (hmap, :171-197)
    std::vector<uint64_t> order_;
};

}  // namespace hnswgpu
