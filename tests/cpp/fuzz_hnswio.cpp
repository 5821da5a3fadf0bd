// Driver of tests/test_hnswio.py::test_mutated_dumps_never_crash_the_reader (built with -fsanitize=address,undefined): loads
// <dir>/<base>.hnsw.{graph,data} for every basename on stdin; a dump that still loads is written out again.
#include <cstdio>
#include <iostream>
#include <string>
#include "hnswio.hpp"
#include "flat_index.hpp"
#include "datamap.hpp"
using namespace hnswgpu;
int main(int argc, char** argv) {
    std::string dir = argv[1], base;
    long ok = 0, bad = 0, maps = 0;
    volatile double sink = 0;
    while (std::getline(std::cin, base)) {
        FlatIndex f; std::string err;
        int rc = load_dump(dir, base, -1, f, err);
        if (rc == 0) {
            ++ok;
            // touch what a consumer would: re-dump must not crash either
            std::string e2; (void)write_dump(f, dir, base + "_rt", e2);
        } else ++bad;
        DumpDescription d; std::string e3; (void)load_description_file(dir + "/" + base + ".hnsw.graph", d, e3);
        {   // DataMap: whatever opens must serve every id it lists, within the mapping (a sum over the vector touches the bytes)
            DataMap dm; std::string e4;
            if (dm.open(dir, base, e4) == 0) {
                ++maps;
                for (uint64_t id : dm.ids_in_file_order()) {
                    const float* v = dm.get_data(id);
                    if (v) for (uint64_t j = 0; j < dm.dimension(); ++j) sink += v[j];
                }
                (void)dm.get_data(0xFFFFFFFFFFFFull);
            }
        }
    }
    std::printf("loaded %ld, rejected %ld, datamaps opened %ld\n", ok, bad, maps);
    return 0;
}
