/* A plain C99 caller of libhnsw_mi355x.so, the way a C / Julia user of the crate's libext.rs FFI would use it:
 * build an index through the reference-style symbols, dump it, reload it through get_hnswio / load_hnswdump_f32_DistL2,
 * read the description back.  No search here (that needs the GPU; tests/test_gpu_parity.py does it through the same
 * symbols).  Exit code 0 = everything as expected. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "hnsw_mi355x.h"

_Static_assert(sizeof(Neighbour_api) == 16, "Neighbour_api {usize id; f32 d} (src/libext.rs:64-71)");
_Static_assert(sizeof(Neighbourhood_api) == 16, "Neighbourhood_api {i64 nbgh; ptr} (src/libext.rs:82-87)");
_Static_assert(sizeof(DescriptionFFI) == 64, "DescriptionFFI (src/libext.rs:1121-1141)");

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    if (chdir(argv[1]) != 0) return 3; /* the reference's get_hnswio always uses directory "." */
    const size_t n = 500, d = 8;
    float* data = (float*)malloc(n * d * sizeof(float));
    const float** rows = (const float**)malloc(n * sizeof(float*));
    size_t* ids = (size_t*)malloc(n * sizeof(size_t));
    unsigned s = 12345u;
    for (size_t i = 0; i < n * d; ++i) { s = s * 1664525u + 1013904223u; data[i] = (float)(s >> 8) / 16777216.0f; }
    for (size_t i = 0; i < n; ++i) { rows[i] = data + i * d; ids[i] = 1000 + i; }
    const char* dist = "DistL2";
    const HnswApif32* h = new_hnsw_f32(16, 100, strlen(dist), (const uint8_t*)dist, n, 16);
    if (!h) return 10;
    parallel_insert_f32((HnswApif32*)h, n, d, rows, ids);
    const char* base = "c_caller";
    if (file_dump_f32(h, strlen(base), (const uint8_t*)base) != 1) return 11;
    drop_hnsw_f32(h);

    const HnswIo* io = get_hnswio(strlen(base), (const uint8_t*)base);
    if (!io) return 12;
    const HnswApif32* h2 = load_hnswdump_f32_DistL2((HnswIo*)io);
    if (!h2) return 13;
    hnswgpu_index* idx = hnswgpu_from_api(h2);
    if (!idx || hnswgpu_nb_point(idx) != n || hnswgpu_dimension(idx) != d) return 14;
    const HnswApif32* wrong = load_hnswdump_f32_DistCosine((HnswIo*)io); /* distance mismatch => NULL (:298-301) */
    if (wrong) return 15;

    const char* graph = "c_caller.hnsw.graph";
    const DescriptionFFI* desc = load_hnsw_description(strlen(graph), (const uint8_t*)graph);
    if (!desc) return 16;
    /* like the reference (src/libext.rs:1198-1206), load_hnsw_description reports dumpmode 1 and leaves nb_point 0 */
    if (desc->data_dimension != d || desc->max_nb_connection != 16 || desc->nb_layer != 16 || desc->ef != 100) return 17;
    if (desc->distname_len == 0 || memcmp(desc->distname + desc->distname_len - 6, "DistL2", 6) != 0) return 18;
    if (desc->t_name_len != 3 || memcmp(desc->t_name, "f32", 3) != 0) return 19;
    hnswgpu_free_description(desc);
    drop_hnsw_f32(h2);
    hnswgpu_free_hnswio(io);
    free(data); free(rows); free(ids);
    printf("ok\n");
    return 0;
}
