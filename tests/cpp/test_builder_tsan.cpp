// ThreadSanitizer stress of the host builder (csrc/builder.cpp): parallel_insert with many threads -- searches read neighbour
// lists without a lock (EdgeList: a publish-once buffer behind an atomic pointer) while other threads rewrite them -- then a
// second batch on top, then the flattening.  Built with -fsanitize=thread by tests/test_cpp_mirror.py; any report fails the test.
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "builder.hpp"
#include "flat_index.hpp"

using namespace hnswgpu;

int main() {
    const uint64_t n = 6000, d = 12;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::vector<float> x(n * d);
    for (auto& v : x) v = u(rng);
    for (uint64_t i = 0; i < 300; ++i)  // duplicated points: equal distances on the lists being rewritten
        for (uint64_t j = 0; j < d; ++j) x[(n - 1 - i) * d + j] = x[i * d + j];
    BuildParams p;
    p.max_nb_connection = 8;
    p.ef_construction = 40;
    p.dist = DIST_L2;
    GraphBuilder b(p);
    std::string err;
    if (b.insert_batch(x.data(), n / 2, d, nullptr, 8, err) != 0) { std::printf("first batch: %s\n", err.c_str()); return 1; }
    if (b.insert_batch(x.data() + (n / 2) * d, n - n / 2, d, nullptr, 8, err) != 0) { std::printf("second batch: %s\n", err.c_str()); return 1; }
    FlatIndex f;
    b.finalize(f);
    if (f.n != n) { std::printf("nb_point %llu != %llu\n", (unsigned long long)f.n, (unsigned long long)n); return 1; }
    std::printf("builder under tsan OK\n");
    return 0;
}
