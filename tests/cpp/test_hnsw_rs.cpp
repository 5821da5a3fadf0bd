// C++ twin of the reference's own tests, written against include/hnsw_rs.hpp (the C++ mirror of the crate's
// interface over the C ABI):
//   reload_with_dist / load/dump round trip     src/hnswio.rs:1413-1459
//   test_sparse_search                          src/hnsw.rs:1870-1881      (gpu mode)
//   self retrieval, serial == parallel          tests/equality.rs:83-160, src/hnsw.rs:1601-1620   (gpu mode)
// usage: test_hnsw_rs <cpu|gpu> <tmpdir>
#include <cmath>
#include <cstring>
#include <cassert>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>

#include "hnsw_rs.hpp"

using hnsw_rs::Hnsw; using hnsw_rs::Error; using hnsw_rs::DistL1; using hnsw_rs::DistL2;

static std::vector<char> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const bool gpu = std::strcmp(argv[1], "gpu") == 0;
    const std::string dir = argv[2];
    const size_t nb_elem = 1000, dim = 10;
    std::mt19937 rng(42);
    std::uniform_real_distribution<float> unif(0.f, 1.f);
    std::vector<float> data(nb_elem * dim);
    for (auto& x : data) x = unif(rng);

    // let hnsw = Hnsw::<f32, DistL1>::new(10, nb_elem, 16, 25, DistL1{}); insert; file_dump; reload; compare
    Hnsw<float, DistL1> hnsw(10, nb_elem, 16, 25, DistL1{});
    hnsw.parallel_insert(data, dim, {}, 1);
    REQUIRE(hnsw.get_nb_point() == nb_elem);
    REQUIRE(hnsw.file_dump(dir, "dumpreloadtest") == "dumpreloadtest");
    hnsw_rs::HnswIo reloader(dir, "dumpreloadtest");
    auto loaded = reloader.load_hnsw<float, DistL1>();
    REQUIRE(loaded.get_nb_point() == nb_elem);
    REQUIRE(loaded.get_max_level_observed() == hnsw.get_max_level_observed());
    loaded.file_dump(dir, "dumpreloadtest2");
    {   // check_graph_equality: every byte but the level scale (offset 6, f64), which a RELOADED index dumps divided
        // by ln(M) -- the reference's reload rule (src/hnswio.rs:773-777)
        std::vector<char> g1 = slurp(dir + "/dumpreloadtest.hnsw.graph"), g2 = slurp(dir + "/dumpreloadtest2.hnsw.graph");
        REQUIRE(g1.size() == g2.size() && g1.size() > 14);
        double s1, s2;
        std::memcpy(&s1, g1.data() + 6, 8);
        std::memcpy(&s2, g2.data() + 6, 8);
        REQUIRE(s2 == s1 / std::log(10.0));
        std::memcpy(&g2[6], g1.data() + 6, 8);
        REQUIRE(g1 == g2);
    }
    REQUIRE(slurp(dir + "/dumpreloadtest.hnsw.data") == slurp(dir + "/dumpreloadtest2.hnsw.data"));
    // a dump written for DistL1 cannot be reloaded as DistL2 (src/hnswio.rs:473-490)
    bool refused = false;
    try { (void)reloader.load_hnsw<float, DistL2>(); } catch (const Error& e) { refused = e.code == HNSWGPU_ERR_DISTANCE; }
    REQUIRE(refused);
    // an empty index cannot be dumped and answers nothing
    Hnsw<float, DistL2> empty(8, 10, 16, 20);
    bool threw = false;
    try { empty.file_dump(dir, "empty"); } catch (const Error&) { threw = true; }
    REQUIRE(threw);
    REQUIRE(empty.search(std::vector<float>(4, 0.f), 2, 10).empty());
    if (!gpu) { std::printf("cpu mode OK\n"); return 0; }

    // ---- searches (MI355X)
    loaded.upload(0);
    std::vector<std::vector<float>> queries;
    for (size_t i = 0; i < 200; ++i) queries.emplace_back(data.begin() + i * dim, data.begin() + (i + 1) * dim);
    auto answers = loaded.parallel_search_neighbours(queries, 5, 40);
    REQUIRE(answers.size() == 200);
    size_t self_found = 0;
    for (size_t i = 0; i < 200; ++i) {
        REQUIRE(answers[i].size() == 5);
        for (size_t j = 1; j < 5; ++j) REQUIRE(answers[i][j - 1].distance <= answers[i][j].distance);
        if (answers[i][0].d_id == i) { ++self_found; REQUIRE(answers[i][0].distance == 0.f); }   // src/hnswio.rs:1639-1640
        auto one = loaded.search_neighbours(queries[i], 5, 40);                                   // serial == parallel
        REQUIRE(one.size() == 5);
        for (size_t j = 0; j < 5; ++j) REQUIRE(one[j].d_id == answers[i][j].d_id && one[j].distance == answers[i][j].distance && one[j].p_id == answers[i][j].p_id);
    }
    REQUIRE(self_found >= 190);
    // test_sparse_search: one point, k=2, ef=10 -> exactly one neighbour at distance 0
    for (int t = 0; t < 20; ++t) {
        Hnsw<float, DistL2> h1(3 + t % 5, 10, 16, 20);
        std::vector<float> p{unif(rng), unif(rng), unif(rng), unif(rng)};
        h1.parallel_insert(p, 4, {}, 1);
        auto r = h1.search(p, 2, 10);
        REQUIRE(r.size() == 1 && r[0].distance == 0.f && r[0].d_id == 0);
    }
    std::printf("gpu mode OK\n");
    return 0;
}
