// The HOST side of GPU-assisted construction (GraphBuilder::insert_batch_gpu, csrc/builder.cpp) against a mock of the device
// side: HostMockBackend keeps the frozen snapshot of the lists exactly as the device does (patch records in, window searches
// out) and searches it on the CPU with the builder's own search_layer semantics.  No GPU, no HIP: what is exercised is the
// window protocol -- bootstrap, snapshot patching, the layers above the frozen entry point, dirty-list bookkeeping, the
// fall-back to the host builder when the backend fails half way -- and, built with -fsanitize=thread, the parallel section
// that applies a window (lock-free list reads, reverse updates, the worker pool).
//   1. window = 1 reproduces the serial insertion: dumps byte-identical;
//   2. growing windows on several threads: a well-formed graph of the right size that finds its own points;
//   3. a backend that fails at its third window: the call succeeds, says so (last_warning) and every point is linked;
//   4. a backend that cannot serve the build at all (check fails): error, index unchanged.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "builder.hpp"
#include "flat_index.hpp"
#include "hnswio.hpp"

using namespace hnswgpu;

namespace {

struct EdgeLessT {
    bool operator()(const Edge& a, const Edge& b) const { return a.dist < b.dist; }
};
struct EdgeGreaterT {
    bool operator()(const Edge& a, const Edge& b) const { return a.dist > b.dist; }
};

class HostMockBackend : public BuildSearchBackend {
public:
    bool do_select = false;    // run select_neighbours here as well when the builder asks for it (WindowSelect::on_device)
    int fail_at_window = -1;   // search_window number (0-based) that reports a device failure
    bool refuse = false;       // check() refuses
    int windows = 0;
    uint64_t records_patched = 0;

    int check(uint64_t, std::string& err) override {
        if (refuse) { err = "mock: no device"; return ERR_DEVICE; }
        return OK;
    }
    int begin(const float* const* chunks, uint64_t chunk_rows, uint64_t n, uint64_t d, const uint8_t* levels, int dist,
              uint64_t max_nb_connection, uint64_t ef_construction, unsigned, uint64_t, std::string& err) override {
        if (dist != DIST_L2) { err = "mock: L2 only"; return ERR_ARG; }
        chunks_.assign(chunks, chunks + (n + chunk_rows - 1) / chunk_rows);
        chunk_rows_ = chunk_rows; n_ = n; d_ = d; m_ = max_nb_connection; efc_ = ef_construction;
        levels_.assign(levels, levels + n);
        for (auto& l : lists_) l.assign(n, {});
        stamp_.assign(n, 0u);
        return OK;
    }
    uint32_t rec_words() const override { return 2u + 2u * (uint32_t)m_; }
    uint32_t* patch_buffer(uint64_t n_records, std::string&) override {
        buf_.resize(std::max<uint64_t>(1, n_records) * rec_words());
        return buf_.data();
    }
    int patch(uint64_t n_records, std::string&) override {
        const uint32_t rw = rec_words();
        for (uint64_t k = 0; k < n_records; ++k) {
            const uint32_t* r = buf_.data() + k * rw;
            auto& lst = lists_[r[1]][r[0]];
            lst.clear();
            for (uint32_t j = 2; j < rw && r[j] != NO_POINT; ++j) lst.push_back(r[j]);
        }
        records_patched += n_records;
        return OK;
    }
    int search_window(uint32_t first, uint32_t count, uint32_t entry, uint32_t entry_level, uint32_t layer_mask, const WindowSelect& wsel,
                      WindowSearchResults& out, std::string& err) override {
        if (windows++ == fail_at_window) { err = "mock: device lost"; return ERR_DEVICE; }
        out = WindowSearchResults();
        out.selected = do_select && wsel.on_device;  // else select_neighbours stays with the host
        out.slot0.resize(count);
        out.hit_ids.assign((size_t)count * NB_LAYER_MAX, NO_POINT);
        out.hit_d.assign((size_t)count * NB_LAYER_MAX, 0.f);
        uint32_t slots = 0;
        for (uint32_t wi = 0; wi < count; ++wi) {
            out.slot0[wi] = slots;
            slots += std::min<uint32_t>(levels_[first + wi], entry_level) + 1u;
        }
        if (out.selected) {
            out.sel_stride = std::max(wsel.nb_layer0, wsel.nb_upper);
            out.sel_ids.assign((size_t)slots * out.sel_stride, NO_POINT);
            out.sel_d.assign((size_t)slots * out.sel_stride, 0.f);
            out.sel_n.assign(slots, 0u);
        } else {
            out.out_ids.assign((size_t)slots * efc_, NO_POINT);
            out.out_d.assign((size_t)slots * efc_, 0.f);
            out.out_n.assign(slots, 0u);
        }
        std::vector<Edge> res, sel;
        for (uint32_t wi = 0; wi < count; ++wi) {
            const uint32_t id = first + wi;
            const unsigned level = levels_[id];
            const float* q = vec(id);
            uint32_t enter = entry;
            float dist_to_entry = l2(q, vec(entry));
            for (int l = (int)entry_level; l >= (int)level + 1; --l) {  // src/hnsw.rs:1114-1155
                search(q, enter, 1, (unsigned)l, ((layer_mask >> l) & 1u) != 0u, res);
                if (res.empty()) continue;
                out.hit_ids[(size_t)wi * NB_LAYER_MAX + (unsigned)l] = res[0].id;
                out.hit_d[(size_t)wi * NB_LAYER_MAX + (unsigned)l] = res[0].dist;
                if (res[0].dist < dist_to_entry) { enter = res[0].id; dist_to_entry = res[0].dist; }
            }
            for (int l = (int)std::min<unsigned>(level, entry_level); l >= 0; --l) {  // :1158-1205
                // (a layer counts as populated for this point when it is its own level: generate_new_point pushed it first)
                search(q, enter, efc_, (unsigned)l, ((layer_mask >> l) & 1u) != 0u || (unsigned)l == level, res);
                const size_t slot = (size_t)out.slot0[wi] + (size_t)l;
                if (out.selected) {  // select_neighbours (src/hnsw.rs:1299-1421; no extension, no kept pruned entries here)
                    const size_t nb = l == 0 ? wsel.nb_layer0 : wsel.nb_upper;
                    sel.clear();
                    if (res.size() <= nb) {
                        sel = res;
                    } else {
                        for (size_t i = 0; i < res.size() && sel.size() < nb; ++i) {
                            bool keep = true;
                            for (const Edge& s : sel)
                                if (l2(vec(res[i].id), vec(s.id)) <= res[i].dist) { keep = false; break; }
                            if (keep) sel.push_back(res[i]);
                        }
                    }
                    out.sel_n[slot] = (uint32_t)sel.size();
                    for (size_t j = 0; j < sel.size(); ++j) {
                        out.sel_ids[slot * out.sel_stride + j] = sel[j].id;
                        out.sel_d[slot * out.sel_stride + j] = sel[j].dist;
                    }
                } else {
                    out.out_n[slot] = (uint32_t)res.size();
                    for (size_t j = 0; j < res.size(); ++j) {
                        out.out_ids[slot * efc_ + j] = res[j].id;
                        out.out_d[slot * efc_ + j] = res[j].dist;
                    }
                }
                if (!res.empty()) enter = res[0].id;  // select_neighbours keeps the nearest candidate first
            }
        }
        return OK;
    }

private:
    const float* vec(uint32_t id) const { return chunks_[id / chunk_rows_] + (uint64_t)(id % chunk_rows_) * d_; }
    float l2(const float* a, const float* b) const {
        float norm = 0.f;
        for (uint64_t i = 0; i < d_; ++i) {
            const float t = a[i] - b[i];
            norm = norm + t * t;
        }
        return std::sqrt(norm);
    }
    bool visit(uint32_t id) {
        if (stamp_[id] == epoch_) return false;
        stamp_[id] = epoch_;
        return true;
    }
    // GraphBuilder::search_layer on the snapshot's lists
    void search(const float* q, uint32_t entry, size_t ef, unsigned layer, bool populated, std::vector<Edge>& out_sorted) {
        out_sorted.clear();
        if (!populated) return;
        std::vector<Edge> C, R;
        ++epoch_;
        const float d0 = l2(q, vec(entry));
        visit(entry);
        C.push_back({entry, d0});
        R.push_back({entry, d0});
        while (!C.empty()) {
            std::pop_heap(C.begin(), C.end(), EdgeGreaterT());
            const Edge c = C.back();
            C.pop_back();
            if (c.dist > R.front().dist) break;
            for (uint32_t e : lists_[layer][c.id]) {
                if (!visit(e)) continue;
                const float de = l2(q, vec(e));
                if (de < R.front().dist || R.size() < ef) {
                    C.push_back({e, de});
                    std::push_heap(C.begin(), C.end(), EdgeGreaterT());
                    R.push_back({e, de});
                    std::push_heap(R.begin(), R.end(), EdgeLessT());
                    if (R.size() > ef) {
                        std::pop_heap(R.begin(), R.end(), EdgeLessT());
                        R.pop_back();
                    }
                }
            }
        }
        out_sorted.assign(R.begin(), R.end());
        std::sort(out_sorted.begin(), out_sorted.end(), EdgeLessT());
    }

    std::vector<const float*> chunks_;
    uint64_t chunk_rows_ = 1, n_ = 0, d_ = 0, m_ = 0, efc_ = 0;
    std::vector<uint8_t> levels_;
    std::vector<std::vector<uint32_t>> lists_[NB_LAYER_MAX];
    std::vector<uint32_t> stamp_;
    uint32_t epoch_ = 0;
    std::vector<uint32_t> buf_;
};

std::vector<float> data_set(uint64_t n, uint64_t d, unsigned seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::vector<float> x(n * d);
    for (auto& v : x) v = u(rng);
    return x;
}
std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
BuildParams params() {
    BuildParams p;
    p.max_nb_connection = 8;
    p.ef_construction = 40;
    p.dist = DIST_L2;
    return p;
}
int well_formed(const FlatIndex& f, uint64_t n, const char* what) {
    if (f.n != n) { std::printf("%s: %llu points, expected %llu\n", what, (unsigned long long)f.n, (unsigned long long)n); return 1; }
    uint64_t empty0 = 0;
    for (uint64_t i = 0; i < f.n; ++i) {
        const uint64_t b = f.nbr_ptr[i * NB_LAYER_MAX + 0], e = f.nbr_ptr[i * NB_LAYER_MAX + 1];  // the layer-0 list
        for (uint64_t j = b; j < e; ++j)
            if (f.nbr_flat[j] >= f.n || f.nbr_flat[j] == i) { std::printf("%s: point %llu has neighbour %u\n", what, (unsigned long long)i, f.nbr_flat[j]); return 1; }
        if (e == b) ++empty0;
    }
    if (empty0 > 1) { std::printf("%s: %llu points without a layer-0 neighbour\n", what, (unsigned long long)empty0); return 1; }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const uint64_t d = 10;
    std::string err;
    // ---- 1. window = 1 is the serial insertion
    for (int pass = 0; pass < 2; ++pass) {
        const uint64_t n = 1500;
        const std::vector<float> x = data_set(n, d, 11);
        GraphBuilder serial(params()), windowed(params());
        if (serial.insert_batch(x.data(), n, d, nullptr, 1, err) != OK) { std::printf("serial: %s\n", err.c_str()); return 1; }
        HostMockBackend dev;
        dev.do_select = pass == 1;  // second pass: select_neighbours on the "device" too
        if (windowed.insert_batch_gpu(x.data(), n, d, nullptr, 1, dev, 1, err) != OK) { std::printf("window 1: %s\n", err.c_str()); return 1; }
        if (!windowed.last_warning().empty()) { std::printf("window 1: unexpected warning %s\n", windowed.last_warning().c_str()); return 1; }
        FlatIndex a, b;
        serial.finalize(a);
        windowed.finalize(b);
        if (write_dump(a, dir, "serial", err) != OK || write_dump(b, dir, "window1", err) != OK) { std::printf("dump: %s\n", err.c_str()); return 1; }
        if (slurp(dir + "/serial.hnsw.graph") != slurp(dir + "/window1.hnsw.graph") || slurp(dir + "/serial.hnsw.data") != slurp(dir + "/window1.hnsw.data")) {
            std::printf("window 1: the dump differs from the serial insertion's\n");
            return 1;
        }
        if (dev.windows != (int)n - 1) { std::printf("window 1: %d windows for %llu points\n", dev.windows, (unsigned long long)n); return 1; }
    }
    // ---- 2. growing windows, several host threads; then a second batch on top of the first
    {
        const uint64_t n = 9000;
        const std::vector<float> x = data_set(n, d, 12);
        GraphBuilder b(params());
        HostMockBackend dev;
        if (b.insert_batch_gpu(x.data(), 6000, d, nullptr, 8, dev, 0, err) != OK) { std::printf("windows: %s\n", err.c_str()); return 1; }
        HostMockBackend dev2;
        dev2.do_select = true;
        if (b.insert_batch_gpu(x.data() + 6000 * d, n - 6000, d, nullptr, 8, dev2, 512, err) != OK) { std::printf("windows, second batch: %s\n", err.c_str()); return 1; }
        FlatIndex f;
        b.finalize(f);
        if (well_formed(f, n, "windows")) return 1;
        if (dev.windows < 5 || dev2.windows < 5 || dev2.records_patched == 0) { std::printf("windows: %d + %d windows\n", dev.windows, dev2.windows); return 1; }
    }
    // ---- 3. the backend fails at its third window: the host builder finishes the batch
    {
        const uint64_t n = 5000;
        const std::vector<float> x = data_set(n, d, 13);
        GraphBuilder b(params());
        HostMockBackend dev;
        dev.fail_at_window = 2;
        if (b.insert_batch_gpu(x.data(), n, d, nullptr, 4, dev, 0, err) != OK) { std::printf("failing backend: %s\n", err.c_str()); return 1; }
        if (b.last_warning().find("fell back to the host builder") == std::string::npos || b.last_warning().find("device lost") == std::string::npos) {
            std::printf("failing backend: warning '%s'\n", b.last_warning().c_str());
            return 1;
        }
        FlatIndex f;
        b.finalize(f);
        if (well_formed(f, n, "failing backend")) return 1;
    }
    // ---- 4. a backend that cannot serve the build: refused before a point is accepted
    {
        const std::vector<float> x = data_set(100, d, 14);
        GraphBuilder b(params());
        HostMockBackend dev;
        dev.refuse = true;
        if (b.insert_batch_gpu(x.data(), 100, d, nullptr, 2, dev, 0, err) != ERR_DEVICE || b.nb_point() != 0) {
            std::printf("refusing backend: nb_point %llu, err '%s'\n", (unsigned long long)b.nb_point(), err.c_str());
            return 1;
        }
    }
    std::printf("window logic OK\n");
    return 0;
}
