// builder.cpp -- see builder.hpp.  Compile with -ffp-contract=off: in reference-order mode the
// stored edge distances must equal the crate's scalar arithmetic bit for bit.
#include "builder.hpp"
#include "worker_pool.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <unordered_set>

#include "hnswio.hpp"
#include "ln_f32.hpp"

namespace hnswgpu {

// ---------------------------------------------------------------------------------------
// Distance<f32>::eval on the host (anndists 0.1, default scalar build).
// ---------------------------------------------------------------------------------------
namespace {

float l2_ref(const float* a, const float* b, size_t d) {
    float norm = 0.f;
    for (size_t i = 0; i < d; ++i) {
        float t = a[i] - b[i];
        norm = norm + t * t;
    }
    return std::sqrt(norm);
}
float l1_ref(const float* a, const float* b, size_t d) {
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) s = s + std::fabs(a[i] - b[i]);
    return s;
}
float cosine_ref(const float* a, const float* b, size_t d) {
    double s0 = 0., s1 = 0., s2 = 0.;
    for (size_t i = 0; i < d; ++i) {
        float ab = a[i] * b[i], aa = a[i] * a[i], bb = b[i] * b[i];
        s0 = s0 + (double)ab;
        s1 = s1 + (double)aa;
        s2 = s2 + (double)bb;
    }
    if (s1 > 0. && s2 > 0.) return (float)std::max(1. - s0 / std::sqrt(s1 * s2), 0.);
    return 0.f;
}
float dot_ref(const float* a, const float* b, size_t d) {
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) s = s + a[i] * b[i];
    return std::max(1.f - s, 0.f);
}
// the distances between probability vectors of the crate's f32 FFI (src/libext.rs:334-345, :491-513); anndists 0.1
float hellinger_ref(const float* a, const float* b, size_t d) {
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) s = s + std::sqrt(a[i]) * std::sqrt(b[i]);
    return std::sqrt(std::max(1.f - s, 0.f));
}
float jeffreys_ref(const float* a, const float* b, size_t d) {
    const float M_MIN = 1.0e-30f;
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) s = s + (a[i] - b[i]) * ln_f32(std::max(a[i], M_MIN) / std::max(b[i], M_MIN));
    return s;
}
float jensenshannon_ref(const float* a, const float* b, size_t d) {
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) {
        const float mean_ab = 0.5f * (a[i] + b[i]);
        if (a[i] > 0.f) s = s + a[i] * ln_f32(a[i] / mean_ab);
        if (b[i] > 0.f) s = s + b[i] * ln_f32(b[i] / mean_ab);
    }
    return std::sqrt(0.5f * s);
}

// (function multi-versioning resolves through ifuncs, which run before a sanitizer's runtime is up: plain functions there)
#if defined(__SANITIZE_THREAD__) || defined(__SANITIZE_ADDRESS__)
#define HNSW_CLONES
#else
#define HNSW_CLONES __attribute__((target_clones("avx2", "default")))
#endif
// "fast" mode: 8 vertical accumulators over floor(d/8)*8 elements, horizontal add, scalar tail --
// the summation order of the crate's simdeez_f (AVX2) build.  Lane arithmetic is independent of
// the vector width the compiler picks, so both clones return the same bits.
HNSW_CLONES float l2_fast(const float* a, const float* b, size_t d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= d; i += 8)
        for (int j = 0; j < 8; ++j) {
            float t = a[i + j] - b[i + j];
            acc[j] += t * t;
        }
    float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    for (; i < d; ++i) {
        float t = a[i] - b[i];
        s += t * t;
    }
    return std::sqrt(s);
}
HNSW_CLONES float dot_fast(const float* a, const float* b, size_t d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= d; i += 8)
        for (int j = 0; j < 8; ++j) acc[j] += a[i + j] * b[i + j];
    float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    for (; i < d; ++i) s += a[i] * b[i];
    return std::max(1.f - s, 0.f);
}
HNSW_CLONES float l1_fast(const float* a, const float* b, size_t d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= d; i += 8)
        for (int j = 0; j < 8; ++j) acc[j] += std::fabs(a[i + j] - b[i + j]);
    float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    for (; i < d; ++i) s += std::fabs(a[i] - b[i]);
    return s;
}
HNSW_CLONES float cosine_fast(const float* a, const float* b, size_t d) {
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    size_t i = 0;
    for (; i + 4 <= d; i += 4)
        for (int j = 0; j < 4; ++j) {
            s0[j] += (double)(a[i + j] * b[i + j]);
            s1[j] += (double)(a[i + j] * a[i + j]);
            s2[j] += (double)(b[i + j] * b[i + j]);
        }
    double t0 = (s0[0] + s0[1]) + (s0[2] + s0[3]), t1 = (s1[0] + s1[1]) + (s1[2] + s1[3]),
           t2 = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    for (; i < d; ++i) {
        t0 += (double)(a[i] * b[i]);
        t1 += (double)(a[i] * a[i]);
        t2 += (double)(b[i] * b[i]);
    }
    if (t1 > 0. && t2 > 0.) return (float)std::max(1. - t0 / std::sqrt(t1 * t2), 0.);
    return 0.f;
}

// a polite spin: the architecture's pause / yield hint where there is one
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    std::this_thread::yield();
#endif
}
struct SpinGuard {
    std::atomic<uint8_t>& l;
    explicit SpinGuard(std::atomic<uint8_t>& x) : l(x) {
        while (l.exchange(1, std::memory_order_acquire)) {
            while (l.load(std::memory_order_relaxed)) cpu_relax();
        }
    }
    ~SpinGuard() { l.store(0, std::memory_order_release); }
};

// Lists are read far more often than written, and the popular nodes (the entry point, the upper layers, hubs) by every
// thread at once: readers take no lock (read_list copies under a sequence counter and retries when a writer was
// inside), writers exclude each other with the spin lock and keep every list inside its reserved capacity, so that a
// reader never follows a pointer to a buffer that was freed long ago.  (Exclusive locks for readers as well: 1M x 128
// took 48 s on 256 threads and 200k x 128 scaled no further than 16 threads.)
struct WriteGuard {
    SpinGuard g;
    std::atomic<uint32_t>& seq;
    template <class NodeT>
    explicit WriteGuard(NodeT& nd) : g(nd.lock), seq(nd.seq) {
        seq.store(seq.load(std::memory_order_relaxed) + 1u, std::memory_order_relaxed);
        std::atomic_thread_fence(std::memory_order_release);
    }
    ~WriteGuard() { seq.store(seq.load(std::memory_order_relaxed) + 1u, std::memory_order_release); }
};

struct EdgeLess {
    bool operator()(const Edge& a, const Edge& b) const { return a.dist < b.dist; }
};
struct EdgeGreater {
    bool operator()(const Edge& a, const Edge& b) const { return a.dist > b.dist; }
};

}  // namespace

// A neighbour list that lock-free readers may copy while a writer (node lock held, sequence counter odd) changes it: the
// buffer is published ONCE through an atomic pointer and keeps its place until the builder dies -- a reader never follows a
// pointer to freed memory, and every word it can see is read and written atomically (its copy is validated by the sequence
// counter).  A list that outgrows its buffer (only a foreign dump with over-long lists can do that) moves to a larger one; the
// old buffer is retired, not freed.  The writer-side interface is the slice of std::vector the builder uses.
class EdgeList {
public:
    EdgeList() = default;
    EdgeList(const EdgeList&) = delete;
    EdgeList& operator=(const EdgeList&) = delete;
    ~EdgeList() {
        delete[] data_.load(std::memory_order_relaxed);
        for (Edge* p : retired_) delete[] p;
    }
    // ---- readers (no lock)
    void snapshot(const Edge*& p, uint32_t& n) const {
        n = size_.load(std::memory_order_acquire);
        p = data_.load(std::memory_order_acquire);
    }
    // ---- writers (node lock held) and single-threaded phases
    size_t size() const { return size_.load(std::memory_order_relaxed); }
    bool empty() const { return size() == 0; }
    const Edge* begin() const { return data_.load(std::memory_order_relaxed); }
    const Edge* end() const { return begin() + size(); }
    const Edge& operator[](size_t i) const { return begin()[i]; }
    void reserve(size_t cap) {
        if (cap <= cap_) return;
        Edge* fresh = new Edge[cap];
        Edge* old = data_.load(std::memory_order_relaxed);
        const size_t n = size();
        for (size_t i = 0; i < n; ++i) put(fresh, i, old[i]);
        data_.store(fresh, std::memory_order_release);
        cap_ = (uint32_t)cap;
        if (old) retired_.push_back(old);
    }
    void push_back(const Edge& e) {
        const size_t n = size();
        if (n + 1 > cap_) reserve(n + 2);
        put(data_.load(std::memory_order_relaxed), n, e);
        size_.store((uint32_t)(n + 1), std::memory_order_release);
    }
    void pop_back() { size_.store((uint32_t)(size() - 1), std::memory_order_release); }
    void insert_at(size_t pos, const Edge& e) {  // keeps the order of the others
        const size_t n = size();
        if (n + 1 > cap_) reserve(n + 2);
        Edge* d = data_.load(std::memory_order_relaxed);
        for (size_t i = n; i > pos; --i) put(d, i, d[i - 1]);
        put(d, pos, e);
        size_.store((uint32_t)(n + 1), std::memory_order_release);
    }
    void assign(const std::vector<Edge>& v) {
        if (v.size() > cap_) reserve(v.size() + 2);
        Edge* d = data_.load(std::memory_order_relaxed);
        for (size_t i = 0; i < v.size(); ++i) put(d, i, v[i]);
        size_.store((uint32_t)v.size(), std::memory_order_release);
    }

private:
    static void put(Edge* d, size_t i, const Edge& e) {
        static_assert(sizeof(Edge) == 8, "an edge is one 64-bit word");
        uint64_t w;
        std::memcpy(&w, &e, 8);
        __atomic_store_n(reinterpret_cast<uint64_t*>(d + i), w, __ATOMIC_RELAXED);
    }
    std::atomic<Edge*> data_{nullptr};
    std::atomic<uint32_t> size_{0};
    uint32_t cap_ = 0;
    std::vector<Edge*> retired_;
};

struct GraphBuilder::Node {
    std::atomic<uint8_t> lock{0};      // writers (one at a time)
    std::atomic<uint32_t> seq{0};      // odd while a writer is inside: readers copy a list without any lock and retry
    uint8_t level = 0;
    int32_t rank = 0;
    uint64_t origin = 0;
    EdgeList l0;                           // neighbours[0]
    std::atomic<EdgeList*> up{nullptr};    // neighbours[1..15], allocated on first use (published once, freed with the node)
    ~Node() { delete[] up.load(std::memory_order_relaxed); }
    EdgeList& list(unsigned l) {
        if (l == 0) return l0;
        EdgeList* u = up.load(std::memory_order_acquire);
        if (!u) {
            u = new EdgeList[NB_LAYER_MAX - 1];
            up.store(u, std::memory_order_release);
        }
        return u[l - 1];
    }
    const EdgeList* list_if(unsigned l) const {
        if (l == 0) return &l0;
        const EdgeList* u = up.load(std::memory_order_acquire);
        return u ? &u[l - 1] : nullptr;
    }
};

struct GraphBuilder::Tls {
    std::vector<uint32_t> stamp;
    uint32_t epoch = 0;
    std::vector<Edge> cand_heap, res_heap, nbuf, res, sel, tmp, discarded;
    double t_select = 0, t_reverse = 0;  // HNSWGPU_BUILD_TIMING: seconds this thread spent in select_neighbours / reverse updates
    bool timing = false;
    void begin_visit(size_t n) {
        if (stamp.size() < n) stamp.resize(n, 0);
        if (++epoch == 0) {
            std::fill(stamp.begin(), stamp.end(), 0);
            epoch = 1;
        }
    }
    bool visit(uint32_t id) {  // true if newly visited
        if (stamp[id] == epoch) return false;
        stamp[id] = epoch;
        return true;
    }
};

GraphBuilder::Node& GraphBuilder::node(uint32_t id) const { return chunks_[id >> 16][id & (CHUNK - 1)]; }

GraphBuilder::GraphBuilder(const BuildParams& p) : p_(p) {
    max_layer_ = (unsigned)std::min<uint64_t>(NB_LAYER_MAX, p.max_layer);  // src/hnsw.rs:778
    if (max_layer_ == 0) max_layer_ = 1;
    double f = std::min(1.0, std::max(0.2, p.level_scale_factor));          // src/hnsw.rs:884-904
    scale_ = f / std::log((double)p.max_nb_connection);                      // src/hnsw.rs:327
    for (auto& a : layer_inserted_) a.store(0);
    layer_rank_next_.fill(0);
}
GraphBuilder::~GraphBuilder() = default;

GraphBuilder::GraphBuilder(const FlatIndex& f, bool fast_arithmetic) {
    p_.max_nb_connection = f.max_nb_connection;
    p_.ef_construction = f.ef_construction;
    p_.max_layer = f.nb_layer;
    p_.dist = f.dist;
    p_.extend_candidates = true;  // what load_hnsw sets on a reloaded index (src/hnswio.rs:510-511)
    p_.keep_pruned = false;
    p_.fast_arithmetic = fast_arithmetic;
    // a reloaded reference index rebuilds its LayerGenerator with maxlevel = NB_LAYER_MAX whatever the dumped nb_layer was
    // (src/hnswio.rs:773-777); the description that gets dumped again keeps the loaded nb_layer (dumped_nb_layer_)
    max_layer_ = NB_LAYER_MAX;
    dumped_nb_layer_ = (unsigned)std::min<uint64_t>(NB_LAYER_MAX, std::max<uint64_t>(1, f.nb_layer));
    // (the loader already applied the reference's reload rule to the dumped scale: see load_dump)
    scale_ = f.level_scale;
    p_.level_scale_factor = scale_ * std::log((double)std::max<uint64_t>(2, f.max_nb_connection));
    for (auto& a : layer_inserted_) a.store(0);
    layer_rank_next_.fill(0);
    n_ = f.n;
    d_ = f.dimension;
    for (unsigned l = 0; l < NB_LAYER_MAX; ++l) {
        layer_rank_next_[l] = f.layer_count(l);
        layer_inserted_[l].store(f.layer_count(l));
    }
    for (uint64_t i = 0; i < n_; ++i) {
        const uint32_t id = (uint32_t)i;
        if ((id >> 16) >= chunks_.size()) {
            chunks_.emplace_back(new Node[CHUNK]);
            vecs_.emplace_back(new float[CHUNK * d_]);
        }
        Node& nd = node(id);
        nd.level = (uint8_t)f.layer_of(id);
        nd.rank = f.rank_of(id);
        nd.origin = f.origin_id[i];
        std::memcpy(vecs_[id >> 16].get() + (uint64_t)(id & (CHUNK - 1)) * d_, f.vectors.data() + i * d_, d_ * sizeof(float));
        for (unsigned l = 0; l < NB_LAYER_MAX; ++l) {
            const uint64_t b = f.nbr_ptr[i * NB_LAYER_MAX + l], e = f.nbr_ptr[i * NB_LAYER_MAX + l + 1];
            if (e == b) continue;
            EdgeList& lst = nd.list(l);
            // sized here, before any worker thread exists, for the most the list can reach (a longer list of a foreign dump
            // included)
            const size_t cap = (l == 0 ? 2 * (size_t)p_.max_nb_connection : (size_t)p_.max_nb_connection) + 2;
            lst.reserve(std::max<size_t>(cap, (size_t)(e - b) + 2));
            for (uint64_t j = b; j < e; ++j) lst.push_back(Edge{f.nbr_flat[j], f.nbr_dist[j]});
        }
    }
    if (f.entry_flat != NO_POINT) {
        entry_.store((int64_t)f.entry_flat);
        entry_level_.store((int)f.layer_of(f.entry_flat));
    }
}

float GraphBuilder::eval(const float* a, const float* b) const {
    switch (p_.dist) {  // (no SIMD-order variant of the probability distances)
        case DIST_HELLINGER: return hellinger_ref(a, b, d_);
        case DIST_JEFFREYS: return jeffreys_ref(a, b, d_);
        case DIST_JENSENSHANNON: return jensenshannon_ref(a, b, d_);
        default: break;
    }
    if (!p_.fast_arithmetic) {
        switch (p_.dist) {
            case DIST_L2: return l2_ref(a, b, d_);
            case DIST_COSINE: return cosine_ref(a, b, d_);
            case DIST_DOT: return dot_ref(a, b, d_);
            default: return l1_ref(a, b, d_);
        }
    }
    switch (p_.dist) {
        case DIST_L2: return l2_fast(a, b, d_);
        case DIST_COSINE: return cosine_fast(a, b, d_);
        case DIST_DOT: return dot_fast(a, b, d_);
        default: return l1_fast(a, b, d_);
    }
}

// LayerGenerator::generate (src/hnsw.rs:363-374) on the documented SplitMix64(397) stream.
size_t GraphBuilder::draw_level() {
    auto next = [&]() {
        uint64_t z = (rng_state_ += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    double xsi = (double)(next() >> 11) * (1.0 / 9007199254740992.0);
    double level = -std::log(xsi) * scale_;
    double fl = std::floor(level);
    size_t ulevel = (fl >= (double)max_layer_ || !(fl == fl)) ? max_layer_ : (size_t)fl;
    if (ulevel >= max_layer_) ulevel = (size_t)(next() % (uint64_t)max_layer_);
    return ulevel;
}

void GraphBuilder::read_list(uint32_t id, unsigned layer, std::vector<Edge>& out) const {
    Node& nd = node(id);
    static_assert(sizeof(Edge) == 8, "an edge is copied as one 64-bit word");
    for (;;) {
        const uint32_t s1 = nd.seq.load(std::memory_order_acquire);
        if (s1 & 1u) { cpu_relax(); continue; }
        const EdgeList* l = nd.list_if(layer);
        uint32_t n = 0;
        const Edge* p = nullptr;
        if (l) l->snapshot(p, n);
        if (n != 0 && p == nullptr) continue;  // (length and buffer of two different moments: a writer is inside)
        out.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const uint64_t w = __atomic_load_n(reinterpret_cast<const uint64_t*>(p + i), __ATOMIC_RELAXED);
            std::memcpy(&out[i], &w, 8);
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (nd.seq.load(std::memory_order_relaxed) == s1) return;
    }
}

// a node's list for writing (under a WriteGuard): its buffer is reserved once, to the most it can ever hold
EdgeList& GraphBuilder::wlist(Node& nd, unsigned layer) const {
    EdgeList& v = nd.list(layer);
    const size_t cap = (layer == 0 ? 2 * (size_t)p_.max_nb_connection : (size_t)p_.max_nb_connection) + 2;
    v.reserve(std::max(cap, v.size() + 2));  // (no-op once the buffer exists with that capacity)
    return v;
}

// search_layer, unfiltered (src/hnsw.rs:922-1064) with flat heaps.  Result ascending by distance.
void GraphBuilder::search_layer(const float* q, uint32_t entry, size_t ef, unsigned layer, Tls& t,
                                std::vector<Edge>& out_sorted) {
    out_sorted.clear();
    if (layer_inserted_[layer].load(std::memory_order_acquire) == 0) return;  // :942-946
    auto& C = t.cand_heap;  // min-heap on dist
    auto& R = t.res_heap;   // max-heap on dist
    C.clear();
    R.clear();
    t.begin_visit(n_);
    float d0 = eval(q, vec(entry));
    t.visit(entry);
    C.push_back({entry, d0});
    R.push_back({entry, d0});
    while (!C.empty()) {
        std::pop_heap(C.begin(), C.end(), EdgeGreater());
        Edge c = C.back();
        C.pop_back();
        if (c.dist > R.front().dist) break;  // :981-993
        read_list(c.id, layer, t.nbuf);
        for (const Edge& e : t.nbuf) {
            if (!t.visit(e.id)) continue;
            float de = eval(q, vec(e.id));
            if (de < R.front().dist || R.size() < ef) {
                C.push_back({e.id, de});
                std::push_heap(C.begin(), C.end(), EdgeGreater());
                R.push_back({e.id, de});
                std::push_heap(R.begin(), R.end(), EdgeLess());
                if (R.size() > ef) {
                    std::pop_heap(R.begin(), R.end(), EdgeLess());
                    R.pop_back();
                }
            }
        }
    }
    out_sorted.assign(R.begin(), R.end());
    std::sort(out_sorted.begin(), out_sorted.end(), EdgeLess());
}

// select_neighbours (src/hnsw.rs:1299-1421).  cands_sorted ascending == pop order of the negated heap.
void GraphBuilder::select_neighbours(const float* q, std::vector<Edge>& cands, size_t nb_asked, bool extend_asked,
                                     unsigned layer, Tls& t, std::vector<Edge>& out) {
    out.clear();
    bool extend = false;
    if (cands.size() <= nb_asked) {
        if (!extend_asked) {
            out = cands;
            return;
        }
        extend = true;
    }
    if (extend) {
        std::unordered_set<uint32_t> in_set;
        for (const Edge& c : cands) in_set.insert(c.id);
        std::vector<uint32_t> fresh;
        size_t n0 = cands.size();
        for (size_t i = 0; i < n0; ++i) {
            read_list(cands[i].id, layer, t.tmp);
            for (const Edge& e : t.tmp)
                if (in_set.insert(e.id).second) fresh.push_back(e.id);
        }
        for (uint32_t id : fresh) cands.push_back({id, eval(q, vec(id))});
        std::sort(cands.begin(), cands.end(), EdgeLess());
    }
    auto& discarded = t.discarded;
    discarded.clear();
    for (size_t i = 0; i < cands.size() && out.size() < nb_asked; ++i) {
        const Edge& e = cands[i];
        bool e_to_insert = true;
        const float* ev = vec(e.id);
        for (const Edge& dn : out)
            if (eval(ev, vec(dn.id)) <= e.dist) {  // :1373-1375
                e_to_insert = false;
                break;
            }
        if (e_to_insert) out.push_back(e);
        else if (p_.keep_pruned) discarded.push_back(e);
    }
    if (p_.keep_pruned)
        for (size_t i = 0; i < discarded.size() && out.size() < nb_asked; ++i) out.push_back(discarded[i]);
}

// reverse_update_neighborhood_simple (src/hnsw.rs:1241-1289)
void GraphBuilder::reverse_update(uint32_t id, Tls& t) {
    Node& np = node(id);
    const unsigned level = np.level;
    for (int l = (int)level; l >= 0; --l) {
        read_list(id, (unsigned)l, t.tmp);
        for (const Edge& q : t.tmp) {
            if (q.id == id) continue;
            Node& qn = node(q.id);
            WriteGuard g(qn);
            EdgeList& lst = wlist(qn, level);  // list at the NEW point's level (:1257)
            bool already = false;
            for (const Edge& old : lst)
                if (old.id == id) { already = true; break; }
            if (already) continue;
            const size_t threshold = level > 0 ? p_.max_nb_connection : 2 * p_.max_nb_connection;
            // push + sort_unstable + pop-if-over (:1268-1283): lists are always ascending, so this is
            // an insertion after the last element <= the new distance
            Edge ne{id, q.dist};
            const Edge* pos = std::upper_bound(lst.begin(), lst.end(), ne, EdgeLess());
            lst.insert_at((size_t)(pos - lst.begin()), ne);
            if (lst.size() > threshold) lst.pop_back();
        }
    }
}

// insert_slice (src/hnsw.rs:1077-1215); generate_new_point's bookkeeping was done by insert_batch.
void GraphBuilder::insert_one(uint32_t id, Tls& t) {
    Node& np = node(id);
    const float* data = vec(id);
    const unsigned level = np.level;
    layer_inserted_[level].fetch_add(1, std::memory_order_acq_rel);  // points_by_layer[level].push (:516)
    int64_t ep = entry_.load(std::memory_order_acquire);
    if (ep < 0) {  // :1106-1109
        std::lock_guard<std::mutex> g(entry_mutex_);
        if (entry_.load() < 0) {
            entry_level_.store((int)level);
            entry_.store(id, std::memory_order_release);
            return;
        }
        ep = entry_.load();
    }
    uint32_t enter = (uint32_t)ep;
    const unsigned max_level_observed = node(enter).level;
    float dist_to_entry = eval(data, vec(enter));  // :1110-1112
    for (int l = (int)max_level_observed; l >= (int)level + 1; --l) {  // :1114-1155
        search_layer(data, enter, 1, (unsigned)l, t, t.res);
        if (!t.res.empty()) {
            Edge hit = t.res[0];
            {
                WriteGuard g(np);
                EdgeList& lst = wlist(np, (unsigned)l);
                if (lst.size() < (size_t)(uint8_t)p_.max_nb_connection) lst.push_back(hit);  // :1140-1144
            }
            if (hit.dist < dist_to_entry) {
                enter = hit.id;
                dist_to_entry = hit.dist;
            }
        }
    }
    for (int l = (int)level; l >= 0; --l) {  // :1158-1205
        search_layer(data, enter, p_.ef_construction, (unsigned)l, t, t.res);
        if (!t.res.empty()) {
            size_t nb_conn = l == 0 ? 2 * p_.max_nb_connection : p_.max_nb_connection;
            bool extend_c = l == 0 ? p_.extend_candidates : false;
            select_neighbours(data, t.res, nb_conn, extend_c, (unsigned)l, t, t.sel);
            std::stable_sort(t.sel.begin(), t.sel.end(), EdgeLess());  // :1195
            {
                WriteGuard g(np);
                wlist(np, (unsigned)l).assign(t.sel);  // :1197
            }
            if (!t.sel.empty()) enter = t.sel[0].id;  // :1201-1203
        }
    }
    reverse_update(id, t);  // :1210
    {                       // check_entry_point (src/hnsw.rs:534-557)
        std::lock_guard<std::mutex> g(entry_mutex_);
        if ((int)level > entry_level_.load()) {
            entry_level_.store((int)level);
            entry_.store(id, std::memory_order_release);
        }
    }
}

// generate_new_point for a whole batch, in input order (src/hnsw.rs:503-531): level, rank, origin id, vector
int GraphBuilder::append_points(const float* data, uint64_t n, uint64_t d, const uint64_t* ids, std::string& err) {
    if (!data || d == 0) { err = "insert: null data or zero dimension"; return ERR_ARG; }
    if (d_ == 0) d_ = d;
    if (d != d_) { err = "insert: dimension differs from the index dimension"; return ERR_ARG; }
    if (n_ + n >= NO_POINT) { err = "insert: too many points for 32-bit ids"; return ERR_ARG; }
    if (p_.max_nb_connection > 256 || p_.max_nb_connection < 2) { err = "max_nb_connection must be in [2, 256]"; return ERR_ARG; }
    const uint64_t first = n_;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t id = (uint32_t)(first + i);
        if ((id >> 16) >= chunks_.size()) {
            chunks_.emplace_back(new Node[CHUNK]);
            vecs_.emplace_back(new float[CHUNK * d_]);
        }
        Node& nd = node(id);
        nd.level = (uint8_t)draw_level();
        nd.rank = (int32_t)layer_rank_next_[nd.level]++;
        nd.origin = ids ? ids[i] : (first + i);
        std::memcpy(vecs_[id >> 16].get() + (uint64_t)(id & (CHUNK - 1)) * d_, data + i * d, d * sizeof(float));
    }
    n_ = first + n;
    return OK;
}

int GraphBuilder::insert_batch(const float* data, uint64_t n, uint64_t d, const uint64_t* ids, int nthreads,
                               std::string& err) {
    if (n == 0) return OK;
    const uint64_t first = n_;
    int rc = append_points(data, n, d, ids, err);
    if (rc != OK) return rc;
    // 32 threads unless told otherwise: construction is bound by random reads of vectors and lists from DRAM, and past
    // that many threads the caches only thrash (200k x 128 on a 256-thread host: 16 threads 4.8 s, 32 4.5 s, 64 4.8 s,
    // 128 6.3 s, 256 9.4 s)
    if (nthreads <= 0) nthreads = (int)std::min<unsigned>(std::thread::hardware_concurrency(), 32u);
    if (nthreads < 1) nthreads = 1;
    uint64_t start = first;
    Tls t0;
    if (first == 0) {  // the very first point only becomes the entry point (:1096-1109)
        insert_one(0, t0);
        start = 1;
    }
    if (nthreads == 1 || n_ - start < 64) {
        for (uint64_t i = start; i < n_; ++i) insert_one((uint32_t)i, t0);
        return OK;
    }
    std::atomic<uint64_t> next{start};
    WorkerPool::instance().run((unsigned)nthreads, (unsigned)nthreads, [&](unsigned) {
        Tls t;
        for (;;) {
            uint64_t i = next.fetch_add(1);
            if (i >= n_) break;
            insert_one((uint32_t)i, t);
        }
    });
    return OK;
}

// insert_slice for one point of a window, with the device's search results in place of the host searches
// (src/hnsw.rs:1106-1213): hits above the point's level, select_neighbours + list per layer, reverse update, entry point.
void GraphBuilder::apply_window_point(uint32_t id, uint32_t wi, uint32_t frozen_entry, unsigned frozen_entry_level, uint32_t layer_mask,
                                      const WindowSearchResults& r, uint64_t ef_c, Tls& t, std::vector<uint32_t>& dirty) {
    Node& np = node(id);
    const float* data = vec(id);
    const unsigned level = np.level;
    layer_inserted_[level].fetch_add(1, std::memory_order_acq_rel);  // points_by_layer[level].push (:516)
    for (int l = (int)frozen_entry_level; l >= (int)level + 1; --l) {  // :1114-1155, searched on the device with ef = 1
        const uint32_t hid = r.hit_ids[(size_t)wi * NB_LAYER_MAX + (unsigned)l];
        if (hid == NO_POINT) continue;
        WriteGuard g(np);
        EdgeList& lst = wlist(np, (unsigned)l);
        if (lst.size() < (size_t)(uint8_t)p_.max_nb_connection) lst.push_back(Edge{hid, r.hit_d[(size_t)wi * NB_LAYER_MAX + (unsigned)l]});  // :1140-1144
        dirty.push_back((id << 4) | (uint32_t)l);
    }
    for (int l = (int)level; l >= 0; --l) {  // :1158-1205
        t.res.clear();
        bool have_sel = false;
        if ((unsigned)l > frozen_entry_level) {
            // a layer above the snapshot's entry point: search_layer finds the entry point alone, if the layer has a point
            // at all (this one counts: generate_new_point pushed it before the search, :516)
            // (the serial reference sees this point in the layer when l is its own level; other empty layers return nothing)
            if ((unsigned)l == level || ((layer_mask >> l) & 1u)) t.res.push_back(Edge{frozen_entry, eval(data, vec(frozen_entry))});
        } else if (r.selected) {
            // the device ran select_neighbours for this slot as well (hnsw_build_select_kernel)
            const size_t slot = (size_t)r.slot0[wi] + (size_t)l;
            const uint32_t cnt = r.sel_n[slot];
            t.sel.clear();
            for (uint32_t j = 0; j < cnt; ++j) t.sel.push_back(Edge{r.sel_ids[slot * r.sel_stride + j], r.sel_d[slot * r.sel_stride + j]});
            have_sel = cnt > 0;
        } else {
            const size_t slot = (size_t)r.slot0[wi] + (size_t)l;
            const uint32_t cnt = r.out_n[slot];
            for (uint32_t j = 0; j < cnt; ++j) t.res.push_back(Edge{r.out_ids[slot * ef_c + j], r.out_d[slot * ef_c + j]});
        }
        if (!t.res.empty()) {
            size_t nb_conn = l == 0 ? 2 * p_.max_nb_connection : p_.max_nb_connection;
            bool extend_c = l == 0 ? p_.extend_candidates : false;
            const auto ts0 = t.timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
            select_neighbours(data, t.res, nb_conn, extend_c, (unsigned)l, t, t.sel);
            if (t.timing) t.t_select += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
            have_sel = true;
        }
        if (have_sel) {
            std::stable_sort(t.sel.begin(), t.sel.end(), EdgeLess());  // :1195
            {
                WriteGuard g(np);
                wlist(np, (unsigned)l).assign(t.sel);  // :1197
            }
            dirty.push_back((id << 4) | (uint32_t)l);
        }
    }
    // reverse_update_neighborhood_simple (:1210), remembering which lists of the snapshot are now out of date
    const auto tr0 = t.timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    for (int l = (int)level; l >= 0; --l) {
        read_list(id, (unsigned)l, t.tmp);
        for (const Edge& q : t.tmp)
            if (q.id != id) dirty.push_back((q.id << 4) | level);
    }
    reverse_update(id, t);
    if (t.timing) t.t_reverse += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
    {   // check_entry_point (src/hnsw.rs:534-557)
        std::lock_guard<std::mutex> g(entry_mutex_);
        if ((int)level > entry_level_.load()) {
            entry_level_.store((int)level);
            entry_.store(id, std::memory_order_release);
        }
    }
}

int GraphBuilder::insert_batch_gpu(const float* data, uint64_t n, uint64_t d, const uint64_t* ids, int nthreads,
                                   BuildSearchBackend& dev, uint64_t max_window, std::string& err) {
    if (n == 0) return OK;
    if (n_ + n >= (1ull << 28)) { err = "GPU-assisted construction: too many points (dirty-list ids are 28 bits)"; return ERR_ARG; }
    warning_.clear();
    // what can be refused up front is refused before a single point is accepted: the index is then unchanged
    int rc = dev.check(p_.ef_construction, err);
    if (rc != OK) return rc;
    const uint64_t first = n_;
    rc = append_points(data, n, d, ids, err);
    if (rc != OK) return rc;
    // host side of a window: 32 threads unless told otherwise -- measured on a 256-thread host, 1M x 128: 32 threads
    // 5.5 s for the whole build, 64 threads 6.3 s, 256 threads 10.9 s (the spin locks of popular nodes and the per-window
    // thread start-up cost more than the extra cores give)
    if (nthreads <= 0) nthreads = (int)std::min<unsigned>(std::thread::hardware_concurrency(), 32u);
    if (nthreads < 1) nthreads = 1;
    if (max_window == 0) max_window = 16384;
    // bootstrap on the host: the first points (a window that cannot see itself needs a graph to search in)
    uint64_t start = first;
    Tls t0;
    const uint64_t boot_until = max_window == 1 ? std::min<uint64_t>(n_, std::max<uint64_t>(first, 1)) : std::min<uint64_t>(n_, std::max<uint64_t>(first, 1024));
    for (; start < boot_until; ++start) insert_one((uint32_t)start, t0);
    if (start >= n_) return OK;
    // A device failure from here on (allocation, copy, kernel) finds the batch half inserted.  The points were accepted,
    // so they are linked: the host builder finishes [start, n_) (every window before `start` is complete), the call
    // succeeds, and the device's message is kept as a warning (last_warning()).
    auto finish_on_host = [&](const std::string& why) -> int {
        warning_ = "GPU-assisted construction fell back to the host builder for " + std::to_string(n_ - start) + " points: " + why;
        err.clear();
        std::atomic<uint64_t> next{start};
        const unsigned nt = n_ - start < 64 ? 1u : (unsigned)nthreads;
        WorkerPool::instance().run(nt, nt, [&](unsigned) {
            Tls t;
            for (;;) {
                const uint64_t i = next.fetch_add(1);
                if (i >= n_) break;
                insert_one((uint32_t)i, t);
            }
        });
        return OK;
    };
    // the device gets every vector and level, and the lists as they are now
    unsigned top_layer = 0;
    std::vector<uint8_t> levels(n_);
    for (uint64_t i = 0; i < n_; ++i) { levels[i] = node((uint32_t)i).level; top_layer = std::max<unsigned>(top_layer, levels[i]); }
    std::vector<const float*> chunk_ptrs;
    for (auto& c : vecs_) chunk_ptrs.push_back(c.get());
    rc = dev.begin(chunk_ptrs.data(), CHUNK, n_, d_, levels.data(), p_.dist, p_.max_nb_connection, p_.ef_construction, top_layer,
                   max_window, err);
    if (rc != OK) return finish_on_host(err);
    const uint32_t rw = dev.rec_words();
    // one record = {node, layer, the list's ids, NO_POINT padding}, written whole (no separate fill pass) into the backend's
    // pinned buffer
    auto pack_record = [&](uint32_t* rec, uint32_t key) {
        const uint32_t id = key >> 4, l = key & 15u;
        rec[0] = id;
        rec[1] = l;
        size_t j = 0;
        {
            Node& nd = node(id);
            SpinGuard g(nd.lock);
            const EdgeList* lst = nd.list_if(l);
            if (lst)
                for (; j < lst->size() && j + 2 < rw; ++j) rec[2 + j] = (*lst)[j].id;
        }
        for (; j + 2 < rw; ++j) rec[2 + j] = NO_POINT;
    };
    {   // initial snapshot: every list of the points inserted so far
        std::vector<uint32_t> dirty;
        for (uint64_t i = 0; i < start; ++i)
            for (unsigned l = 0; l < NB_LAYER_MAX; ++l) {
                const EdgeList* lst = node((uint32_t)i).list_if(l);
                if (lst && !lst->empty()) dirty.push_back(((uint32_t)i << 4) | l);
            }
        uint32_t* rec = dev.patch_buffer(dirty.size(), err);
        if (!dirty.empty() && !rec) return finish_on_host(err);
        for (size_t k = 0; k < dirty.size(); ++k) pack_record(rec + k * rw, dirty[k]);
        rc = dev.patch(dirty.size(), err);
        if (rc != OK) return finish_on_host(err);
    }
    WindowSearchResults res;
    // select_neighbours runs on the device too, unless candidates have to be extended from the lists of the graph under
    // construction (extend_candidates: what a reloaded index asks for at layer 0) -- that stays with the host's lists
    WindowSelect wsel;
    wsel.on_device = !p_.extend_candidates && std::getenv("HNSWGPU_HOST_SELECT") == nullptr;
    wsel.nb_layer0 = (uint32_t)(2 * p_.max_nb_connection);
    wsel.nb_upper = (uint32_t)p_.max_nb_connection;
    wsel.keep_pruned = p_.keep_pruned;
    std::vector<std::vector<uint32_t>> dirty_t((size_t)nthreads);
    // HNSWGPU_BUILD_TIMING=1: where the wall time of the windows goes (stderr, once per call)
    const bool timing = std::getenv("HNSWGPU_BUILD_TIMING") != nullptr;
    double t_search = 0, t_apply = 0, t_patch = 0, t_select_sum = 0, t_reverse_sum = 0;
    uint64_t n_windows = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    while (start < n_) {
        const uint64_t grown = max_window == 1 ? 1 : std::max<uint64_t>(256, start / 8);
        const uint32_t count = (uint32_t)std::min<uint64_t>({n_ - start, max_window, grown});
        const uint32_t frozen_entry = (uint32_t)entry_.load(std::memory_order_acquire);
        const unsigned frozen_level = (unsigned)entry_level_.load();
        uint32_t layer_mask = 0;
        for (unsigned l = 0; l < NB_LAYER_MAX; ++l)
            if (layer_inserted_[l].load(std::memory_order_acquire) > 0) layer_mask |= 1u << l;
        const double w0 = now();
        rc = dev.search_window((uint32_t)start, count, frozen_entry, frozen_level, layer_mask, wsel, res, err);
        if (rc != OK) return finish_on_host(err);
        const double w1 = now();
        for (auto& v : dirty_t) v.clear();
        std::atomic<uint32_t> next{0};
        const unsigned nt = (unsigned)std::min<uint64_t>((uint64_t)nthreads, std::max<uint32_t>(1, count / 8));
        WorkerPool& pool = WorkerPool::instance();
        pool.run(nt, nt, [&](unsigned tid) {
            Tls t;
            t.timing = timing;
            for (;;) {
                const uint32_t wi = next.fetch_add(1);
                if (wi >= count) break;
                apply_window_point((uint32_t)start + wi, wi, frozen_entry, frozen_level, layer_mask, res, p_.ef_construction, t, dirty_t[(size_t)tid]);
            }
            // the lists this thread changed: sorted and de-duplicated here, packed below (a list two threads touched is
            // sent twice, with the same content)
            auto& v = dirty_t[(size_t)tid];
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
            if (timing) {
                std::lock_guard<std::mutex> g(entry_mutex_);
                t_select_sum += t.t_select;
                t_reverse_sum += t.t_reverse;
            }
        });
        const double w2 = now();
        uint64_t n_records = 0;
        {
            std::vector<size_t> offs((size_t)nt + 1, 0);
            for (unsigned k = 0; k < nt; ++k) offs[(size_t)k + 1] = offs[(size_t)k] + dirty_t[(size_t)k].size();
            n_records = offs[(size_t)nt];
            uint32_t* rec = dev.patch_buffer(n_records, err);
            if (n_records != 0 && !rec) { start += count; if (start < n_) return finish_on_host(err); warning_ = err; err.clear(); return OK; }
            pool.run(nt, nt, [&](unsigned tid) {
                uint32_t* out = rec + offs[(size_t)tid] * rw;
                for (uint32_t key : dirty_t[(size_t)tid]) {
                    pack_record(out, key);
                    out += rw;
                }
            });
        }
        start += count;  // this window is linked on the host whatever happens to the snapshot
        rc = dev.patch(n_records, err);
        if (rc != OK) { if (start < n_) return finish_on_host(err); warning_ = err; err.clear(); return OK; }
        t_search += w1 - w0;
        t_apply += w2 - w1;
        t_patch += now() - w2;
        ++n_windows;
    }
    if (timing)
        std::fprintf(stderr, "[hnswgpu build] %llu windows: device searches %.2f s, host select/reverse-update (%d threads) %.2f s, "
                             "dirty lists packed + patched on the device %.2f s; thread-seconds inside select_neighbours %.2f, inside the "
                             "reverse updates %.2f\n",
                     (unsigned long long)n_windows, t_search, nthreads, t_apply, t_patch, t_select_sum, t_reverse_sum);
    return OK;
}

void GraphBuilder::finalize(FlatIndex& out) const {
    out = FlatIndex();
    out.format_version = 4;
    out.dumpmode = 1;
    out.max_nb_connection = p_.max_nb_connection;
    out.level_scale = scale_;
    out.nb_layer = (uint8_t)(dumped_nb_layer_ ? dumped_nb_layer_ : max_layer_);
    out.ef_construction = p_.ef_construction;
    out.dimension = d_;
    out.dist = p_.dist;
    out.distname = dist_type_name(p_.dist);
    out.extend_candidates = p_.extend_candidates;
    out.keep_pruned = p_.keep_pruned;
    out.n = n_;
    uint64_t acc = 0;
    for (unsigned l = 0; l < NB_LAYER_MAX; ++l) {
        out.layer_offset[l] = acc;
        acc += layer_rank_next_[l];
    }
    out.layer_offset[NB_LAYER_MAX] = acc;
    std::vector<uint32_t> flat_of(n_);
    for (uint64_t i = 0; i < n_; ++i) {
        const Node& nd = node((uint32_t)i);
        flat_of[i] = (uint32_t)(out.layer_offset[nd.level] + (uint64_t)nd.rank);
    }
    out.origin_id.resize(n_);
    out.vectors.resize(n_ * d_);
    std::vector<uint32_t> id_of_flat(n_);
    for (uint64_t i = 0; i < n_; ++i) id_of_flat[flat_of[i]] = (uint32_t)i;
    out.nbr_ptr.assign(n_ * NB_LAYER_MAX + 1, 0);
    uint64_t total = 0;
    for (uint64_t f = 0; f < n_; ++f) {
        const Node& nd = node(id_of_flat[f]);
        for (unsigned l = 0; l < NB_LAYER_MAX; ++l) {
            const EdgeList* lst = nd.list_if(l);
            total += lst ? lst->size() : 0;
            out.nbr_ptr[f * NB_LAYER_MAX + l + 1] = total;
        }
    }
    out.nbr_flat.resize(total);
    out.nbr_dist.resize(total);
    for (uint64_t f = 0; f < n_; ++f) {
        uint32_t id = id_of_flat[f];
        const Node& nd = node(id);
        out.origin_id[f] = nd.origin;
        std::memcpy(out.vectors.data() + f * d_, vec(id), d_ * sizeof(float));
        for (unsigned l = 0; l < NB_LAYER_MAX; ++l) {
            const EdgeList* lst = nd.list_if(l);
            if (!lst) continue;
            uint64_t b = out.nbr_ptr[f * NB_LAYER_MAX + l];
            for (size_t j = 0; j < lst->size(); ++j) {
                out.nbr_flat[b + j] = flat_of[(*lst)[j].id];
                out.nbr_dist[b + j] = (*lst)[j].dist;
            }
        }
    }
    int64_t ep = entry_.load();
    out.entry_flat = ep < 0 ? NO_POINT : flat_of[ep];
}

int build_index(const float* data, uint64_t n, uint64_t d, const uint64_t* ids, const BuildParams& p, FlatIndex& out,
                std::string& err) {
    GraphBuilder b(p);
    int rc = b.insert_batch(data, n, d, ids, p.nthreads, err);
    if (rc != OK) return rc;
    b.finalize(out);
    return OK;
}

}  // namespace hnswgpu
