// search_device.hip -- CDNA4 (gfx950) kernels for the batched-search hot path of hnsw_rs and
// the host driver that launches them.  Written for MI355X only: wave64, LDS, 8 XCDs.
//
// Reference path (file:line under /root/reference):
//   Hnsw::parallel_search          src/hnsw.rs:1612-1635   -> one wavefront per query, persistent grid
//   Hnsw::search_filter(None)      src/hnsw.rs:1487-1580   -> descent prologue + result epilogue
//   Hnsw::search_layer             src/hnsw.rs:922-1064    -> expansion loop (visited set in LDS,
//                                                             ef-bounded result/candidate set in VGPRs)
//   Distance<f32>::eval            anndists 0.1            -> dist_row<METRIC>: one LANE per neighbour,
//                                                             summed left-to-right exactly like the
//                                                             crate's scalar build (bit-identical)
//
// Arithmetic contract: every distance is accumulated in the reference's order (sequential over
// the vector index, no FMA contraction), so ids AND f32 distances equal the CPU oracle bit for
// bit on tie-free inputs.  Build with -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hnswio.hpp"
#include "search_device.hpp"

#pragma clang fp contract(off)

// tuning knobs (overridable with make TUNE=-D...)
#ifndef HNSW_LB_WAVES
#define HNSW_LB_WAVES 4   // __launch_bounds__ waves per SIMD: LDS already caps residency near 4
#endif
#ifndef HNSW_PHASE_TIMING
#define HNSW_PHASE_TIMING 0  // 1: per-query cycle counts of the expansion phases go to stats[8..15] (profiling builds)
#endif
#if HNSW_PHASE_TIMING
#define PH_T(var) const unsigned long long var = clock64()
#define PH_ACC(idx, t0, t1) ph[idx] += (uint32_t)((t1) - (t0))
#else
#define PH_T(var)
#define PH_ACC(idx, t0, t1)
#endif
#ifndef HNSW_ACC_SPLIT
#define HNSW_ACC_SPLIT 0  // 1: all 32 squares first, then the add chain (needs 32 more VGPRs); 0: per element
#endif

namespace hnswgpu {

namespace {

constexpr uint32_t EXPANDED = 0x80000000u;  // flag bit on a result entry whose neighbour list was read
constexpr uint32_t EMPTY_SLOT = 0xFFFFFFFFu;
// visited-set representations (see visit_*)
constexpr int TABLE_LDS_CELL16 = 0;
constexpr int TABLE_LDS_CELL32 = 1;
constexpr int TABLE_GLOBAL_BITMAP = 2;

// LDS carve (bytes) in front of the visited table
constexpr uint32_t TILE_ROWS = 16;                       // rows transposed per sub-batch
constexpr uint32_t TILE_PITCH = TILE_ROWS + 1;           // float4 units; +1 keeps ds_write_b128 conflict-free
constexpr uint32_t TILE_BYTES = 2 * 8 * TILE_PITCH * 16; // two buffers of 8 chunks x 16 B per row and pass
constexpr uint32_t IDS_BYTES = 64 * 4;

struct SearchArgs {
    const float* queries;   // [nq][row_stride], zero padded
    const uint32_t* qlist;  // optional: indices of the queries to run (retry pass), else nullptr
    uint32_t nq;            // number of work items
    uint32_t k;
    uint32_t ef;            // already max(ef_arg, k)
    uint32_t tbits;         // visited table = 1 << tbits cells
    uint32_t idbits;        // ceil(log2(n))
    uint32_t restbits;      // CELL16: idbits - tbits bits of the mixed id kept in the cell
    uint32_t* work_counter; // persistent-grid work queue head
    uint32_t* overflow_count;
    uint32_t* retry_out;    // queries whose visited table overflowed
    uint32_t* bitmap;       // [bitmap_blocks][bitmap_words]: per-workgroup visited bitmaps in HBM
    uint32_t bitmap_words;
    uint32_t bitmap_blocks; // workgroups with blockIdx.x < bitmap_blocks own a slice
    uint32_t* tie_list;     // strict ties: queries that met an exact distance tie (count at overflow_count + 3)
    uint64_t* out_ids;
    float* out_dists;
    uint8_t* out_layer;
    int32_t* out_rank;
    uint32_t* out_counts;
    uint32_t* stats;        // [nq_total][8] = n_dist, n_expand, n_ids_read, status, t_start, t_end (10 ns ticks), bitmap_used, 0
};

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ uint32_t popc64(unsigned long long m) { return (uint32_t)__popcll(m); }
__device__ __forceinline__ int ctz64(unsigned long long m) { return __ffsll((long long)m) - 1; }
__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// ---------------------------------------------------------------------------------------
// Distance<f32>::eval accumulators (anndists 0.1, scalar build).  add4 consumes four consecutive
// vector elements IN ORDER; rows and queries are zero padded and x + 0 == x, so running over the
// padding leaves every sum bit-identical to the reference's d-term left-to-right sum.
// ---------------------------------------------------------------------------------------
template <int METRIC>
struct Acc;
template <>
struct Acc<DIST_L2> {
    float a = 0.f;
    __device__ __forceinline__ void add4(const float4 s, const float4 r) {
        float t;
        t = s.x - r.x; a = a + t * t;
        t = s.y - r.y; a = a + t * t;
        t = s.z - r.z; a = a + t * t;
        t = s.w - r.w; a = a + t * t;
    }
    // 8 consecutive float4 (32 elements): all differences and squares first (independent, packed
    // math, full VALU rate), then the 32 dependent adds of the reference's left-to-right sum
    __device__ __forceinline__ void add32(const float4 (&s)[8], const float4 (&r)[8]) {
        float sq[32];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float t;
            t = s[c].x - r[c].x; sq[4 * c + 0] = t * t;
            t = s[c].y - r[c].y; sq[4 * c + 1] = t * t;
            t = s[c].z - r[c].z; sq[4 * c + 2] = t * t;
            t = s[c].w - r[c].w; sq[4 * c + 3] = t * t;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) a = a + sq[i];
    }
    __device__ __forceinline__ float fin() const { return __builtin_sqrtf(a); }
};
template <>
struct Acc<DIST_L1> {
    float a = 0.f;
    __device__ __forceinline__ void add4(const float4 s, const float4 r) {
        a = a + fabsf(s.x - r.x);
        a = a + fabsf(s.y - r.y);
        a = a + fabsf(s.z - r.z);
        a = a + fabsf(s.w - r.w);
    }
    __device__ __forceinline__ void add32(const float4 (&s)[8], const float4 (&r)[8]) {
        float v[32];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            v[4 * c + 0] = fabsf(s[c].x - r[c].x);
            v[4 * c + 1] = fabsf(s[c].y - r[c].y);
            v[4 * c + 2] = fabsf(s[c].z - r[c].z);
            v[4 * c + 3] = fabsf(s[c].w - r[c].w);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) a = a + v[i];
    }
    __device__ __forceinline__ float fin() const { return a; }
};
template <>
struct Acc<DIST_DOT> {
    float a = 0.f;
    __device__ __forceinline__ void add4(const float4 s, const float4 r) {
        a = a + s.x * r.x;
        a = a + s.y * r.y;
        a = a + s.z * r.z;
        a = a + s.w * r.w;
    }
    __device__ __forceinline__ void add32(const float4 (&s)[8], const float4 (&r)[8]) {
        float v[32];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            v[4 * c + 0] = s[c].x * r[c].x;
            v[4 * c + 1] = s[c].y * r[c].y;
            v[4 * c + 2] = s[c].z * r[c].z;
            v[4 * c + 3] = s[c].w * r[c].w;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) a = a + v[i];
    }
    __device__ __forceinline__ float fin() const { return fmaxf(1.f - a, 0.f); }
};
template <>
struct Acc<DIST_COSINE> {  // f32 products widened to f64, three f64 running sums
    double s0 = 0., s1 = 0., s2 = 0.;
    __device__ __forceinline__ void add4(const float4 s, const float4 r) {
        s0 = s0 + (double)(s.x * r.x); s1 = s1 + (double)(s.x * s.x); s2 = s2 + (double)(r.x * r.x);
        s0 = s0 + (double)(s.y * r.y); s1 = s1 + (double)(s.y * s.y); s2 = s2 + (double)(r.y * r.y);
        s0 = s0 + (double)(s.z * r.z); s1 = s1 + (double)(s.z * s.z); s2 = s2 + (double)(r.z * r.z);
        s0 = s0 + (double)(s.w * r.w); s1 = s1 + (double)(s.w * s.w); s2 = s2 + (double)(r.w * r.w);
    }
    __device__ __forceinline__ void add32(const float4 (&s)[8], const float4 (&r)[8]) {
#pragma unroll
        for (int c = 0; c < 8; ++c) add4(s[c], r[c]);
    }
    __device__ __forceinline__ float fin() const {
        if (s1 > 0. && s2 > 0.) {
            double du = 1. - s0 / __builtin_sqrt(s1 * s2);
            return (float)fmax(du, 0.);
        }
        return 0.f;
    }
};

// one lane per pair, straight from global memory (used by the arithmetic test kernel)
template <int METRIC>
__device__ __forceinline__ float dist_row(const float4* __restrict__ q, const float4* __restrict__ row, uint32_t nchunk) {
    Acc<METRIC> acc;
    for (uint32_t c = 0; c < nchunk; ++c) acc.add4(q[c], row[c]);
    return acc.fin();
}

// ---------------------------------------------------------------------------------------
// batch_dist: distances from the query (staged in LDS) to `nf` <= 64 rows whose flat ids sit in
// ids_lds[0..nf).  Returns the distance to row r in LANE r.
//
// HBM side: every load instruction fetches 8 rows x one full 128-byte line (8 lanes x 16 B per
// row), and all loads of up to 4 passes (16 rows x 512 B) are in flight before the first is used.
// Arithmetic side: the reference sums each distance left to right over the vector index, which a
// wave-wide reduction cannot reproduce bit for bit.  So the tile is transposed through LDS
// (tile[chunk][row], pitch 17 float4: conflict-free for both the 8-lane ds_write_b128 groups and
// the row-per-lane ds_read_b128) and lane r then walks row r sequentially -- one LANE per neighbour.
// ---------------------------------------------------------------------------------------
// Single-wavefront workgroups: LDS instructions of one wave execute in program order, so a
// ds_write followed by a ds_read of another lane's data needs no s_barrier and no drained counter --
// only a fence that keeps the compiler from reordering the two.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef float v4f __attribute__((ext_vector_type(4)));
// the query row is wave-uniform and read-only for the whole launch: constant address space =>
// s_load_dwordx16 into SGPRs, consumed directly as VALU scalar operands (no VGPRs, no LDS)
typedef __attribute__((address_space(4))) const v4f* qptr_t;

// one pass = 32 consecutive elements of every row of the sub-batch
__device__ __forceinline__ void tile_write(float4* buf, const float4 x0, const float4 x1, uint32_t lrow, uint32_t lchunk) {
    buf[lchunk * TILE_PITCH + lrow] = x0;
    buf[lchunk * TILE_PITCH + 8u + lrow] = x1;
}
template <int METRIC>
__device__ __forceinline__ void pass_read_accumulate(Acc<METRIC>& acc, const float4* buf, uint32_t rr, uint32_t pass, qptr_t q) {
    qptr_t qp = q + (size_t)pass * 8u;
    float4 sv[8], rv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const v4f s = qp[c];
        sv[c] = make_float4(s.x, s.y, s.z, s.w);
        rv[c] = buf[(uint32_t)c * TILE_PITCH + rr];
    }
#if HNSW_ACC_SPLIT
    acc.add32(sv, rv);
#else
#pragma unroll
    for (int c = 0; c < 8; ++c) acc.add4(sv[c], rv[c]);
#endif
}

// G passes (G x 16 rows x 128 B) are loaded before the first is consumed; the transposing tile is
// double buffered so that the LDS write+read of pass i+1 overlaps the dependent add chain of pass i.
// Named scalars on purpose: arrays of loads end up in scratch memory.
template <int METRIC, int G>
__device__ __forceinline__ void pass_group(Acc<METRIC>& acc, const float* __restrict__ p0, const float* __restrict__ p1,
                                           uint32_t pg, qptr_t q, float4* tile, uint32_t lrow, uint32_t lchunk,
                                           uint32_t rr, bool mine) {
    const float4* a0 = reinterpret_cast<const float4*>(p0 + (size_t)pg * 32u);
    const float4* a1 = reinterpret_cast<const float4*>(p1 + (size_t)pg * 32u);
    float4 x00, x01, x10, x11, x20, x21, x30, x31;
    x00 = a0[0]; x01 = a1[0];
    if constexpr (G > 1) { x10 = a0[8]; x11 = a1[8]; }
    if constexpr (G > 2) { x20 = a0[16]; x21 = a1[16]; }
    if constexpr (G > 3) { x30 = a0[24]; x31 = a1[24]; }
    float4* t0 = tile;
    float4* t1 = tile + 8 * TILE_PITCH;
    // pass i+1 is written into the other buffer BEFORE pass i is consumed (LDS ops of one wave run in
    // program order, so no hazard: the reads of a buffer always precede its next overwrite)
    wave_lds_fence();
    tile_write(t0, x00, x01, lrow, lchunk);
    if constexpr (G > 1) tile_write(t1, x10, x11, lrow, lchunk);
    wave_lds_fence();
    if (mine) pass_read_accumulate<METRIC>(acc, t0, rr, pg, q);
    if constexpr (G > 2) { wave_lds_fence(); tile_write(t0, x20, x21, lrow, lchunk); wave_lds_fence(); }
    if constexpr (G > 1) { if (mine) pass_read_accumulate<METRIC>(acc, t1, rr, pg + 1, q); }
    if constexpr (G > 3) { wave_lds_fence(); tile_write(t1, x30, x31, lrow, lchunk); wave_lds_fence(); }
    if constexpr (G > 2) { if (mine) pass_read_accumulate<METRIC>(acc, t0, rr, pg + 2, q); }
    if constexpr (G > 3) { if (mine) pass_read_accumulate<METRIC>(acc, t1, rr, pg + 3, q); }
}

template <int METRIC>
__device__ __forceinline__ float batch_dist(const float* __restrict__ vec, uint32_t row_stride, qptr_t q,
                                            float4* tile, const uint32_t* ids_lds, uint32_t nf, int lane) {
    const uint32_t npass = row_stride >> 5;  // 32 floats (8 x 16 B) per row and pass
    const uint32_t lrow = (uint32_t)lane >> 3, lchunk = (uint32_t)lane & 7u;
    float result = INFINITY;
    for (uint32_t s0 = 0; s0 < nf; s0 += TILE_ROWS) {
        // lanes past the last row re-read row s0 (same cache lines as the lanes that own it: free)
        const uint32_t r0 = s0 + lrow, r1 = r0 + 8;
        const float* p0 = vec + (size_t)ids_lds[r0 < nf ? r0 : s0] * row_stride + lchunk * 4u;
        const float* p1 = vec + (size_t)ids_lds[r1 < nf ? r1 : s0] * row_stride + lchunk * 4u;
        const bool mine = ((uint32_t)lane >> 4) == (s0 >> 4) && (uint32_t)lane < nf;
        const uint32_t rr = (uint32_t)lane & 15u;
        Acc<METRIC> acc;
        uint32_t pg = 0;
        for (; pg + 4 <= npass; pg += 4) pass_group<METRIC, 4>(acc, p0, p1, pg, q, tile, lrow, lchunk, rr, mine);
        const uint32_t rem = npass - pg;
        if (rem == 3) pass_group<METRIC, 3>(acc, p0, p1, pg, q, tile, lrow, lchunk, rr, mine);
        else if (rem == 2) pass_group<METRIC, 2>(acc, p0, p1, pg, q, tile, lrow, lchunk, rr, mine);
        else if (rem == 1) pass_group<METRIC, 1>(acc, p0, p1, pg, q, tile, lrow, lchunk, rr, mine);
        if (mine) result = acc.fin();
    }
    return result;
}

// ---------------------------------------------------------------------------------------
// Visited set (reference: hashbrown::HashMap<PointId, Arc<Point>>, src/hnsw.rs:955-956, :1016-1017),
// one per wavefront.  All three are EXACT (no false positives) and return true when `id` was not yet
// visited, marking it.
//   CELL16 : open addressing in LDS with 16-bit cells.  The id is passed through a bijection of
//            [0, 2^idbits); its top tbits select the home cell, and the cell stores
//            {valid, displacement from home, remaining restbits} -- enough to identify the id, at
//            half the LDS of a table of full ids (LDS per wave is what bounds occupancy here).
//   CELL32 : same with full 32-bit ids (large indexes where restbits would not fit).
//   GLOBAL_BITMAP : one bit per point in a per-workgroup HBM slice; cannot overflow; last resort.
// A lane that cannot place its id within the displacement budget reports overflow; the query is
// then re-run from scratch with a larger representation (never a silent miss).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix_id(uint32_t id, uint32_t idbits) {
    const uint32_t mask = idbits >= 32 ? 0xFFFFFFFFu : ((1u << idbits) - 1u);
    const uint32_t sh = (idbits + 1u) >> 1;
    uint32_t h = (id * 0x9E3779B1u) & mask;
    h ^= h >> sh;
    h = (h * 0x85EBCA6Bu) & mask;
    h ^= h >> sh;
    return h;
}
// inverse of mix_id (each step is a bijection of [0, 2^idbits): odd multipliers have inverses mod 2^k,
// and x ^= x >> s is an involution once 2s >= idbits)
__device__ __forceinline__ uint32_t unmix_id(uint32_t h, uint32_t idbits) {
    const uint32_t mask = idbits >= 32 ? 0xFFFFFFFFu : ((1u << idbits) - 1u);
    const uint32_t sh = (idbits + 1u) >> 1;
    h ^= h >> sh;
    h = (h * 0xA5CB9243u) & mask;  // 0x85EBCA6B^-1 mod 2^32
    h ^= h >> sh;
    h = (h * 0x0E8B2F51u) & mask;  // 0x9E3779B1^-1 mod 2^32
    return h;
}
// returns 0 = already visited, 1 = newly marked, 2 = no room.
// Linear probing, but a probe fetches the whole aligned 16-byte block (8 cells) with ONE ds_read_b128 and
// scans it in registers: a wave waits for its slowest lane, and the dependent LDS round trips of that
// lane were the cost of this function.
__device__ __forceinline__ int visit_cell16(uint32_t* words, uint32_t tbits, uint32_t idbits, uint32_t restbits, uint32_t id) {
    const uint32_t h = mix_id(id, idbits);
    const uint32_t home = h >> restbits;
    const uint32_t rest = h & ((1u << restbits) - 1u);
    const uint32_t tmask = (1u << tbits) - 1u;
    const uint32_t maxdisp = 1u << (15u - restbits);
    const uint4* blocks = reinterpret_cast<const uint4*>(words);
    uint32_t disp = 0;
    while (disp < maxdisp) {
        const uint32_t pos = (home + disp) & tmask;
        const uint4 v = blocks[pos >> 3];
        const unsigned long long lo = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
        const unsigned long long hi = (unsigned long long)v.z | ((unsigned long long)v.w << 32);
        bool reread = false;
        for (uint32_t j = pos & 7u; j < 8u; ++j, ++disp) {
            if (disp >= maxdisp) return 2;
            const uint32_t cellv = (uint32_t)(((j < 4u ? lo : hi) >> ((j & 3u) * 16u)) & 0xFFFFull);
            const uint32_t expect = 0x8000u | (disp << restbits) | rest;
            if (cellv == expect) return 0;
            if (cellv == 0u) {
                const uint32_t cell = (pos & ~7u) + j;          // (no wrap inside an aligned block)
                const uint32_t w32 = j < 2u ? v.x : j < 4u ? v.y : j < 6u ? v.z : v.w;
                const uint32_t sh = (j & 1u) * 16u;
                if (atomicCAS(&words[cell >> 1], w32, w32 | (expect << sh)) == w32) return 1;
                reread = true;  // this word changed under us (another lane): look at the block again
                break;
            }
        }
        (void)reread;
    }
    return 2;
}
__device__ __forceinline__ int visit_cell32(uint32_t* tab, uint32_t tbits, uint32_t id) {
    const uint32_t mask = (1u << tbits) - 1u;
    uint32_t h = (id * 0x9E3779B1u) >> (32u - tbits);
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const uint32_t old = atomicCAS(&tab[h], EMPTY_SLOT, id);
        if (old == EMPTY_SLOT) return 1;
        if (old == id) return 0;
        h = (h + 1u) & mask;
    }
    return 2;
}
__device__ __forceinline__ int visit_bitmap(uint32_t* bm, uint32_t id) {
    const uint32_t bit = 1u << (id & 31u);
    const uint32_t old = atomicOr(&bm[id >> 5], bit);
    return (old & bit) == 0u ? 1 : 0;
}

// ---------------------------------------------------------------------------------------
// Result set R (reference: return_points, a max-heap capped at ef, plus candidate_points).
// Kept as ONE array sorted ascending by distance, entry j in VGPR slot j/64 of lane j%64, with an
// EXPANDED flag: the candidates of the reference are exactly the not-yet-expanded members of R
// (an entry evicted from R can only terminate the loop when popped: SURVEY.md section 3.1).
// Insertion keeps arrival order among equal distances and reports whether an equal distance was
// already present (a tie: the reference's answer then depends on its heaps' internal order).
// ---------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ bool r_insert(float (&rd)[S], uint32_t (&ri)[S], uint32_t& len, uint32_t ef, float xd,
                                         uint32_t xi, int lane) {
    uint32_t pos = 0;
    bool tie = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint32_t j = (uint32_t)s * 64u + (uint32_t)lane;
        pos += popc64(__ballot(j < len && rd[s] <= xd));
        tie = tie || (__ballot(j < len && rd[s] == xd) != 0ull);
    }
#pragma unroll
    for (int s = S - 1; s >= 0; --s) {
        // entry j-1 -> j: DPP wave_shr:1 inside a slot, lane 63 of the previous slot into lane 0
        int od = __float_as_int(rd[s]), oi = (int)ri[s];
        if (s > 0) {
            od = __builtin_amdgcn_readlane(__float_as_int(rd[s - 1]), 63);
            oi = __builtin_amdgcn_readlane((int)ri[s - 1], 63);
        }
        const float pd = __int_as_float(__builtin_amdgcn_update_dpp(od, __float_as_int(rd[s]), 0x138, 0xf, 0xf, false));
        const uint32_t pi = (uint32_t)__builtin_amdgcn_update_dpp(oi, (int)ri[s], 0x138, 0xf, 0xf, false);
        const uint32_t j = (uint32_t)s * 64u + (uint32_t)lane;
        if (j > pos) { rd[s] = pd; ri[s] = pi; }
        else if (j == pos) { rd[s] = xd; ri[s] = xi; }
    }
    len = len + 1 > ef ? ef : len + 1;  // the entry pushed past ef-1 is the evicted worst (src/hnsw.rs:1051-1053)
    return tie;
}
template <int S>
__device__ __forceinline__ float r_worst(const float (&rd)[S], uint32_t len) {
    float w = 0.f;
    const uint32_t j = len - 1;
#pragma unroll
    for (int s = 0; s < S; ++s)
        if ((j >> 6) == (uint32_t)s) w = readlane_f(rd[s], (int)(j & 63));
    return w;
}
// nearest not-yet-expanded member of R: returns its index, or -1
template <int S>
__device__ __forceinline__ int r_next(const uint32_t (&ri)[S], uint32_t len, int lane) {
    int j0 = -1;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint32_t j = (uint32_t)s * 64u + (uint32_t)lane;
        const unsigned long long m = __ballot(j < len && (ri[s] & EXPANDED) == 0u);
        if (j0 < 0 && m != 0ull) j0 = s * 64 + ctz64(m);
    }
    return j0;
}
template <int S>
__device__ __forceinline__ uint32_t r_id_at(const uint32_t (&ri)[S], int j) {
    uint32_t v = 0;
#pragma unroll
    for (int s = 0; s < S; ++s)
        if ((j >> 6) == s) v = readlane_u(ri[s], j & 63);
    return v & ~EXPANDED;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}

// ---------------------------------------------------------------------------------------
// The search kernel: one wavefront (= one 64-thread workgroup) per query, persistent grid pulling
// query indices from a global counter.
// status per query: 0 ok, 1 visited-set overflow (re-run bigger), 2 ok but an exact distance tie
// was met while inserting (answer is a valid search result; order among equals may differ from the
// reference's heap order -- see DESIGN.md "ties").
// ---------------------------------------------------------------------------------------
template <int METRIC, int S, int TABLE>
__global__ __launch_bounds__(64, HNSW_LB_WAVES) void hnsw_search_kernel(DeviceIndexView ix, SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float4* tile = reinterpret_cast<float4*>(lds_raw);
    uint32_t* ids_lds = reinterpret_cast<uint32_t*>(lds_raw + TILE_BYTES);
    uint32_t* table = reinterpret_cast<uint32_t*>(lds_raw + TILE_BYTES + IDS_BYTES);
    const int lane = (int)threadIdx.x;
    const uint32_t table_cells = 1u << a.tbits;
    const uint32_t table_words = TABLE == TABLE_LDS_CELL16 ? table_cells >> 1 : table_cells;
    const uint32_t table_limit = table_cells - (table_cells >> 2);  // stop inserting at 75 % load
    // this workgroup's private bitmap slice (in-launch fallback when the LDS table overflows)
    uint32_t* bitmap = blockIdx.x < a.bitmap_blocks ? a.bitmap + (size_t)blockIdx.x * a.bitmap_words : nullptr;
    const bool single_batch = ix.deg_stride <= 64u;
    bool use_bm = TABLE == TABLE_GLOBAL_BITMAP;

    auto visit = [&](uint32_t id) -> int {
        if (TABLE == TABLE_GLOBAL_BITMAP || use_bm) return visit_bitmap(bitmap, id);
        if constexpr (TABLE == TABLE_LDS_CELL16) return visit_cell16(table, a.tbits, a.idbits, a.restbits, id);
        else return visit_cell32(table, a.tbits, id);
    };

    // A query that outgrows its LDS table moves its visited set into the HBM bitmap and carries on
    // there (cells identify their ids exactly, so nothing is recomputed).
    auto migrate_to_bitmap = [&]() {
        for (uint32_t i = (uint32_t)lane; i < a.bitmap_words; i += 64) bitmap[i] = 0u;
        __syncthreads();
        for (uint32_t i = (uint32_t)lane; i < table_cells; i += 64) {
            uint32_t id = EMPTY_SLOT;
            if constexpr (TABLE == TABLE_LDS_CELL16) {
                const uint32_t half = (table[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu;
                if (half != 0u) {
                    const uint32_t disp = (half & 0x7FFFu) >> a.restbits;
                    const uint32_t rest = half & ((1u << a.restbits) - 1u);
                    const uint32_t home = (i - disp) & (table_cells - 1u);
                    id = unmix_id((home << a.restbits) | rest, a.idbits);
                }
            } else if constexpr (TABLE == TABLE_LDS_CELL32) {
                id = table[i];
            }
            if (id != EMPTY_SLOT) atomicOr(&bitmap[id >> 5], 1u << (id & 31u));
        }
        __syncthreads();
        use_bm = true;
    };

    for (;;) {
        uint32_t wi = 0;
        if (lane == 0) wi = atomicAdd(a.work_counter, 1u);
        wi = readlane_u(wi, 0);
        if (wi >= a.nq) break;
        const uint32_t q = a.qlist ? a.qlist[wi] : wi;

        const qptr_t qrow = (qptr_t)(a.queries + (size_t)q * ix.row_stride);  // wave-uniform => scalar loads
        const uint32_t t_start = (uint32_t)wall_clock64();

        uint32_t n_dist, n_expand, n_ids, status, len, n_visited_final = 0;
        bool tie;
#if HNSW_PHASE_TIMING
        uint32_t ph[4] = {0, 0, 0, 0};
#endif
        float rd[S];
        uint32_t ri[S];
        use_bm = TABLE == TABLE_GLOBAL_BITMAP;
        for (;;) {
        // ---- reset the visited set
        if (TABLE == TABLE_GLOBAL_BITMAP || use_bm) {
            for (uint32_t i = (uint32_t)lane; i < a.bitmap_words; i += 64) bitmap[i] = 0u;
        } else if constexpr (TABLE == TABLE_LDS_CELL16) {
            for (uint32_t i = (uint32_t)lane; i < table_words; i += 64) table[i] = 0u;
        } else {
            for (uint32_t i = (uint32_t)lane; i < table_words; i += 64) table[i] = EMPTY_SLOT;
        }
        __syncthreads();

        n_dist = 0; n_expand = 0; n_ids = 0; status = 0;
        tie = false;

        // ---- greedy descent: ONE scan of the pivot's list per layer (src/hnsw.rs:1506-1529)
        uint32_t pivot = ix.entry;
        if (lane == 0) ids_lds[0] = pivot;
        __syncthreads();
        float dcur = readlane_f(batch_dist<METRIC>(ix.vec, ix.row_stride, qrow, tile, ids_lds, 1u, lane), 0);
        n_dist += 1;
        for (int layer = (int)ix.entry_level; layer >= 1; --layer) {
            uint32_t b = 0, e = 0;
            if ((uint32_t)layer <= ix.n_up_layers) {
                const uint32_t* ptr = ix.up_ptr + (size_t)(layer - 1) * ((size_t)ix.n + 1);
                b = ptr[pivot];
                e = ptr[pivot + 1];
            }
            n_expand += 1;
            n_ids += e - b;
            float best = INFINITY;
            uint32_t best_id = pivot;
            for (uint32_t base = b; base < e; base += 64) {
                const uint32_t j = base + (uint32_t)lane;
                const bool valid = j < e;
                const uint32_t id = valid ? ix.up_ids[j] : 0u;
                const uint32_t nf = e - base < 64u ? e - base : 64u;
                __syncthreads();
                if (valid) ids_lds[lane] = id;
                __syncthreads();
                const float dl = batch_dist<METRIC>(ix.vec, ix.row_stride, qrow, tile, ids_lds, nf, lane);  // INF in lanes >= nf
                n_dist += nf;
                const float m = wave_min(dl);
                const unsigned long long eq = __ballot(valid && dl == m);
                if (eq != 0ull && m < best) {  // strict '<': the first index wins ties (:1519)
                    best = m;
                    best_id = readlane_u(id, ctz64(eq));
                }
            }
            if (best < dcur) {  // pivot replaced once per layer, only if strictly better (:1519-1528)
                dcur = best;
                pivot = best_id;
            }
        }

        // ---- search_layer at the lowest non-empty layer (src/hnsw.rs:1542, :922-1064)
#pragma unroll
        for (int s = 0; s < S; ++s) { rd[s] = 0.f; ri[s] = 0u; }
        len = 1;
        if (lane == 0) { rd[0] = dcur; ri[0] = pivot; }  // dist_to_entry_point == eval(q, pivot) (:952)
        uint32_t n_visited = 1;
        if (lane == 0) (void)visit(pivot);
        __syncthreads();

        uint32_t spec_for = EMPTY_SLOT, spec_ids = EMPTY_SLOT;  // prefetched id row of the likely next candidate
        for (;;) {
            PH_T(pt0);
            // c = nearest unexpanded member of R (candidate_points.pop(), :971)
            const int cj = r_next<S>(ri, len, lane);
            if (cj < 0) break;  // every remaining candidate is farther than R's worst (:981-993)
            const uint32_t c = r_id_at<S>(ri, cj);
#pragma unroll
            for (int s = 0; s < S; ++s)
                if ((cj >> 6) == s && lane == (cj & 63)) ri[s] |= EXPANDED;
            n_expand += 1;
            const uint32_t* nrow = ix.nbr0 + (size_t)c * ix.deg_stride;
            const uint32_t nbatch = (ix.deg_stride + 63u) >> 6;
            for (uint32_t bi = 0; bi < nbatch; ++bi) {
                const uint32_t j = bi * 64u + (uint32_t)lane;
                uint32_t id;
                if (single_batch && spec_for == c) id = spec_ids;
                else id = j < ix.deg_stride ? nrow[j] : EMPTY_SLOT;
                if (single_batch) {
                    // speculate on the next candidate: its id row is fetched while this one is processed
                    const int nj = r_next<S>(ri, len, lane);
                    if (nj >= 0) {
                        spec_for = r_id_at<S>(ri, nj);
                        spec_ids = (uint32_t)lane < ix.deg_stride ? ix.nbr0[(size_t)spec_for * ix.deg_stride + (uint32_t)lane] : EMPTY_SLOT;
                    } else {
                        spec_for = EMPTY_SLOT;
                    }
                }
                const bool valid = id != EMPTY_SLOT;
                const unsigned long long vm = __ballot(valid);
                PH_T(pt1);
                PH_ACC(0, pt0, pt1);  // candidate selection + wait for its id row
                if (vm == 0ull) break;  // lists are padded at the end only
                n_ids += popc64(vm);
                if (TABLE != TABLE_GLOBAL_BITMAP && !use_bm && n_visited + 64 > table_limit) {
                    if (bitmap == nullptr) { status = 1; break; }  // no slice for this workgroup: host re-runs it
                    migrate_to_bitmap();
                }
                int vr = 0;
                if (valid) vr = visit(id);
                if (__ballot(vr == 2) != 0ull) {  // displacement budget exhausted inside the table (rare)
                    if (use_bm || bitmap == nullptr) { status = 1; break; }
                    migrate_to_bitmap();  // includes the ids this batch already placed
                    if (vr == 2) vr = visit_bitmap(bitmap, id);
                }
                const bool fresh = vr == 1;
                const unsigned long long fm = __ballot(fresh);
                const uint32_t nf = popc64(fm);
                PH_T(pt2);
                PH_ACC(1, pt1, pt2);  // visited-set probes
                if (nf == 0u) continue;
                n_visited += nf;
                n_dist += nf;
                // compact the fresh ids in list order: rank r -> lane r
                __syncthreads();
                if (fresh) ids_lds[popc64(fm & lanemask_lt(lane))] = id;
                __syncthreads();
                const uint32_t idc = (uint32_t)lane < nf ? ids_lds[lane] : 0u;
                const float de = batch_dist<METRIC>(ix.vec, ix.row_stride, qrow, tile, ids_lds, nf, lane);
                PH_T(pt3);
                PH_ACC(2, pt2, pt3);  // compaction + row loads + transposed accumulate
                // accept rule applied sequentially in list order (:1028-1053)
                float worst = r_worst<S>(rd, len);
                unsigned long long cand = __ballot((uint32_t)lane < nf && (len < a.ef || de < worst));
                while (cand != 0ull) {
                    const int jl = ctz64(cand);
                    cand &= cand - 1ull;
                    const float xd = readlane_f(de, jl);
                    if (xd < worst || len < a.ef) {
                        const uint32_t xi = readlane_u(idc, jl);
                        tie = r_insert<S>(rd, ri, len, a.ef, xd, xi, lane) || tie;
                        worst = r_worst<S>(rd, len);
                    }
                }
                PH_T(pt4);
                PH_ACC(3, pt3, pt4);  // result-set insertions
            }
            if (status != 0) break;
        }
        n_visited_final = n_visited;
        break;
        }  // (single pass; kept as a block so that the reset above stays next to the body)
        if (status == 0 && tie) {
            status = 2;
            if (a.tie_list != nullptr && lane == 0) a.tie_list[atomicAdd(a.overflow_count + 3, 1u)] = q;
        }

        // ---- into_sorted_vec + truncate to min(knbn, ef, len) (:1544-1547, :1567-1578)
        if (status != 1) {
            const uint32_t cnt = len < a.k ? len : a.k;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const uint32_t j = (uint32_t)s * 64u + (uint32_t)lane;
                if (j < a.k) {
                    const size_t o = (size_t)q * a.k + j;
                    if (j < cnt) {
                        const uint32_t flat = ri[s] & ~EXPANDED;
                        uint32_t l = 0;
                        while (l + 1 < NB_LAYER_MAX && flat >= ix.layer_offset[l + 1]) ++l;
                        a.out_ids[o] = ix.origin_id[flat];
                        a.out_dists[o] = rd[s];
                        if (a.out_layer) a.out_layer[o] = (uint8_t)l;
                        if (a.out_rank) a.out_rank[o] = (int32_t)(flat - ix.layer_offset[l]);
                    } else {
                        a.out_ids[o] = 0ull;
                        a.out_dists[o] = 0.f;
                        if (a.out_layer) a.out_layer[o] = 0;
                        if (a.out_rank) a.out_rank[o] = 0;
                    }
                }
            }
            if (lane == 0) a.out_counts[q] = cnt;
        } else if (lane == 0) {
            const uint32_t slot = atomicAdd(a.overflow_count, 1u);
            a.retry_out[slot] = q;
        }
        if (lane == 0 && TABLE != TABLE_GLOBAL_BITMAP) {
            // feedback for the host's table sizing (ctrl[2], ctrl[3])
            if (use_bm) atomicAdd(a.overflow_count + 1, 1u);
            if (n_visited_final > (table_limit >> 1)) atomicAdd(a.overflow_count + 2, 1u);
        }
        if (lane == 0) {
            uint32_t* st = a.stats + (size_t)q * 8;
            st[0] = n_dist; st[1] = n_expand; st[2] = n_ids; st[3] = status;
            st[4] = t_start; st[5] = (uint32_t)wall_clock64(); st[6] = use_bm ? 1u : 0u; st[7] = 0u;
#if HNSW_PHASE_TIMING
            st[7] = ph[0]; st[3] = ph[1]; st[2] = ph[2]; st[6] = ph[3];  // profiling build: overwrites status/n_ids/bitmap flag
#endif
        }
        __syncthreads();
    }
}


// =======================================================================================
// Exact replay of tie-affected queries ("strict ties").
//
// With two EQUAL f32 distances the reference's choice depends on the sift history of Rust's
// std::collections::BinaryHeap.  Queries flagged status 2 by the main kernel are
// re-run here with both heaps (candidate_points on -dist, return_points on +dist) emulated literally:
// push = append + sift_up, pop = swap-remove + sift_down_to_bottom + sift_up, into_sorted_vec = repeated
// swap + sift_down_range.  A sift only ever touches one root-to-leaf path, so the wave performs each heap
// operation cooperatively: lanes load the ancestors (push) or the 62 descendants of the current node
// within 5 levels (pop), the path is chased with readlane, and the shifted entries are stored in
// parallel.  Heaps live in an L2-resident scratch slice of the workgroup (entries = {key f32, id u32}),
// accessed with L1-bypassing relaxed agent-scope atomics.
// =======================================================================================
typedef unsigned long long hent_t;
__device__ __forceinline__ hent_t hmake(float key, uint32_t id) { return ((hent_t)__float_as_uint(key) << 32) | id; }
__device__ __forceinline__ float hkey(hent_t e) { return __uint_as_float((uint32_t)(e >> 32)); }
__device__ __forceinline__ uint32_t hid(hent_t e) { return (uint32_t)e; }
// A heap array: entries [0, lds_cap) live in LDS (the top levels, touched by every operation), the rest in
// the workgroup's global scratch slice (L1-bypassing relaxed agent-scope accesses, served by the L2).
struct HeapMem {
    hent_t* lds;
    hent_t* glb;
    uint32_t lds_cap;
};
__device__ __forceinline__ hent_t hload(const HeapMem& H, uint32_t i) {
    if (i < H.lds_cap) return H.lds[i];
    return __hip_atomic_load(H.glb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void hstore(const HeapMem& H, uint32_t i, hent_t v) {
    if (i < H.lds_cap) H.lds[i] = v;
    else __hip_atomic_store(H.glb + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// orders the stores of one heap operation before the loads of the next (single wave: LDS is in order,
// global stores are waited for)
__device__ __forceinline__ void hfence() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ hent_t readlane_h(hent_t v, int lane) {
    const uint32_t lo = readlane_u((uint32_t)v, lane), hi = readlane_u((uint32_t)(v >> 32), lane);
    return ((hent_t)hi << 32) | lo;
}

// BinaryHeap::push: data.push(item); sift_up(0, old_len): while pos > 0 { if elt <= data[parent] break; move parent down }
__device__ __forceinline__ void heap_push(const HeapMem& H, uint32_t& len, hent_t item, int lane) {
    const uint32_t pos = len;
    len += 1;
    // ancestor j of pos is ((pos+1) >> j) - 1
    const uint32_t p1 = lane < 32 ? (pos + 1u) >> lane : 0u;
    const bool anc = lane >= 1 && p1 >= 1u;
    const hent_t e = anc ? hload(H, p1 - 1u) : 0ull;
    const unsigned long long am = __ballot(anc);
    const unsigned long long stop = __ballot(anc && hkey(item) <= hkey(e));
    const int depth = (int)popc64(am);                       // ancestors are lanes 1..depth
    const int moved = stop != 0ull ? ctz64(stop) - 1 : depth;  // how many ancestors move down one step
    if (lane >= 1 && lane <= moved) hstore(H, ((pos + 1u) >> (lane - 1)) - 1u, e);
    if (lane == 0) hstore(H, ((pos + 1u) >> moved) - 1u, item);
    hfence();
}

// Greater-child path from the root of H[0..end): at a node with two children take the right one when
// data[left] <= data[right], with only a left child take it.  Returns the number m of path nodes below the
// root; lane j (1..m) receives the j-th node's index and entry, lane 0 index 0.
__device__ __forceinline__ int heap_chase(const HeapMem& H, uint32_t end, int lane, uint32_t& my_pos, hent_t& my_ent) {
    int m = 0;
    uint32_t p = 0;
    my_pos = 0;
    my_ent = 0ull;
    for (;;) {
        // lane L <= 62 holds the descendant of p at relative depth t = floor(log2(L+1)), offset L+1-2^t
        const uint32_t L1 = (uint32_t)lane + 1u;
        const uint32_t t = 31u - (uint32_t)__clz((int)L1);
        const unsigned long long idx64 = (((unsigned long long)p + 1ull) << t) - 1ull + (unsigned long long)(L1 - (1u << t));
        const bool valid = lane < 63 && idx64 < (unsigned long long)end;
        const uint32_t idx = (uint32_t)idx64;
        const hent_t e = valid ? hload(H, idx) : 0ull;
        const unsigned long long vm = __ballot(valid);
        int cur = 0;
        bool bottom = false;
        for (int step = 0; step < 5; ++step) {
            const int l = 2 * cur + 1, r = l + 1;
            if (((vm >> l) & 1ull) == 0ull) { bottom = true; break; }
            int nxt = l;
            if ((vm >> r) & 1ull) {
                const float kl = hkey(readlane_h(e, l)), kr = hkey(readlane_h(e, r));
                nxt = (kl <= kr) ? r : l;
            }
            m += 1;
            const hent_t en = readlane_h(e, nxt);
            const uint32_t ip = readlane_u(idx, nxt);
            if (lane == m) { my_pos = ip; my_ent = en; }
            cur = nxt;
        }
        if (bottom) break;
        p = readlane_u(idx, cur);
    }
    return m;
}

// BinaryHeap::pop: swap-remove the root with the last element, sift_down_to_bottom(0), then sift_up.
__device__ __forceinline__ hent_t heap_pop(const HeapMem& H, uint32_t& len, int lane) {
    const hent_t last = hload(H, len - 1u);
    len -= 1;
    if (len == 0u) return last;
    const hent_t root = hload(H, 0u);
    uint32_t my_pos;
    hent_t my_ent;
    const int m = heap_chase(H, len, lane, my_pos, my_ent);
    // the element re-inserted at the bottom climbs back while it is > the entry above it
    const unsigned long long le = __ballot(lane >= 1 && lane <= m && hkey(last) <= hkey(my_ent));
    const int jstar = le != 0ull ? 63 - __clzll((long long)le) : 0;
    const uint32_t prev_pos = (uint32_t)__shfl_up((int)my_pos, 1);
    if (lane >= 1 && lane <= jstar) hstore(H, prev_pos, my_ent);
    if (lane == jstar) hstore(H, my_pos, last);
    hfence();
    return root;
}

// sift_down_range(0, end) of into_sorted_vec: descend along the greater child, stop as soon as elt >= child
__device__ __forceinline__ void heap_sift_down_range(const HeapMem& H, uint32_t end, int lane) {
    const hent_t elt = hload(H, 0u);
    uint32_t my_pos;
    hent_t my_ent;
    const int m = heap_chase(H, end, lane, my_pos, my_ent);
    const unsigned long long ge = __ballot(lane >= 1 && lane <= m && hkey(elt) >= hkey(my_ent));
    const int nshift = ge != 0ull ? ctz64(ge) - 1 : m;
    const uint32_t prev_pos = (uint32_t)__shfl_up((int)my_pos, 1);
    if (lane >= 1 && lane <= nshift) hstore(H, prev_pos, my_ent);
    if (lane == nshift) hstore(H, my_pos, elt);
    hfence();
}

struct ExactArgs {
    hent_t* heaps;        // [gridDim.x][heap_stride]: return_points (ef + 2 entries) then candidate_points
    uint64_t heap_stride; // entries per workgroup
    uint32_t cand_cap;    // capacity of candidate_points
    uint32_t r_lds_cap;   // entries of return_points kept in LDS
    uint32_t cand_lds;    // entries of candidate_points kept in LDS
};

template <int METRIC>
__global__ __launch_bounds__(64) void hnsw_search_exact_kernel(DeviceIndexView ix, SearchArgs a, ExactArgs x) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float4* tile = reinterpret_cast<float4*>(lds_raw);
    uint32_t* ids_lds = reinterpret_cast<uint32_t*>(lds_raw + TILE_BYTES);
    const int lane = (int)threadIdx.x;
    uint32_t* bitmap = a.bitmap + (size_t)blockIdx.x * a.bitmap_words;
    // LDS: [tile][ids][R heap: ef+2 entries][candidate heap: first cand_lds entries]
    hent_t* lds_heaps = reinterpret_cast<hent_t*>(lds_raw + TILE_BYTES + IDS_BYTES);
    hent_t* glb = x.heaps + (size_t)blockIdx.x * x.heap_stride;
    const uint32_t r_lds = a.ef + 2u <= x.r_lds_cap ? a.ef + 2u : x.r_lds_cap;
    const HeapMem R{lds_heaps, glb, r_lds};
    const HeapMem Cq{lds_heaps + r_lds, glb + (a.ef + 2u), x.cand_lds};

    for (;;) {
        uint32_t wi = 0;
        if (lane == 0) wi = atomicAdd(a.work_counter, 1u);
        wi = readlane_u(wi, 0);
        if (wi >= a.nq) break;
        const uint32_t q = a.qlist[wi];
        const qptr_t qrow = (qptr_t)(a.queries + (size_t)q * ix.row_stride);
        for (uint32_t i = (uint32_t)lane; i < a.bitmap_words; i += 64) bitmap[i] = 0u;
        __syncthreads();
        uint32_t n_dist = 0, n_expand = 0, n_ids = 0, status = 0;

        // ---- descent, identical to the main kernel (src/hnsw.rs:1506-1529)
        uint32_t pivot = ix.entry;
        if (lane == 0) ids_lds[0] = pivot;
        __syncthreads();
        float dcur = readlane_f(batch_dist<METRIC>(ix.vec, ix.row_stride, qrow, tile, ids_lds, 1u, lane), 0);
        n_dist += 1;
        for (int layer = (int)ix.entry_level; layer >= 1; --layer) {
            uint32_t b = 0, e = 0;
            if ((uint32_t)layer <= ix.n_up_layers) {
                const uint32_t* ptr = ix.up_ptr + (size_t)(layer - 1) * ((size_t)ix.n + 1);
                b = ptr[pivot];
                e = ptr[pivot + 1];
            }
            n_expand += 1;
            n_ids += e - b;
            float best = INFINITY;
            uint32_t best_id = pivot;
            for (uint32_t base = b; base < e; base += 64) {
                const uint32_t j = base + (uint32_t)lane;
                const bool valid = j < e;
                const uint32_t id = valid ? ix.up_ids[j] : 0u;
                const uint32_t nf = e - base < 64u ? e - base : 64u;
                __syncthreads();
                if (valid) ids_lds[lane] = id;
                __syncthreads();
                const float dl = batch_dist<METRIC>(ix.vec, ix.row_stride, qrow, tile, ids_lds, nf, lane);
                n_dist += nf;
                const float m = wave_min(dl);
                const unsigned long long eq = __ballot(valid && dl == m);
                if (eq != 0ull && m < best) { best = m; best_id = readlane_u(id, ctz64(eq)); }
            }
            if (best < dcur) { dcur = best; pivot = best_id; }
        }

        // ---- search_layer with literal heaps (src/hnsw.rs:938-1063)
        uint32_t lenR = 0, lenC = 0;
        if (lane == 0) (void)visit_bitmap(bitmap, pivot);
        __syncthreads();
        heap_push(Cq, lenC, hmake(-dcur, pivot), lane);
        heap_push(R, lenR, hmake(dcur, pivot), lane);
        while (lenC > 0u) {
            const hent_t c = heap_pop(Cq, lenC, lane);                      // :971
            float worst = hkey(hload(R, 0u));                               // :973 peek
            if (-hkey(c) > worst) break;                                    // :981-993
            n_expand += 1;
            const uint32_t* nrow = ix.nbr0 + (size_t)hid(c) * ix.deg_stride;
            const uint32_t nbatch = (ix.deg_stride + 63u) >> 6;
            for (uint32_t bi = 0; bi < nbatch; ++bi) {
                const uint32_t j = bi * 64u + (uint32_t)lane;
                const uint32_t id = j < ix.deg_stride ? nrow[j] : EMPTY_SLOT;
                const bool valid = id != EMPTY_SLOT;
                const unsigned long long vm = __ballot(valid);
                if (vm == 0ull) break;
                n_ids += popc64(vm);
                const bool fresh = valid && visit_bitmap(bitmap, id) == 1;  // :1016-1017
                const unsigned long long fm = __ballot(fresh);
                const uint32_t nf = popc64(fm);
                if (nf == 0u) continue;
                n_dist += nf;
                __syncthreads();
                if (fresh) ids_lds[popc64(fm & lanemask_lt(lane))] = id;
                __syncthreads();
                const uint32_t idc = (uint32_t)lane < nf ? ids_lds[lane] : 0u;
                const float de = batch_dist<METRIC>(ix.vec, ix.row_stride, qrow, tile, ids_lds, nf, lane);
                for (uint32_t r = 0; r < nf; ++r) {                          // list order (:1013)
                    const float xd = readlane_f(de, (int)r);
                    if (xd < worst || lenR < a.ef) {                        // :1028
                        const uint32_t xi = readlane_u(idc, (int)r);
                        if (lenC >= x.cand_cap) { status = 1; break; }
                        heap_push(Cq, lenC, hmake(-xd, xi), lane);          // :1035-1036
                        heap_push(R, lenR, hmake(xd, xi), lane);            // :1038
                        if (lenR > a.ef) (void)heap_pop(R, lenR, lane);     // :1051-1053
                        worst = hkey(hload(R, 0u));
                    }
                }
                if (status != 0) break;
            }
            if (status != 0) break;
        }

        // ---- into_sorted_vec (:1544) + truncate (:1547)
        if (status == 0) {
            uint32_t end = lenR;
            while (end > 1u) {
                end -= 1;
                {
                    const hent_t r0 = hload(R, 0u), re = hload(R, end);
                    hfence();
                    if (lane == 0) {
                        hstore(R, 0u, re);
                        hstore(R, end, r0);
                    }
                }
                hfence();
                heap_sift_down_range(R, end, lane);
            }
            const uint32_t cnt = lenR < a.k ? lenR : a.k;
            for (uint32_t j = (uint32_t)lane; j < a.k; j += 64) {
                const size_t o = (size_t)q * a.k + j;
                if (j < cnt) {
                    const hent_t e = hload(R, j);
                    const uint32_t flat = hid(e);
                    uint32_t l = 0;
                    while (l + 1 < NB_LAYER_MAX && flat >= ix.layer_offset[l + 1]) ++l;
                    a.out_ids[o] = ix.origin_id[flat];
                    a.out_dists[o] = hkey(e);
                    if (a.out_layer) a.out_layer[o] = (uint8_t)l;
                    if (a.out_rank) a.out_rank[o] = (int32_t)(flat - ix.layer_offset[l]);
                } else {
                    a.out_ids[o] = 0ull;
                    a.out_dists[o] = 0.f;
                    if (a.out_layer) a.out_layer[o] = 0;
                    if (a.out_rank) a.out_rank[o] = 0;
                }
            }
            if (lane == 0) a.out_counts[q] = cnt;
        } else if (lane == 0) {
            atomicAdd(a.overflow_count, 1u);
        }
        if (lane == 0) {
            uint32_t* st = a.stats + (size_t)q * 8;
            st[0] = n_dist; st[1] = n_expand; st[2] = n_ids; st[3] = status == 0 ? 3u : 4u;  // 3 = exact replay done
        }
        __syncthreads();
    }
}

// queries [nq][d] -> [nq][row_stride] zero padded
__global__ void pad_queries_kernel(const float* __restrict__ src, float* __restrict__ dst, uint32_t nq, uint32_t d,
                                   uint32_t row_stride) {
    const size_t total = (size_t)nq * row_stride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / row_stride), c = (uint32_t)(i % row_stride);
        dst[i] = c < d ? src[(size_t)r * d + c] : 0.f;
    }
}

template <int METRIC>
__global__ void eval_pairs_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                  uint32_t n, uint32_t row_stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = dist_row<METRIC>(reinterpret_cast<const float4*>(a + (size_t)i * row_stride),
                              reinterpret_cast<const float4*>(b + (size_t)i * row_stride), row_stride >> 2);
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
            return ERR_DEVICE;                                                                 \
        }                                                                                      \
    } while (0)

using KernelFn = void (*)(DeviceIndexView, SearchArgs);

template <int METRIC, int TABLE>
KernelFn pick_slots(int slots) {
    switch (slots) {
        case 1: return hnsw_search_kernel<METRIC, 1, TABLE>;
        case 2: return hnsw_search_kernel<METRIC, 2, TABLE>;
        case 4: return hnsw_search_kernel<METRIC, 4, TABLE>;
        case 8: return hnsw_search_kernel<METRIC, 8, TABLE>;
        default: return hnsw_search_kernel<METRIC, 16, TABLE>;
    }
}
template <int TABLE>
KernelFn pick_metric(int metric, int slots) {
    switch (metric) {
        case DIST_L2: return pick_slots<DIST_L2, TABLE>(slots);
        case DIST_COSINE: return pick_slots<DIST_COSINE, TABLE>(slots);
        case DIST_DOT: return pick_slots<DIST_DOT, TABLE>(slots);
        default: return pick_slots<DIST_L1, TABLE>(slots);
    }
}
KernelFn pick_kernel(int metric, int slots, int table) {
    switch (table) {
        case TABLE_LDS_CELL16: return pick_metric<TABLE_LDS_CELL16>(metric, slots);
        case TABLE_LDS_CELL32: return pick_metric<TABLE_LDS_CELL32>(metric, slots);
        default: return pick_metric<TABLE_GLOBAL_BITMAP>(metric, slots);
    }
}

uint32_t ceil_log2(uint64_t x) {
    uint32_t b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}

}  // namespace

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

DeviceIndex::~DeviceIndex() { release(); }

void DeviceIndex::release() {
    if (device_ >= 0) (void)hipSetDevice(device_);
    void** ptrs[] = {&d_vec_, &d_nbr0_, &d_up_ptr_, &d_up_ids_, &d_origin_, &d_qpad_, &d_ctrl_, &d_retry_[0], &d_retry_[1],
                     &d_stats_, &d_bitmap_, &d_tie_, &d_heaps_, &d_hostio_[0], &d_hostio_[1], &d_hostio_[2], &d_hostio_[3], &d_hostio_[4]};
    for (void** p : ptrs)
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (ev_start_) { (void)hipEventDestroy((hipEvent_t)ev_start_); ev_start_ = nullptr; }
    if (ev_stop_) { (void)hipEventDestroy((hipEvent_t)ev_stop_); ev_stop_ = nullptr; }
    ready_ = false;
}

int DeviceIndex::upload(const FlatIndex& x, int device, std::string& err) {
    if (const char* e = std::getenv("HNSWGPU_STRICT_TIES")) strict_ties_ = std::atoi(e) != 0;
    if (x.n == 0 || x.entry_flat == NO_POINT) { err = "cannot upload an empty index"; return ERR_EMPTY; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { err = "no HIP device visible (a gfx950 GPU is required; there is no CPU fallback)"; return ERR_DEVICE; }
    if (device < 0 || device >= ndev) { err = "bad device ordinal"; return ERR_ARG; }
    release();
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    num_cu_ = prop.multiProcessorCount;
    device_ = device;
    dist_ = x.dist;

    const uint64_t n = x.n, d = x.dimension;
    DeviceIndexView v{};
    v.n = (uint32_t)n;
    v.d = (uint32_t)d;
    v.row_stride = (uint32_t)((d + 31) / 32 * 32);  // 128-byte lines
    v.entry = x.entry_flat;
    v.entry_level = x.layer_of(x.entry_flat);
    v.search_layer = x.layer_to_search();
    for (unsigned l = 0; l <= NB_LAYER_MAX; ++l) v.layer_offset[l] = (uint32_t)x.layer_offset[l];

    // vectors, padded rows
    {
        std::vector<float> pad((size_t)n * v.row_stride, 0.f);
        for (uint64_t f = 0; f < n; ++f) std::memcpy(pad.data() + f * v.row_stride, x.vectors.data() + f * d, d * sizeof(float));
        HIP_TRY(hipMalloc(&d_vec_, pad.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(d_vec_, pad.data(), pad.size() * sizeof(float), hipMemcpyHostToDevice));
        bytes_ += pad.size() * sizeof(float);
    }
    // search-layer lists, fixed stride ("padded CSR": row_ptr is implicit, one aligned row per point)
    {
        const unsigned sl = v.search_layer;
        uint64_t maxdeg = 1;
        for (uint64_t f = 0; f < n; ++f)
            maxdeg = std::max<uint64_t>(maxdeg, x.nbr_ptr[f * NB_LAYER_MAX + sl + 1] - x.nbr_ptr[f * NB_LAYER_MAX + sl]);
        v.deg_stride = (uint32_t)((maxdeg + 15) / 16 * 16);
        std::vector<uint32_t> ell((size_t)n * v.deg_stride, EMPTY_SLOT);
        for (uint64_t f = 0; f < n; ++f) {
            uint64_t b = x.nbr_ptr[f * NB_LAYER_MAX + sl], e = x.nbr_ptr[f * NB_LAYER_MAX + sl + 1];
            std::memcpy(ell.data() + f * v.deg_stride, x.nbr_flat.data() + b, (e - b) * sizeof(uint32_t));
        }
        HIP_TRY(hipMalloc(&d_nbr0_, ell.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(d_nbr0_, ell.data(), ell.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        bytes_ += ell.size() * sizeof(uint32_t);
    }
    // upper layers (>= 1): CSR per layer over all flat ids (lists may exist above a point's own level)
    {
        unsigned top = 0;
        for (uint64_t f = 0; f < n; ++f)
            for (unsigned l = NB_LAYER_MAX - 1; l > top; --l)
                if (x.nbr_ptr[f * NB_LAYER_MAX + l + 1] > x.nbr_ptr[f * NB_LAYER_MAX + l]) { top = l; break; }
        v.n_up_layers = top;
        std::vector<uint32_t> ptr((size_t)std::max(1u, top) * (n + 1), 0u);
        std::vector<uint32_t> ids;
        for (unsigned l = 1; l <= top; ++l) {
            uint32_t* p = ptr.data() + (size_t)(l - 1) * (n + 1);
            for (uint64_t f = 0; f < n; ++f) {
                p[f] = (uint32_t)ids.size();
                uint64_t b = x.nbr_ptr[f * NB_LAYER_MAX + l], e = x.nbr_ptr[f * NB_LAYER_MAX + l + 1];
                ids.insert(ids.end(), x.nbr_flat.begin() + b, x.nbr_flat.begin() + e);
            }
            p[n] = (uint32_t)ids.size();
        }
        if (ids.empty()) ids.push_back(0);
        HIP_TRY(hipMalloc(&d_up_ptr_, ptr.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(d_up_ptr_, ptr.data(), ptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&d_up_ids_, ids.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(d_up_ids_, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        bytes_ += (ptr.size() + ids.size()) * sizeof(uint32_t);
    }
    HIP_TRY(hipMalloc(&d_origin_, n * sizeof(uint64_t)));
    HIP_TRY(hipMemcpy(d_origin_, x.origin_id.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice));
    bytes_ += n * sizeof(uint64_t);
    HIP_TRY(hipMalloc(&d_ctrl_, 64));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    ev_start_ = e0;
    ev_stop_ = e1;

    v.vec = static_cast<const float*>(d_vec_);
    v.nbr0 = static_cast<const uint32_t*>(d_nbr0_);
    v.up_ptr = static_cast<const uint32_t*>(d_up_ptr_);
    v.up_ids = static_cast<const uint32_t*>(d_up_ids_);
    v.origin_id = static_cast<const uint64_t*>(d_origin_);
    v_ = v;
    ready_ = true;
    return OK;
}

int DeviceIndex::ensure_workspace(uint64_t nq, uint64_t /*k*/, std::string& err) {
    const uint64_t qpad_need = nq * v_.row_stride * sizeof(float);
    if (qpad_need > qpad_cap_) {
        if (d_qpad_) (void)hipFree(d_qpad_);
        d_qpad_ = nullptr;
        HIP_TRY(hipMalloc(&d_qpad_, qpad_need));
        qpad_cap_ = qpad_need;
    }
    if (nq > tie_cap_) {
        if (d_tie_) (void)hipFree(d_tie_);
        d_tie_ = nullptr;
        HIP_TRY(hipMalloc(&d_tie_, nq * sizeof(uint32_t)));
        tie_cap_ = nq;
    }
    if (nq > retry_cap_) {
        for (int i = 0; i < 2; ++i) {
            if (d_retry_[i]) (void)hipFree(d_retry_[i]);
            d_retry_[i] = nullptr;
            HIP_TRY(hipMalloc(&d_retry_[i], nq * sizeof(uint32_t)));
        }
        retry_cap_ = nq;
    }
    if (nq * 8 * sizeof(uint32_t) > stats_cap_) {
        if (d_stats_) (void)hipFree(d_stats_);
        d_stats_ = nullptr;
        HIP_TRY(hipMalloc(&d_stats_, nq * 8 * sizeof(uint32_t)));
        stats_cap_ = nq * 8 * sizeof(uint32_t);
    }
    return OK;
}

int DeviceIndex::search_device(const float* d_queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef_arg,
                               uint64_t* d_out_ids, float* d_out_dists, uint8_t* d_out_layer, int32_t* d_out_rank,
                               uint32_t* d_out_counts, uint32_t* d_stats, void* stream_v, std::string& err) {
    if (!ready_) { err = "index is not resident on a device: call hnswgpu_upload first"; return ERR_DEVICE; }
    if (d != v_.d) { err = "query dimension differs from the index dimension"; return ERR_ARG; }
    if (nq == 0) { last_ms_ = 0; last_launches_ = 0; return OK; }
    if (!d_queries || !d_out_ids || !d_out_dists || !d_out_counts) { err = "null buffer"; return ERR_ARG; }
    if (k == 0) { err = "knbn must be > 0"; return ERR_ARG; }
    const uint64_t ef = std::max(ef_arg, k);  // src/hnsw.rs:1531
    if (ef > 1024) { err = "ef (= max(ef, knbn)) above 1024 is not supported by the register-resident result set"; return ERR_ARG; }
    if (nq > 0xFFFFFFF0ull) { err = "too many queries in one batch"; return ERR_ARG; }
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    HIP_TRY(hipSetDevice(device_));
    int rc = ensure_workspace(nq, k, err);
    if (rc != OK) return rc;
    uint32_t* stats = d_stats ? d_stats : static_cast<uint32_t*>(d_stats_);

    int slots = 1;
    while ((uint64_t)slots * 64 < ef) slots *= 2;

    HIP_TRY(hipEventRecord((hipEvent_t)ev_start_, stream));
    // pad queries to the row stride (tiny, stays on the launch stream)
    {
        const uint64_t total = nq * v_.row_stride;
        const int blocks = (int)std::min<uint64_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(pad_queries_kernel, dim3(blocks), dim3(256), 0, stream, d_queries,
                           static_cast<float*>(d_qpad_), (uint32_t)nq, v_.d, v_.row_stride);
    }

    // Visited-set sizing.  LDS per wavefront is what bounds occupancy, so the table is sized for the
    // typical query (ef x degree cells, ~2.4x the median number of visited points, measured); the few
    // per cent of queries that outgrow it start over on the HBM bitmap inside the same launch.
    const uint32_t idbits = std::max<uint32_t>(1u, ceil_log2(v_.n));
    const uint64_t expect = ef * std::min<uint64_t>(v_.deg_stride, 64);
    uint32_t tbits = std::min<uint32_t>(14u, std::max<uint32_t>(8u, ceil_log2(expect)));
    // ... then follows what the previous batches with the same ef measured (adapt_* below)
    if (adapt_ef_ == ef && adapt_tbits_ != 0) tbits = adapt_tbits_;
    bool env_forced = false;
    if (const char* e = std::getenv("HNSWGPU_HASH_BITS")) {  // tuning / test hook: initial table size
        int b = std::atoi(e);
        if (b >= 6 && b <= 14) { tbits = (uint32_t)b; env_forced = true; }
    }
    const uint32_t tbits_first = tbits;
    const size_t lds_fixed = TILE_BYTES + IDS_BYTES;
    int table = TABLE_LDS_CELL16;
    bool grown = false;

    uint32_t launches = 0;
    uint32_t work = (uint32_t)nq;
    uint32_t n_ties = 0;
    const uint32_t* qlist = nullptr;
    int pingpong = 0;
    SearchArgs last_args{};
    for (;;) {
        SearchArgs a{};
        size_t lds = lds_fixed;
        if (table != TABLE_GLOBAL_BITMAP) {
            uint32_t tb = std::min(tbits, idbits);  // a table with one cell per possible id never probes
            if (idbits - tb <= 11u) {
                table = TABLE_LDS_CELL16;
                a.restbits = idbits - tb;
                lds += (size_t)2 << tb;
            } else {
                table = TABLE_LDS_CELL32;
                lds += (size_t)4 << tb;
            }
            a.tbits = tb;
        }
        a.idbits = idbits;
        KernelFn fn = pick_kernel(dist_, slots, table);
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, lds));
        if (per_cu < 1) per_cu = 1;
        uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)per_cu * (uint64_t)num_cu_, work);
        a.queries = static_cast<const float*>(d_qpad_);
        a.qlist = qlist;
        a.nq = work;
        a.k = (uint32_t)k;
        a.ef = (uint32_t)ef;
        a.work_counter = static_cast<uint32_t*>(d_ctrl_);
        a.overflow_count = static_cast<uint32_t*>(d_ctrl_) + 1;
        a.retry_out = static_cast<uint32_t*>(d_retry_[pingpong]);
        a.out_ids = d_out_ids;
        a.out_dists = d_out_dists;
        a.out_layer = d_out_layer;
        a.out_rank = d_out_rank;
        a.out_counts = d_out_counts;
        a.stats = stats;
        {
            // HBM bitmaps for the in-launch fallback: one slice per workgroup, within a 4 GiB budget
            a.bitmap_words = (v_.n + 31) / 32;
            const uint64_t slice = (uint64_t)a.bitmap_words * sizeof(uint32_t);
            uint64_t blocks = std::min<uint64_t>(grid, std::max<uint64_t>(1, (4ull << 30) / slice));
            if (table == TABLE_GLOBAL_BITMAP) grid = (uint32_t)blocks;  // every workgroup needs one
            const uint64_t need = blocks * slice;
            if (need > bitmap_cap_) {
                if (d_bitmap_) (void)hipFree(d_bitmap_);
                d_bitmap_ = nullptr;
                bitmap_cap_ = 0;
                HIP_TRY(hipMalloc(&d_bitmap_, need));
                bitmap_cap_ = need;
            }
            a.bitmap = static_cast<uint32_t*>(d_bitmap_);
            a.bitmap_blocks = (uint32_t)blocks;
        }
        a.tie_list = strict_ties_ ? static_cast<uint32_t*>(d_tie_) : nullptr;
        HIP_TRY(hipMemsetAsync(d_ctrl_, 0, launches == 0 ? 32 : 16, stream));  // the tie list spans relaunches
        hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds, stream, v_, a);
        HIP_TRY(hipGetLastError());
        last_args = a;
        ++launches;
        uint32_t ctrl[5] = {0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(ctrl, d_ctrl_, 20, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        n_ties = ctrl[4];
        if (launches == 1 && table != TABLE_GLOBAL_BITMAP && !env_forced && nq >= 256) {
            // Table sizing feedback for the next batch: grow when more than ~1 query in 8 had to move to the
            // HBM bitmap, shrink when a half-size table would have overflowed for fewer than 1 in 32.
            uint32_t next = tbits_first;
            if ((uint64_t)ctrl[2] * 8 > nq && tbits_first < 14u) next = tbits_first + 1;
            else if ((uint64_t)ctrl[3] * 32 < nq && tbits_first > 8u) next = tbits_first - 1;
            adapt_ef_ = ef;
            adapt_tbits_ = next;
        }
        if (ctrl[1] == 0) break;
        // some queries visited more points than the table holds: rerun only those
        work = ctrl[1];
        qlist = static_cast<const uint32_t*>(d_retry_[pingpong]);
        pingpong ^= 1;
        if (table == TABLE_GLOBAL_BITMAP) { err = "internal error: bitmap visited set reported an overflow"; return ERR_DEVICE; }
        if (!grown && tbits < 14u) {
            tbits = std::min<uint32_t>(14u, tbits + 2u);
            grown = true;
        } else {
            table = TABLE_GLOBAL_BITMAP;
        }
    }
    last_ties_ = n_ties;
    if (strict_ties_ && n_ties > 0) {
        // Exact replay of the tie-affected queries with literal binary heaps (hnsw_search_exact_kernel).
        const uint64_t bm_slice = (uint64_t)last_args.bitmap_words * sizeof(uint32_t);
        const uint64_t cand_cap = v_.n;                                   // every point is accepted at most once
        const uint64_t heap_stride = ef + 2 + cand_cap;
        const uint64_t per_block = bm_slice + heap_stride * sizeof(hent_t);
        uint32_t grid = (uint32_t)std::min<uint64_t>(n_ties, std::max<uint64_t>(1, (2ull << 30) / per_block));
        grid = std::min<uint32_t>(grid, (uint32_t)num_cu_ * 4u);
        if ((uint64_t)grid * bm_slice > bitmap_cap_) {
            if (d_bitmap_) (void)hipFree(d_bitmap_);
            d_bitmap_ = nullptr;
            bitmap_cap_ = 0;
            HIP_TRY(hipMalloc(&d_bitmap_, (uint64_t)grid * bm_slice));
            bitmap_cap_ = (uint64_t)grid * bm_slice;
        }
        if ((uint64_t)grid * heap_stride * sizeof(hent_t) > heaps_cap_) {
            if (d_heaps_) (void)hipFree(d_heaps_);
            d_heaps_ = nullptr;
            heaps_cap_ = 0;
            HIP_TRY(hipMalloc(&d_heaps_, (uint64_t)grid * heap_stride * sizeof(hent_t)));
            heaps_cap_ = (uint64_t)grid * heap_stride * sizeof(hent_t);
        }
        SearchArgs a = last_args;
        a.qlist = static_cast<const uint32_t*>(d_tie_);
        a.nq = n_ties;
        a.bitmap = static_cast<uint32_t*>(d_bitmap_);
        a.tie_list = nullptr;
        ExactArgs x{};
        x.heaps = static_cast<hent_t*>(d_heaps_);
        x.heap_stride = heap_stride;
        x.cand_cap = (uint32_t)cand_cap;
        // heaps' top levels in LDS: up to ~56 KiB per workgroup (the exact replay is latency-, not occupancy-bound)
        const uint64_t lds_budget = 56 * 1024 - (TILE_BYTES + IDS_BYTES);
        x.r_lds_cap = (uint32_t)std::min<uint64_t>(ef + 2, lds_budget / 2 / sizeof(hent_t));
        x.cand_lds = (uint32_t)std::min<uint64_t>(cand_cap, (lds_budget - (uint64_t)x.r_lds_cap * sizeof(hent_t)) / sizeof(hent_t));
        HIP_TRY(hipMemsetAsync(d_ctrl_, 0, 8, stream));
        const size_t lds = TILE_BYTES + IDS_BYTES + ((size_t)x.r_lds_cap + x.cand_lds) * sizeof(hent_t);
        switch (dist_) {
            case DIST_L2: hipLaunchKernelGGL(hnsw_search_exact_kernel<DIST_L2>, dim3(grid), dim3(64), lds, stream, v_, a, x); break;
            case DIST_COSINE: hipLaunchKernelGGL(hnsw_search_exact_kernel<DIST_COSINE>, dim3(grid), dim3(64), lds, stream, v_, a, x); break;
            case DIST_DOT: hipLaunchKernelGGL(hnsw_search_exact_kernel<DIST_DOT>, dim3(grid), dim3(64), lds, stream, v_, a, x); break;
            default: hipLaunchKernelGGL(hnsw_search_exact_kernel<DIST_L1>, dim3(grid), dim3(64), lds, stream, v_, a, x); break;
        }
        HIP_TRY(hipGetLastError());
        ++launches;
        uint32_t ctrl2[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(ctrl2, d_ctrl_, 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (ctrl2[1] != 0) { err = "internal error: candidate heap overflow in the exact replay"; return ERR_DEVICE; }
    }
    HIP_TRY(hipEventRecord((hipEvent_t)ev_stop_, stream));
    HIP_TRY(hipEventSynchronize((hipEvent_t)ev_stop_));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, (hipEvent_t)ev_start_, (hipEvent_t)ev_stop_));
    last_ms_ = ms;
    last_launches_ = launches;
    return OK;
}

int DeviceIndex::search_host(const float* queries, uint64_t nq, uint64_t d, uint64_t k, uint64_t ef, uint64_t* out_ids,
                             float* out_dists, uint8_t* out_layer, int32_t* out_rank, uint32_t* out_counts,
                             std::string& err) {
    if (!ready_) { err = "index is not resident on a device: call hnswgpu_upload first"; return ERR_DEVICE; }
    if (nq == 0) return OK;
    if (!queries || !out_ids || !out_dists || !out_counts) { err = "null buffer"; return ERR_ARG; }
    if (d != v_.d) { err = "query dimension differs from the index dimension"; return ERR_ARG; }
    HIP_TRY(hipSetDevice(device_));
    if (nq * d > hostio_cap_q_ || nq * k > hostio_cap_k_ || nq > hostio_cap_n_) {
        for (auto& p : d_hostio_) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
        hostio_cap_q_ = hostio_cap_k_ = hostio_cap_n_ = 0;
        HIP_TRY(hipMalloc(&d_hostio_[0], nq * d * sizeof(float)));
        HIP_TRY(hipMalloc(&d_hostio_[1], nq * k * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&d_hostio_[2], nq * k * sizeof(float)));
        HIP_TRY(hipMalloc(&d_hostio_[3], nq * k * (sizeof(int32_t) + 1)));
        HIP_TRY(hipMalloc(&d_hostio_[4], nq * sizeof(uint32_t)));
        hostio_cap_q_ = nq * d;
        hostio_cap_k_ = nq * k;
        hostio_cap_n_ = nq;
    }
    float* dq = static_cast<float*>(d_hostio_[0]);
    uint64_t* dids = static_cast<uint64_t*>(d_hostio_[1]);
    float* ddist = static_cast<float*>(d_hostio_[2]);
    int32_t* drank = static_cast<int32_t*>(d_hostio_[3]);
    uint8_t* dlayer = reinterpret_cast<uint8_t*>(drank + nq * k);
    uint32_t* dcnt = static_cast<uint32_t*>(d_hostio_[4]);
    HIP_TRY(hipMemcpy(dq, queries, nq * d * sizeof(float), hipMemcpyHostToDevice));
    int rc = search_device(dq, nq, d, k, ef, dids, ddist, dlayer, drank, dcnt, nullptr, nullptr, err);
    if (rc != OK) return rc;
    HIP_TRY(hipMemcpy(out_ids, dids, nq * k * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_dists, ddist, nq * k * sizeof(float), hipMemcpyDeviceToHost));
    if (out_layer) HIP_TRY(hipMemcpy(out_layer, dlayer, nq * k, hipMemcpyDeviceToHost));
    if (out_rank) HIP_TRY(hipMemcpy(out_rank, drank, nq * k * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_counts, dcnt, nq * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return OK;
}

int eval_distances_device(int dist, const float* a, const float* b, uint64_t n, uint64_t d, float* out, std::string& err) {
    if (n == 0) return OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { err = "no HIP device visible"; return ERR_DEVICE; }
    const uint32_t rs = (uint32_t)((d + 31) / 32 * 32);
    std::vector<float> pa((size_t)n * rs, 0.f), pb((size_t)n * rs, 0.f);
    for (uint64_t i = 0; i < n; ++i) {
        std::memcpy(pa.data() + i * rs, a + i * d, d * sizeof(float));
        std::memcpy(pb.data() + i * rs, b + i * d, d * sizeof(float));
    }
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&da, pa.size() * sizeof(float)));
    HIP_TRY(hipMalloc(&db, pb.size() * sizeof(float)));
    HIP_TRY(hipMalloc(&dout, n * sizeof(float)));
    HIP_TRY(hipMemcpy(da, pa.data(), pa.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, pb.data(), pb.size() * sizeof(float), hipMemcpyHostToDevice));
    const int blocks = (int)((n + 63) / 64);
    switch (dist) {
        case DIST_L2: hipLaunchKernelGGL(eval_pairs_kernel<DIST_L2>, dim3(blocks), dim3(64), 0, 0, da, db, dout, (uint32_t)n, rs); break;
        case DIST_COSINE: hipLaunchKernelGGL(eval_pairs_kernel<DIST_COSINE>, dim3(blocks), dim3(64), 0, 0, da, db, dout, (uint32_t)n, rs); break;
        case DIST_DOT: hipLaunchKernelGGL(eval_pairs_kernel<DIST_DOT>, dim3(blocks), dim3(64), 0, 0, da, db, dout, (uint32_t)n, rs); break;
        default: hipLaunchKernelGGL(eval_pairs_kernel<DIST_L1>, dim3(blocks), dim3(64), 0, 0, da, db, dout, (uint32_t)n, rs); break;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    return OK;
}

}  // namespace hnswgpu
