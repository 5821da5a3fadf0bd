// gather_bw.hip -- what HBM delivers for the access pattern of the search path: random rows of `row_bytes` bytes
// (default 512 = one d=128 f32 vector) gathered from a table far larger than the caches, 4 lanes per row with 16-byte
// loads exactly like batch_dist's G = 4 path, nothing else in the loop.  The number is the practical ceiling that
// roofline.frac (priced against the 8 TB/s datasheet peak) should be read against.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bw.hip -o /tmp/gather_bw && /tmp/gather_bw [rows_millions] [row_bytes] [waves_per_cu]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// each wave: `iters` rounds of 16 rows (4 lanes per row, chunks of 16 bytes interleaved over the 4 lanes), ROUNDS rounds in flight
template <int CHUNKS, int ROUNDS, bool RANDOM>
__global__ __launch_bounds__(64) void gather_kernel(const float4* __restrict__ table, uint32_t n_rows, uint32_t iters, float* __restrict__ sink) {
    const uint32_t lane = threadIdx.x, g = lane >> 2, sub = lane & 3u;
    float acc = 0.f;
    for (uint32_t it = 0; it < iters; it += ROUNDS) {
        float4 v[ROUNDS][CHUNKS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t seq = (blockIdx.x * iters + it + r) * 16u + g;
            const uint32_t row = (RANDOM ? mix(seq + 0x9e3779b9u) : seq) % n_rows;
            const float4* p = table + (size_t)row * (CHUNKS * 4) + sub;
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c) v[r][c] = p[c * 4];
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c) acc += v[r][c].x + v[r][c].y + v[r][c].z + v[r][c].w;
    }
    if (acc == 12345.678f) sink[0] = acc;  // keeps the loads alive
}

template <int CHUNKS, int ROUNDS, bool RANDOM = true>
int run(const float4* table, uint32_t n_rows, int waves_per_cu, int num_cu, float* sink) {
    const uint32_t grid = (uint32_t)(num_cu * waves_per_cu);
    // random: 2048 rounds per wave; sequential: one pass over the table (no row is read twice)
    const uint32_t iters = RANDOM ? 2048u : std::max(2u, n_rows / (grid * 16u)) / ROUNDS * ROUNDS;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gather_kernel<CHUNKS, ROUNDS, RANDOM>), dim3(grid), dim3(64), 0, 0, table, n_rows, 64u, sink);  // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((gather_kernel<CHUNKS, ROUNDS, RANDOM>), dim3(grid), dim3(64), 0, 0, table, n_rows, iters, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)grid * iters * 16.0 * CHUNKS * 64.0;
    std::printf("%s rows of %4d B  waves/CU %2d  rounds in flight %d (%2d KB per wave)  %8.3f ms  %7.1f GB/s  (%.3f of 8000)\n", RANDOM ? "random    " : "sequential", CHUNKS * 64, waves_per_cu,
                ROUNDS, ROUNDS * CHUNKS, best, bytes / best * 1e-6, bytes / best * 1e-6 / 8000.0);
    return 0;
}

int main(int argc, char** argv) {
    const uint32_t rows_m = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 4;   // table of rows_m Mi rows
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int num_cu = prop.multiProcessorCount;
    const size_t max_row = 512;
    const uint32_t n_rows = rows_m << 20;
    float4* table = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc(&table, (size_t)n_rows * max_row));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(table, 0, (size_t)n_rows * max_row));
    std::printf("%s, %d CUs, table %u Mi rows x up to 512 B = %.1f GB\n", prop.name, num_cu, rows_m, (double)n_rows * max_row * 1e-9);
    for (int w : {8, 16, 20, 32}) {
        if (run<8, 1>(table, n_rows, w, num_cu, sink)) return 1;   // 512-byte rows, one round of 16 rows per wave in flight (the search kernel's shape)
        if (run<8, 2>(table, n_rows, w, num_cu, sink)) return 1;
    }
    for (int w : {16, 32}) {
        if (run<2, 4>(table, n_rows * 4u, w, num_cu, sink)) return 1;   // 128-byte rows (d = 25 padded to 32)
        if (run<4, 2>(table, n_rows * 2u, w, num_cu, sink)) return 1;   // 256-byte rows
    }
    // streaming reference: the same loads, consecutive rows
    for (int w : {16, 32})
        if (run<8, 2, false>(table, n_rows, w, num_cu, sink)) return 1;
    return 0;
}
