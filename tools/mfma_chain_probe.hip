// mfma_chain_probe.hip -- does D = A x 1 + C on the matrix pipe reproduce the reference's left-to-right f32 (f64) sum bit for bit?
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_chain_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// v_mfma_f32_16x16x4_f32: D[i][j] = C[i][j] + sum_k A[i][k] B[k][j]; A[i][k] sits in lane 16 k + i, B[k][j] in lane 16 k + j,
// D[4 (l / 16) + v][l % 16] in register v of lane l.  With B = 1 every column holds the row's sum; if the hardware adds the four
// products one after the other (k = 0, 1, 2, 3), each rounded to f32, a chain of such instructions IS `Iterator::sum` over the row.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));

// terms[row][e], 16 rows, n terms (multiple of 4); out[row] = chain sum
__global__ void chain_f32(const float* terms, int n, float* out) {
    const int lane = threadIdx.x, i = lane & 15, k = lane >> 4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < n / 4; ++m) {
        const float a = terms[i * n + 4 * m + k];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1.0f, acc, 0, 0, 0);
    }
    // row r = 4 (l / 16) + v, any column
    if ((lane & 15) == 0)
        for (int v = 0; v < 4; ++v) out[4 * (lane >> 4) + v] = acc[v];
}
__global__ void chain_f64(const double* terms, int n, double* out) {
    const int lane = threadIdx.x, i = lane & 15, k = lane >> 4;
    d4 acc = {0., 0., 0., 0.};
    for (int m = 0; m < n / 4; ++m) {
        const double a = terms[i * n + 4 * m + k];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, 1.0, acc, 0, 0, 0);
    }
    // every (lane, register): 256 values, the host works the layout out
    for (int v = 0; v < 4; ++v) out[lane * 4 + v] = acc[v];
}
// throughput: W waves per block, each running `reps` chains of n/4 dependent MFMAs (or n dependent v_add_f32)
__global__ void time_mfma(int n4, int reps, float* sink) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f;
    for (int r = 0; r < reps; ++r)
        for (int m = 0; m < n4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1.0f, acc, 0, 0, 0);
    if (acc[0] == 123.f) sink[0] = acc[1];
}
__global__ void time_mfma64(int n4, int reps, float* sink) {
    d4 acc = {0., 0., 0., 0.};
    double a = threadIdx.x * 1e-3;
    for (int r = 0; r < reps; ++r)
        for (int m = 0; m < n4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, 1.0, acc, 0, 0, 0);
    if (acc[0] == 123.) sink[0] = (float)acc[1];
}
__global__ void time_valu(int n, int reps, float* sink) {
    float acc = 0.f;
    float a = threadIdx.x * 1e-3f;
    for (int r = 0; r < reps; ++r)
#pragma unroll 16
        for (int m = 0; m < n; ++m) { acc = acc + a; asm volatile("" : "+v"(acc)); }
    if (acc == 123.f) sink[0] = acc;
}

template <typename T>
static T host_chain(const T* t, int n) {
    volatile T s = 0;
    for (int e = 0; e < n; ++e) s = s + t[e];
    return s;
}

int main() {
    const int n = 128;
    std::mt19937_64 rng(7);
    int bad32 = 0, bad64 = 0, cases = 0;
    float *dt, *dout;
    double *dt64, *dout64;
    hipMalloc(&dt, 16 * n * 4); hipMalloc(&dout, 16 * 4);
    hipMalloc(&dt64, 16 * n * 8); hipMalloc(&dout64, 256 * 8);
    for (int rep = 0; rep < 400; ++rep) {
        std::vector<float> t(16 * n);
        std::vector<double> t64(16 * n);
        const int kind = rep % 8;
        for (int r = 0; r < 16; ++r)
            for (int e = 0; e < n; ++e) {
                std::uniform_real_distribution<double> u(0., 1.);
                double v;
                switch (kind) {
                    case 0: { double x = u(rng) - u(rng); v = x * x; break; }                 // squared differences (DistL2)
                    case 1: v = (u(rng) - 0.5) * std::ldexp(1.0, (int)(u(rng) * 60) - 30); break;  // signed, wide exponent range (DistDot)
                    case 2: v = std::ldexp(u(rng), -140 + (int)(u(rng) * 20)); break;              // subnormal terms
                    case 3: v = (e % 7 == 0) ? std::ldexp(u(rng), 100) : std::ldexp(u(rng), -20); break;  // absorption
                    case 4: v = (e & 1) ? u(rng) : -u(rng); break;                                 // cancellation
                    case 5: v = std::ldexp(1.0 + std::ldexp((double)(rng() & 0x7FFFFF), -23), -126 - (int)(rng() % 3)); break;  // around FLT_MIN
                    case 6: v = std::ldexp(u(rng), 120); break;                                    // towards overflow
                    default: { double x = (u(rng) - u(rng)); v = std::fabs(x); break; }           // DistL1
                }
                t[r * n + e] = (float)v;
                t64[r * n + e] = (double)((float)v) * (double)((float)u(rng));                   // products widened to f64 (DistCosine)
            }
        hipMemcpy(dt, t.data(), t.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dt64, t64.data(), t64.size() * 8, hipMemcpyHostToDevice);
        chain_f32<<<1, 64>>>(dt, n, dout);
        chain_f64<<<1, 64>>>(dt64, n, dout64);
        float o[16]; double o64[16]; double all64[256];
        hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
        hipMemcpy(all64, dout64, sizeof all64, hipMemcpyDeviceToHost);
        // f64 layout: find, once, which row every (lane, register) holds
        static int row_of[256]; static bool have_layout = false;
        if (!have_layout) {
            have_layout = true;
            for (int s = 0; s < 256; ++s) {
                row_of[s] = -1;
                for (int r = 0; r < 16; ++r) { double w = host_chain<double>(&t64[r * n], n); if (memcmp(&w, &all64[s], 8) == 0) row_of[s] = r; }
            }
            printf("f64 D layout (row held by register v of lane l):\n");
            for (int l = 0; l < 64; l += 1) if (l % 16 == 0 || l % 16 == 1 || l % 16 == 15) printf("  lane %2d: %2d %2d %2d %2d\n", l, row_of[4 * l], row_of[4 * l + 1], row_of[4 * l + 2], row_of[4 * l + 3]);
        }
        for (int r = 0; r < 16; ++r) { o64[r] = 0; for (int s = 0; s < 256; ++s) if (row_of[s] == r) { o64[r] = all64[s]; break; } }
        for (int r = 0; r < 16; ++r) {
            const float want = host_chain<float>(&t[r * n], n);
            const double want64 = host_chain<double>(&t64[r * n], n);
            uint32_t a, b; memcpy(&a, &want, 4); memcpy(&b, &o[r], 4);
            uint64_t a6, b6; memcpy(&a6, &want64, 8); memcpy(&b6, &o64[r], 8);
            ++cases;
            if (a != b) { if (bad32 < 8) printf("f32 kind %d row %d: host %a (%08x) mfma %a (%08x)\n", kind, r, want, a, o[r], b); ++bad32; }
            if (a6 != b6) { if (bad64 < 8) printf("f64 kind %d row %d: host %a mfma %a\n", kind, r, want64, o64[r]); ++bad64; }
        }
    }
    printf("chains compared: %d, f32 mismatches %d, f64 mismatches %d\n", cases, bad32, bad64);
    // throughput with 4 waves per SIMD on every CU (16 waves per CU: 4 blocks of 256 threads)
    float* sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_cu : {4, 16}) {
        const int blocks = 256 * waves_per_cu / 4;
        float ms;
        time_mfma<<<blocks, 256>>>(32, 10, sink); hipDeviceSynchronize();
        hipEventRecord(e0); time_mfma<<<blocks, 256>>>(32, 2000, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%2d waves/CU: 32-MFMA chain  %.1f ns per chain per wave-slot (%.0f cycles @2.4GHz per chain, per SIMD share)\n", waves_per_cu, ms * 1e6 / 2000, ms * 1e6 / 2000 * 2.4);
        time_mfma64<<<blocks, 256>>>(32, 10, sink); hipDeviceSynchronize();
        hipEventRecord(e0); time_mfma64<<<blocks, 256>>>(32, 2000, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%2d waves/CU: 32-MFMA f64 chain %.1f ns per chain per wave-slot\n", waves_per_cu, ms * 1e6 / 2000);
        time_valu<<<blocks, 256>>>(128, 10, sink); hipDeviceSynchronize();
        hipEventRecord(e0); time_valu<<<blocks, 256>>>(128, 2000, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%2d waves/CU: 128-v_add chain %.1f ns per chain per wave-slot (%.0f cycles)\n", waves_per_cu, ms * 1e6 / 2000, ms * 1e6 / 2000 * 2.4);
    }
    return bad32 || bad64;
}
