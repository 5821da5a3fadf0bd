// dispatch_probe.hip -- how fast does the dispatcher hand workgroups to the CUs?  A kernel whose waves spin for a fixed time and leave:
// the launch lasts (time to place every workgroup) + spin.  hipcc --offload-arch=gfx950 -O2 tools/dispatch_probe.hip -o tools/dispatch_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(uint64_t ticks, uint32_t* sink) {
    extern __shared__ uint32_t lds[];
    const uint64_t t0 = wall_clock64();
    if (threadIdx.x == 0) lds[0] = (uint32_t)t0;
    while (wall_clock64() - t0 < ticks) {}
    if (lds[0] == 0x12345u && ticks == 77) sink[0] = 1;
}
int main() {
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int rate_khz = 0; CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    printf("wall clock %d kHz\n", rate_khz);
    struct Cfg { int grid, block, lds; double spin_us; };
    const Cfg cfgs[] = {
        {7168, 64, 2816, 0.0}, {7168, 64, 2816, 5.0}, {7168, 64, 0, 0.0}, {7168, 64, 64, 0.0},
        {3584, 128, 5632, 0.0}, {1792, 256, 11264, 0.0}, {1792, 256, 11264, 5.0}, {896, 512, 22528, 0.0}, {448, 1024, 45056, 0.0},
        {4096, 64, 9000, 0.0}, {1024, 256, 36000, 0.0}, {2048, 64, 2816, 0.0}, {1024, 64, 2816, 0.0}, {256, 64, 2816, 0.0},
        {20000, 64, 2816, 0.0}, {5000, 256, 11264, 0.0},
    };
    for (const Cfg& c : cfgs) {
        const uint64_t ticks = (uint64_t)(c.spin_us * rate_khz / 1000.0);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(spin, dim3(c.grid), dim3(c.block), c.lds, 0, ticks, sink);
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0.f;
        const int reps = 20;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(spin, dim3(c.grid), dim3(c.block), c.lds, 0, ticks, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        printf("grid %6d x %4d threads, %6d B LDS, spin %4.1f us: %7.2f us min, %7.2f us mean  -> %.0f workgroups/us, %.0f waves/us (after spin)\n", c.grid, c.block, c.lds, c.spin_us,
               best * 1e3, sum / reps * 1e3, c.grid / (best * 1e3 - c.spin_us), c.grid * (c.block / 64) / (best * 1e3 - c.spin_us));
    }
    return 0;
}
