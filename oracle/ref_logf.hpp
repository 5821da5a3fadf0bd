// ref_logf.hpp -- f32::ln as the reference computes it: Rust's f32::ln lowers to the platform libm's logf; on Linux that
// is glibc's logf (sysdeps/ieee754/flt-32/e_logf.c, the ARM "optimized routines" algorithm: 16-entry table of
// {1/c, log c}, degree-3 polynomial, everything in double, one final rounding).  Restated here so that the oracle does
// not depend on the build host's libm; `tests/test_oracle.py::test_ref_logf_is_the_hosts_logf` compares it with the
// host's logf on all 2 139 095 039 positive finite floats (0 mismatches on glibc 2.35, with or without FMA contraction),
// and the device kernels carry the same table (search_kernels.inc: dev_logf).
// TEST INFRASTRUCTURE (oracle).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace oracle {

struct LogfTab { double invc, logc; };
static const LogfTab LOGF_T[16] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2},
};
static const double LOGF_LN2 = 0x1.62e42fefa39efp-1;
static const double LOGF_A[3] = {-0x1.00ea348b88334p-2, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2};

inline float ref_logf(float x) {
    uint32_t ix;
    std::memcpy(&ix, &x, 4);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {  // x < 0x1p-126 or inf or nan
        if (ix * 2 == 0) return -INFINITY;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return NAN;
        const float xs = x * 0x1p23f;  // subnormal: normalise
        std::memcpy(&ix, &xs, 4);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) % 16);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    float zf;
    std::memcpy(&zf, &iz, 4);
    const double z = (double)zf;
    const double r = z * LOGF_T[i].invc - 1;
    const double y0 = LOGF_T[i].logc + (double)k * LOGF_LN2;
    const double r2 = r * r;
    double y = LOGF_A[1] * r + LOGF_A[2];
    y = LOGF_A[0] * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

}  // namespace oracle
