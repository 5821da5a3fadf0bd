// ln_f32.hpp -- f32::ln as the reference computes it, for the host builder and the device kernels alike.
// Rust's f32::ln is the platform libm's logf; glibc's (sysdeps/ieee754/flt-32/e_logf.c) is a 16-entry table of
// {1/c, log c}, a degree-3 polynomial, all in double, one final rounding -- restated here.  No FMA contraction is needed
// to match it: the restatement equals glibc 2.35's logf on every positive finite float with and without contraction
// (checked exhaustively on the oracle's twin of this function, tests/test_oracle.py).  DistJeffreys and
// DistJensenShannon are the only users.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define HNSW_HD __host__ __device__
#else
#define HNSW_HD
#endif

namespace hnswgpu {

HNSW_HD inline float ln_f32(float x) {
    // {invc, logc} for the 16 subintervals of [0x1.66p-1, 0x1.66p0)
    const double T[32] = {
        0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,
        0x1.49539f0f010bp+0,  -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,
        0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0,  -0x1.1aa2bc79c81p-3,
        0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,
        0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1p+0,               0x0p+0,
        0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5,  0x1.ca4b31f026aap-1,  0x1.c5e53aa362eb4p-4,
        0x1.b2036576afce6p-1, 0x1.526e57720db08p-3,  0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3,
        0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,  0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2,
    };
    const double LN2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix;
    memcpy(&ix, &x, 4);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {  // x < 0x1p-126, or inf, or nan
        if (ix * 2u == 0u) return -__builtin_inff();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");
        const float xs = x * 0x1p23f;  // subnormal: normalise
        memcpy(&ix, &xs, 4);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    float zf;
    memcpy(&zf, &iz, 4);
    const double z = (double)zf;
    const double r = z * T[2 * i] - 1.0;
    const double y0 = T[2 * i + 1] + (double)k * LN2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

}  // namespace hnswgpu
