// libm_f32.h -- atan2f as the platform's libm evaluates it, restated so that device code can evaluate it too.
//
// The reference's angle_degree (src/math_util.rs:31-33) is f32::atan2, i.e. the C library's atan2f.  On glibc up to 2.40
// (this image: 2.35) that is the FreeBSD / fdlibm single-precision routine (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c):
// argument reduction into five intervals, an odd polynomial of degree 23 split in two halves, every operation a binary32
// operation -- so the same operations in the same order give the same bits on any IEEE-754 machine, the GPU included
// (this file is only ever compiled with -ffp-contract=off; division is correctly rounded on both sides).  The published
// algorithm is restated below from its description (constants are the published ones).
//
// Whether THIS process's libm is that routine is checked, not assumed: agx::libm_atan2f_matches() compares the two on a
// fixed set of inputs (tests/test_abi_cpu.py: tens of millions); the device tail is refused where they differ.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)  // (the header is shared by the HIP kernels and the plain C++ host tail)
#define AGX_HD __host__ __device__ inline
#else
#define AGX_HD inline
#endif

namespace agx {

AGX_HD uint32_t f32_bits(float v)
{
    uint32_t u;
    memcpy(&u, &v, 4);
    return u;
}
AGX_HD float f32_from_bits(uint32_t u)
{
    float v;
    memcpy(&v, &u, 4);
    return v;
}

AGX_HD float fdlibm_atanf(float x)
{
    const float atanhi0 = 4.6364760399e-01f, atanhi1 = 7.8539812565e-01f, atanhi2 = 9.8279368877e-01f, atanhi3 = 1.5707962513e+00f;
    const float atanlo0 = 5.0121582440e-09f, atanlo1 = 3.7748947079e-08f, atanlo2 = 3.4473217170e-08f, atanlo3 = 7.5497894159e-08f;
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const uint32_t hx = f32_bits(x), ix = hx & 0x7fffffffu;
    const bool neg = (hx >> 31) != 0;
    if (ix >= 0x4c000000u) {  // |x| >= 2^25
        if (ix > 0x7f800000u) return x + x;  // NaN
        return neg ? -atanhi3 - atanlo3 : atanhi3 + atanlo3;
    }
    int id;
    float hi = 0.0f, lo = 0.0f;
    if (ix < 0x3ee00000u) {  // |x| < 0.4375
        if (ix < 0x31000000u) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = f32_from_bits(ix);  // fabsf
        if (ix < 0x3f980000u) {  // |x| < 1.1875
            if (ix < 0x3f300000u) {  // 7/16 <= |x| < 11/16
                id = 0; hi = atanhi0; lo = atanlo0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {  // 11/16 <= |x| < 19/16
                id = 1; hi = atanhi1; lo = atanlo1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000u) {  // |x| < 2.4375
                id = 2; hi = atanhi2; lo = atanlo2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else {  // 2.4375 <= |x| < 2^25
                id = 3; hi = atanhi3; lo = atanlo3;
                x = -1.0f / x;
            }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return neg ? -r : r;
}

AGX_HD float fdlibm_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    const uint32_t hx = f32_bits(x), hy = f32_bits(y), ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;  // NaN
    if (hx == 0x3f800000u) return fdlibm_atanf(y);           // x = 1.0
    const int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);  // 2 * sign(x) + sign(y)
    if (iy == 0) {  // y = 0
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;  // x = 0
    if (ix == 0x7f800000u) {  // x = INF
        if (iy == 0x7f800000u) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000u) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;  // y = INF
    const int k = ((int)iy - (int)ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;         // |y / x| > 2^60
    else if ((hx >> 31) && k < -60) z = 0.0f;      // |y| / x < -2^60
    else {
        const float q = y / x;
        z = fdlibm_atanf(f32_from_bits(f32_bits(q) & 0x7fffffffu));
    }
    switch (m) {
    case 0: return z;
    case 1: return f32_from_bits(f32_bits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

}  // namespace agx
