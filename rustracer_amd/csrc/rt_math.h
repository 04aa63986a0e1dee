// rt_math.h — arithmetic shared by the host scene code and the gfx950 kernels.
//
// Everything here is built from single IEEE-754 operations (no contraction: the library is
// compiled with -ffp-contract=off; HIP's f32 divide and sqrt are the correctly rounded forms),
// so a value computed on the host and on the device is the same bit pattern.
//
// Vector operations follow nalgebra 0.23.2 as wrapped by the reference's Point/Vector
// (src/scene/geo.rs:33-235): dot = (x*x' + y*y') + z*z', normalize = v / norm (three divides).
// sin/cos/acos/pow stand in for the libm calls behind f32::sin, f32::cos, f32::acos, f32::powf
// (src/scene/geo.rs:106-143, src/common.rs:57): fixed double-precision evaluations rounded once
// to f32 (correctly rounded on all sampled inputs; tests/test_oracle_math.py).
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIP__)
#define RT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define RT_HD inline
#endif

namespace rt {

struct vec3 {
    float x, y, z;
};
RT_HD vec3 mk(float x, float y, float z) { vec3 v; v.x = x; v.y = y; v.z = z; return v; }
RT_HD vec3 add(vec3 a, vec3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD vec3 sub(vec3 a, vec3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD vec3 scale(vec3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RT_HD float dot3(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD vec3 cross3(vec3 a, vec3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_HD float length3(vec3 a) { return sqrtf(dot3(a, a)); }
RT_HD vec3 unit3(vec3 a) { float n = length3(a); return mk(a.x / n, a.y / n, a.z / n); }

RT_HD float bits_to_float(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
RT_HD uint32_t float_to_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
RT_HD double bits_to_double(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
RT_HD uint64_t double_to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

// ---- sin / cos: Cody-Waite reduction by pi/2 (33-bit head), Taylor kernels on |r| <= pi/4 ----
RT_HD void sincos_f(float xf, float* s_out, float* c_out) {
    const double x = (double)xf;
    const double kd = floor(x * 0.63661977236758134308 + 0.5);
    const int k = (int)kd;
    const double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    const double z = r * r;
    double ps = 1.0 / 6227020800.0;
    ps = -1.0 / 39916800.0 + z * ps;
    ps = 1.0 / 362880.0 + z * ps;
    ps = -1.0 / 5040.0 + z * ps;
    ps = 1.0 / 120.0 + z * ps;
    ps = -1.0 / 6.0 + z * ps;
    const double s = r + r * (z * ps);
    double pc = 1.0 / 87178291200.0;
    pc = -1.0 / 479001600.0 + z * pc;
    pc = 1.0 / 3628800.0 + z * pc;
    pc = -1.0 / 40320.0 + z * pc;
    pc = 1.0 / 720.0 + z * pc;
    pc = -1.0 / 24.0 + z * pc;
    pc = 0.5 + z * pc;
    const double c = 1.0 - z * pc;
    const int q = k & 3;
    const double so = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
    const double co = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
    *s_out = (float)so;
    *c_out = (float)co;
}
RT_HD float sin_f(float x) { float s, c; sincos_f(x, &s, &c); return s; }
RT_HD float cos_f(float x) { float s, c; sincos_f(x, &s, &c); return c; }

// ---- sampler trigonometry: single precision, fused multiply-add, branch-free ----
// random_sphere / random_hemisphere (src/scene/geo.rs:101-145) call f32::sin, f32::cos and f32::acos once per light sample
// and per bounce: the innermost arithmetic of the path tracer.  These stand in for the platform libm there.  Every
// operation is one IEEE single-precision operation (fmaf is the exactly rounded fused multiply-add on host and device),
// so host and device agree bit for bit.  Error against the exact value, measured on EVERY input the samplers can produce
// (theta = 2*pi*xi and u = 1 - 2*xi for all 2^23 values of xi, and phi = acos_s(u)) and on every float of [2^-32, 2*pi] /
// [-1, 1] (tools/mathcheck.cpp): sin <= 0.87 ulp, cos <= 0.81 ulp, acos <= 0.79 ulp.
//
// sincos_s, 0 <= x <= 2*pi (k <= 4): three-term Cody-Waite reduction by pi/2 = H + M + L.  x - k*H is exact (both are
// multiples of 2^-23 below 8 and the difference is below 1); M has 21 significant bits, so k*M is exact and the rounding
// error of r = r1 - k*M is recovered exactly in `lo` (Fast2Sum when |r1| >= |k*M|, and when |r1| < |k*M| the subtraction
// itself is exact); minimax kernels on |r| <= pi/4, sin(r) = r + r^3 P(r^2), cos(r) = 1 - r^2/2 + r^4 Q(r^2) with the
// rounding error of 1 - r^2/2 carried along.
#if defined(__HIPCC__)
// The short exact forms of 1/x, a/b and sqrt(x) for mid-range operands (derivation and tests: rt_kernels.h, "IEEE f32 reciprocal,
// division and square root"); device code only.
__device__ __forceinline__ bool mid_range(float x) { const float a = fabsf(x); return a >= 0x1p-96f && a <= 0x1p+96f; }
__device__ __forceinline__ bool div_range(float x) { const float a = fabsf(x); return a >= 0x1p-60f && a <= 0x1p+60f; }   // quotients and residuals stay normal
__device__ __forceinline__ float rcp_mid(float x) {
    float y = __builtin_amdgcn_rcpf(x);
    float e = fmaf(-x, y, 1.0f);
    y = fmaf(y, e, y);
    e = fmaf(-x, y, 1.0f);
    return fmaf(y, e, y);
}
__device__ __forceinline__ float div_mid(float a, float b, float rcp_b) {
    const float q = a * rcp_b;
    const float r = fmaf(-b, q, a);
    return fmaf(r, rcp_b, q);
}
__device__ __forceinline__ float sqrt_mid(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);
    float out = rm <= 0.0f ? sm : s;
    out = rp > 0.0f ? sp : out;
    return out;
}
#endif

#define RT_PIO2_H 1.57079637050628662109375f        // float(pi/2)
#define RT_PIO2_M (-4.3711395392165286466e-08f)     // pi/2 - H rounded to 21 bits
#define RT_PIO2_L 5.3903029534742383927e-15f        // pi/2 - H - M
RT_HD void sincos_s(float x, float* s_out, float* c_out) {
    const float k = rintf(x * 0.636619746685028076171875f);
    const float r1 = fmaf(k, -RT_PIO2_H, x);
    const float p = k * RT_PIO2_M;
    const float r = r1 - p;
    float lo = (r1 - r) - p;
    lo = fmaf(k, -RT_PIO2_L, lo);
    const float z = r * r;
    float ps = fmaf(-0.00019508754485286772f, z, 0.008332112804055214f);
    ps = fmaf(ps, z, -0.16666653752326965f);
    const float s = r + fmaf(r * z, ps, lo);
    float pc = fmaf(2.4425688025075942e-05f, z, -0.0013887251261621714f);
    pc = fmaf(pc, z, 0.04166664555668831f);
    const float w = fmaf(-0.5f, z, 1.0f);
    const float e = fmaf(-0.5f, z, 1.0f - w);
    float t = fmaf(z * z, pc, e);
    t = fmaf(-r, lo, t);
    const float c = w + t;
    const int q = (int)k;
    const float s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *s_out = (q & 2) ? -s1 : s1;
    *c_out = ((q + 1) & 2) ? -c1 : c1;
}
// acos_s, |u| <= 1: asin(x) = x + x z P(z), z = x^2 <= 1/4 (minimax).  |u| <= 1/2: pi/2 - asin(u).  Above: 2 asin(sqrt((1-|u|)/2)),
// with the rounding error of the square root put back through its exact residual z - s^2 (fmaf) over 2s.
RT_HD float acos_s(float u) {
    const float a = u < 0.0f ? -u : u;
    const bool small = a <= 0.5f;
    const float z = small ? u * u : (1.0f - a) * 0.5f;
    // z is 0 or in [2^-46, 1/4], the divisor 2 sqrt(z) in [2^-22, 1], the residual 0 or in [2^-72, 2^-25]: always inside the range of the
    // short exact forms, so the device takes them without a range check (same bits as sqrtf and `/`; acos_s is compared with the
    // oracle's on every input the samplers can produce, tests/test_gpu_parity.py::test_device_math_bit_exact)
#if defined(__HIP_DEVICE_COMPILE__)
    const float sq = sqrt_mid(z);
#else
    const float sq = sqrtf(z);
#endif
    float p = fmaf(0.04216630756855011f, z, 0.024179520085453987f);
    p = fmaf(p, z, 0.04547037556767464f);
    p = fmaf(p, z, 0.07495297491550446f);
    p = fmaf(p, z, 0.16666752099990845f);
    const float R = z * p;
#if defined(__HIP_DEVICE_COMPILE__)
    const float res = div_mid(fmaf(-sq, sq, z), sq + sq, rcp_mid(sq + sq));
#else
    const float res = fmaf(-sq, sq, z) / (sq + sq);
#endif
    const float c = sq > 0.0f ? res : 0.0f;
    const float w = fmaf(sq, R, c);
    const float r_small = RT_PIO2_H - (u - fmaf(-u, R, RT_PIO2_M));
    const float r_pos = 2.0f * (sq + w);
    const float r_neg = 2.0f * RT_PIO2_H - 2.0f * (sq + (w - RT_PIO2_M));
    return small ? r_small : (u > 0.0f ? r_pos : r_neg);
}

// ---- acos: fdlibm rational asin(x) = x + x*R(x^2) on |x| <= 1/2, half-angle form above ----
RT_HD double asin_ratio(double z) {
    const double p = z * (1.66666666666666657415e-01 +
                          z * (-3.25565818622400915405e-01 +
                               z * (2.01212532134862925881e-01 +
                                    z * (-4.00555345006794114027e-02 + z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
    const double q = 1.0 + z * (-2.40339491173441421878e+00 +
                                z * (2.02094576023350569471e+00 + z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
    return p / q;
}
RT_HD float acos_f(float xf) {
    const double x = (double)xf;
    const double PI_D = 3.14159265358979311600e+00, PIO2_D = 1.57079632679489655800e+00;
    if (x != x) return xf;
    if (x >= 1.0) return x == 1.0 ? 0.0f : bits_to_float(0x7fc00000u);
    if (x <= -1.0) return x == -1.0 ? (float)PI_D : bits_to_float(0x7fc00000u);
    // One evaluation of the rational for both ranges (a wave holds lanes of both): the operations applied to
    // each value are those of the two-branch form, so the result is the same bit pattern.
    const double ax = x < 0.0 ? -x : x;
    const bool small = ax <= 0.5;
    const double z = small ? x * x : (1.0 - ax) * 0.5;
    const double r = asin_ratio(z);
    if (small) return (float)(PIO2_D - (x + x * r));
    const double s = sqrt(z);
    const double w = s + s * r;
    if (x > 0.0) return (float)(2.0 * w);
    return (float)(PI_D - 2.0 * w);
}

// ---- pow(x, y) = exp(y * log(x)) for x >= 0 (tone map base is clamped to [0,1]) ----
RT_HD float pow_f(float xf, float yf) {
    if (xf != xf) return xf;
    if (xf == 0.0f) return 0.0f;
    if (xf == 1.0f) return 1.0f;
    const double x = (double)xf, y = (double)yf;
    uint64_t bits = double_to_bits(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = bits_to_double((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.41421356237309514547) { m *= 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 17.0;
    p = 1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p;
    p = 1.0 / 11.0 + z * p;
    p = 1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p;
    p = 1.0 / 3.0 + z * p;
    const double logm = 2.0 * (s + s * (z * p));
    const double LN2 = 6.93147180559945286227e-01;
    const double t = y * ((double)e * LN2 + logm);
    const double kd = floor(t * 1.44269504088896338700e+00 + 0.5);
    const double r = t - kd * LN2;
    double q = 1.0 / 479001600.0;
    q = 1.0 / 39916800.0 + r * q;
    q = 1.0 / 3628800.0 + r * q;
    q = 1.0 / 362880.0 + r * q;
    q = 1.0 / 40320.0 + r * q;
    q = 1.0 / 5040.0 + r * q;
    q = 1.0 / 720.0 + r * q;
    q = 1.0 / 120.0 + r * q;
    q = 1.0 / 24.0 + r * q;
    q = 1.0 / 6.0 + r * q;
    q = 0.5 + r * q;
    const double er = 1.0 + (r + (r * r) * q);
    const int k = (int)kd;
    if (k < -1000) return 0.0f;
    return (float)(er * bits_to_double((uint64_t)(k + 1023) << 52));
}

// Spectrum::to_color (src/common.rs:46-58): clamp, gamma 1/2.2, *255, `as u8` (truncate, NaN -> 0).
RT_HD uint32_t to_color(float intensity) {
    float c = intensity;
    if (c < 0.0f) c = 0.0f;
    if (c > 1.0f) c = 1.0f;
    const float v = pow_f(c, 1.0f / 2.2f) * 255.0f;
    if (v > 255.0f) return 255u;
    if (!(v == v)) return 0u;
    if (v <= 0.0f) return 0u;
    return (uint32_t)(int)v;
}

// ---- RNG: the fastrand 1.4.0 generator (PCG-XSH-RR 64/32) with one stream per (pixel, sample) ----
RT_HD uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
#define RT_PCG_MUL 6364136223846793005ULL
#define RT_PCG_INC 1442695040888963407ULL
RT_HD uint64_t stream_seed(uint64_t seed_mixed, uint32_t pixel_index, uint32_t sample) {
    return mix64(seed_mixed ^ (((uint64_t)pixel_index << 32) | (uint64_t)sample));   // seed_mixed = mix64(seed)
}
RT_HD float next_f32(uint64_t* state) {
    const uint64_t s = *state;
    *state = s * RT_PCG_MUL + RT_PCG_INC;
    const uint32_t x = (uint32_t)((s ^ (s >> 18)) >> 27);
    const uint32_t rot = (uint32_t)(s >> 59);
    const uint32_t u = (x >> rot) | (x << ((32u - rot) & 31u));
    return bits_to_float(0x3F800000u | (u >> 9)) - 1.0f;
}

}  // namespace rt
