// oracle_rt.cpp — CPU ORACLE (TEST INFRASTRUCTURE ONLY).
//
// A C++ restatement of stephenjayakar/rustracer's per-pixel trace/shade path, following the
// reference source function by function (citations are path:line under /root/reference).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (rustracer_amd/, include/) never includes, links or calls anything here.
//
// PARITY STATUS: "parity unpinned".  The reference ships no golden vectors, no assertions
// (src/scene/objects.rs:323-343 checks nothing) and cannot be compiled here (no cargo/rustc).
// Three pieces of arithmetic live in crates whose source is not on disk; they are restated from
// their published algorithms and are the build's *chosen definition*:
//   * bvh 0.3.2      (Cargo.lock:108-111)  SAH-bucket binary BVH, Williams slab test
//   * nalgebra 0.23.2 (Cargo.lock:481-484) dot = (x*x' + y*y') + z*z', normalize = v / norm
//   * fastrand 1.4.0 (Cargo.lock:296-299)  PCG-XSH-RR 64/32, f32 = bits(0x3F800000|u>>9) - 1
// Deliberate deviations from the reference, each forced by reproducibility:
//   * RNG: the reference generator is thread-local and time-seeded (common.rs:137,
//     geo.rs:102-103,136-137).  Here every (pixel, sample) owns a stream seeded from
//     hash(seed, pixel_index, sample); the generator and float conversion are unchanged.
//   * sin/cos/acos/powf: the reference calls the platform libm through f32::sin etc.  Here they
//     are fixed sequences of IEEE operations, so that a CPU and a GPU evaluating the same formula
//     agree bit for bit: the samplers (geo.rs:101-145) use single-precision fused-multiply-add
//     kernels (orc_sincos_s, orc_acos_s: < 0.9 ulp from the exact value on every float of their
//     domains, tools/mathcheck.cpp), ScreenParams and the tone map use double-precision polynomial
//     evaluations rounded once to f32 (orc_sinf, orc_powf: <= 0.5001 ulp).  tests/ bound their
//     distance to libm in ulps.
//
// Build: oracle/Makefile  (g++ -O2 -ffp-contract=off, no fast-math).

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------
// common.rs:6
const float EPS = 0.0000001f;
const float PI_F = 3.14159265358979323846f;   // std::f32::consts::PI
const float RUSSIAN_ROULETTE_PROBABILITY = 0.7f;  // raytracer.rs:11

// ------------------------------------------------------------------------------------------
// Deterministic elementary functions (see header).  All arithmetic is IEEE double with no
// contraction; the result is rounded to f32 once.
const double PIO2_HI = 1.57079632673412561417e+00;  // first 33 bits of pi/2
const double PIO2_LO = 6.07710050650619224932e-11;  // pi/2 - PIO2_HI

inline void sincos_reduced(double r, double* s, double* c) {
    // |r| <= pi/4 (+ a hair).  Taylor series; truncation < 3e-14 relative, far below f32 ulp.
    double z = r * r;
    double ps = 1.0 / 6227020800.0;                // 1/13!
    ps = -1.0 / 39916800.0 + z * ps;                // -1/11!
    ps = 1.0 / 362880.0 + z * ps;                   // 1/9!
    ps = -1.0 / 5040.0 + z * ps;                    // -1/7!
    ps = 1.0 / 120.0 + z * ps;                      // 1/5!
    ps = -1.0 / 6.0 + z * ps;                       // -1/3!
    *s = r + r * (z * ps);
    double pc = 1.0 / 87178291200.0;                // 1/14!
    pc = -1.0 / 479001600.0 + z * pc;               // -1/12!
    pc = 1.0 / 3628800.0 + z * pc;                  // 1/10!
    pc = -1.0 / 40320.0 + z * pc;                   // -1/8!
    pc = 1.0 / 720.0 + z * pc;                      // 1/6!
    pc = -1.0 / 24.0 + z * pc;                      // -1/4!
    pc = 0.5 + z * pc;
    *c = 1.0 - z * pc;
}

inline void orc_sincos(float xf, float* sf, float* cf) {
    double x = (double)xf;
    // k = nearest integer to x * 2/pi; arguments here are in [-pi/4, 2*pi], |k| <= 4.
    double kd = std::floor(x * 0.63661977236758134308 + 0.5);
    int k = (int)kd;
    double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    double s, c;
    sincos_reduced(r, &s, &c);
    double so, co;
    switch (k & 3) {
        case 0: so = s; co = c; break;
        case 1: so = c; co = -s; break;
        case 2: so = -s; co = -c; break;
        default: so = -c; co = s; break;
    }
    *sf = (float)so;
    *cf = (float)co;
}
inline float orc_sinf(float x) { float s, c; orc_sincos(x, &s, &c); return s; }
inline float orc_cosf(float x) { float s, c; orc_sincos(x, &s, &c); return c; }

// asin on |x| <= 0.5 by the fdlibm rational form  asin(x) = x + x*R(x^2).
inline double asin_R(double z) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    return p / q;
}
inline float orc_acosf(float xf) {
    double x = (double)xf;
    const double PI_D = 3.14159265358979311600e+00, PIO2_D = 1.57079632679489655800e+00;
    if (x != x) return xf;
    if (x >= 1.0) return x == 1.0 ? 0.0f : std::numeric_limits<float>::quiet_NaN();
    if (x <= -1.0) return x == -1.0 ? (float)PI_D : std::numeric_limits<float>::quiet_NaN();
    double ax = x < 0.0 ? -x : x;
    if (ax <= 0.5) {
        return (float)(PIO2_D - (x + x * asin_R(x * x)));
    }
    double z = (1.0 - ax) * 0.5;
    double s = std::sqrt(z);
    double w = s + s * asin_R(z);  // asin(sqrt(z))
    if (x > 0.0) return (float)(2.0 * w);
    return (float)(PI_D - 2.0 * w);
}

// f32::sin / f32::cos / f32::acos as the samplers call them (geo.rs:106-107, 138-143): the build's single-precision
// definition (see header; rustracer_amd/csrc/rt_math.h holds the device's statement of the same operations).
// pi/2 = H + M + L with M cut to 21 bits, so that k*M is exact for the k <= 4 that occur (0 <= x <= 2 pi).
const float S_PIO2_H = 1.57079637050628662109375f, S_PIO2_M = -4.3711395392165286466e-08f, S_PIO2_L = 5.3903029534742383927e-15f;
inline void orc_sincos_s(float x, float* sf, float* cf) {
    float k = std::rint(x * 0.636619746685028076171875f);   // round to nearest, ties to even
    float r1 = std::fma(k, -S_PIO2_H, x);                    // exact
    float p = k * S_PIO2_M;                                  // exact
    float r = r1 - p;
    float lo = (r1 - r) - p;                                 // the rounding error of r, exactly
    lo = std::fma(k, -S_PIO2_L, lo);
    float z = r * r;
    float ps = std::fma(-0.00019508754485286772f, z, 0.008332112804055214f);
    ps = std::fma(ps, z, -0.16666653752326965f);
    float s = r + std::fma(r * z, ps, lo);                   // sin(r + lo)
    float pc = std::fma(2.4425688025075942e-05f, z, -0.0013887251261621714f);
    pc = std::fma(pc, z, 0.04166664555668831f);
    float w = std::fma(-0.5f, z, 1.0f);
    float e = std::fma(-0.5f, z, 1.0f - w);                  // what rounding w lost
    float t = std::fma(z * z, pc, e);
    t = std::fma(-r, lo, t);
    float c = w + t;                                         // cos(r + lo)
    switch ((int)k & 3) {
        case 0: *sf = s; *cf = c; break;
        case 1: *sf = c; *cf = -s; break;
        case 2: *sf = -s; *cf = -c; break;
        default: *sf = -c; *cf = s; break;
    }
}
inline float orc_acos_s(float u) {
    float a = u < 0.0f ? -u : u;
    bool small = a <= 0.5f;
    float z = small ? u * u : (1.0f - a) * 0.5f;
    float p = std::fma(0.04216630756855011f, z, 0.024179520085453987f);
    p = std::fma(p, z, 0.04547037556767464f);
    p = std::fma(p, z, 0.07495297491550446f);
    p = std::fma(p, z, 0.16666752099990845f);
    float R = z * p;                                         // asin(x) = x + x R(x^2)
    if (small) return S_PIO2_H - (u - std::fma(-u, R, S_PIO2_M));
    float sq = std::sqrt(z);
    float c = sq > 0.0f ? std::fma(-sq, sq, z) / (sq + sq) : 0.0f;   // sqrt(z) = sq + c
    float w = std::fma(sq, R, c);
    if (u > 0.0f) return 2.0f * (sq + w);
    return 2.0f * S_PIO2_H - 2.0f * (sq + (w - S_PIO2_M));
}

// powf for the tone map (common.rs:57): base in [0,1], exponent 1/2.2.  exp(y*log(x)) in double.
inline float orc_powf(float xf, float yf) {
    if (xf != xf) return xf;
    if (xf == 0.0f) return 0.0f;
    if (xf == 1.0f) return 1.0f;
    double x = (double)xf, y = (double)yf;
    // log(x): x = m * 2^e with m in [sqrt(1/2), sqrt(2))
    uint64_t bits;
    std::memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;       // xf is a normal or subnormal f32 => normal double
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    std::memcpy(&m, &bits, 8);
    if (m > 1.41421356237309514547) { m *= 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 17.0;
    p = 1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p;
    p = 1.0 / 11.0 + z * p;
    p = 1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p;
    p = 1.0 / 3.0 + z * p;
    double logm = 2.0 * (s + s * (z * p));
    const double LN2 = 6.93147180559945286227e-01;
    double t = y * ((double)e * LN2 + logm);
    // exp(t): t <= 0 here for x in (0,1); general enough for |t| < 700.
    double kd = std::floor(t * 1.44269504088896338700e+00 + 0.5);
    double r = t - kd * LN2;
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
    double er = 1.0 + (r + (r * r) * q);
    int k = (int)kd;
    if (k < -1000) return 0.0f;
    uint64_t sb = (uint64_t)(k + 1023) << 52;
    double sc;
    std::memcpy(&sc, &sb, 8);
    return (float)(er * sc);
}

// ------------------------------------------------------------------------------------------
// nalgebra 0.23.2 Vector3<f32>/Point3<f32> as wrapped by geo.rs:33-235.
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { V3 v = {x, y, z}; return v; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   // geo.rs:197-219
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   // geo.rs:189-227
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }       // geo.rs:229-235
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }       // geo.rs:158
inline V3 cross(V3 a, V3 b) {                                                      // geo.rs:163
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float norm(V3 a) { return std::sqrt(dot(a, a)); }                           // geo.rs:148
inline V3 normalized(V3 a) { float n = norm(a); return v3(a.x / n, a.y / n, a.z / n); }  // geo.rs:153

// common.rs:10-132
struct Spectrum {
    float r, g, b;
};
inline Spectrum sp(float r, float g, float b) { Spectrum s = {r, g, b}; return s; }
inline Spectrum black() { return sp(0.f, 0.f, 0.f); }
inline Spectrum operator+(Spectrum a, Spectrum b) { return sp(a.r + b.r, a.g + b.g, a.b + b.b); }
inline Spectrum operator*(Spectrum a, Spectrum b) { return sp(a.r * b.r, a.g * b.g, a.b * b.b); }
inline Spectrum operator*(Spectrum a, float s) { return sp(a.r * s, a.g * s, a.b * s); }
inline bool is_black(Spectrum s) { float c = s.r + s.g + s.b; return c <= EPS; }   // common.rs:40-43

// common.rs:46-58.  clamp keeps NaN; `val as u8` saturates and maps NaN to 0.
inline uint8_t to_u8(float v) {
    if (v > 255.0f) return 255;
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    return (uint8_t)(int)v;   // truncation
}
inline uint8_t to_color(float intensity) {
    float c = intensity;
    if (c < 0.0f) c = 0.0f;
    if (c > 1.0f) c = 1.0f;
    return to_u8(orc_powf(c, 1.0f / 2.2f) * 255.0f);
}

// ------------------------------------------------------------------------------------------
// RNG: fastrand 1.4.0 generator, per-(pixel,sample) stream (see header).
struct Rng {
    uint64_t s;
};
inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline Rng rng_for(uint64_t seed, uint32_t pixel_index, uint32_t sample) {
    Rng r;
    r.s = splitmix64(splitmix64(seed) ^ (((uint64_t)pixel_index << 32) | (uint64_t)sample));
    return r;
}
inline uint32_t rng_u32(Rng* r) {
    uint64_t s = r->s;
    r->s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    uint32_t x = (uint32_t)(((s ^ (s >> 18)) >> 27));
    uint32_t rot = (uint32_t)(s >> 59);
    return (x >> rot) | (x << ((32 - rot) & 31));
}
inline float rng_f32(Rng* r) {
    uint32_t bits = 0x3F800000u | (rng_u32(r) >> 9);
    float f;
    std::memcpy(&f, &bits, 4);
    return f - 1.0f;
}

// ------------------------------------------------------------------------------------------
// geo.rs:10-31
struct Ray {
    V3 origin, direction;
};
inline Ray ray_new(V3 o, V3 d) { Ray r = {o, normalized(d)}; return r; }          // geo.rs:19-24
inline Ray ray_prenorm(V3 o, V3 d) { Ray r = {o, d}; return r; }                   // geo.rs:28-30

// Test switch (orc_set_double_trig): the samplers evaluate sin / cos / acos with the double-precision definitions (orc_sincos, orc_acosf:
// <= 0.5001 ulp) instead of the single-precision forms the kernels share.  The two must give statistically the same image
// (tests/test_oracle_math.py), so that a defect of the single-precision forms cannot hide behind goldens regenerated with them.
static bool g_double_trig = false;
inline void sampler_sincos(float x, float* s, float* c) { if (g_double_trig) orc_sincos(x, s, c); else orc_sincos_s(x, s, c); }
inline float sampler_acos(float u) { return g_double_trig ? orc_acosf(u) : orc_acos_s(u); }

// geo.rs:101-114
inline V3 random_hemisphere(Rng* rng) {
    float xi1 = rng_f32(rng);
    float xi2 = rng_f32(rng);
    float cos_theta = xi1;
    float sin_theta = std::sqrt(1.0f - xi1 * xi1);
    float phi = 2.0f * PI_F * xi2;
    float sin_phi, cos_phi;
    sampler_sincos(phi, &sin_phi, &cos_phi);
    return v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
}
// geo.rs:118-132
inline V3 to_coord_space(V3 v, V3 n) {
    V3 t, b;
    if (n.z < -0.9999999f) {
        t = v3(0.0f, -1.0f, 0.0f);
        b = v3(-1.0f, 0.0f, 0.0f);
    } else {
        float a = 1.0f / (1.0f + n.z);
        float b_val = -n.x * n.y * a;
        t = v3(1.0f - n.x * n.x * a, b_val, -n.x);
        b = v3(b_val, 1.0f - n.y * n.y * a, -n.y);
    }
    return (t * v.x + b * v.y) + n * v.z;
}
// geo.rs:135-145
inline V3 random_sphere(Rng* rng) {
    float xi1 = rng_f32(rng);
    float xi2 = rng_f32(rng);
    float theta = 2.0f * PI_F * xi1;
    float phi = sampler_acos(1.0f - 2.0f * xi2);
    float sp_, cp_, st_, ct_;
    sampler_sincos(phi, &sp_, &cp_);
    sampler_sincos(theta, &st_, &ct_);
    return v3(sp_ * ct_, sp_ * st_, cp_);
}

// ------------------------------------------------------------------------------------------
// objects.rs:9-31, 178-211
enum { BSDF_DIFFUSE = 0, BSDF_SPECULAR = 1 };
struct Material {
    uint32_t bsdf;
    Spectrum reflectance, emittance;
};
inline Material material(uint32_t bsdf, Spectrum refl, Spectrum emit) { Material m = {bsdf, refl, emit}; return m; }
inline bool same_material(const Material& a, const Material& b) { return std::memcmp(&a, &b, sizeof(Material)) == 0; }

struct Object {
    int is_sphere;
    V3 p1, p2, p3, vn1, vn2, vn3;  // triangle (objects.rs:185-195)
    V3 center;
    float radius;                  // sphere   (objects.rs:178-183)
    Material mat;
    size_t node_index;
};

// objects.rs:240-267
Object triangle_new(V3 p1, V3 p2, V3 p3, V3 vn1, V3 vn2, V3 vn3, Material m) {
    Object o;
    std::memset(&o, 0, sizeof(o));
    o.is_sphere = 0;
    o.p1 = p1; o.p2 = p2; o.p3 = p3;
    o.vn1 = normalized(vn1); o.vn2 = normalized(vn2); o.vn3 = normalized(vn3);
    o.mat = m;
    return o;
}
Object triangle_new_without_vn(V3 p1, V3 p2, V3 p3, Material m) {
    V3 n = cross(p2 - p1, p3 - p1);
    return triangle_new(p1, p2, p3, n, n, n, m);
}
Object sphere_new(V3 c, float r, Material m) {                                      // objects.rs:214-221
    Object o;
    std::memset(&o, 0, sizeof(o));
    o.is_sphere = 1;
    o.center = c; o.radius = r; o.mat = m;
    return o;
}

// objects.rs:36-78.  Returns false for None.
inline bool object_intersect(const Object& obj, const Ray& ray, float* out) {
    if (!obj.is_sphere) {
        V3 p1 = obj.p1, p2 = obj.p2, p3 = obj.p3;
        V3 direction = ray.direction;
        V3 e1 = p2 - p1;
        V3 e2 = p3 - p1;
        V3 s = ray.origin - p1;
        V3 s1 = cross(direction, e2);
        V3 s2 = cross(s, e1);
        float scalar = 1.0f / dot(s1, e1);
        float t = dot(s2, e2) * scalar;
        float b1 = dot(s1, s) * scalar;
        float b2 = dot(s2, direction) * scalar;
        if (b1 < 0.0f || b2 < 0.0f || b1 > 1.0f || b2 > 1.0f || b1 + b2 > 1.0f + EPS || t < EPS) return false;
        *out = t;
        return true;
    }
    V3 l = obj.center - ray.origin;
    float adj = dot(l, ray.direction);
    float d2 = dot(l, l) - (adj * adj);
    float radius2 = obj.radius * obj.radius;
    if (d2 > radius2) return false;
    float thc = std::sqrt(radius2 - d2);
    float t0 = adj - thc;
    float t1 = adj + thc;
    if (t0 < 0.0f && t1 < 0.0f) return false;
    *out = t0 < t1 ? t0 : t1;
    return true;
}

// objects.rs:228-230, 270-290
inline V3 surface_normal(const Object& obj, V3 point) {
    if (obj.is_sphere) return normalized(point - obj.center);
    V3 v0 = obj.p2 - obj.p1;
    V3 v1 = obj.p3 - obj.p1;
    V3 v2 = point - obj.p1;
    float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    float u = 1.0f - v - w;
    return (obj.vn1 * u + obj.vn2 * v) + obj.vn3 * w;
}

// objects.rs:96-118 (sphere lights only; triangle lights are unimplemented!() in the reference)
struct LightSample {
    float pdf;
    V3 wi;
    float distance;
};
inline LightSample sample_l(const Object& light, V3 p, Rng* rng) {
    V3 s = light.center + random_sphere(rng) * light.radius;     // objects.rs:223-226
    V3 ps = s - p;
    float d_s = norm(ps);
    V3 wi = ps * (1.0f / d_s);
    float d_c = norm(light.center - p);
    float cos_a = (d_c * d_c + light.radius * light.radius - d_s * d_s) / (2.0f * d_c * light.radius);
    LightSample ls;
    ls.pdf = 2.0f * PI_F * (1.0f - cos_a);
    ls.wi = wi;
    ls.distance = d_s;
    return ls;
}

// objects.rs:121-127
inline Spectrum object_bsdf(const Object& obj) {
    if (obj.mat.bsdf == BSDF_DIFFUSE) return obj.mat.reflectance * (1.0f / PI_F);
    return black();
}
// objects.rs:130-148
struct BSDFSample {
    V3 wi;
    float pdf;
    Spectrum reflected;
};
inline BSDFSample sample_bsdf(const Object& obj, V3 wo, V3 normal, Rng* rng) {
    BSDFSample s;
    if (obj.mat.bsdf == BSDF_DIFFUSE) {
        s.wi = to_coord_space(random_hemisphere(rng), normal);
        s.pdf = 2.0f * PI_F;
        s.reflected = object_bsdf(obj);
    } else {
        s.wi = wo - (normal * 2.0f) * dot(wo, normal);
        s.pdf = 1.0f;
        float cos_theta = std::fabs(dot(s.wi, normal));
        s.reflected = obj.mat.reflectance * (1.0f / cos_theta);
    }
    return s;
}

// ------------------------------------------------------------------------------------------
// bvh 0.3.2 (restated; see header).  aabb.rs / bvh/bvh_impl.rs / ray.rs of that crate.
struct AABB {
    V3 mn, mx;
};
inline AABB aabb_empty() {
    float inf = std::numeric_limits<float>::infinity();
    AABB a = {v3(inf, inf, inf), v3(-inf, -inf, -inf)};
    return a;
}
inline float fmin_(float a, float b) { return a < b ? a : b; }   // f32::min for non-NaN inputs
inline float fmax_(float a, float b) { return a > b ? a : b; }
inline AABB aabb_join(AABB a, AABB b) {
    AABB r = {v3(fmin_(a.mn.x, b.mn.x), fmin_(a.mn.y, b.mn.y), fmin_(a.mn.z, b.mn.z)),
              v3(fmax_(a.mx.x, b.mx.x), fmax_(a.mx.y, b.mx.y), fmax_(a.mx.z, b.mx.z))};
    return r;
}
inline AABB aabb_grow(AABB a, V3 p) {
    AABB r = {v3(fmin_(a.mn.x, p.x), fmin_(a.mn.y, p.y), fmin_(a.mn.z, p.z)),
              v3(fmax_(a.mx.x, p.x), fmax_(a.mx.y, p.y), fmax_(a.mx.z, p.z))};
    return r;
}
inline V3 aabb_size(AABB a) { return a.mx - a.mn; }
inline V3 aabb_center(AABB a) { V3 s = aabb_size(a); return a.mn + v3(s.x / 2.0f, s.y / 2.0f, s.z / 2.0f); }
inline float aabb_surface_area(AABB a) {
    V3 s = aabb_size(a);
    return 2.0f * (s.x * s.y + s.x * s.z + s.y * s.z);
}
inline int aabb_largest_axis(AABB a) {
    V3 s = aabb_size(a);
    if (s.x > s.y && s.x > s.z) return 0;
    if (s.y > s.z) return 1;
    return 2;
}
inline float axis_of(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// objects.rs:293-316
inline AABB object_aabb(const Object& o) {
    if (o.is_sphere) {
        V3 half = v3(o.radius, o.radius, o.radius);
        AABB a = {o.center - half, o.center + half};
        return a;
    }
    AABB a = {v3(fmin_(fmin_(o.p1.x, o.p2.x), o.p3.x), fmin_(fmin_(o.p1.y, o.p2.y), o.p3.y), fmin_(fmin_(o.p1.z, o.p2.z), o.p3.z)),
              v3(fmax_(fmax_(o.p1.x, o.p2.x), o.p3.x), fmax_(fmax_(o.p1.y, o.p2.y), o.p3.y), fmax_(fmax_(o.p1.z, o.p2.z), o.p3.z))};
    return a;
}

struct BVHNode {
    int is_leaf;
    AABB child_l_aabb, child_r_aabb;
    size_t child_l_index, child_r_index;
    size_t shape_index;
    uint32_t depth;
};

struct BVH {
    std::vector<BVHNode> nodes;
};

const float BVH_EPSILON = 0.00001f;
const int NUM_BUCKETS = 6;

size_t bvh_build_rec(std::vector<Object>& shapes, const std::vector<size_t>& indices, std::vector<BVHNode>& nodes,
                     uint32_t depth) {
    AABB aabb_bounds = aabb_empty(), centroid_bounds = aabb_empty();
    for (size_t k = 0; k < indices.size(); ++k) {
        AABB sa = object_aabb(shapes[indices[k]]);
        aabb_bounds = aabb_join(aabb_bounds, sa);
        centroid_bounds = aabb_grow(centroid_bounds, aabb_center(sa));
    }
    if (indices.size() == 1) {
        size_t node_index = nodes.size();
        BVHNode leaf;
        std::memset(&leaf, 0, sizeof(leaf));
        leaf.is_leaf = 1;
        leaf.shape_index = indices[0];
        leaf.depth = depth;
        nodes.push_back(leaf);
        shapes[indices[0]].node_index = node_index;
        return node_index;
    }
    size_t node_index = nodes.size();
    {
        BVHNode dummy;
        std::memset(&dummy, 0, sizeof(dummy));
        nodes.push_back(dummy);
    }
    int split_axis = aabb_largest_axis(centroid_bounds);
    float split_axis_size = axis_of(centroid_bounds.mx, split_axis) - axis_of(centroid_bounds.mn, split_axis);

    AABB child_l_aabb = aabb_empty(), child_r_aabb = aabb_empty();
    size_t child_l_index, child_r_index;
    if (split_axis_size < BVH_EPSILON) {
        size_t half = indices.size() / 2;
        std::vector<size_t> li(indices.begin(), indices.begin() + half), ri(indices.begin() + half, indices.end());
        for (size_t k = 0; k < li.size(); ++k) child_l_aabb = aabb_join(child_l_aabb, object_aabb(shapes[li[k]]));
        for (size_t k = 0; k < ri.size(); ++k) child_r_aabb = aabb_join(child_r_aabb, object_aabb(shapes[ri[k]]));
        child_l_index = bvh_build_rec(shapes, li, nodes, depth + 1);
        child_r_index = bvh_build_rec(shapes, ri, nodes, depth + 1);
    } else {
        AABB bucket_aabb[NUM_BUCKETS];
        size_t bucket_size[NUM_BUCKETS];
        std::vector<size_t> assign[NUM_BUCKETS];
        for (int b = 0; b < NUM_BUCKETS; ++b) { bucket_aabb[b] = aabb_empty(); bucket_size[b] = 0; }
        for (size_t k = 0; k < indices.size(); ++k) {
            AABB sa = object_aabb(shapes[indices[k]]);
            V3 c = aabb_center(sa);
            float rel = (axis_of(c, split_axis) - axis_of(centroid_bounds.mn, split_axis)) / split_axis_size;
            int bn = (int)(rel * ((float)NUM_BUCKETS - 0.01f));
            if (bn < 0) bn = 0;                    // `as usize` saturates
            if (bn > NUM_BUCKETS - 1) bn = NUM_BUCKETS - 1;  // unreachable for rel <= 1; guards the index
            bucket_aabb[bn] = aabb_join(bucket_aabb[bn], sa);
            bucket_size[bn] += 1;
            assign[bn].push_back(indices[k]);
        }
        int min_bucket = 0;
        float min_cost = std::numeric_limits<float>::infinity();
        for (int i = 0; i < NUM_BUCKETS - 1; ++i) {
            AABB la = aabb_empty(), ra = aabb_empty();
            size_t ls = 0, rs = 0;
            for (int b = 0; b <= i; ++b) { la = aabb_join(la, bucket_aabb[b]); ls += bucket_size[b]; }
            for (int b = i + 1; b < NUM_BUCKETS; ++b) { ra = aabb_join(ra, bucket_aabb[b]); rs += bucket_size[b]; }
            float cost = ((float)ls * aabb_surface_area(la) + (float)rs * aabb_surface_area(ra)) / aabb_surface_area(aabb_bounds);
            if (cost < min_cost) {
                min_bucket = i;
                min_cost = cost;
                child_l_aabb = la;
                child_r_aabb = ra;
            }
        }
        std::vector<size_t> li, ri;
        for (int b = 0; b <= min_bucket; ++b) li.insert(li.end(), assign[b].begin(), assign[b].end());
        for (int b = min_bucket + 1; b < NUM_BUCKETS; ++b) ri.insert(ri.end(), assign[b].begin(), assign[b].end());
        for (int b = 0; b < NUM_BUCKETS; ++b) std::vector<size_t>().swap(assign[b]);
        if (li.empty() || ri.empty()) {
            // cost was NaN for every split (degenerate bounds): the crate would assert; split in half.
            std::vector<size_t> all(li);
            all.insert(all.end(), ri.begin(), ri.end());
            size_t half = all.size() / 2;
            li.assign(all.begin(), all.begin() + half);
            ri.assign(all.begin() + half, all.end());
            child_l_aabb = aabb_empty(); child_r_aabb = aabb_empty();
            for (size_t k = 0; k < li.size(); ++k) child_l_aabb = aabb_join(child_l_aabb, object_aabb(shapes[li[k]]));
            for (size_t k = 0; k < ri.size(); ++k) child_r_aabb = aabb_join(child_r_aabb, object_aabb(shapes[ri[k]]));
        }
        child_l_index = bvh_build_rec(shapes, li, nodes, depth + 1);
        child_r_index = bvh_build_rec(shapes, ri, nodes, depth + 1);
    }
    BVHNode n;
    std::memset(&n, 0, sizeof(n));
    n.is_leaf = 0;
    n.depth = depth;
    n.child_l_aabb = child_l_aabb; n.child_l_index = child_l_index;
    n.child_r_aabb = child_r_aabb; n.child_r_index = child_r_index;
    nodes[node_index] = n;
    return node_index;
}

// bvh::ray::Ray (mod.rs:506-512 builds one per scene ray): direction re-normalised.
struct BvhRay {
    V3 origin, inv_direction;
    int sign_x, sign_y, sign_z;
};
inline BvhRay ray_to_bvh_ray(const Ray& r) {
    V3 d = normalized(r.direction);
    BvhRay b;
    b.origin = r.origin;
    b.inv_direction = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    b.sign_x = d.x < 0.0f; b.sign_y = d.y < 0.0f; b.sign_z = d.z < 0.0f;
    return b;
}
inline bool intersects_aabb(const BvhRay& r, const AABB& a) {
    const V3* bnd[2] = {&a.mn, &a.mx};
    float ray_min = (bnd[r.sign_x]->x - r.origin.x) * r.inv_direction.x;
    float ray_max = (bnd[1 - r.sign_x]->x - r.origin.x) * r.inv_direction.x;
    float y_min = (bnd[r.sign_y]->y - r.origin.y) * r.inv_direction.y;
    float y_max = (bnd[1 - r.sign_y]->y - r.origin.y) * r.inv_direction.y;
    if ((ray_min > y_max) || (y_min > ray_max)) return false;
    if (y_min > ray_min) ray_min = y_min;
    if (y_max < ray_max) ray_max = y_max;
    float z_min = (bnd[r.sign_z]->z - r.origin.z) * r.inv_direction.z;
    float z_max = (bnd[1 - r.sign_z]->z - r.origin.z) * r.inv_direction.z;
    if ((ray_min > z_max) || (z_min > ray_max)) return false;
    if (z_max < ray_max) ray_max = z_max;
    return ray_max > 0.0f;
}

// ------------------------------------------------------------------------------------------
// scene/mod.rs:20-24, 82-106
struct Scene {
    std::vector<Object> objects;
    BVH bvh;
    std::vector<size_t> light_indexes;
};

struct Counters {
    uint64_t intersect_calls, occluded_calls, nodes_visited, prims_tested;
};

Scene* scene_new(const std::vector<Object>& triangles, const std::vector<Object>& spheres) {
    Scene* s = new Scene();
    for (size_t i = 0; i < triangles.size(); ++i) s->objects.push_back(triangles[i]);
    for (size_t i = 0; i < spheres.size(); ++i) s->objects.push_back(spheres[i]);
    for (size_t i = 0; i < s->objects.size(); ++i)
        if (!is_black(s->objects[i].mat.emittance)) s->light_indexes.push_back(i);
    if (!s->objects.empty()) {
        std::vector<size_t> idx(s->objects.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
        bvh_build_rec(s->objects, idx, s->bvh.nodes, 0);
    }
    return s;
}

struct Hit {
    const Object* object;
    size_t index;
    float distance;
};

// mod.rs:406-447
inline bool scene_intersect(const Scene& sc, const Ray& ray, Hit* hit, std::vector<size_t>& stack, Counters* cnt) {
    cnt->intersect_calls++;
    BvhRay bray = ray_to_bvh_ray(ray);
    const std::vector<BVHNode>& nodes = sc.bvh.nodes;
    stack.clear();
    if (nodes.empty()) return false;
    stack.push_back(0);
    float min_dist = std::numeric_limits<float>::infinity();
    const Object* min_object = 0;
    size_t min_index = 0;
    while (!stack.empty()) {
        size_t ni = stack.back();
        stack.pop_back();
        const BVHNode& n = nodes[ni];
        cnt->nodes_visited++;
        if (!n.is_leaf) {
            if (intersects_aabb(bray, n.child_l_aabb)) stack.push_back(n.child_l_index);
            if (intersects_aabb(bray, n.child_r_aabb)) stack.push_back(n.child_r_index);
        } else {
            const Object& obj = sc.objects[n.shape_index];
            cnt->prims_tested++;
            float d;
            if (object_intersect(obj, ray, &d)) {
                if (d < min_dist) {
                    min_dist = d;
                    min_object = &obj;
                    min_index = n.shape_index;
                }
            }
        }
    }
    if (!min_object) return false;
    hit->object = min_object;
    hit->index = min_index;
    hit->distance = min_dist;
    return true;
}

// mod.rs:452-491
inline bool scene_is_occluded(const Scene& sc, const Ray& ray, float max_dist, std::vector<size_t>& stack, Counters* cnt) {
    cnt->occluded_calls++;
    BvhRay bray = ray_to_bvh_ray(ray);
    const std::vector<BVHNode>& nodes = sc.bvh.nodes;
    stack.clear();
    if (nodes.empty()) return false;
    stack.push_back(0);
    while (!stack.empty()) {
        size_t ni = stack.back();
        stack.pop_back();
        const BVHNode& n = nodes[ni];
        cnt->nodes_visited++;
        if (!n.is_leaf) {
            if (intersects_aabb(bray, n.child_l_aabb)) stack.push_back(n.child_l_index);
            if (intersects_aabb(bray, n.child_r_aabb)) stack.push_back(n.child_r_index);
        } else {
            const Object& obj = sc.objects[n.shape_index];
            cnt->prims_tested++;
            float d;
            if (object_intersect(obj, ray, &d)) {
                if (d > 0.0f && d < max_dist) {
                    if (is_black(obj.mat.emittance)) return true;
                }
            }
        }
    }
    return false;
}

// mod.rs:58-67
inline V3 hit_point(const Ray& ray, float distance) {
    float min_dist = distance - EPS;
    V3 scaled = ray.direction * min_dist;
    return ray.origin + scaled;
}

// ------------------------------------------------------------------------------------------
// Render parameters: same byte layout as rt_render_params in include/rustracer_hip.h.
struct RenderParams {
    uint32_t width, height;
    float inv_w, inv_h, start, total, aspect_ratio, z;     // raytracer.rs:100-124
    float cam[3];
    uint32_t spp, light_samples, bounces;
    uint32_t mode;              // 0 = Debug (raytracer.rs:353), 1 = Full (raytracer.rs:167)
    uint32_t direct_estimator;  // 0 = light importance (raytracer.rs:251), 1 = hemisphere (build-defined)
    uint64_t seed;
    uint32_t tile_w, tile_h, shard_index, shard_count;  // screen-tile sharding (no reference counterpart)
};

// raytracer.rs:110-124.  sin through orc_sinf (see header).
void screen_params(uint32_t w, uint32_t h, float fov, RenderParams* p) {
    float wf = (float)w, hf = (float)h;
    float half_fov = fov * 0.5f;
    float start = orc_sinf(-half_fov);
    p->width = w; p->height = h;
    p->inv_w = 1.0f / wf; p->inv_h = 1.0f / hf;
    p->start = start;
    p->total = -2.0f * start;
    p->aspect_ratio = wf / hf;
    p->z = 1.7f;
}
// raytracer.rs:127-133
inline V3 screen_to_world(const RenderParams& p, uint32_t i, uint32_t j) {
    float iw = ((float)i + 0.5f) * p.inv_w;
    float jh = ((float)j + 0.5f) * p.inv_h;
    float xi = (p.start + iw * p.total) * p.aspect_ratio;
    float yi = -p.start - jh * p.total;
    return normalized(v3(xi, yi, -p.z));
}

struct Ctx {
    const Scene* scene;
    const RenderParams* params;
    Rng rng;
    std::vector<size_t> stack, shadow_stack;
    Counters cnt;
};

Spectrum cast_ray(Ctx& c, Ray ray, uint32_t bounces_left);

// raytracer.rs:251-286 (+ the build-defined hemisphere estimator)
Spectrum one_bounce_radiance(Ctx& c, const Hit& hit, const Ray& ray, V3 point, V3 normal) {
    const Scene& sc = *c.scene;
    Spectrum l = black();
    const Object& object = *hit.object;
    uint32_t num_light_samples = c.params->light_samples;
    float inv_light_samples = 1.0f / (float)num_light_samples;
    if (c.params->direct_estimator == 0) {
        for (size_t li = 0; li < sc.light_indexes.size(); ++li) {
            const Object& light = sc.objects[sc.light_indexes[li]];
            Spectrum light_emittance = light.mat.emittance;
            Spectrum color = black();
            for (uint32_t s = 0; s < num_light_samples; ++s) {
                LightSample sample = sample_l(light, point, &c.rng);
                Ray shadow_ray = ray_prenorm(point, sample.wi);
                if (!scene_is_occluded(sc, shadow_ray, sample.distance, c.shadow_stack, &c.cnt)) {
                    Spectrum reflected = object_bsdf(object);
                    float cos_theta = std::fabs(dot(sample.wi, normal));
                    color = color + light_emittance * reflected * cos_theta * sample.pdf;
                }
            }
            l = l + color * inv_light_samples;
        }
    } else {
        Spectrum color = black();
        for (uint32_t s = 0; s < num_light_samples; ++s) {
            V3 wi = to_coord_space(random_hemisphere(&c.rng), normal);
            Ray r = ray_new(point, wi);
            Hit h2;
            if (scene_intersect(sc, r, &h2, c.stack, &c.cnt)) {
                Spectrum reflected = object_bsdf(object);
                float cos_theta = std::fabs(dot(wi, normal));
                color = color + h2.object->mat.emittance * reflected * cos_theta * (2.0f * PI_F);
            }
        }
        l = l + color * inv_light_samples;
    }
    (void)ray;
    l = l + object.mat.emittance;   // raytracer.rs:284 (zero_bounce_radiance, raytracer.rs:245-247)
    return l;
}

// raytracer.rs:289-326
Spectrum global_illumination(Ctx& c, const Hit& hit, const Ray& ray, uint32_t bounces_left) {
    const Object& object = *hit.object;
    V3 point = hit_point(ray, hit.distance);
    V3 normal = surface_normal(object, point);
    Spectrum l = one_bounce_radiance(c, hit, ray, point, normal);
    if (!(rng_f32(&c.rng) <= RUSSIAN_ROULETTE_PROBABILITY)) return l;   // common.rs:136-138
    V3 wo = ray.direction;
    BSDFSample sample = sample_bsdf(object, wo, normal, &c.rng);
    Ray bounced = ray_new(point, sample.wi);
    Spectrum color = cast_ray(c, bounced, bounces_left - 1);
    if (!is_black(color)) {
        float cos_theta = std::fabs(dot(sample.wi, normal));
        color = color * sample.reflected * cos_theta * sample.pdf;
    }
    l = l + color;
    return l;
}

// raytracer.rs:330-350
Spectrum cast_ray(Ctx& c, Ray ray, uint32_t bounces_left) {
    Hit hit;
    if (scene_intersect(*c.scene, ray, &hit, c.stack, &c.cnt)) {
        if (bounces_left == 0) return hit.object->mat.emittance;
        if (bounces_left == 1) {
            V3 pt = hit_point(ray, hit.distance);
            V3 n = surface_normal(*hit.object, pt);
            return one_bounce_radiance(c, hit, ray, pt, n);
        }
        return global_illumination(c, hit, ray, bounces_left);
    }
    return black();
}

inline void set_pixel(uint8_t* rgba, uint32_t width, uint32_t x, uint32_t y, Spectrum s) {  // raytracer.rs:43-51
    size_t index = ((size_t)y * width + x) * 4;
    rgba[index] = to_color(s.r);
    rgba[index + 1] = to_color(s.g);
    rgba[index + 2] = to_color(s.b);
    rgba[index + 3] = 255;
}

inline bool pixel_owned(const RenderParams& p, uint32_t i, uint32_t j) {
    if (p.shard_count <= 1) return true;
    uint32_t tiles_x = (p.width + p.tile_w - 1) / p.tile_w;
    uint32_t t = (j / p.tile_h) * tiles_x + (i / p.tile_w);
    return (t % p.shard_count) == p.shard_index;
}

// raytracer.rs:223-241
Spectrum render_helper(Ctx& c, uint32_t i, uint32_t j, float* radiance_out) {
    const RenderParams& p = *c.params;
    V3 cam = v3(p.cam[0], p.cam[1], p.cam[2]);
    Ray ray = ray_prenorm(cam, screen_to_world(p, i, j));
    Spectrum color = black();
    uint32_t pixel_index = j * p.width + i;
    for (uint32_t s = 0; s < p.spp; ++s) {
        c.rng = rng_for(p.seed, pixel_index, s);
        color = color + cast_ray(c, ray, p.bounces);
    }
    color = color * (1.0f / (float)p.spp);
    if (radiance_out) { radiance_out[0] = color.r; radiance_out[1] = color.g; radiance_out[2] = color.b; }
    return color;
}
// raytracer.rs:393-416
Spectrum debug_render_helper(Ctx& c, uint32_t i, uint32_t j) {
    const RenderParams& p = *c.params;
    V3 cam = v3(p.cam[0], p.cam[1], p.cam[2]);
    Ray ray = ray_prenorm(cam, screen_to_world(p, i, j));
    Hit hit;
    if (scene_intersect(*c.scene, ray, &hit, c.stack, &c.cnt)) {
        float max_distance = 100.0f;
        float d = hit.distance;
        float dm = (d != d) ? max_distance : (d < max_distance ? d : max_distance);   // f32::min
        float distance_factor = 1.0f - (dm / max_distance);
        return sp(0.7f * distance_factor, 0.7f * distance_factor, 0.7f * distance_factor);
    }
    return black();
}

// ------------------------------------------------------------------------------------------
// Scenes (mod.rs:108-401) and the OBJ loader (mod.rs:135-189; tobj 2.0.3 restated).
struct CornellBox {
    std::vector<Object> triangles;
    float half_length, box_z_offset;
    Material red, green, blue, grey;
    Object sphere_light;
};
Spectrum grey_s() { return sp(0.78f, 0.78f, 0.78f); }   // common.rs:80-82

CornellBox cornell_box() {   // mod.rs:266-331
    CornellBox cb;
    float half_length = 20.0f, box_z_offset = -48.0f;
    cb.half_length = half_length; cb.box_z_offset = box_z_offset;
    cb.red = material(BSDF_DIFFUSE, sp(1, 0, 0), black());
    cb.blue = material(BSDF_DIFFUSE, sp(0, 0, 1), black());
    cb.green = material(BSDF_DIFFUSE, sp(0, 1, 0), black());
    cb.grey = material(BSDF_DIFFUSE, grey_s(), black());
    Material white_light = material(BSDF_DIFFUSE, black(), sp(1, 1, 1));
    float light_radius = 7.0f;
    cb.sphere_light = sphere_new(v3(0.0f, half_length + light_radius * 0.6f, box_z_offset - half_length / 2.0f), light_radius, white_light);
    float z = box_z_offset - half_length;
    V3 p0 = v3(-half_length, -half_length, 1.0f), p1 = v3(-half_length, -half_length, z), p2 = v3(half_length, -half_length, z),
       p3 = v3(half_length, -half_length, 1.0f), p4 = v3(-half_length, half_length, z), p5 = v3(half_length, half_length, z),
       p6 = v3(-half_length, half_length, 1.0f), p7 = v3(half_length, half_length, 1.0f), p8 = v3(-half_length, half_length, 1.0f),
       p9 = v3(-half_length, half_length, z), p10 = v3(half_length, half_length, z), p11 = v3(half_length, half_length, 1.0f);
    std::vector<Object>& t = cb.triangles;
    t.push_back(triangle_new_without_vn(p1, p0, p2, cb.grey));
    t.push_back(triangle_new_without_vn(p3, p2, p0, cb.grey));
    t.push_back(triangle_new_without_vn(p4, p5, p6, cb.grey));
    t.push_back(triangle_new_without_vn(p7, p6, p5, cb.grey));
    t.push_back(triangle_new_without_vn(p4, p1, p2, cb.green));
    t.push_back(triangle_new_without_vn(p2, p5, p4, cb.green));
    t.push_back(triangle_new_without_vn(p8, p0, p9, cb.red));
    t.push_back(triangle_new_without_vn(p1, p9, p0, cb.red));
    t.push_back(triangle_new_without_vn(p3, p11, p2, cb.blue));
    t.push_back(triangle_new_without_vn(p10, p2, p11, cb.blue));
    return cb;
}

// Corner-wise OBJ reader equivalent to tobj::load_obj(path, triangulate=true) + mod.rs:135-189:
// only the first model, fan triangulation, corner normal = the corner's vn.
bool load_obj(const char* filename, float scale, V3 offset, Material m, std::vector<Object>* out, std::string* err) {
    FILE* f = std::fopen(filename, "r");
    if (!f) { *err = std::string("cannot open ") + filename; return false; }
    std::vector<V3> pos, nrm;
    struct Corner { long v, n; };
    std::vector<std::vector<Corner> > faces;
    char line[4096];
    bool have_face = false;
    while (std::fgets(line, sizeof line, f)) {
        char* s = line;
        while (*s == ' ' || *s == '\t') ++s;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            V3 p;
            if (std::sscanf(s + 1, "%f %f %f", &p.x, &p.y, &p.z) == 3) pos.push_back(p);
        } else if (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t')) {
            V3 p;
            if (std::sscanf(s + 2, "%f %f %f", &p.x, &p.y, &p.z) == 3) nrm.push_back(p);
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            std::vector<Corner> cs;
            char* q = s + 1;
            while (*q) {
                while (*q == ' ' || *q == '\t') ++q;
                if (*q == '\0' || *q == '\n' || *q == '\r') break;
                Corner c = {0, 0};
                c.v = std::strtol(q, &q, 10);
                if (*q == '/') {
                    ++q;
                    if (*q != '/') std::strtol(q, &q, 10);  // vt ignored
                    if (*q == '/') { ++q; c.n = std::strtol(q, &q, 10); }
                }
                if (c.v < 0) c.v = (long)pos.size() + c.v + 1;
                if (c.n < 0) c.n = (long)nrm.size() + c.n + 1;
                cs.push_back(c);
                while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') ++q;
            }
            if (cs.size() >= 3) { faces.push_back(cs); have_face = true; }
        } else if ((s[0] == 'o' || s[0] == 'g') && (s[1] == ' ' || s[1] == '\t' || s[1] == '\n' || s[1] == '\r')) {
            if (have_face) break;   // second model starts: mod.rs:137 uses models[0] only
        }
    }
    std::fclose(f);
    bool has_normals = false;
    for (size_t i = 0; i < faces.size() && !has_normals; ++i) has_normals = faces[i][0].n > 0;
    for (size_t i = 0; i < faces.size(); ++i) {
        const std::vector<Corner>& cs = faces[i];
        for (size_t k = 1; k + 1 < cs.size(); ++k) {
            const Corner c3[3] = {cs[0], cs[k], cs[k + 1]};
            V3 p[3], n[3] = {v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)};
            for (int j = 0; j < 3; ++j) {
                if (c3[j].v < 1 || (size_t)c3[j].v > pos.size()) { *err = "face index out of range"; return false; }
                p[j] = offset + (pos[c3[j].v - 1] * scale);            // mod.rs:147
                if (has_normals) {
                    if (c3[j].n < 1 || (size_t)c3[j].n > nrm.size()) { *err = "normal index out of range"; return false; }
                    n[j] = nrm[c3[j].n - 1];
                }
            }
            out->push_back(has_normals ? triangle_new(p[0], p[1], p[2], n[0], n[1], n[2], m)
                                       : triangle_new_without_vn(p[0], p[1], p[2], m));
        }
    }
    return true;
}

// Mesh given as arrays (used for the procedural dragon-class stand-in): same placement arithmetic
// as load_obj (mod.rs:140-187).
void mesh_to_triangles(const float* vpos, const float* vnrm, size_t nv, const uint32_t* faces, size_t nf, float scale,
                       V3 offset, Material m, std::vector<Object>* out) {
    (void)nv;
    for (size_t f = 0; f < nf; ++f) {
        V3 p[3], n[3] = {v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)};
        for (int j = 0; j < 3; ++j) {
            uint32_t i = faces[3 * f + j];
            p[j] = offset + (v3(vpos[3 * i], vpos[3 * i + 1], vpos[3 * i + 2]) * scale);
            if (vnrm) n[j] = v3(vnrm[3 * i], vnrm[3 * i + 1], vnrm[3 * i + 2]);
        }
        out->push_back(vnrm ? triangle_new(p[0], p[1], p[2], n[0], n[1], n[2], m) : triangle_new_without_vn(p[0], p[1], p[2], m));
    }
}

thread_local std::string g_err;

Scene* scene_named(const char* name, const char* obj_path, const float* vpos, const float* vnrm, size_t nv,
                   const uint32_t* faces, size_t nf) {
    std::string n(name);
    CornellBox cb = cornell_box();
    float half_length = cb.half_length, box_z_offset = cb.box_z_offset;
    float sphere_radius = 6.0f;
    std::vector<Object> tris, sph;
    if (n == "triangle") {   // mod.rs:108-133
        Material mat = material(BSDF_SPECULAR, sp(1, 1, 1), black());
        Object light = sphere_new(v3(0.0f, 0.0f, 10.0f), 8.0f, material(BSDF_DIFFUSE, black(), sp(1, 1, 1)));
        tris.push_back(triangle_new(v3(-5, -5, -20), v3(5, -5, -20), v3(5, 5, -20), normalized(v3(-0.4f, 0.0f, 1.0f)),
                                    normalized(v3(0.4f, 0.0f, 1.0f)), normalized(v3(0.0f, 0.0f, 1.0f)), mat));
        sph.push_back(light);
    } else if (n == "specular" || n == "diffuse") {   // mod.rs:333-401
        tris = cb.triangles;
        Material first = n == "specular" ? material(BSDF_SPECULAR, sp(1, 1, 1), black()) : cb.grey;
        sph.push_back(cb.sphere_light);
        sph.push_back(sphere_new(v3(-half_length / 3.0f, -half_length + sphere_radius, box_z_offset - 2.0f * half_length / 3.0f), sphere_radius, first));
        sph.push_back(sphere_new(v3(half_length / 3.0f, -half_length + sphere_radius, box_z_offset - half_length / 3.0f), sphere_radius, cb.red));
    } else if (n == "teapot" || n == "specular_teapot") {   // mod.rs:229-264; "specular_teapot" = BASELINE config 5 (composed)
        tris = cb.triangles;
        Material mat = material(BSDF_DIFFUSE, grey_s(), black());
        if (!load_obj(obj_path, 0.13f, v3(-half_length / 3.0f - 2.0f, -15.0f, box_z_offset - 2.5f * half_length / 3.0f), mat, &tris, &g_err)) return 0;
        sph.push_back(cb.sphere_light);
        if (n == "specular_teapot") {
            // the mirror sphere of new_specular (mod.rs:345-353) moved clear of the teapot, plus the red one
            sph.push_back(sphere_new(v3(half_length / 3.0f + 2.0f, -half_length + sphere_radius, box_z_offset - 2.0f * half_length / 3.0f), sphere_radius,
                                     material(BSDF_SPECULAR, sp(1, 1, 1), black())));
            sph.push_back(sphere_new(v3(half_length / 3.0f, -half_length + 4.0f, box_z_offset + half_length / 3.0f), 4.0f, cb.red));
        } else {
            sph.push_back(sphere_new(v3(half_length / 3.0f, -half_length + sphere_radius, box_z_offset - half_length / 3.0f), sphere_radius, cb.red));
        }
    } else if (n == "dragon") {   // mod.rs:191-227
        tris = cb.triangles;
        Material mat = material(BSDF_DIFFUSE, grey_s(), black());
        V3 off = v3(-half_length / 3.0f, -half_length, box_z_offset - 2.0f * half_length / 3.0f);
        if (faces) mesh_to_triangles(vpos, vnrm, nv, faces, nf, 2.0f, off, mat, &tris);
        else if (!load_obj(obj_path, 2.0f, off, mat, &tris, &g_err)) return 0;
        sph.push_back(cb.sphere_light);
        sph.push_back(sphere_new(v3(half_length / 3.0f + 2.0f, -half_length + sphere_radius, box_z_offset - half_length / 3.0f + 2.0f), sphere_radius, cb.red));
    } else if (n == "sphere_on_plane") {   // BASELINE config 1 (new scene; screenshots/sphere_on_top_of_plane.png)
        V3 a = v3(-60.0f, -20.0f, 10.0f), b = v3(-60.0f, -20.0f, -140.0f), c2 = v3(60.0f, -20.0f, -140.0f), d = v3(60.0f, -20.0f, 10.0f);
        tris.push_back(triangle_new_without_vn(b, a, c2, cb.grey));
        tris.push_back(triangle_new_without_vn(d, c2, a, cb.grey));
        sph.push_back(cb.sphere_light);
        sph.push_back(sphere_new(v3(0.0f, -20.0f + 8.0f, -48.0f), 8.0f, cb.grey));
    } else {
        g_err = "unknown scene " + n;
        return 0;
    }
    return scene_new(tris, sph);
}

void render_columns(const Scene* sc, const RenderParams* p, uint8_t* rgba, float* radiance, uint32_t* ids, float* ts,
                    uint32_t nthreads, Counters* total) {
    std::atomic<uint32_t> next(0);
    std::vector<Counters> per(nthreads);
    auto work = [&](uint32_t tid) {
        Ctx c;
        c.scene = sc; c.params = p;
        std::memset(&c.cnt, 0, sizeof c.cnt);
        c.stack.reserve(64); c.shadow_stack.reserve(64);
        for (;;) {
            uint32_t i = next.fetch_add(1);                     // one task per column: raytracer.rs:198
            if (i >= p->width) break;
            for (uint32_t j = 0; j < p->height; ++j) {
                if (!pixel_owned(*p, i, j)) continue;
                size_t pi = (size_t)j * p->width + i;
                if (ids) {
                    V3 cam = v3(p->cam[0], p->cam[1], p->cam[2]);
                    Ray ray = ray_prenorm(cam, screen_to_world(*p, i, j));
                    Hit h;
                    if (scene_intersect(*sc, ray, &h, c.stack, &c.cnt)) { ids[pi] = (uint32_t)h.index; if (ts) ts[pi] = h.distance; }
                    else { ids[pi] = 0xFFFFFFFFu; if (ts) ts[pi] = std::numeric_limits<float>::infinity(); }
                    continue;
                }
                Spectrum color = p->mode == 1 ? render_helper(c, i, j, radiance ? radiance + 3 * pi : 0) : debug_render_helper(c, i, j);
                if (rgba) set_pixel(rgba, p->width, i, j, color);
            }
        }
        per[tid] = c.cnt;
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < nthreads; ++t) th.push_back(std::thread(work, t));
    work(0);
    for (size_t t = 0; t < th.size(); ++t) th[t].join();
    std::memset(total, 0, sizeof *total);
    for (uint32_t t = 0; t < nthreads; ++t) {
        total->intersect_calls += per[t].intersect_calls;
        total->occluded_calls += per[t].occluded_calls;
        total->nodes_visited += per[t].nodes_visited;
        total->prims_tested += per[t].prims_tested;
    }
}

}  // namespace

// ==========================================================================================
// C ABI for ctypes (tests/, smoke(), bench.py cpu_baseline).
extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// Raw scene: same argument layout as rt_scene_create (include/rustracer_hip.h).
// tri_pos/tri_nrm: 9 floats per triangle (stored, i.e. already normalised normals); mats: 8 x u32/f32 words
// {bsdf, refl rgb, emit rgb, pad}.
void* orc_scene_create(const float* tri_pos, const float* tri_nrm, const uint32_t* tri_mat, uint32_t T, const float* sph,
                       const uint32_t* sph_mat, uint32_t S, const void* mats, uint32_t M) {
    const uint32_t* mw = (const uint32_t*)mats;
    std::vector<Material> ms(M);
    for (uint32_t i = 0; i < M; ++i) {
        ms[i].bsdf = mw[8 * i];
        std::memcpy(&ms[i].reflectance, mw + 8 * i + 1, 12);
        std::memcpy(&ms[i].emittance, mw + 8 * i + 4, 12);
    }
    std::vector<Object> tris(T), sphs(S);
    for (uint32_t i = 0; i < T; ++i) {
        Object o;
        std::memset(&o, 0, sizeof o);
        const float* p = tri_pos + 9 * i; const float* n = tri_nrm + 9 * i;
        o.p1 = v3(p[0], p[1], p[2]); o.p2 = v3(p[3], p[4], p[5]); o.p3 = v3(p[6], p[7], p[8]);
        o.vn1 = v3(n[0], n[1], n[2]); o.vn2 = v3(n[3], n[4], n[5]); o.vn3 = v3(n[6], n[7], n[8]);
        if (tri_mat[i] >= M) { g_err = "material index out of range"; return 0; }
        o.mat = ms[tri_mat[i]];
        tris[i] = o;
    }
    for (uint32_t i = 0; i < S; ++i) {
        if (sph_mat[i] >= M) { g_err = "material index out of range"; return 0; }
        sphs[i] = sphere_new(v3(sph[4 * i], sph[4 * i + 1], sph[4 * i + 2]), sph[4 * i + 3], ms[sph_mat[i]]);
    }
    return scene_new(tris, sphs);
}

void* orc_scene_new_named(const char* name, const char* obj_path) { return scene_named(name, obj_path ? obj_path : "", 0, 0, 0, 0, 0); }
void* orc_scene_new_dragon_mesh(const float* vpos, const float* vnrm, uint32_t nv, const uint32_t* faces, uint32_t nf) {
    return scene_named("dragon", "", vpos, vnrm, nv, faces, nf);
}
void orc_scene_destroy(void* h) { delete (Scene*)h; }

void orc_scene_counts(void* h, uint32_t* T, uint32_t* S, uint32_t* M, uint32_t* nlights, uint32_t* nnodes) {
    Scene* s = (Scene*)h;
    uint32_t t = 0, sp_ = 0;
    std::vector<Material> ms;
    for (size_t i = 0; i < s->objects.size(); ++i) {
        if (s->objects[i].is_sphere) ++sp_; else ++t;
        bool found = false;
        for (size_t k = 0; k < ms.size() && !found; ++k) found = same_material(ms[k], s->objects[i].mat);
        if (!found) ms.push_back(s->objects[i].mat);
    }
    *T = t; *S = sp_; *M = (uint32_t)ms.size(); *nlights = (uint32_t)s->light_indexes.size(); *nnodes = (uint32_t)s->bvh.nodes.size();
}
// Export the object list in the raw layout (materials de-duplicated in order of first use).
void orc_scene_export(void* h, float* tri_pos, float* tri_nrm, uint32_t* tri_mat, float* sph, uint32_t* sph_mat, void* mats, uint32_t* lights) {
    Scene* s = (Scene*)h;
    std::vector<Material> ms;
    uint32_t t = 0, sp_ = 0;
    for (size_t i = 0; i < s->objects.size(); ++i) {
        const Object& o = s->objects[i];
        uint32_t mi = 0;
        for (; mi < ms.size(); ++mi) if (same_material(ms[mi], o.mat)) break;
        if (mi == ms.size()) ms.push_back(o.mat);
        if (!o.is_sphere) {
            const V3 p[3] = {o.p1, o.p2, o.p3}, n[3] = {o.vn1, o.vn2, o.vn3};
            for (int k = 0; k < 3; ++k) {
                tri_pos[9 * t + 3 * k] = p[k].x; tri_pos[9 * t + 3 * k + 1] = p[k].y; tri_pos[9 * t + 3 * k + 2] = p[k].z;
                tri_nrm[9 * t + 3 * k] = n[k].x; tri_nrm[9 * t + 3 * k + 1] = n[k].y; tri_nrm[9 * t + 3 * k + 2] = n[k].z;
            }
            tri_mat[t++] = mi;
        } else {
            sph[4 * sp_] = o.center.x; sph[4 * sp_ + 1] = o.center.y; sph[4 * sp_ + 2] = o.center.z; sph[4 * sp_ + 3] = o.radius;
            sph_mat[sp_++] = mi;
        }
    }
    uint32_t* mw = (uint32_t*)mats;
    for (size_t i = 0; i < ms.size(); ++i) {
        mw[8 * i] = ms[i].bsdf;
        std::memcpy(mw + 8 * i + 1, &ms[i].reflectance, 12);
        std::memcpy(mw + 8 * i + 4, &ms[i].emittance, 12);
        mw[8 * i + 7] = 0;
    }
    for (size_t i = 0; i < s->light_indexes.size(); ++i) lights[i] = (uint32_t)s->light_indexes[i];
}
// BVH dump: per node 16 words {is_leaf, l_index, r_index, shape_index, l_min[3], l_max[3], r_min[3], r_max[3]}.
void orc_scene_bvh(void* h, uint32_t* words) {
    Scene* s = (Scene*)h;
    for (size_t i = 0; i < s->bvh.nodes.size(); ++i) {
        const BVHNode& n = s->bvh.nodes[i];
        uint32_t* w = words + 16 * i;
        w[0] = n.is_leaf; w[1] = (uint32_t)n.child_l_index; w[2] = (uint32_t)n.child_r_index; w[3] = (uint32_t)n.shape_index;
        std::memcpy(w + 4, &n.child_l_aabb, 24);
        std::memcpy(w + 10, &n.child_r_aabb, 24);
    }
}

void orc_screen_params(uint32_t w, uint32_t h, float fov, void* params) { screen_params(w, h, fov, (RenderParams*)params); }
void orc_screen_to_world(const void* params, uint32_t i, uint32_t j, float* out3) {
    V3 v = screen_to_world(*(const RenderParams*)params, i, j);
    out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
}

// Full or debug render into rgba (w*h*4, caller zero-fills; only owned pixels are written).
// radiance (optional, w*h*3 f32) receives the pre-tone-map pixel value. counters: 4 x u64.
int orc_render(void* h, const void* params, uint8_t* rgba, float* radiance, uint32_t nthreads, uint64_t* counters) {
    const RenderParams* p = (const RenderParams*)params;
    if (p->mode == 1 && (p->spp == 0 || p->light_samples == 0)) { g_err = "spp and light_samples must be > 0"; return 1; }
    Scene* s = (Scene*)h;
    if (p->mode == 1 && p->direct_estimator == 0)
        for (size_t i = 0; i < s->light_indexes.size(); ++i)
            if (!s->objects[s->light_indexes[i]].is_sphere) { g_err = "triangle lights: sample_l unimplemented (objects.rs:98-100)"; return 2; }
    if (nthreads == 0) nthreads = std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 1;
    Counters c;
    render_columns(s, p, rgba, radiance, 0, 0, nthreads, &c);
    if (counters) { counters[0] = c.intersect_calls; counters[1] = c.occluded_calls; counters[2] = c.nodes_visited; counters[3] = c.prims_tested; }
    return 0;
}
// Primary-ray cast: object index per pixel (0xFFFFFFFF = miss) and distance.
int orc_cast_primary(void* h, const void* params, uint32_t* ids, float* ts, uint32_t nthreads, uint64_t* counters) {
    const RenderParams* p = (const RenderParams*)params;
    if (nthreads == 0) nthreads = std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 1;
    Counters c;
    render_columns((Scene*)h, p, 0, 0, ids, ts, nthreads, &c);
    if (counters) { counters[0] = c.intersect_calls; counters[1] = c.occluded_calls; counters[2] = c.nodes_visited; counters[3] = c.prims_tested; }
    return 0;
}

// ---- unit-level entry points for the known-answer tests ----
// Arbitrary rays against the scene: closest hit (mod.rs:406) and occlusion (mod.rs:452).
void orc_intersect_rays(void* h, const float* org, const float* dir, uint32_t n, uint32_t* ids, float* ts) {
    Scene* s = (Scene*)h;
    std::vector<size_t> st;
    Counters c;
    std::memset(&c, 0, sizeof c);
    for (uint32_t i = 0; i < n; ++i) {
        Ray r = ray_prenorm(v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
        Hit hit;
        if (scene_intersect(*s, r, &hit, st, &c)) { ids[i] = (uint32_t)hit.index; ts[i] = hit.distance; }
        else { ids[i] = 0xFFFFFFFFu; ts[i] = std::numeric_limits<float>::infinity(); }
    }
}
void orc_occluded_rays(void* h, const float* org, const float* dir, const float* maxd, uint32_t n, uint8_t* out) {
    Scene* s = (Scene*)h;
    std::vector<size_t> st;
    Counters c;
    std::memset(&c, 0, sizeof c);
    for (uint32_t i = 0; i < n; ++i) {
        Ray r = ray_prenorm(v3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
        out[i] = scene_is_occluded(*s, r, maxd[i], st, &c) ? 1 : 0;
    }
}
int orc_tri_intersect(const float* p9, const float* org, const float* dir, float* t) {
    Object o = triangle_new_without_vn(v3(p9[0], p9[1], p9[2]), v3(p9[3], p9[4], p9[5]), v3(p9[6], p9[7], p9[8]), material(0, black(), black()));
    return object_intersect(o, ray_prenorm(v3(org[0], org[1], org[2]), v3(dir[0], dir[1], dir[2])), t) ? 1 : 0;
}
int orc_sphere_intersect(const float* c4, const float* org, const float* dir, float* t) {
    Object o = sphere_new(v3(c4[0], c4[1], c4[2]), c4[3], material(0, black(), black()));
    return object_intersect(o, ray_prenorm(v3(org[0], org[1], org[2]), v3(dir[0], dir[1], dir[2])), t) ? 1 : 0;
}
void orc_tri_normal(const float* p9, const float* n9, const float* point, float* out3) {
    Object o = triangle_new(v3(p9[0], p9[1], p9[2]), v3(p9[3], p9[4], p9[5]), v3(p9[6], p9[7], p9[8]), v3(n9[0], n9[1], n9[2]),
                            v3(n9[3], n9[4], n9[5]), v3(n9[6], n9[7], n9[8]), material(0, black(), black()));
    V3 n = surface_normal(o, v3(point[0], point[1], point[2]));
    out3[0] = n.x; out3[1] = n.y; out3[2] = n.z;
}
uint8_t orc_to_color(float v) { return to_color(v); }
void orc_rng_stream(uint64_t seed, uint32_t pixel_index, uint32_t sample, uint32_t n, float* out) {
    Rng r = rng_for(seed, pixel_index, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng_f32(&r);
}
void orc_rng_raw(uint64_t state, uint32_t n, uint32_t* out) {
    Rng r; r.s = state;
    for (uint32_t i = 0; i < n; ++i) out[i] = rng_u32(&r);
}
void orc_set_double_trig(int on) { g_double_trig = on != 0; }
float orc_math_sin(float x) { return orc_sinf(x); }
float orc_math_cos(float x) { return orc_cosf(x); }
float orc_math_acos(float x) { return orc_acosf(x); }
float orc_math_pow(float x, float y) { return orc_powf(x, y); }
// op: 0 sin, 1 cos, 2 acos (double-precision definitions); 5 sin, 6 cos, 7 acos (the samplers' single-precision definitions)
void orc_math_batch(uint32_t op, const float* x, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; ++i) {
        float s, c;
        switch (op) {
            case 0: out[i] = orc_sinf(x[i]); break;
            case 1: out[i] = orc_cosf(x[i]); break;
            case 2: out[i] = orc_acosf(x[i]); break;
            case 5: orc_sincos_s(x[i], &s, &c); out[i] = s; break;
            case 6: orc_sincos_s(x[i], &s, &c); out[i] = c; break;
            default: out[i] = orc_acos_s(x[i]); break;
        }
    }
}
void orc_random_sphere(uint64_t seed, uint32_t pixel, uint32_t sample, float* out3) {
    Rng r = rng_for(seed, pixel, sample); V3 v = random_sphere(&r); out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
}
void orc_random_hemisphere(uint64_t seed, uint32_t pixel, uint32_t sample, float* out3) {
    Rng r = rng_for(seed, pixel, sample); V3 v = random_hemisphere(&r); out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
}
void orc_to_coord_space(const float* v, const float* n, float* out3) {
    V3 r = to_coord_space(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2])); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
uint32_t orc_hardware_threads() { return std::thread::hardware_concurrency(); }

}  // extern "C"
