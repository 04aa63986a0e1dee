// mathcheck.cpp — exhaustive error measurement of the sampler trigonometry in rustracer_amd/csrc/rt_math.h
// (sincos_s, acos_s) against double-precision libm.  Host only; the functions are single IEEE operations, so the device
// computes the same bits (tests/test_gpu_parity.py::test_device_math_bit_exact compares them).
//   g++ -O2 -mfma -ffp-contract=off -pthread tools/mathcheck.cpp -o /tmp/mathcheck && /tmp/mathcheck [stride]
// Prints the maximum error in ulps over (a) every input the samplers can produce, (b) every `stride`-th float of the domain.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <thread>
#include <vector>

#include "../rustracer_amd/csrc/rt_math.h"

using namespace rt;

static double ulp_err(float got, double ref) {
    if (got != got) return 1e30;               // a NaN where a number is due
    if (ref == 0.0) return got == 0.0f ? 0.0 : 1e30;
    int e;
    frexp(ref, &e);
    return fabs((double)got - ref) / ldexp(1.0, e - 24);
}
struct Max {
    double sin_e = 0, cos_e = 0, acos_e = 0;
    float sin_x = 0, cos_x = 0, acos_x = 0;
    void sc(float x) {
        float s, c;
        sincos_s(x, &s, &c);
        double e = ulp_err(s, sin((double)x));
        if (e > sin_e) { sin_e = e; sin_x = x; }
        e = ulp_err(c, cos((double)x));
        if (e > cos_e) { cos_e = e; cos_x = x; }
    }
    void ac(float u) {
        const double e = ulp_err(acos_s(u), acos((double)u));
        if (e > acos_e) { acos_e = e; acos_x = u; }
    }
    void merge(const Max& o) {
        if (o.sin_e > sin_e) { sin_e = o.sin_e; sin_x = o.sin_x; }
        if (o.cos_e > cos_e) { cos_e = o.cos_e; cos_x = o.cos_x; }
        if (o.acos_e > acos_e) { acos_e = o.acos_e; acos_x = o.acos_x; }
    }
};

int main(int argc, char** argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
    const int T = (int)std::thread::hardware_concurrency() > 0 ? (int)std::thread::hardware_concurrency() : 4;
    const float TWO_PI = 2.0f * 3.14159265358979323846f;
    std::vector<Max> reach(T), sweep(T);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            // (a) reachable: xi = i * 2^-23 (fastrand f32), theta = 2 pi xi (geo.rs:106, 138), u = 1 - 2 xi (geo.rs:139), phi = acos(u)
            for (uint32_t i = (uint32_t)t; i < (1u << 23); i += (uint32_t)T) {
                const float xi = (float)i * (1.0f / 8388608.0f);
                reach[t].sc(TWO_PI * xi);
                const float u = 1.0f - 2.0f * xi;
                reach[t].ac(u);
                reach[t].sc(acos_s(u));
            }
            // (b) the domains: floats of [2^-32, 2 pi] and of [-1, 1]
            for (uint64_t b = 0x2f800000ull + (uint64_t)t * stride; b <= float_to_bits(TWO_PI); b += (uint64_t)T * stride) sweep[t].sc(bits_to_float((uint32_t)b));
            for (uint64_t b = (uint64_t)t * stride; b <= 0x3f800000ull; b += (uint64_t)T * stride) {
                sweep[t].ac(bits_to_float((uint32_t)b));
                sweep[t].ac(bits_to_float((uint32_t)b | 0x80000000u));
            }
        });
    for (auto& x : th) x.join();
    Max r, s;
    for (int t = 0; t < T; ++t) { r.merge(reach[t]); s.merge(sweep[t]); }
    printf("reachable inputs (2^23 values of xi): sin %.4f ulp at %.9g, cos %.4f ulp at %.9g, acos %.4f ulp at %.9g\n", r.sin_e, r.sin_x, r.cos_e, r.cos_x,
           r.acos_e, r.acos_x);
    printf("domain sweep (every %u-th float): sin %.4f ulp at %.9g, cos %.4f ulp at %.9g, acos %.4f ulp at %.9g\n", stride, s.sin_e, s.sin_x, s.cos_e, s.cos_x,
           s.acos_e, s.acos_x);
    float s0, c0;
    sincos_s(0.0f, &s0, &c0);
    printf("sincos_s(0) = (%g, %g); acos_s(1) = %g; acos_s(-1) = %.9g; acos_s(0) = %.9g\n", s0, c0, acos_s(1.0f), acos_s(-1.0f), acos_s(0.0f));
    return (r.sin_e < 1.0 && r.cos_e < 1.0 && r.acos_e < 1.0 && s.sin_e < 1.0 && s.cos_e < 1.0 && s.acos_e < 1.0) ? 0 : 1;
}
