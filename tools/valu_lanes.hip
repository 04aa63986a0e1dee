// Follow-up to valu_rate.hip: how the cost of a wave64 VALU instruction depends on the number and position of enabled lanes,
// with the shader clock measured inside the kernel (s_memtime ticks per s_memrealtime tick, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_lanes.hip -o tools/valu_lanes && tools/valu_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BLOCK64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define I_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(c1), "v"(c2));
#define I_MUL32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(c1));
#define I_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u##i) : "v"(u0));
#define I_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(c1), "v"(c2));
#define I_RCP32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
// a realistic mix (what the node test of k_trace_p is made of)
#define I_MIX(i) asm volatile("v_cvt_f32_ubyte1 %0, %3\n v_fma_f32 %0, %0, %1, %2\n v_max3_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(c1), "v"(c2), "v"(u##i));
// 8 instructions with every lane enabled, then 8 with EXEC = mask given in s[22:23], inside one block
#define I_ALT(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n s_mov_b64 exec, %4\n v_fma_f32 %1, %1, %2, %3\n s_mov_b64 exec, -1" : "+v"(a##i), "+v"(b##i) : "v"(c1), "v"(c2), "s"(mask));

#define KERNEL(NAME, INSN_MACRO, PER)                                                                                    \
    __global__ void __launch_bounds__(256) NAME(float* out, unsigned long long* cyc, int rep, unsigned long long mask, float seed) { \
        float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        float b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;                                     \
        unsigned u0 = threadIdx.x * 3u + 1u, u1 = u0 + 1u, u2 = u0 + 2u, u3 = u0 + 3u, u4 = u0 + 4u, u5 = u0 + 5u, u6 = u0 + 6u, u7 = u0 + 7u; \
        const float c1 = 1.0000001f, c2 = 1e-9f;                                                                           \
        const unsigned lane = threadIdx.x & 63u;                                                                           \
        const bool on = (mask >> lane) & 1ull;                                                                             \
        __syncthreads();                                                                                                   \
        const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();                 \
        if (PER == 2) {                                                                                                    \
            for (int r = 0; r < rep; ++r) { BLOCK64(INSN_MACRO) }                                                          \
        } else if (on) {                                                                                                   \
            for (int r = 0; r < rep; ++r) { BLOCK64(INSN_MACRO) }                                                          \
        }                                                                                                                  \
        const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();                 \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7); \
        if ((threadIdx.x & 63u) == 0u) { cyc[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; } \
    }
KERNEL(k_fma32, I_FMA32, 1) KERNEL(k_mul32, I_MUL32, 1) KERNEL(k_mullo, I_MULLO, 1) KERNEL(k_max3, I_MAX3, 1) KERNEL(k_rcp32, I_RCP32, 1) KERNEL(k_mix, I_MIX, 1)
KERNEL(k_alt, I_ALT, 2)

typedef void (*kern_t)(float*, unsigned long long*, int, unsigned long long, float);
struct Entry { const char* name; kern_t k; int insts_per_macro; };

int main() {
    const Entry all[] = {{"v_fma_f32", k_fma32, 1}, {"v_mul_f32", k_mul32, 1}, {"v_mul_lo_u32", k_mullo, 1}, {"v_max3_f32", k_max3, 1}, {"v_rcp_f32", k_rcp32, 1},
                         {"cvt+fma+max3+mul", k_mix, 4}, {"fma all / fma masked", k_alt, 2}};
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, waves = 4, blocks = cus * waves, rep = 8000;
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
    CHECK(hipMalloc(&cyc, (size_t)blocks * 8 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h((size_t)blocks * 8);
    hipEvent_t ev0, ev1;
    CHECK(hipEventCreate(&ev0));
    CHECK(hipEventCreate(&ev1));
    struct M { const char* name; unsigned long long m; };
    const M masks[] = {{"64", ~0ull}, {"48", (1ull << 48) - 1}, {"32lo", 0xFFFFFFFFull}, {"32even", 0x5555555555555555ull}, {"16lo", 0xFFFFull}, {"16/4th", 0x1111111111111111ull},
                       {"12", 0xFFFull}, {"8lo", 0xFFull}, {"8/8th", 0x0101010101010101ull}, {"6", 0x3Full}, {"5", 0x1Full}, {"4lo", 0xFull}, {"4/16th", 0x0001000100010001ull}, {"3", 7ull}, {"2", 3ull}, {"1", 1ull},
                       {"1hi", 1ull << 63}, {"rand20", 0x0010842108421084ull | 0x8000000000000001ull}};
    printf("# %s, %d CUs, %d waves per SIMD, nominal %d MHz.  Per cell: ns per wave64 instruction per SIMD (HIP events) | measured shader MHz (s_memtime / s_memrealtime)\n", prop.gcnArchName, cus, waves, prop.clockRate / 1000);
    printf("%-22s", "lanes enabled:");
    for (auto& m : masks) printf(" %12s", m.name);
    printf("\n");
    for (const Entry& e : all) {
        printf("%-22s", e.name);
        for (auto& m : masks) {
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, 10, m.m, 1.0f);
            CHECK(hipEventRecord(ev0));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, rep, m.m, 1.0f);
            CHECK(hipEventRecord(ev1));
            CHECK(hipEventSynchronize(ev1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, ev0, ev1));
            CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double st = 0, sr = 0;
            for (int i = 0; i < blocks * 4; ++i) { st += (double)h[2 * i]; sr += (double)h[2 * i + 1]; }
            const double ns = ms * 1e6 / ((double)waves * rep * 64.0 * e.insts_per_macro);
            printf(" %6.3f|%5.0f", ns, sr > 0 ? st / sr * 100.0 : 0.0);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
