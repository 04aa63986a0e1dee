// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for the instruction classes the
// path-tracing kernels are made of, at 1 / 2 / 4 / 8 waves per SIMD and with full, half, quarter and single-lane EXEC.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate && tools/valu_rate
// Every kernel runs REP iterations of a block of 64 independent instructions of one kind (8 accumulators round-robin, so a
// dependent instruction is 8 issues away); s_memtime brackets the loop.  Output: cycles (of the 100 MHz-independent shader
// clock counter: s_memrealtime is not used) per instruction per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BLOCK64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// one kernel per instruction kind; `lanes` selects the EXEC pattern: 64 = all, 32 = low half, 16 = low quarter, 1 = lane 0, 65 = even lanes
#define KERNEL(NAME, INSN_MACRO)                                                                                         \
    __global__ void __launch_bounds__(256) NAME(float* out, unsigned long long* cyc, int rep, int lanes, float seed) {    \
        float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        float b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;                                     \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;                                     \
        unsigned u0 = threadIdx.x * 3u + 1u, u1 = u0 + 1u, u2 = u0 + 2u, u3 = u0 + 3u, u4 = u0 + 4u, u5 = u0 + 5u, u6 = u0 + 6u, u7 = u0 + 7u; \
        unsigned long long q0 = u0, q1 = u1, q2 = u2, q3 = u3, q4 = u4, q5 = u5, q6 = u6, q7 = u7;                         \
        const float c1 = 1.0000001f + seed * 0.f, c2 = 1e-9f + seed * 0.f; const unsigned u0s = (unsigned)rep * 3u + 1u;                                                                           \
        const double e1 = 1.0000001, e2 = 1e-9;                                                                            \
        const unsigned lane = threadIdx.x & 63u;                                                                           \
        const bool on = lanes == 64 ? true : lanes == 65 ? !(lane & 1u) : lanes == 66 ? (lane & 15u) == 0u : lane < (unsigned)lanes; \
        unsigned long long t0 = 0, t1 = 0;                                                                                 \
        __syncthreads();                                                                                                   \
        t0 = __builtin_readcyclecounter();                                                                                 \
        if (on) {                                                                                                          \
            for (int r = 0; r < rep; ++r) { BLOCK64(INSN_MACRO) }                                                          \
        }                                                                                                                  \
        t1 = __builtin_readcyclecounter();                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + \
            (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7);               \
        if ((threadIdx.x & 63u) == 0u) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                 \
    }

#define I_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(c1), "v"(c2));
#define I_MUL32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(c1));
#define I_ADD32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(c2));
#define I_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##i) : "v"(e1), "v"(e2));
#define I_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##i) : "v"(e1));
#define I_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##i) : "v"(e2));
#define I_RCP32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
#define I_SQRT32(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##i));
#define I_RSQ32(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a##i));
#define I_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u##i) : "v"(u0));
#define I_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u##i) : "v"(u0));
#define I_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##i) : "v"(u##i), "v"(u0) : "vcc");
#define I_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u##i) : "v"(u0));
#define I_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u##i) : "v"(u0));
#define I_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u##i));
#define I_CVTUB(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a##i) : "v"(u##i));
#define I_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(c1), "v"(c2));
#define I_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(c1) : );
#define I_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a##i), "v"(c1) : "vcc");
#define I_LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a##i) : "v"(u0));
#define I_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d##i) : "v"(e1), "v"(e2));
#define I_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d##i) : "v"(e1));
#define I_DIVSCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a##i) : "v"(c1) : "vcc");
#define I_DIVFMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(c1), "v"(c2) : "vcc");
#define I_DIVFIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(c1), "v"(c2));
#define I_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d##i));
#define I_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a##i) : "v"(c1));
#define I_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(u##i));
#define I_MBCNT(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(u##i) : "v"(u0));
#define I_FMA32_SALU(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_add_u32 s20, s20, 1" : "+v"(a##i) : "v"(c1), "v"(c2) : "s20", "scc");
#define I_FMA32_2SALU(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_add_u32 s20, s20, 1\n s_and_b32 s21, s20, 7" : "+v"(a##i) : "v"(c1), "v"(c2) : "s20", "s21", "scc");
#define I_FMA32_LDS(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(c1), "v"(c2));


#define I_SUB32(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a##i) : "v"(c2));
#define I_MAX32(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##i) : "v"(c1));
#define I_MIN32(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a##i) : "v"(c1));
#define I_MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(c1), "v"(c2));
#define I_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(c1), "v"(c2));
#define I_FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##i) : "v"(c1), "v"(c2));
#define I_FMAAK(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f800001" : "+v"(a##i) : "v"(c1));
#define I_MULS(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "s"(c1));
#define I_MULABS(i) asm volatile("v_mul_f32 %0, |%0|, %1" : "+v"(a##i) : "v"(c1));
#define I_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u##i) : "v"(u0));
#define I_OR(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u##i) : "v"(u0));
#define I_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u##i) : "v"(u0));
#define I_LSHR(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(u##i));
#define I_SUBU(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u##i) : "v"(u0));
#define I_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(u##i) : "v"(u0) : "vcc");
#define I_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u##i) : "v"(u0));
#define I_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u##i) : "v"(u0));
#define I_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u##i) : "v"(u0));
#define I_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u##i) : "v"(u0));
#define I_BFI(i) asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(u##i) : "v"(u0));
#define I_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u##i) : "v"(u0));
#define I_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(u##i) : "v"(u0));
#define I_CVTU(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a##i) : "v"(u##i));
#define I_CVTUB0(i) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a##i) : "v"(u##i));
#define I_CVTF2U(i) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u##i) : "v"(a##i));
#define I_RNDNE(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a##i));
#define I_CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(c1) : "vcc");
#define I_CMPS(i) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a##i), "v"(c1) : "s20", "s21");
#define I_CMPCNDS(i) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a##i) : "v"(c1) : "s20", "s21");
#define I_CNDS(i) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a##i) : "v"(c1));
#define I_CMPX(i) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(a##i), "v"(u0) : "vcc");
#define I_RDFL(i) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(a##i) : "s20");
#define I_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i));
#define I_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a##i) : "v"(u0));
// independent pairs: a fast-class and a slow-class instruction on different accumulators
#define I_P_FMA_MAX3(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_max3_f32 %1, %1, %2, %3" : "+v"(a##i), "+v"(b##i) : "v"(c1), "v"(c2));
#define I_P_FMA_CVT(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_cvt_f32_ubyte1 %1, %4" : "+v"(a##i), "=v"(b##i) : "v"(c1), "v"(c2), "v"(u##i));
#define I_P_MUL_ADD(i) asm volatile("v_mul_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(a##i), "+v"(b##i) : "v"(c1), "v"(c2));
#define I_P_MAX3_CVT(i) asm volatile("v_max3_f32 %0, %0, %2, %3\n v_cvt_f32_ubyte1 %1, %4" : "+v"(a##i), "=v"(b##i) : "v"(c1), "v"(c2), "v"(u##i));
#define I_P_FMA_RCP(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_rcp_f32 %1, %1" : "+v"(a##i), "+v"(b##i) : "v"(c1), "v"(c2));
#define I_P_MAX3_RCP(i) asm volatile("v_max3_f32 %0, %0, %2, %3\n v_rcp_f32 %1, %1" : "+v"(a##i), "+v"(b##i) : "v"(c1), "v"(c2));
#define I_P_FMA_FMA64(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f64 %1, %1, %4, %5" : "+v"(a##i), "+v"(d##i) : "v"(c1), "v"(c2), "v"(e1), "v"(e2));
#define I_P_FMA_AND(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_and_b32 %1, %1, %4" : "+v"(a##i), "+v"(u##i) : "v"(c1), "v"(c2), "v"(u0));
#define I_T_FMA_FMA_MAX3(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_mul_f32 %0, %0, %2\n v_max3_f32 %1, %1, %2, %3" : "+v"(a##i), "+v"(b##i) : "v"(c1), "v"(c2));

#define I_FMAS(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "s"(c1), "v"(c2));
#define I_FMASS(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "s"(c1));
#define I_ADDS(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a##i) : "s"(c2));
#define I_SUBS(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a##i) : "s"(c2));
#define I_ANDS(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u##i) : "s"(u0s));
#define I_ADDUS(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u##i) : "s"(u0s));
#define I_MOVS(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a##i) : "s"(c1));
#define I_MULC(i) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a##i));
#define I_MULL(i) asm volatile("v_mul_f32 %0, 0x3f800001, %0" : "+v"(a##i));
#define I_ADDC(i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a##i));
#define I_FMAC2(i) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a##i) : "v"(c2));
#define I_MAXS(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a##i) : "s"(c1));
#define I_LSHLV(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(u##i) : "v"(u0));
#define I_LSHRV(i) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(u##i) : "v"(u0));
#define I_ASHR(i) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(u##i));
#define I_NOT(i) asm volatile("v_not_b32 %0, %0" : "+v"(u##i));
#define I_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(u##i) : "v"(u0));
#define I_FFBL(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(u##i));
#define I_MINU(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u##i) : "v"(u0));
#define I_CMPU(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u##i), "v"(u0) : "vcc");
#define I_ADDF16(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u##i) : "v"(u0));

#define I_FMAMIX(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a##i) : "v"(u##i), "v"(c1));
#define I_FMAMIXHI(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a##i) : "v"(u##i), "v"(c1));
#define I_P_FMAMIX_MAX3(i) asm volatile("v_fma_mix_f32 %0, %4, %2, %0 op_sel_hi:[1,0,0]\n v_max3_f32 %1, %1, %2, %3" : "+v"(a##i), "+v"(b##i) : "v"(c1), "v"(c2), "v"(u##i));
#define I_CVTF16(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a##i) : "v"(u##i));

KERNEL(k_fma32, I_FMA32) KERNEL(k_mul32, I_MUL32) KERNEL(k_add32, I_ADD32) KERNEL(k_fma64, I_FMA64) KERNEL(k_mul64, I_MUL64) KERNEL(k_add64, I_ADD64)
KERNEL(k_rcp32, I_RCP32) KERNEL(k_sqrt32, I_SQRT32) KERNEL(k_rsq32, I_RSQ32) KERNEL(k_mullo, I_MULLO) KERNEL(k_mulhi, I_MULHI) KERNEL(k_mad64, I_MAD64)
KERNEL(k_mul24, I_MUL24) KERNEL(k_addu, I_ADDU) KERNEL(k_lshl, I_LSHL) KERNEL(k_cvtub, I_CVTUB) KERNEL(k_max3, I_MAX3) KERNEL(k_cndmask, I_CNDMASK)
KERNEL(k_cmp, I_CMP) KERNEL(k_ldexp, I_LDEXP) KERNEL(k_pkfma, I_PKFMA) KERNEL(k_pkmul, I_PKMUL) KERNEL(k_divscale, I_DIVSCALE) KERNEL(k_divfmas, I_DIVFMAS)
KERNEL(k_divfixup, I_DIVFIXUP) KERNEL(k_rcp64, I_RCP64) KERNEL(k_mov, I_MOV) KERNEL(k_bfe, I_BFE) KERNEL(k_mbcnt, I_MBCNT)
KERNEL(k_fma32_salu, I_FMA32_SALU) KERNEL(k_fma32_2salu, I_FMA32_2SALU)

KERNEL(kx0, I_SUB32) KERNEL(kx1, I_MAX32) KERNEL(kx2, I_MIN32) KERNEL(kx3, I_MIN3) KERNEL(kx4, I_MED3) KERNEL(kx5, I_FMAC) KERNEL(kx6, I_FMAAK) KERNEL(kx7, I_MULS) KERNEL(kx8, I_MULABS) KERNEL(kx9, I_AND) KERNEL(kx10, I_OR) KERNEL(kx11, I_XOR) KERNEL(kx12, I_LSHR) KERNEL(kx13, I_SUBU) KERNEL(kx14, I_ADDCO) KERNEL(kx15, I_ADD3) KERNEL(kx16, I_LSHLADD) KERNEL(kx17, I_LSHLOR) KERNEL(kx18, I_ANDOR) KERNEL(kx19, I_BFI) KERNEL(kx20, I_PERM) KERNEL(kx21, I_ALIGNBIT) KERNEL(kx22, I_CVTU) KERNEL(kx23, I_CVTUB0) KERNEL(kx24, I_CVTF2U) KERNEL(kx25, I_RNDNE) KERNEL(kx26, I_CMPCND) KERNEL(kx27, I_CMPS) KERNEL(kx28, I_CMPCNDS) KERNEL(kx29, I_CNDS) KERNEL(kx30, I_CMPX) KERNEL(kx31, I_RDFL) KERNEL(kx32, I_DPP) KERNEL(kx33, I_BPERM) KERNEL(kx34, I_P_FMA_MAX3) KERNEL(kx35, I_P_FMA_CVT) KERNEL(kx36, I_P_MUL_ADD) KERNEL(kx37, I_P_MAX3_CVT) KERNEL(kx38, I_P_FMA_RCP) KERNEL(kx39, I_P_MAX3_RCP) KERNEL(kx40, I_P_FMA_FMA64) KERNEL(kx41, I_P_FMA_AND) KERNEL(kx42, I_T_FMA_FMA_MAX3) KERNEL(ky0, I_FMAS) KERNEL(ky1, I_FMASS) KERNEL(ky2, I_ADDS) KERNEL(ky3, I_SUBS) KERNEL(ky4, I_ANDS) KERNEL(ky5, I_ADDUS) KERNEL(ky6, I_MOVS) KERNEL(ky7, I_MULC) KERNEL(ky8, I_MULL) KERNEL(ky9, I_ADDC) KERNEL(ky10, I_FMAC2) KERNEL(ky11, I_MAXS) KERNEL(ky12, I_LSHLV) KERNEL(ky13, I_LSHRV) KERNEL(ky14, I_ASHR) KERNEL(ky15, I_NOT) KERNEL(ky16, I_BCNT) KERNEL(ky17, I_FFBL) KERNEL(ky18, I_MINU) KERNEL(ky19, I_CMPU) KERNEL(ky20, I_ADDF16) KERNEL(kz0, I_FMAMIX) KERNEL(kz1, I_FMAMIXHI) KERNEL(kz2, I_P_FMAMIX_MAX3) KERNEL(kz3, I_CVTF16)
typedef void (*kern_t)(float*, unsigned long long*, int, int, float);
struct Entry { const char* name; kern_t k; int per; };

int main(int argc, char** argv) {
    const Entry all[] = {
        {"v_fma_f32", k_fma32, 1}, {"v_mul_f32", k_mul32, 1}, {"v_add_f32", k_add32, 1}, {"v_fma_f64", k_fma64, 1}, {"v_mul_f64", k_mul64, 1}, {"v_add_f64", k_add64, 1},
        {"v_pk_fma_f32", k_pkfma, 1}, {"v_pk_mul_f32", k_pkmul, 1}, {"v_rcp_f32", k_rcp32, 1}, {"v_sqrt_f32", k_sqrt32, 1}, {"v_rsq_f32", k_rsq32, 1}, {"v_rcp_f64", k_rcp64, 1},
        {"v_mul_lo_u32", k_mullo, 1}, {"v_mul_hi_u32", k_mulhi, 1}, {"v_mad_u64_u32", k_mad64, 1}, {"v_mul_u32_u24", k_mul24, 1}, {"v_add_u32", k_addu, 1},
        {"v_lshlrev_b32", k_lshl, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_mbcnt_lo", k_mbcnt, 1}, {"v_cvt_f32_ubyte1", k_cvtub, 1}, {"v_max3_f32", k_max3, 1}, {"v_cndmask_b32", k_cndmask, 1},
        {"v_cmp_lt_f32", k_cmp, 1}, {"v_ldexp_f32", k_ldexp, 1}, {"v_div_scale_f32", k_divscale, 1}, {"v_div_fmas_f32", k_divfmas, 1}, {"v_div_fixup_f32", k_divfixup, 1},
        {"v_mov_b32", k_mov, 1}, {"v_fma_f32 + 1 s_add", k_fma32_salu, 1}, {"v_fma_f32 + 2 salu", k_fma32_2salu, 1},
        {"v_fma_f32 1 sgpr", ky0, 1},
        {"v_fma_f32 sgpr twice", ky1, 1},
        {"v_add_f32 sgpr", ky2, 1},
        {"v_sub_f32 sgpr", ky3, 1},
        {"v_and_b32 sgpr", ky4, 1},
        {"v_add_u32 sgpr", ky5, 1},
        {"v_mov_b32 sgpr", ky6, 1},
        {"v_mul_f32 inline 2.0", ky7, 1},
        {"v_mul_f32 literal", ky8, 1},
        {"v_add_f32 inline 1.0", ky9, 1},
        {"v_fma_f32 inline 2.0", ky10, 1},
        {"v_max_f32 sgpr", ky11, 1},
        {"v_lshlrev_b32 vgpr amt", ky12, 1},
        {"v_lshrrev_b32 vgpr amt", ky13, 1},
        {"v_ashrrev_i32", ky14, 1},
        {"v_not_b32", ky15, 1},
        {"v_bcnt_u32_b32", ky16, 1},
        {"v_ffbl_b32", ky17, 1},
        {"v_min_u32", ky18, 1},
        {"v_cmp_lt_u32", ky19, 1},
        {"v_pk_add_f16", ky20, 1},
        {"v_fma_mix_f32 (f16 lo, f32, f32)", kz0, 1},
        {"v_fma_mix_f32 (f16 hi, f32, f32)", kz1, 1},
        {"PAIR fma_mix | max3", kz2, 2},
        {"v_cvt_f32_f16", kz3, 1},
        {"v_sub_f32", kx0, 1},
        {"v_max_f32", kx1, 1},
        {"v_min_f32", kx2, 1},
        {"v_min3_f32", kx3, 1},
        {"v_med3_f32", kx4, 1},
        {"v_fmac_f32", kx5, 1},
        {"v_fmaak_f32", kx6, 1},
        {"v_mul_f32 sgpr src", kx7, 1},
        {"v_mul_f32 |abs| (VOP3)", kx8, 1},
        {"v_and_b32", kx9, 1},
        {"v_or_b32", kx10, 1},
        {"v_xor_b32", kx11, 1},
        {"v_lshrrev_b32", kx12, 1},
        {"v_sub_u32", kx13, 1},
        {"v_add_co_u32", kx14, 1},
        {"v_add3_u32", kx15, 1},
        {"v_lshl_add_u32", kx16, 1},
        {"v_lshl_or_b32", kx17, 1},
        {"v_and_or_b32", kx18, 1},
        {"v_bfi_b32", kx19, 1},
        {"v_perm_b32", kx20, 1},
        {"v_alignbit_b32", kx21, 1},
        {"v_cvt_f32_u32", kx22, 1},
        {"v_cvt_f32_ubyte0", kx23, 1},
        {"v_cvt_u32_f32", kx24, 1},
        {"v_rndne_f32", kx25, 1},
        {"v_cmp + v_cndmask vcc", kx26, 2},
        {"v_cmp -> sgpr pair", kx27, 1},
        {"v_cmp + v_cndmask sgpr", kx28, 2},
        {"v_cndmask sgpr mask", kx29, 1},
        {"v_cmp_class_f32", kx30, 1},
        {"v_readfirstlane", kx31, 1},
        {"v_mov_b32 dpp", kx32, 1},
        {"ds_bpermute + wait", kx33, 1},
        {"PAIR fma | max3", kx34, 2},
        {"PAIR fma | cvt_ubyte", kx35, 2},
        {"PAIR mul | add", kx36, 2},
        {"PAIR max3 | cvt_ubyte", kx37, 2},
        {"PAIR fma | rcp", kx38, 2},
        {"PAIR max3 | rcp", kx39, 2},
        {"PAIR fma32 | fma64", kx40, 2},
        {"PAIR fma | and", kx41, 2},
        {"TRIPLE fma mul | max3", kx42, 3},

    };
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs, shader clock %d kHz, wall clock counter %d kHz\n", prop.gcnArchName, cus, prop.clockRate, prop.clockRate);
    const int rep = 4000;
    std::vector<unsigned long long> h_cyc((size_t)cus * 8 * 4);
    float* out;
    unsigned long long* cyc;
    const int max_blocks = cus * 8;
    CHECK(hipMalloc(&out, (size_t)max_blocks * 256 * sizeof(float)));
    CHECK(hipMalloc(&cyc, (size_t)max_blocks * 4 * sizeof(unsigned long long)));
    hipEvent_t ev0, ev1;
    CHECK(hipEventCreate(&ev0));
    CHECK(hipEventCreate(&ev1));
    // (waves per SIMD, lanes pattern)
    const int cfgs[][2] = {{1, 64}, {2, 64}, {4, 64}, {8, 64}, {4, 16}};
    printf("%-22s", "instruction");
    for (auto& c : cfgs) printf("  w%d/l%-3d      ", c[0], c[1]);
    printf("   (a/b: a = ns per instruction per SIMD x nominal shader GHz, from HIP events; b = s_memtime ticks per instruction per SIMD; wN = waves per SIMD, lN = lanes enabled, l65 = even lanes, l66 = every 16th lane)\n");
    for (const Entry& e : all) {
        if (argc > 1 && !strstr(e.name, argv[1]) && !(argc > 2 && strstr(e.name, argv[2]))) continue;
        printf("%-22s", e.name);
        for (auto& c : cfgs) {
            const int blocks = cus * c[0];            // one workgroup of 4 waves = 1 wave per SIMD of a CU
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, 10, c[1], 1.0f);   // warm
            CHECK(hipEventRecord(ev0));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, rep, c[1], 1.0f);
            CHECK(hipEventRecord(ev1));
            CHECK(hipEventSynchronize(ev1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, ev0, ev1));
            // instructions issued per SIMD: waves/SIMD * rep * 64
            const double per_simd = (double)c[0] * rep * 64.0 * e.per;
            const double ns = ms * 1e6 / per_simd;
            CHECK(hipMemcpy(h_cyc.data(), cyc, (size_t)blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < blocks * 4; ++i) sum += (double)h_cyc[i];
            // counter ticks per instruction per SIMD: a wave's ticks / (its instructions * waves sharing the SIMD)
            const double ticks = sum / (blocks * 4.0) / (rep * 64.0 * e.per) / c[0];
            printf("  %6.2f/%-6.2f", ns * (prop.clockRate / 1e6), ticks);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
