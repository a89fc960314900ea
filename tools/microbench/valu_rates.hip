// Issue cost of individual VALU instructions on gfx950, wave64, eight waves per SIMD, independent chains:
//   mkdir -p _build && hipcc --offload-arch=gfx950 -O2 -o _build/valu_rates valu_rates.hip && _build/valu_rates
// Prints ns per wave instruction per SIMD and that figure relative to v_mul_f32 (DESIGN.md section 5 quotes the table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define REP16(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#define KERNEL(NAME, ASM, ...)                                                                                         \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                                              \
        float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const float m = 1.0000001f, n = 0.5f; unsigned long long msk = 0x5555aaaa3333ccccull; unsigned long long so = 0; unsigned long long dd = threadIdx.x; unsigned si = 0;  \
        for (int i = 0; i < iters; i++) {                                                                              \
            REP16(ASM)                                                                                                 \
        }                                                                                                              \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)so + (float)dd + (float)si;                      \
    }
#define A2(OP, x) asm volatile(OP " %0, %0, %1" : "+v"(x) : "v"(m));
#define S(x) A2("v_mul_f32", x)
KERNEL(k_mul, S)
#undef S
#define S(x) A2("v_add_f32", x)
KERNEL(k_add, S)
#undef S
#define S(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
KERNEL(k_fma, S)
#undef S
#define S(x) A2("v_max_f32", x)
KERNEL(k_max, S)
#undef S
#define S(x) A2("v_min_f32", x)
KERNEL(k_min, S)
#undef S
#define S(x) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
KERNEL(k_max3, S)
#undef S
#define S(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
KERNEL(k_med3, S)
#undef S
#define S(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(m), "s"(msk));
KERNEL(k_cndmask, S)
#undef S
#define S(x) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(so) : "v"(x), "v"(m));
KERNEL(k_cmp, S)
#undef S
#define S(x) A2("v_and_b32", x)
KERNEL(k_and, S)
#undef S
#define S(x) A2("v_xor_b32", x)
KERNEL(k_xor, S)
#undef S
#define S(x) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));
KERNEL(k_lshl, S)
#undef S
#define S(x) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x));
KERNEL(k_alignbit, S)
#undef S
#define S(x) A2("v_add_u32", x)
KERNEL(k_addu, S)
#undef S
#define S(x) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(m));
KERNEL(k_lshl_add, S)
#undef S
#define S(x) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
KERNEL(k_add3, S)
#undef S
#define S(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(m));
KERNEL(k_mov, S)
#undef S
#define S(x) A2("v_mul_lo_u32", x)
KERNEL(k_mul_lo, S)
#undef S
#define S(x) A2("v_mul_u32_u24", x)
KERNEL(k_mul24, S)
#undef S
#define S(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
KERNEL(k_mad24, S)
#undef S
#define S(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
KERNEL(k_rcp, S)
#undef S
#define S(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
KERNEL(k_sqrt, S)
#undef S
#define S(x) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));
KERNEL(k_cvt, S)
#undef S
#define S(x) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(x));
KERNEL(k_bfe, S)
#undef S
#define S(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "s"(m));
KERNEL(k_mul_sgpr, S)
#undef S
#define S(x) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x) : "s"(m));
KERNEL(k_sub_sgpr, S)
#undef S
#define S(x) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
KERNEL(k_div_fixup, S)
#undef S
#define S(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
KERNEL(k_fmac, S)
#undef S
#define S(x) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x));
KERNEL(k_cvt_ub, S)
#undef S
#define S(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
KERNEL(k_perm, S)
#undef S

#define S(x) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(x));
KERNEL(k_mul_inl, S)
#undef S
#define S(x) asm volatile("v_add_f32 %0, 0xc0400000, %0" : "+v"(x));
KERNEL(k_add_lit, S)
#undef S
#define S(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(m), "v"(n));
KERNEL(k_fma_sgpr, S)
#undef S
#define S(x) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(x) : "v"(n));
KERNEL(k_fma_inl, S)
#undef S
#define S(x) asm volatile("v_add_f32_e64 %0, %0, -%1" : "+v"(x) : "v"(m));
KERNEL(k_add_neg, S)
#undef S
#define S(x) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(x) : "v"(m));
KERNEL(k_add_abs, S)
#undef S
#define S(x) A2("v_sub_f32", x)
KERNEL(k_sub, S)
#undef S
#define S(x) A2("v_or_b32", x)
KERNEL(k_or, S)
#undef S
#define S(x) A2("v_sub_u32", x)
KERNEL(k_subu, S)
#undef S
#define S(x) asm volatile("v_not_b32 %0, %0" : "+v"(x));
KERNEL(k_not, S)
#undef S
#define S(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(m) : "vcc");
KERNEL(k_cmp_vcc, S)
#undef S
#define S(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(m) : "vcc");
KERNEL(k_cndmask_vcc, S)
#undef S
#define S(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "s"(m));
KERNEL(k_mov_sgpr, S)
#undef S
#define S(x) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "s"(m));
KERNEL(k_and_sgpr, S)
#undef S
#define S(x) asm volatile("v_and_b32 %0, 0x7fffff, %0" : "+v"(x));
KERNEL(k_and_lit, S)
#undef S
#define S(x) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
KERNEL(k_and_or, S)
#undef S
#define S(x) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(m));
KERNEL(k_lshl_or, S)
#undef S
#define S(x) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
KERNEL(k_xad, S)
#undef S
#define S(x) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(x) : "v"(m));
KERNEL(k_mul_legacy, S)
#undef S
#define S(x) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
KERNEL(k_min3, S)
#undef S
#define S(x) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
KERNEL(k_dpp, S)
#undef S
#define S(x) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(m));
KERNEL(k_add_dpp, S)
#undef S
#define S(x) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(si) : "v"(x));
KERNEL(k_rfl, S)
#undef S
#define S(x) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(m) : "vcc");
KERNEL(k_add_co, S)
#undef S
#define S(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
KERNEL(k_exp, S)
#undef S
#define S(x) asm volatile("v_fract_f32 %0, %0" : "+v"(x));
KERNEL(k_fract, S)
#undef S
#define S(x) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x));
KERNEL(k_cvt_u, S)
#undef S
#define S(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(dd) : "v"(x), "v"(m) : "vcc");
KERNEL(k_mad64, S)
#undef S

// mixed streams: do a 2-pass and a 4-pass instruction cost their sum when they alternate?
#define S(x) asm volatile("v_mul_f32 %0, %0, %1\n\tv_max_f32 %2, %2, %1" : "+v"(x), "+v"(b##x) : "v"(m));
#define KERNEL2(NAME, ASM) \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) { \
        float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float ba0 = a0, ba1 = a1, ba2 = a2, ba3 = a3, ba4 = a4, ba5 = a5, ba6 = a6, ba7 = a7; \
        const float m = 1.0000001f; \
        for (int i = 0; i < iters; i++) { ASM(a0) ASM(a1) ASM(a2) ASM(a3) ASM(a4) ASM(a5) ASM(a6) ASM(a7) } \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + ba0 + ba1 + ba2 + ba3 + ba4 + ba5 + ba6 + ba7; \
    }
KERNEL2(k_mix_mul_max, S)
#undef S
#define S(x) asm volatile("v_mul_f32 %0, %0, %1\n\tv_mul_f32 %2, %2, %1" : "+v"(x), "+v"(b##x) : "v"(m));
KERNEL2(k_mix_mul_mul, S)
#undef S
#define S(x) asm volatile("v_max_f32 %0, %0, %1\n\tv_max_f32 %2, %2, %1" : "+v"(x), "+v"(b##x) : "v"(m));
KERNEL2(k_mix_max_max, S)
#undef S
#define S(x) asm volatile("v_mul_f32 %0, %0, %1\n\tv_rcp_f32 %2, %2" : "+v"(x), "+v"(b##x) : "v"(m));
KERNEL2(k_mix_mul_rcp, S)
#undef S

typedef void (*kern_t)(float*, int);
static float run(kern_t f, float* d, int blocks, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, iters);
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    const int blocks = 256 * 8, iters = 10000; float* d; (void)hipMalloc(&d, blocks * 256 * 4);
    const double insts_per_simd = (double)blocks * 4 * iters * 16 / 1024.0;
    struct E { const char* name; kern_t f; };
    std::vector<E> es = { {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_mul_f32 (sgpr operand)", k_mul_sgpr}, {"v_sub_f32 (sgpr operand)", k_sub_sgpr},
        {"v_max_f32", k_max}, {"v_min_f32", k_min}, {"v_max3_f32", k_max3}, {"v_med3_f32", k_med3}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32 -> sgpr", k_cmp},
        {"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_lshlrev_b32", k_lshl}, {"v_alignbit_b32", k_alignbit}, {"v_add_u32", k_addu}, {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3},
        {"v_mov_b32", k_mov}, {"v_bfe_u32", k_bfe}, {"v_perm_b32", k_perm}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24},
        {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_cvt_f32_u32", k_cvt}, {"v_cvt_f32_ubyte0", k_cvt_ub}, {"v_div_fixup_f32", k_div_fixup},
        {"v_mul_f32 (inline constant)", k_mul_inl}, {"v_add_f32 (32-bit literal)", k_add_lit}, {"v_fma_f32 (sgpr operand)", k_fma_sgpr}, {"v_fma_f32 (inline constant)", k_fma_inl},
        {"v_add_f32 (neg modifier)", k_add_neg}, {"v_add_f32 (abs modifier)", k_add_abs}, {"v_sub_f32", k_sub}, {"v_or_b32", k_or}, {"v_sub_u32", k_subu}, {"v_not_b32", k_not},
        {"v_cmp_lt_f32 -> vcc", k_cmp_vcc}, {"v_cndmask_b32 (vcc)", k_cndmask_vcc}, {"v_mov_b32 (from sgpr)", k_mov_sgpr}, {"v_and_b32 (sgpr operand)", k_and_sgpr}, {"v_and_b32 (literal)", k_and_lit},
        {"v_and_or_b32", k_and_or}, {"v_lshl_or_b32", k_lshl_or}, {"v_xad_u32", k_xad}, {"v_mul_legacy_f32", k_mul_legacy}, {"v_min3_f32", k_min3},
        {"v_mov_b32_dpp", k_dpp}, {"v_add_f32_dpp", k_add_dpp}, {"v_readfirstlane_b32", k_rfl}, {"v_add_co_u32", k_add_co}, {"v_exp_f32", k_exp}, {"v_fract_f32", k_fract}, {"v_cvt_u32_f32", k_cvt_u}, {"v_mad_u64_u32", k_mad64},
        {"pairs: v_mul + v_max (per pair)", k_mix_mul_max}, {"pairs: v_mul + v_mul (per pair)", k_mix_mul_mul}, {"pairs: v_max + v_max (per pair)", k_mix_max_max}, {"pairs: v_mul + v_rcp (per pair)", k_mix_mul_rcp} };
    const float base = run(k_mul, d, blocks, iters);
    for (auto& e : es) { const float ms = run(e.f, d, blocks, iters); printf("%-28s %7.3f ms  %6.3f ns per wave instruction per SIMD  %5.2f x v_mul_f32\n", e.name, ms, ms * 1e6 / insts_per_simd, ms / base); }
    return 0;
}
