// valu_cost.hip -- issue cost (cycles per wave64 instruction per SIMD) of the
// VALU instructions the ensemble kernels are made of, measured on the GPU
// with s_memtime.  Evidence for the per-operation cycle budgets in DESIGN.md /
// profiles/README.md: on gfx950 a wave64 fp64 FMA/ADD/MUL occupies the SIMD
// for 4 cycles, 32-bit operations for 2, and several fp64 "helper" opcodes
// (conversions, ldexp, rndne, the division and root seeds) for more.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_cost valu_cost.hip && ./valu_cost
//
// For every instruction two numbers are printed:
//   thr  8 independent accumulators, W waves on each SIMD of ONE CU:
//        cycles the slowest wave needed / (W * instructions per wave)
//        = sustained issue cost per instruction per SIMD;
//   dep  one wave per SIMD, every instruction depending on the previous one
//        = issue-to-use latency.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#define REP 4
#define CHAINS 8

// One asm statement per group of 8 instructions (separate statements make
// hipcc put an s_nop between them).  Operands: %0..%7 accumulators (VGPR
// pairs, in/out), %8 int VGPR (in/out), %9 SGPR pair (in/out), %10 / %11
// double VGPR pairs, %12 double SGPR pair.
#define UB_OPERANDS(A0, A1, A2, A3, A4, A5, A6, A7)                          \
    : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6),  \
      "+v"(A7), "+v"(iv), "+s"(sm) : "v"(b), "v"(c), "s"(sc) : "vcc", "s2"

#define UB_KERNEL(NAME, THR, DEP)                                            \
    __global__ void NAME##_thr(unsigned long long *out, double b, double c,  \
                               int iters, unsigned long long mask)           \
    {                                                                        \
        double a[CHAINS];                                                    \
        for (int k = 0; k < CHAINS; ++k) a[k] = 1.0 + 0.001 * (threadIdx.x + k); \
        int iv = threadIdx.x & 3;                                            \
        unsigned long long sm = mask;                                        \
        const double sc = c;                                                 \
        asm volatile("s_mov_b64 exec, %0" : : "s"(mask));                    \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();          \
        for (int it = 0; it < iters; ++it) {                                 \
            _Pragma("unroll") for (int r = 0; r < REP; ++r)                  \
                asm volatile(THR UB_OPERANDS(a[0], a[1], a[2], a[3], a[4],   \
                                             a[5], a[6], a[7]));             \
        }                                                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();          \
        asm volatile("s_mov_b64 exec, -1");                                  \
        double s = 0;                                                        \
        for (int k = 0; k < CHAINS; ++k) s += a[k];                          \
        if (s == 12345.678 && iv == 77 && sm == 3) out[4096] = 1;            \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;        \
    }                                                                        \
    __global__ void NAME##_dep(unsigned long long *out, double b, double c,  \
                               int iters, unsigned long long mask)           \
    {                                                                        \
        double a[CHAINS];                                                    \
        for (int k = 0; k < CHAINS; ++k) a[k] = 1.0 + 0.001 * (threadIdx.x + k); \
        int iv = threadIdx.x & 3;                                            \
        unsigned long long sm = mask;                                        \
        const double sc = c;                                                 \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();          \
        for (int it = 0; it < iters; ++it) {                                 \
            _Pragma("unroll") for (int r = 0; r < REP; ++r)                  \
                asm volatile(DEP UB_OPERANDS(a[0], a[1], a[2], a[3], a[4],   \
                                             a[5], a[6], a[7]));             \
        }                                                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();          \
        double s = 0;                                                        \
        for (int k = 0; k < CHAINS; ++k) s += a[k];                          \
        if (s == 12345.678 && iv == 77 && sm == 3) out[4096] = 1;            \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;        \
    }

UB_KERNEL(fma_f64,
    "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %1, %1, %10, %11\n" "v_fma_f64 %2, %2, %10, %11\n" "v_fma_f64 %3, %3, %10, %11\n" "v_fma_f64 %4, %4, %10, %11\n" "v_fma_f64 %5, %5, %10, %11\n" "v_fma_f64 %6, %6, %10, %11\n" "v_fma_f64 %7, %7, %10, %11\n",
    "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n")
UB_KERNEL(fma_f64_sgpr,
    "v_fma_f64 %0, %0, %10, %12\n" "v_fma_f64 %1, %1, %10, %12\n" "v_fma_f64 %2, %2, %10, %12\n" "v_fma_f64 %3, %3, %10, %12\n" "v_fma_f64 %4, %4, %10, %12\n" "v_fma_f64 %5, %5, %10, %12\n" "v_fma_f64 %6, %6, %10, %12\n" "v_fma_f64 %7, %7, %10, %12\n",
    "v_fma_f64 %0, %0, %10, %12\n" "v_fma_f64 %0, %0, %10, %12\n" "v_fma_f64 %0, %0, %10, %12\n" "v_fma_f64 %0, %0, %10, %12\n" "v_fma_f64 %0, %0, %10, %12\n" "v_fma_f64 %0, %0, %10, %12\n" "v_fma_f64 %0, %0, %10, %12\n" "v_fma_f64 %0, %0, %10, %12\n")
UB_KERNEL(add_f64,
    "v_add_f64 %0, %0, %10\n" "v_add_f64 %1, %1, %10\n" "v_add_f64 %2, %2, %10\n" "v_add_f64 %3, %3, %10\n" "v_add_f64 %4, %4, %10\n" "v_add_f64 %5, %5, %10\n" "v_add_f64 %6, %6, %10\n" "v_add_f64 %7, %7, %10\n",
    "v_add_f64 %0, %0, %10\n" "v_add_f64 %0, %0, %10\n" "v_add_f64 %0, %0, %10\n" "v_add_f64 %0, %0, %10\n" "v_add_f64 %0, %0, %10\n" "v_add_f64 %0, %0, %10\n" "v_add_f64 %0, %0, %10\n" "v_add_f64 %0, %0, %10\n")
UB_KERNEL(mul_f64,
    "v_mul_f64 %0, %0, %10\n" "v_mul_f64 %1, %1, %10\n" "v_mul_f64 %2, %2, %10\n" "v_mul_f64 %3, %3, %10\n" "v_mul_f64 %4, %4, %10\n" "v_mul_f64 %5, %5, %10\n" "v_mul_f64 %6, %6, %10\n" "v_mul_f64 %7, %7, %10\n",
    "v_mul_f64 %0, %0, %10\n" "v_mul_f64 %0, %0, %10\n" "v_mul_f64 %0, %0, %10\n" "v_mul_f64 %0, %0, %10\n" "v_mul_f64 %0, %0, %10\n" "v_mul_f64 %0, %0, %10\n" "v_mul_f64 %0, %0, %10\n" "v_mul_f64 %0, %0, %10\n")
UB_KERNEL(max_f64,
    "v_max_f64 %0, %0, %10\n" "v_max_f64 %1, %1, %10\n" "v_max_f64 %2, %2, %10\n" "v_max_f64 %3, %3, %10\n" "v_max_f64 %4, %4, %10\n" "v_max_f64 %5, %5, %10\n" "v_max_f64 %6, %6, %10\n" "v_max_f64 %7, %7, %10\n",
    "v_max_f64 %0, %0, %10\n" "v_max_f64 %0, %0, %10\n" "v_max_f64 %0, %0, %10\n" "v_max_f64 %0, %0, %10\n" "v_max_f64 %0, %0, %10\n" "v_max_f64 %0, %0, %10\n" "v_max_f64 %0, %0, %10\n" "v_max_f64 %0, %0, %10\n")
UB_KERNEL(min_f64,
    "v_min_f64 %0, %0, %10\n" "v_min_f64 %1, %1, %10\n" "v_min_f64 %2, %2, %10\n" "v_min_f64 %3, %3, %10\n" "v_min_f64 %4, %4, %10\n" "v_min_f64 %5, %5, %10\n" "v_min_f64 %6, %6, %10\n" "v_min_f64 %7, %7, %10\n",
    "v_min_f64 %0, %0, %10\n" "v_min_f64 %0, %0, %10\n" "v_min_f64 %0, %0, %10\n" "v_min_f64 %0, %0, %10\n" "v_min_f64 %0, %0, %10\n" "v_min_f64 %0, %0, %10\n" "v_min_f64 %0, %0, %10\n" "v_min_f64 %0, %0, %10\n")
UB_KERNEL(mov_b64,
    "v_mov_b64 %0, %10\n" "v_mov_b64 %1, %10\n" "v_mov_b64 %2, %10\n" "v_mov_b64 %3, %10\n" "v_mov_b64 %4, %10\n" "v_mov_b64 %5, %10\n" "v_mov_b64 %6, %10\n" "v_mov_b64 %7, %10\n",
    "v_mov_b64 %0, %10\n" "v_mov_b64 %0, %10\n" "v_mov_b64 %0, %10\n" "v_mov_b64 %0, %10\n" "v_mov_b64 %0, %10\n" "v_mov_b64 %0, %10\n" "v_mov_b64 %0, %10\n" "v_mov_b64 %0, %10\n")
UB_KERNEL(ldexp_f64,
    "v_ldexp_f64 %0, %0, %8\n" "v_ldexp_f64 %1, %1, %8\n" "v_ldexp_f64 %2, %2, %8\n" "v_ldexp_f64 %3, %3, %8\n" "v_ldexp_f64 %4, %4, %8\n" "v_ldexp_f64 %5, %5, %8\n" "v_ldexp_f64 %6, %6, %8\n" "v_ldexp_f64 %7, %7, %8\n",
    "v_ldexp_f64 %0, %0, %8\n" "v_ldexp_f64 %0, %0, %8\n" "v_ldexp_f64 %0, %0, %8\n" "v_ldexp_f64 %0, %0, %8\n" "v_ldexp_f64 %0, %0, %8\n" "v_ldexp_f64 %0, %0, %8\n" "v_ldexp_f64 %0, %0, %8\n" "v_ldexp_f64 %0, %0, %8\n")
UB_KERNEL(rndne_f64,
    "v_rndne_f64 %0, %0\n" "v_rndne_f64 %1, %1\n" "v_rndne_f64 %2, %2\n" "v_rndne_f64 %3, %3\n" "v_rndne_f64 %4, %4\n" "v_rndne_f64 %5, %5\n" "v_rndne_f64 %6, %6\n" "v_rndne_f64 %7, %7\n",
    "v_rndne_f64 %0, %0\n" "v_rndne_f64 %0, %0\n" "v_rndne_f64 %0, %0\n" "v_rndne_f64 %0, %0\n" "v_rndne_f64 %0, %0\n" "v_rndne_f64 %0, %0\n" "v_rndne_f64 %0, %0\n" "v_rndne_f64 %0, %0\n")
UB_KERNEL(fract_f64,
    "v_fract_f64 %0, %0\n" "v_fract_f64 %1, %1\n" "v_fract_f64 %2, %2\n" "v_fract_f64 %3, %3\n" "v_fract_f64 %4, %4\n" "v_fract_f64 %5, %5\n" "v_fract_f64 %6, %6\n" "v_fract_f64 %7, %7\n",
    "v_fract_f64 %0, %0\n" "v_fract_f64 %0, %0\n" "v_fract_f64 %0, %0\n" "v_fract_f64 %0, %0\n" "v_fract_f64 %0, %0\n" "v_fract_f64 %0, %0\n" "v_fract_f64 %0, %0\n" "v_fract_f64 %0, %0\n")
UB_KERNEL(cvt_f64_i32,
    "v_cvt_f64_i32 %0, %8\n" "v_cvt_f64_i32 %1, %8\n" "v_cvt_f64_i32 %2, %8\n" "v_cvt_f64_i32 %3, %8\n" "v_cvt_f64_i32 %4, %8\n" "v_cvt_f64_i32 %5, %8\n" "v_cvt_f64_i32 %6, %8\n" "v_cvt_f64_i32 %7, %8\n",
    "v_cvt_f64_i32 %0, %8\n" "v_cvt_f64_i32 %0, %8\n" "v_cvt_f64_i32 %0, %8\n" "v_cvt_f64_i32 %0, %8\n" "v_cvt_f64_i32 %0, %8\n" "v_cvt_f64_i32 %0, %8\n" "v_cvt_f64_i32 %0, %8\n" "v_cvt_f64_i32 %0, %8\n")
UB_KERNEL(cvt_i32_f64,
    "v_cvt_i32_f64 %8, %0\n" "v_cvt_i32_f64 %8, %1\n" "v_cvt_i32_f64 %8, %2\n" "v_cvt_i32_f64 %8, %3\n" "v_cvt_i32_f64 %8, %4\n" "v_cvt_i32_f64 %8, %5\n" "v_cvt_i32_f64 %8, %6\n" "v_cvt_i32_f64 %8, %7\n",
    "v_cvt_i32_f64 %8, %0\n" "v_cvt_i32_f64 %8, %0\n" "v_cvt_i32_f64 %8, %0\n" "v_cvt_i32_f64 %8, %0\n" "v_cvt_i32_f64 %8, %0\n" "v_cvt_i32_f64 %8, %0\n" "v_cvt_i32_f64 %8, %0\n" "v_cvt_i32_f64 %8, %0\n")
UB_KERNEL(frexp_mant_f64,
    "v_frexp_mant_f64 %0, %0\n" "v_frexp_mant_f64 %1, %1\n" "v_frexp_mant_f64 %2, %2\n" "v_frexp_mant_f64 %3, %3\n" "v_frexp_mant_f64 %4, %4\n" "v_frexp_mant_f64 %5, %5\n" "v_frexp_mant_f64 %6, %6\n" "v_frexp_mant_f64 %7, %7\n",
    "v_frexp_mant_f64 %0, %0\n" "v_frexp_mant_f64 %0, %0\n" "v_frexp_mant_f64 %0, %0\n" "v_frexp_mant_f64 %0, %0\n" "v_frexp_mant_f64 %0, %0\n" "v_frexp_mant_f64 %0, %0\n" "v_frexp_mant_f64 %0, %0\n" "v_frexp_mant_f64 %0, %0\n")
UB_KERNEL(frexp_exp_f64,
    "v_frexp_exp_i32_f64 %8, %0\n" "v_frexp_exp_i32_f64 %8, %1\n" "v_frexp_exp_i32_f64 %8, %2\n" "v_frexp_exp_i32_f64 %8, %3\n" "v_frexp_exp_i32_f64 %8, %4\n" "v_frexp_exp_i32_f64 %8, %5\n" "v_frexp_exp_i32_f64 %8, %6\n" "v_frexp_exp_i32_f64 %8, %7\n",
    "v_frexp_exp_i32_f64 %8, %0\n" "v_frexp_exp_i32_f64 %8, %0\n" "v_frexp_exp_i32_f64 %8, %0\n" "v_frexp_exp_i32_f64 %8, %0\n" "v_frexp_exp_i32_f64 %8, %0\n" "v_frexp_exp_i32_f64 %8, %0\n" "v_frexp_exp_i32_f64 %8, %0\n" "v_frexp_exp_i32_f64 %8, %0\n")
UB_KERNEL(rcp_f64,
    "v_rcp_f64 %0, %0\n" "v_rcp_f64 %1, %1\n" "v_rcp_f64 %2, %2\n" "v_rcp_f64 %3, %3\n" "v_rcp_f64 %4, %4\n" "v_rcp_f64 %5, %5\n" "v_rcp_f64 %6, %6\n" "v_rcp_f64 %7, %7\n",
    "v_rcp_f64 %0, %0\n" "v_rcp_f64 %0, %0\n" "v_rcp_f64 %0, %0\n" "v_rcp_f64 %0, %0\n" "v_rcp_f64 %0, %0\n" "v_rcp_f64 %0, %0\n" "v_rcp_f64 %0, %0\n" "v_rcp_f64 %0, %0\n")
UB_KERNEL(rsq_f64,
    "v_rsq_f64 %0, %0\n" "v_rsq_f64 %1, %1\n" "v_rsq_f64 %2, %2\n" "v_rsq_f64 %3, %3\n" "v_rsq_f64 %4, %4\n" "v_rsq_f64 %5, %5\n" "v_rsq_f64 %6, %6\n" "v_rsq_f64 %7, %7\n",
    "v_rsq_f64 %0, %0\n" "v_rsq_f64 %0, %0\n" "v_rsq_f64 %0, %0\n" "v_rsq_f64 %0, %0\n" "v_rsq_f64 %0, %0\n" "v_rsq_f64 %0, %0\n" "v_rsq_f64 %0, %0\n" "v_rsq_f64 %0, %0\n")
// one quarter-rate instruction among seven full-rate ones (per instruction of
// the group; 8 x 4.3 would be 4.3, 16 + 7 x 4.3 would be 5.8)
UB_KERNEL(mix_rsq_7fma,
    "v_rsq_f64 %0, %0\n" "v_fma_f64 %1, %1, %10, %11\n" "v_fma_f64 %2, %2, %10, %11\n" "v_fma_f64 %3, %3, %10, %11\n" "v_fma_f64 %4, %4, %10, %11\n" "v_fma_f64 %5, %5, %10, %11\n" "v_fma_f64 %6, %6, %10, %11\n" "v_fma_f64 %7, %7, %10, %11\n",
    "v_rsq_f64 %0, %0\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n")
// ... and among three
UB_KERNEL(mix_rsq_3fma,
    "v_rsq_f64 %0, %0\n" "v_fma_f64 %1, %1, %10, %11\n" "v_fma_f64 %2, %2, %10, %11\n" "v_fma_f64 %3, %3, %10, %11\n" "v_rsq_f64 %4, %4\n" "v_fma_f64 %5, %5, %10, %11\n" "v_fma_f64 %6, %6, %10, %11\n" "v_fma_f64 %7, %7, %10, %11\n",
    "v_rsq_f64 %0, %0\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_rsq_f64 %0, %0\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n" "v_fma_f64 %0, %0, %10, %11\n")
UB_KERNEL(sqrt_f64,
    "v_sqrt_f64 %0, %0\n" "v_sqrt_f64 %1, %1\n" "v_sqrt_f64 %2, %2\n" "v_sqrt_f64 %3, %3\n" "v_sqrt_f64 %4, %4\n" "v_sqrt_f64 %5, %5\n" "v_sqrt_f64 %6, %6\n" "v_sqrt_f64 %7, %7\n",
    "v_sqrt_f64 %0, %0\n" "v_sqrt_f64 %0, %0\n" "v_sqrt_f64 %0, %0\n" "v_sqrt_f64 %0, %0\n" "v_sqrt_f64 %0, %0\n" "v_sqrt_f64 %0, %0\n" "v_sqrt_f64 %0, %0\n" "v_sqrt_f64 %0, %0\n")
UB_KERNEL(div_scale_f64,
    "v_div_scale_f64 %0, vcc, %0, %10, %11\n" "v_div_scale_f64 %1, vcc, %1, %10, %11\n" "v_div_scale_f64 %2, vcc, %2, %10, %11\n" "v_div_scale_f64 %3, vcc, %3, %10, %11\n" "v_div_scale_f64 %4, vcc, %4, %10, %11\n" "v_div_scale_f64 %5, vcc, %5, %10, %11\n" "v_div_scale_f64 %6, vcc, %6, %10, %11\n" "v_div_scale_f64 %7, vcc, %7, %10, %11\n",
    "v_div_scale_f64 %0, vcc, %0, %10, %11\n" "v_div_scale_f64 %0, vcc, %0, %10, %11\n" "v_div_scale_f64 %0, vcc, %0, %10, %11\n" "v_div_scale_f64 %0, vcc, %0, %10, %11\n" "v_div_scale_f64 %0, vcc, %0, %10, %11\n" "v_div_scale_f64 %0, vcc, %0, %10, %11\n" "v_div_scale_f64 %0, vcc, %0, %10, %11\n" "v_div_scale_f64 %0, vcc, %0, %10, %11\n")
UB_KERNEL(div_fmas_f64,
    "v_div_fmas_f64 %0, %0, %10, %11\n" "v_div_fmas_f64 %1, %1, %10, %11\n" "v_div_fmas_f64 %2, %2, %10, %11\n" "v_div_fmas_f64 %3, %3, %10, %11\n" "v_div_fmas_f64 %4, %4, %10, %11\n" "v_div_fmas_f64 %5, %5, %10, %11\n" "v_div_fmas_f64 %6, %6, %10, %11\n" "v_div_fmas_f64 %7, %7, %10, %11\n",
    "v_div_fmas_f64 %0, %0, %10, %11\n" "v_div_fmas_f64 %0, %0, %10, %11\n" "v_div_fmas_f64 %0, %0, %10, %11\n" "v_div_fmas_f64 %0, %0, %10, %11\n" "v_div_fmas_f64 %0, %0, %10, %11\n" "v_div_fmas_f64 %0, %0, %10, %11\n" "v_div_fmas_f64 %0, %0, %10, %11\n" "v_div_fmas_f64 %0, %0, %10, %11\n")
UB_KERNEL(div_fixup_f64,
    "v_div_fixup_f64 %0, %0, %10, %11\n" "v_div_fixup_f64 %1, %1, %10, %11\n" "v_div_fixup_f64 %2, %2, %10, %11\n" "v_div_fixup_f64 %3, %3, %10, %11\n" "v_div_fixup_f64 %4, %4, %10, %11\n" "v_div_fixup_f64 %5, %5, %10, %11\n" "v_div_fixup_f64 %6, %6, %10, %11\n" "v_div_fixup_f64 %7, %7, %10, %11\n",
    "v_div_fixup_f64 %0, %0, %10, %11\n" "v_div_fixup_f64 %0, %0, %10, %11\n" "v_div_fixup_f64 %0, %0, %10, %11\n" "v_div_fixup_f64 %0, %0, %10, %11\n" "v_div_fixup_f64 %0, %0, %10, %11\n" "v_div_fixup_f64 %0, %0, %10, %11\n" "v_div_fixup_f64 %0, %0, %10, %11\n" "v_div_fixup_f64 %0, %0, %10, %11\n")
UB_KERNEL(trig_preop_f64,
    "v_trig_preop_f64 %0, %0, %8\n" "v_trig_preop_f64 %1, %1, %8\n" "v_trig_preop_f64 %2, %2, %8\n" "v_trig_preop_f64 %3, %3, %8\n" "v_trig_preop_f64 %4, %4, %8\n" "v_trig_preop_f64 %5, %5, %8\n" "v_trig_preop_f64 %6, %6, %8\n" "v_trig_preop_f64 %7, %7, %8\n",
    "v_trig_preop_f64 %0, %0, %8\n" "v_trig_preop_f64 %0, %0, %8\n" "v_trig_preop_f64 %0, %0, %8\n" "v_trig_preop_f64 %0, %0, %8\n" "v_trig_preop_f64 %0, %0, %8\n" "v_trig_preop_f64 %0, %0, %8\n" "v_trig_preop_f64 %0, %0, %8\n" "v_trig_preop_f64 %0, %0, %8\n")
UB_KERNEL(cmp_lt_f64_vcc,
    "v_cmp_lt_f64 vcc, %0, %10\n" "v_cmp_lt_f64 vcc, %1, %10\n" "v_cmp_lt_f64 vcc, %2, %10\n" "v_cmp_lt_f64 vcc, %3, %10\n" "v_cmp_lt_f64 vcc, %4, %10\n" "v_cmp_lt_f64 vcc, %5, %10\n" "v_cmp_lt_f64 vcc, %6, %10\n" "v_cmp_lt_f64 vcc, %7, %10\n",
    "v_cmp_lt_f64 vcc, %0, %10\n" "v_cmp_lt_f64 vcc, %0, %10\n" "v_cmp_lt_f64 vcc, %0, %10\n" "v_cmp_lt_f64 vcc, %0, %10\n" "v_cmp_lt_f64 vcc, %0, %10\n" "v_cmp_lt_f64 vcc, %0, %10\n" "v_cmp_lt_f64 vcc, %0, %10\n" "v_cmp_lt_f64 vcc, %0, %10\n")
UB_KERNEL(cmp_lt_f64_sgpr,
    "v_cmp_lt_f64 %9, %0, %10\n" "v_cmp_lt_f64 %9, %1, %10\n" "v_cmp_lt_f64 %9, %2, %10\n" "v_cmp_lt_f64 %9, %3, %10\n" "v_cmp_lt_f64 %9, %4, %10\n" "v_cmp_lt_f64 %9, %5, %10\n" "v_cmp_lt_f64 %9, %6, %10\n" "v_cmp_lt_f64 %9, %7, %10\n",
    "v_cmp_lt_f64 %9, %0, %10\n" "v_cmp_lt_f64 %9, %0, %10\n" "v_cmp_lt_f64 %9, %0, %10\n" "v_cmp_lt_f64 %9, %0, %10\n" "v_cmp_lt_f64 %9, %0, %10\n" "v_cmp_lt_f64 %9, %0, %10\n" "v_cmp_lt_f64 %9, %0, %10\n" "v_cmp_lt_f64 %9, %0, %10\n")
UB_KERNEL(cmp_class_f64_vcc,
    "v_cmp_class_f64 vcc, %0, %8\n" "v_cmp_class_f64 vcc, %1, %8\n" "v_cmp_class_f64 vcc, %2, %8\n" "v_cmp_class_f64 vcc, %3, %8\n" "v_cmp_class_f64 vcc, %4, %8\n" "v_cmp_class_f64 vcc, %5, %8\n" "v_cmp_class_f64 vcc, %6, %8\n" "v_cmp_class_f64 vcc, %7, %8\n",
    "v_cmp_class_f64 vcc, %0, %8\n" "v_cmp_class_f64 vcc, %0, %8\n" "v_cmp_class_f64 vcc, %0, %8\n" "v_cmp_class_f64 vcc, %0, %8\n" "v_cmp_class_f64 vcc, %0, %8\n" "v_cmp_class_f64 vcc, %0, %8\n" "v_cmp_class_f64 vcc, %0, %8\n" "v_cmp_class_f64 vcc, %0, %8\n")
UB_KERNEL(cmp_class_f64_sgpr,
    "v_cmp_class_f64 %9, %0, %8\n" "v_cmp_class_f64 %9, %1, %8\n" "v_cmp_class_f64 %9, %2, %8\n" "v_cmp_class_f64 %9, %3, %8\n" "v_cmp_class_f64 %9, %4, %8\n" "v_cmp_class_f64 %9, %5, %8\n" "v_cmp_class_f64 %9, %6, %8\n" "v_cmp_class_f64 %9, %7, %8\n",
    "v_cmp_class_f64 %9, %0, %8\n" "v_cmp_class_f64 %9, %0, %8\n" "v_cmp_class_f64 %9, %0, %8\n" "v_cmp_class_f64 %9, %0, %8\n" "v_cmp_class_f64 %9, %0, %8\n" "v_cmp_class_f64 %9, %0, %8\n" "v_cmp_class_f64 %9, %0, %8\n" "v_cmp_class_f64 %9, %0, %8\n")
UB_KERNEL(cndmask_b32_vcc,
    "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n",
    "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n" "v_cndmask_b32 %8, %8, %8, vcc\n")
UB_KERNEL(cndmask_b32_sgpr,
    "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n",
    "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n" "v_cndmask_b32 %8, %8, %8, %9\n")
UB_KERNEL(mov_b32,
    "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n",
    "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n" "v_mov_b32 %8, %8\n")
UB_KERNEL(add_u32,
    "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n",
    "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n" "v_add_u32 %8, %8, %8\n")
UB_KERNEL(and_b32,
    "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n",
    "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n" "v_and_b32 %8, %8, %8\n")
UB_KERNEL(lshl_add_u32,
    "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n",
    "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n" "v_lshl_add_u32 %8, %8, 1, %8\n")
UB_KERNEL(ashrrev_i32,
    "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n",
    "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n" "v_ashrrev_i32 %8, 1, %8\n")
UB_KERNEL(lshl_add_u64,
    "v_lshl_add_u64 %0, %0, 0, %10\n" "v_lshl_add_u64 %1, %1, 0, %10\n" "v_lshl_add_u64 %2, %2, 0, %10\n" "v_lshl_add_u64 %3, %3, 0, %10\n" "v_lshl_add_u64 %4, %4, 0, %10\n" "v_lshl_add_u64 %5, %5, 0, %10\n" "v_lshl_add_u64 %6, %6, 0, %10\n" "v_lshl_add_u64 %7, %7, 0, %10\n",
    "v_lshl_add_u64 %0, %0, 0, %10\n" "v_lshl_add_u64 %0, %0, 0, %10\n" "v_lshl_add_u64 %0, %0, 0, %10\n" "v_lshl_add_u64 %0, %0, 0, %10\n" "v_lshl_add_u64 %0, %0, 0, %10\n" "v_lshl_add_u64 %0, %0, 0, %10\n" "v_lshl_add_u64 %0, %0, 0, %10\n" "v_lshl_add_u64 %0, %0, 0, %10\n")
UB_KERNEL(mad_u64_u32,
    "v_mad_u64_u32 %0, vcc, %8, %8, %0\n" "v_mad_u64_u32 %1, vcc, %8, %8, %1\n" "v_mad_u64_u32 %2, vcc, %8, %8, %2\n" "v_mad_u64_u32 %3, vcc, %8, %8, %3\n" "v_mad_u64_u32 %4, vcc, %8, %8, %4\n" "v_mad_u64_u32 %5, vcc, %8, %8, %5\n" "v_mad_u64_u32 %6, vcc, %8, %8, %6\n" "v_mad_u64_u32 %7, vcc, %8, %8, %7\n",
    "v_mad_u64_u32 %0, vcc, %8, %8, %0\n" "v_mad_u64_u32 %0, vcc, %8, %8, %0\n" "v_mad_u64_u32 %0, vcc, %8, %8, %0\n" "v_mad_u64_u32 %0, vcc, %8, %8, %0\n" "v_mad_u64_u32 %0, vcc, %8, %8, %0\n" "v_mad_u64_u32 %0, vcc, %8, %8, %0\n" "v_mad_u64_u32 %0, vcc, %8, %8, %0\n" "v_mad_u64_u32 %0, vcc, %8, %8, %0\n")
UB_KERNEL(fma_f32,
    "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n",
    "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n" "v_fma_f32 %8, %8, %8, %8\n")
UB_KERNEL(cvt_f32_f64,
    "v_cvt_f32_f64 %8, %0\n" "v_cvt_f32_f64 %8, %1\n" "v_cvt_f32_f64 %8, %2\n" "v_cvt_f32_f64 %8, %3\n" "v_cvt_f32_f64 %8, %4\n" "v_cvt_f32_f64 %8, %5\n" "v_cvt_f32_f64 %8, %6\n" "v_cvt_f32_f64 %8, %7\n",
    "v_cvt_f32_f64 %8, %0\n" "v_cvt_f32_f64 %8, %0\n" "v_cvt_f32_f64 %8, %0\n" "v_cvt_f32_f64 %8, %0\n" "v_cvt_f32_f64 %8, %0\n" "v_cvt_f32_f64 %8, %0\n" "v_cvt_f32_f64 %8, %0\n" "v_cvt_f32_f64 %8, %0\n")
UB_KERNEL(cvt_f64_f32,
    "v_cvt_f64_f32 %0, %8\n" "v_cvt_f64_f32 %1, %8\n" "v_cvt_f64_f32 %2, %8\n" "v_cvt_f64_f32 %3, %8\n" "v_cvt_f64_f32 %4, %8\n" "v_cvt_f64_f32 %5, %8\n" "v_cvt_f64_f32 %6, %8\n" "v_cvt_f64_f32 %7, %8\n",
    "v_cvt_f64_f32 %0, %8\n" "v_cvt_f64_f32 %0, %8\n" "v_cvt_f64_f32 %0, %8\n" "v_cvt_f64_f32 %0, %8\n" "v_cvt_f64_f32 %0, %8\n" "v_cvt_f64_f32 %0, %8\n" "v_cvt_f64_f32 %0, %8\n" "v_cvt_f64_f32 %0, %8\n")
UB_KERNEL(readlane_b32,
    "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n",
    "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n" "v_readlane_b32 s2, %8, 0\n")
UB_KERNEL(readfirstlane_b32,
    "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n",
    "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n" "v_readfirstlane_b32 s2, %8\n")
UB_KERNEL(writelane_b32,
    "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n",
    "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n" "v_writelane_b32 %8, s2, 0\n")


#define UB_OPERANDS_I(A0, A1, A2, A3, A4, A5, A6, A7)                        \
    : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6),  \
      "+v"(A7), "+v"(iv), "+s"(sm), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3),  \
      "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(b), "v"(c), "v"(d0)       \
    : "vcc", "s2"

// int accumulators x[0..7] (%0..%7), y[0..7] (%10..%17), %8 int, %9 SGPR pair,
// %18 / %19 / %20 double VGPR pairs
#define UB_KERNEL_I(NAME, THR, NPER)                                         \
    __global__ void NAME##_thr(unsigned long long *out, double b, double c,  \
                               int iters, unsigned long long mask)           \
    {                                                                        \
        int x[CHAINS];                                                       \
        for (int k = 0; k < CHAINS; ++k) x[k] = threadIdx.x + k;             \
        int y0 = 1, y1 = 2, y2 = 3, y3 = 4, y4 = 5, y5 = 6, y6 = 7, y7 = 8;  \
        int iv = threadIdx.x & 3;                                            \
        unsigned long long sm = mask;                                        \
        double d0 = 1.0 + 0.001 * threadIdx.x;                               \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();          \
        for (int it = 0; it < iters; ++it) {                                 \
            _Pragma("unroll") for (int r = 0; r < REP; ++r)                  \
                asm volatile(THR UB_OPERANDS_I(x[0], x[1], x[2], x[3], x[4], \
                                               x[5], x[6], x[7]));           \
        }                                                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();          \
        int s = y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;                       \
        for (int k = 0; k < CHAINS; ++k) s += x[k];                          \
        if (s == 123456789 && iv == 77 && sm == 3) out[4096] = 1;            \
        if ((threadIdx.x & 63) == 0)                                         \
            out[threadIdx.x >> 6] = (t1 - t0) / NPER;                        \
    }                                                                        \
    __global__ void NAME##_dep(unsigned long long *out, double b, double c,  \
                               int iters, unsigned long long mask)           \
    {                                                                        \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = 0;              \
    }

UB_KERNEL_I(sel_vop2_vcc,
    "v_cndmask_b32_e32 %0, %0, %8, vcc\n" "v_cndmask_b32_e32 %1, %1, %8, vcc\n" "v_cndmask_b32_e32 %2, %2, %8, vcc\n" "v_cndmask_b32_e32 %3, %3, %8, vcc\n" "v_cndmask_b32_e32 %4, %4, %8, vcc\n" "v_cndmask_b32_e32 %5, %5, %8, vcc\n" "v_cndmask_b32_e32 %6, %6, %8, vcc\n" "v_cndmask_b32_e32 %7, %7, %8, vcc\n", 1)
UB_KERNEL_I(sel_vop3_vcc,
    "v_cndmask_b32_e64 %0, %0, %8, vcc\n" "v_cndmask_b32_e64 %1, %1, %8, vcc\n" "v_cndmask_b32_e64 %2, %2, %8, vcc\n" "v_cndmask_b32_e64 %3, %3, %8, vcc\n" "v_cndmask_b32_e64 %4, %4, %8, vcc\n" "v_cndmask_b32_e64 %5, %5, %8, vcc\n" "v_cndmask_b32_e64 %6, %6, %8, vcc\n" "v_cndmask_b32_e64 %7, %7, %8, vcc\n", 1)
UB_KERNEL_I(sel_vop3_sgpr,
    "v_cndmask_b32_e64 %0, %0, %8, %9\n" "v_cndmask_b32_e64 %1, %1, %8, %9\n" "v_cndmask_b32_e64 %2, %2, %8, %9\n" "v_cndmask_b32_e64 %3, %3, %8, %9\n" "v_cndmask_b32_e64 %4, %4, %8, %9\n" "v_cndmask_b32_e64 %5, %5, %8, %9\n" "v_cndmask_b32_e64 %6, %6, %8, %9\n" "v_cndmask_b32_e64 %7, %7, %8, %9\n", 1)
UB_KERNEL_I(cmp_sel2_vcc,
    "v_cmp_lt_f64 vcc, %20, %18\nv_cndmask_b32_e32 %0, %0, %8, vcc\nv_cndmask_b32_e32 %10, %10, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_cndmask_b32_e32 %1, %1, %8, vcc\nv_cndmask_b32_e32 %11, %11, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_cndmask_b32_e32 %2, %2, %8, vcc\nv_cndmask_b32_e32 %12, %12, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_cndmask_b32_e32 %3, %3, %8, vcc\nv_cndmask_b32_e32 %13, %13, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_cndmask_b32_e32 %4, %4, %8, vcc\nv_cndmask_b32_e32 %14, %14, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_cndmask_b32_e32 %5, %5, %8, vcc\nv_cndmask_b32_e32 %15, %15, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_cndmask_b32_e32 %6, %6, %8, vcc\nv_cndmask_b32_e32 %16, %16, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_cndmask_b32_e32 %7, %7, %8, vcc\nv_cndmask_b32_e32 %17, %17, %8, vcc\n", 3)
UB_KERNEL_I(cmp_sel2_sgpr,
    "v_cmp_lt_f64 %9, %20, %18\nv_cndmask_b32_e64 %0, %0, %8, %9\nv_cndmask_b32_e64 %10, %10, %8, %9\n" "v_cmp_lt_f64 %9, %20, %18\nv_cndmask_b32_e64 %1, %1, %8, %9\nv_cndmask_b32_e64 %11, %11, %8, %9\n" "v_cmp_lt_f64 %9, %20, %18\nv_cndmask_b32_e64 %2, %2, %8, %9\nv_cndmask_b32_e64 %12, %12, %8, %9\n" "v_cmp_lt_f64 %9, %20, %18\nv_cndmask_b32_e64 %3, %3, %8, %9\nv_cndmask_b32_e64 %13, %13, %8, %9\n" "v_cmp_lt_f64 %9, %20, %18\nv_cndmask_b32_e64 %4, %4, %8, %9\nv_cndmask_b32_e64 %14, %14, %8, %9\n" "v_cmp_lt_f64 %9, %20, %18\nv_cndmask_b32_e64 %5, %5, %8, %9\nv_cndmask_b32_e64 %15, %15, %8, %9\n" "v_cmp_lt_f64 %9, %20, %18\nv_cndmask_b32_e64 %6, %6, %8, %9\nv_cndmask_b32_e64 %16, %16, %8, %9\n" "v_cmp_lt_f64 %9, %20, %18\nv_cndmask_b32_e64 %7, %7, %8, %9\nv_cndmask_b32_e64 %17, %17, %8, %9\n", 3)
// a compare, 1 independent VOP3 instructions, then the VOP2 selects on VCC:
// does the select's cost depend on how long ago VCC was written?
UB_KERNEL_I(cmp_gap1_sel2_vcc,
    "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %11, %11, 0, %8\nv_cndmask_b32_e32 %0, %0, %8, vcc\nv_cndmask_b32_e32 %10, %10, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %12, %12, 0, %8\nv_cndmask_b32_e32 %1, %1, %8, vcc\nv_cndmask_b32_e32 %11, %11, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %13, %13, 0, %8\nv_cndmask_b32_e32 %2, %2, %8, vcc\nv_cndmask_b32_e32 %12, %12, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %14, %14, 0, %8\nv_cndmask_b32_e32 %3, %3, %8, vcc\nv_cndmask_b32_e32 %13, %13, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %15, %15, 0, %8\nv_cndmask_b32_e32 %4, %4, %8, vcc\nv_cndmask_b32_e32 %14, %14, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %16, %16, 0, %8\nv_cndmask_b32_e32 %5, %5, %8, vcc\nv_cndmask_b32_e32 %15, %15, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %17, %17, 0, %8\nv_cndmask_b32_e32 %6, %6, %8, vcc\nv_cndmask_b32_e32 %16, %16, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %10, %10, 0, %8\nv_cndmask_b32_e32 %7, %7, %8, vcc\nv_cndmask_b32_e32 %17, %17, %8, vcc\n", 4)
// a compare, 2 independent VOP3 instructions, then the VOP2 selects on VCC:
// does the select's cost depend on how long ago VCC was written?
UB_KERNEL_I(cmp_gap2_sel2_vcc,
    "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %11, %11, 0, %8\nv_lshl_add_u32 %12, %12, 0, %8\nv_cndmask_b32_e32 %0, %0, %8, vcc\nv_cndmask_b32_e32 %10, %10, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %12, %12, 0, %8\nv_lshl_add_u32 %13, %13, 0, %8\nv_cndmask_b32_e32 %1, %1, %8, vcc\nv_cndmask_b32_e32 %11, %11, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %13, %13, 0, %8\nv_lshl_add_u32 %14, %14, 0, %8\nv_cndmask_b32_e32 %2, %2, %8, vcc\nv_cndmask_b32_e32 %12, %12, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %14, %14, 0, %8\nv_lshl_add_u32 %15, %15, 0, %8\nv_cndmask_b32_e32 %3, %3, %8, vcc\nv_cndmask_b32_e32 %13, %13, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %15, %15, 0, %8\nv_lshl_add_u32 %16, %16, 0, %8\nv_cndmask_b32_e32 %4, %4, %8, vcc\nv_cndmask_b32_e32 %14, %14, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %16, %16, 0, %8\nv_lshl_add_u32 %17, %17, 0, %8\nv_cndmask_b32_e32 %5, %5, %8, vcc\nv_cndmask_b32_e32 %15, %15, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %17, %17, 0, %8\nv_lshl_add_u32 %10, %10, 0, %8\nv_cndmask_b32_e32 %6, %6, %8, vcc\nv_cndmask_b32_e32 %16, %16, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %10, %10, 0, %8\nv_lshl_add_u32 %11, %11, 0, %8\nv_cndmask_b32_e32 %7, %7, %8, vcc\nv_cndmask_b32_e32 %17, %17, %8, vcc\n", 5)
// a compare, 4 independent VOP3 instructions, then the VOP2 selects on VCC:
// does the select's cost depend on how long ago VCC was written?
UB_KERNEL_I(cmp_gap4_sel2_vcc,
    "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %11, %11, 0, %8\nv_lshl_add_u32 %12, %12, 0, %8\nv_lshl_add_u32 %13, %13, 0, %8\nv_lshl_add_u32 %14, %14, 0, %8\nv_cndmask_b32_e32 %0, %0, %8, vcc\nv_cndmask_b32_e32 %10, %10, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %12, %12, 0, %8\nv_lshl_add_u32 %13, %13, 0, %8\nv_lshl_add_u32 %14, %14, 0, %8\nv_lshl_add_u32 %15, %15, 0, %8\nv_cndmask_b32_e32 %1, %1, %8, vcc\nv_cndmask_b32_e32 %11, %11, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %13, %13, 0, %8\nv_lshl_add_u32 %14, %14, 0, %8\nv_lshl_add_u32 %15, %15, 0, %8\nv_lshl_add_u32 %16, %16, 0, %8\nv_cndmask_b32_e32 %2, %2, %8, vcc\nv_cndmask_b32_e32 %12, %12, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %14, %14, 0, %8\nv_lshl_add_u32 %15, %15, 0, %8\nv_lshl_add_u32 %16, %16, 0, %8\nv_lshl_add_u32 %17, %17, 0, %8\nv_cndmask_b32_e32 %3, %3, %8, vcc\nv_cndmask_b32_e32 %13, %13, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %15, %15, 0, %8\nv_lshl_add_u32 %16, %16, 0, %8\nv_lshl_add_u32 %17, %17, 0, %8\nv_lshl_add_u32 %10, %10, 0, %8\nv_cndmask_b32_e32 %4, %4, %8, vcc\nv_cndmask_b32_e32 %14, %14, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %16, %16, 0, %8\nv_lshl_add_u32 %17, %17, 0, %8\nv_lshl_add_u32 %10, %10, 0, %8\nv_lshl_add_u32 %11, %11, 0, %8\nv_cndmask_b32_e32 %5, %5, %8, vcc\nv_cndmask_b32_e32 %15, %15, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %17, %17, 0, %8\nv_lshl_add_u32 %10, %10, 0, %8\nv_lshl_add_u32 %11, %11, 0, %8\nv_lshl_add_u32 %12, %12, 0, %8\nv_cndmask_b32_e32 %6, %6, %8, vcc\nv_cndmask_b32_e32 %16, %16, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %10, %10, 0, %8\nv_lshl_add_u32 %11, %11, 0, %8\nv_lshl_add_u32 %12, %12, 0, %8\nv_lshl_add_u32 %13, %13, 0, %8\nv_cndmask_b32_e32 %7, %7, %8, vcc\nv_cndmask_b32_e32 %17, %17, %8, vcc\n", 7)
// a compare, 8 independent VOP3 instructions, then the VOP2 selects on VCC:
// does the select's cost depend on how long ago VCC was written?
UB_KERNEL_I(cmp_gap8_sel2_vcc,
    "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %11, %11, 0, %8\nv_lshl_add_u32 %12, %12, 0, %8\nv_lshl_add_u32 %13, %13, 0, %8\nv_lshl_add_u32 %14, %14, 0, %8\nv_lshl_add_u32 %15, %15, 0, %8\nv_lshl_add_u32 %16, %16, 0, %8\nv_lshl_add_u32 %17, %17, 0, %8\nv_lshl_add_u32 %10, %10, 0, %8\nv_cndmask_b32_e32 %0, %0, %8, vcc\nv_cndmask_b32_e32 %10, %10, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %12, %12, 0, %8\nv_lshl_add_u32 %13, %13, 0, %8\nv_lshl_add_u32 %14, %14, 0, %8\nv_lshl_add_u32 %15, %15, 0, %8\nv_lshl_add_u32 %16, %16, 0, %8\nv_lshl_add_u32 %17, %17, 0, %8\nv_lshl_add_u32 %10, %10, 0, %8\nv_lshl_add_u32 %11, %11, 0, %8\nv_cndmask_b32_e32 %1, %1, %8, vcc\nv_cndmask_b32_e32 %11, %11, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %13, %13, 0, %8\nv_lshl_add_u32 %14, %14, 0, %8\nv_lshl_add_u32 %15, %15, 0, %8\nv_lshl_add_u32 %16, %16, 0, %8\nv_lshl_add_u32 %17, %17, 0, %8\nv_lshl_add_u32 %10, %10, 0, %8\nv_lshl_add_u32 %11, %11, 0, %8\nv_lshl_add_u32 %12, %12, 0, %8\nv_cndmask_b32_e32 %2, %2, %8, vcc\nv_cndmask_b32_e32 %12, %12, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %14, %14, 0, %8\nv_lshl_add_u32 %15, %15, 0, %8\nv_lshl_add_u32 %16, %16, 0, %8\nv_lshl_add_u32 %17, %17, 0, %8\nv_lshl_add_u32 %10, %10, 0, %8\nv_lshl_add_u32 %11, %11, 0, %8\nv_lshl_add_u32 %12, %12, 0, %8\nv_lshl_add_u32 %13, %13, 0, %8\nv_cndmask_b32_e32 %3, %3, %8, vcc\nv_cndmask_b32_e32 %13, %13, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %15, %15, 0, %8\nv_lshl_add_u32 %16, %16, 0, %8\nv_lshl_add_u32 %17, %17, 0, %8\nv_lshl_add_u32 %10, %10, 0, %8\nv_lshl_add_u32 %11, %11, 0, %8\nv_lshl_add_u32 %12, %12, 0, %8\nv_lshl_add_u32 %13, %13, 0, %8\nv_lshl_add_u32 %14, %14, 0, %8\nv_cndmask_b32_e32 %4, %4, %8, vcc\nv_cndmask_b32_e32 %14, %14, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %16, %16, 0, %8\nv_lshl_add_u32 %17, %17, 0, %8\nv_lshl_add_u32 %10, %10, 0, %8\nv_lshl_add_u32 %11, %11, 0, %8\nv_lshl_add_u32 %12, %12, 0, %8\nv_lshl_add_u32 %13, %13, 0, %8\nv_lshl_add_u32 %14, %14, 0, %8\nv_lshl_add_u32 %15, %15, 0, %8\nv_cndmask_b32_e32 %5, %5, %8, vcc\nv_cndmask_b32_e32 %15, %15, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %17, %17, 0, %8\nv_lshl_add_u32 %10, %10, 0, %8\nv_lshl_add_u32 %11, %11, 0, %8\nv_lshl_add_u32 %12, %12, 0, %8\nv_lshl_add_u32 %13, %13, 0, %8\nv_lshl_add_u32 %14, %14, 0, %8\nv_lshl_add_u32 %15, %15, 0, %8\nv_lshl_add_u32 %16, %16, 0, %8\nv_cndmask_b32_e32 %6, %6, %8, vcc\nv_cndmask_b32_e32 %16, %16, %8, vcc\n" "v_cmp_lt_f64 vcc, %20, %18\nv_lshl_add_u32 %10, %10, 0, %8\nv_lshl_add_u32 %11, %11, 0, %8\nv_lshl_add_u32 %12, %12, 0, %8\nv_lshl_add_u32 %13, %13, 0, %8\nv_lshl_add_u32 %14, %14, 0, %8\nv_lshl_add_u32 %15, %15, 0, %8\nv_lshl_add_u32 %16, %16, 0, %8\nv_lshl_add_u32 %17, %17, 0, %8\nv_cndmask_b32_e32 %7, %7, %8, vcc\nv_cndmask_b32_e32 %17, %17, %8, vcc\n", 11)
UB_KERNEL_I(addc_co_vcc,
    "v_addc_co_u32 %0, vcc, %0, %8, vcc\n" "v_addc_co_u32 %1, vcc, %1, %8, vcc\n" "v_addc_co_u32 %2, vcc, %2, %8, vcc\n" "v_addc_co_u32 %3, vcc, %3, %8, vcc\n" "v_addc_co_u32 %4, vcc, %4, %8, vcc\n" "v_addc_co_u32 %5, vcc, %5, %8, vcc\n" "v_addc_co_u32 %6, vcc, %6, %8, vcc\n" "v_addc_co_u32 %7, vcc, %7, %8, vcc\n", 1)
UB_KERNEL_I(add_co_vcc,
    "v_add_co_u32 %0, vcc, %0, %8\n" "v_add_co_u32 %1, vcc, %1, %8\n" "v_add_co_u32 %2, vcc, %2, %8\n" "v_add_co_u32 %3, vcc, %3, %8\n" "v_add_co_u32 %4, vcc, %4, %8\n" "v_add_co_u32 %5, vcc, %5, %8\n" "v_add_co_u32 %6, vcc, %6, %8\n" "v_add_co_u32 %7, vcc, %7, %8\n", 1)
UB_KERNEL_I(mov_b32_indep,
    "v_mov_b32 %0, %8\n" "v_mov_b32 %1, %8\n" "v_mov_b32 %2, %8\n" "v_mov_b32 %3, %8\n" "v_mov_b32 %4, %8\n" "v_mov_b32 %5, %8\n" "v_mov_b32 %6, %8\n" "v_mov_b32 %7, %8\n", 1)
UB_KERNEL_I(fma_f32_indep,
    "v_fma_f32 %0, %0, %8, %8\n" "v_fma_f32 %1, %1, %8, %8\n" "v_fma_f32 %2, %2, %8, %8\n" "v_fma_f32 %3, %3, %8, %8\n" "v_fma_f32 %4, %4, %8, %8\n" "v_fma_f32 %5, %5, %8, %8\n" "v_fma_f32 %6, %6, %8, %8\n" "v_fma_f32 %7, %7, %8, %8\n", 1)
// 32-bit transcendentals (operands are bit patterns; only the issue cost
// matters): would an fp32 seed be cheaper than the fp64 one?
UB_KERNEL_I(rsq_f32_indep,
    "v_rsq_f32 %0, %0\n" "v_rsq_f32 %1, %1\n" "v_rsq_f32 %2, %2\n" "v_rsq_f32 %3, %3\n" "v_rsq_f32 %4, %4\n" "v_rsq_f32 %5, %5\n" "v_rsq_f32 %6, %6\n" "v_rsq_f32 %7, %7\n", 1)
UB_KERNEL_I(rcp_f32_indep,
    "v_rcp_f32 %0, %0\n" "v_rcp_f32 %1, %1\n" "v_rcp_f32 %2, %2\n" "v_rcp_f32 %3, %3\n" "v_rcp_f32 %4, %4\n" "v_rcp_f32 %5, %5\n" "v_rcp_f32 %6, %6\n" "v_rcp_f32 %7, %7\n", 1)
UB_KERNEL_I(sqrt_f32_indep,
    "v_sqrt_f32 %0, %0\n" "v_sqrt_f32 %1, %1\n" "v_sqrt_f32 %2, %2\n" "v_sqrt_f32 %3, %3\n" "v_sqrt_f32 %4, %4\n" "v_sqrt_f32 %5, %5\n" "v_sqrt_f32 %6, %6\n" "v_sqrt_f32 %7, %7\n", 1)
UB_KERNEL_I(lshl_add_u32_indep,
    "v_lshl_add_u32 %0, %0, 1, %8\n" "v_lshl_add_u32 %1, %1, 1, %8\n" "v_lshl_add_u32 %2, %2, 1, %8\n" "v_lshl_add_u32 %3, %3, 1, %8\n" "v_lshl_add_u32 %4, %4, 1, %8\n" "v_lshl_add_u32 %5, %5, 1, %8\n" "v_lshl_add_u32 %6, %6, 1, %8\n" "v_lshl_add_u32 %7, %7, 1, %8\n", 1)

typedef void (*kern_t)(unsigned long long *, double, double, int,
                       unsigned long long);
struct Entry { const char *name; kern_t thr, dep; };
#define E(NAME) {#NAME, NAME##_thr, NAME##_dep}
static const Entry entries[] = {
    E(fma_f64), E(fma_f64_sgpr), E(add_f64), E(mul_f64), E(max_f64),
    E(min_f64), E(mov_b64), E(ldexp_f64), E(rndne_f64), E(fract_f64),
    E(cvt_f64_i32), E(cvt_i32_f64), E(frexp_mant_f64), E(frexp_exp_f64),
    E(rcp_f64), E(rsq_f64), E(mix_rsq_7fma), E(mix_rsq_3fma), E(sqrt_f64), E(div_scale_f64), E(div_fmas_f64),
    E(div_fixup_f64), E(trig_preop_f64), E(cmp_lt_f64_vcc),
    E(cmp_lt_f64_sgpr), E(cmp_class_f64_vcc), E(cmp_class_f64_sgpr),
    E(cndmask_b32_vcc), E(cndmask_b32_sgpr), E(mov_b32), E(add_u32),
    E(and_b32), E(lshl_add_u32), E(ashrrev_i32), E(lshl_add_u64),
    E(mad_u64_u32), E(fma_f32), E(cvt_f32_f64), E(cvt_f64_f32),
    E(readlane_b32), E(readfirstlane_b32), E(writelane_b32),
    E(sel_vop2_vcc), E(sel_vop3_vcc), E(sel_vop3_sgpr), E(cmp_sel2_vcc), E(cmp_sel2_sgpr), E(cmp_gap1_sel2_vcc), E(cmp_gap2_sel2_vcc), E(cmp_gap4_sel2_vcc), E(cmp_gap8_sel2_vcc), E(addc_co_vcc), E(add_co_vcc), E(mov_b32_indep), E(fma_f32_indep), E(rsq_f32_indep), E(rcp_f32_indep), E(sqrt_f32_indep), E(lshl_add_u32_indep)};

static double run(kern_t k, int waves_per_simd, int iters,
                  unsigned long long *d_out, unsigned long long mask)
{
    const int nw = 4 * waves_per_simd;
    unsigned long long h[64];
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * nw), 0, 0, d_out, 1.0000001,
                           1e-9, iters, mask);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d_out, sizeof(unsigned long long) * nw,
                  hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int w = 0; w < nw; ++w) mx = h[w] > mx ? h[w] : mx;
        const double c = (double)mx /
                         ((double)waves_per_simd * iters * REP * CHAINS);
        best = c < best ? c : best;
    }
    return best;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned long long *d_out;
    (void)hipMalloc(&d_out, 8 * 8192);
    (void)hipMemset(d_out, 0, 8 * 8192);
    printf("# cycles per wave64 instruction per SIMD (s_memtime ticks); "
           "thr@W = W waves per SIMD, 8 independent chains; dep = one "
           "dependent chain, 1 wave per SIMD\n");
    printf("%-22s %8s %8s %8s %8s\n", "instruction", "thr@1", "thr@2",
           "thr@4", "dep");
    for (const Entry &e : entries) {
        const double t1 = run(e.thr, 1, iters, d_out, ~0ull);
        const double t2 = run(e.thr, 2, iters, d_out, ~0ull);
        const double t4 = run(e.thr, 4, iters, d_out, ~0ull);
        const double d = run(e.dep, 1, iters, d_out, ~0ull);
        printf("%-22s %8.2f %8.2f %8.2f %8.2f\n", e.name, t1, t2, t4, d);
    }
    // partially filled waves: does the SIMD skip passes with no active lane?
    printf("# v_fma_f64 with a partial EXEC mask (thr@4)\n");
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull,
                                        0xffull, 0x1ull,
                                        0xffff0000ffffull};
    for (unsigned long long m : masks)
        printf("exec=%016llx     %8.2f\n", m,
               run(fma_f64_thr, 4, iters, d_out, m));
    (void)hipFree(d_out);
    return 0;
}
