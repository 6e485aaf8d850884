// fastmath.h -- the transcendental pieces of the ensemble kernels in fp64:
//   fastpow_soil      (soil / FC)**Beta from the soil alone, table-driven in
//                     plain double, 25 vector instructions (OCML pow: ~224)
//   fast_tanh parts   tanh as a numerator / denominator pair: a [9/8] Pade
//                     approximant inside |a| <= 1, expm1-based beyond
//   inv_fourth_root   b**(-1/4), b >= 1, ~1.5 ulp, ~16 instructions; a
//                     degree-7 polynomial for b - 1 <= 0.0416
//
// Why: after the scalar forcing loads and coalesced stores, the kernels are
// fp64-VALU-issue bound, and the reference's libm calls (numba ->
// llvm.pow.f64 / tanh) were most of their instructions.  The platform libm is
// itself only faithful to <1 ulp, so an implementation of comparable accuracy
// is an equally valid realisation of the same statement; parity is asserted
// at 1e-10 relative on the discharge (observed ~1e-14 ... 3e-13).  Arguments
// outside a form's domain are voted out by its caller and take the general
// libm function, so IEEE special cases stay exactly the reference's.
// (Three earlier generations of the power -- double-double series, table-driven
// double-double, table-driven plain double behind the quotient soil / FC: 71,
// 50 and 34 instructions -- were removed in round 6 with their tables; their
// error analyses and A/B numbers: profiles/README.md rounds 1-4.)
//
// The same source compiles for the host (tests/test_fastmath_cpu.py builds a
// small harness with g++ and checks it against 80-bit powl): only the
// primitives below differ.
#pragma once

#include <math.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define FP_FN __device__ __forceinline__
#define FP_RCP(g) __builtin_amdgcn_rcp(g)
#define FP_FMA(a, b, c) __builtin_fma((a), (b), (c))
// Horner step a*b + CONSTANT as ONE v_fma_f64 with the constant in an SGPR
// pair.  Left to itself hipcc keeps each polynomial coefficient in a VGPR pair
// (42 VGPRs for the two polynomials) and issues v_mov_b64 + v_fmac_f64 per
// step; this form halves the instructions of both Horner chains and frees
// the VGPRs.  Register-only VALU: no wait states or counters involved.
static __device__ __forceinline__ double fp_fma_sconst_(double a, double b,
                                                        double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
#define FP_FMA_C(a, b, c) fp_fma_sconst_((a), (b), (c))
// ... or in a VGPR pair: kernels that run at one or two waves per SIMD have
// registers to spare but need their SGPRs for the prefetched forcing record
// (hbvedu.hip, small sweeps).  Written as asm so that the constant stays a
// loop-invariant register instead of becoming v_mov_b64 + v_fmac_f64.
static __device__ __forceinline__ double fp_fma_vconst_(double a, double b,
                                                        double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
#define FP_FMA_CV(a, b, c) fp_fma_vconst_((a), (b), (c))
// a * K + 0.5 with K in an SGPR pair and 0.5 as the ISA's inline constant
static __device__ __forceinline__ double fp_fma_half_(double a, double k)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, 0.5" : "=v"(d) : "v"(a), "s"(k));
    return d;
}
#define FP_FMA_HALF(a, k) fp_fma_half_((a), (k))
#define FP_RINT(v) __builtin_rint(v)
#define FP_FREXP_MANT(x) __builtin_amdgcn_frexp_mant(x)
#define FP_FREXP_EXP(x) __builtin_amdgcn_frexp_exp(x)
#define FP_LDEXP(v, n) __builtin_amdgcn_ldexp((v), (n))
#define FP_RSQ_APPROX(v) __builtin_amdgcn_rsq(v)      /* v_rsq_f64, ~2^-26 */
#define FP_SQRT_APPROX(v) __builtin_amdgcn_sqrt(v)    /* v_sqrt_f64, ~2^-26 */
#define FP_HI32(x) __double2hiint(x)
#define FP_LO32(x) __double2loint(x)
#define FP_FROM_HILO(hi, lo) __hiloint2double((hi), (lo))
#elif defined(__HIPCC__)
// host pass of a .hip translation unit: the function is never called there
#define FP_FN __device__ __forceinline__
#define FP_RCP(g) (1.0 / (g))
#define FP_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define FP_FMA_C(a, b, c) __builtin_fma((a), (b), (c))
#define FP_FMA_CV(a, b, c) __builtin_fma((a), (b), (c))
#define FP_FMA_HALF(a, k) __builtin_fma((a), (k), 0.5)
#define FP_RINT(v) __builtin_rint(v)
#define FP_FREXP_MANT(x) (x)
#define FP_FREXP_EXP(x) 0
#define FP_LDEXP(v, n) (v)
#define FP_RSQ_APPROX(v) (v)
#define FP_SQRT_APPROX(v) (v)
#define FP_HI32(x) 0
#define FP_LO32(x) 0
#define FP_FROM_HILO(hi, lo) 0.0
#else
#define FP_FN static inline
#define FP_FMA(a, b, c) fma((a), (b), (c))
#define FP_FMA_C(a, b, c) fma((a), (b), (c))
#define FP_FMA_CV(a, b, c) fma((a), (b), (c))
#define FP_FMA_HALF(a, k) fma((a), (k), 0.5)
#define FP_RINT(v) rint(v)
static inline double fp_frexp_mant_(double x) { int e; return frexp(x, &e); }
static inline int fp_frexp_exp_(double x) { int e; (void)frexp(x, &e); return e; }
#define FP_FREXP_MANT(x) fp_frexp_mant_(x)
#define FP_FREXP_EXP(x) fp_frexp_exp_(x)
#define FP_LDEXP(v, n) ldexp((v), (n))
/* host stand-ins for the hardware's ~26-bit estimates: the exact value with
 * its significand cut to 24 bits (any exponent, unlike a float) */
#include <stdint.h>
#include <string.h>
static inline double fp_cut24_(double v) {
    uint64_t b; memcpy(&b, &v, 8); b &= ~((1ULL << 29) - 1); memcpy(&v, &b, 8); return v;
}
#define FP_RCP(g) fp_cut24_(1.0 / (g))
#define FP_RSQ_APPROX(v) fp_cut24_(1.0 / sqrt(v))
#define FP_SQRT_APPROX(v) fp_cut24_(sqrt(v))
static inline int fp_hi32_(double x) { uint64_t b; memcpy(&b, &x, 8); return (int)(uint32_t)(b >> 32); }
static inline int fp_lo32_(double x) { uint64_t b; memcpy(&b, &x, 8); return (int)(uint32_t)b; }
static inline double fp_from_hilo_(int hi, int lo) {
    uint64_t b = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double d; memcpy(&d, &b, 8); return d;
}
#define FP_HI32(x) fp_hi32_(x)
#define FP_LO32(x) fp_lo32_(x)
#define FP_FROM_HILO(hi, lo) fp_from_hilo_((hi), (lo))
#endif

// ---------------------------------------------------------------------------
// (soil / FC) ** Beta for HBV-Edu's effective precipitation
// (hbvedu_model.py:99), round 5: the power of a QUOTIENT BY A LOOP INVARIANT,
// evaluated on the numerator alone --
//     (soil / FC) ** Beta = 2 ** ((Beta / ln 2) (ln soil - ln FC)),
//     N zz = sN = fma(y2N, ln soil, -cF),   y2N = N Beta / ln 2,
//                                           cF  = y2N ln FC   (per lane, once)
// -- so the quotient is never formed (one multiply and one link of the
// dependent chain less), with both halves table-driven in plain double:
//   * ln soil: soil = 2^e m, m in [1/2, 1) (v_frexp_exp / v_frexp_mant -- two
//     instructions where the bit-pattern split of round 4's table lookup took
//     five), the top 9 mantissa bits select {invc, lnc} (pow2_tables.h, 512
//     entries of 16 bytes: one ds_read_b128), r = fma(m, invc, -1), |r| <=
//     2^-10, ln(1 + r) = r + r^2 (A0 + A1 r + A2 r^2) (4.5e-17 absolute: two
//     Horner steps where 128 subintervals needed five);
//   * 2 ** (sN / N), N = 256: k = RN(sN) and the table index come out of ONE
//     addition -- t = sN + 1.5 2^52 holds k in its low mantissa bits (round
//     to nearest even of the hardware), kd = t - 1.5 2^52, d = sN - kd exact
//     -- instead of v_ldexp + v_rndne + v_cvt; q of degree 3 on |d| / N <=
//     1/512 (fit error 5e-18).
// 28 vector instructions where round 4's power behind the quotient took
// 34.  What the subtraction ln soil - ln FC costs: the rounding of ln soil (a
// few 2^-53 |ln soil|) is no longer relative to |ln(soil / FC)|, so the
// relative error of the result is at most
//     (6 + 3 |zz| + |y| (1 + 3 |log2 FC| + 3 |log2 soil|)) 2^-53,  zz = y log2 x
// -- 5e-15 in a sane run's box (Beta <= 6, FC <= 1000 mm), 1e-12 at the far
// corners of the guard box (Beta = 64, FC = 1e6) -- against the 1e-10 the
// discharge has to meet; measured by tests/native/fastmath_harness.cpp
// ("soil_*").  Domain: soil a positive normal number, |sN| < 1000 N
// (fastpow_soil_ok); y2N and cF finite.
#include "pow2_tables.h"
struct FpSoilEntry { double invc, lnc; };
#define FP_INVLN2HI 0x1.71547652b82fep+0          /* RN(1 / ln 2) */
// VCONST: polynomial coefficients in VGPRs instead of SGPRs (see FP_FMA_CV)
#define FP_FMA_K(a, b, c) \
    (VCONST ? FP_FMA_CV((a), (b), (c)) : FP_FMA_C((a), (b), (c)))
#define FP_SOIL_MAGIC 0x1.8p52
static_assert(FP_SOIL_LOG_A0 == -0.5, "the inline constant below");
#if !defined(__HIPCC__)
static inline int fp_soil_lo32_(double t) {
    uint64_t b; memcpy(&b, &t, 8); return (int)(uint32_t)b;
}
#endif
// ln x for a positive normal x, the first half of fastpow_soil
template <bool VCONST = false>
FP_FN double fastpow_soil_log(double x, const FpSoilEntry *tabl)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double m = FP_FREXP_MANT(x);
    const double ed = (double)FP_FREXP_EXP(x);
    const int i = (FP_HI32(x) >> (20 - FP_SOIL_LOG_BITS)) & (FP_SOIL_LOG_N - 1);
#elif defined(__HIPCC__)
    const double m = x, ed = 0.0;
    const int i = 0;
#else
    int e_;
    const double m = frexp(x, &e_);
    const double ed = (double)e_;
    const int i = (FP_HI32(x) >> (20 - FP_SOIL_LOG_BITS)) & (FP_SOIL_LOG_N - 1);
#endif
    const FpSoilEntry en = tabl[i];
    const double r = FP_FMA(m, en.invc, -1.0);
    const double t = FP_FMA(ed, FP_SOIL_LN2, en.lnc);
    double h = FP_SOIL_LOG_A2;
    h = FP_FMA_K(h, r, FP_SOIL_LOG_A1);
    h = FP_FMA(h, r, -0.5);                           // inline constant (A0)
    return t + FP_FMA(r * r, h, r);
}
// the per-lane constants: y2N = N Beta / ln 2, cF = y2N ln FC (ln FC by the
// same table-driven logarithm: the kernels need no second one).  An FC that
// is not a positive normal number gives cF = NaN, and fastpow_soil_ok then
// rejects every result.
FP_FN void fastpow_soil_exponent(double y, double fc, const FpSoilEntry *tabl,
                                 double *y2N, double *cF)
{
    const double h = (y * FP_INVLN2HI) * (double)FP_SOIL_EXP_N;
    const bool ok = (fc >= 0x1p-1022) && (fc < __builtin_inf());
    *y2N = h;
    *cF = ok ? h * fastpow_soil_log<false>(ok ? fc : 1.0, tabl)
             : __builtin_nan("");
}
template <bool VCONST = false>
FP_FN double fastpow_soil(double x, double y2N, double cF,
                          const FpSoilEntry *tabl, const double *tabe,
                          double *sN_out)
{
    const double l = fastpow_soil_log<VCONST>(x, tabl);
    const double sN = FP_FMA(y2N, l, -cF);
    *sN_out = sN;
    const double tm = sN + FP_SOIL_MAGIC;
    const double kd = tm - FP_SOIL_MAGIC;
    const double d = sN - kd;                         // exact
#if defined(__HIP_DEVICE_COMPILE__)
    const int k = FP_LO32(tm);
#elif defined(__HIPCC__)
    const int k = 0;
#else
    const int k = fp_soil_lo32_(tm);
#endif
    const double tj = tabe[k & (FP_SOIL_EXP_N - 1)];
    constexpr double I = 1.0 / FP_SOIL_EXP_N;         // a power of two
    double q = FP_SOIL_EXP_C3 * (I * I * I * I);
    q = FP_FMA_K(q, d, FP_SOIL_EXP_C2 * (I * I * I));
    q = FP_FMA_K(q, d, FP_SOIL_EXP_C1 * (I * I));
    q = FP_FMA_K(q, d, FP_SOIL_EXP_C0 * I);
    return FP_LDEXP(FP_FMA(tj, d * q, tj), k >> FP_SOIL_EXP_BITS);
}
FP_FN bool fastpow_soil_ok(double x, double sN)
{
    return (x >= 0x1p-1022) && (x < __builtin_inf()) &&
           (__builtin_fabs(sN) < 1000.0 * FP_SOIL_EXP_N);
}

// ---------------------------------------------------------------------------
// tanh(a), any a.  tanh(a) = E / (E + 2) with E = expm1(2|a|), sign restored.
//   2|a| = n ln2 + r, |r| <= ln2/2 (ln2 split hi/lo so n*ln2_hi is exact);
//   expm1(r) = r + r^2 q(r), q of degree 10 through the Chebyshev nodes of
//   the interval (csrc/tools/gen_exp_poly.py, 0.008 x 2^-53; the Taylor
//   polynomial of that accuracy has degree 11);
//   E = 2^n expm1(r) + (2^n - 1) in one FMA (2^n - 1 is exact for n <= 53).
// |a| is clamped to 20 first (tanh(20) rounds to 1.0; keeps 2^n finite and
// makes +-inf give +-1); NaN propagates; tanh(+-0) = +-0.  Error <= ~2.5 ulp
// (libm: 1 ulp), measured by tests/native/fastmath_harness.cpp.
// Used for np.tanh(p_n / x1) of GR4J (reference: gr4j_model.py:95-96, 107-108).
// fast_tanh_parts gives numerator and denominator (tanh(a) = num / den,
// den >= 1) so that a caller can fold the quotient into one of its own.
//
// JIT_CONST (device only): the 14 constants are fetched from constant memory
// with scalar loads at the point of use instead of living in SGPRs for the
// whole time loop.  The big fused kernels (snow routine + GR4J) run out of
// SGPRs otherwise and hipcc parks the overflow in VGPR lanes, paying a
// v_readlane -- a VALU slot -- per use.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const double __attribute__((address_space(4))) *fp_cptr_t;
static __device__ __constant__ const double FP_TANH_TABLE[16] = {
    2.0914679376583935e-09, 2.510520637395701e-08, 2.7557273661348637e-07,
    2.7557255425746435e-06, 2.4801587325533363e-05, 0.00019841269874800493,
    0.0013888888888883752, 0.008333333333326141, 0.04166666666666667,
    0.1666666666666667, 0.5,
    1.4426950408889634, 6.93147180369123816490e-01,
    1.90821492927058770002e-10, 0.0, 0.0};
#endif

// CONSTS: where the 14 constants live -- 0 SGPR pairs, 1 VGPR pairs (sweeps of
// at most two waves per SIMD: registers to spare, and no scalar load whose
// latency nobody hides), 2 constant memory at the point of use (JIT_CONST).
template <int CONSTS = 0>
FP_FN void fast_tanh_parts(double a, double &num, double &den)
{
    const double ax = __builtin_fabs(a);
    const double x = (ax > 20.0) ? 20.0 : ax;        // NaN stays NaN
    const double y = 2.0 * x;
    double p, n;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (CONSTS == 2) {
        fp_cptr_t c = (fp_cptr_t)FP_TANH_TABLE;
        asm volatile("" : "+s"(c));   // keeps the loads inside the time loop
        n = FP_RINT(y * c[11]);
        const double r = FP_FMA(-n, c[13], FP_FMA(-n, c[12], y));
        double q = c[0];
#pragma unroll
        for (int j = 1; j < 11; ++j) q = FP_FMA_C(q, r, c[j]);
        p = FP_FMA(r * r, q, r);
    } else
#endif
    {
        n = FP_RINT(y * 1.4426950408889634);
        const double r = FP_FMA(-n, 1.90821492927058770002e-10,
                                FP_FMA(-n, 6.93147180369123816490e-01, y));
        double q = 2.0914679376583935e-09;           // see FP_TANH_TABLE
#define FP_TANH_STEP(c) \
    q = (CONSTS == 1) ? FP_FMA_CV(q, r, (c)) : FP_FMA_C(q, r, (c))
        FP_TANH_STEP(2.510520637395701e-08);
        FP_TANH_STEP(2.7557273661348637e-07);
        FP_TANH_STEP(2.7557255425746435e-06);
        FP_TANH_STEP(2.4801587325533363e-05);
        FP_TANH_STEP(0.00019841269874800493);
        FP_TANH_STEP(0.0013888888888883752);
        FP_TANH_STEP(0.008333333333326141);
        FP_TANH_STEP(0.04166666666666667);
        FP_TANH_STEP(0.1666666666666667);
        FP_TANH_STEP(0.5);
#undef FP_TANH_STEP
        p = FP_FMA(r * r, q, r);                     // expm1(r)
    }
    // H = expm1(2|a|) / 2 and H + 1: tanh = H / (H + 1).  (Halved so that a
    // caller multiplying num by a huge factor -- GR4J with x1 ~ 1e308 and a
    // net rainfall tiny next to it -- overflows exactly where the product
    // with tanh itself would: for |a| -> 0, H + 1 -> 1.)
    const double half_two_n = FP_LDEXP(1.0, (int)n - 1);
    const double H = FP_FMA(half_two_n, p, half_two_n - 0.5);
    num = __builtin_copysign(H, a);
    den = H + 1.0;
}

// tanh(a) = num / den for |a| <= FP_TANHR_AMAX as the [9/8] Pade approximant
// (the continued fraction a / (1 + a^2 / (3 + a^2 / (5 + ...))) cut after
// its ninth level, scaled so that both polynomials start with 1):
//     num = a (1 + n1 a^2 + n2 a^4 + n3 a^6 + n4 a^8),
//     den =    1 + d1 a^2 + d2 a^4 + d3 a^6 + d4 a^8,
// approximation error 0.23 x 2^-53 at |a| = 1 (0.002 at 0.75), odd in a (no
// sign handling), no exponential, no range reduction: TEN instructions
// against fast_tanh_parts' 28, and like it a numerator / denominator pair
// for the caller to fold its own quotient into.  GR4J's argument is net
// rainfall over the production store's capacity: below 1 on every day of a
// sane run; anything else (NaN included) is the caller's vote and
// fast_tanh_parts.  Error of num / den evaluated in double: <= 3.7 ulp
// (tests/native/fastmath_harness.cpp).
#define FP_TANHR_AMAX 1.0
#define FP_TANHR_N1 0.13725490196078433      /* 4729725 / 34459425 */
#define FP_TANHR_N2 0.00392156862745098      /* 135135 / 34459425  */
#define FP_TANHR_N3 2.8729440494146376e-05   /* 990 / 34459425     */
#define FP_TANHR_N4 2.901963686277412e-08    /* 1 / 34459425       */
#define FP_TANHR_D1 0.47058823529411764      /* 16216200 / 34459425 */
#define FP_TANHR_D2 0.027450980392156862     /* 945945 / 34459425  */
#define FP_TANHR_D3 0.00040221216691804925   /* 13860 / 34459425   */
#define FP_TANHR_D4 1.3058836588248353e-06   /* 45 / 34459425      */
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __constant__ const double FP_TANHR_TABLE[8] = {
    FP_TANHR_N4, FP_TANHR_N3, FP_TANHR_N2, FP_TANHR_N1,
    FP_TANHR_D4, FP_TANHR_D3, FP_TANHR_D2, FP_TANHR_D1};
#endif
template <int CONSTS = 0>
FP_FN void fast_tanh_rational_parts(double a, double &num, double &den)
{
    const double a2 = a * a;
    double p, q;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (CONSTS == 2) {
        fp_cptr_t c = (fp_cptr_t)FP_TANHR_TABLE;
        asm volatile("" : "+s"(c));   // keeps the loads inside the time loop
        p = c[0];
        p = FP_FMA_C(p, a2, c[1]);
        p = FP_FMA_C(p, a2, c[2]);
        p = FP_FMA_C(p, a2, c[3]);
        q = c[4];
        q = FP_FMA_C(q, a2, c[5]);
        q = FP_FMA_C(q, a2, c[6]);
        q = FP_FMA_C(q, a2, c[7]);
    } else
#endif
    {
#define FP_TANHR_STEP(v, c) \
    v = (CONSTS == 1) ? FP_FMA_CV(v, a2, (c)) : FP_FMA_C(v, a2, (c))
        p = FP_TANHR_N4;
        FP_TANHR_STEP(p, FP_TANHR_N3);
        FP_TANHR_STEP(p, FP_TANHR_N2);
        FP_TANHR_STEP(p, FP_TANHR_N1);
        q = FP_TANHR_D4;
        FP_TANHR_STEP(q, FP_TANHR_D3);
        FP_TANHR_STEP(q, FP_TANHR_D2);
        FP_TANHR_STEP(q, FP_TANHR_D1);
#undef FP_TANHR_STEP
    }
    p = FP_FMA(p, a2, 1.0);                          // inline constant
    den = FP_FMA(q, a2, 1.0);
    num = a * p;
}

FP_FN double fast_tanh(double a)
{
    double num, den;
    fast_tanh_parts(a, num, den);
    return num / den;
}

// ---------------------------------------------------------------------------
// sqrt(x) for normal x >= 2^-500: hardware estimate y ~ 1/sqrt(x) (~2^-26),
// one coupled Newton step on g ~ sqrt(x), h ~ 1/(2 sqrt(x)) and a final
// residual correction -- 8 instructions, error <= ~0.6 ulp (the compiler's
// correctly rounded sqrt, with its range scaling and fix-ups, takes 18).
// No range handling of its own: callers guard (0, inf, NaN, negatives).
FP_FN double fast_sqrt_core(double x)
{
    const double y = FP_RSQ_APPROX(x);
    double g = x * y, h = 0.5 * y;
    const double r = FP_FMA(-h, g, 0.5);
    g = FP_FMA(g, r, g);
    h = FP_FMA(h, r, h);
    const double d = FP_FMA(-g, g, x);
    return FP_FMA(d, h, g);
}

// ---------------------------------------------------------------------------
// b**(-1/4) for b >= 1 (or NaN).  Hardware estimate y0 = sqrt(rsq(b))
// (~2^-25), then two Newton steps for y^-4 = b:
//     y <- y + (y/4) (1 - b y^4)
// (quadratic: 2^-25 -> 2^-50 -> below rounding), ~16 instructions instead of
// sqrt, sqrt, divide (56).  Error <= ~1.5 ulp.  b is clamped to 1e300: for
// b >= 1e300 (inf included) the result is < 1e-75, which every caller only
// uses as 1 - result == 1.0 exactly; NaN propagates.
// Used for (1 + v**4)**(-0.25) of GR4J (reference: gr4j_model.py:117, 145).
// (core: finite b >= 1 only -- +inf would give NaN)
FP_FN double inv_fourth_root_core(double bb)
{
    double y = FP_SQRT_APPROX(FP_RSQ_APPROX(bb));
    double y2 = y * y;
    double e = FP_FMA(-bb, y2 * y2, 1.0);
    y = FP_FMA(y * 0.25, e, y);
    y2 = y * y;
    e = FP_FMA(-bb, y2 * y2, 1.0);
    y = FP_FMA(y * 0.25, e, y);
    return y;
}

// The same root with ONE third-order step instead of two Newton steps (6
// instructions after the estimate instead of 10): with e = 1 - b y^4,
//     b^(-1/4) = y (1 - e)^(-1/4) = y (1 + e/4 + 5 e^2/32 + 15 e^3/128 ...),
// and |e| <= ~2^-23 from the hardware estimates, so the cubic term is below
// 2^-70.  Error <= ~1 ulp (the last FMA's rounding plus 2^-53 from y e).
FP_FN double inv_fourth_root_core3(double bb)
{
    const double y = FP_SQRT_APPROX(FP_RSQ_APPROX(bb));
    const double y2 = y * y;
    const double e = FP_FMA(-bb, y2 * y2, 1.0);
    // y (1 + e/4 + 5e^2/32) = y + (y e / 2) (1/2 + 5e/16): 0.5 is an inline
    // constant of the ISA, 0.25 is not
    const double p = FP_FMA_HALF(e, 0.3125);
    return FP_FMA(y * e * 0.5, p, y);
}

// (1 + u)**(-1/4) for 0 <= u <= FP_R4_UMAX as 1 + u P(u), P of degree 7
// (csrc/tools/gen_root_poly.py; approximation error 0.009 x 2^-53): eight
// full-rate FMAs and no hardware estimate -- the Newton form above costs two
// quarter-rate instructions (16 cycles each) + 8, i.e. about twice as much.
// The result is the rounding of 1 + u P(u) with u P(u) exact to ~2^-60: error
// <= 0.51 ulp against the exact root of 1 + u.  (The reference rounds
// 1 + v**4 first; the caller therefore passes u = (1 + v**4) - 1, exact.)  GR4J's percolation
// (gr4j_model.py:117) has u = (4/9 S/x1)**4 <= 0.0391 whenever the production
// store is not above its capacity, which the model's equations maintain.
// CONSTS: 0 coefficients in SGPR pairs, 1 in VGPR pairs, 2 fetched from
// constant memory at the point of use (see fast_tanh_parts' JIT_CONST).
#define FP_R4_UMAX 0.0416
#define FP_R4_C0 -0.25
#define FP_R4_C1 0.15624999999996197
#define FP_R4_C2 -0.11718749998077582
#define FP_R4_C3 0.0952148400420375
#define FP_R4_C4 -0.08093226527063785
#define FP_R4_C5 0.0707978435375625
#define FP_R4_C6 -0.06270408403556514
#define FP_R4_C7 0.04930569258051168
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __constant__ const double FP_R4_TABLE[8] = {
    FP_R4_C7, FP_R4_C6, FP_R4_C5, FP_R4_C4,
    FP_R4_C3, FP_R4_C2, FP_R4_C1, FP_R4_C0};
#endif

template <int CONSTS = 0>
FP_FN double inv_fourth_root_1p_small(double u)
{
    double p;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (CONSTS == 2) {
        fp_cptr_t c = (fp_cptr_t)FP_R4_TABLE;
        asm volatile("" : "+s"(c));   // keeps the load inside the time loop
        p = c[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) p = FP_FMA_C(p, u, c[j]);
    } else if constexpr (CONSTS == 1) {
        p = FP_R4_C7;
        p = FP_FMA_CV(p, u, FP_R4_C6);
        p = FP_FMA_CV(p, u, FP_R4_C5);
        p = FP_FMA_CV(p, u, FP_R4_C4);
        p = FP_FMA_CV(p, u, FP_R4_C3);
        p = FP_FMA_CV(p, u, FP_R4_C2);
        p = FP_FMA_CV(p, u, FP_R4_C1);
        p = FP_FMA_CV(p, u, FP_R4_C0);
    } else
#endif
    {
        p = FP_R4_C7;
        p = FP_FMA_C(p, u, FP_R4_C6);
        p = FP_FMA_C(p, u, FP_R4_C5);
        p = FP_FMA_C(p, u, FP_R4_C4);
        p = FP_FMA_C(p, u, FP_R4_C3);
        p = FP_FMA_C(p, u, FP_R4_C2);
        p = FP_FMA_C(p, u, FP_R4_C1);
        p = FP_FMA_C(p, u, FP_R4_C0);
    }
    return FP_FMA(u, p, 1.0);
}

// n / d for a finite normal d in [1, 2^200] and finite n with |n| < 2^800:
// hardware reciprocal estimate, one Newton step on it, and a residual
// correction of the quotient -- 6 instructions (one of them quarter rate)
// instead of the IEEE sequence's 11 (one quarter rate); error <= ~1 ulp
// (correctly rounded except in rare double-rounding cases).  No scaling and
// no special cases: callers guard the domain.
FP_FN double fast_div_core(double n, double d)
{
    double y = FP_RCP(d);
    y = FP_FMA(FP_FMA(-d, y, 1.0), y, y);
    const double q0 = n * y;
    return FP_FMA(FP_FMA(-d, q0, n), y, q0);
}

FP_FN double inv_fourth_root(double b)
{
    const double bb = (b > 1e300) ? 1e300 : b;       // NaN stays NaN
    return inv_fourth_root_core(bb);
}
