// invdiv.h -- correctly rounded a / b for a loop-invariant b in 3 FMA-class
// instructions (see common.h for the why).  Portable: the host build is used
// by tests/test_fastmath_cpu.py to check bit-identity with `/`.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define ID_FN __host__ __device__ __forceinline__
#else
#define ID_FN static inline
#endif

struct InvDivisor {
    double b;     // the divisor
    double rb;    // RN(1 / b)
    bool ok;      // |b| in [2^-100, 2^100]
};

ID_FN InvDivisor make_inv_divisor(double b) {
    InvDivisor d;
    d.b = b;
    d.rb = 1.0 / b;
    d.ok = (fabs(b) >= 0x1p-100) && (fabs(b) <= 0x1p100);
    return d;
}

// |a| in [2^-900, 2^900]: shared by all quotients with the same numerator
ID_FN bool inv_div_numerator_ok(double a) {
    return (fabs(a) >= 0x1p-900) && (fabs(a) <= 0x1p900);
}

// ... or a is +0: q0 = +-0, r = +0 and the final FMA returns q0, the
// correctly signed zero a / b (stores that have run dry, an empty snow pack:
// common numerators, worth their own compare).  Not -0: there the final FMA
// gives (+0) + (-0) = +0 where the quotient is -0.
ID_FN bool inv_div_numerator_ok0(double a) {
    return inv_div_numerator_ok(a) || (a == 0.0 && !__builtin_signbit(a));
}

// The FAITHFUL form of the quotient: a * RN(1 / b), one instruction.  For d.ok
// it serves EVERY numerator -- NaN, infinities and signed zeros come out as
// the division gives them, nothing can be lost in a residual -- within
// (1 + 2^-53)^2 of a / b: 1.5 ulp at worst where the IEEE quotient is within
// 0.5 (tests/native/invdiv_harness.cpp).  What HBV-Edu's and GR4J's own
// quotients use (their results are few-ulp restatements of the reference's
// libm calls anyway); the snow routines, whose states
// are bit-identical to the reference's, keep inv_div_core.
ID_FN double inv_mul_core(double a, const InvDivisor &d) {
    return a * d.rb;
}

// valid (== RN(a / b)) when inv_div_numerator_ok[0](a) && d.ok
ID_FN double inv_div_core(double a, const InvDivisor &d) {
    const double q0 = a * d.rb;
    const double r = __builtin_fma(-d.b, q0, a);
    return __builtin_fma(r, d.rb, q0);
}
