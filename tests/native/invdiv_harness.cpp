// CPU harness for rrmpg_amd/csrc/invdiv.h (built and run by
// tests/test_fastmath_cpu.py): inv_div_core must be bit-identical to a / b
// wherever its guards hold.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../rrmpg_amd/csrc/invdiv.h"

static uint64_t s = 0x9E3779B97F4A7C15ULL;
static inline uint64_t rnd() {
    s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
    return s * 2685821237005ULL;
}
static inline double mk(uint64_t mant, int e, int neg) {
    uint64_t b = ((uint64_t)neg << 63) | ((uint64_t)(1023 + e) << 52) |
                 (mant & ((1ULL << 52) - 1));
    double d; memcpy(&d, &b, 8); return d;
}

int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 100000000L;
    long bad = 0, checked = 0;
    for (long i = 0; i < n; ++i) {
        uint64_t ma = rnd(), mb = rnd();
        const int mode = (int)(i & 7);
        if (mode == 1) mb &= 0xFF;            // divisor significand just above 1
        if (mode == 2) mb = ~(mb & 0xFF);     // ... just below 2
        if (mode == 3) ma &= 0xFF;
        if (mode == 4) ma = ~(ma & 0xFF);
        if (mode == 5) mb = ~0ULL;            // all ones
        if (mode == 6) { ma = mb; }           // quotient near a power of two
        const int ea = (int)(rnd() % 1801) - 900, eb = (int)(rnd() % 201) - 100;
        double a = mk(ma, ea, (int)(rnd() & 1));
        const double b = mk(mb, eb, (int)(rnd() & 1));
        if ((i & 1023) == 7) a = (rnd() & 1) ? 0.0 : -0.0;   // exact zeros
        const InvDivisor d = make_inv_divisor(b);
        if (!(d.ok && inv_div_numerator_ok0(a))) continue;
        checked++;
        const double q = inv_div_core(a, d), want = a / b;
        if (memcmp(&q, &want, 8) != 0) {
            if (bad < 5) printf("MISMATCH a=%a b=%a got=%a want=%a\n", a, b, q, want);
            bad++;
        }
    }
    printf("checked %ld\nmismatches %ld\n", checked, bad);

    // GR4J's own quotients drop the numerator's lower bound (gr4j_core.h
    // gr4j_num_ok): positive numerators of ANY magnitude below 2^196,
    // subnormals included.  Below 2^-900 the residual a - b q0 can underflow,
    // and the 3-FMA form is then only FAITHFUL: measured here in ulps of the
    // IEEE quotient (a subnormal quotient's ulp is 2^-1074).
    long tiny_checked = 0, tiny_off = 0;
    double tiny_worst = 0.0;
    for (long i = 0; i < n / 4; ++i) {
        const uint64_t ma = rnd(), mb = rnd();
        const int eb = (int)(rnd() % 201) - 100;
        const double b = mk(mb, eb, 0);
        double a;
        if (i & 1) {                     // positive subnormal
            uint64_t bits = ma >> (12 + rnd() % 52);
            if (!bits) bits = 1;
            memcpy(&a, &bits, 8);
        } else {                         // tiny normal, 2^-1022 .. 2^-850
            a = mk(ma, -1022 + (int)(rnd() % 172), 0);
        }
        const InvDivisor d = make_inv_divisor(b);
        if (!d.ok) continue;
        tiny_checked++;
        const double q = inv_div_core(a, d), want = a / b;
        int e;
        frexp(want, &e);
        const double ulp = (want < 0x1p-1022) ? 0x1p-1074 : ldexp(1.0, e - 53);
        const double diff = fabs(q - want) / ulp;
        if (diff > tiny_worst) tiny_worst = diff;
        if (q != want) tiny_off++;
    }
    printf("tiny_checked %ld\ntiny_worst_ulps %.0f\ntiny_not_rn %ld\n",
           tiny_checked, tiny_worst, tiny_off);

    // The faithful form a * RN(1/b) (inv_mul_core: HBV-Edu's and GR4J's own
    // quotients) serves EVERY numerator once the divisor is ok: error in
    // ulps of the IEEE quotient over the whole exponent range (quotients
    // that overflow or are subnormal measured on their own grid), and the
    // special numerators must come out exactly as the division gives them.
    long f_checked = 0, f_special_bad = 0;
    double f_worst = 0.0;
    for (long i = 0; i < n / 2; ++i) {
        const uint64_t ma = rnd(), mb = rnd();
        const int ea = (int)(rnd() % 2045) - 1022, eb = (int)(rnd() % 201) - 100;
        const double a = mk(ma, ea, (int)(rnd() & 1));
        const double b = mk(mb, eb, (int)(rnd() & 1));
        const InvDivisor d = make_inv_divisor(b);
        if (!d.ok) continue;
        const double q = inv_mul_core(a, d), want = a / b;
        if (std::isinf(want) || std::isinf(q)) {
            // at the very top of the range the product may round to inf one
            // ulp before the quotient does (or after): same magnitude class
            if (!(fabs(want) >= 0x1.ffffffffffff0p1023 &&
                  fabs(q) >= 0x1.ffffffffffff0p1023)) f_special_bad++;
            continue;
        }
        f_checked++;
        int e;
        frexp(want, &e);
        const double ulp = (fabs(want) < 0x1p-1022) ? 0x1p-1074
                                                    : ldexp(1.0, e - 53);
        const double diff = fabs(q - want) / ulp;
        if (diff > f_worst) f_worst = diff;
    }
    {
        const double specials[] = {0.0, -0.0, INFINITY, -INFINITY, NAN,
                                   0x1p-1074, -0x1p-1074, 0x1.fffffffffffffp1023};
        const double divisors[] = {1.0, -3.0, 0x1p-100, 0x1p100, 350.0, -0.7};
        for (double b : divisors) {
            const InvDivisor d = make_inv_divisor(b);
            for (double a : specials) {
                const double q = inv_mul_core(a, d), want = a / b;
                const bool same = (std::isnan(q) && std::isnan(want)) ||
                                  (memcmp(&q, &want, 8) == 0) ||
                                  (std::isfinite(want) && want != 0.0 &&
                                   fabs(q - want) <= 2 * fabs(want) * 0x1p-52) ||
                                  (fabs(want) < 0x1p-1022 &&
                                   fabs(q - want) <= 0x1p-1074);
                if (!same || !d.ok) {
                    printf("SPECIAL a=%a b=%a got=%a want=%a\n", a, b, q, want);
                    f_special_bad++;
                }
            }
        }
    }
    printf("faithful_checked %ld\nfaithful_worst_ulps_x100 %.0f\n"
           "faithful_special_bad %ld\n", f_checked, f_worst * 100, f_special_bad);
    return bad != 0;
}
