// CPU accuracy harness for rrmpg_amd/csrc/fastmath.h (built and run by
// tests/test_fastmath_cpu.py): compares every fast form with 80-bit long-double
// libm on random arguments and prints the worst errors.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../rrmpg_amd/csrc/fastmath.h"

static uint64_t s = 88172645463325252ULL;
static double u01() {          // xorshift64*, uniform in [0,1)
    s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
    return (double)((s * 2685821237005ULL) >> 11) * (1.0 / 9007199254740992.0);
}

static double ulp_err(double got, long double want) {
    double w = (double)want;
    int e; frexp(w, &e);
    long double ulp = ldexpl(1.0L, e - 53);
    return (double)(fabsl((long double)got - want) / ulp);
}

int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 2000000;
    // fastpow_soil (HBV-Edu's default since round 5): (soil / FC) ** Beta
    // from the soil alone, against powl of the EXACT quotient, relative
    // error in units of 2^-53 -- (a) a sane run's box: FC 50..1000 mm, soil
    // = FC U(0.05, 1.5), Beta 0.5..8; (b) the whole guard box: FC =
    // 10^U(-3, 6), soil = FC 2^U(-9, 9), Beta U(-64, 64) -- and the stated
    // bound (6 + 3 |zz| + |y| (1 + 3 |log2 FC| + 3 |log2 soil|)) 2^-53
    {
        static const FpSoilEntry tabl[FP_SOIL_LOG_N] = FP_SOIL_LOG_TABLE_INIT;
        static const double tabe[FP_SOIL_EXP_N] = FP_SOIL_EXP_TABLE_INIT;
        double sw[2] = {0, 0}, sbound = 0;
        long srej = 0;
        s = 88172645463325252ULL;
        for (int set = 0; set < 2; ++set)
            for (long i = 0; i < n; ++i) {
                double fc, x, y;
                if (set == 0) {
                    fc = 50 + 950 * u01(); x = fc * (0.05 + 1.45 * u01());
                    y = 0.5 + 7.5 * u01();
                } else {
                    fc = pow(10.0, -3 + 9 * u01());
                    x = fc * exp2(-9 + 18 * u01()); y = -64 + 128 * u01();
                }
                double y2N, cF, sN;
                fastpow_soil_exponent(y, fc, tabl, &y2N, &cF);
                const double got = fastpow_soil(x, y2N, cF, tabl, tabe, &sN);
                if (!fastpow_soil_ok(x, sN)) { srej++; continue; }
                const long double want =
                    powl((long double)x / (long double)fc, (long double)y);
                const double rel = (double)(fabsl((long double)got - want) /
                                            want) * 0x1p53;
                if (rel > sw[set]) sw[set] = rel;
                const double zz = sN / FP_SOIL_EXP_N;
                const double over = rel / (6 + 3 * fabs(zz) + fabs(y) *
                    (1 + 3 * fabs(log2(fc)) + 3 * fabs(log2(x))));
                if (over > sbound) sbound = over;
            }
        // special arguments: the guard refuses what the split cannot serve,
        // and an FC that is no positive normal number poisons cF
        double y2N, cF, sN;
        fastpow_soil_exponent(2.0, 100.0, tabl, &y2N, &cF);
        int sok = fabs(fastpow_soil(100.0, y2N, cF, tabl, tabe, &sN) - 1.0)
                      < 1e-14 &&
                  fabs(fastpow_soil(50.0, y2N, cF, tabl, tabe, &sN) - 0.25)
                      < 1e-14 &&
                  !fastpow_soil_ok(0.0, 0.0) && !fastpow_soil_ok(-1.0, 0.0) &&
                  !fastpow_soil_ok(4e-320, 0.0) && !fastpow_soil_ok(NAN, 0.0) &&
                  !fastpow_soil_ok(INFINITY, 0.0) && !fastpow_soil_ok(1.0, NAN) &&
                  !fastpow_soil_ok(1.0, 3e5) && fastpow_soil_ok(0x1p-1022, 3.0);
        fastpow_soil_exponent(2.0, 0.0, tabl, &y2N, &cF);
        sok = sok && std::isnan(cF);
        fastpow_soil_exponent(2.0, -5.0, tabl, &y2N, &cF);
        sok = sok && std::isnan(cF);
        fastpow_soil_exponent(2.0, INFINITY, tabl, &y2N, &cF);
        sok = sok && std::isnan(cF);
        fastpow_soil_exponent(2.0, NAN, tabl, &y2N, &cF);
        sok = sok && std::isnan(cF);
        (void)fastpow_soil(1.0, 1.0, NAN, tabl, tabe, &sN);
        sok = sok && !fastpow_soil_ok(1.0, sN);
        printf("soil_worst_rel53_sane %.2f\nsoil_worst_rel53_box %.2f\n"
               "soil_worst_over_bound_x100 %.0f\nsoil_guard_rejected %ld\n"
               "soil_special_ok %d\n", sw[0], sw[1], sbound * 100, srej, sok);
    }

    // tanh: arguments as GR4J produces them (net / x1 in [0, ~1]) and wide
    double wt = 0, wtx = 0, wtw = 0, wtwx = 0;
    for (long i = 0; i < n; ++i) {
        double a = 1.2 * u01() * u01();
        double err = ulp_err(fast_tanh(a), tanhl((long double)a));
        if (err > wt) { wt = err; wtx = a; }
        a = exp2(-40 + 46 * u01()) * (u01() < 0.5 ? -1 : 1);
        err = ulp_err(fast_tanh(a), tanhl((long double)a));
        if (err > wtw) { wtw = err; wtwx = a; }
    }
    printf("worst_ulp_tanh_gr4j %.4f at a=%.17g\n", wt, wtx);
    printf("worst_ulp_tanh_wide %.4f at a=%.17g\n", wtw, wtwx);
    int tanh_special = fast_tanh(0.0) == 0.0 && signbit(fast_tanh(-0.0)) &&
                       fast_tanh(INFINITY) == 1.0 && fast_tanh(-INFINITY) == -1.0 &&
                       fast_tanh(25.0) == 1.0 && fast_tanh(-1e300) == -1.0 &&
                       std::isnan(fast_tanh(NAN)) &&
                       fast_tanh(1e-300) == 1e-300 && fast_tanh(-4e-320) == -4e-320;
    printf("tanh_special_ok %d\n", tanh_special);
    // the [9/8] Pade pair (fast_tanh_rational_parts, GR4J's default inside
    // |a| <= 1): num / den against tanhl over its whole range, both signs;
    // zeros keep their sign, tiny arguments come back unchanged
    {
        double wr = 0, wrx = 0;
        for (long i = 0; i < n; ++i) {
            const double a = FP_TANHR_AMAX * u01() * (u01() < 0.5 ? -1 : 1);
            double nu, de;
            fast_tanh_rational_parts(a, nu, de);
            const double err = ulp_err(nu / de, tanhl((long double)a));
            if (err > wr) { wr = err; wrx = a; }
        }
        double nu, de;
        fast_tanh_rational_parts(0.0, nu, de);
        int ok = nu == 0.0 && !signbit(nu) && de == 1.0;
        fast_tanh_rational_parts(-0.0, nu, de);
        ok = ok && nu == 0.0 && signbit(nu);
        fast_tanh_rational_parts(1e-300, nu, de);
        ok = ok && nu / de == 1e-300;
        fast_tanh_rational_parts(NAN, nu, de);
        ok = ok && std::isnan(nu / de);
        printf("worst_ulp_tanh_rational %.4f at a=%.17g\n", wr, wrx);
        printf("tanh_rational_special_ok %d\n", ok);
    }

    // fast_sqrt_core on its domain
    double ws = 0, wsx = 0;
    for (long i = 0; i < n; ++i) {
        double x = (i & 1) ? 4.0 * u01() + 1e-9 : exp2(-499 + 1522 * u01());
        double err = ulp_err(fast_sqrt_core(x), sqrtl((long double)x));
        if (err > ws) { ws = err; wsx = x; }
    }
    printf("worst_ulp_fast_sqrt %.4f at x=%.17g\n", ws, wsx);

    // b**(-1/4), b = 1 + v^4 >= 1
    double wr = 0, wrx = 0;
    for (long i = 0; i < n; ++i) {
        double v = (i & 1) ? 3.0 * u01() : exp2(-30 + 60 * u01());
        double b = 1 + (v * v) * (v * v);
        double err = ulp_err(inv_fourth_root(b), powl((long double)b, -0.25L));
        if (err > wr) { wr = err; wrx = b; }
    }
    printf("worst_ulp_inv_fourth_root %.4f at b=%.17g\n", wr, wrx);
    int r4_special = inv_fourth_root(1.0) == 1.0 && inv_fourth_root(16.0) == 0.5 &&
                     inv_fourth_root(INFINITY) < 1e-70 && inv_fourth_root(1e308) < 1e-70 &&
                     std::isnan(inv_fourth_root(NAN));
    printf("r4_special_ok %d\n", r4_special);

    // the one-step third-order variant the GR4J kernels use (finite b >= 1)
    double wr3 = 0, wr3x = 0;
    for (long i = 0; i < n; ++i) {
        double v = (i & 1) ? 3.0 * u01() : exp2(-30 + 60 * u01());
        double b = 1 + (v * v) * (v * v);
        double err = ulp_err(inv_fourth_root_core3(b),
                             powl((long double)b, -0.25L));
        if (err > wr3) { wr3 = err; wr3x = b; }
    }
    printf("worst_ulp_inv_fourth_root3 %.4f at b=%.17g\n", wr3, wr3x);
    printf("r4_3_exact_ok %d\n", inv_fourth_root_core3(1.0) == 1.0 &&
                                  inv_fourth_root_core3(16.0) == 0.5);

    // the polynomial root of b = 1 + v^4 (rounded, as the reference has it),
    // evaluated at u = b - 1 <= FP_R4_UMAX (GR4J's percolation)
    double wrp = 0, wrpx = 0;
    for (long i = 0; i < n; ++i) {
        double v = (i & 1) ? 0.4516 * u01() : exp2(-40 + 38.85 * u01());
        double b = 1 + (v * v) * (v * v);
        double u = b - 1;                  // exact; what the kernels pass
        if (u > FP_R4_UMAX) continue;
        double err = ulp_err(inv_fourth_root_1p_small<0>(u),
                             powl((long double)b, -0.25L));
        if (err > wrp) { wrp = err; wrpx = u; }
    }
    printf("worst_ulp_inv_fourth_root_poly %.4f at u=%.17g\n", wrp, wrpx);
    printf("r4_poly_ends_ok %d\n",
           inv_fourth_root_1p_small<0>(0.0) == 1.0 &&
           fabs(inv_fourth_root_1p_small<0>(FP_R4_UMAX) -
                pow(1 + FP_R4_UMAX, -0.25)) < 2.3e-16);

    // fast_div_core: denominators as the folded store update produces them
    // (D + k E in [1, 1e18]) and wide, numerators of either sign
    double wd = 0, wdn = 0, wdd = 0;
    for (long i = 0; i < n; ++i) {
        double d = (i & 1) ? 1.0 + 3.0 * u01() : exp2(200 * u01());
        if (i % 7 == 0) d = -d;
        double nn = exp2(-300 + 900 * u01()) * (u01() < 0.5 ? -1 : 1);
        if (i % 5 == 0) nn = 1500.0 * u01();
        double err = ulp_err(fast_div_core(nn, d),
                             (long double)nn / (long double)d);
        if (err > wd) { wd = err; wdn = nn; wdd = d; }
    }
    printf("worst_ulp_fast_div %.4f at n=%.17g d=%.17g\n", wd, wdn, wdd);
    printf("fast_div_exact_ok %d\n", fast_div_core(0.0, 3.0) == 0.0 &&
           fast_div_core(6.0, 3.0) == 2.0 && fast_div_core(-1.0, 4.0) == -0.25 &&
           std::isnan(fast_div_core(NAN, 2.0)));
    return 0;
}
