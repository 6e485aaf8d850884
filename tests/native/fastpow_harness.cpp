// CPU accuracy harness for rrmpg_amd/csrc/fastpow.h (built and run by
// tests/test_fastpow_cpu.py): compares fastpow_core with 80-bit powl on random
// arguments and prints the worst error in ulp of the double result.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../rrmpg_amd/csrc/fastpow.h"

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
    // set 1: the HBV-Edu box  x = soil/FC in (0.02, 2.5), y = Beta in (0.5, 8)
    // set 2: wide             x = 2^U(-60,60), y = U(-12, 12)
    // set 3: near 1           x = 1 + U(-1e-3,1e-3), y = U(-300, 300)
    double worst[3] = {0, 0, 0};
    double wx[3] = {0, 0, 0}, wy[3] = {0, 0, 0};
    long bad_guard = 0;
    for (int set = 0; set < 3; ++set)
        for (long i = 0; i < n; ++i) {
            double x, y;
            if (set == 0) { x = 0.02 + 2.48 * u01(); y = 0.5 + 7.5 * u01(); }
            else if (set == 1) { x = exp2(-60 + 120 * u01()); y = -12 + 24 * u01(); }
            else { x = 1 + 2e-3 * (u01() - 0.5); y = -300 + 600 * u01(); }
            double z;
            double got = fastpow_core(x, y, &z);
            if (!fastpow_ok(x, z)) { bad_guard++; continue; }
            double err = ulp_err(got, powl((long double)x, (long double)y));
            if (err > worst[set]) { worst[set] = err; wx[set] = x; wy[set] = y; }
        }
    // exact cases
    double z;
    int exact_ok = fastpow_core(1.0, 3.7, &z) == 1.0 &&
                   fastpow_core(2.5, 0.0, &z) == 1.0 &&
                   fastpow_core(2.0, 3.0, &z) == 8.0 &&
                   fastpow_core(0.25, 0.5, &z) == 0.5 &&
                   fastpow_core(4.0, -1.0, &z) == 0.25;
    printf("worst_ulp_hbv %.4f at x=%.17g y=%.17g\n", worst[0], wx[0], wy[0]);
    printf("worst_ulp_wide %.4f at x=%.17g y=%.17g\n", worst[1], wx[1], wy[1]);
    printf("worst_ulp_near1 %.4f at x=%.17g y=%.17g\n", worst[2], wx[2], wy[2]);
    printf("guard_rejected %ld\nexact_ok %d\n", bad_guard, exact_ok);
    // libm's own pow for scale
    double wl = 0; s = 88172645463325252ULL;
    for (long i = 0; i < n; ++i) {
        double x = 0.02 + 2.48 * u01(), y = 0.5 + 7.5 * u01();
        double err = ulp_err(pow(x, y), powl((long double)x, (long double)y));
        if (err > wl) wl = err;
    }
    printf("libm_pow_worst_ulp_hbv %.4f\n", wl);
    return 0;
}
