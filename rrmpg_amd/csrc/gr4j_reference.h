// gr4j_reference.h -- the reference's own GR4J day, for the parameter sets the
// fast forms are not meant for.
//
// The fast forms of the GR4J family (gr4j_core.h) are restatements of the
// reference's statements that agree with them to a few ulp AS LONG AS
// EVERYTHING IS A NUMBER: a product that overflows on its own is still a
// number inside an FMA, x - x (1 - y) is x y unless x is infinite, a store of
// x3 = 1e-200 mm runs away through cancellations that another grouping
// rounds differently.  So only CIVIL sets keep the fast kernels' results:
// capacities x1, x3 within [1e-2, 1e6] mm, the exchange coefficient within
// +-1e3, initial fillings in [0, 1], forcing that is a number of at most 1e6
// (x4 only sets the hydrographs' lengths and is not part of it; in the coupled
// models the snow routine's parameters have to be numbers of at most 1e6 as
// well, or its outflow -- GR4J's precipitation -- need not be).  For every
// other set -- zeros, negatives, subnormals, 1e+-200, infinities, NaN -- a
// second kernel, launched right behind the fast ones, runs the reference's
// own sequence (gr4j_model.py:60-157) statement by statement -- IEEE
// quotients, tanh and pow of the math library, separate multiply and add
// (the files are built -ffp-contract=off) -- and OVERWRITES that set's
// columns and score: whatever the reference does with infinities and NaN,
// day by day, this does too (tests/test_gpu_fuzz.py compares wild sets' NaN
// / inf pattern with the CPU restatement's over the whole series).  Which
// result a set gets depends on ITS parameters (and the launch's forcing and
// initial states) only; the fast kernels are untouched -- they carry nothing
// of this path.  One lane per set, the hydrographs in private memory: unit
// hydrographs of more than 20 days (x4 > 20) keep the fast kernels' result.
#pragma once

#include <math.h>

#include "common.h"

#define GR4J_CIVIL 1e6
#define GR4J_REF_N1 20
#define GR4J_REF_N2 (2 * GR4J_REF_N1 + 1)

// (NaN fails every comparison)
__device__ __forceinline__ bool gr4j_civil_set(double x1, double x2, double x3,
                                               double s_init, double r_init)
{
    return x1 >= 1e-2 && x1 <= GR4J_CIVIL && fabs(x2) <= 1e3 && x3 >= 1e-2 &&
           x3 <= GR4J_CIVIL && s_init >= 0.0 && s_init <= 1.0 &&
           r_init >= 0.0 && r_init <= 1.0;
}
// a snow-routine parameter (CTG, Kf, Thacc, Rsp, DDF) under which the routine
// hands GR4J a liquid-water series that is civil forcing: a number of at most
// 1e6 (the coupled kernels' sets are civil only with these)
__device__ __forceinline__ bool gr4j_civil_snow_par(double v)
{
    return fabs(v) <= GR4J_CIVIL;
}
// a forcing value the fast forms are meant for
__device__ __forceinline__ bool gr4j_civil_forcing(double v)
{
    return fabs(v) <= GR4J_CIVIL;
}

// gr4j_model.py:159-173, :176-192
__device__ __forceinline__ double gr4j_ref_s_curve1(int t, double x4)
{
    if (t <= 0) return 0.0;
    else if ((double)t < x4) return pow((double)t / x4, 2.5);
    else return 1.0;
}
__device__ __forceinline__ double gr4j_ref_s_curve2(int t, double x4)
{
    if (t <= 0) return 0.0;
    else if ((double)t <= x4) return 0.5 * pow((double)t / x4, 2.5);
    else if ((double)t < 2 * x4)
        return 1 - 0.5 * pow(2 - (double)t / x4, 2.5);
    else return 1.0;
}

struct Gr4jRef {
    double x1, x2, x3, s, r;
    int n1, n2;
    double o1[GR4J_REF_N1], o2[GR4J_REF_N2], u1[GR4J_REF_N1], u2[GR4J_REF_N2];

    // false: hydrographs this path does not keep (or none at all: the
    // launch's plan reports those)
    __device__ bool init(double x1_, double x2_, double x3_, double x4,
                         double s_init, double r_init)
    {
        x1 = x1_; x2 = x2_; x3 = x3_;
        const double c1 = ceil(x4), c2 = ceil(2 * x4 + 1);   // :68-69
        if (!(c1 >= 1.0) || !(c1 <= GR4J_REF_N1) || !(c2 >= 1.0) ||
            !(c2 <= GR4J_REF_N2))
            return false;
        n1 = (int)c1;
        n2 = (int)c2;
        for (int j = 1; j <= n1; ++j)                         // :75-76
            o1[j - 1] = gr4j_ref_s_curve1(j, x4) - gr4j_ref_s_curve1(j - 1, x4);
        for (int j = 1; j <= n2; ++j)                         // :78-79
            o2[j - 1] = gr4j_ref_s_curve2(j, x4) - gr4j_ref_s_curve2(j - 1, x4);
        for (int j = 0; j < n1; ++j) u1[j] = 0.0;
        for (int j = 0; j < n2; ++j) u2[j] = 0.0;
        s = s_init * x1;                                      // :64
        r = r_init * x3;                                      // :65
        return true;
    }

    // one day (:86-154); returns the discharge, s and r are the day's stores
    __device__ double day(double prec, double etp)
    {
        double p_n, p_s, e_s;
        if (prec >= etp) {                                    // :89-99
            p_n = prec - etp;
            const double sx = s / x1;
            const double th = tanh(p_n / x1);
            p_s = (x1 * (1 - sx * sx) * th) / (1 + sx * th);
            e_s = 0.0;
        } else {                                              // :101-111
            p_n = 0.0;
            const double pe_n = etp - prec;
            const double sx = s / x1;
            const double th = tanh(pe_n / x1);
            e_s = (s * (2 - sx) * th) / (1 + (1 - sx) * th);
            p_s = 0.0;
        }
        double sn = s - e_s + p_s;                            // :114
        const double v = 4.0 / 9.0 * sn / x1;                 // :117
        const double v2 = v * v;
        const double perc = sn * (1 - pow(1 + v2 * v2, -0.25));
        sn = sn - perc;                                       // :120
        const double p_r = perc + (p_n - p_s);                // :123
        const double p_r_uh1 = 0.9 * p_r;                     // :126-127
        const double p_r_uh2 = 0.1 * p_r;
        for (int j = 0; j < n1 - 1; ++j)                      // :130-132
            u1[j] = u1[j + 1] + o1[j] * p_r_uh1;
        u1[n1 - 1] = o1[n1 - 1] * p_r_uh1;
        for (int j = 0; j < n2 - 1; ++j)                      // :134-136
            u2[j] = u2[j + 1] + o2[j] * p_r_uh2;
        u2[n2 - 1] = o2[n2 - 1] * p_r_uh2;
        const double gw_exchange = x2 * pow(r / x3, 3.5);     // :139
        double rn = nb_max(0.0, r + u1[0] + gw_exchange);     // :142
        const double w = rn / x3;                             // :145
        const double w2 = w * w;
        const double q_r = rn * (1 - pow(1 + w2 * w2, -0.25));
        rn = rn - q_r;                                        // :148
        const double q_d = nb_max(0.0, u2[0] + gw_exchange);  // :151
        s = sn;
        r = rn;
        return q_r + q_d;                                     // :154
    }
};
