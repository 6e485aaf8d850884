/*
 * rr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, scalar, CPU restatement of the reference's hot path: the @njit
 * per-timestep loops of kratzert/RRMPG.  It exists so the HIP kernels can be
 * checked on a GPU box where the Python reference cannot travel.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it; nothing under rrmpg_amd/ links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against (a) the reference's own known-answer data (MATLAB / Excel
 * outputs held by the reference's unit tests) and (b) outputs of the
 * reference's own source executed in the build container
 * (tests/golden/gen_golden.py), to 1e-12 relative.
 *
 * Numerics contract followed (what numba generates for the reference):
 *   - fp64 everywhere, no FMA contraction (build with -ffp-contract=off),
 *     Python left-to-right evaluation order;
 *   - x**2 / x**4 with a literal int exponent = repeated squaring, x*x and
 *     (x*x)*(x*x) (numba/cpython/numbers.py static_power_impl);
 *   - float exponents and np.tanh -> the platform libm pow / tanh;
 *   - max(a,b) = (b > a) ? b : a, min(a,b) = (b < a) ? b : a
 *     (numba/cpython/builtins.py do_minmax) -- so max(0, NaN) == 0;
 *   - np.mean = sequential left-to-right sum / size (numba/np/arraymath.py).
 *
 * Each function cites the reference file:line it restates (paths relative to
 * the reference checkout).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline double nb_max(double a, double b) { return (b > a) ? b : a; }
static inline double nb_min(double a, double b) { return (b < a) ? b : a; }

/* ------------------------------------------------------------------ ABC
 * reference: rrmpg/models/abcmodel_model.py:15-60
 * params = {a, b, c}; the loop starts at t = 1 (qsim[0] = 0,
 * storage[0] = initial_state, prec[0] unused). */
void oracle_run_abcmodel(const double *prec, int64_t T, double initial_state,
                         const double *params, double *qsim, double *storage)
{
    const double a = params[0], b = params[1], c = params[2];
    if (T <= 0) return;
    qsim[0] = 0.0;
    storage[0] = initial_state;
    for (int64_t t = 1; t < T; ++t) {
        /* abcmodel_model.py:56 */
        qsim[t] = (1 - a - b) * prec[t] + c * storage[t - 1];
        /* abcmodel_model.py:59 */
        storage[t] = (1 - c) * storage[t - 1] + a * prec[t];
    }
}

/* -------------------------------------------------------------- HBV-Edu
 * reference: rrmpg/models/hbvedu_model.py:15-129
 * params = {T_t, DD, FC, Beta, C, PWP, K_0, K_1, K_2, K_p, L};
 * month holds 0..11 (already decremented by the caller, hbvedu.py:164). */
void oracle_run_hbvedu(const double *temp, const double *prec,
                       const int8_t *month, const double *PE_m,
                       const double *T_m, int64_t T, double snow_init,
                       double soil_init, double s1_init, double s2_init,
                       const double *params, double *qsim, double *snow,
                       double *soil, double *s1, double *s2)
{
    const double T_t = params[0], DD = params[1], FC = params[2],
                 Beta = params[3], C = params[4], PWP = params[5],
                 K_0 = params[6], K_1 = params[7], K_2 = params[8],
                 K_p = params[9], L = params[10];
    if (T <= 0) return;
    qsim[0] = 0.0;                      /* hbvedu_model.py:71-81 */
    snow[0] = snow_init;
    soil[0] = soil_init;
    s1[0] = s1_init;
    s2[0] = s2_init;
    for (int64_t t = 1; t < T; ++t) {
        double liquid_water;
        if (temp[t] < T_t) {            /* :87-91 */
            snow[t] = snow[t - 1] + prec[t];
            liquid_water = 0.0;
        } else {                        /* :92-96 */
            const double melt = DD * (temp[t] - T_t);
            snow[t] = nb_max(0.0, snow[t - 1] - melt);
            liquid_water = prec[t] + nb_min(snow[t - 1], melt);
        }
        /* :99 */
        const double prec_eff = liquid_water * pow(soil[t - 1] / FC, Beta);
        /* :102 */
        const int m = month[t];
        const double pe = (1 + C * (temp[t] - T_m[m])) * PE_m[m];
        /* :105-108 */
        double ea;
        if (soil[t - 1] > PWP) ea = pe;
        else ea = pe * (soil[t - 1] / PWP);
        /* :111 */
        soil[t] = soil[t - 1] + liquid_water - prec_eff - ea;
        /* :114-118 */
        s1[t] = s1[t - 1] + prec_eff - nb_max(0.0, s1[t - 1] - L) * K_0
                - s1[t - 1] * K_1 - s1[t - 1] * K_p;
        /* :121-123 */
        s2[t] = s2[t - 1] + s1[t - 1] * K_p - s2[t - 1] * K_2;
        /* :125-127 */
        qsim[t] = nb_max(0.0, s1[t - 1] - L) * K_0 + s1[t] * K_1
                  + s2[t] * K_2;
    }
}

/* ----------------------------------------------------------------- GR4J
 * reference: rrmpg/models/gr4j_model.py:159-173 (_s_curve1) */
static double s_curve1(int64_t t, double x4)
{
    if (t <= 0) return 0.0;
    else if ((double)t < x4) return pow((double)t / x4, 2.5);
    else return 1.0;
}

/* reference: rrmpg/models/gr4j_model.py:176-192 (_s_curve2) */
static double s_curve2(int64_t t, double x4)
{
    if (t <= 0) return 0.0;
    else if ((double)t <= x4) return 0.5 * pow((double)t / x4, 2.5);
    else if ((double)t < 2 * x4)
        return 1 - 0.5 * pow(2 - (double)t / x4, 2.5);
    else return 1.0;
}

/* Number of unit-hydrograph ordinates, gr4j_model.py:68-69. */
void oracle_gr4j_num_uh(double x4, int64_t *num_uh1, int64_t *num_uh2)
{
    *num_uh1 = (int64_t)ceil(x4);
    *num_uh2 = (int64_t)ceil(2 * x4 + 1);
}

/* reference: rrmpg/models/gr4j_model.py:15-157
 * params = {x1, x2, x3, x4}; out[k] is the state AFTER day k (the reference
 * prepends an artificial step 0 and drops it again, :60-61, :157).
 * Returns 0, or -1 if x4 gives an empty unit hydrograph (the reference
 * raises IndexError there under CPython). */
int oracle_run_gr4j(const double *prec, const double *etp, int64_t T,
                    double s_init, double r_init, const double *params,
                    double *qsim, double *s_store, double *r_store)
{
    const double x1 = params[0], x2 = params[1], x3 = params[2],
                 x4 = params[3];
    int64_t n1, n2;
    oracle_gr4j_num_uh(x4, &n1, &n2);
    if (!(n1 >= 1) || !(n2 >= 1) || n1 > (1 << 20) || n2 > (1 << 21))
        return -1;
    double *o1 = (double *)calloc((size_t)n1, sizeof(double));
    double *o2 = (double *)calloc((size_t)n2, sizeof(double));
    double *uh1 = (double *)calloc((size_t)n1, sizeof(double));
    double *uh2 = (double *)calloc((size_t)n2, sizeof(double));
    for (int64_t j = 1; j <= n1; ++j)   /* :75-76 */
        o1[j - 1] = s_curve1(j, x4) - s_curve1(j - 1, x4);
    for (int64_t j = 1; j <= n2; ++j)   /* :78-79 */
        o2[j - 1] = s_curve2(j, x4) - s_curve2(j - 1, x4);

    double s = s_init * x1;             /* :64 */
    double r = r_init * x3;             /* :65 */
    for (int64_t k = 0; k < T; ++k) {   /* reference t = k + 1 */
        double p_n, p_s, e_s;
        if (prec[k] >= etp[k]) {        /* :89-99 */
            p_n = prec[k] - etp[k];
            const double sx = s / x1;
            const double th = tanh(p_n / x1);
            p_s = (x1 * (1 - sx * sx) * th) / (1 + sx * th);
            e_s = 0.0;
        } else {                        /* :101-111 */
            p_n = 0.0;
            const double pe_n = etp[k] - prec[k];
            const double sx = s / x1;
            const double th = tanh(pe_n / x1);
            e_s = (s * (2 - sx) * th) / (1 + (1 - sx) * th);
            p_s = 0.0;
        }
        double sn = s - e_s + p_s;      /* :114 */
        /* :117  (4/9 * S / x1)**4 by repeated squaring */
        const double v = 4.0 / 9.0 * sn / x1;
        const double v2 = v * v;
        const double perc = sn * (1 - pow(1 + v2 * v2, -0.25));
        sn = sn - perc;                 /* :120 */
        const double p_r = perc + (p_n - p_s);      /* :123 */
        const double p_r_uh1 = 0.9 * p_r;           /* :126-127 */
        const double p_r_uh2 = 0.1 * p_r;
        for (int64_t j = 0; j < n1 - 1; ++j)        /* :130-132 */
            uh1[j] = uh1[j + 1] + o1[j] * p_r_uh1;
        uh1[n1 - 1] = o1[n1 - 1] * p_r_uh1;
        for (int64_t j = 0; j < n2 - 1; ++j)        /* :134-136 */
            uh2[j] = uh2[j + 1] + o2[j] * p_r_uh2;
        uh2[n2 - 1] = o2[n2 - 1] * p_r_uh2;
        /* :139 */
        const double gw_exchange = x2 * pow(r / x3, 3.5);
        /* :142 */
        double rn = nb_max(0.0, r + uh1[0] + gw_exchange);
        /* :145 */
        const double w = rn / x3;
        const double w2 = w * w;
        const double q_r = rn * (1 - pow(1 + w2 * w2, -0.25));
        rn = rn - q_r;                  /* :148 */
        /* :151 */
        const double q_d = nb_max(0.0, uh2[0] + gw_exchange);
        qsim[k] = q_r + q_d;            /* :154 */
        s_store[k] = sn;
        r_store[k] = rn;
        s = sn;
        r = rn;
    }
    free(o1); free(o2); free(uh1); free(uh2);
    return 0;
}

/* ------------------------------------------------------------ Cemaneige
 * reference: rrmpg/models/cemaneige_model.py:15-126
 * prec, mean_temp, frac_solid_prec: [T][L] row-major; params = {CTG, Kf};
 * G, eTG: [T][L]; outflow: [T]. */
void oracle_run_cemaneige(const double *prec, const double *mean_temp,
                          const double *frac_solid_prec, int64_t T, int64_t L,
                          double snow_pack_init, double thermal_state_init,
                          const double *params, double *outflow, double *G,
                          double *eTG)
{
    const double CTG = params[0], Kf = params[1];
    if (T <= 0 || L <= 0) return;
    double *liquid_water = (double *)calloc((size_t)(T * L), sizeof(double));
    double *snow = (double *)malloc((size_t)T * sizeof(double));
    double *rain = (double *)malloc((size_t)T * sizeof(double));
    for (int64_t l = 0; l < L; ++l) {   /* :73 */
        double c = 0.0;
        for (int64_t t = 0; t < T; ++t) {           /* :76-77 */
            snow[t] = prec[t * L + l] * frac_solid_prec[t * L + l];
            rain[t] = prec[t * L + l] - snow[t];
            c += snow[t];               /* np.mean: sequential sum */
        }
        const double G_tresh = 0.9 * 365.25 * (c / (double)T);  /* :80 */
        for (int64_t t = 0; t < T; ++t) {
            double g, e;
            if (t == 0) g = snow_pack_init;         /* :85-88 */
            else g = G[(t - 1) * L + l] + snow[t];
            if (t == 0) e = thermal_state_init;     /* :91-96 */
            else e = CTG * eTG[(t - 1) * L + l]
                     + (1 - CTG) * mean_temp[t * L + l];
            if (e > 0) e = 0.0;
            double pot_melt;                        /* :99-106 */
            if (e == 0 && mean_temp[t * L + l] > 0) {
                pot_melt = Kf * mean_temp[t * L + l];
                if (pot_melt > g) pot_melt = g;
            } else {
                pot_melt = 0.0;
            }
            double G_ratio;                         /* :109-112 */
            if (g < G_tresh) G_ratio = g / G_tresh;
            else G_ratio = 1.0;
            const double melt = (0.9 * G_ratio + 0.1) * pot_melt;  /* :115 */
            g = g - melt;                           /* :118 */
            G[t * L + l] = g;
            eTG[t * L + l] = e;
            liquid_water[t * L + l] = rain[t] + melt;   /* :121 */
        }
    }
    for (int64_t t = 0; t < T; ++t) {   /* :124-125, sequential mean */
        double c = 0.0;
        for (int64_t l = 0; l < L; ++l) c += liquid_water[t * L + l];
        outflow[t] = c / (double)L;
    }
    free(liquid_water); free(snow); free(rain);
}

/* ------------------------------------------------- Cemaneige -> GR4J
 * reference: rrmpg/models/cemaneigegr4j_model.py:16-63
 * params = {CTG, Kf, x1, x2, x3, x4}: run_cemaneige reads the first two by
 * name, run_gr4j the last four. */
int oracle_run_cemaneigegr4j(const double *prec, const double *mean_temp,
                             const double *etp, const double *frac_solid_prec,
                             int64_t T, int64_t L, double snow_pack_init,
                             double thermal_state_init, double s_init,
                             double r_init, const double *params, double *qsim,
                             double *G, double *eTG, double *s_store,
                             double *r_store)
{
    if (T <= 0 || L <= 0) return 0;
    double *liquid_water = (double *)malloc((size_t)T * sizeof(double));
    oracle_run_cemaneige(prec, mean_temp, frac_solid_prec, T, L,
                         snow_pack_init, thermal_state_init, params,
                         liquid_water, G, eTG);             /* :57-59 */
    const int rc = oracle_run_gr4j(liquid_water, etp, T, s_init, r_init,
                                   params + 2, qsim, s_store, r_store); /* :62 */
    free(liquid_water);
    return rc;
}

/* ======================================================================
 * Reference-shaped sweeps: what Model.simulate() does around run_* -- one
 * call per parameter set, fresh zeroed [T] arrays per call, scatter into
 * column i of a [T][N] array (reference: hbvedu.py:190-214, abcmodel.py:
 * 168-186, gr4j.py:162-183, cemaneige.py:218-245, cemaneigegr4j.py:238-273).
 * nthreads = 1 is the single-threaded reference shape; nthreads > 1 spreads
 * the sets over host cores with OpenMP (used only for the "all host cores"
 * CPU baseline).  Storage outputs are nullable.  params: [N][k] row-major =
 * the reference's structured-dtype buffer.
 * ====================================================================== */

static void scatter(double *dst, int64_t ld, int64_t col, const double *src,
                    int64_t T)
{
    if (!dst) return;
    for (int64_t t = 0; t < T; ++t) dst[t * ld + col] = src[t];
}

#define SWEEP_THREADS(nthreads)                                           \
    int _nt = (nthreads);                                                 \
    if (_nt < 1) _nt = 1;

void oracle_simulate_abc(const double *prec, int64_t T, double initial_state,
                         const double *params, int64_t N, double *qsim,
                         double *storage, int nthreads)
{
    SWEEP_THREADS(nthreads)
#pragma omp parallel for num_threads(_nt) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        double *q = (double *)calloc((size_t)(T > 0 ? T : 1), sizeof(double));
        double *s = (double *)calloc((size_t)(T > 0 ? T : 1), sizeof(double));
        oracle_run_abcmodel(prec, T, initial_state, params + 3 * i, q, s);
        scatter(qsim, N, i, q, T);
        scatter(storage, N, i, s, T);
        free(q); free(s);
    }
}

void oracle_simulate_hbvedu(const double *temp, const double *prec,
                            const int8_t *month, const double *PE_m,
                            const double *T_m, int64_t T, double snow_init,
                            double soil_init, double s1_init, double s2_init,
                            const double *params, int64_t N, double *qsim,
                            double *snow, double *soil, double *s1, double *s2,
                            int nthreads)
{
    SWEEP_THREADS(nthreads)
#pragma omp parallel for num_threads(_nt) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const size_t n = (size_t)(T > 0 ? T : 1);
        double *a = (double *)calloc(5 * n, sizeof(double));
        oracle_run_hbvedu(temp, prec, month, PE_m, T_m, T, snow_init,
                          soil_init, s1_init, s2_init, params + 11 * i, a,
                          a + n, a + 2 * n, a + 3 * n, a + 4 * n);
        scatter(qsim, N, i, a, T);
        scatter(snow, N, i, a + n, T);
        scatter(soil, N, i, a + 2 * n, T);
        scatter(s1, N, i, a + 3 * n, T);
        scatter(s2, N, i, a + 4 * n, T);
        free(a);
    }
}

int oracle_simulate_gr4j(const double *prec, const double *etp, int64_t T,
                         double s_init, double r_init, const double *params,
                         int64_t N, double *qsim, double *s_store,
                         double *r_store, int nthreads)
{
    int rc = 0;
    SWEEP_THREADS(nthreads)
#pragma omp parallel for num_threads(_nt) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const size_t n = (size_t)(T > 0 ? T : 1);
        double *a = (double *)calloc(3 * n, sizeof(double));
        if (oracle_run_gr4j(prec, etp, T, s_init, r_init, params + 4 * i, a,
                            a + n, a + 2 * n) != 0) {
#pragma omp atomic write
            rc = -1;
        }
        scatter(qsim, N, i, a, T);
        scatter(s_store, N, i, a + n, T);
        scatter(r_store, N, i, a + 2 * n, T);
        free(a);
    }
    return rc;
}

/* G, eTG: [T][L][N] (reference: cemaneige.py:219-224) */
void oracle_simulate_cemaneige(const double *prec, const double *mean_temp,
                               const double *frac_solid_prec, int64_t T,
                               int64_t L, double snow_pack_init,
                               double thermal_state_init, const double *params,
                               int64_t N, double *outflow, double *G,
                               double *eTG, int nthreads)
{
    SWEEP_THREADS(nthreads)
#pragma omp parallel for num_threads(_nt) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const size_t n = (size_t)(T > 0 ? T : 1), nl = n * (size_t)(L > 0 ? L : 1);
        double *o = (double *)calloc(n, sizeof(double));
        double *g = (double *)calloc(nl, sizeof(double));
        double *e = (double *)calloc(nl, sizeof(double));
        oracle_run_cemaneige(prec, mean_temp, frac_solid_prec, T, L,
                             snow_pack_init, thermal_state_init,
                             params + 2 * i, o, g, e);
        scatter(outflow, N, i, o, T);
        scatter(G, N, i, g, T * L);
        scatter(eTG, N, i, e, T * L);
        free(o); free(g); free(e);
    }
}

int oracle_simulate_cemaneigegr4j(const double *prec, const double *mean_temp,
                                  const double *etp,
                                  const double *frac_solid_prec, int64_t T,
                                  int64_t L, double snow_pack_init,
                                  double thermal_state_init, double s_init,
                                  double r_init, const double *params,
                                  int64_t N, double *qsim, double *G,
                                  double *eTG, double *s_store,
                                  double *r_store, int nthreads)
{
    int rc = 0;
    SWEEP_THREADS(nthreads)
#pragma omp parallel for num_threads(_nt) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const size_t n = (size_t)(T > 0 ? T : 1), nl = n * (size_t)(L > 0 ? L : 1);
        double *a = (double *)calloc(3 * n, sizeof(double));
        double *g = (double *)calloc(nl, sizeof(double));
        double *e = (double *)calloc(nl, sizeof(double));
        if (oracle_run_cemaneigegr4j(prec, mean_temp, etp, frac_solid_prec, T,
                                     L, snow_pack_init, thermal_state_init,
                                     s_init, r_init, params + 6 * i, a, g, e,
                                     a + n, a + 2 * n) != 0) {
#pragma omp atomic write
            rc = -1;
        }
        scatter(qsim, N, i, a, T);
        scatter(G, N, i, g, T * L);
        scatter(eTG, N, i, e, T * L);
        scatter(s_store, N, i, a + n, T);
        scatter(r_store, N, i, a + 2 * n, T);
        free(a); free(g); free(e);
    }
    return rc;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ======================================================================
 * Next tier: hysteresis snow routine, ice melt
 * and their couplings with GR4J.
 * ====================================================================== */

/* reference: rrmpg/models/cemaneigehyst_model.py:5-166
 * params = {CTG, Kf, Thacc, Rsp} (read by name from the coupled record);
 * G, eTG, sca, rain: [T][L]; outflow: [T].  Quirk kept: at t = 0 the
 * accumulation branch reads sca[t-1] = sca[-1], the still-zero LAST row, so
 * sca_init never survives the first step (cemaneigehyst_model.py:96-98,126). */
void oracle_run_cemaneigehyst(const double *prec, const double *mean_temp,
                              const double *frac_solid_prec, int64_t T,
                              int64_t L, double snow_pack_init,
                              double thermal_state_init, double sca_init,
                              double CTG, double Kf, double Thacc, double Rsp,
                              double *outflow, double *G, double *eTG,
                              double *sca, double *rain)
{
    if (T <= 0 || L <= 0) return;
    double *liquid_water = (double *)calloc((size_t)(T * L), sizeof(double));
    double *snow = (double *)malloc((size_t)T * sizeof(double));
    memset(sca, 0, (size_t)(T * L) * sizeof(double));
    for (int64_t l = 0; l < L; ++l) {           /* :86 */
        double c = 0.0;
        for (int64_t t = 0; t < T; ++t) {       /* :89-90 */
            snow[t] = prec[t * L + l] * frac_solid_prec[t * L + l];
            rain[t * L + l] = prec[t * L + l] - snow[t];
            c += snow[t];
        }
        const double Psolannual = 365.25 * (c / (double)T);    /* :93 */
        double swe_max = 0.0, Thmax = 0.0;
        for (int64_t t = 0; t < T; ++t) {
            double g, e, s;
            if (t == 0) {                       /* :98-102 */
                g = snow_pack_init;
                sca[t * L + l] = sca_init;
            } else {
                g = G[(t - 1) * L + l] + snow[t];
            }
            if (t == 0) e = thermal_state_init; /* :105-110 */
            else e = CTG * eTG[(t - 1) * L + l]
                     + (1 - CTG) * mean_temp[t * L + l];
            if (e > 0) e = 0.0;
            double pot_melt;                    /* :113-120 */
            if (e == 0 && mean_temp[t * L + l] > 0) {
                pot_melt = Kf * mean_temp[t * L + l];
                if (pot_melt > g) pot_melt = g;
            } else {
                pot_melt = 0.0;
            }
            const double snow_balance = snow[t] - pot_melt;     /* :123 */
            if (snow_balance >= 0) {            /* :126-129 */
                /* sca[t-1]: row -1 (= T-1, still zero) when t == 0 */
                const double prev = sca[((t == 0) ? (T - 1) : (t - 1)) * L + l];
                s = prev + snow_balance / Thacc;
                swe_max = nb_max(swe_max, g);
            } else {                            /* :130-142 */
                const double Thmelt = Psolannual * Rsp;
                if (swe_max > Thmelt) Thmax = Thmelt;
                else Thmax = swe_max;
                if (Thmax > 0) s = g / Thmax;
                else s = 0.0;
            }
            s = nb_min(nb_max(s, 0.0), 1.0);    /* :145 */
            double melt = (0.9 * s + 0.1) * pot_melt;           /* :148 */
            melt = nb_min(melt, g);             /* :151 */
            g = g - melt;                       /* :154 */
            if (g == 0) swe_max = 0.0;          /* :157-158 */
            G[t * L + l] = g;
            eTG[t * L + l] = e;
            sca[t * L + l] = s;
            liquid_water[t * L + l] = rain[t * L + l] + melt;   /* :162 */
        }
    }
    for (int64_t t = 0; t < T; ++t) {           /* :165-166 */
        double c = 0.0;
        for (int64_t l = 0; l < L; ++l) c += liquid_water[t * L + l];
        outflow[t] = c / (double)L;
    }
    free(liquid_water); free(snow);
}

/* reference: rrmpg/models/icemelt_model.py:15-65 followed by the layer
 * weighting of the couplings, np.sum(icemelt * frac_ice[None, :], axis=1)
 * (cemaneigegr4jice_model.py:81-84): total[t] = sum_l ice[t,l]*frac_ice[l],
 * summed left to right. */
void oracle_icemelt_total(const double *temp, const double *snow,
                          const double *frac_ice, int64_t T, int64_t L,
                          double ddf, double *total)
{
    for (int64_t t = 0; t < T; ++t) {
        double c = 0.0;
        for (int64_t l = 0; l < L; ++l) {
            double melt = ddf * (temp[t * L + l] - 0);  /* tbase = 0, :57 */
            if (melt < 0) melt = 0.0;
            const double lw = (snow[t * L + l] > 1) ? 0.0 : melt;  /* :60-63 */
            c += lw * frac_ice[l];
        }
        total[t] = c;
    }
}

/* One parameter set of the three couplings.
 * hyst: params = {CTG,Kf,Thacc,Rsp,x1..x4[,DDF]}, else {CTG,Kf,x1..x4[,DDF]}.
 * reference: cemaneigehystgr4j_model.py:17-79, cemaneigegr4jice_model.py:
 * 20-93, cemaneigehystgr4jice_model.py:22-104.  Outputs not produced by a
 * variant may be NULL (sca: hyst only; icemelt: ice only; snowmelt: the snow
 * routine's outflow before the ice melt is added). */
int oracle_run_snow_gr4j(int hyst, int ice, const double *prec,
                         const double *mean_temp, const double *etp,
                         const double *frac_ice,
                         const double *frac_solid_prec, int64_t T, int64_t L,
                         double snow_pack_init, double thermal_state_init,
                         double sca_init, double s_init, double r_init,
                         const double *params, double *qsim, double *G,
                         double *eTG, double *s_store, double *r_store,
                         double *sca, double *icemelt, double *snowmelt,
                         double *rain)
{
    if (T <= 0 || L <= 0) return 0;
    const size_t tl = (size_t)(T * L);
    double *snow_out = (double *)malloc((size_t)T * sizeof(double));
    double *sca_b = sca ? sca : (double *)malloc(tl * sizeof(double));
    double *rain_b = rain ? rain : (double *)malloc(tl * sizeof(double));
    const double *gp;
    double ddf = 0.0;
    if (hyst) {
        oracle_run_cemaneigehyst(prec, mean_temp, frac_solid_prec, T, L,
                                 snow_pack_init, thermal_state_init, sca_init,
                                 params[0], params[1], params[2], params[3],
                                 snow_out, G, eTG, sca_b, rain_b);
        gp = params + 4;
        if (ice) ddf = params[8];
    } else {
        oracle_run_cemaneige(prec, mean_temp, frac_solid_prec, T, L,
                             snow_pack_init, thermal_state_init, params,
                             snow_out, G, eTG);
        gp = params + 2;
        if (ice) ddf = params[6];
    }
    double *liquid = (double *)malloc((size_t)T * sizeof(double));
    if (ice) {
        double *tot = icemelt ? icemelt : (double *)malloc((size_t)T * 8);
        oracle_icemelt_total(mean_temp, G, frac_ice, T, L, ddf, tot);
        for (int64_t t = 0; t < T; ++t) liquid[t] = snow_out[t] + tot[t];
        if (!icemelt) free(tot);
    } else {
        memcpy(liquid, snow_out, (size_t)T * sizeof(double));
    }
    if (snowmelt) memcpy(snowmelt, snow_out, (size_t)T * sizeof(double));
    const int rc = oracle_run_gr4j(liquid, etp, T, s_init, r_init, gp, qsim,
                                   s_store, r_store);
    free(liquid); free(snow_out);
    if (!sca) free(sca_b);
    if (!rain) free(rain_b);
    return rc;
}

/* Reference-shaped sweep of a coupling over N parameter sets (npar doubles
 * each).  2-D outputs [T][N], 3-D outputs [T][L][N]; all but qsim nullable. */
int oracle_simulate_snow_gr4j(int hyst, int ice, const double *prec,
                              const double *mean_temp, const double *etp,
                              const double *frac_ice,
                              const double *frac_solid_prec, int64_t T,
                              int64_t L, double snow_pack_init,
                              double thermal_state_init, double sca_init,
                              double s_init, double r_init,
                              const double *params, int64_t N, double *qsim,
                              double *G, double *eTG, double *s_store,
                              double *r_store, double *sca, double *icemelt,
                              double *snowmelt, double *rain, int nthreads)
{
    const int npar = 6 + (hyst ? 2 : 0) + (ice ? 1 : 0);
    int rc = 0;
    SWEEP_THREADS(nthreads)
#pragma omp parallel for num_threads(_nt) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        const size_t n = (size_t)(T > 0 ? T : 1), nl = n * (size_t)(L > 0 ? L : 1);
        double *a = (double *)calloc(5 * n, sizeof(double));
        double *b = (double *)calloc(4 * nl, sizeof(double));
        if (oracle_run_snow_gr4j(hyst, ice, prec, mean_temp, etp, frac_ice,
                                 frac_solid_prec, T, L, snow_pack_init,
                                 thermal_state_init, sca_init, s_init, r_init,
                                 params + npar * i, a, b, b + nl, a + n,
                                 a + 2 * n, b + 2 * nl, a + 3 * n, a + 4 * n,
                                 b + 3 * nl) != 0) {
#pragma omp atomic write
            rc = -1;
        }
        scatter(qsim, N, i, a, T);
        scatter(s_store, N, i, a + n, T);
        scatter(r_store, N, i, a + 2 * n, T);
        scatter(icemelt, N, i, a + 3 * n, T);
        scatter(snowmelt, N, i, a + 4 * n, T);
        scatter(G, N, i, b, T * L);
        scatter(eTG, N, i, b + nl, T * L);
        scatter(sca, N, i, b + 2 * nl, T * L);
        scatter(rain, N, i, b + 3 * nl, T * L);
        free(a); free(b);
    }
    return rc;
}
