// snow_core.h -- pieces shared by cemaneige.hip (Cemaneige, CemaneigeGR4J) and
// snownext_kernels.h (hysteresis / ice-melt couplings): the per-day Cemaneige step,
// the layer-count dispatch, workspace sizing and the forcing pre-pass.
#pragma once

#include "gr4j_core.h"

// defined in gr4j.hip: enqueues the scan of x4 that leaves the plan
// {max ceil(x4), #bad sets} at the start of the workspace (gr4j_core.h)
int rr_gr4j_plan_async(const double *params, int64_t N, int stride,
                       int x4_index, int *d_plan, int mem_cap, hipStream_t st);
// defined in gr4j.hip: bytes of unit-hydrograph scratch (UhMem, gr4j_core.h)
// behind a workspace whose launch may hold x4 up to max_x4 (0 up to 20)
size_t rr_gr4j_uh_scratch_bytes(int64_t N, double max_x4);
// Side streams for the tier kernels of one launch (gr4j.hip): with the waves
// choosing their own tier (gr4j_core.h gr4j_wave_selects) every tier's kernel
// has work, and enqueued on ONE stream they would run one after the other,
// each with a fraction of the GPU.  rr_tier_fork makes the calling thread's
// side streams of the current device wait for what `st` holds so far,
// rr_tier_stream(k) hands out side stream k (0..2), rr_tier_join makes `st`
// wait for all of them.  Events only: nothing blocks the host, and a stream
// capture of `st` follows the fork into the side streams.
int rr_tier_fork(hipStream_t st);
hipStream_t rr_tier_stream(int k);
int rr_tier_join(hipStream_t st);
// the sets of a launch ordered by ceil(x4) (gr4j.hip): `perm` [N] ints,
// `bins` 128 ints of scratch
int rr_gr4j_tier_sort_async(const double *params, int64_t N, int stride,
                            int x4_index, int *bins, int *perm,
                            hipStream_t st);

// Day record of the snow kernels, D = cema_record_len(L, with_etp) doubles:
//   [0, L) snow   [L, 2L) rain   [2L, 3L) mean temperature   [3L] etp (if any)
//   [D - 1] the day's observed discharge (0 when no score is fused): rides
//           along so that the score needs no second scalar load + wait at the
//           end of every day
// One record is wave-uniform and arrives with one burst of scalar loads.
// (Measured and dropped, round 2: a flag word per day -- temp[l] > 0, snow[l]
// == +0 -- with wave-uniform shortcuts for frost and bare ground that cut the
// snow routine from ~17 to 6-8 vector instructions per layer-day, bit for
// bit.  Decided per layer (two scalar branches per layer and day) every snow
// kernel got 10-19 % SLOWER; decided once per day (all layers frost / all
// bare) nothing was gained either.  Nor does it pay to take the wave-uniform
// `temp > 0` off the vector unit: as w = (temp > 0) ? 0.0 : NaN formed by four
// scalar integer instructions, with `e == w` as the whole melt condition, one
// vector compare per layer and day is saved and every snow kernel is 1-2 %
// slower; as a signed compare of the temperature's high word (round 4: one
// s_cmp) hipcc makes it a scalar branch INSIDE the masked melt block -- a
// taken branch on every warm day, 10 cycles for the compare's 4 -- dropped on
// reading the ISA.  profiles/README.md.)
static __host__ __device__ constexpr int cema_record_len(int L, bool with_etp)
{
    return 3 * L + (with_etp ? 1 : 0) + 1;
}

// defined in cemaneige.hip: packs the day records and the per-layer
// G_tresh[L] / Psolannual[L] / CemaGt table into the workspace (layout
// below); returns device pointers into it.  qobs: device, [T], or NULL.
int rr_cema_prepass(const double *prec, const double *mean_temp,
                    const double *frac, const double *etp, const double *qobs,
                    int64_t T, int L, void *workspace, hipStream_t st,
                    double **days_out, double **gt_out, double **state_out,
                    int *uncivil = nullptr,
                    int reg_layers = RR_CEMANEIGE_MAX_LAYERS);

static inline size_t cema_days_bytes(int64_t T, int64_t L, bool with_etp)
{
    if (T < 1) T = 1;
    if (L < 1) L = 1;
    // + one spare record: the fused kernels request the next day's record in
    // the middle of a day, the last day included
    return rr_align256((size_t)(T + 1) *
                       (size_t)cema_record_len((int)L, with_etp) * 8);
}

// per-layer constants: G_tresh[L], Psolannual[L], the CemaGt table [L] the
// register kernels read (below), one flag: every threshold suits the
// 3-FMA quotient, and one counter (a 64-bit integer): forcing values that
// rule out the SANE form of the snow routine (cema_day)
static inline size_t cema_gt_bytes(int64_t L)
{
    if (L < 1) L = 1;
    return rr_align256((size_t)(4 * L + 2) * 8);
}

// + the [nstate][L][N] snow-state scratch when the layers do not fit in
// registers (nstate = 2: G, eTG; 4 with the hysteresis' sca and SWE maximum)
// ... or, for the register kernels, the tiled kernels' work queue and
// hand-over scratch (common.h RrTiles): both snow states per layer, both
// GR4J stores, up to 5 + 11 hydrograph slots, the score sum
static inline int cema_tile_states(int64_t L) { return (int)(2 * L + 20); }
static inline size_t cema_tile_offset(int64_t T, int64_t L, bool with_etp)
{
    return 512 + cema_gt_bytes(L) + cema_days_bytes(T, L, with_etp);
}
// reg_layers: up to this many layers the model's kernels keep the snow states
// in registers (RR_CEMANEIGE_MAX_LAYERS; the hysteresis / ice couplings:
// RR_SNOWNEXT_REG_LAYERS)
static inline size_t cema_ws_bytes(int64_t T, int64_t L, bool with_etp,
                                   int64_t N, int nstate = 2,
                                   int reg_layers = RR_CEMANEIGE_MAX_LAYERS)
{
    size_t b = cema_tile_offset(T, L, with_etp);
    if (L > reg_layers && N > 0)
        b += rr_align256((size_t)nstate * (size_t)L * (size_t)N * 8);
    else if (N > 0)
        b += rr_tile_bytes(N, cema_tile_states(L));
    return b;
}


// Per-layer melt threshold as the kernels read it: {G_tresh, RN(1/G_tresh)}
// per layer, written by the pre-pass (cema_gtresh) behind the thresholds
// themselves.  Parameter independent, so it is fetched with a scalar load at
// its point of use (the days on which a layer melts) instead of occupying
// 4 SGPRs per layer for the whole time loop -- the fused kernels are short of
// SGPRs, and every one that overflows into a VGPR lane costs a v_readlane
// (a VALU slot) per use.
struct CemaGt { double gt, rgt; };
// Tolerance-driven forms of the snow routine: the
// quotient G / G_tresh and the layer mean as ONE multiply by the rounded
// reciprocal (1.5 ulp, any numerator: no numerator vote), 0.9 ratio + 0.1 as
// one FMA.  The thermal state keeps the reference's own sequence and stays
// bit-identical to it; snow pack and outflow are within 1e-14 of it over 30
// years (tests/conftest.py SNOW_TOL = 1e-12).  Cemaneige 1M sets 38.9 -> 35.0
// ms, scores 33.4 -> 27.9; fused 83.1 -> 77.6, its 125k shard 12.9 -> 11.7.
typedef const CemaGt __attribute__((address_space(4))) *cema_gt_ptr_t;

// c / L (the layer mean, np.mean's division by the size): L is a constant --
// one multiply by RN(1/L).
template <int L, class V = CarefulVotes>
__device__ __forceinline__ double cema_layer_mean(double c, V &&votes = V())
{
    const InvDivisor inv_L = {(double)L, 1.0 / (double)L, true};
    return inv_mul_core(c, inv_L);
}

// One day of the snow routine for all L layers of one parameter set
// (cemaneige_model.py:83-125).  Returns the layer-mean liquid outflow.
// FIRST: day 0, whose states are the initial values (:85-96) -- a template
// argument because the kernels peel that day off their time loop (as a
// run-time flag it costs two selects per state and layer on EVERY day).
// gt_tab: the CemaGt table; gt_ok: lanes (all or none) for which every
// threshold suits the 3-FMA quotient.
// GT_REGS: the thresholds come from `gt_regs` ({G_tresh, 1/G_tresh} per layer
// in VGPR pairs, cema_gt_to_regs) instead of a scalar load from the table at
// the point of use -- the small-sweep kernels, which have registers to spare
// and nobody to hide a load's latency behind.
// The pack's update and the layer sum of the outflow formed inside the melt
// block only (cema_day_io; needs the faithful forms' block).  Measured and
// left off: two vector instructions fewer on a layer's idle days bought
// Cemaneige 1M sets 28.92 -> 28.68 ms and cost the small sweeps (125k sets
// 6.49 -> 6.60, fused 10.75 -> 11.0; profiles/r04_lazy_sums_ab.txt).
template <int L>
struct CemaGtRegs { double gt[L], rgt[L]; };

// "No layer's temperature is above zero", read off the record's high words
// (scalar integer maxima).  For the forcing of a SANE wave only: every finite
// temperature but a positive subnormal one has `temp > 0` == `high word > 0`
// (cemaneige.hip cema_pack counts those among what rules SANE out).
template <int L>
__device__ __forceinline__ bool cema_frost_everywhere(
    const double *__restrict__ day)
{
    // (s_max_i32 written out: from C++ hipcc forms the maximum on the vector
    // unit -- three v_mov, two v_max3_i32 and a v_cmp a day, 80.1 -> 82.3
    // instructions per set-day instead of fewer)
    int warmest = __double2hiint(day[2 * L]);
#pragma unroll
    for (int l = 1; l < L; ++l)
        asm("s_max_i32 %0, %1, %2"
            : "=s"(warmest)
            : "s"(warmest), "s"(__double2hiint(day[2 * L + l]))
            : "scc");
    return warmest <= 0;
}

template <int L>
__device__ __forceinline__ void cema_gt_to_regs(cema_gt_ptr_t gt_tab,
                                                CemaGtRegs<L> &g)
{
#pragma unroll
    for (int l = 0; l < L; ++l) {
        g.gt[l] = gt_tab[l].gt;
        g.rgt[l] = gt_tab[l].rgt;
        // (pins them in VGPRs: as uniform values hipcc would hold them in
        // SGPRs, which these kernels do not have)
        asm volatile("" : "+v"(g.gt[l]), "+v"(g.rgt[l]));
    }
}

// SANE: the wave has established (cema_wave_is_sane) that in every lane
//   * the thermal state can never be NaN -- 0 <= CTG <= 1, a finite initial
//     state and finite temperatures make it a convex combination of finite
//     values --, so `if e > 0: e = 0` (:93-96) is one v_min_f64 (which would
//     turn a NaN into 0) instead of a compare and a 64-bit select;
//   * the potential melt Kf * temp can never be NaN (Kf is not), so its cap
//     by the pack is the hardware minimum;
//   * the snow pack can never be negative -- an initial pack and snowfall
//     that are not negative, and melt <= pack --, so the factor
//     0.9 ratio + 0.1 of a day without melt is finite and positive without
//     asking (the idle vote's second compare).
// Three vector instructions per layer and day; bit-identical by construction
// (and checked: every snow fixture is bit-exact in both forms).
// G_in / eTG_in -> G / eTG: the states' two generations (the optimistic time
// loops, common.h OptimisticVotes), or the same arrays (a layer's state is
// read before it is written).
template <int L, bool FIRST, bool GT_REGS = false, bool SANE = false,
          class V = CarefulVotes, bool COUPLED = false>
__device__ __forceinline__ double cema_day_io(
    const double *__restrict__ day, cema_gt_ptr_t gt_tab, lanemask_t gt_ok,
    double snow_pack_init, double thermal_state_init, double CTG,
    double one_minus_CTG, double Kf, const double (&G_in)[L],
    const double (&eTG_in)[L], double (&G)[L], double (&eTG)[L],
    const CemaGtRegs<L> *gt_regs = nullptr, V &&votes = V())
{
    double c = 0.0;
    if constexpr (SANE && !FIRST) {
        // Frost in every layer (a third of the days of a temperate year): no
        // lane melts anything, whatever its thermal state -- `temp > 0` is
        // false (:99) --, so melt = pot_melt = 0, the pack keeps g = G + snow
        // (g - (+0), g never -0) and the layer gives its rain (rain + (+0),
        // rain never -0: the pre-pass counts such forcing among what rules out
        // SANE, as it does a positive subnormal temperature -- for every other
        // finite one `temp > 0` is `high word > 0`).  One scalar question per
        // day, five vector instructions a layer instead of ten.
        // (the question spelled out HERE, not asked through
        // cema_frost_everywhere: behind the call's bool hipcc lays the frost
        // days' block out of line, allocates 20 VGPRs more and the sweep takes
        // 28.3 ms instead of 22.8 -- 25.9 without the frost days;
        // profiles/r05_hyst_days_ab.txt)
        int warmest = __double2hiint(day[2 * L]);
#pragma unroll
        for (int l = 1; l < L; ++l)
            asm("s_max_i32 %0, %1, %2" : "=s"(warmest)
                : "s"(warmest), "s"(__double2hiint(day[2 * L + l])) : "scc");
        if (warmest <= 0) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const double snow = day[l], rain = day[L + l],
                             temp = day[2 * L + l];
                // (no clamp: CTG and 1 - CTG in [0, 1], yesterday's state
                // and the temperature not above zero -- neither is their
                // weighted sum, and `if e > 0: e = 0` (:93-96) leaves a zero of
                // either sign alone)
                const double e = CTG * eTG_in[l] + one_minus_CTG * temp;
                G[l] = G_in[l] + snow;
                eTG[l] = e;
                c = (l == 0) ? rain : c + rain;
            }
            return cema_layer_mean<L>(c, votes);
        }
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const double snow = day[l], rain = day[L + l], temp = day[2 * L + l];
        double g, e;
        if (FIRST) {                                       // :85-96
            g = snow_pack_init;
            e = thermal_state_init;
        } else {
            g = G_in[l] + snow;
            // (not contracted: the thermal state crosses zero, where an FMA's
            // other rounding is 1e-9 of the value -- and it would buy one
            // instruction)
            e = CTG * eTG_in[l] + one_minus_CTG * temp;
        }
        if (SANE && !FIRST) {
            // (written out: from C++ hipcc quiets the operand with a
            // v_max x, x of its own first)
            asm("v_min_f64 %0, %1, 0" : "=v"(e) : "v"(e));
        } else {
            if (e > 0) e = 0.0;
        }
        double pot_melt = 0.0;                             // :99-106
        // (`temp > 0`: in a SANE wave the sign of the record's high word --
        // cema_frost_everywhere --, one scalar compare where the vector unit
        // would compare a uniform value in every lane)
        // (COUPLED -- the kernels with a GR4J day behind the snow routine --
        // only: they gain 1 % from it, 125k and 1M sets alike, the plain
        // Cemaneige kernel loses 2 %; profiles/r05_hyst_days_ab.txt)
        const bool warm = (COUPLED && SANE && !FIRST)
                              ? __double2hiint(temp) > 0 : temp > 0;
        if (SANE && !FIRST && GT_REGS) {
            // (the small-sweep kernels evaluate it for every lane and select:
            // no exec-masked block, no branch over it -- at two waves per
            // SIMD the scalar work of a branch is not hidden: 125k sets
            // 14.98 -> 14.68 ms; at a million sets the block wins, 92.2 vs
            // 92.8)
            const double pm = rr_hw_min(Kf * temp, g);
            pot_melt = (e == 0 && warm) ? pm : 0.0;
        } else if (e == 0 && warm) {
            pot_melt = Kf * temp;
            // (SANE: Kf is not NaN, so neither is the product, and numba's
            // `if pot_melt > G: pot_melt = G` is the hardware minimum)
            if (SANE) pot_melt = rr_hw_min(pot_melt, g);
            else if (pot_melt > g) pot_melt = g;
        }
        // Most days of a year nothing melts in a layer: frost (temp <= 0,
        // the same for every lane) or no snow left (G == 0 in every lane), so
        // pot_melt is a zero in all lanes of the wave.  Then
        // melt = (0.9*ratio + 0.1) * pot_melt is that same zero -- the factor
        // is finite and positive whenever G >= 0 (ratio in [0, 1]) -- and the
        // wave skips the quotient and the melt arithmetic.
        const lanemask_t idle =
            SANE ? RR_LANES(pot_melt == 0.0)
                 : (RR_LANES(pot_melt == 0.0) & RR_LANES(g >= 0.0));
        double melt = pot_melt;
        if (rr_exec() & ~idle) {
            // G / G_tresh: the threshold is fixed for the whole run, so the
            // quotient is the 3-instruction correctly rounded form of
            // common.h
            InvDivisor inv_gt;
            inv_gt.ok = gt_ok != 0;
            if constexpr (GT_REGS) {
                inv_gt.b = gt_regs->gt[l];
                inv_gt.rb = gt_regs->rgt[l];
            } else {
                cema_gt_ptr_t pg = gt_tab + l;
                asm volatile("" : "+s"(pg)); // keeps the load inside the branch
                inv_gt.b = pg->gt;
                inv_gt.rb = pg->rgt;
                // (both fields now: one s_load_dwordx4 and one wait, not a
                // second load + wait where the reciprocal is first used)
                asm volatile("" : : "s"(inv_gt.b), "s"(inv_gt.rb));
            }
            // (SANE: the thresholds are not negative -- no snowfall is -- so
            // `G / G_tresh if G < G_tresh else 1` is the hardware minimum of
            // the quotient and 1: a quotient of inf or NaN, G_tresh = 0, is
            // dropped for the 1 the reference takes there)
            // (the quotient -- and its vote -- for every lane: a vote inside
            // a per-lane conditional would make the vote mask a per-lane
            // value)
            const double gq = mul_by_inverse_m(g, inv_gt, gt_ok, votes);
            // (the faithful quotient of a pack a few ulp below its threshold
            // can round to 1 + ulp, and a ratio above 1 would melt more than
            // the potential melt, the pack: it is capped in both forms -- a
            // NaN quotient stays NaN in the general one, as the reference's)
            const double ratio =                           // :109-112
                SANE ? rr_hw_min(gq, 1.0)
                     : ((g < inv_gt.b) ? ((gq > 1.0) ? 1.0 : gq) : 1.0);
            // (one v_fma_f64 with 0.9 in an SGPR pair and 0.1 in a VGPR
            // pair: from __builtin_fma hipcc holds them in those very
            // registers and then issues v_mov_b64 + v_fmac_f64, the VOP2 form
            // whose sum is its destination)
            // (not in the small-sweep form, GT_REGS: 125k sets 6.50 -> 6.65
            // ms with it)
            double factor;
            if constexpr (GT_REGS)
                factor = __builtin_fma(0.9, ratio, 0.1);
            else
                asm("v_fma_f64 %0, %1, %2, %3"
                    : "=v"(factor) : "v"(ratio), "s"(0.9), "v"(0.1));
            melt = factor * pot_melt;                      // :115
        }
        g = g - melt;                                      // :118
        G[l] = g;
        eTG[l] = e;
        c = (l == 0) ? rain + melt : c + (rain + melt);    // :121, :125
    }
    return cema_layer_mean<L>(c, votes);
}

template <int L, bool FIRST, bool GT_REGS = false, bool SANE = false,
          bool COUPLED = false>
__device__ __forceinline__ double cema_day(
    const double *__restrict__ day, cema_gt_ptr_t gt_tab, lanemask_t gt_ok,
    double snow_pack_init, double thermal_state_init, double CTG,
    double one_minus_CTG, double Kf, double (&G)[L], double (&eTG)[L],
    const CemaGtRegs<L> *gt_regs = nullptr)
{
    return cema_day_io<L, FIRST, GT_REGS, SANE, CarefulVotes, COUPLED>(
        day, gt_tab, gt_ok, snow_pack_init, thermal_state_init, CTG,
        one_minus_CTG, Kf, G, eTG, G, eTG, gt_regs);
}


// The reference's own snow day (cemaneige_model.py:85-125), statement by
// statement -- IEEE quotients G / G_tresh and c / L, 0.9 ratio + 0.1 as a
// multiply and an add (the files are built -ffp-contract=off), the layer sum
// started from 0.0 as np.mean starts it -- for the one-lane-per-set kernels
// that run the sets the fast forms are not meant for
// (cemaneigegr4j_reference_kernel, snow_gr4j_reference_kernel): the outflow
// they hand to the reference's GR4J day is the reference's to the bit.  It
// has to be: a set with x1 = 1e308 forms tanh(p_n / x1) among the subnormals,
// p_n - p_s is then a rounding residue of +-1e-17 mm, and under a negative x3
// its SIGN decides whether the routing store's exchange term is a NaN the
// next day (tests/test_gpu_fuzz.py, seed 200: one ulp of the outflow turned
// a discharge of 9e302 into 0).  gt: G_tresh[L] as cema_gtresh wrote it
// (summed left to right like the reference's mean).
template <int L, bool FIRST>
__device__ __forceinline__ double cema_ref_day(
    const double *__restrict__ day, const double *__restrict__ gt,
    double snow_pack_init, double thermal_state_init, double CTG, double Kf,
    double (&G)[L], double (&eTG)[L])
{
    double c = 0.0;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const double snow = day[l], rain = day[L + l], temp = day[2 * L + l];
        double g, e;
        if (FIRST) {                                       // :85-96
            g = snow_pack_init;
            e = thermal_state_init;
        } else {
            g = G[l] + snow;
            e = CTG * eTG[l] + (1 - CTG) * temp;
        }
        if (e > 0) e = 0.0;
        double pot_melt = 0.0;                             // :99-106
        if (e == 0 && temp > 0) {
            pot_melt = Kf * temp;
            if (pot_melt > g) pot_melt = g;
        }
        const double G_tresh = gt[l];
        const double ratio = (g < G_tresh) ? g / G_tresh : 1.0;  // :109-112
        const double melt = (0.9 * ratio + 0.1) * pot_melt;      // :115
        g = g - melt;                                      // :118
        G[l] = g;
        eTG[l] = e;
        c += rain + melt;                                  // :121, :125
    }
    return c / (double)L;
}

// Whether the wave may run the SANE form of cema_day: the conditions of its
// comment, for every lane and for the whole forcing (`gtresh` + 4L + 1: the
// pre-pass's count of temperatures that are not finite and snowfalls that are
// negative, a 64-bit integer).
__device__ __forceinline__ bool cema_wave_is_sane(const double *gtresh, int L,
                                                  double CTG, double Kf,
                                                  double snow_pack_init,
                                                  double thermal_state_init)
{
    const long long bad_forcing =
        *(const long long *)(gtresh + 4 * L + 1);
    // (Kf: anything but NaN -- v_cmp_class mask 0x3fc)
    const lanemask_t ctg_ok = RR_LANES(CTG >= 0.0) & RR_LANES(CTG <= 1.0) &
                              lanes_of_class(Kf, 0x3fc);
    return bad_forcing == 0 && !(snow_pack_init < 0.0) &&
           fabs(thermal_state_init) <= 1e300 && (rr_exec() & ~ctg_ok) == 0;
}

// calls f(std::integral_constant<int, L>) for the runtime L in
// 1..RR_CEMANEIGE_MAX_LAYERS (the callers send more layers to the kernels
// that run from the HBM state scratch: per-layer kernels for L = 6..8 were
// 72 of the library's kernels and 5 MB of its code for configurations nobody
// has asked for; removed in round 6)
template <int MAXL = RR_CEMANEIGE_MAX_LAYERS, class F>
static inline void dispatch_layers(int L, F &&f)
{
    static_assert(MAXL == 5, "dispatch_layers");
    switch (L) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    default: f(std::integral_constant<int, 5>{}); break;
    }
}

