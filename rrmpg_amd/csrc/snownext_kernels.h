// snownext_kernels.h -- next-tier ensemble kernels for gfx950: the SWE-SCA
// hysteresis snow routine, the degree-day ice melt, and their couplings with
// GR4J (the reference's CemaneigeHystGR4J, CemaneigeGR4JIce,
// CemaneigeHystGR4JIce).  One fused kernel template, one lane per parameter
// set, all snow states in registers; see snow_core.h / gr4j_core.h.  Included
// by the two translation units that hold the C-ABI entry points
// (snownext.hip, snownext_hyst.hip).
#pragma once
#include "snow_core.h"
#include "gr4j_reference.h"

// ===========================================================================
// Next tier: SWE-SCA hysteresis snow routine, ice
// melt, and their couplings with GR4J -- the reference's CemaneigeHystGR4J,
// CemaneigeGR4JIce and CemaneigeHystGR4JIce.
// ===========================================================================

// One day of the hysteresis snow routine for all L layers of one parameter
// set (reference: rrmpg/models/cemaneigehyst_model.py:95-162).  Four states
// per layer: snow pack G, thermal state eTG, snow-covered area sca and the
// maximum SWE before melt.  Returns the layer-mean liquid outflow.
// sca_prev0: what the reference reads as sca[t-1] at t = 0 -- row -1, i.e. 0
// (or sca_init when T == 1); sca_init itself never survives (quirk Q8).
// FIRST: day 0 (peeled off the kernels' time loops, see snow_core.h cema_day).
// SANE: the wave has established (cema_hyst_wave_is_sane) what snow_core.h's
// cema_day asks for -- the thermal state cannot be NaN, its clamp is one
// v_min_f64 -- and, for the hysteresis: the pack G is a finite number >= +0
// (initial pack <= 1e300 and snowfall <= 1e290 are, melt <= pack), Kf >= +0,
// Thacc > 0 and within the
// 3-FMA quotient's range, Rsp and Psolannual finite, the covered area of "day
// -1" a number in [+0, 1e300].  Then, in either branch, the covered area
// before its clamp is a number >= +0 (previous area + a quotient >= +0, or
// G / Thmax with Thmax > 0 finite, or 0): the clamp to [0, 1] (:145, numba's
// min(max(.)): two compares and two 64-bit selects) is ONE v_min_f64 with 1;
// the melt factor 0.9 sca + 0.1 is then in [0.1, 1.0] (0.9 + 0.1 rounds to
// 1.0, and rounding is monotonic), so melt = factor * pot_melt <= pot_melt
// <= G and `melt = min(melt, G)` (:151) is the identity; and the two
// maxima / minima of non-negative numbers (:128, :134-137) are the hardware's.
// Ten vector instructions per layer and day, bit-identical by construction.
// REF (the one-lane kernel of the sets the fast forms are not meant for,
// snow_gr4j_reference_kernel): the layer sum from 0.0 and the IEEE quotient
// c / L as the reference's mean forms them, and a plain snow_balance / Thacc
// -- the outflow is then the reference's to the bit (snow_core.h
// cema_ref_day says why it has to be).
// IDLE DAYS (SANE waves, every day but the first; HYST_IDLE_DAYS): most days of
// a year no layer sees snowfall and no lane of the wave melts anything --
// summer without a pack, dry frost.  The day is therefore evaluated in two
// steps: pack + snowfall, thermal state and potential melt of every layer
// (eleven vector instructions a layer, straight line), then ONE wave-uniform
// question -- any snowfall (the record's own bits), any lane with a potential
// melt that is not zero?  If not, the reference's statements reduce to: the
// covered area keeps its bits (prev + (+0) / Thacc, and min(., 1) of a value
// that was clamped the day before), the SWE maximum takes the pack's (:128),
// melt = factor * (+0) = +0 with a finite positive factor, the pack keeps its
// bits, `if G == 0: max = 0` finds the zero the day before left, and the
// outflow is the layers' rain (rain + (+0), rain never -0: the pre-pass
// counts a rain with the sign bit set among the forcing values that rule out
// SANE).  Two instructions per layer instead of thirty.
// What the synthetic forcing of the bench has of such days: 22 % -- dry frost.
// A warm day is never idle: a pack melts by a tenth of itself a day at the end
// (factor 0.9 sca + 0.1 -> 0.1) and is never exactly gone, in no lane.
// (Measured and dropped, round 5: the other days split the same way -- every
// lane of every layer accumulating / melting -- with the five layers' quotients
// in one basic block for the scheduler to interleave: 16-24 more VGPRs, spills
// to scratch in the 3- and 10-slot tiers, 135 -> 159 ms.  Nor a frost-and-dry
// day decided from the record alone in front of the layers' first step (what
// snow_core.h cema_day_io gains from, 25.9 -> 23.8 ms): 135 -> 197 ms.
// profiles/r05_hyst_days_ab.txt)
// (measured, off: 135.0 -> 140.0 ms -- these kernels wait, at two or three
// waves per SIMD, for their dependent chains, not for issue slots, and the
// masked move sits on the chain)
template <int L, bool FIRST, bool SANE = false, bool REF = false>
__device__ __forceinline__ double cema_hyst_day(
    const double *__restrict__ day, const double *__restrict__ psol,
    double snow_pack_init, double thermal_state_init,
    double sca_prev0, double CTG, double one_minus_CTG, double Kf,
    const InvDivisor &inv_Thacc, lanemask_t thacc_m, double Rsp,
    double (&G)[L], double (&eTG)[L],
    double (&sca)[L], double (&swe_max)[L])
{
    constexpr bool TWO_STEPS = SANE && !FIRST && !REF;
    double g_[L], e_[L], pot_[L];
    if constexpr (TWO_STEPS) {
        lanemask_t busy = 0;
        unsigned snowfall = 0;
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const double snow = day[l], temp = day[2 * L + l];
            snowfall |= (unsigned)__double2hiint(snow) |
                        (unsigned)__double2loint(snow);
            g_[l] = G[l] + snow;
            double e = CTG * eTG[l] + one_minus_CTG * temp;
            asm("v_min_f64 %0, %1, 0" : "=v"(e) : "v"(e));
            e_[l] = e;
            const double pm = rr_hw_min(Kf * temp, g_[l]);
            // (`temp > 0` off the record's high word -- a SANE wave's
            // temperatures are finite and not positive subnormals, snow_core.h
            // cema_frost_everywhere --: a scalar compare instead of a vector
            // one per layer and day)
            const bool warm = __double2hiint(temp) > 0;
            pot_[l] = (e == 0 && warm) ? pm : 0.0;
            busy |= RR_LANES(pot_[l] != 0.0);
        }
        if (snowfall == 0 && busy == 0) {
            double c = 0.0;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                swe_max[l] = rr_hw_max(swe_max[l], g_[l]);
                G[l] = g_[l];
                eTG[l] = e_[l];
                const double rain = day[L + l];
                c = (l == 0) ? rain : c + rain;
            }
            return cema_layer_mean<L>(c);
        }
    }
    double c = 0.0;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const double snow = day[l], rain = day[L + l], temp = day[2 * L + l];
        double g, e;
        double pot_melt = 0.0;                             // :113-120
        if constexpr (TWO_STEPS) {
            g = g_[l];
            e = e_[l];
            pot_melt = pot_[l];
        } else {
            if (FIRST) {                                   // :98-110
                g = snow_pack_init;
                e = thermal_state_init;
            } else {
                g = G[l] + snow;
                e = CTG * eTG[l] + one_minus_CTG * temp;
            }
            if (SANE && !FIRST) {
                asm("v_min_f64 %0, %1, 0" : "=v"(e) : "v"(e));
            } else {
                if (e > 0) e = 0.0;
            }
            if (e == 0 && temp > 0) {
                pot_melt = Kf * temp;
                if (SANE) pot_melt = rr_hw_min(pot_melt, g);  // (Kf is not NaN)
                else if (pot_melt > g) pot_melt = g;
            }
        }
        const double snow_balance = snow - pot_melt;       // :123
        double sc;
        if (snow_balance >= 0) {                           // :126-129
            const double prev = FIRST ? sca_prev0 : sca[l];
            // (a day without snowfall or melt has snow_balance == 0; the
            // balance is not negative here, so the cheap integer form of the
            // numerator vote -- +0 or [2^-900, 2^196) -- applies)
            if constexpr (REF)
                sc = prev + snow_balance / inv_Thacc.b;
            else
                sc = prev + div_by_invariant_m(
                                snow_balance, gr4j_num_mask(snow_balance),
                                inv_Thacc, thacc_m);
            swe_max[l] = SANE ? rr_hw_max(swe_max[l], g)
                              : nb_max(swe_max[l], g);
        } else {                                           // :130-142
            const double Thmelt = psol[l] * Rsp;
            const double Thmax =
                SANE ? rr_hw_min(swe_max[l], Thmelt)
                     : ((swe_max[l] > Thmelt) ? Thmelt : swe_max[l]);
            sc = (Thmax > 0) ? g / Thmax : 0.0;
        }
        double melt;
        if constexpr (SANE) {
            sc = rr_hw_min(sc, 1.0);                       // :145
            melt = (0.9 * sc + 0.1) * pot_melt;            // :148, :151
        } else {
            sc = nb_min(nb_max(sc, 0.0), 1.0);             // :145
            melt = (0.9 * sc + 0.1) * pot_melt;            // :148
            melt = nb_min(melt, g);                        // :151
        }
        g = g - melt;                                      // :154
        if (g == 0) swe_max[l] = 0.0;                      // :157-158
        G[l] = g;
        eTG[l] = e;
        sca[l] = sc;
        if constexpr (REF) c += rain + melt;
        else c = (l == 0) ? rain + melt : c + (rain + melt);  // :162, :166
    }
    if constexpr (REF) return c / (double)L;
    else return cema_layer_mean<L>(c);
}

// Whether the wave may run the SANE form of cema_hyst_day (its comment).
__device__ __forceinline__ bool cema_hyst_wave_is_sane(
    const double *gtresh, int L, double CTG, double Kf,
    const InvDivisor &inv_Thacc, double Rsp, double snow_pack_init,
    double thermal_state_init, double sca_prev0)
{
    if (!cema_wave_is_sane(gtresh, L, CTG, Kf, snow_pack_init,
                           thermal_state_init))
        return false;
    bool psol_ok = true;                       // Psolannual[l], wave-uniform
    for (int l = 0; l < L; ++l) psol_ok = psol_ok && fabs(gtresh[L + l]) <= 1e300;
    // Kf: +0, positive or +inf (v_cmp_class mask 0x3c0)
    const lanemask_t lanes_ok = lanes_of_class(Kf, 0x3c0) &
                                RR_LANES(inv_Thacc.ok) &
                                RR_LANES(inv_Thacc.b > 0.0) &
                                RR_LANES(fabs(Rsp) <= 1e300);
    return psol_ok && snow_pack_init >= 0.0 && snow_pack_init <= 1e300 &&
           sca_prev0 >= 0.0 &&
           !__builtin_signbit(sca_prev0) && sca_prev0 <= 1e300 &&
           (rr_exec() & ~lanes_ok) == 0;
}

// Where the optional parameters sit in a record of `npar` doubles:
// {CTG, Kf, [Thacc, Rsp,] x1, x2, x3, x4 [, DDF]}.
struct SnowParLayout {
    int npar, i_x1, i_ddf;
};

// Minimum waves per SIMD the register allocation is held to.  At least 2: the
// x4 <= 10 register tier of the hysteresis + ice variant would otherwise take
// a few AGPRs more than 256 registers and drop to one wave per SIMD.  Small
// configurations (<= 5 layers, unit hydrographs in 3+7 registers or in LDS)
// are held at 3 (hysteresis: four states per layer) or 4 waves; a handful of
// spills cost less than the lost wave.
// (the hysteresis couplings' 5-slot tier held to three waves per SIMD: 180 ->
// 168 VGPRs and a few spills, 135.4 -> 133.3 ms, hysteresis + ice 155.4 ->
// 152.9 -- these kernels wait for their chains, a third wave hides more of
// them than the spills cost; the 3-slot tier at four waves 144.6 ms, the
// 10-slot tier at three 412.8)
// (round 6, under the iterative-ilp scheduling of snownext_hyst.hip: held to
// three waves 129.6 ms, not held 132.8)
#define SNOW_TIER5_WAVES 3
template <int L, class UH, bool HYST>
constexpr int snow_min_waves()
{
    if (SNOW_TIER5_WAVES > 2 && HYST && L <= 5 &&
        std::is_same<UH, UhRegs<5>>::value)
        return SNOW_TIER5_WAVES;
    return (L <= 5 && (std::is_same<UH, UhRegs<3>>::value ||
                       uh_is_indexed<UH>)) ? (HYST ? 3 : 4) : 2;
}

// Output pointers.  Passed as the FIRST kernel argument and never touched by
// name: on the days something is stored the kernel re-reads the struct from
// offset 0 of its kernarg segment with one scalar load.  As ordinary
// arguments the eight pointers + ld would sit in 18 SGPRs for the whole time
// loop of a kernel that is already short of them (the overflow goes to VGPR
// lanes and every use then costs a v_readlane, i.e. a VALU slot).
struct SnowOut {
    double *qsim, *G, *eTG, *s_store, *r_store, *sca, *icemelt, *snowmelt;
    int64_t ld;
};
typedef const SnowOut __attribute__((address_space(4))) *snow_out_ptr_t;

template <int L, class UH, bool HYST, bool ICE>
__global__ __launch_bounds__(RR_BLOCK, (snow_min_waves<L, UH, HYST>())) void
snow_gr4j_kernel(
    SnowOut /* read through the kernarg segment, see above */,
    const double *__restrict__ days, const double *__restrict__ gtresh,
    const double *__restrict__ frac_ice, int64_t T, double snow_pack_init,
    double thermal_state_init, double sca_init, double s_init, double r_init,
    const double *__restrict__ params, SnowParLayout lay, int64_t N,
    const int *__restrict__ plan, int force_lds, int wq, int ws,
    const double *__restrict__ qobs, double *__restrict__ sse,
    double *__restrict__ uh_mem, const int *__restrict__ perm)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // `perm` (score-only sweeps, or NULL): the sets ordered by ceil(x4), so
    // that most waves need a narrower hydrograph tier than the launch's
    // widest (gr4j_core.h gr4j_wave_selects); lane g simulates set perm[g]
    const int64_t g_lane = (int64_t)blockIdx.x * RR_BLOCK + threadIdx.x;
    const bool active = g_lane < N;
    const int64_t g_set = active ? g_lane : N - 1;
    const int64_t i = perm ? (int64_t)perm[g_set] : g_lane;
    const double *p = params + (perm ? i : g_set) * lay.npar;
    int n1cap, n2cap;
    // per-wave tiers only with the sets ordered (perm: the by-tier launch,
    // whose tier kernels share the GPU on streams of their own); any other
    // launch keeps the launch's tier -- one kernel does the work, the others
    // return at once -- as before round 5: on ONE stream the tier kernels of
    // a block that happens to be ordered by x4 would take turns
    if (!(perm ? gr4j_wave_selects<UH>(plan, force_lds, p[lay.i_x1 + 3], n1cap,
                                       n2cap)
               : gr4j_plan_selects<UH>(plan, force_lds, n1cap, n2cap)))
        return;
    const double CTG = p[0], Kf = p[1];
    const double Rsp = HYST ? p[3] : 0.0;
    const InvDivisor inv_Thacc = make_inv_divisor(HYST ? p[2] : 1.0);
    const lanemask_t thacc_m = RR_LANES(inv_Thacc.ok);
    const double ddf = ICE ? p[lay.i_ddf] : 0.0;
    Gr4jPar P;
    P.set(p[lay.i_x1], p[lay.i_x1 + 1], p[lay.i_x1 + 2], p[lay.i_x1 + 3]);
    const double omc = 1 - CTG;
    double G[L], eTG[L], sca[L], swe_max[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        G[l] = 0.0; eTG[l] = 0.0; sca[l] = 0.0; swe_max[l] = 0.0;
    }
    const cema_gt_ptr_t gt_tab = (cema_gt_ptr_t)(gtresh + 2 * L);
    const lanemask_t gt_ok = gtresh[4 * L] != 0.0 ? ~0ull : 0ull;
    const double *psol = gtresh + L;
    const double sca_prev0 = (T == 1) ? sca_init : 0.0;
    UH uh;
    gr4j_uh_init(uh, lds, uh_mem, n1cap, n2cap, P.x4);
    double s = s_init * P.x1, r = r_init * P.x3;
    double acc = 0.0;
    const bool we = sse != nullptr;
    constexpr int D = cema_record_len(L, true);
    // (ice melt: every lane's factor in [+0, 1e300], every layer's glaciated
    // fraction finite -- what the frost days' shortcut below asks for)
    bool ice_tame = false;
    if constexpr (ICE) {
        ice_tame = (rr_exec() & ~(RR_LANES(ddf >= 0.0) &
                                  RR_LANES(ddf <= 1e300))) == 0;
        for (int l = 0; l < L; ++l)
            ice_tame = ice_tame && fabs(frac_ice[l]) <= 1e300;
    }
    // one day; `first` (a std::bool_constant) marks day 0, which is peeled
    // off the time loop
    auto one_day = [&](auto first, auto sane, int64_t t) {
        constexpr bool FIRST = decltype(first)::value;
        constexpr bool SANE = decltype(sane)::value;
        double day[D];          // by value: one wide scalar load per day
#pragma unroll
        for (int k = 0; k < D; ++k) day[k] = days[t * D + k];
        double snowmelt;
        if constexpr (HYST)
            snowmelt = cema_hyst_day<L, FIRST, SANE>(
                day, psol, snow_pack_init, thermal_state_init, sca_prev0, CTG,
                omc, Kf, inv_Thacc, thacc_m, Rsp, G, eTG, sca, swe_max);
        else
            snowmelt = cema_day<L, FIRST, false, SANE, true>(day, gt_tab, gt_ok, snow_pack_init,
                                          thermal_state_init, CTG, omc, Kf, G,
                                          eTG);
        double liquid = snowmelt;
        double ice_total = 0.0;
        if constexpr (ICE) {
            // degree-day ice melt where the layer is (nearly) snow free
            // (icemelt_model.py:55-63), weighted by the glaciated fraction
            // and summed over the layers left to right
            // (cemaneigegr4jice_model.py:81-87)
            // (under frost in every layer -- snow_core.h
            // cema_frost_everywhere -- a factor in [+0, 1e300] melts +-0 of
            // ice, the finite fractions' sum of it from +0 is +0, and the
            // snow routine's outflow, the layers' rain, never -0, keeps its
            // bits: the loop is skipped)
            bool frost = false;
            if constexpr (SANE && !FIRST)
                frost = ice_tame && cema_frost_everywhere<L>(day);
            if (!frost) {
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    double melt = ddf * day[2 * L + l];
                    if (melt < 0) melt = 0.0;
                    const double lw = (G[l] > 1) ? 0.0 : melt;
                    ice_total += lw * frac_ice[l];
                }
                liquid = snowmelt + ice_total;
            }
        }
        const double q = gr4j_step<UH, ICE ? GR4J_CONSTS_JIT_EXP : GR4J_CONSTS_JIT>(
            P, s, r, uh, liquid, day[3 * L]);
        if (active && (wq | ws)) {
            snow_out_ptr_t po =
                (snow_out_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(po));     // keeps the load at this spot
            SnowOut o;                       // one s_load_dwordx16 (+x2)
            o.qsim = po->qsim; o.G = po->G; o.eTG = po->eTG;
            o.s_store = po->s_store; o.r_store = po->r_store;
            o.sca = po->sca; o.icemelt = po->icemelt;
            o.snowmelt = po->snowmelt;
            const int64_t ld = po->ld;
            if (wq) rr_out(&o.qsim[t * ld + i], q);
            if (ws) {
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    rr_out(&o.G[(t * L + l) * ld + i], G[l]);
                    rr_out(&o.eTG[(t * L + l) * ld + i], eTG[l]);
                    if (HYST) rr_out(&o.sca[(t * L + l) * ld + i], sca[l]);
                }
                rr_out(&o.s_store[t * ld + i], s);
                rr_out(&o.r_store[t * ld + i], r);
                if (ICE) rr_out(&o.icemelt[t * ld + i], ice_total);
                if (HYST && ICE) rr_out(&o.snowmelt[t * ld + i], snowmelt);
            }
        }
        if (we) {
            const double d = day[D - 1] - q;   // the day's observation
            acc = __builtin_fma(d, d, acc);
        }
    };
    // (two copies of the time loop, see cemaneige.hip cemaneige_kernel)
    const bool sane_wave =
        HYST ? cema_hyst_wave_is_sane(gtresh, L, CTG, Kf, inv_Thacc, Rsp,
                                      snow_pack_init, thermal_state_init,
                                      sca_prev0)
             : cema_wave_is_sane(gtresh, L, CTG, Kf, snow_pack_init,
                                 thermal_state_init);
    if (sane_wave) {
        one_day(std::true_type{}, std::true_type{}, 0);
        for (int64_t t = 1; t < T; ++t)
            one_day(std::false_type{}, std::true_type{}, t);
    } else {
        one_day(std::true_type{}, std::false_type{}, 0);
        for (int64_t t = 1; t < T; ++t)
            one_day(std::false_type{}, std::false_type{}, t);
    }
    if (we && active) sse[i] = acc;
}

// ---- the reference's own GR4J sequence for the sets that are not civil ------
// (gr4j_reference.h)  One lane per set, launched behind the fast kernels; a
// civil set's lane returns at once.  Every other one runs the snow routine
// (and the ice melt) as the fast kernels do and the reference's own run_gr4j
// on their outflow (cemaneigehystgr4j_model.py:74-78,
// cemaneigegr4jice_model.py:88-92), and overwrites its columns and its score.
template <int L, bool HYST, bool ICE>
__global__ __launch_bounds__(RR_BLOCK) void snow_gr4j_reference_kernel(
    SnowOut o, const double *__restrict__ days,
    const double *__restrict__ gtresh, const double *__restrict__ frac_ice,
    int64_t T, double snow_pack_init, double thermal_state_init,
    double sca_init, double s_init, double r_init,
    const double *__restrict__ params, SnowParLayout lay, int64_t N,
    const int *__restrict__ plan, double *__restrict__ sse)
{
    const int64_t i = (int64_t)blockIdx.x * RR_BLOCK + threadIdx.x;
    if (i >= N) return;
    if (gr4j_plan_tier(plan[0], plan[1], 0, plan[2]) < 0) return;
    const double *p = params + i * lay.npar;
    bool snow_civil = gr4j_civil_snow_par(p[0]) && gr4j_civil_snow_par(p[1]) &&
                      gr4j_civil_snow_par(snow_pack_init) &&
                      gr4j_civil_snow_par(thermal_state_init) &&
                      gr4j_civil_snow_par(sca_init);
    if (HYST)
        snow_civil = snow_civil && gr4j_civil_snow_par(p[2]) &&
                     gr4j_civil_snow_par(p[3]);
    if (ICE) snow_civil = snow_civil && gr4j_civil_snow_par(p[lay.i_ddf]);
    if (plan[3] == 0 && snow_civil &&
        gr4j_civil_set(p[lay.i_x1], p[lay.i_x1 + 1], p[lay.i_x1 + 2], s_init,
                       r_init))
        return;
    Gr4jRef g;
    if (!g.init(p[lay.i_x1], p[lay.i_x1 + 1], p[lay.i_x1 + 2],
                p[lay.i_x1 + 3], s_init, r_init))
        return;
    const double CTG = p[0], Kf = p[1], omc = 1 - CTG;
    const double Rsp = HYST ? p[3] : 0.0;
    const InvDivisor inv_Thacc = make_inv_divisor(HYST ? p[2] : 1.0);
    const lanemask_t thacc_m = RR_LANES(inv_Thacc.ok);
    const double ddf = ICE ? p[lay.i_ddf] : 0.0;
    double G[L], eTG[L], sca[L], swe_max[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        G[l] = 0.0; eTG[l] = 0.0; sca[l] = 0.0; swe_max[l] = 0.0;
    }
    const double *psol = gtresh + L;
    const double sca_prev0 = (T == 1) ? sca_init : 0.0;
    constexpr int D = cema_record_len(L, true);
    double acc = 0.0;
    for (int64_t t = 0; t < T; ++t) {
        double day[D];
#pragma unroll
        for (int k = 0; k < D; ++k) day[k] = days[t * D + k];
        double snowmelt;
        if constexpr (HYST) {
            snowmelt = t == 0
                ? cema_hyst_day<L, true, false, true>(day, psol, snow_pack_init,
                                         thermal_state_init, sca_prev0, CTG,
                                         omc, Kf, inv_Thacc, thacc_m, Rsp, G,
                                         eTG, sca, swe_max)
                : cema_hyst_day<L, false, false, true>(day, psol, snow_pack_init,
                                          thermal_state_init, sca_prev0, CTG,
                                          omc, Kf, inv_Thacc, thacc_m, Rsp, G,
                                          eTG, sca, swe_max);
        } else {
            // (the reference's own snow day too: its outflow to the bit)
            snowmelt = t == 0
                ? cema_ref_day<L, true>(day, gtresh, snow_pack_init,
                                        thermal_state_init, CTG, Kf, G, eTG)
                : cema_ref_day<L, false>(day, gtresh, snow_pack_init,
                                         thermal_state_init, CTG, Kf, G, eTG);
        }
        double liquid = snowmelt;
        double ice_total = 0.0;
        if constexpr (ICE) {            // icemelt_model.py:55-63
#pragma unroll
            for (int l = 0; l < L; ++l) {
                double melt = ddf * day[2 * L + l];
                if (melt < 0) melt = 0.0;
                const double lw = (G[l] > 1) ? 0.0 : melt;
                ice_total += lw * frac_ice[l];
            }
            liquid = snowmelt + ice_total;
        }
        const double q = g.day(liquid, day[3 * L]);
        if (o.qsim) rr_out(&o.qsim[t * o.ld + i], q);
        if (o.G) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                rr_out(&o.G[(t * L + l) * o.ld + i], G[l]);
                rr_out(&o.eTG[(t * L + l) * o.ld + i], eTG[l]);
                if (HYST) rr_out(&o.sca[(t * L + l) * o.ld + i], sca[l]);
            }
            rr_out(&o.s_store[t * o.ld + i], g.s);
            rr_out(&o.r_store[t * o.ld + i], g.r);
            if (ICE) rr_out(&o.icemelt[t * o.ld + i], ice_total);
            if (HYST && ICE) rr_out(&o.snowmelt[t * o.ld + i], snowmelt);
        }
        if (sse) {
            const double d = day[D - 1] - q;   // the day's observation
            acc = __builtin_fma(d, d, acc);
        }
    }
    if (sse) sse[i] = acc;
}

// ---- more than RR_SNOWNEXT_REG_LAYERS elevation layers ----------------------
// The same day step with a run-time layer count (as cemaneige_dyn_kernel does
// for the plain snow routine): the per-layer states -- G, eTG and, with the
// hysteresis, sca and the pre-melt SWE maximum -- live in an HBM scratch
// [4][L][N] (lane-contiguous: every access is a coalesced 512-byte row per
// wave) instead of registers.  Rare, so it favours simplicity: plain `/`
// (bit-identical to the 3-FMA quotients by construction), no unrolling.
template <class UH, bool HYST, bool ICE>
__global__ __launch_bounds__(RR_BLOCK) void snow_gr4j_dyn_kernel(
    SnowOut /* read through the kernarg segment, see SnowOut */,
    const double *__restrict__ days, const double *__restrict__ gtresh,
    const double *__restrict__ frac_ice, int64_t T, int L,
    double snow_pack_init, double thermal_state_init, double sca_init,
    double s_init, double r_init, const double *__restrict__ params,
    SnowParLayout lay, int64_t N, const int *__restrict__ plan,
    int force_lds, int wq, int ws,
    double *__restrict__ state, const double *__restrict__ qobs,
    double *__restrict__ sse, double *__restrict__ uh_mem)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int n1cap, n2cap;
    if (!gr4j_plan_selects<UH, false>(plan, force_lds, n1cap, n2cap)) return;
    const int64_t i = (int64_t)blockIdx.x * RR_BLOCK + threadIdx.x;
    const bool active = i < N;
    // tail lanes of the last wave share (and rewrite identically) set N-1's
    // scratch column, like the register kernels recompute it
    const int64_t ii = active ? i : N - 1;
    const double *p = params + ii * lay.npar;
    const double CTG = p[0], Kf = p[1];
    const double Thacc = HYST ? p[2] : 1.0, Rsp = HYST ? p[3] : 0.0;
    const double ddf = ICE ? p[lay.i_ddf] : 0.0;
    Gr4jPar P;
    P.set(p[lay.i_x1], p[lay.i_x1 + 1], p[lay.i_x1 + 2], p[lay.i_x1 + 3]);
    const double omc = 1 - CTG;
    const int64_t plane = (int64_t)L * N;
    double *Gs = state + ii, *Es = Gs + plane, *Ss = Es + plane,
           *Ms = Ss + plane;
    const double *psol = gtresh + L;
    const double sca_prev0 = (T == 1) ? sca_init : 0.0;
    UH uh;
    gr4j_uh_init(uh, lds, uh_mem, n1cap, n2cap, P.x4);
    double s = s_init * P.x1, r = r_init * P.x3;
    double acc = 0.0;
    const bool we = sse != nullptr;
    const int D = cema_record_len(L, true);
    for (int64_t t = 0; t < T; ++t) {
        const double *day = days + t * D;
        const bool first = t == 0;
        SnowOut o = {};
        int64_t ld = 0;
        if (wq | ws) {
            snow_out_ptr_t po =
                (snow_out_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(po));
            o.qsim = po->qsim; o.G = po->G; o.eTG = po->eTG;
            o.s_store = po->s_store; o.r_store = po->r_store;
            o.sca = po->sca; o.icemelt = po->icemelt;
            o.snowmelt = po->snowmelt;
            ld = po->ld;
        }
        double c = 0.0, ice_total = 0.0;
        for (int l = 0; l < L; ++l) {
            const double snow = day[l], rain = day[L + l],
                         temp = day[2 * L + l];
            const int64_t at = (int64_t)l * N;
            double g, e;
            if (first) {
                g = snow_pack_init;
                e = thermal_state_init;
            } else {
                g = Gs[at] + snow;
                e = CTG * Es[at] + omc * temp;
            }
            if (e > 0) e = 0.0;
            double pot_melt = 0.0;
            if (e == 0 && temp > 0) {
                pot_melt = Kf * temp;
                if (pot_melt > g) pot_melt = g;
            }
            double melt, sc = 0.0;
            if constexpr (HYST) {           // cemaneigehyst_model.py:123-158
                double mx = first ? 0.0 : Ms[at];
                const double snow_balance = snow - pot_melt;
                if (snow_balance >= 0) {
                    const double prev = first ? sca_prev0 : Ss[at];
                    sc = prev + snow_balance / Thacc;
                    mx = nb_max(mx, g);
                } else {
                    const double Thmelt = psol[l] * Rsp;
                    const double Thmax = (mx > Thmelt) ? Thmelt : mx;
                    sc = (Thmax > 0) ? g / Thmax : 0.0;
                }
                sc = nb_min(nb_max(sc, 0.0), 1.0);
                melt = (0.9 * sc + 0.1) * pot_melt;
                melt = nb_min(melt, g);
                g = g - melt;
                if (g == 0) mx = 0.0;
                Ss[at] = sc;
                Ms[at] = mx;
            } else {                        // cemaneige_model.py:109-118
                const double gt = gtresh[l];
                const double ratio = (g < gt) ? g / gt : 1.0;
                melt = (0.9 * ratio + 0.1) * pot_melt;
                g = g - melt;
            }
            Gs[at] = g;
            Es[at] = e;
            if constexpr (ICE) {            // icemelt_model.py:55-63
                double im = ddf * temp;
                if (im < 0) im = 0.0;
                const double lw = (g > 1) ? 0.0 : im;
                ice_total += lw * frac_ice[l];
            }
            if (ws && active) {
                rr_out(&o.G[(t * L + l) * ld + i], g);
                rr_out(&o.eTG[(t * L + l) * ld + i], e);
                if (HYST) rr_out(&o.sca[(t * L + l) * ld + i], sc);
            }
            c += rain + melt;
        }
        const double snowmelt = c / (double)L;
        const double liquid = ICE ? snowmelt + ice_total : snowmelt;
        // (no reference kernel behind this one: the excess by select)
        const double q = gr4j_step<UH, ICE ? GR4J_CONSTS_JIT_EXP : GR4J_CONSTS_JIT,
                                   true>(P, s, r, uh, liquid, day[3 * L]);
        if (active) {
            if (wq) rr_out(&o.qsim[t * ld + i], q);
            if (ws) {
                rr_out(&o.s_store[t * ld + i], s);
                rr_out(&o.r_store[t * ld + i], r);
                if (ICE) rr_out(&o.icemelt[t * ld + i], ice_total);
                if (HYST && ICE) rr_out(&o.snowmelt[t * ld + i], snowmelt);
            }
        }
        if (we) {
            const double d = qobs[t] - q;
            acc = __builtin_fma(d, d, acc);
        }
    }
    if (we && active) sse[i] = acc;
}

template <bool HYST, bool ICE>
static int snow_gr4j_dev(const char *who, const double *prec,
                         const double *mean_temp, const double *etp,
                         const double *frac_ice, const double *frac_solid_prec,
                         int64_t T, int64_t L, double snow_pack_init,
                         double thermal_state_init, double sca_init,
                         double s_init, double r_init, const double *params,
                         int64_t N, double *qsim, double *G, double *eTG,
                         double *s_store, double *r_store, double *sca,
                         double *icemelt, double *snowmelt, int64_t ld,
                         const double *qobs, double *sse, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    int rc = rr_check_common(who, T, N, ld, params, qobs, sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (L < 1 || L > 100000) {
        rr_set_error("%s: %lld elevation layers", who, (long long)L);
        return RR_E_PARAM;
    }
    if (!prec || !mean_temp || !etp || !frac_solid_prec || (ICE && !frac_ice)) {
        rr_set_error("%s: NULL forcing pointer", who);
        return RR_E_NULL;
    }
    const int want = 4 + (HYST ? 1 : 0) + (ICE ? 1 : 0) + (HYST && ICE ? 1 : 0);
    const int got = (G != nullptr) + (eTG != nullptr) + (s_store != nullptr) +
                    (r_store != nullptr) + (HYST && sca != nullptr) +
                    (ICE && icemelt != nullptr) +
                    (HYST && ICE && snowmelt != nullptr);
    if (got != 0 && got != want) {
        rr_set_error("%s: pass all %d storage outputs or none", who, want);
        return RR_E_NULL;
    }
    if ((rc = rr_check_outputs(who, qsim, got != 0)) != RR_OK) return rc;
    if (!workspace || workspace_bytes < cema_ws_bytes(T, L, true, N, 4, RR_SNOWNEXT_REG_LAYERS)) {
        rr_set_error("%s: workspace too small", who);
        return RR_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    SnowParLayout lay;
    lay.npar = 6 + (HYST ? 2 : 0) + (ICE ? 1 : 0);
    lay.i_x1 = HYST ? 4 : 2;
    lay.i_ddf = lay.npar - 1;
    const int *d_plan = (const int *)workspace;
    // whatever lies behind the base workspace is unit-hydrograph scratch
    const size_t base_ws = cema_ws_bytes(T, L, true, N, 4, RR_SNOWNEXT_REG_LAYERS);
    double *uh_mem = (double *)((char *)workspace + base_ws);
    const int mem_cap = gr4j_mem_cap(workspace_bytes - base_ws, N);
    rc = rr_gr4j_plan_async(params, N, lay.npar, lay.i_x1 + 3,
                            (int *)workspace, mem_cap, st);
    if (rc != RR_OK) return rc;
    // a block the plan cannot run (no tier selected) writes nothing: its
    // scores then read NaN, not whatever the buffer held
    if (qobs && sse)
        RR_HIP(hipMemsetAsync(sse, 0xFF, (size_t)N * sizeof(double), st));
    const int force_lds = (int)rr_option(RR_OPT_GR4J_FORCE_LDS);
    double *days, *gt, *state;
    rc = rr_cema_prepass(prec, mean_temp, frac_solid_prec, etp,
                         (qobs && sse) ? qobs : nullptr, T, (int)L, workspace,
                         st, &days, &gt, &state, (int *)workspace + 3,
                         RR_SNOWNEXT_REG_LAYERS);
    if (rc != RR_OK) return rc;
    const dim3 grid((unsigned)rr_ceil_div(N, RR_BLOCK)), block(RR_BLOCK);
    const double *qo = (qobs && sse) ? qobs : nullptr;
    // every unit-hydrograph tier is enqueued; the kernels pick the one the
    // plan selects (gr4j_core.h)
    const size_t lds_bytes = GR4J_LDS_BYTES;
    const SnowOut out = {qsim, G, eTG, s_store, r_store, sca, icemelt,
                         snowmelt, ld};
    // (the couplings of this file keep up to RR_SNOWNEXT_REG_LAYERS = 5
    // layers -- Cemaneige's own five equal-area zones -- in registers; more
    // run from the HBM state scratch: per-layer kernels for L = 6..8 were 45
    // of the library's 525 kernels and 3.5 MB of its code for configurations
    // nobody has asked for; removed in round 6)
    if (L > RR_SNOWNEXT_REG_LAYERS) {
        gr4j_for_each_indexed_tier([&](auto uh) {
            using UH = decltype(uh);
            snow_gr4j_dyn_kernel<UH, HYST, ICE>
                <<<grid, block, std::is_same<UH, UhLds>::value ? lds_bytes : 0,
                   st>>>(out, days, gt, frac_ice, T, (int)L, snow_pack_init,
                         thermal_state_init, sca_init, s_init, r_init, params,
                         lay, N, d_plan, /*force_lds=*/1, qsim != nullptr,
                         G != nullptr, state, qo, sse, uh_mem);
        });
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    // A sweep that writes nothing but scores takes its sets in the order of
    // their ceil(x4): the waves then pick their own hydrograph tier
    // (gr4j_core.h gr4j_wave_selects).  The order lives where the tiled
    // kernels of cemaneige.hip keep their hand-over scratch (unused here).
    // From four waves per SIMD on: there every tier's share of the waves still
    // fills the GPU (1M sets under the hysteresis couplings' default bounds:
    // 146.8 -> 140.0 ms, hysteresis + ice 165.4 -> 156.5,
    // profiles/r05_n1_tiers_ab.txt); a smaller sweep keeps its order, and its
    // waves -- 64 sets drawn from the whole range of x4 -- the launch's tier.
    const int *perm = nullptr;
    const bool by_tier = !qsim && !G && qo && N < 0x7fffffff &&
                         rr_ceil_div(N, RR_BLOCK) >= 4 * (int64_t)rr_simd_count();
    if (by_tier) {
        int *bins = (int *)((char *)workspace + cema_tile_offset(T, L, true));
        int *pm = bins + 128;
        rc = rr_gr4j_tier_sort_async(params, N, lay.npar, lay.i_x1 + 3, bins,
                                     pm, st);
        if (rc != RR_OK) return rc;
        perm = pm;
    }
    // every tier's kernel on a stream of its own (the narrowest register tier
    // stays on the caller's: a launch inside the default bounds of the plain
    // GR4J family runs as it always has): with the waves choosing their tiers
    // all of them have work, and they are to share the GPU, not to take turns
    int join_rc = RR_OK;
    if (by_tier) {
        rc = rr_tier_fork(st);
        if (rc != RR_OK) return rc;
    }
    // (no early return between the fork and the join inside dispatch_layers)
    dispatch_layers<RR_SNOWNEXT_REG_LAYERS>((int)L, [&](auto LL) {
        gr4j_for_each_tier([&](auto uh) {
            using UH = decltype(uh);
            hipStream_t ts = st;
            if (std::is_same<UH, UhRegs<5>>::value) ts = rr_tier_stream(0);
            else if (std::is_same<UH, UhRegs<10>>::value) ts = rr_tier_stream(1);
            else if (std::is_same<UH, UhLds>::value) ts = rr_tier_stream(2);
            if (!ts || !by_tier) ts = st;
            snow_gr4j_kernel<LL.value, UH, HYST, ICE>
                <<<grid, block, std::is_same<UH, UhLds>::value ? lds_bytes : 0,
                   ts>>>(out, days, gt, frac_ice, T, snow_pack_init,
                         thermal_state_init, sca_init, s_init, r_init, params,
                         lay, N, d_plan, force_lds, qsim != nullptr,
                         G != nullptr, qo, sse, uh_mem, perm);
        });
        if (by_tier) join_rc = rr_tier_join(st);
        // ... and behind them the sets that are not civil
        // (gr4j_reference.h)
        snow_gr4j_reference_kernel<LL.value, HYST, ICE>
            <<<grid, block, 0, st>>>(
                out, days, gt, frac_ice, T, snow_pack_init,
                thermal_state_init, sca_init, s_init, r_init, params, lay, N,
                d_plan, qo ? sse : nullptr);
    });
    if (join_rc != RR_OK) return join_rc;
    RR_HIP(hipGetLastError());
    return RR_OK;
}
