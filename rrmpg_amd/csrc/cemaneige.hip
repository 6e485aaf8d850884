// cemaneige.hip -- Cemaneige snow routine and the fused Cemaneige->GR4J
// ensemble kernels for gfx950.
//
// Replaces run_cemaneige (reference: rrmpg/models/cemaneige_model.py:15-126),
// run_cemaneigegr4j (reference: rrmpg/models/cemaneigegr4j_model.py:16-63)
// and the per-set Python loops around them (reference:
// rrmpg/models/cemaneige.py:218-245, rrmpg/models/cemaneigegr4j.py:238-273).
//
// One lane per parameter set.  Per elevation layer the snow pack G and its
// thermal state eTG live in registers (L <= 5 layers, kernels are
// instantiated per L so the layer loop is fully unrolled).  Everything that
// does not depend on the parameters is hoisted out of the N-fold sweep by two
// small pre-pass kernels, with the very operations the reference performs
// per set:
//   cema_pack    snow = prec * frac, rain = prec - snow     (:76-77)
//   cema_gtresh  G_tresh[l] = 0.9 * 365.25 * mean_t(snow)   (:80), summed
//                strictly left to right like numba's np.mean.
// The per-day record {snow[L], rain[L], temp[L] (, etp)} is wave-uniform and
// arrives through the scalar cache.
//
// In the coupled kernel the layer-mean outflow of the snow routine feeds
// GR4J's precipitation in registers the same day; the [T] liquid-water
// intermediate of the reference (cemaneigegr4j_model.py:57-62) never exists.
#include <math.h>

#include <vector>

#include "snow_core.h"
#include "gr4j_reference.h"

// days: [T][D] doubles, D = cema_record_len(L, with_etp) (snow_core.h):
//   [0,L) snow, [L,2L) rain, [2L,3L) mean_temp, [3L] etp, then the day's
//   observation (cema_day_meta)
// `insane`: counts the values that rule out the SANE form of the snow routine
// (snow_core.h cema_day, snownext_kernels.h cema_hyst_day): a snowfall that is not
// a number in [0, 1e290], a temperature that is not finite (or beyond 1e300),
// a rain whose sign bit is set, a positive subnormal temperature.
// Zeroed by rr_cema_prepass before the launch.
__global__ void cema_pack(const double *__restrict__ prec,
                          const double *__restrict__ mean_temp,
                          const double *__restrict__ frac,
                          const double *__restrict__ etp, int64_t T, int L,
                          int D, double *__restrict__ days,
                          unsigned long long *__restrict__ insane,
                          int *__restrict__ uncivil)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T * L) return;
    const int64_t t = g / L;
    const int l = (int)(g - t * L);
    const double p = prec[g];
    const double snow = p * frac[g];            // cemaneige_model.py:76
    const double rain = p - snow;               // :77
    const double temp = mean_temp[g];
    double *d = days + t * D;
    d[l] = snow;
    d[L + l] = rain;
    d[2 * L + l] = temp;
    if (etp && l == 0) d[3 * L] = etp[t];
    // (... or a rain of -0 or below: on a day without melt the hysteresis
    // routine's outflow is the rain itself, snownext_kernels.h)
    if (!(snow >= 0.0 && snow <= 1e290) || !(fabs(temp) <= 1e300) ||
        __builtin_signbit(rain) || (temp > 0.0 && temp < 0x1p-1022))
        atomicAdd(insane, 1ull);
    // forcing the GR4J half's fast forms are not meant for
    // (gr4j_reference.h): with any, every set takes the reference's sequence
    if (uncivil &&
        (!gr4j_civil_forcing(snow) || !gr4j_civil_forcing(rain) ||
         !gr4j_civil_forcing(temp) ||
         (etp && l == 0 && !gr4j_civil_forcing(etp[t]))))
        atomicAdd(uncivil, 1);
}

// The trailing slot of every record: the day's observed discharge
// (snow_core.h).  One thread per day.
__global__ void cema_day_meta(double *__restrict__ days, int64_t T, int D,
                              const double *__restrict__ qobs)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    days[t * D + D - 1] = qobs ? qobs[t] : 0.0;
}

// One workgroup per layer: the wave stages 256 days of snow in LDS, lane 0
// adds them strictly in time order (numba's np.mean is a sequential sum).
__global__ __launch_bounds__(256) void cema_gtresh(
    const double *__restrict__ days, int64_t T, int D,
    double *__restrict__ gtresh)
{
    __shared__ double buf[256];
    const int l = blockIdx.x;
    double c = 0.0;
    for (int64_t t0 = 0; t0 < T; t0 += 256) {
        const int64_t t = t0 + threadIdx.x;
        buf[threadIdx.x] = (t < T) ? days[t * D + l] : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int n = (int)((T - t0 < 256) ? (T - t0) : 256);
            for (int k = 0; k < n; ++k) c += buf[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        gtresh[l] = 0.9 * 365.25 * (c / (double)T);       // :80
        // hysteresis variant: Psolannual = 365.25 * mean(snow)
        // (cemaneigehyst_model.py:93), kept next to it at [L + l]
        gtresh[gridDim.x + l] = 365.25 * (c / (double)T);
    }
}

// The CemaGt table {G_tresh, RN(1 / G_tresh)} of the register kernels
// (snow_core.h) and the flag "every threshold suits the 3-FMA quotient",
// behind G_tresh[L] and Psolannual[L].
__global__ void cema_gt_table(double *__restrict__ gtresh, int L, int nreg)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool all_ok = true;
    for (int l = 0; l < nreg; ++l) {
        const InvDivisor d = make_inv_divisor(gtresh[l]);
        gtresh[2 * L + 2 * l] = d.b;
        gtresh[2 * L + 2 * l + 1] = d.rb;
        all_ok = all_ok && d.ok;
    }
    gtresh[4 * L] = all_ok ? 1.0 : 0.0;
}

// TILED: the time axis in pieces, one workgroup per ticket (common.h RrTiles:
// million-set sweeps); handed over: both snow states of every layer and the
// score sum.
// GTR (sweeps of at most two waves per SIMD): the melt thresholds in VGPR
// pairs and the potential melt by select instead of a scalar load and an
// exec-masked block per melting layer -- nobody hides the load's latency
// there (125k sets, scores: 7.6 -> 6.9 ms); with a SIMD full of waves the
// scalar form is the faster one (1M sets 34.1 vs 38.8 ms).  Same bits.
template <int L, bool TILED = false, bool GTR = false>
__global__ __launch_bounds__(RR_BLOCK) void cemaneige_kernel(
    const double *__restrict__ days, const double *__restrict__ gtresh,
    int64_t T, double snow_pack_init, double thermal_state_init,
    const double *__restrict__ params, int64_t N,
    double *__restrict__ outflow, double *__restrict__ G_out,
    double *__restrict__ eTG_out, int64_t ld,
    const double *__restrict__ qobs, double *__restrict__ sse, RrTiles tiles)
{
    const int njobs = (int)((N + RR_BLOCK - 1) / RR_BLOCK);
    int job = blockIdx.x, piece = 0;
    if constexpr (TILED) {
        const int item = rr_tile_ticket(tiles);
        piece = __builtin_amdgcn_readfirstlane(item / njobs);
        job = __builtin_amdgcn_readfirstlane(item - piece * njobs);
    }
    const int64_t i = (int64_t)job * RR_BLOCK + threadIdx.x;
    const bool active = i < N;
    const double *p = params + (active ? i : N - 1) * 2;
    const double CTG = p[0], Kf = p[1];
    const double omc = 1 - CTG;
    double G[L], eTG[L];
#pragma unroll
    for (int l = 0; l < L; ++l) { G[l] = 0.0; eTG[l] = 0.0; }
    const cema_gt_ptr_t gt_tab = (cema_gt_ptr_t)(gtresh + 2 * L);
    const lanemask_t gt_ok = gtresh[4 * L] != 0.0 ? ~0ull : 0ull;
    CemaGtRegs<L> gt_regs;
    if (GTR) cema_gt_to_regs<L>(gt_tab, gt_regs);
    double acc = 0.0;
    const bool wq = outflow != nullptr, ws = G_out != nullptr,
               we = sse != nullptr;
    const int lane_off = threadIdx.x * 8;
    const int64_t first = (int64_t)job * RR_BLOCK;
    const unsigned row_bytes = rr_row_bytes(first, N);
    int t_begin = 0, t_end = (int)T;
    double *const hand = TILED ? tiles.state + ((int64_t)job * RR_BLOCK +
                                                threadIdx.x) : nullptr;
    const int64_t hs = (int64_t)njobs * RR_BLOCK;
    if constexpr (TILED) {
        rr_tile_range(0, (int)T, tiles.pieces, piece, 1, t_begin, t_end);
        t_begin = __builtin_amdgcn_readfirstlane(t_begin);
        t_end = __builtin_amdgcn_readfirstlane(t_end);
        if (piece > 0) {
            rr_tile_wait(tiles, job, piece);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                G[l] = hand[(2 * l) * hs];
                eTG[l] = hand[(2 * l + 1) * hs];
            }
            acc = hand[(2 * L) * hs];
        }
    }
    // one day; `is_first` (a std::bool_constant) marks day 0, which is peeled
    // off the time loop
    auto one_day = [&](auto is_first, auto sane, int64_t t) {
        // the whole day record by value, up front: one wide scalar load and
        // one wait per day (read through the pointer, hipcc fetches every
        // field at its use site with its own s_load + wait)
        constexpr int D = cema_record_len(L, false);
        double rec[D];
#pragma unroll
        for (int k = 0; k < D; ++k) rec[k] = days[t * D + k];
        const double q =
            cema_day<L, decltype(is_first)::value, GTR,
                     decltype(sane)::value>(
                rec, gt_tab, gt_ok, snow_pack_init, thermal_state_init, CTG,
                omc, Kf, G, eTG, GTR ? &gt_regs : nullptr);
        // output rows: wave-uniform base + lane offset (common.h
        // rr_store_row): no per-lane address arithmetic, no exec masking of
        // the tail wave, and -- unlike eleven strength-reduced row pointers --
        // nothing to advance on the days and in the modes that store nothing
        if (wq) rr_store_row(outflow + first + t * ld, row_bytes, lane_off, q);
        if (ws) {
            // (opaque day index: the addresses are formed here, on the days
            // and in the mode that stores, instead of as eleven induction
            // pointers advanced every day)
            int64_t ts = __builtin_amdgcn_readfirstlane((int)t);
            asm volatile("" : "+s"(ts));
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int64_t at = first + (ts * L + l) * ld;
                rr_store_row(G_out + at, row_bytes, lane_off, G[l]);
                rr_store_row(eTG_out + at, row_bytes, lane_off, eTG[l]);
            }
        }
        // (unconditional: as `if (we)` on the run-time flag hipcc computes it
        // anyway and then selects, two VOP3 selects a day; the record's
        // observation slot holds 0 when no score is asked for, and the sum
        // is only stored when one is)
        const double d = rec[D - 1] - q;         // the day's observation
        acc = __builtin_fma(d, d, acc);
    };
    // the time loop exists twice: for waves that may use the SANE form of
    // the snow routine (any sane run) and for the rest
    // (a piece that starts at day 0 peels it; the others start mid-run)
    const bool from_start = t_begin == 0 && t_begin < t_end;
    // The day count is compared in 32 bits: there is no 64-bit signed scalar
    // compare, and the vector one (v_mov_b64 + v_cmp_ge_i64) costs the loop
    // two vector instructions a day -- 1M sets 29.7 -> 28.6 ms, scores 27.0
    // -> 25.8.  (Not in the small-sweep form: 125k sets 6.49 -> 6.62 ms with
    // the scalar compare, measured twice.)
    auto more = [&](int64_t t) __attribute__((always_inline)) {
        if constexpr (GTR) return t < (int64_t)t_end;
        else return (int)t < t_end;
    };
    if (cema_wave_is_sane(gtresh, L, CTG, Kf, snow_pack_init,
                          thermal_state_init)) {
        if (from_start) one_day(std::true_type{}, std::true_type{}, 0);
        for (int64_t t = t_begin + (from_start ? 1 : 0); more(t); ++t)
            one_day(std::false_type{}, std::true_type{}, t);
    } else {
        if (from_start) one_day(std::true_type{}, std::false_type{}, 0);
        for (int64_t t = t_begin + (from_start ? 1 : 0); more(t); ++t)
            one_day(std::false_type{}, std::false_type{}, t);
    }
    if (TILED && piece + 1 < tiles.pieces) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
            hand[(2 * l) * hs] = G[l];
            hand[(2 * l + 1) * hs] = eTG[l];
        }
        hand[(2 * L) * hs] = acc;
        rr_tile_publish(tiles, job, piece);
    } else {
        if (we && active) sse[i] = acc;
    }
}

// Small configurations (<= 5 layers, unit hydrographs in 3+7 registers or in
// LDS) are held at 128 registers = 4 waves per SIMD: a handful of spills cost
// less than the lost wave (137 -> 130 ms at L = 5).
// (The many-waves kernel executes 38 lane moves a day -- scalar values parked
// in VGPR lanes -- at four waves per SIMD; round 5 measured what giving it
// registers instead costs: 1M sets, scores, 72.6 ms as shipped, 78.1 at three
// waves per SIMD, 74.7 with the constants in VGPRs as well, 73.1 for an
// optimistic many-waves form, 77.9 / 82.7 for the small-sweep forms;
// profiles/r05_fused_forms_ab.txt)
#define COUPLED_BIG_MINWAVES 4
template <int L, class UH, bool SMALL = false>
constexpr int coupled_min_waves()
{
    return (!SMALL && (std::is_same<UH, UhRegs<3>>::value ||
                       uh_is_indexed<UH>))
               ? (std::is_same<UH, UhRegs<3>>::value ? COUPLED_BIG_MINWAVES : 4)
               : 2;
}

// SMALL variant of the fused kernel, for sweeps of at most two waves per SIMD
// (<= 131,072 sets: one GPU's shard of BASELINE configs[3], and every `fit`
// population): occupancy is not a concern there and exposed latencies are,
// so the polynomial constants and the melt thresholds sit in VGPRs and the
// only scalar load left in the time loop is the prefetched day record.
// Instantiated for the model's own five layers, where it was measured (fewer
// layers run the many-waves form at any size), and where the register file
// allows it.
template <int L, class UH>
constexpr bool coupled_has_small()
{
    return L == 5 && (std::is_same<UH, UhRegs<3>>::value ||
                      std::is_same<UH, UhRegs<5>>::value);
}

// Output pointers.  Passed as the FIRST kernel argument and never touched by
// name: on the days something is stored the kernel re-reads the struct from
// offset 0 of its kernarg segment with one scalar load.  As ordinary
// arguments the five pointers + ld would sit in 12 SGPRs for the whole time
// loop of a kernel that is already short of them (the overflow goes to VGPR
// lanes and every use then costs a v_readlane, i.e. a VALU slot).
struct CoupledOut {
    double *qsim, *G, *eTG, *s_store, *r_store;
    int64_t ld;
};
typedef const CoupledOut __attribute__((address_space(4))) *coupled_out_ptr_t;

typedef const double __attribute__((address_space(4))) *cema_rec_ptr_t;
// (A time-tiled form of this kernel was measured slower -- 1M sets 89.8 -> 99.9
// ms: 168 VGPRs, three waves per SIMD -- and removed in round 6.)
template <int L, class UH, bool SMALL = false>
__global__ __launch_bounds__(RR_BLOCK, (coupled_min_waves<L, UH, SMALL>())) void
cemaneigegr4j_kernel(
    CoupledOut /* read through the kernarg segment, see above */,
    const double *__restrict__ days, const double *__restrict__ gtresh,
    int64_t T, double snow_pack_init, double thermal_state_init,
    double s_init, double r_init, const double *__restrict__ params,
    int64_t N, const int *__restrict__ plan, int force_lds, int wq, int ws,
    const double *__restrict__ qobs, double *__restrict__ sse,
    double *__restrict__ uh_mem, int warm)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int n1cap, n2cap;
    if (!gr4j_plan_selects<UH>(plan, force_lds, n1cap, n2cap)) return;
    if (warm)
        rr_warm_l2(days, (T + 1) * (int64_t)cema_record_len(L, true) * 8);
    const int64_t i = (int64_t)blockIdx.x * RR_BLOCK + threadIdx.x;
    const bool active = i < N;
    const double *p = params + (active ? i : N - 1) * 6;
    const double CTG = p[0], Kf = p[1];
    Gr4jPar P;
    P.set(p[2], p[3], p[4], p[5]);
    const double omc = 1 - CTG;
    double G[L], eTG[L];
#pragma unroll
    for (int l = 0; l < L; ++l) { G[l] = 0.0; eTG[l] = 0.0; }
    const cema_gt_ptr_t gt_tab = (cema_gt_ptr_t)(gtresh + 2 * L);
    const lanemask_t gt_ok = gtresh[4 * L] != 0.0 ? ~0ull : 0ull;
    UH uh;
    gr4j_uh_init(uh, lds, uh_mem, n1cap, n2cap, P.x4);
    double s = s_init * P.x1, r = r_init * P.x3;
    double acc = 0.0;
    const bool we = sse != nullptr;
    constexpr int D = cema_record_len(L, true);
    // The day's record (wave-uniform, SGPRs).  The snow routine is its only
    // reader apart from the two trailing slots, so the NEXT day's record is
    // requested into the registers that fall free after it -- in the middle
    // of the GR4J day, once that step's constant tables are through (they
    // need the SGPRs) -- and arrives while the rest of the day runs.
    // Requested at the top of the day, every wave would sit out the
    // scalar-load latency once per day, which a sweep of one or two waves per
    // SIMD cannot hide (measured: 65k sets 16.2 -> 14.0 ms, 125k 18.3 -> 17.9,
    // a million unchanged; the hysteresis / ice kernels of snownext_kernels.h,
    // shorter of SGPRs still, lose with it and keep the load at the top).
    CemaGtRegs<L> gt_regs;
    if constexpr (SMALL) cema_gt_to_regs<L>(gt_tab, gt_regs);
    constexpr int CONSTS = SMALL ? GR4J_CONSTS_VGPR : GR4J_CONSTS_JIT;
    const cema_rec_ptr_t drec = (cema_rec_ptr_t)days;
    double day[D];
#pragma unroll
    for (int k = 0; k < D; ++k) day[k] = drec[k];
    // one day; `first` (a std::bool_constant) marks day 0, which is peeled
    // off the time loop
    auto one_day = [&](auto first, auto sane, int64_t t) {
        const double liquid =
            cema_day<L, decltype(first)::value, SMALL, decltype(sane)::value,
                     true>(
                day, gt_tab, gt_ok, snow_pack_init, thermal_state_init, CTG,
                omc, Kf, G, eTG, &gt_regs);
        const double etp_t = day[3 * L], qobs_t = day[D - 1];
        auto fetch_next = [&]() {
            // (day T-1 requests the spare record behind the last one)
            cema_rec_ptr_t nx = drec + (t + 1) * D;
            asm volatile("" : "+s"(nx));     // keeps the loads at this spot
#pragma unroll
            for (int k = 0; k < D; ++k) day[k] = nx[k];
        };
        const double q =
            gr4j_step<UH, CONSTS>(P, s, r, uh, liquid, etp_t, fetch_next);
        // (per-lane addresses here: row stores through a buffer descriptor,
        // as in cemaneige_kernel, cost this kernel 1.5-2.5 % -- four more
        // SGPRs it does not have)
        if (active && (wq | ws)) {
            coupled_out_ptr_t po =
                (coupled_out_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(po));     // keeps the load at this spot
            CoupledOut o;                    // one s_load_dwordx16
            o.qsim = po->qsim; o.G = po->G; o.eTG = po->eTG;
            o.s_store = po->s_store; o.r_store = po->r_store;
            const int64_t ld = po->ld;
            if (wq) rr_out(&o.qsim[t * ld + i], q);
            if (ws) {
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    rr_out(&o.G[(t * L + l) * ld + i], G[l]);
                    rr_out(&o.eTG[(t * L + l) * ld + i], eTG[l]);
                }
                rr_out(&o.s_store[t * ld + i], s);
                rr_out(&o.r_store[t * ld + i], r);
            }
        }
        // (as a branch on the run-time flag; unconditional -- as in
        // cemaneige_kernel -- measured 0.5 % slower here)
        if (we) {
            const double d = qobs_t - q;   // the day's observation
            acc = __builtin_fma(d, d, acc);
        }
    };
    // (two copies of the time loop, see cemaneige_kernel)
    if (cema_wave_is_sane(gtresh, L, CTG, Kf, snow_pack_init,
                          thermal_state_init)) {
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

// ---- optimistic variant of the fused kernel -----------------------------------
// gr4j.hip gr4j_opt_kernel's idea for the GR4J half of the coupled day: its
// nine votes only note their lanes, the half runs straight through, and one
// branch at its end redoes it -- from the untouched start state, every vote
// decided on the spot as in cemaneigegr4j_kernel -- if a vote failed.  Both
// stores and the hydrograph slots exist in two generations; a day reads one
// and writes the other, two days per trip.  The snow routine in front of it
// stays as it is, in place (measured: with its six votes optimistic as well
// -- two generations of the ten snow states, the day's record fetched again
// for the redo -- the kernel got 4-9 % slower, not faster).  Bit-identical to
// cemaneigegr4j_kernel by construction.
template <class UH>
struct Gr4jGen {
    double s, r;
    typename UH::Slots u;
};

template <int L, class UH>
constexpr bool coupled_has_optimistic()
{
    return coupled_has_small<L, UH>();
}

// The small-sweep form only (polynomial constants and melt thresholds in
// VGPRs, two waves per SIMD): the many-waves form with an optimistic GR4J
// half was measured slower than the careful kernel (97 vs 90 ms at a million
// sets) and removed in round 6.
template <int L, class UH>
__global__ __launch_bounds__(RR_BLOCK, 2) void
cemaneigegr4j_opt_kernel(
    CoupledOut /* read through the kernarg segment, see above */,
    const double *__restrict__ days, const double *__restrict__ gtresh,
    int64_t T, double snow_pack_init, double thermal_state_init,
    double s_init, double r_init, const double *__restrict__ params,
    int64_t N, const int *__restrict__ plan, int force_lds, int wq, int ws,
    const double *__restrict__ qobs, double *__restrict__ sse, int warm)
{
    int n1cap, n2cap;
    if (!gr4j_plan_selects<UH>(plan, force_lds, n1cap, n2cap)) return;
    if (warm)
        rr_warm_l2(days, (T + 1) * (int64_t)cema_record_len(L, true) * 8);
    const int64_t i = (int64_t)blockIdx.x * RR_BLOCK + threadIdx.x;
    const bool active = i < N;
    const double *p = params + (active ? i : N - 1) * 6;
    const double CTG = p[0], Kf = p[1];
    Gr4jPar P;
    P.set(p[2], p[3], p[4], p[5]);
    const double omc = 1 - CTG;
    double G[L], eTG[L];
#pragma unroll
    for (int l = 0; l < L; ++l) { G[l] = 0.0; eTG[l] = 0.0; }
    const cema_gt_ptr_t gt_tab = (cema_gt_ptr_t)(gtresh + 2 * L);
    const lanemask_t gt_ok = gtresh[4 * L] != 0.0 ? ~0ull : 0ull;
    typedef Gr4jGen<UH> Gen;
    Gen A, B;
    UH uh;
    uh.init(P.x4, A.u);
    A.s = s_init * P.x1;
    A.r = r_init * P.x3;
    double acc = 0.0;
    const bool we = sse != nullptr;
    constexpr int D = cema_record_len(L, true);
    CemaGtRegs<L> gt_regs;
    cema_gt_to_regs<L>(gt_tab, gt_regs);
    constexpr int CONSTS = GR4J_CONSTS_VGPR;
    const cema_rec_ptr_t drec = (cema_rec_ptr_t)days;
    double day[D];
#pragma unroll
    for (int k = 0; k < D; ++k) day[k] = drec[k];
    // (always_inline: with its three call sites per copy of the time loop
    // the inliner would otherwise leave the day as a FUNCTION, its states
    // and the day record passed through memory)
    auto one_day = [&](auto first, auto sane, const Gen &in, Gen &out,
                       int64_t t) __attribute__((always_inline)) {
        const double liquid =
            cema_day<L, decltype(first)::value, true, decltype(sane)::value,
                     true>(
                day, gt_tab, gt_ok, snow_pack_init, thermal_state_init, CTG,
                omc, Kf, G, eTG, &gt_regs);
        const double etp_t = day[3 * L], qobs_t = day[D - 1];
        auto fetch_next = [&]() {
            // (day T-1 requests the spare record behind the last one)
            cema_rec_ptr_t nx = drec + (t + 1) * D;
            asm volatile("" : "+s"(nx));     // keeps the loads at this spot
#pragma unroll
            for (int k = 0; k < D; ++k) day[k] = nx[k];
        };
        const bool wet = liquid >= etp_t;                   // gr4j_model.py:89
        // liquid - etp on a wet day, etp - liquid = -(liquid - etp) on a dry
        // one (:90, :102): the magnitude of one difference (gr4j_core.h
        // gr4j_step)
        const double net = fabs(liquid - etp_t);
        const lanemask_t net_m = gr4j_num_lanes(net);
        OptimisticVotes votes;
        double s = in.s, r = in.r;
        double p_r;
        p_r = gr4j_production<UH, CONSTS>(P, s, net, wet, net_m,
                                          fetch_next, votes);
        double q = gr4j_routing<UH>(P, r, uh, in.u, out.u, p_r, votes);
        if (RR_VOTES_FAILED(votes)) {
            // some lane left a fast form's domain: the GR4J day again from
            // its untouched start state, every vote decided on the spot
            asm volatile("");
            s = in.s;
            r = in.r;
            p_r = gr4j_production<UH, CONSTS>(P, s, net, wet, net_m);
            q = gr4j_routing<UH>(P, r, uh, in.u, out.u, p_r);
        }
        out.s = s;
        out.r = r;
        if (active && (wq | ws)) {
            coupled_out_ptr_t po =
                (coupled_out_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(po));     // keeps the load at this spot
            CoupledOut o;                    // one s_load_dwordx16
            o.qsim = po->qsim; o.G = po->G; o.eTG = po->eTG;
            o.s_store = po->s_store; o.r_store = po->r_store;
            const int64_t ld = po->ld;
            if (wq) rr_out(&o.qsim[t * ld + i], q);
            if (ws) {
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    rr_out(&o.G[(t * L + l) * ld + i], G[l]);
                    rr_out(&o.eTG[(t * L + l) * ld + i], eTG[l]);
                }
                rr_out(&o.s_store[t * ld + i], s);
                rr_out(&o.r_store[t * ld + i], r);
            }
        }
        if (we) {
            const double d = qobs_t - q;   // the day's observation
            acc = __builtin_fma(d, d, acc);
        }
    };
    auto run = [&](auto sane) __attribute__((always_inline)) {
        one_day(std::true_type{}, sane, A, B, 0);        // day 0, peeled
        // (the day count compared in 32 bits: there is no 64-bit signed
        // scalar compare, and the vector one costs two instructions a day;
        // the launcher rejects T >= 2^31)
        const int Ti = (int)T;
        for (int64_t t = 1; (int)t < Ti; t += 2) {
            one_day(std::false_type{}, sane, B, A, t);
            if ((int)t + 1 < Ti)
                one_day(std::false_type{}, sane, A, B, t + 1);
        }
    };
    // (two copies of the time loop, see cemaneige_kernel)
    if (cema_wave_is_sane(gtresh, L, CTG, Kf, snow_pack_init,
                          thermal_state_init))
        run(std::true_type{});
    else
        run(std::false_type{});
    if (we && active) sse[i] = acc;
}

// (A wave-specialised form -- snow routine and optimistic GR4J day in two
// waves of a workgroup, the outflow through an LDS ring, 128 VGPRs, 0 lane
// moves -- was built in round 5, is bit-identical and no faster: the fused day
// is bound by the issue of its vector instructions, not by scalar loads or
// lane moves.  Removed in round 6; profiles/r05_fused_pipe_ab.txt.)

// ---- more than RR_CEMANEIGE_MAX_LAYERS elevation layers ----------------------
// Same day step with a run-time layer count; the per-layer snow states live
// in an HBM scratch [2][L][N] (lane-contiguous, so every access is a coalesced
// 512-byte row per wave) instead of registers.  Rare (Cemaneige is defined
// with 5 layers), so it favours simplicity: plain `/`, no unrolling.
__device__ __forceinline__ double cema_day_dyn(
    const double *__restrict__ day, const double *__restrict__ gtresh, int L,
    bool first, double snow_pack_init, double thermal_state_init, double CTG,
    double one_minus_CTG, double Kf, double *__restrict__ Gs,
    double *__restrict__ Es, int64_t stride, double *__restrict__ G_out,
    double *__restrict__ eTG_out, int64_t out_stride, bool write)
{
    double c = 0.0;
    for (int l = 0; l < L; ++l) {
        const double snow = day[l], rain = day[L + l], temp = day[2 * L + l];
        double g, e;
        if (first) {
            g = snow_pack_init;
            e = thermal_state_init;
        } else {
            g = Gs[l * stride] + snow;
            e = CTG * Es[l * stride] + one_minus_CTG * temp;
        }
        if (e > 0) e = 0.0;
        double pot_melt = 0.0;
        if (e == 0 && temp > 0) {
            pot_melt = Kf * temp;
            if (pot_melt > g) pot_melt = g;
        }
        const double gt = gtresh[l];
        const double ratio = (g < gt) ? g / gt : 1.0;
        const double melt = (0.9 * ratio + 0.1) * pot_melt;
        g = g - melt;
        Gs[l * stride] = g;
        Es[l * stride] = e;
        if (write) {
            rr_out(&G_out[l * out_stride], g);
            rr_out(&eTG_out[l * out_stride], e);
        }
        c += rain + melt;
    }
    return c / (double)L;
}

// UH = void: snow routine only; otherwise the fused Cemaneige -> GR4J sweep.
template <class UH>
__global__ __launch_bounds__(RR_BLOCK) void cemaneige_dyn_kernel(
    const double *__restrict__ days, const double *__restrict__ gtresh,
    int64_t T, int L, int D, double snow_pack_init, double thermal_state_init,
    double s_init, double r_init, const double *__restrict__ params, int npar,
    int64_t N, const int *__restrict__ plan, int force_lds,
    double *__restrict__ state,
    double *__restrict__ qsim, double *__restrict__ G_out,
    double *__restrict__ eTG_out, double *__restrict__ s_store,
    double *__restrict__ r_store, int64_t ld,
    const double *__restrict__ qobs, double *__restrict__ sse,
    double *__restrict__ uh_mem)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr bool coupled = !std::is_same<UH, void>::value;
    int n1cap = 0, n2cap = 0;
    if constexpr (coupled) {
        if (!gr4j_plan_selects<UH, false>(plan, force_lds, n1cap, n2cap)) return;
    }
    const int64_t i = (int64_t)blockIdx.x * RR_BLOCK + threadIdx.x;
    const bool active = i < N;
    const int64_t ii = active ? i : N - 1;
    const double *p = params + ii * npar;
    const double CTG = p[0], Kf = p[1];
    const double omc = 1 - CTG;
    // tail lanes of the last wave share (and rewrite identically) set N-1's
    // scratch column, like the register kernels recompute it
    double *Gs = state + ii, *Es = state + (int64_t)L * N + ii;
    Gr4jPar P;
    typename std::conditional<coupled, UH, int>::type uh;
    double s = 0.0, r = 0.0;
    if constexpr (coupled) {
        P.set(p[2], p[3], p[4], p[5]);
        gr4j_uh_init(uh, lds, uh_mem, n1cap, n2cap, P.x4);
        s = s_init * P.x1;
        r = r_init * P.x3;
    }
    double acc = 0.0;
    const bool wq = qsim != nullptr, ws = G_out != nullptr, we = sse != nullptr;
    for (int64_t t = 0; t < T; ++t) {
        const double *day = days + t * D;
        double q = cema_day_dyn(day, gtresh, L, t == 0, snow_pack_init,
                                thermal_state_init, CTG, omc, Kf, Gs, Es, N,
                                ws ? G_out + (t * L) * ld + i : nullptr,
                                ws ? eTG_out + (t * L) * ld + i : nullptr, ld,
                                ws && active);
        if constexpr (coupled) q = gr4j_step<UH, GR4J_CONSTS_JIT, true>(P, s, r, uh, q, day[3 * L]);
        if (active) {
            if (wq) rr_out(&qsim[t * ld + i], q);
            if (coupled && ws) {
                rr_out(&s_store[t * ld + i], s);
                rr_out(&r_store[t * ld + i], r);
            }
        }
        if (we) {
            const double d = qobs[t] - q;
            acc = __builtin_fma(d, d, acc);
        }
    }
    if (we && active) sse[i] = acc;
}

// workspace: [0,256) x4 scan | G_tresh[L], Psolannual[L] | days |
// (L > 8: state[2][L][N])
extern "C" size_t rr_cemaneige_workspace_bytes(int64_t T, int64_t L, int64_t N)
{
    return cema_ws_bytes(T, L, false, N);
}

extern "C" size_t rr_cemaneigegr4j_workspace_bytes(int64_t T, int64_t L,
                                                   int64_t N)
{
    return cema_ws_bytes(T, L, true, N);
}

extern "C" size_t rr_cemaneigegr4j_workspace_bytes_x4(int64_t T, int64_t L,
                                                      int64_t N, double max_x4)
{
    return cema_ws_bytes(T, L, true, N) + rr_gr4j_uh_scratch_bytes(N, max_x4);
}

int rr_cema_prepass(const double *prec, const double *mean_temp,
                        const double *frac, const double *etp,
                        const double *qobs, int64_t T, int L, void *workspace,
                        hipStream_t st, double **days_out, double **gt_out,
                        double **state_out, int *uncivil, int reg_layers)
{
    const int D = cema_record_len(L, etp != nullptr);
    double *gt = (double *)((char *)workspace + 512);
    double *days = (double *)((char *)workspace + 512 + cema_gt_bytes(L));
    unsigned long long *insane = (unsigned long long *)(gt + 4 * L + 1);
    RR_HIP(hipMemsetAsync(insane, 0, sizeof(*insane), st));
    hipLaunchKernelGGL(cema_pack, dim3((unsigned)rr_ceil_div(T * L, 256)),
                       dim3(256), 0, st, prec, mean_temp, frac, etp, T, L, D,
                       days, insane, uncivil);
    hipLaunchKernelGGL(cema_day_meta, dim3((unsigned)rr_ceil_div(T, 256)),
                       dim3(256), 0, st, days, T, D, qobs);
    hipLaunchKernelGGL(cema_gtresh, dim3((unsigned)L), dim3(256), 0, st, days,
                       T, D, gt);
    hipLaunchKernelGGL(cema_gt_table, dim3(1), dim3(1), 0, st, gt, L,
                       L <= reg_layers ? L : 0);
    *days_out = days;
    *gt_out = gt;
    *state_out = (double *)((char *)days + cema_days_bytes(T, L, etp != nullptr));
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---- forcing preprocessing on the device -------------------------------------
// rrmpg/models/cemaneige_utils.py of the reference: extrapolate_precipitation
// (:100-158), extrapolate_temperature (:160-207), calculate_solid_fraction
// (:15-98) -- one thread per (day, layer), the same fp64 operations in the
// same order; the per-layer constants (precipitation factor, temperature
// offset, which solid-fraction rule) come from the host.
struct CemaLayerConst {
    double prec_factor;    // exp(...) of :143/:151, or 1 (:156)
    double temp_delta;     // (z - z_station) * -0.0065 (:200-205)
    int low;               // z < 1500 m: min/max rule (:52-96)
    int pad;
};

__global__ void cema_layers_kernel(const double *__restrict__ prec,
                                   const double *__restrict__ mean_temp,
                                   const double *__restrict__ min_temp,
                                   const double *__restrict__ max_temp,
                                   int64_t T, int L,
                                   const CemaLayerConst *__restrict__ lc,
                                   double *__restrict__ layer_prec,
                                   double *__restrict__ layer_mean,
                                   double *__restrict__ frac)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T * L) return;
    const int64_t t = g / L;
    const int l = (int)(g - t * L);
    const CemaLayerConst c = lc[l];
    const double tmean = mean_temp[t] + c.temp_delta;
    const double tmin = min_temp[t] + c.temp_delta;
    const double tmax = max_temp[t] + c.temp_delta;
    double f;
    if (c.low) {
        if (tmax <= 0) f = 1.0;
        else if (tmin >= 0) f = 0.0;
        else f = 1 - (tmax / (tmax - tmin));
    } else {
        if (tmean >= 3) f = 0.0;
        else if (tmean <= 0) f = 1.0;
        else f = 1 - (tmean + 1) / 4;
    }
    layer_prec[g] = prec[t] * c.prec_factor;
    layer_mean[g] = tmean;
    frac[g] = f;
}

extern "C" size_t rr_cemaneige_layers_workspace_bytes(int64_t L)
{
    if (L < 1) L = 1;
    return rr_align256((size_t)L * sizeof(CemaLayerConst));
}

extern "C" int rr_cemaneige_layers_dev(
    const double *prec, const double *mean_temp, const double *min_temp,
    const double *max_temp, int64_t T, const double *altitudes, int64_t L,
    double met_station_height, const double *prec_factor, double *layer_prec,
    double *layer_mean_temp, double *frac_solid_prec, void *workspace,
    size_t workspace_bytes, void *stream)
{
    const char *who = "rr_cemaneige_layers_dev";
    if (T < 0 || L < 1 || L > 4096) {
        rr_set_error("%s: T=%lld, L=%lld", who, (long long)T, (long long)L);
        return RR_E_SIZE;
    }
    if (T == 0) return RR_OK;
    if (!prec || !mean_temp || !min_temp || !max_temp || !altitudes ||
        !layer_prec || !layer_mean_temp || !frac_solid_prec) {
        rr_set_error("%s: NULL pointer", who);
        return RR_E_NULL;
    }
    if (!workspace || workspace_bytes < rr_cemaneige_layers_workspace_bytes(L)) {
        rr_set_error("%s: workspace too small", who);
        return RR_E_WORKSPACE;
    }
    std::vector<CemaLayerConst> lc((size_t)L);
    for (int64_t l = 0; l < L; ++l) {
        const double z = altitudes[l];
        double factor;                               // cemaneige_utils.py:139-156
        if (prec_factor) factor = prec_factor[l];
        else if (z <= 4000) factor = exp((z - met_station_height) * 0.0004);
        else if (met_station_height <= 4000)
            factor = exp((4000 - met_station_height) * 0.0004);
        else factor = 1.0;
        lc[(size_t)l].prec_factor = factor;
        lc[(size_t)l].temp_delta = (z - met_station_height) * -0.0065;
        lc[(size_t)l].low = z < 1500;
        lc[(size_t)l].pad = 0;
    }
    hipStream_t st = (hipStream_t)stream;
    // (pageable source: the runtime stages the L records before returning)
    RR_HIP(hipMemcpyAsync(workspace, lc.data(), (size_t)L * sizeof(CemaLayerConst),
                          hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(cema_layers_kernel,
                       dim3((unsigned)rr_ceil_div(T * L, 256)), dim3(256), 0,
                       st, prec, mean_temp, min_temp, max_temp, T, (int)L,
                       (const CemaLayerConst *)workspace, layer_prec,
                       layer_mean_temp, frac_solid_prec);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

static int cema_check_layers(const char *who, int64_t L)
{
    if (L < 1 || L > 4096) {
        rr_set_error("%s: %lld elevation layers; supported: 1..4096", who,
                     (long long)L);
        return RR_E_PARAM;
    }
    return RR_OK;
}

extern "C" int rr_cemaneige_simulate_dev(
    const double *prec, const double *mean_temp, const double *frac_solid_prec,
    int64_t T, int64_t L, double snow_pack_init, double thermal_state_init,
    const double *params, int64_t N, double *outflow, double *G, double *eTG,
    int64_t ld, const double *qobs, double *sse, void *workspace,
    size_t workspace_bytes, void *stream)
{
    int rc = rr_check_common("rr_cemaneige_simulate_dev", T, N, ld, params,
                             qobs, sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    if ((rc = cema_check_layers("rr_cemaneige_simulate_dev", L)) != RR_OK)
        return rc;
    if (!prec || !mean_temp || !frac_solid_prec) {
        rr_set_error("rr_cemaneige_simulate_dev: NULL forcing pointer");
        return RR_E_NULL;
    }
    if ((G == nullptr) != (eTG == nullptr)) {
        rr_set_error("rr_cemaneige_simulate_dev: pass both G and eTG or none");
        return RR_E_NULL;
    }
    if ((rc = rr_check_outputs("rr_cemaneige_simulate_dev", outflow,
                               G != nullptr)) != RR_OK)
        return rc;
    if (!workspace || workspace_bytes < cema_ws_bytes(T, L, false, N)) {
        rr_set_error("rr_cemaneige_simulate_dev: workspace too small");
        return RR_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    double *days, *gt, *state;
    rc = rr_cema_prepass(prec, mean_temp, frac_solid_prec, nullptr,
                         (qobs && sse) ? qobs : nullptr, T, (int)L, workspace,
                         st, &days, &gt, &state);
    if (rc != RR_OK) return rc;
    const dim3 grid((unsigned)rr_ceil_div(N, RR_BLOCK)), block(RR_BLOCK);
    const double *qo = (qobs && sse) ? qobs : nullptr;
    if (L > RR_CEMANEIGE_MAX_LAYERS) {
        cemaneige_dyn_kernel<void><<<grid, block, 0, st>>>(
            days, gt, T, (int)L, cema_record_len((int)L, false), snow_pack_init,
            thermal_state_init, 0., 0., params, 2, N, nullptr, 0, state,
            outflow, G, eTG, nullptr, nullptr, ld, qo, sse, nullptr);
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    // time tiles (common.h RrTiles) for sweeps of many rounds of waves
    RrTiles tiles = {nullptr, nullptr, 0, 0};
    {
        const int64_t opt = rr_option(RR_OPT_TIME_TILES);
        int pieces = 0;
        if (T > 16 && T < 2000000000) {
            if (opt > 1) pieces = (int)opt;
            else if (opt < 0 && (int64_t)grid.x > 6 * (int64_t)rr_simd_count())
                pieces = 4;
        }
        if (pieces > 1) {
            tiles.queue = (int *)((char *)workspace +
                                  cema_tile_offset(T, L, false));
            tiles.state = (double *)((char *)tiles.queue +
                                     rr_tile_queue_bytes(N));
            tiles.pieces = pieces;
            RR_HIP(hipMemsetAsync(tiles.queue, 0, rr_tile_queue_bytes(N), st));
        }
    }
    // (RR_OPT_FUSED_VARIANT pins the form here too: 1 the many-waves kernel,
    // 2 / 3 the small-sweep one)
    const int64_t pinned = rr_option(RR_OPT_FUSED_VARIANT);
    const bool small_form =
        pinned == 1 ? false : (pinned == 2 || pinned == 3) ? true
        : (int64_t)grid.x <= 2 * (int64_t)rr_simd_count();
    dispatch_layers((int)L, [&](auto LL) {
        if (tiles.pieces > 1)
            cemaneige_kernel<LL.value, true>
                <<<dim3((unsigned)((int64_t)tiles.pieces * grid.x)), block, 0,
                   st>>>(days, gt, T, snow_pack_init, thermal_state_init,
                         params, N, outflow, G, eTG, ld, qo, sse, tiles);
        else if (small_form)
            cemaneige_kernel<LL.value, false, true><<<grid, block, 0, st>>>(
                days, gt, T, snow_pack_init, thermal_state_init, params, N,
                outflow, G, eTG, ld, qo, sse, tiles);
        else
            cemaneige_kernel<LL.value><<<grid, block, 0, st>>>(
                days, gt, T, snow_pack_init, thermal_state_init, params, N,
                outflow, G, eTG, ld, qo, sse, tiles);
    });
    RR_HIP(hipGetLastError());
    return RR_OK;
}

// ---- the reference's own GR4J sequence for the sets that are not civil ------
// (gr4j_reference.h)  One lane per set, launched behind the fast kernels; a
// civil set's lane returns at once.  Every other one runs the snow routine as
// the fast kernels do (it is the reference's sequence but for two quotients
// taken as one multiply) and the reference's own run_gr4j on its outflow
// (cemaneigegr4j_model.py:57-62), and overwrites its columns and its score.
template <int L>
__global__ __launch_bounds__(RR_BLOCK) void cemaneigegr4j_reference_kernel(
    CoupledOut o, const double *__restrict__ days,
    const double *__restrict__ gtresh, int64_t T, double snow_pack_init,
    double thermal_state_init, double s_init, double r_init,
    const double *__restrict__ params, int64_t N,
    const int *__restrict__ plan, double *__restrict__ sse)
{
    const int64_t i = (int64_t)blockIdx.x * RR_BLOCK + threadIdx.x;
    if (i >= N) return;
    if (gr4j_plan_tier(plan[0], plan[1], 0, plan[2]) < 0) return;
    const double *p = params + i * 6;
    if (plan[3] == 0 && gr4j_civil_set(p[2], p[3], p[4], s_init, r_init) &&
        gr4j_civil_snow_par(p[0]) && gr4j_civil_snow_par(p[1]) &&
        gr4j_civil_snow_par(snow_pack_init) &&
        gr4j_civil_snow_par(thermal_state_init))
        return;
    Gr4jRef g;
    if (!g.init(p[2], p[3], p[4], p[5], s_init, r_init)) return;
    const double CTG = p[0], Kf = p[1];
    double G[L], eTG[L];
#pragma unroll
    for (int l = 0; l < L; ++l) { G[l] = 0.0; eTG[l] = 0.0; }
    constexpr int D = cema_record_len(L, true);
    double acc = 0.0;
    for (int64_t t = 0; t < T; ++t) {
        double rec[D];
#pragma unroll
        for (int k = 0; k < D; ++k) rec[k] = days[t * D + k];
        // (the reference's own snow day too: its outflow to the bit)
        const double liquid = t == 0
            ? cema_ref_day<L, true>(rec, gtresh, snow_pack_init,
                                    thermal_state_init, CTG, Kf, G, eTG)
            : cema_ref_day<L, false>(rec, gtresh, snow_pack_init,
                                     thermal_state_init, CTG, Kf, G, eTG);
        const double q = g.day(liquid, rec[3 * L]);
        if (o.qsim) rr_out(&o.qsim[t * o.ld + i], q);
        if (o.G) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                rr_out(&o.G[(t * L + l) * o.ld + i], G[l]);
                rr_out(&o.eTG[(t * L + l) * o.ld + i], eTG[l]);
            }
            rr_out(&o.s_store[t * o.ld + i], g.s);
            rr_out(&o.r_store[t * o.ld + i], g.r);
        }
        if (sse) {
            const double d = rec[D - 1] - q;     // the day's observation
            acc = __builtin_fma(d, d, acc);
        }
    }
    if (sse) sse[i] = acc;
}

extern "C" int rr_cemaneigegr4j_simulate_dev(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_solid_prec, int64_t T, int64_t L,
    double snow_pack_init, double thermal_state_init, double s_init,
    double r_init, const double *params, int64_t N, double *qsim, double *G,
    double *eTG, double *s_store, double *r_store, int64_t ld,
    const double *qobs, double *sse, void *workspace, size_t workspace_bytes,
    void *stream)
{
    int rc = rr_check_common("rr_cemaneigegr4j_simulate_dev", T, N, ld, params,
                             qobs, sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    if ((rc = cema_check_layers("rr_cemaneigegr4j_simulate_dev", L)) != RR_OK)
        return rc;
    if (!prec || !mean_temp || !etp || !frac_solid_prec) {
        rr_set_error("rr_cemaneigegr4j_simulate_dev: NULL forcing pointer");
        return RR_E_NULL;
    }
    const int ns = (G != nullptr) + (eTG != nullptr) + (s_store != nullptr)
                   + (r_store != nullptr);
    if (ns != 0 && ns != 4) {
        rr_set_error("rr_cemaneigegr4j_simulate_dev: pass all four storage "
                     "outputs or none");
        return RR_E_NULL;
    }
    if ((rc = rr_check_outputs("rr_cemaneigegr4j_simulate_dev", qsim,
                               ns != 0)) != RR_OK)
        return rc;
    if (!workspace || workspace_bytes < cema_ws_bytes(T, L, true, N)) {
        rr_set_error("rr_cemaneigegr4j_simulate_dev: workspace too small");
        return RR_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int *d_plan = (const int *)workspace;
    // whatever lies behind the base workspace is unit-hydrograph scratch
    const size_t base_ws = cema_ws_bytes(T, L, true, N);
    double *uh_mem = (double *)((char *)workspace + base_ws);
    const int mem_cap = gr4j_mem_cap(workspace_bytes - base_ws, N);
    rc = rr_gr4j_plan_async(params, N, 6, 5, (int *)workspace, mem_cap, st);
    if (rc != RR_OK) return rc;
    // a block the plan cannot run (no tier selected) writes nothing: its
    // scores then read NaN, not whatever the buffer held
    if (qobs && sse)
        RR_HIP(hipMemsetAsync(sse, 0xFF, (size_t)N * sizeof(double), st));
    const int force_lds = (int)rr_option(RR_OPT_GR4J_FORCE_LDS);
    double *days, *gt, *state;
    rc = rr_cema_prepass(prec, mean_temp, frac_solid_prec, etp,
                         (qobs && sse) ? qobs : nullptr, T, (int)L, workspace,
                         st, &days, &gt, &state, (int *)workspace + 3);
    if (rc != RR_OK) return rc;
    const dim3 grid((unsigned)rr_ceil_div(N, RR_BLOCK)), block(RR_BLOCK);
    const double *qo = (qobs && sse) ? qobs : nullptr;
    // every unit-hydrograph tier is enqueued; the kernels pick the one the
    // plan selects (gr4j_core.h)
    const size_t lds_bytes = GR4J_LDS_BYTES;
    if (L > RR_CEMANEIGE_MAX_LAYERS) {
        gr4j_for_each_indexed_tier([&](auto uh) {
            using UH = decltype(uh);
            cemaneige_dyn_kernel<UH>
                <<<grid, block, std::is_same<UH, UhLds>::value ? lds_bytes : 0,
                   st>>>(days, gt, T, (int)L, cema_record_len((int)L, true),
                         snow_pack_init,
                         thermal_state_init, s_init, r_init, params, 6, N,
                         d_plan, /*force_lds=*/1, state, qsim, G, eTG, s_store,
                         r_store, ld, qo, sse, uh_mem);
        });
        RR_HIP(hipGetLastError());
        return RR_OK;
    }
    const CoupledOut out = {qsim, G, eTG, s_store, r_store, ld};
    // At most two waves per SIMD (1024 SIMDs on a whole MI355X; one GPU's
    // shard of BASELINE configs[3], every `fit` population): the small-sweep
    // forms where there is one -- optimistic GR4J half in the register
    // hydrograph tiers 3 and 5 (125k sets 14.35 -> 13.96 ms) --, the
    // many-waves kernel otherwise.  RR_OPT_FUSED_VARIANT pins one (tests: they
    // agree bit for bit): 1 many-waves, 2 small-sweep careful, 3 small-sweep
    // optimistic.
    // (0: by sweep size; the setters accept 0..3 only)
    const int64_t fv = rr_option(RR_OPT_FUSED_VARIANT);
    const bool small = fv == 2 || fv == 3 ||
                       (fv != 1 &&
                        (int64_t)grid.x <= 2 * (int64_t)rr_simd_count());
    // (the day records -- 17 doubles a day -- prefetched into the XCDs' L2,
    // common.h rr_warm_l2: 65,536 sets, scores, 8.91 -> 7.71 ms; 125k 10.91
    // -> 10.82; a million unchanged, profiles/r05_warm_family_ab.txt)
    const int warm = rr_warm_choice((int64_t)grid.x, rr_simd_count(),
                                    qsim == nullptr && G == nullptr);
    dispatch_layers((int)L, [&](auto LL) {
        gr4j_for_each_tier([&](auto uh) {
            using UH = decltype(uh);
            const size_t lds = std::is_same<UH, UhLds>::value ? lds_bytes : 0;
            if constexpr (coupled_has_optimistic<LL.value, UH>()) {
                if (small && fv != 2) {
                    cemaneigegr4j_opt_kernel<LL.value, UH>
                        <<<grid, block, 0, st>>>(
                            out, days, gt, T, snow_pack_init,
                            thermal_state_init, s_init, r_init, params, N,
                            d_plan, force_lds, qsim != nullptr, G != nullptr,
                            qo, sse, warm);
                    return;
                }
            }
            if constexpr (coupled_has_small<LL.value, UH>()) {
                if (small) {
                    cemaneigegr4j_kernel<LL.value, UH, true>
                        <<<grid, block, lds, st>>>(
                            out, days, gt, T, snow_pack_init,
                            thermal_state_init, s_init, r_init, params, N,
                            d_plan, force_lds, qsim != nullptr, G != nullptr,
                            qo, sse, uh_mem, warm);
                    return;
                }
            }
            cemaneigegr4j_kernel<LL.value, UH, false>
                <<<grid, block, lds, st>>>(
                    out, days, gt, T, snow_pack_init, thermal_state_init,
                    s_init, r_init, params, N, d_plan, force_lds,
                    qsim != nullptr, G != nullptr, qo, sse, uh_mem, warm);
        });
        // ... and behind them the sets that are not civil
        // (gr4j_reference.h)
        cemaneigegr4j_reference_kernel<LL.value><<<grid, block, 0, st>>>(
            out, days, gt, T, snow_pack_init, thermal_state_init, s_init,
            r_init, params, N, d_plan, qo ? sse : nullptr);
    });
    RR_HIP(hipGetLastError());
    return RR_OK;
}
