// hbvedu.hip -- HBV-Edu ensemble kernel for gfx950.
//
// Replaces run_hbvedu (reference: rrmpg/models/hbvedu_model.py:15-129) and
// the per-parameter-set Python loop around it (reference:
// rrmpg/models/hbvedu.py:199-209): one lane per parameter set, the four
// reservoir states (snow, soil, s1, s2) live in registers for the whole
// 30-year scan, the time loop runs inside the kernel.
//
// Data layout
//   forcing  : a packed per-day record {temp, prec, temp - T_m[month],
//              PE_m[month], qobs} (40 B) built once by hbv_pack_forcing -- the
//              month lookup and the (temp - T_m) subtraction are parameter
//              independent, so they are hoisted out of the N-fold sweep with
//              the identical fp64 operation the reference performs per set.
//              The record address is wave-uniform, so it is fetched with ONE
//              scalar load burst (s_load_dwordx8 + x2) per day into SGPRs
//              (scalar cache -> L2), never through the vector memory path.
//   params   : the reference's AoS double[N][11]; each lane reads its 11
//              values once (88 B per set, amortised over T steps).
//   outputs  : [T][ld] row-major, lane i <-> column i, so each wave stores
//              512 contiguous bytes per output per day -- as a buffer store
//              whose descriptor is the wave's row segment (common.h
//              rr_store_row): scalar address arithmetic, tail columns
//              dropped by the hardware.
//   tables   : the logarithm table of the power function, 4 KiB in LDS.
//
// Arithmetic follows the reference statement by statement in fp64; the snow
// routine is bit-identical to it.  The soil and reservoir updates are
// tolerance-driven (the discharge has to be within 1e-10 relative; measured
// 7e-15 over 30 years): the power (soil/FC)**Beta is fastmath.h's table-driven
// evaluation in plain double (a few ulp; general pow as fallback), the two
// quotients by per-lane constants are one multiply by the rounded reciprocal
// (1.5 ulp), and a product that feeds a sum is contracted into it (the file
// is built with -ffp-contract=off: only the FMAs written out below).
#include "common.h"
#include "fastmath.h"

// Tolerance-driven forms: the power evaluated on the
// soil alone in plain double arithmetic (fastmath.h fastpow_soil: no quotient,
// 512 / 256-entry tables, 28 instructions), the reservoir updates contracted.
// The reference's own sequence is hbv_reference_day below; the A/B record of
// the forms that lost is profiles/README.md.

struct __attribute__((aligned(8))) HbvDay {
    double temp;   // temp[t]
    double prec;   // prec[t]
    double dtemp;  // temp[t] - T_m[month[t]]        (hbvedu_model.py:102);
                   // times PE_m[month[t]], see day_step
    double pe_m;   // PE_m[month[t]]
    double qobs;   // observed discharge of the day (0 if no score is wanted):
                   // rides along so the score needs no second load + wait
};

// Magnitude up to which a forcing value or an initial state counts as CIVIL
// (see hbv_civil_lane): beyond it -- or not finite -- every lane of the launch
// takes the reference's own sequence.
#define HBV_CIVIL 1e6

// blockIdx.y = catchment (forcing arrays are [C][T], monthly tables [C][12])
// flags[c][block], one word per block of 256 days, written unconditionally
// (nothing to zero beforehand):
//   bit 0  a precipitation value that is negative or -0 (the wrapper rejects
//          negative values; the C-ABI takes anything): rules out the
//          kernel's TAME loop copy;
//   bit 1  a forcing value (temperature, precipitation, temp - T_m, PE_m)
//          that is not finite or beyond HBV_CIVIL: every lane of the launch
//          then takes the reference's own sequence (hbv_civil_lane);
//   bit 2  an observation that is not finite (the run-away test of the
//          reference instantiation then leaves the sums of squares aside).
// dtemp_raw[c][t] = temp - T_m[month] as the reference forms it, for that
// sequence (the day record carries the product with PE_m).
__global__ void hbv_pack_forcing(const double *__restrict__ temp,
                                 const double *__restrict__ prec,
                                 const int8_t *__restrict__ month,
                                 const double *__restrict__ PE_m,
                                 const double *__restrict__ T_m,
                                 const double *__restrict__ qobs, int64_t T,
                                 HbvDay *__restrict__ days,
                                 int *__restrict__ flags,
                                 double *__restrict__ dtemp_raw)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t c = blockIdx.y;
    const int64_t g = c * T + (t < T ? t : T - 1);
    int m = month[g];
    m = m < 0 ? 0 : (m > 11 ? 11 : m);   // memory safety only; the wrapper
                                         // has already validated 1..12
    HbvDay d;
    d.temp = temp[g];
    d.prec = prec[g];
    const double dtemp = temp[g] - T_m[c * 12 + m];      // :102
    d.pe_m = PE_m[c * 12 + m];
    d.dtemp = dtemp;
    d.dtemp *= d.pe_m;
    d.qobs = qobs ? qobs[g] : 0.0;
    const bool odd = d.prec < 0.0 || (d.prec == 0.0 && __builtin_signbit(d.prec));
    const bool uncivil = !(fabs(d.temp) <= HBV_CIVIL) ||
                         !(fabs(d.prec) <= HBV_CIVIL) ||
                         !(fabs(dtemp) <= HBV_CIVIL) ||
                         !(fabs(d.pe_m) <= HBV_CIVIL) ||
                         !(fabs(d.dtemp) <= HBV_CIVIL);
    const int any_odd = __syncthreads_or(t < T && odd);
    const int any_uncivil = __syncthreads_or(t < T && uncivil);
    // (bit 2: an observation that is not finite -- every sum of squares of
    // the catchment then is not, whatever the set: hbvedu_kernel's run-away
    // test cannot go by the sums)
    const int any_gap = __syncthreads_or(t < T && !__builtin_isfinite(d.qobs));
    if (threadIdx.x == 0)
        flags[c * gridDim.x + blockIdx.x] = (any_odd ? 1 : 0) |
                                            (any_uncivil ? 2 : 0) |
                                            (any_gap ? 4 : 0);
    if (t >= T) return;
    days[g] = d;
    dtemp_raw[g] = dtemp;
}

// The two tables of fastpow_soil (pow2_tables.h): 8 + 2 KiB in constant
// memory, copied into LDS by every wave at kernel start (each lane indexes
// them with its own subinterval, which only LDS serves at full rate); sixteen
// single-wave workgroups per CU = four waves per SIMD
static __device__ __constant__ const FpSoilEntry HBV_SOIL_LOG_TABLE[FP_SOIL_LOG_N] =
    FP_SOIL_LOG_TABLE_INIT;
static __device__ __constant__ const double HBV_SOIL_EXP_TABLE[FP_SOIL_EXP_N] =
    FP_SOIL_EXP_TABLE_INIT;

// General pow for the (never expected) arguments outside fastpow's domain.
// Out of line on purpose: inlined, OCML's pow raised the kernel from ~100 to
// 148 VGPRs (3 instead of 4-5 waves per SIMD) for a path that never runs.
__device__ __attribute__((noinline)) double pow_general(double x, double y)
{
    return pow(x, y);
}

// ---- which sets the fast forms serve, and the reference's own day ---------
// The fast forms of the time loop are restatements of the reference's
// statements that agree with them to an ulp or two AS LONG AS EVERYTHING IS A
// NUMBER: a product that overflows on its own is still a number inside an
// FMA, an infinite store times its retention factor is not the reference's
// inf - inf, a run-away store (K_0 = 7.5) cancels to an exact zero in one
// grouping and to 1e-11 in another.  So only CIVIL sets get them: parameters
// inside a (generous) box of what the model's equations mean -- recession and
// retention factors in [0, 1], 0 <= Beta <= 64, 0 <= C <= 1, field capacity
// and wilting point in [1e-3, 1e6] mm, thresholds and the degree-day factor
// within +-1e6 (DD not negative) -- with forcing and initial states that are
// numbers of at most 1e6.  Every other set -- zeros, negatives, subnormals,
// 1e+-200, infinities, NaN -- is computed with the reference's own sequence
// (hbvedu_model.py:84-127) statement by statement: IEEE quotients, the
// general pow, separate multiply and add (the file is built
// -ffp-contract=off), so whatever the reference does with infinities and NaN,
// day by day, this does too (tests/test_gpu_fuzz.py compares wild sets' NaN /
// inf pattern with the CPU restatement's over the whole series).
// Which sequence a lane gets depends on ITS parameters (and on the launch's
// forcing and initial states) only, never on its wave-mates.  The split is
// made per WAVE between two kernels, so that the loops that matter carry
// nothing of the other path: hbvedu_kernel<..., REFERENCE = false> runs the
// waves whose lanes are all civil and skips the others; the instantiation
// with REFERENCE = true, launched right behind it, skips the all-civil waves
// and runs the others -- each lane its own sequence.  (Inlined into the fast
// loops, or called from them, the reference's day cost the million-set sweep
// 2-3 ms of 19.7: 17 VGPRs, a wave per SIMD; measured three ways in round 4.)
__device__ __forceinline__ bool hbv_civil_lane(const double *__restrict__ p)
{
    const double T_t = p[0], DD = p[1], FC = p[2], Beta = p[3], C = p[4],
                 PWP = p[5], K_0 = p[6], K_1 = p[7], K_2 = p[8], K_p = p[9],
                 L = p[10];
    // (NaN fails every comparison)
    return fabs(T_t) <= HBV_CIVIL && DD >= 0.0 && DD <= HBV_CIVIL &&
           !__builtin_signbit(DD) &&
           FC >= 1e-3 && FC <= HBV_CIVIL && Beta >= 0.0 && Beta <= 64.0 &&
           C >= 0.0 && C <= 1.0 && PWP >= 1e-3 && PWP <= HBV_CIVIL &&
           K_0 >= 0.0 && K_0 <= 1.0 && !__builtin_signbit(K_0) &&
           K_1 >= 0.0 && K_p >= 0.0 && K_1 + K_p <= 1.0 && K_2 >= 0.0 &&
           K_2 <= 1.0 && fabs(L) <= HBV_CIVIL;
}

struct HbvRefDay { double snow, soil, s1, s2, q; };
// One day of the reference's own sequence (hbvedu_model.py:84-127).
__device__ __forceinline__ HbvRefDay hbv_reference_day(
    const double *__restrict__ p, double temp, double prec, double dtemp,
    double pe_m, double snow, double soil, double s1, double s2)
{
    const double T_t = p[0], DD = p[1], FC = p[2], Beta = p[3], C = p[4],
                 PWP = p[5], K_0 = p[6], K_1 = p[7], K_2 = p[8], K_p = p[9],
                 L = p[10];
    HbvRefDay r;
    double lw;
    if (temp < T_t) {                                          // :87-91
        r.snow = snow + prec;
        lw = 0.0;
    } else {                                                   // :92-96
        const double melt = DD * (temp - T_t);
        r.snow = nb_max(0.0, snow - melt);
        lw = prec + nb_min(snow, melt);
    }
    const double prec_eff = lw * pow(soil / FC, Beta);         // :99
    const double pe = (1 + C * dtemp) * pe_m;                  // :102
    double ea;                                                 // :105-108
    if (soil > PWP) ea = pe;
    else ea = pe * (soil / PWP);
    r.soil = soil + lw - prec_eff - ea;                        // :111
    const double spill = nb_max(0.0, s1 - L) * K_0;
    r.s1 = s1 + prec_eff - spill - s1 * K_1 - s1 * K_p;        // :114-118
    r.s2 = s2 + s1 * K_p - s2 * K_2;                           // :121-123
    r.q = spill + r.s1 * K_1 + r.s2 * K_2;                     // :125-127
    return r;
}

// A day record through the CONSTANT address space: the records are read-only
// for the time-loop kernels, and a wave-uniform load from address space 4 is
// a scalar load whatever else the kernel does (from the global address space
// hipcc scalarises a uniform load only while it can prove that nothing in the
// kernel has written memory before it -- the prologue's inline-asm prefetch
// already counts, and the records then arrive by vector loads in VGPRs).
__device__ __forceinline__ HbvDay hbv_load_day(const HbvDay *days, int64_t t)
{
    typedef const HbvDay __attribute__((address_space(4))) *cp_t;
    const cp_t p = (cp_t)(days + t);
    HbvDay d;
    d.temp = p->temp; d.prec = p->prec; d.dtemp = p->dtemp;
    d.pe_m = p->pe_m; d.qobs = p->qobs;
    return d;
}

// blockIdx.y = catchment.  A single-catchment launch has gridDim.y == 1; a
// multi-catchment launch (rr_hbvedu_simulate_catchments_dev) lays every array
// out catchment-major: days [C][T], params [C][N][11], outputs [C][T][ld],
// qobs [C][T], sse [C][N], inits [C][4].
// FORCING: the time loop.  0: two days per trip, each day's record one scalar
// load at its top (sweeps of more than six waves per SIMD, in time tiles);
// 3: three records rotating, a day's record asked for two days ahead (up to
// six waves per SIMD).  (The LDS-staged records north_star sketched and the
// mid-day prefetch of round 2 -- variants 1 and 2 -- lost to these and were
// removed in round 6: profiles/README.md has their A/B tables.)
// TAME: the kernel carries a second copy of the time loop for waves that
// qualify for it (see day_step); without it -- the REFERENCE instantiation --
// the kernel is the general loop alone.
//
// TILED: the time axis in PIECES (common.h "the time axis in pieces"), here
// in the PERSISTENT form -- measured faster than grid-order items in the
// qsim mode (25.3 vs 27.0 ms; scores 19.6 vs 19.9) --: the waves persistent.  A million-set sweep
// is 15,625 waves of equal duration on 1,024 SIMDs: 15.26 per SIMD, so 265
// SIMDs run a sixteenth wave while the others idle -- the kernel takes 16
// wave slots for 15.26 slots of work (4.6 %).  Here exactly as many waves
// are launched as are resident at once, and each pulls work items (piece p
// of the 64 sets of job j: days [t_p, t_p+1)) from one atomic counter, in
// piece-major order; the states (four stores + the score sum) travel from
// piece to piece through a small HBM scratch, handed over with a release /
// acquire pair at agent scope and a per-job flag.  Items are a quarter as
// long, slots refill as they free, and the SIMDs finish within a fraction
// of a wave of each other.  No deadlock: a wave waits only for an item with
// a smaller index, which a resident wave took before it.  Bit-identical to
// the untiled loop (same operations in the same order, the score summed in
// time order across the pieces).
#define HBV_TILED_MINWAVES 1
// Every wave reads a slice of the day records with ordinary vector loads when
// it starts (where the launch says so: `warm`), so that the scalar loads of
// the time loop find their lines in this XCD's L2 (see the kernel's
// prologue): loads per lane
#define HBV_WARM_L2_LOADS 4
template <bool WRITE_Q, bool WRITE_S, bool WITH_SSE, int FORCING = 0,
          bool TAME = true, int TILED = 0, bool REFERENCE = false>
__global__ __launch_bounds__(RR_BLOCK, (TILED ? HBV_TILED_MINWAVES : 1)) void
hbvedu_kernel(
    const HbvDay *__restrict__ days, int64_t T, double snow_init,
    double soil_init, double s1_init, double s2_init,
    const double *__restrict__ inits, const double *__restrict__ params,
    int64_t N, double *__restrict__ qsim, double *__restrict__ snow_out,
    double *__restrict__ soil_out, double *__restrict__ s1_out,
    double *__restrict__ s2_out, int64_t ld, const double *__restrict__ qobs,
    double *__restrict__ sse, const int *__restrict__ day_flags,
    const double *__restrict__ dtemp_raw, int *__restrict__ queue,
    double *__restrict__ tile_state, int pieces, int ncatch, int warm)
{
    __shared__ FpSoilEntry soillog[FP_SOIL_LOG_N];
    __shared__ double soilexp[FP_SOIL_EXP_N];
    for (int j = threadIdx.x; j < FP_SOIL_LOG_N; j += RR_BLOCK)
        soillog[j] = HBV_SOIL_LOG_TABLE[j];
    for (int j = threadIdx.x; j < FP_SOIL_EXP_N; j += RR_BLOCK)
        soilexp[j] = HBV_SOIL_EXP_TABLE[j];
    __syncthreads();
    const int njobs = (int)((N + RR_BLOCK - 1) / RR_BLOCK);
    // TILED == 2: several catchments (the jobs of catchment c are the slots
    // [c * njobs, (c + 1) * njobs) of the queue); every trip starts from the
    // arguments as they were passed
    const int nslots = TILED == 2 ? njobs * ncatch : njobs;
    const HbvDay *const days0 = days;
    const double *const params0 = params, *const qobs0 = qobs;
    double *const qsim0 = qsim, *const snow0 = snow_out, *const soil0 = soil_out,
                 *const s10 = s1_out, *const s20 = s2_out, *const sse0 = sse;
    if (!REFERENCE && warm != 0) {
        // the day records into this XCD's L2 (common.h rr_warm_l2); `warm`:
        // the launch's choice (hbv_launch)
        const int64_t ctotal = TILED == 2 ? ncatch : (int64_t)gridDim.y;
        rr_warm_l2(days, T * ctotal * (int64_t)sizeof(HbvDay),
                   HBV_WARM_L2_LOADS);
    }
  for (;;) {           // TILED: one work item per trip; otherwise one trip
    int job = blockIdx.x, piece = 0, slot = 0, catchment = blockIdx.y;
    if constexpr (TILED) {
        int item = 0;
        if (threadIdx.x == 0) item = atomicAdd(queue, 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= pieces * nslots) break;
        piece = __builtin_amdgcn_readfirstlane(item / nslots);
        slot = __builtin_amdgcn_readfirstlane(item - piece * nslots);
        job = slot;
        catchment = 0;
        if constexpr (TILED == 2) {
            catchment = __builtin_amdgcn_readfirstlane(slot / njobs);
            job = __builtin_amdgcn_readfirstlane(slot - catchment * njobs);
            days = days0; params = params0; qobs = qobs0; qsim = qsim0;
            snow_out = snow0; soil_out = soil0; s1_out = s10; s2_out = s20;
            sse = sse0;
        }
    }
    const int64_t i = (int64_t)job * RR_BLOCK + threadIdx.x;
    const bool active = i < N;
    if constexpr (TILED != 1) {
        const int64_t c = catchment;           // wave-uniform
        days += c * T;
        if (REFERENCE) dtemp_raw += c * T;
        params += c * N * 11;
        const int64_t out_off = c * T * ld;
        if (WRITE_Q) qsim += out_off;
        if (WRITE_S) {
            snow_out += out_off; soil_out += out_off;
            s1_out += out_off; s2_out += out_off;
        }
        if (WITH_SSE) { qobs += c * T; sse += c * N; }
        if (inits) {
            snow_init = inits[c * 4 + 0]; soil_init = inits[c * 4 + 1];
            s1_init = inits[c * 4 + 2]; s2_init = inits[c * 4 + 3];
        }
    }
    // tail lanes recompute the last set and simply do not store
    const double *p = params + (active ? i : N - 1) * 11;
    const double T_t = p[0], DD = p[1], FC = p[2], Beta = p[3], C = p[4],
                 PWP = p[5], K_0 = p[6], K_1 = p[7], K_2 = p[8], K_p = p[9],
                 L = p[10];
    // Which kernel runs this wave (hbv_civil_lane): this one if REFERENCE
    // says so.  (bit 1 of the pre-pass's flags: a forcing value that is not
    // civil.)
    bool lane_civil;
    {
        const int nb = (int)((T + 255) / 256);
        const int *flags = day_flags + (int64_t)catchment * nb;
        lanemask_t uncivil = 0, gaps = 0;
        for (int k = threadIdx.x; k - (int)threadIdx.x < nb; k += RR_BLOCK) {
            uncivil |= RR_LANES(k < nb && (flags[k] & 2) != 0);
            if (REFERENCE) gaps |= RR_LANES(k < nb && (flags[k] & 4) != 0);
        }
        lane_civil = uncivil == 0 && fabs(snow_init) <= HBV_CIVIL &&
                     fabs(soil_init) <= HBV_CIVIL &&
                     fabs(s1_init) <= HBV_CIVIL &&
                     fabs(s2_init) <= HBV_CIVIL && hbv_civil_lane(p);
        bool wave_civil = (rr_exec() & ~RR_LANES(lane_civil)) == 0;
        if constexpr (REFERENCE) {
            // ... and an all-civil wave in which a set RAN AWAY is this
            // kernel's as well.  A civil set still can: FC = 1 mm under a
            // soil of 100 mm makes (soil / FC)**Beta 1e4, the soil overshoots
            // to -1e6, 1e14 ... and within weeks the stores are infinite --
            // where the fast forms are no longer the reference's statements
            // (inf * K_0 with K_0 = 0 is NaN in max(0, s1 - L) * K_0 and 0 in
            // the folded form's hardware maximum; fuzz seed 684: NaN a day
            // late).  Infinities and NaN stay in these recurrences (the
            // reservoirs carry them into the discharge, the discharge into
            // the sum of squares), so what the fast kernel -- which has just
            // run this wave -- left in the LAST row of each output it wrote,
            // and in the sums, tells; the fast kernel itself carries nothing
            // for it (a flag store at the end of its sweep cost the
            // multi-catchment kernels 12 VGPRs and their fourth wave per
            // SIMD: 15.9 -> 18.7 ms for 125 catchments x 10k sets).  The wave
            // then is computed again below, each lane with the reference's
            // own day from the first day that starts with a store that is
            // not finite (day_step) -- the rule a civil set gets among
            // wave-mates that are not, so its bits do not depend on the
            // company.
            bool ran = false;
            if (wave_civil && active) {
                const int64_t last = (T - 1) * ld + i;
                if (WRITE_Q) ran = ran || !__builtin_isfinite(qsim[last]);
                if (WRITE_S)
                    ran = ran || !__builtin_isfinite(snow_out[last]) ||
                          !__builtin_isfinite(soil_out[last]) ||
                          !__builtin_isfinite(s1_out[last]) ||
                          !__builtin_isfinite(s2_out[last]);
                if (WITH_SSE && gaps == 0)
                    ran = ran || !__builtin_isfinite(sse[i]);
            }
            wave_civil = wave_civil && RR_LANES(ran) == 0;
        }
        if (wave_civil == REFERENCE) {
            // not this kernel's wave.  (A piece of the time axis still tells
            // the next one, which is skipped just the same, not to wait.)
            if constexpr (TILED) {
                if (piece + 1 < pieces && threadIdx.x == 0)
                    __hip_atomic_store(queue + 1 + slot, piece + 1,
                                       __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                continue;
            } else {
                break;
            }
        }
    }

    const InvDivisor inv_FC = make_inv_divisor(FC);
    const InvDivisor inv_PWP = make_inv_divisor(PWP);
    // box in which (soil/FC)**Beta is certainly finite (see the time loop):
    // FC 2^-9 < soil < FC 2^9, |Beta| <= 64, FC a sane positive number.
    // Tested on soil's HIGH WORD with two 32-bit integer instructions (a
    // subtraction and an unsigned compare against the box's span) instead of
    // two fp64 compares: for positive finite doubles the high word is
    // monotonic, a negative, zero, subnormal, infinite or NaN soil wraps to a
    // difference beyond any span, and a lane whose FC or Beta rule the box
    // out has span 0 -- the box is a sufficient condition only, so being a
    // few ulps of 2^32 tighter at both ends costs nothing.
    const bool box_ok = (FC > 0x1p-500) && (FC < 0x1p500) &&
                        (fabs(Beta) <= 64.0);
    unsigned box_lo = (unsigned)__double2hiint(FC * 0x1p-9) + 1u;
    unsigned box_span =
        box_ok ? (unsigned)__double2hiint(FC * 0x1p9) - box_lo : 0u;
    // (opaque to the compiler, which would otherwise re-derive both every day)
    asm("" : "+v"(box_lo), "+v"(box_span));
    // N Beta / ln 2 and its product with ln FC, for fastpow_soil
    double beta_y2N, beta_cF;
    fastpow_soil_exponent(Beta, FC, soillog, &beta_y2N, &beta_cF);
    // loop-invariant lane masks for the wave votes (common.h)
    const lanemask_t fc_m = RR_LANES(inv_FC.ok), pwp_m = RR_LANES(inv_PWP.ok);
    const bool pwp_pos = inv_PWP.ok && PWP > 0.0;
    // what a day leaves of the two linear stores
    const double keep_1 = 1 - K_1 - K_p, keep_2 = 1 - K_2;
    const double neg_LK0 = -(L * K_0);
    // K_0: +0 or a positive number; L finite (v_cmp_class masks)
    const lanemask_t k0_folds_m = lanes_of_class(K_0, 0x1c0) & lanes_finite(L);
    const bool k0_folds = (k0_folds_m >> (threadIdx.x & 63)) & 1;

    double snow = snow_init, soil = soil_init, s1 = s1_init, s2 = s2_init;
    double acc = 0.0;
    // Outputs are addressed as (wave-uniform row base, advanced by ld per day
    // with scalar adds) + (this lane's fixed column inside the wave's 512-byte
    // segment): no vector address arithmetic inside the time loop.
    const int lane_off = threadIdx.x * 8;
    const int64_t first = (int64_t)job * RR_BLOCK;
    const unsigned row_bytes = rr_row_bytes(first, N);
    int64_t row = first;                            // t * ld + first column
    // this trip's days [t_begin, t_end)
    const int Ti = (int)T;
    int t_begin = 1, t_end = Ti;
    // a piece's states in the scratch: [5][njobs * 64], lane-contiguous
    double *const hand = TILED ? tile_state + ((int64_t)slot * RR_BLOCK +
                                               threadIdx.x) : nullptr;
    const int64_t hand_stride = (int64_t)nslots * RR_BLOCK;
    if constexpr (TILED) {
        const int len = (Ti - 1 + pieces - 1) / pieces;
        t_begin = 1 + piece * len;
        t_end = t_begin + len < Ti ? t_begin + len : Ti;
        if (t_begin > Ti) t_begin = Ti;
        row = first + (int64_t)(t_begin - 1) * ld;
    }
    if (TILED && piece > 0) {
        // the piece before this one (a smaller item, taken earlier by a
        // resident wave) publishes flag = piece when its states are out
        int *flag = queue + 1 + slot;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT) < piece)
            __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        snow = hand[0];
        soil = hand[hand_stride];
        s1 = hand[2 * hand_stride];
        s2 = hand[3 * hand_stride];
        acc = hand[4 * hand_stride];
    } else {
        // t = 0: qsim[0] = 0, state[0] = init (hbvedu_model.py:71-81)
        if (WRITE_Q) rr_store_row(qsim + row, row_bytes, lane_off, 0.0);
        if (WRITE_S) {
            rr_store_row(snow_out + row, row_bytes, lane_off, snow);
            rr_store_row(soil_out + row, row_bytes, lane_off, soil);
            rr_store_row(s1_out + row, row_bytes, lane_off, s1);
            rr_store_row(s2_out + row, row_bytes, lane_off, s2);
        }
        if (WITH_SSE) {
            const double d = qobs[0] - 0.0;
            acc = d * d;
        }
    }

    // (the record is taken BY VALUE: one s_load_dwordx8 per day up front; a
    // reference lets hipcc re-load single fields at their use sites, each
    // with its own wait)
    // `mid` runs between the effective-precipitation block and the rest of
    // the day: the small-sweep variant requests the next day's record there
    // (32-bit day counters: hbv_launch rejects T >= 2^31; a 64-bit count
    // costs a second scalar add per day and, in the unrolled loop, a VALU
    // compare -- there is no 64-bit signed scalar compare)
    // `tame` (a std::bool_constant): the wave runs the copy of the time loop
    // in which `min(snow, melt)` (:94) is one v_min_f64.  numba's min(a, b) is
    // `b if b < a else a`: a compare and a 64-bit select, because the
    // hardware minimum differs from it for a NaN snow pack (it would return
    // the melt) and for a melt of -0 against an empty pack (it would return
    // -0).  Neither matters in a wave whose lanes all have a degree-day factor
    // that is not negative, whose initial pack is a number that is not
    // negative, and whose precipitation -- the pre-pass flags this -- is
    // never NaN, negative or -0: the pack stays in {+0} u (0, inf], and the
    // melt DD * (temp - T_t) of a day with temp >= T_t is -0 only for
    // temp = -0 against T_t = +0, where the one use of the minimum is its
    // sum with a precipitation that is not -0 -- the same +0, or prec,
    // either way.  Bit-identical by construction; any other wave runs the
    // general copy.
    // `soff`: the day's row as a byte offset from `row` (the stores' SOFFSET
    // field, common.h rr_store_row_at): the unrolled loops advance `row`
    // once per trip and pass loop-invariant multiples of ld * 8 -- three
    // scalar instructions of address arithmetic per trip instead of per day
    auto day_step = [&](const HbvDay f, int t, auto &&mid, auto tame,
                        unsigned soff) {

        // snow routine (hbvedu_model.py:87-96)
        const double melt = DD * (f.temp - T_t);
        const bool cold = f.temp < T_t;
        double snow_n, liquid_water;
        if constexpr (decltype(tame)::value) {
            // The snow routine of a tame wave as ONE quantity m that leaves
            // the pack for the soil: min(snow, melt) on a day that is not
            // cold, -prec (the precipitation stays) on a cold one --
            //     snow' = snow - m,   liquid_water = prec + m.
            // Bit for bit the reference's four expressions: a tame wave's
            // pack, melt and precipitation are finite numbers (civil lanes,
            // civil forcing), the pack and the precipitation not negative
            // and not -0, so  snow - min(snow, melt)  is  snow - melt  where
            // that is positive and +0 where max(0, .) would have cut it off
            // (x - x = +0; a melt of -0 against any pack gives the pack
            // back);  snow - (-prec)  is  snow + prec;  prec + (-prec) = +0
            // is the reference's `liquid_water = 0`.
            // The cold lanes' m is written by a v_mov_b64 under an exec mask
            // -- one instruction where a 64-bit select costs two VOP3
            // v_cndmask_b32 (8.6 cycles, profiles/ubench/valu_cost.hip), and
            // the two selects, the second difference and the maximum of the
            // general form are not evaluated at all: 6 vector instructions
            // instead of 10 for the snow routine.
            double m;
            asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(snow), "v"(melt));
            const lanemask_t cold_m = RR_LANES(cold);
            // (-prec as v_max_f64 of the negated record field with itself:
            // the negation is an operand modifier, where a v_mov_b64 needs
            // the negative in an SGPR pair of its own -- s_xor + s_mov a day;
            // prec is a number here, so the maximum is the value)
            lanemask_t saved;
            asm("s_and_saveexec_b64 %1, %2\n\t"
                "v_max_f64 %0, -%3, -%3\n\t"
                "s_mov_b64 exec, %1"
                : "+v"(m), "=&s"(saved)
                : "s"(cold_m), "s"(f.prec)
                : "scc");
            snow_n = snow - m;
            liquid_water = f.prec + m;
        } else {
            snow_n = cold ? snow + f.prec : nb_max(0.0, snow - melt);
            double least;
            if constexpr (decltype(tame)::value)
                asm("v_min_f64 %0, %1, %2"
                    : "=v"(least) : "v"(snow), "v"(melt));
            else
                least = nb_min(snow, melt);
            liquid_water = cold ? 0.0 : f.prec + least;
        }
        // first operation of the soil update (:111), taken here so that
        // liquid_water itself is dead after the power block (it lives on as
        // prec_eff, in place, on the days without the power)
        // (the plain loop only: in the loops that request the next record in
        // the middle of the day -- the sweeps of few waves per SIMD -- the
        // second copy of that request cost more than the three instructions:
        // 125k sets 2.63 -> 2.75 ms, 65k 2.37 -> 2.62)
        // (nor in the multi-catchment launch: 125 x 10k sets, scores, 17.96
        // -> 18.29)
        constexpr bool split_tail = decltype(tame)::value && FORCING == 0 &&
                                    TILED != 2;
        double soil_lw = soil;
        if constexpr (!split_tail) {
            soil_lw = soil + liquid_water;
            asm("" : "+v"(soil_lw));  // (evaluated HERE, not sunk below)
        }

        // effective precipitation (:99): liquid_water * (soil/FC)**Beta.
        // On dry or frozen days liquid_water is exactly 0 for every lane of
        // the wave (the forcing is shared), and 0 * pow(..) is 0 whenever
        // pow(..) is finite -- guaranteed when soil/FC lies in [2^-10, 2^10]
        // (tested without dividing: FC * 2^-9 <= soil <= FC * 2^9, FC > 0)
        // and |Beta| <= 64 -- so the wave skips the division and the power
        // altogether.  Outside that box (NaN/inf/zero/negative operands, huge
        // Beta) both are evaluated and 0 * inf / 0 * NaN propagate exactly as
        // in the reference.
        // lanes inside the box.  (The two quotients by FC and PWP are
        // FAITHFUL -- soil * RN(1/FC), within 1.5 ulp for every numerator,
        // invdiv.h inv_mul_core; their only vote is on the divisor, a loop
        // invariant.  HBV-Edu 1M sets 25.25 -> 24.70 ms, deviation from the
        // reference semantics 5e-15 -> 6e-15.)
        const lanemask_t soil_m = RR_LANES(
            (unsigned)__double2hiint(soil) - box_lo < box_span);
        // lanes that need the power: wet, or outside the box (votes are done
        // on lane masks, common.h)
        const lanemask_t wet_m = RR_LANES(liquid_water != 0.0);
        double prec_eff = liquid_water;   // == liquid_water * finite (it is 0)
        // (measured and removed in round 4: this part of the day written
        // inside both arms of the power's branch, fenced by scheduling
        // barriers behind the power's LDS table read -- hipcc hoists half of
        // it above the branch again: 125k sets 2.85 -> 2.91 ms, 1M 19.8 ->
        // 20.2)
        double pe, dry, over, s2_n;
        auto independent_of_the_power = [&]() __attribute__((always_inline)) {
            // potential / actual evapotranspiration (:102-108); the select
            // picks the factor, 1 or soil/PWP, so that the product with pe
            // goes into the soil update's FMA
            // (1 + C dtemp) PE_m as PE_m + C (dtemp PE_m): the forcing
            // record carries the product
            pe = __builtin_fma(C, f.dtemp, f.pe_m);
            // min(soil/PWP, 1) for a lane with a positive, usable PWP -- what
            // the reference's `if soil > PWP` selects, in one instruction (a
            // NaN soil has made soil_lw NaN already) --, the select itself
            // for any other lane; a tame wave has only lanes of the first kind
            if constexpr (decltype(tame)::value) {
                const double ratio = inv_mul_core(soil, inv_PWP);
                asm("v_min_f64 %0, %1, 1.0" : "=v"(dry) : "v"(ratio));
            } else {
                const double ratio = mul_by_inverse_m(soil, inv_PWP, pwp_m);
                double least1;
                asm("v_min_f64 %0, %1, 1.0" : "=v"(least1) : "v"(ratio));
                dry = pwp_pos ? least1 : ((soil > PWP) ? 1.0 : ratio);
            }
            // near-surface reservoir's overflow (:114-115)
            // (a tame wave -- K_0 not negative, L finite --: max(0, s1 - L)
            // K_0 as max(0, s1 K_0 - L K_0), the product L K_0 a loop
            // invariant; a lane's form does not depend on the loop copy its
            // wave runs: in the general copy a lane with such K_0 and L takes
            // the same value, any other lane the reference's sequence.
            // The folded form's error is ABSOLUTE: L K_0 is rounded once, so
            // the spill is within ulp(L K_0) of the reference's -- close to
            // the threshold, s1 ~ L, that is not a relative bound, and a
            // spill of that size can come out as 0 or the other way round
            // (the reference's (s1 - L) K_0 is exact there).  1e-16 L K_0 mm
            // a day against a discharge compared at 1e-10; pinned by
            // tests/test_gpu_parity.py test_hbvedu_overflow_term_alone.)
            if constexpr (decltype(tame)::value) {
                over = rr_hw_max(__builtin_fma(s1, K_0, neg_LK0), 0.0);
            } else {
                const double folded =
                    rr_hw_max(__builtin_fma(s1, K_0, neg_LK0), 0.0);
                over = k0_folds ? folded : nb_max(0.0, s1 - L) * K_0;
            }
            // base-flow reservoir (:121-123): s2 (1 - K_2) + s1 K_p
            s2_n = __builtin_fma(s2, keep_2, s1 * K_p);
        };
        double soil_n, s1_n;
        // (every wave runs with all 64 lanes: tail lanes recompute the last
        // set, workgroups are RR_BLOCK threads -- no AND with exec)
        if ((wet_m | ~soil_m) != 0) {
            if constexpr (split_tail) {
                // (first operation of the soil update, before liquid_water
                // becomes prec_eff in place)
                soil_lw = soil + liquid_water;
                asm("" : "+v"(soil_lw));
            }
            // fastmath.h fastpow_soil: the power from the soil alone -- the
            // quotient by FC lives in a per-lane constant of the exponent --,
            // table-driven in plain double, 28 instructions.  Inside the box
            // its arguments are in its domain by construction (soil a
            // positive normal number within 2^9 of FC, |zz| <= 64 * 9.1): the
            // box mask is the vote.  If any lane is outside the box the wave
            // also evaluates the general pow of the reference's own quotient
            // and every lane fastpow_soil cannot serve takes it.
            double sN;
            double pw = fastpow_soil<FORCING == 3>(soil, beta_y2N, beta_cF,
                                                   soillog, soilexp, &sN);
            if (RR_ANY_OUTSIDE(soil_m)) {
                double soil_again = soil;
                asm volatile("" : "+v"(soil_again));
                const double general = pow_general(soil_again / FC, Beta);
                pw = fastpow_soil_ok(soil_again, sN) ? pw : general;
            }
            // lanes of this wave that did not need the power sit inside the
            // box with liquid_water == 0: their pw is finite (|z| <= 64 * 9.1)
            // and 0 * pw is the 0 they already hold, so no select is needed
            // (in place: spelled as `prec_eff = liquid_water * pw` hipcc
            // gives the product a register of its own and pays a v_mov_b64
            // on every day WITHOUT the power to join the two)
            asm("v_mul_f64 %0, %0, %1" : "+v"(prec_eff) : "v"(pw));
            if constexpr (split_tail) {
                // (the tame copy: the rest of the day inside the branch's
                // arm, see below)
                independent_of_the_power();
                mid();
                soil_n = __builtin_fma(-pe, dry, soil_lw - prec_eff);
                s1_n = __builtin_fma(s1, keep_1, prec_eff - over);
                asm("" : "+v"(soil_n), "+v"(s1_n));
            }
        }
        else if constexpr (split_tail) {
            // (a day without the power, see below)
            independent_of_the_power();
            mid();
            soil_n = __builtin_fma(-pe, dry, soil);
            s1_n = __builtin_fma(s1, keep_1, -over);
            asm("" : "+v"(soil_n), "+v"(s1_n));
        }
        // The reservoir updates with their multiply-adds CONTRACTED -- each
        // product fused into the sum that takes it, one rounding instead of
        // two -- and the two linear stores regrouped around their
        // loop-invariant retention factors: 12 instructions instead of 23 a
        // day, every result within an ulp or two of the reference's.
        // (The reference's own sequence: hbv_reference_day.)
        // soil moisture (:111); near-surface reservoir (:114-118): s1 - s1
        // K_1 - s1 K_p as s1 (1 - K_1 - K_p), the factor a loop invariant
        //
        // The tame copy forms both INSIDE the arms of the power's branch: on
        // a day without the power -- three days of five -- the liquid water is
        // a zero in every lane and the soil inside the box (a positive
        // number), so soil + 0 - 0 is the soil and 0 - over is -over: three
        // vector instructions that day does not issue.  (The asm pins keep
        // hipcc from joining the arms again behind a register copy.)
        if constexpr (!split_tail) {
            independent_of_the_power();
            mid();
            soil_n = __builtin_fma(-pe, dry, soil_lw - prec_eff);
            s1_n = __builtin_fma(s1, keep_1, prec_eff - over);
        }

        // discharge mixes old and new states (:125-127)
        const double q =
            __builtin_fma(s2_n, K_2, __builtin_fma(s1_n, K_1, over));

        double q_c = q;
        if constexpr (REFERENCE) {
            // each lane its own sequence: a civil lane the fast forms (the
            // bits it has in any other launch), every other lane the
            // reference's own day, from the same start states -- and a civil
            // lane too from the day on which it starts with a store that is
            // not finite (a set that ran away, see the wave's choice above:
            // infinities and NaN stay, so does the choice)
            const HbvRefDay r = hbv_reference_day(
                p, f.temp, f.prec, dtemp_raw[t], f.pe_m, snow, soil, s1, s2);
            const bool fast_day =
                lane_civil && __builtin_isfinite(snow) &&
                __builtin_isfinite(soil) && __builtin_isfinite(s1) &&
                __builtin_isfinite(s2);
            snow = fast_day ? snow_n : r.snow;
            soil = fast_day ? soil_n : r.soil;
            s1 = fast_day ? s1_n : r.s1;
            s2 = fast_day ? s2_n : r.s2;
            q_c = fast_day ? q : r.q;
        } else {
            snow = snow_n; soil = soil_n; s1 = s1_n; s2 = s2_n;
        }

        if (WRITE_Q) rr_store_row_at(qsim + row, row_bytes, lane_off, soff, q_c);
        if (WRITE_S) {
            rr_store_row_at(snow_out + row, row_bytes, lane_off, soff, snow);
            rr_store_row_at(soil_out + row, row_bytes, lane_off, soff, soil);
            rr_store_row_at(s1_out + row, row_bytes, lane_off, soff, s1);
            rr_store_row_at(s2_out + row, row_bytes, lane_off, soff, s2);
        }
        if (WITH_SSE) {
            const double d = f.qobs - q_c;
            acc = __builtin_fma(d, d, acc);   // one rounding per day
        }
        (void)t;
    };

    // (hbv_launch rejects 3 * ld * 8 >= 2^32)
    const unsigned ld8 = (unsigned)ld * 8u, ld16 = 2u * ld8, ld24 = 3u * ld8;
    (void)ld16; (void)ld24;
    auto time_loop = [&](auto tame) {
        if constexpr (FORCING == 3) {
            // sweeps of at most two waves per SIMD, round 4: a lone wave's
            // day is a latency chain (a day takes as long with one wave on
            // the SIMD as with two), and the longest link that nothing covers
            // is the record's scalar load, requested and waited for in the
            // same day by the plain loop below.  Here THREE records rotate (loop unrolled by three: no
            // copies) and the middle of day t requests the record of day
            // t + 2: by the time it is used a whole day has passed.  Scalar
            // loads return out of order, so every wait is for all of them:
            // the `use` of the NEXT day's record right before the request
            // makes that wait explicit at a point where the record, asked
            // for a day ago, has long arrived -- and the compiler then knows
            // it to be there at the top of the next day.  The fetch runs two
            // records ahead: the workspace holds two spare records.
            typedef const HbvDay __attribute__((address_space(4))) *cp_t;
            cp_t pn = (cp_t)(days + t_begin + 2);
            auto fetch = [&](HbvDay &dst) {
                asm volatile("" : "+s"(pn));
                dst.temp = pn->temp; dst.prec = pn->prec;     // one load burst
                dst.dtemp = pn->dtemp; dst.pe_m = pn->pe_m; dst.qobs = pn->qobs;
                pn += 1;
            };
            auto use = [](const HbvDay &r) {
                asm volatile("" : : "s"(r.temp), "s"(r.prec), "s"(r.dtemp),
                             "s"(r.pe_m), "s"(r.qobs));
            };
            HbvDay a = hbv_load_day(days, t_begin),
                   b = hbv_load_day(days, t_begin + 1), c;
            use(a);
            int t = t_begin;
            for (; t + 2 < t_end; t += 3) {
                day_step(a, t, [&] { use(b); fetch(c); }, tame, ld8);
                day_step(b, t + 1, [&] { use(c); fetch(a); }, tame, ld16);
                day_step(c, t + 2, [&] { use(a); fetch(b); }, tame, ld24);
                row += 3 * ld;
            }
            if (t < t_end) {
                day_step(a, t, [&] { use(b); }, tame, ld8);
                if (t + 1 < t_end) day_step(b, t + 1, [] {}, tame, ld16);
            }
        } else {
            // (two days per trip, written out -- the votes are convergent
            // operations, which keeps hipcc from unrolling a loop with a
            // remainder on its own: one taken branch per two days)
            int t = t_begin;
            for (; t + 1 < t_end; t += 2) {
                const HbvDay f0 = hbv_load_day(days, t);   // s_load_dwordx8 + x2
                day_step(f0, t, [] {}, tame, ld8);
                const HbvDay f1 = hbv_load_day(days, t + 1);
                day_step(f1, t + 1, [] {}, tame, ld16);
                row += 2 * ld;
            }
            if (t < t_end) {
                const HbvDay f = hbv_load_day(days, t);
                day_step(f, t, [] {}, tame, ld8);
            }
        }
    };
    // DD: +0, positive, +inf or NaN (v_cmp_class mask 0x3c3)
    bool tame_wave = false;
    if constexpr (TAME) {
        // any flag of this catchment's pre-pass blocks set?
        const int nb = (int)((T + 255) / 256);
        const int *flags = day_flags + (int64_t)catchment * nb;
        lanemask_t odd = 0;
        for (int k = threadIdx.x; k - (int)threadIdx.x < nb; k += RR_BLOCK)
            odd |= RR_LANES(k < nb && (flags[k] & 1) != 0);
        tame_wave = odd == 0 && snow_init >= 0.0 &&
                    !__builtin_signbit(snow_init) &&
                    (rr_exec() & ~lanes_of_class(DD, 0x3c3)) == 0;
        // ... and both divisors usable, PWP positive; K_0 +0 or a positive
        // number, L finite (day_step)
        tame_wave = tame_wave &&
                    (rr_exec() & ~(fc_m & RR_LANES(pwp_pos) &
                                   k0_folds_m)) == 0;
    }
    if constexpr (TAME) {
        if (tame_wave) time_loop(std::true_type{});
        else time_loop(std::false_type{});
    } else {
        (void)tame_wave;
        time_loop(std::false_type{});
    }
    if (TILED && piece + 1 < pieces) {
        // hand the states to the next piece: stores, release at agent scope
        // (the next piece may run on another XCD), then the flag
        hand[0] = snow;
        hand[hand_stride] = soil;
        hand[2 * hand_stride] = s1;
        hand[3 * hand_stride] = s2;
        hand[4 * hand_stride] = acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0)
            __hip_atomic_store(queue + 1 + slot, piece + 1, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (WITH_SSE && active) sse[i] = acc;
    }
    if constexpr (!TILED) break;
  }
}

// forcing records (+ 2: the spare records the prefetching variants may
// touch), the pre-pass's flags (one int per 256 days) and, behind them, the
// raw temp - T_m series of the reference's own day
static size_t hbv_forcing_bytes(int64_t T, int64_t C)
{
    if (T < 1) T = 1;
    if (C < 1) C = 1;
    return rr_align256((size_t)(T * C + 2) * sizeof(HbvDay) +
                       (size_t)rr_ceil_div(T, 256) * (size_t)C * sizeof(int)) +
           rr_align256((size_t)T * (size_t)C * 8);
}
// the tiled kernel's work queue {counter, flag per job} and its hand-over
// scratch [5][jobs * 64]
static size_t hbv_queue_bytes(int64_t N, int64_t C)
{
    if (C < 1) C = 1;
    return rr_align256((size_t)(rr_ceil_div(N > 0 ? N : 1, RR_BLOCK) * C + 1) *
                       sizeof(int));
}
static size_t hbv_tile_bytes(int64_t N, int64_t C)
{
    if (C < 1) C = 1;
    return hbv_queue_bytes(N, C) +
           rr_align256((size_t)5 * (size_t)rr_ceil_div(N > 0 ? N : 1, RR_BLOCK) *
                       (size_t)C * RR_BLOCK * sizeof(double));
}

extern "C" size_t rr_hbvedu_workspace_bytes(int64_t T, int64_t N)
{
    if (T < 0) T = 0;
    // + 2: the spare records the three-record loop may touch; + the
    // pre-pass's flags of odd precipitation values (one int per 256 days)
    // behind it
    if (T < 1) T = 1;
    return hbv_forcing_bytes(T, 1) + hbv_tile_bytes(N, 1);
}

// Shared by the single- and multi-catchment entry points.
static int hbv_launch(const double *temp, const double *prec,
                      const int8_t *month, const double *PE_m,
                      const double *T_m, int64_t T, int64_t C,
                      double snow_init, double soil_init, double s1_init,
                      double s2_init, const double *inits,
                      const double *params, int64_t N, double *qsim,
                      double *snow, double *soil, double *s1, double *s2,
                      int64_t ld, const double *qobs, double *sse,
                      void *workspace, hipStream_t st)
{
    if (T > 0x7fffffff) {
        rr_set_error("HBV-Edu: T=%lld timesteps; supported up to 2^31 - 1",
                     (long long)T);
        return RR_E_SIZE;
    }
    if (ld > 0x7ffffff) {
        // (the unrolled loops address a trip's rows through 32-bit offsets
        // of up to 3 * ld * 8 bytes; 134 million columns x 8 B x T > any HBM)
        rr_set_error("HBV-Edu: ld=%lld columns; supported up to 2^27 - 1",
                     (long long)ld);
        return RR_E_SIZE;
    }
    HbvDay *days = (HbvDay *)workspace;
    int *day_flags = (int *)(days + (size_t)T * (size_t)C + 2);
    double *dtemp_raw = (double *)((char *)workspace +
                                   hbv_forcing_bytes(T, C) -
                                   rr_align256((size_t)T * (size_t)C * 8));
    hipLaunchKernelGGL(hbv_pack_forcing,
                       dim3((unsigned)rr_ceil_div(T, 256), (unsigned)C),
                       dim3(256), 0, st, temp, prec, month, PE_m, T_m,
                       (qobs && sse) ? qobs : nullptr, T, days, day_flags,
                       dtemp_raw);
    const dim3 grid((unsigned)rr_ceil_div(N, RR_BLOCK), (unsigned)C);
    const bool any_s = snow != nullptr;
    const int64_t waves = rr_ceil_div(N, RR_BLOCK) * C;
    // (waves per SIMD of the device the process sees: 1024 SIMDs on a whole
    // MI355X)
    const int64_t simds = rr_simd_count();
    // Loop form by sweep size.  Up to six waves per SIMD a day is a latency
    // chain per wave, and the loop that asks for its record two days ahead
    // (3) wins (kernel ms, plain loop / three records, with qsim: 65k sets
    // 3.04 / 2.48, 125k 3.08 / 2.88, 250k 4.96 / 4.76, 375k 7.83 / 6.97);
    // beyond, where a SIMD always has a wave ready, the plain loop in time
    // tiles (400k 7.02 / 8.01, 750k 12.84 / 13.61; profiles/r05_mid_sizes.txt).
    // RR_OPT_HBV_VARIANT pins 0 or 3 (tests: the two must agree bit for bit).
    // The records' lines are fetched into every XCD's L2 by the waves
    // themselves when they start (the kernel's prologue) where a SIMD holds
    // one wave, and in the score-only mode at any size (kernel ms without /
    // with, profiles/r05_hbv_warm_ab.txt: 65,536 sets 2.20 / 1.51; scores only
    // 125k 2.16 / 1.98) -- but not with qsim written at two to four waves per
    // SIMD (100k 2.19 / 2.27, 250k 4.85 / 5.12): there the store stream is
    // what the sweep waits for, and waves that queue behind their XCD's
    // leading wave write the same rows at the same time, which the memory
    // system likes better than 2,000 waves each at a row of its own.
    const bool score_only = qsim == nullptr && snow == nullptr;
    const int warm = rr_warm_choice(waves, simds, score_only);
    const int64_t many_waves = 6 * simds;
    int variant = waves > many_waves ? 0 : 3;
    const int64_t pinned = rr_option(RR_OPT_HBV_VARIANT);
    if (pinned == 0 || pinned == 3) variant = (int)pinned;   // (the setters
                                                // accept -1, 0 and 3 only)
    // Time-tiled persistent form (hbvedu_kernel's TILED) of the plain loop:
    // sweeps of many rounds of waves, where equal-length waves quantise the
    // kernel time to whole wave slots (1M sets: 16 slots for 15.26 slots of
    // work): four pieces, as many waves as are resident.  (At two waves per
    // SIMD tiles lose -- nobody covers an item's own latencies: 125k sets 2.82
    // ms untiled, 3.31 with 11 pieces.)  RR_OPT_TIME_TILES: -1 by sweep size,
    // 0 never, k > 1 pieces.
    // The plain loop exists in the persistent form only: where it is not
    // tiled (a pinned variant 0 on a small sweep, RR_OPT_TIME_TILES = 0, a
    // series of a few days) the same kernel runs with one piece -- the tests
    // that pin variant 0 exercise the kernel the big sweeps run.
    int pieces = 0;
    if (variant == 0) {
        const int64_t opt = rr_option(RR_OPT_TIME_TILES);
        pieces = 1;
        if (T > 16 && waves * (opt > 1 ? opt : 64) < 0x7fffffff) {
            if (opt > 1) pieces = (int)opt;
            else if (opt < 0 && waves > many_waves) pieces = 4;
        }
    }
    int *queue = nullptr;
    double *tile_state = nullptr;
    if (pieces > 0) {
        queue = (int *)((char *)workspace + hbv_forcing_bytes(T, C));
        tile_state = (double *)((char *)queue + hbv_queue_bytes(N, C));
        RR_HIP(hipMemsetAsync(queue, 0, hbv_queue_bytes(N, C), st));
    }
    rr_dispatch3(qsim != nullptr, any_s, qobs && sse,
                 [&](auto Q, auto S, auto E) {
        auto go = [&](auto V) {
            if constexpr (V.value == 0) {
                auto kern = C == 1 ? hbvedu_kernel<Q.value, S.value, E.value,
                                                   0, true, 1>
                                   : hbvedu_kernel<Q.value, S.value, E.value,
                                                   0, true, 2>;
                // as many waves as are resident at once, no more
                int per_cu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
                        &per_cu, kern, RR_BLOCK, 0) != hipSuccess ||
                    per_cu < 1) {
                    (void)hipGetLastError();
                    per_cu = 16;
                }
                int64_t resident = (int64_t)per_cu * (simds / 4);
                const int64_t items = (int64_t)pieces * waves;
                if (resident > items) resident = items;
                kern<<<dim3((unsigned)resident), dim3(RR_BLOCK), 0, st>>>(
                    days, T, snow_init, soil_init, s1_init, s2_init, inits,
                    params, N, qsim, snow, soil, s1, s2, ld, qobs, sse,
                    day_flags, dtemp_raw, queue, tile_state, pieces, (int)C,
                    warm);
            } else {
                hbvedu_kernel<Q.value, S.value, E.value, V.value, true>
                    <<<grid, dim3(RR_BLOCK), 0, st>>>(
                        days, T, snow_init, soil_init, s1_init, s2_init,
                        inits, params, N, qsim, snow, soil, s1, s2, ld, qobs,
                        sse, day_flags, dtemp_raw, nullptr, nullptr, 0,
                        (int)C, warm);
            }
        };
        if (variant == 3) go(std::integral_constant<int, 3>{});
        else go(std::integral_constant<int, 0>{});
        // ... and, right behind it, the kernel of the waves that hold a set
        // which is not civil (hbv_civil_lane): every other wave returns at
        // once (30 us of a million-set sweep)
        hbvedu_kernel<Q.value, S.value, E.value, 0, false, 0, true>
            <<<grid, dim3(RR_BLOCK), 0, st>>>(
                days, T, snow_init, soil_init, s1_init, s2_init, inits,
                params, N, qsim, snow, soil, s1, s2, ld, qobs, sse,
                day_flags, dtemp_raw, nullptr, nullptr, 0, (int)C, 0);
    });
    RR_HIP(hipGetLastError());
    return RR_OK;
}

static int hbv_check(const char *who, const void *temp, const void *prec,
                     const void *month, const void *PE_m, const void *T_m,
                     const double *qsim, const double *snow,
                     const double *soil, const double *s1, const double *s2)
{
    if (!temp || !prec || !month || !PE_m || !T_m) {
        rr_set_error("%s: NULL forcing pointer", who);
        return RR_E_NULL;
    }
    const bool any_s = snow || soil || s1 || s2;
    if (any_s && !(snow && soil && s1 && s2)) {
        rr_set_error("%s: pass all four storage outputs or none", who);
        return RR_E_NULL;
    }
    return rr_check_outputs(who, qsim, any_s);
}

extern "C" int rr_hbvedu_simulate_dev(
    const double *temp, const double *prec, const int8_t *month,
    const double *PE_m, const double *T_m, int64_t T, double snow_init,
    double soil_init, double s1_init, double s2_init, const double *params,
    int64_t N, double *qsim, double *snow, double *soil, double *s1,
    double *s2, int64_t ld, const double *qobs, double *sse, void *workspace,
    size_t workspace_bytes, void *stream)
{
    const char *who = "rr_hbvedu_simulate_dev";
    int rc = rr_check_common(who, T, N, ld, params, qobs, sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    rc = hbv_check(who, temp, prec, month, PE_m, T_m, qsim, snow, soil, s1,
                   s2);
    if (rc != RR_OK) return rc;
    if (!workspace || workspace_bytes < rr_hbvedu_workspace_bytes(T, N)) {
        rr_set_error("%s: workspace too small", who);
        return RR_E_WORKSPACE;
    }
    return hbv_launch(temp, prec, month, PE_m, T_m, T, 1, snow_init,
                      soil_init, s1_init, s2_init, nullptr, params, N, qsim,
                      snow, soil, s1, s2, ld, qobs, sse, workspace,
                      (hipStream_t)stream);
}

extern "C" size_t rr_hbvedu_catchments_workspace_bytes(int64_t T, int64_t C,
                                                       int64_t N)
{
    if (T < 1) T = 1;
    if (C < 1) C = 1;
    return hbv_forcing_bytes(T, C) + hbv_tile_bytes(N, C);
}

extern "C" int rr_hbvedu_simulate_catchments_dev(
    const double *temp, const double *prec, const int8_t *month,
    const double *PE_m, const double *T_m, int64_t T, int64_t C,
    const double *inits, const double *params, int64_t N, double *qsim,
    double *snow, double *soil, double *s1, double *s2, int64_t ld,
    const double *qobs, double *sse, void *workspace, size_t workspace_bytes,
    void *stream)
{
    const char *who = "rr_hbvedu_simulate_catchments_dev";
    int rc = rr_check_common(who, T, N, ld, params, qobs, sse);
    if (rc != RR_OK) return rc;
    if (C < 0 || C > 65535) {
        rr_set_error("%s: C=%lld catchments; supported 0..65535", who,
                     (long long)C);
        return RR_E_SIZE;
    }
    if (T == 0 || N == 0 || C == 0) return RR_OK;
    rc = hbv_check(who, temp, prec, month, PE_m, T_m, qsim, snow, soil, s1,
                   s2);
    if (rc != RR_OK) return rc;
    if (!inits) {
        rr_set_error("%s: inits is NULL", who);
        return RR_E_NULL;
    }
    if (!workspace ||
        workspace_bytes < rr_hbvedu_catchments_workspace_bytes(T, C, N)) {
        rr_set_error("%s: workspace too small", who);
        return RR_E_WORKSPACE;
    }
    return hbv_launch(temp, prec, month, PE_m, T_m, T, C, 0., 0., 0., 0.,
                      inits, params, N, qsim, snow, soil, s1, s2, ld, qobs,
                      sse, workspace, (hipStream_t)stream);
}
