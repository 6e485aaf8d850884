// gr4j.hip -- GR4J ensemble kernel for gfx950.
//
// Replaces run_gr4j (reference: rrmpg/models/gr4j_model.py:15-157) and the
// per-set Python loop in GR4J.simulate (reference: rrmpg/models/gr4j.py:
// 162-183; its early `return` after the first set, gr4j.py:176-178, is NOT
// reproduced -- every column is filled).
//
// One lane per parameter set; S and R in registers; the unit-hydrograph
// convolution state in registers (x4 <= 3 launch-wide) or staged in LDS
// (gr4j_core.h).  The shared {prec, etp} day record is wave-uniform and is
// fetched with one scalar s_load_dwordx4 per day.
#include "gr4j_core.h"
#include "gr4j_reference.h"

// One day of shared forcing as the kernel wants it (fetched by value with one
// s_load_dwordx8 per day).  Which branch of the reference's net-rainfall /
// net-evaporation split applies (gr4j_model.py:89-111) and the net amount do
// not depend on the parameters, so the pre-pass evaluates them once per day
// instead of every lane doing it with vector instructions.
struct __attribute__((aligned(32))) GrDay {
    double net;      // prec - etp if wet else etp - prec (:90, :102)
    double qobs;     // the day's observation (0 when no metric is fused)
    int wet;         // prec >= etp (:89)
    int net_ok;      // net is a numerator the 3-FMA quotient net/x1 serves
                     // (gr4j_core.h gr4j_num_ok: +0 or positive and below
                     // 2^196), decided here once per day instead of by
                     // vector instructions in every wave
    int pad[2];
};

// plan[3] counts the forcing values that are not civil (gr4j_reference.h):
// with any, every set of the launch gets the reference's own sequence.
__global__ void gr4j_pack_forcing(const double *__restrict__ prec,
                                  const double *__restrict__ etp,
                                  const double *__restrict__ qobs, int64_t T,
                                  GrDay *__restrict__ days,
                                  int *__restrict__ plan)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const double p = prec[t], e = etp[t];
    if (!gr4j_civil_forcing(p) || !gr4j_civil_forcing(e))
        atomicAdd(&plan[3], 1);
    GrDay d;
    d.wet = p >= e;
    d.net = d.wet ? p - e : e - p;
    d.qobs = qobs ? qobs[t] : 0.0;
    d.net_ok = gr4j_num_ok(d.net);
    d.pad[0] = d.pad[1] = 0;
    days[t] = d;
}

// scan[0] = max over sets of ceil(x4) (as int, saturated), scan[1] = number
// of sets whose x4 gives no ordinates (ceil(x4) < 1 or NaN).
__global__ void gr4j_scan_x4(const double *__restrict__ params, int64_t N,
                             int stride, int x4_index, int mem_cap,
                             int *__restrict__ scan)
{
    // scan[2]: what the unit-hydrograph scratch of this launch can hold
    if (blockIdx.x == 0 && threadIdx.x == 0) scan[2] = mem_cap;
    int mx = 0, bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double x4 = params[i * stride + x4_index];
        const int n1 = gr4j_num_uh1(x4);
        if (n1 < 1) bad++;
        mx = n1 > mx ? n1 : mx;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const int o = __shfl_xor(mx, m, 64);
        mx = o > mx ? o : mx;
        bad += __shfl_xor(bad, m, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&scan[0], mx);
        if (bad) atomicAdd(&scan[1], bad);
    }
}

// Minimum waves per SIMD the register allocation is held to: 5 (<= 96 VGPRs)
// for the 3+7-register tier -- the polynomial's coefficients sit in VGPR pairs
// there (gr4j_core.h), and above four waves the fp64 issue rate no longer
// depends on the count (profiles/r02_valu_cost.txt) --, 6 for the LDS tier.
template <class UH>
constexpr int gr4j_min_waves()
{
    return std::is_same<UH, UhRegs<3>>::value ? 5 : uh_is_indexed<UH> ? 6 : 2;
}

template <class UH, bool Q, bool S, bool E>
__global__ __launch_bounds__(RR_BLOCK, gr4j_min_waves<UH>())
void gr4j_kernel(
    const GrDay *__restrict__ days, int64_t T, double s_init, double r_init,
    const double *__restrict__ params, int64_t N,
    const int *__restrict__ plan, int force_lds,
    double *__restrict__ qsim, double *__restrict__ s_store,
    double *__restrict__ r_store, int64_t ld,
    const double *__restrict__ qobs, double *__restrict__ sse,
    double *__restrict__ uh_mem)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int n1cap, n2cap;
    if (!gr4j_plan_selects<UH>(plan, force_lds, n1cap, n2cap)) return;
    const int lane = (int)threadIdx.x;
    const int64_t first = (int64_t)blockIdx.x * RR_BLOCK;
    const int64_t i = first + lane;
    const bool active = i < N;
    const double *p = params + (active ? i : N - 1) * 4;
    Gr4jPar P;
    P.set(p[0], p[1], p[2], p[3]);

    UH uh;
    gr4j_uh_init(uh, lds, uh_mem, n1cap, n2cap, P.x4);

    double s = s_init * P.x1;   // gr4j_model.py:64
    double r = r_init * P.x3;   // gr4j_model.py:65
    double acc = 0.0;
    // output rows: wave-uniform base + lane offset (common.h rr_store_row)
    const int lane_off = lane * 8;
    const unsigned row_bytes = rr_row_bytes(first, N);
    int64_t row = first;

    // The day record is wave-uniform (one s_load_dwordx8).  Day k+1's is
    // requested in the middle of day k, so that no wave sits out the load's
    // latency at the top of every day (sweeps of one or two waves per SIMD
    // cannot hide it: 65k sets 7.5 -> 5.8 ms, 125k 8.8 -> 8.0, a million
    // 56.6 -> 56.2).
    typedef const GrDay __attribute__((address_space(4))) *day_ptr_t;
    const day_ptr_t dp = (day_ptr_t)days;
    GrDay f;
    f.net = dp[0].net; f.qobs = dp[0].qobs; f.wet = dp[0].wet;
    f.net_ok = dp[0].net_ok;
    for (int64_t k = 0; k < T; ++k) {
        const double net = f.net, qobs_k = f.qobs;
        const bool wet = f.wet != 0;
        const lanemask_t net_m = f.net_ok ? ~0ull : 0ull;
        auto fetch_next = [&]() {
            day_ptr_t nx = dp + (k + 1);
            asm volatile("" : "+s"(nx));         // keeps the load at this spot
            f.net = nx->net; f.qobs = nx->qobs; f.wet = nx->wet;
            f.net_ok = nx->net_ok;
        };
        const double q = gr4j_step_net<UH, GR4J_CONSTS_SGPR>(
            P, s, r, uh, net, wet, net_m, fetch_next);
        if (Q) rr_store_row(qsim + row, row_bytes, lane_off, q);
        if (S) {
            rr_store_row(s_store + row, row_bytes, lane_off, s);
            rr_store_row(r_store + row, row_bytes, lane_off, r);
        }
        if (E) {
            const double d = qobs_k - q;
            acc = __builtin_fma(d, d, acc);
        }
        row += ld;
    }
    if (E && active) sse[i] = acc;
}

// ---- optimistic variant: a branch-free day ----------------------------------
// Every fast form of the day is guarded by a wave vote, and each vote costs
// the loop two scalar instructions and a branch (twelve branches a day in
// gr4j_kernel).  With four or more waves on a SIMD somebody else's vector
// work fills those issue turns; a sweep of one or two waves per SIMD -- one
// GPU's shard of a strong-scaled sweep, every `fit` population -- pays for
// each of them (measured: 2.4 ns per vector instruction at two waves per
// SIMD against 1.94 at five, with the vector pipe 20 % idle).  Here the votes
// only note their lanes (OptimisticVotes, common.h: one scalar AND each), the
// day runs the fast forms straight through, and ONE branch at its end asks
// whether any vote failed; if so the day is redone from its start state with
// the deciding votes of gr4j_kernel.  That start state is still there
// because the states exist in two generations -- production store, routing
// store and the hydrograph slots of day k are read from one and written to
// the other, two days per trip -- which costs 12 register pairs (3+7 slots)
// and not a single move.  Bit-identical to gr4j_kernel: a lane inside every
// domain gets the fast forms' values either way, and a day with a lane
// outside one IS gr4j_kernel's day.
template <class UH>
constexpr bool gr4j_has_optimistic()
{
    return std::is_same<UH, UhRegs<3>>::value ||
           std::is_same<UH, UhRegs<5>>::value;
}

#define GR4J_OPT_CONSTS GR4J_CONSTS_SGPR
#define GR4J_OPT_MINWAVES (std::is_same<UH, UhRegs<3>>::value ? 4 : 3)
// TILED: the time axis in pieces handed out by a ticket counter
// (common.h RrTiles: million-set sweeps); handed over: both stores, the
// hydrograph slots, the score sum.  Pieces hold an even number of days, so
// every piece starts in the same state generation.
template <class UH>
constexpr int gr4j_tile_states()
{
    return 3 + UH::TIER + (2 * UH::TIER + 1);       // s, r, acc + slots
}

// TILED: one workgroup per item (the item: a ticket from the queue's counter).
template <class UH, bool Q, bool S, bool E, bool TILED = false>
__global__ __launch_bounds__(RR_BLOCK, GR4J_OPT_MINWAVES)
void gr4j_opt_kernel(
    const GrDay *__restrict__ days, int64_t T, double s_init, double r_init,
    const double *__restrict__ params, int64_t N,
    const int *__restrict__ plan, int force_lds,
    double *__restrict__ qsim, double *__restrict__ s_store,
    double *__restrict__ r_store, int64_t ld,
    const double *__restrict__ qobs, double *__restrict__ sse, RrTiles tiles)
{
    int n1cap, n2cap;
    if (!gr4j_plan_selects<UH>(plan, force_lds, n1cap, n2cap)) return;
    const int njobs = (int)((N + RR_BLOCK - 1) / RR_BLOCK);
    typedef const GrDay __attribute__((address_space(4))) *day_ptr_t;
    const day_ptr_t dp = (day_ptr_t)days;
    if (tiles.warm) rr_warm_l2(days, T * (int64_t)sizeof(GrDay));
    // (TILED: one single-wave workgroup per item, the item a ticket drawn
    // when the wave starts -- common.h "the time axis in pieces"; a
    // persistent loop around this kernel's two-generation day cost scratch
    // spills)
    int job = blockIdx.x, piece = 0;
    if constexpr (TILED) {
        const int item = rr_tile_ticket(tiles);
        piece = __builtin_amdgcn_readfirstlane(item / njobs);
        job = __builtin_amdgcn_readfirstlane(item - piece * njobs);
    }
    {
    const int64_t i = (int64_t)job * RR_BLOCK + threadIdx.x;
    const bool active = i < N;
    const double *p = params + (active ? i : N - 1) * 4;
    Gr4jPar P;
    P.set(p[0], p[1], p[2], p[3]);
    UH uh;
    typename UH::Slots ua, ub;
    uh.init(P.x4, ua);
    double sa = s_init * P.x1, sb;   // gr4j_model.py:64
    double ra = r_init * P.x3, rb;   // gr4j_model.py:65
    double acc = 0.0;
    const int lane_off = threadIdx.x * 8;
    const int64_t first = (int64_t)job * RR_BLOCK;
    const unsigned row_bytes = rr_row_bytes(first, N);
    int k_begin = 0, k_end = (int)T;
    double *const hand = TILED ? tiles.state + ((int64_t)job * RR_BLOCK +
                                                threadIdx.x) : nullptr;
    const int64_t hs = (int64_t)njobs * RR_BLOCK;
    if constexpr (TILED) {
        rr_tile_range(0, (int)T, tiles.pieces, piece, 2, k_begin, k_end);
        if (piece > 0) {
            rr_tile_wait(tiles, job, piece);
            sa = hand[0];
            ra = hand[hs];
            acc = hand[2 * hs];
#pragma unroll
            for (int j = 0; j < UH::TIER; ++j) ua.u1[j] = hand[(3 + j) * hs];
#pragma unroll
            for (int j = 0; j < UH::N2MAX; ++j)
                ua.u2[j] = hand[(3 + UH::TIER + j) * hs];
        }
    }
    int64_t row = first + (int64_t)k_begin * ld;
    // Two day records alternate with the two state generations: day k reads
    // its record at its top and, in its middle, requests the record of day
    // k + 2 into the same registers -- a day and a half ahead (the records of
    // two more days than the run has are there to be touched).  Scalar loads
    // return out of order, so every wait is for all of them: the `use` of the
    // OTHER record right before the request makes that wait explicit where
    // the record, asked for a day ago, has long arrived.  (Until round 4: one
    // record, requested half a day ahead.)
    GrDay fa, fb;
    fa.net = dp[k_begin].net; fa.qobs = dp[k_begin].qobs;
    fa.wet = dp[k_begin].wet; fa.net_ok = dp[k_begin].net_ok;
    fb.net = dp[k_begin + 1].net; fb.qobs = dp[k_begin + 1].qobs;
    fb.wet = dp[k_begin + 1].wet; fb.net_ok = dp[k_begin + 1].net_ok;
    auto day = [&](GrDay &f, const GrDay &other, const double s_in,
                   const double r_in, const typename UH::Slots &u_in,
                   double &s_out, double &r_out, typename UH::Slots &u_out,
                   int k)
        __attribute__((always_inline)) {
        const double net = f.net, qobs_k = f.qobs;
        const Gr4jUniformWet wet = {f.wet};     // (gr4j_core.h)
        const lanemask_t net_m = f.net_ok ? ~0ull : 0ull;
        auto fetch_next = [&]() {
            asm volatile("" : : "s"(other.net), "s"(other.qobs),
                         "s"(other.wet), "s"(other.net_ok));
            day_ptr_t nx = dp + (k + 2);
            asm volatile("" : "+s"(nx));         // keeps the load at this spot
            f.net = nx->net; f.qobs = nx->qobs; f.wet = nx->wet;
            f.net_ok = nx->net_ok;
        };
        OptimisticVotes votes;
        double s = s_in, r = r_in;
        double p_r = gr4j_production<UH, GR4J_OPT_CONSTS>(
            P, s, net, wet, net_m, fetch_next, votes);
        double q = gr4j_routing<UH>(P, r, uh, u_in, u_out, p_r, votes);
        if (RR_VOTES_FAILED(votes)) {
            // some lane left a fast form's domain: this day again, from its
            // untouched start state, every vote decided on the spot
            asm volatile("");
            s = s_in;
            r = r_in;
            p_r = gr4j_production<UH, GR4J_OPT_CONSTS>(P, s, net, wet, net_m);
            q = gr4j_routing<UH>(P, r, uh, u_in, u_out, p_r);
        }
        s_out = s;
        r_out = r;
        if (Q) rr_store_row(qsim + row, row_bytes, lane_off, q);
        if (S) {
            rr_store_row(s_store + row, row_bytes, lane_off, s);
            rr_store_row(r_store + row, row_bytes, lane_off, r);
        }
        if (E) {
            const double d = qobs_k - q;
            acc = __builtin_fma(d, d, acc);
        }
        row += ld;
    };
    for (int k = k_begin; k < k_end; k += 2) {
        day(fa, fb, sa, ra, ua, sb, rb, ub, k);
        if (k + 1 < k_end) day(fb, fa, sb, rb, ub, sa, ra, ua, k + 1);
    }
    if (TILED && piece + 1 < tiles.pieces) {
        // (an even number of days: the states are back in generation a)
        hand[0] = sa;
        hand[hs] = ra;
        hand[2 * hs] = acc;
#pragma unroll
        for (int j = 0; j < UH::TIER; ++j) hand[(3 + j) * hs] = ua.u1[j];
#pragma unroll
        for (int j = 0; j < UH::N2MAX; ++j)
            hand[(3 + UH::TIER + j) * hs] = ua.u2[j];
        rr_tile_publish(tiles, job, piece);
    } else {
        if (E && active) sse[i] = acc;
    }
    }
}

// (A wave-specialised variant -- the day's production and routing halves in
// two waves of a workgroup, an LDS ring between them -- was built in round 3:
// bit-identical, it wins at one wave per SIMD only and moves nothing at two
// (125k sets).  Removed in round 6 together with gr4j_kernel in workgroups of
// four waves and the persistent-wave form of the tiled optimistic kernel;
// their A/B tables: profiles/README.md.)

// ---- the reference's own sequence for the sets that are not civil ---------
// (gr4j_reference.h)  One lane per set, launched behind the fast kernels;
// a civil set's lane returns at once, every other one runs the reference's
// run_gr4j and overwrites its columns and its score.  A launch whose plan
// runs nothing (a set without unit-hydrograph ordinates) writes nothing here
// either.
__global__ __launch_bounds__(RR_BLOCK) void gr4j_reference_kernel(
    const double *__restrict__ prec, const double *__restrict__ etp,
    int64_t T, double s_init, double r_init,
    const double *__restrict__ params, int64_t N,
    const int *__restrict__ plan, double *__restrict__ qsim,
    double *__restrict__ s_store, double *__restrict__ r_store, int64_t ld,
    const double *__restrict__ qobs, double *__restrict__ sse)
{
    const int64_t i = (int64_t)blockIdx.x * RR_BLOCK + threadIdx.x;
    if (i >= N) return;
    if (gr4j_plan_tier(plan[0], plan[1], 0, plan[2]) < 0) return;
    const double *p = params + i * 4;
    if (plan[3] == 0 && gr4j_civil_set(p[0], p[1], p[2], s_init, r_init))
        return;
    Gr4jRef g;
    if (!g.init(p[0], p[1], p[2], p[3], s_init, r_init)) return;
    double acc = 0.0;
    for (int64_t k = 0; k < T; ++k) {
        const double q = g.day(prec[k], etp[k]);
        if (qsim) rr_out(&qsim[k * ld + i], q);
        if (s_store) {
            rr_out(&s_store[k * ld + i], g.s);
            rr_out(&r_store[k * ld + i], g.r);
        }
        if (sse) {
            const double d = qobs[k] - q;
            acc = __builtin_fma(d, d, acc);
        }
    }
    if (sse) sse[i] = acc;
}

// plan + day records (+ three spare records: the kernels request the record
// of up to two days ahead, the last day included)
static size_t gr4j_days_bytes(int64_t T)
{
    if (T < 1) T = 1;
    return 256 + rr_align256((size_t)(T + 3) * sizeof(GrDay));
}
// ... + the tiled kernels' work queue and hand-over scratch (common.h RrTiles)
#define GR4J_TILE_STATES (gr4j_tile_states<UhRegs<5>>())
extern "C" size_t rr_gr4j_workspace_bytes(int64_t T, int64_t N)
{
    return gr4j_days_bytes(T) + rr_tile_bytes(N, GR4J_TILE_STATES);
}

// ceil(x4) as the kernels count it, for sizing (host)
static int64_t gr4j_host_n1(double max_x4)
{
    if (!(max_x4 > 0)) return 0;
    const double c = ceil(max_x4);
    return c > 1e6 ? 1000000 : (int64_t)c;
}

// bytes of unit-hydrograph scratch behind a workspace whose launch may hold
// sets with x4 up to max_x4 (none up to RR_GR4J_MAX_X4: registers / LDS)
size_t rr_gr4j_uh_scratch_bytes(int64_t N, double max_x4)
{
    const int64_t n1 = gr4j_host_n1(max_x4);
    return n1 > (int64_t)RR_GR4J_MAX_X4 ? rr_align256(gr4j_mem_bytes(N, n1)) : 0;
}

extern "C" size_t rr_gr4j_workspace_bytes_x4(int64_t T, int64_t N,
                                             double max_x4)
{
    return rr_gr4j_workspace_bytes(T, N) + rr_gr4j_uh_scratch_bytes(N, max_x4);
}

// Shared by gr4j.hip, cemaneige.hip and snownext_kernels.h: enqueues the scan of
// x4 that leaves the plan {max ceil(x4), #bad sets, scratch capacity} in
// d_plan (gr4j_core.h).  Asynchronous: nothing is read back.
int rr_gr4j_plan_async(const double *params, int64_t N, int stride,
                       int x4_index, int *d_plan, int mem_cap, hipStream_t st)
{
    RR_HIP(hipMemsetAsync(d_plan, 0, GR4J_PLAN_INTS * sizeof(int), st));
    int blocks = (int)rr_ceil_div(N, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(gr4j_scan_x4, dim3(blocks), dim3(256), 0, st, params, N,
                       stride, x4_index, mem_cap, d_plan);
    return RR_OK;
}

// ---- side streams for a launch's tier kernels (snow_core.h) ------------------
// Per thread and device, created on first use, never destroyed (a handful of
// streams and events per calling thread).
namespace {
struct TierStreams {
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, done[3] = {nullptr, nullptr, nullptr};
    bool ok = false, failed = false;
};
constexpr int RR_TIER_MAX_DEVICES = 64;
thread_local TierStreams tl_tier[RR_TIER_MAX_DEVICES];
TierStreams *tier_streams()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 ||
        dev >= RR_TIER_MAX_DEVICES)
        return nullptr;
    TierStreams &t = tl_tier[dev];
    if (t.failed) return nullptr;       // (tried once: everything on `st`)
    if (!t.ok) {
        bool good = hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) ==
                    hipSuccess;
        for (int k = 0; k < 3 && good; ++k)
            good = hipStreamCreateWithFlags(&t.side[k],
                                            hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&t.done[k],
                                           hipEventDisableTiming) == hipSuccess;
        if (!good) {
            // give back what was created and do not try again: a later call
            // would create -- and leak -- the same handles once more
            for (int k = 0; k < 3; ++k) {
                if (t.side[k]) (void)hipStreamDestroy(t.side[k]);
                if (t.done[k]) (void)hipEventDestroy(t.done[k]);
                t.side[k] = nullptr;
                t.done[k] = nullptr;
            }
            if (t.fork) (void)hipEventDestroy(t.fork);
            t.fork = nullptr;
            t.failed = true;
            return nullptr;
        }
        t.ok = true;
    }
    return &t;
}
thread_local TierStreams *tl_tier_now = nullptr;
}  // namespace
int rr_tier_fork(hipStream_t st)
{
    tl_tier_now = tier_streams();
    if (!tl_tier_now) {
        (void)hipGetLastError();
        return RR_OK;                  // no side streams: everything on `st`
    }
    TierStreams *const t = tl_tier_now;
    bool good = hipEventRecord(t->fork, st) == hipSuccess;
    for (int k = 0; k < 3 && good; ++k)
        good = hipStreamWaitEvent(t->side[k], t->fork, 0) == hipSuccess;
    if (!good) {
        // nothing has been enqueued on a side stream yet: this launch runs
        // on `st` alone
        (void)hipGetLastError();
        tl_tier_now = nullptr;
    }
    return RR_OK;
}
hipStream_t rr_tier_stream(int k)
{
    return tl_tier_now ? tl_tier_now->side[k] : nullptr;
}
// Always leaves the thread without a fork (also on an error, so that the
// next launch starts clean); a side stream whose join could not be enqueued
// is waited for on the host -- `st` must not run ahead of kernels that write
// the same score vector.
int rr_tier_join(hipStream_t st)
{
    TierStreams *const t = tl_tier_now;
    tl_tier_now = nullptr;
    if (!t) return RR_OK;
    int rc = RR_OK;
    for (int k = 0; k < 3; ++k) {
        if (hipEventRecord(t->done[k], t->side[k]) == hipSuccess &&
            hipStreamWaitEvent(st, t->done[k], 0) == hipSuccess)
            continue;
        (void)hipGetLastError();
        if (hipStreamSynchronize(t->side[k]) != hipSuccess) {
            rr_set_error("rr_tier_join: side stream %d could not be joined", k);
            rc = RR_E_HIP;
        }
    }
    return rc;
}

// ---- the sets of a launch ordered by ceil(x4) -------------------------------
// For the per-wave tiers (gr4j_core.h gr4j_wave_selects) of a sweep that
// writes nothing but scores: a counting sort by n1 = ceil(x4) (64 bins; a
// set without ordinates goes to bin 0, n1 >= 63 to bin 63) into `perm` -- wave
// w then simulates the sets perm[64 w .. 64 w + 63] and writes their scores
// to their own places.  Three tiny kernels, about 30 us for a million sets;
// the order inside a bin is whatever the atomics give (a set's result does
// not depend on its wave-mates).
#define GR4J_SORT_BINS 64
__device__ __forceinline__ int gr4j_sort_key(double x4)
{
    const int n1 = gr4j_num_uh1(x4);
    return n1 < 1 ? 0 : (n1 >= GR4J_SORT_BINS ? GR4J_SORT_BINS - 1 : n1);
}
__global__ void gr4j_sort_count(const double *__restrict__ params, int64_t N,
                                int stride, int x4_index,
                                int *__restrict__ bins)
{
    __shared__ int local[GR4J_SORT_BINS];
    if (threadIdx.x < GR4J_SORT_BINS) local[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N;
         i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&local[gr4j_sort_key(params[i * stride + x4_index])], 1);
    __syncthreads();
    if (threadIdx.x < GR4J_SORT_BINS && local[threadIdx.x])
        atomicAdd(&bins[threadIdx.x], local[threadIdx.x]);
}
__global__ void gr4j_sort_offsets(int *__restrict__ bins)
{
    // bins[0..63]: counts -> bins[64..127]: where each bin starts
    if (threadIdx.x == 0) {
        int at = 0;
        for (int b = 0; b < GR4J_SORT_BINS; ++b) {
            bins[GR4J_SORT_BINS + b] = at;
            at += bins[b];
        }
    }
}
__global__ void gr4j_sort_place(const double *__restrict__ params, int64_t N,
                                int stride, int x4_index,
                                int *__restrict__ bins, int *__restrict__ perm)
{
    __shared__ int local[GR4J_SORT_BINS], base[GR4J_SORT_BINS];
    int *cursor = bins + GR4J_SORT_BINS;
    const int64_t chunk = (int64_t)blockIdx.x * blockDim.x;
    if (threadIdx.x < GR4J_SORT_BINS) local[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = chunk + threadIdx.x;
    int key = 0, rank = 0;
    if (i < N) {
        key = gr4j_sort_key(params[i * stride + x4_index]);
        rank = atomicAdd(&local[key], 1);
    }
    __syncthreads();
    if (threadIdx.x < GR4J_SORT_BINS && local[threadIdx.x])
        base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], local[threadIdx.x]);
    __syncthreads();
    if (i < N) perm[base[key] + rank] = (int)i;
}
int rr_gr4j_tier_sort_async(const double *params, int64_t N, int stride,
                            int x4_index, int *bins, int *perm,
                            hipStream_t st)
{
    RR_HIP(hipMemsetAsync(bins, 0, 2 * GR4J_SORT_BINS * sizeof(int), st));
    int blocks = (int)rr_ceil_div(N, 256);
    hipLaunchKernelGGL(gr4j_sort_count, dim3(blocks > 1024 ? 1024 : blocks),
                       dim3(256), 0, st, params, N, stride, x4_index, bins);
    hipLaunchKernelGGL(gr4j_sort_offsets, dim3(1), dim3(64), 0, st, bins);
    hipLaunchKernelGGL(gr4j_sort_place, dim3(blocks), dim3(256), 0, st, params,
                       N, stride, x4_index, bins, perm);
    return RR_OK;
}

extern "C" int rr_gr4j_plan_status(const void *workspace, void *stream)
{
    if (!workspace) {
        rr_set_error("rr_gr4j_plan_status: workspace is NULL");
        return RR_E_NULL;
    }
    hipStream_t st = (hipStream_t)stream;
    int h[GR4J_PLAN_INTS] = {0, 0, 0, 0};
    RR_HIP(hipMemcpyAsync(h, workspace, sizeof(h), hipMemcpyDeviceToHost, st));
    RR_HIP(hipStreamSynchronize(st));
    if (h[1] > 0) {
        rr_set_error("GR4J: %d parameter set(s) have ceil(x4) < 1 (or NaN): "
                     "the unit hydrograph would have no ordinates (the "
                     "reference raises IndexError there)", h[1]);
        return RR_E_PARAM;
    }
    if ((double)h[0] > RR_GR4J_MAX_X4 && h[0] > h[2]) {
        rr_set_error("GR4J: x4 up to %d needs a unit-hydrograph scratch behind "
                     "the workspace (this one holds x4 <= %d): size the "
                     "workspace with rr_*_workspace_bytes_x4", h[0],
                     h[2] > (int)RR_GR4J_MAX_X4 ? h[2] : (int)RR_GR4J_MAX_X4);
        return RR_E_PARAM;
    }
    return RR_OK;
}

extern "C" int rr_gr4j_simulate_dev(const double *prec, const double *etp,
                                    int64_t T, double s_init, double r_init,
                                    const double *params, int64_t N,
                                    double *qsim, double *s_store,
                                    double *r_store, int64_t ld,
                                    const double *qobs, double *sse,
                                    void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    int rc = rr_check_common("rr_gr4j_simulate_dev", T, N, ld, params, qobs,
                             sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (!prec || !etp) {
        rr_set_error("rr_gr4j_simulate_dev: NULL forcing pointer");
        return RR_E_NULL;
    }
    if ((s_store == nullptr) != (r_store == nullptr)) {
        rr_set_error("rr_gr4j_simulate_dev: pass both storage outputs or none");
        return RR_E_NULL;
    }
    if ((rc = rr_check_outputs("rr_gr4j_simulate_dev", qsim,
                               s_store != nullptr)) != RR_OK)
        return rc;
    if (!workspace || workspace_bytes < rr_gr4j_workspace_bytes(T, N)) {
        rr_set_error("rr_gr4j_simulate_dev: workspace too small");
        return RR_E_WORKSPACE;
    }
    if (T > 2000000000) {
        rr_set_error("rr_gr4j_simulate_dev: T = %lld exceeds 2e9 timesteps",
                     (long long)T);
        return RR_E_SIZE;
    }
    hipStream_t st = (hipStream_t)stream;
    int *d_plan = (int *)workspace;
    GrDay *days = (GrDay *)((char *)workspace + 256);
    // whatever lies behind the base workspace is unit-hydrograph scratch
    const size_t base_ws = rr_gr4j_workspace_bytes(T, N);
    double *uh_mem = (double *)((char *)workspace + base_ws);
    const int mem_cap = gr4j_mem_cap(workspace_bytes - base_ws, N);
    rc = rr_gr4j_plan_async(params, N, 4, 3, d_plan, mem_cap, st);
    if (rc != RR_OK) return rc;
    // a block the plan cannot run (no tier selected) writes nothing: its
    // scores then read NaN, not whatever the buffer held
    if (qobs && sse)
        RR_HIP(hipMemsetAsync(sse, 0xFF, (size_t)N * sizeof(double), st));
    hipLaunchKernelGGL(gr4j_pack_forcing, dim3((unsigned)rr_ceil_div(T, 256)),
                       dim3(256), 0, st, prec, etp, qobs, T, days, d_plan);
    const dim3 grid((unsigned)rr_ceil_div(N, RR_BLOCK)), block(RR_BLOCK);
    const bool q = qsim != nullptr, s = s_store != nullptr, e = qobs && sse;
    const int force_lds = (int)rr_option(RR_OPT_GR4J_FORCE_LDS);
    // every tier is enqueued; the kernels pick the one the plan selects
    // (RR_OPT_GR4J_VARIANT 1: gr4j_kernel, every vote decided on the spot, in
    // the tiers that have the optimistic kernel too -- tests: the same bits)
    const bool careful = rr_option(RR_OPT_GR4J_VARIANT) == 1;
    const int64_t waves = rr_ceil_div(N, RR_BLOCK);
    // time tiles of the optimistic kernel (common.h RrTiles)
    // (the day records prefetched into the XCDs' L2, common.h rr_warm_l2:
    // only on request -- GR4J's day is issue-bound and its record asked for a
    // day and a half ahead: 65,536 sets 3.66 / 3.64 ms without / with,
    // scores only 3.37 / 3.34, 125k with qsim 5.71 / 6.00,
    // profiles/r05_warm_family_ab.txt)
    RrTiles tiles = {nullptr, nullptr, 0,
                     rr_option(RR_OPT_WARM_RECORDS) == 1 ? 1 : 0};
    {
        const int64_t opt = rr_option(RR_OPT_TIME_TILES);
        int pieces = 0;
        if (!careful && T > 16) {
            if (opt > 1) pieces = (int)opt;
            else if (opt < 0 && waves > 6 * (int64_t)rr_simd_count()) pieces = 4;
        }
        if (pieces > 1) {
            tiles.queue = (int *)((char *)workspace + gr4j_days_bytes(T));
            tiles.state = (double *)((char *)tiles.queue +
                                     rr_tile_queue_bytes(N));
            tiles.pieces = pieces;
            RR_HIP(hipMemsetAsync(tiles.queue, 0, rr_tile_queue_bytes(N), st));
        }
    }
    rr_dispatch3(q, s, e, [&](auto Q, auto S, auto E) {
        gr4j_for_each_tier([&](auto uh) {
            using UH = decltype(uh);
            constexpr size_t lds =
                std::is_same<UH, UhLds>::value ? GR4J_LDS_BYTES : 0;
            if constexpr (gr4j_has_optimistic<UH>()) {
                // (the default wherever it exists: faster at every sweep
                // size, by 1-2 % at a million sets and 8-15 % at one or two
                // waves per SIMD)
                if (!careful) {
                    if (tiles.pieces > 1) {
                        gr4j_opt_kernel<UH, Q.value, S.value, E.value, true>
                            <<<dim3((unsigned)(tiles.pieces * waves)), block,
                               0, st>>>(
                                days, T, s_init, r_init, params, N, d_plan,
                                force_lds, qsim, s_store, r_store, ld, qobs,
                                sse, tiles);
                        return;
                    }
                    gr4j_opt_kernel<UH, Q.value, S.value, E.value>
                        <<<grid, block, 0, st>>>(
                            days, T, s_init, r_init, params, N, d_plan,
                            force_lds, qsim, s_store, r_store, ld, qobs, sse,
                            tiles);
                    return;
                }
            }
            gr4j_kernel<UH, Q.value, S.value, E.value>
                <<<grid, block, lds, st>>>(
                    days, T, s_init, r_init, params, N, d_plan, force_lds,
                    qsim, s_store, r_store, ld, qobs, sse, uh_mem);
        });
    });
    // ... and behind them the sets that are not civil (gr4j_reference.h)
    gr4j_reference_kernel<<<grid, block, 0, st>>>(
        prec, etp, T, s_init, r_init, params, N, d_plan, qsim, s_store,
        r_store, ld, e ? qobs : nullptr, e ? sse : nullptr);
    RR_HIP(hipGetLastError());
    return RR_OK;
}
