// gr4j_core.h -- per-lane GR4J state machine shared by the GR4J kernel and
// the fused Cemaneige->GR4J kernel.
//
// Restates run_gr4j / _s_curve1 / _s_curve2 (reference:
// rrmpg/models/gr4j_model.py:15-192) for one parameter set per lane.
// Production store S and routing store R live in registers.  The two unit
// hydrographs have a DATA-DEPENDENT length per lane (ceil(x4) and
// ceil(2*x4+1) ordinates, gr4j_model.py:68-69), handled by two storage tiers
// with identical arithmetic:
//
//   UhRegs<M>  x4 <= M for every set of the launch, M = 3 (the reference's
//              default GR4J bounds, rrmpg/models/gr4j.py:51-54), 5 or 10 (the
//              hysteresis models' bounds): M + (2M+1) ordinates and as many
//              convolution slots in registers, loops fully unrolled.
//   UhLds      any x4 <= RR_GR4J_MAX_X4: ordinates and slots are staged in
//              LDS as [slot][lane] (8-byte elements, lane-contiguous ->
//              ds_read_b64 / ds_write_b64 are bank-conflict free) and the
//              slot loop runs to the wave-uniform maximum length.
//   UhMem      any x4 at all: the same [slot][lane] slab in HBM, behind the
//              launch's workspace (rr_*_workspace_bytes_x4).
//
// In both, a slot j of a lane with n ordinates is updated as the reference
// does (gr4j_model.py:130-136):  uh[j] = uh[j+1] + ord[j]*p  for j < n-1,
// uh[n-1] = ord[n-1]*p; slots >= n are never read.
#pragma once

#include "common.h"
#include "fastmath.h"

// Tolerance-driven arithmetic of the GR4J day: a
// product that goes straight into a sum is contracted into it (one rounding
// instead of two), the unit hydrographs' ordinates carry the 0.9 / 0.1 split
// of the routed amount (gr4j_model.py:126-127), and the two store updates
// x - x (1 - y) are taken as x y.  (The reference's own sequence:
// gr4j_reference.h, the kernel of the sets that are not civil.)
#define GR4J_UH1_SHARE 0.9
#define GR4J_UH2_SHARE 0.1

struct Gr4jPar {
    double x1, x2, x3, x4;
    InvDivisor inv_x1, inv_x3;   // x1, x3 divide five quantities every day
    lanemask_t x1_m, x3_m;       // lanes whose x1 / x3 suit the 3-FMA quotient
    __device__ __forceinline__ void set(double a, double b, double c, double d)
    {
        x1 = a; x2 = b; x3 = c; x4 = d;
        inv_x1 = make_inv_divisor(a);
        inv_x3 = make_inv_divisor(c);
        x1_m = RR_LANES(inv_x1.ok);
        x3_m = RR_LANES(inv_x3.ok);
    }
};

// Numerators of the 3-FMA quotients of this model stay below 2^196 (~1e59)
// on the fast form: no store or flux of a sane run comes near, and with it
// the folded update of gr4j_step_net cannot overflow early.
#define GR4J_NUM_HI 0x1p196

// Lanes whose a is a numerator the 3-FMA quotient certainly serves: +0 or
// 2^-900 <= a < 2^196.  Stores and fluxes are never negative in a sane run,
// so the magnitude test is ONE unsigned range check on the high word (two
// 32-bit instructions at 2 cycles each instead of two fp64 compares at 4;
// profiles/ubench): everything else -- negatives, NaN, inf, subnormals --
// falls outside it and sends the wave through the IEEE division, where every
// lane is re-examined with the full test (div_by_invariant_m).  This strict
// form guards the quotients of the snow routine, whose results are
// bit-identical to the reference's.
__device__ __forceinline__ lanemask_t gr4j_num_mask(double a)
{
    const unsigned hi = (unsigned)__double2hiint(a);
    return RR_LANES((hi - 0x07B00000u) < (0x4C300000u - 0x07B00000u)) |
           lanes_plus_zero(a);
}

// GR4J's own quotients (s/x1, the percolation's and the routing store's
// arguments, net/x1) are FAITHFUL: a * RN(1/x), one instruction, within 1.5
// ulp of a/x for every numerator (invdiv.h inv_mul_core) -- the GR4J family
// is a few-ulp restatement of the reference's libm calls to begin with
// (tolerance 1e-10 relative), its measured deviation
// from the reference semantics did not move (1.7e-13 over 30 years), and the two FMAs and
// the numerator vote of the correctly rounded form were a tenth of the day's
// vector work (GR4J 1M sets 53.6 -> 48.1 ms, scores 44.6 -> 40.4, 125k 6.7 ->
// 6.1; fused 90.2 -> 84.7).
// The only vote left is on the divisor, a loop invariant.  One numerator
// test survives, on the production store (gr4j_production: the folded store
// update must not overflow early), and it is the relaxed one: a is +0 or
// positive and below 2^196 -- ONE unsigned compare of the high word (sign
// bit set, NaN, inf and anything >= 2^196 are above the bound).  The
// per-lane choice in the slow path uses the same test, so a set's result
// never depends on its wave neighbours.
#define GR4J_NUM_HI_WORD 0x4C300000u        // high word of 2^196
__device__ __forceinline__ bool gr4j_num_ok(double a)
{
    return (unsigned)__double2hiint(a) < GR4J_NUM_HI_WORD;
}
__device__ __forceinline__ lanemask_t gr4j_num_lanes(double a)
{
    return RR_LANES((unsigned)__double2hiint(a) < GR4J_NUM_HI_WORD);
}

// a / x for the per-lane invariant x; stores run dry, so exact zeros stay on
// the fast form
template <class V = CarefulVotes>
__device__ __forceinline__ double gr4j_div_m(double a, lanemask_t a_ok,
                                             const InvDivisor &d,
                                             lanemask_t d_ok, V &&votes = V())
{
    (void)a_ok;
    return mul_by_inverse_m(a, d, d_ok, votes);
}
template <class V = CarefulVotes>
__device__ __forceinline__ double gr4j_div(double a, const InvDivisor &d,
                                           lanemask_t d_ok, V &&votes = V())
{
    return mul_by_inverse_m(a, d, d_ok, votes);
}

// _s_curve1 (gr4j_model.py:159-173); t is the integer ordinate index
__device__ __forceinline__ double gr4j_s_curve1(int t, double x4)
{
    const double tf = (double)t;
    if (t <= 0) return 0.0;
    else if (tf < x4) return pow(tf / x4, 2.5);
    else return 1.0;
}

// _s_curve2 (gr4j_model.py:176-192)
__device__ __forceinline__ double gr4j_s_curve2(int t, double x4)
{
    const double tf = (double)t;
    if (t <= 0) return 0.0;
    else if (tf <= x4) return 0.5 * pow(tf / x4, 2.5);
    else if (tf < 2 * x4) return 1 - 0.5 * pow(2 - tf / x4, 2.5);
    else return 1.0;
}

// num_uh1 = ceil(x4), num_uh2 = ceil(2*x4+1) (gr4j_model.py:68-69), clamped
// into int range; NaN gives 0 (invalid, rejected by the host scan).
__device__ __forceinline__ int gr4j_num_uh1(double x4)
{
    const double c = ceil(x4);
    return (c >= 1.0) ? ((c > 1e6) ? 1000000 : (int)c) : 0;
}
__device__ __forceinline__ int gr4j_num_uh2(double x4)
{
    const double c = ceil(2 * x4 + 1);
    return (c >= 1.0) ? ((c > 2e6) ? 2000000 : (int)c) : 0;
}

// a * b + c as a three-address v_fma_f64 (see UhRegs::route)
__device__ __forceinline__ double uh_fma(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// ---- register tier ---------------------------------------------------------
template <int N1MAX>
struct UhRegs {
    static constexpr int TIER = N1MAX;
    static constexpr int N2MAX = 2 * N1MAX + 1;
    // the convolution slots; a struct of their own so that a kernel can keep
    // TWO generations of them (the optimistic time loops: a day reads one and
    // writes the other, and the day's start state survives until the day is
    // known to be good -- OptimisticVotes, common.h)
    struct Slots {
        double u1[N1MAX], u2[N2MAX];
    };
    Slots z;                       // the in-place kernels' one generation
    double o1[N1MAX], o2[N2MAX];
    int n1, n2;

    __device__ __forceinline__ void init(double x4) { init(x4, z); }
    __device__ __forceinline__ void init(double x4, Slots &u)
    {
        n1 = gr4j_num_uh1(x4);
        n2 = gr4j_num_uh2(x4);
        // ordinate j = S(j+1) - S(j) (gr4j_model.py:75-79); S(j) is carried
        // over from the previous ordinate instead of being re-evaluated.
        // Invariant kept for the whole run: ordinates and slots at j >= n
        // are exactly 0 (route() relies on it).
        double prev = 0.0;
#pragma unroll
        for (int j = 0; j < N1MAX; ++j) {
            const double cur = gr4j_s_curve1(j + 1, x4);
            o1[j] = (j < n1) ? GR4J_UH1_SHARE * (cur - prev) : 0.0;
            prev = cur;
            u.u1[j] = 0.0;
        }
        prev = 0.0;
#pragma unroll
        for (int j = 0; j < N2MAX; ++j) {
            const double cur = gr4j_s_curve2(j + 1, x4);
            o2[j] = (j < n2) ? GR4J_UH2_SHARE * (cur - prev) : 0.0;
            prev = cur;
            u.u2[j] = 0.0;
        }
    }

    // shift-and-add both hydrographs; returns uh1[0], uh2[0].
    //
    // Reference (gr4j_model.py:130-136), a lane with n ordinates:
    //     uh[j] = uh[j+1] + ord[j]*p  (j < n-1),   uh[n-1] = ord[n-1]*p.
    // With the zero invariant above, the single formula
    //     uh[j] = uh[j+1] + ord[j]*p   for every j < MAX  (uh[MAX] := 0)
    // gives the same values: slot n-1 becomes 0 + ord*p and, for finite p,
    // the padding stays 0 + 0*p = 0 (only the sign of an exact zero can
    // differ).  Each slot is ONE fused multiply-add (the product is not
    // rounded separately as numba's fmul/fadd pair would; a <= 1/2-ulp
    // difference per slot that belongs to the few-ulp budget of the GR4J
    // family, see "transcendental calls" below) instead of 5 instructions
    // with per-lane selects.  A non-finite p (never in a sane run) leaves the real slots
    // right as well -- they only read padding that was still zero -- but
    // writes 0*NaN into the padding, so on such days the wave re-zeroes it.
    // `in` -> `out`: two generations, or the same object (slot j is written
    // after slot j + 1 was read and is not read again).
    template <class V = CarefulVotes>
    __device__ __forceinline__ void route(const Slots &in, Slots &out,
                                          double p1, double p2, double &head1,
                                          double &head2, V &&votes = V())
    {
        // (uh_fma: a three-address v_fma_f64 written out, because hipcc
        // turns __builtin_fma into the two-address v_fmac on the NEXT slot's
        // register and then shifts the whole array back with one v_mov_b64
        // per slot at the end of every day)
#pragma unroll
        for (int j = 0; j < N1MAX; ++j)
            out.u1[j] = (j + 1 < N1MAX)
                ? uh_fma(o1[j], p1, in.u1[(j + 1 < N1MAX) ? j + 1 : j])
                : o1[j] * p1;
#pragma unroll
        for (int j = 0; j < N2MAX; ++j)
            out.u2[j] = (j + 1 < N2MAX)
                ? uh_fma(o2[j], p2, in.u2[(j + 1 < N2MAX) ? j + 1 : j])
                : o2[j] * p2;
        // (p1 = 0.9 p and p2 = 0.1 p of the same p: one is finite iff the
        // other is)
        const lanemask_t finite = lanes_finite(p1);
        if (RR_VOTE(votes, finite)) {
            // (lengths made opaque: hipcc otherwise hoists the 3 * N1MAX + 1
            // slot masks `j < n` of this never-taken path out of the time
            // loop and parks them in that many SGPR pairs)
            int m1 = n1, m2 = n2;
            asm volatile("" : "+v"(m1), "+v"(m2));
#pragma unroll
            for (int j = 0; j < N1MAX; ++j)
                out.u1[j] = (j < m1) ? out.u1[j] : 0.0;
#pragma unroll
            for (int j = 0; j < N2MAX; ++j)
                out.u2[j] = (j < m2) ? out.u2[j] : 0.0;
        }
        head1 = out.u1[0];
        head2 = out.u2[0];
    }
    __device__ __forceinline__ void route(double p1, double p2, double &head1,
                                          double &head2)
    {
        route(z, z, p1, p2, head1, head2);
    }
};

// ---- indexed tiers: LDS (x4 <= 20) and global memory (any x4) -----------------
// Layout inside the wave's slab (doubles, RR_BLOCK lanes wide) -- the
// workgroup's dynamic LDS, or for UhMem the wave's part of the
// unit-hydrograph scratch behind the workspace (HBM; every access a
// coalesced 512-byte row per wave):
//   [0,         n1cap)           uh1 slots
//   [n1cap,     n1cap+n2cap)     uh2 slots
//   then the same again for the ordinates.
// UhMem is what lets ANY x4 run, as in the reference, which simply builds
// ceil(x4) and ceil(2 x4 + 1) ordinates (gr4j_model.py:68-79): slowly (three
// memory accesses per slot and day), but with the same arithmetic and hence
// the same bits as every other tier.
#define GR4J_TIER_LDS 0
#define GR4J_TIER_MEM 1000
template <bool IN_LDS>
struct UhIndexed {
    static constexpr int TIER = IN_LDS ? GR4J_TIER_LDS : GR4J_TIER_MEM;
    struct Slots {};     // (in place only: no second generation)
    double *base;        // this lane's column: base[slot * RR_BLOCK]
    int n1cap, n2cap;    // launch-wide capacities (host scan of max x4)
    int n1, n2;          // this lane's lengths
    int n1w, n2w;        // wave-uniform loop bounds

    __device__ __forceinline__ double &U1(int j) { return base[j * RR_BLOCK]; }
    __device__ __forceinline__ double &U2(int j) {
        return base[(n1cap + j) * RR_BLOCK];
    }
    __device__ __forceinline__ double &O1(int j) {
        return base[(n1cap + n2cap + j) * RR_BLOCK];
    }
    __device__ __forceinline__ double &O2(int j) {
        return base[(2 * n1cap + n2cap + j) * RR_BLOCK];
    }

    static __device__ __forceinline__ int wave_max(int v)
    {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const int o = __shfl_xor(v, m, 64);
            v = o > v ? o : v;
        }
        return v;
    }

    __device__ __forceinline__ void init(double *lds, int n1cap_, int n2cap_,
                                         double x4)
    {
        base = lds + threadIdx.x;
        n1cap = n1cap_;
        n2cap = n2cap_;
        n1 = gr4j_num_uh1(x4);
        n2 = gr4j_num_uh2(x4);
        n1 = n1 > n1cap ? n1cap : n1;     // cannot happen after the host scan
        n2 = n2 > n2cap ? n2cap : n2;
        n1w = __builtin_amdgcn_readfirstlane(wave_max(n1));
        n2w = __builtin_amdgcn_readfirstlane(wave_max(n2));
        double prev = 0.0;
        for (int j = 0; j < n1w; ++j) {
            const double cur = gr4j_s_curve1(j + 1, x4);
            O1(j) = GR4J_UH1_SHARE * (cur - prev);
            prev = cur;
            U1(j) = 0.0;
        }
        prev = 0.0;
        for (int j = 0; j < n2w; ++j) {
            const double cur = gr4j_s_curve2(j + 1, x4);
            O2(j) = GR4J_UH2_SHARE * (cur - prev);
            prev = cur;
            U2(j) = 0.0;
        }
    }

    // Same select-free update as UhRegs::route: ordinates and slots beyond a
    // lane's own length are exactly 0 (the s-curves give S(j+1) - S(j) = 1 - 1
    // there), so uh[j] = uh[j+1] + ord[j]*p for every slot up to the wave's
    // longest hydrograph; a non-finite p makes the wave re-zero the padding.
    // Each slot is read once (as the "next" of its left neighbour) and
    // written once per day.
    __device__ __forceinline__ void route(double p1, double p2, double &head1,
                                          double &head2)
    {
        for (int j = 0; j < n1w; ++j) {
            const double nxt = (j + 1 < n1w) ? U1(j + 1) : 0.0;
            const double nv = __builtin_fma(O1(j), p1, nxt);
            U1(j) = nv;
            if (j == 0) head1 = nv;
        }
        for (int j = 0; j < n2w; ++j) {
            const double nxt = (j + 1 < n2w) ? U2(j + 1) : 0.0;
            const double nv = __builtin_fma(O2(j), p2, nxt);
            U2(j) = nv;
            if (j == 0) head2 = nv;
        }
        // (p1 = 0.9 p and p2 = 0.1 p of the same p: one is finite iff the
        // other is)
        const lanemask_t finite = lanes_finite(p1);
        if (RR_ANY_OUTSIDE(finite)) {
            for (int j = 0; j < n1w; ++j)
                if (j >= n1) U1(j) = 0.0;
            for (int j = 0; j < n2w; ++j)
                if (j >= n2) U2(j) = 0.0;
        }
    }
};
typedef UhIndexed<true> UhLds;
typedef UhIndexed<false> UhMem;
template <class UH>
constexpr bool uh_is_indexed =
    std::is_same<UH, UhLds>::value || std::is_same<UH, UhMem>::value;

// doubles of one wave's slab for hydrographs of up to n1cap / n2cap ordinates
static __host__ __device__ __forceinline__ size_t gr4j_slab_doubles(int n1cap,
                                                                    int n2cap)
{
    return (size_t)2 * ((size_t)n1cap + (size_t)n2cap) * RR_BLOCK;
}
// bytes of unit-hydrograph scratch N sets need when the longest uh1 has n1
// ordinates (uh2: 2 n1 + 1), and the longest n1 a scratch of `bytes` serves
static inline size_t gr4j_mem_bytes(int64_t N, int64_t n1)
{
    if (N < 1 || n1 < 1) return 0;
    return (size_t)rr_ceil_div(N, RR_BLOCK) *
           gr4j_slab_doubles((int)n1, (int)(2 * n1 + 1)) * sizeof(double);
}
static inline int gr4j_mem_cap(size_t bytes, int64_t N)
{
    if (N < 1) return 0;
    const size_t per_wave = bytes / (size_t)rr_ceil_div(N, RR_BLOCK);
    // per wave: 2 * (n1 + 2 n1 + 1) * 64 * 8 = (6 n1 + 2) * 512 bytes
    const size_t units = per_wave / 512;
    if (units < 8) return 0;
    const size_t n1 = (units - 2) / 6;
    return n1 > 1000000 ? 1000000 : (int)n1;
}


// ---- which tier runs: decided ON THE DEVICE ----------------------------------
// The tier depends on the largest ceil(x4) of the launch, which only the GPU
// knows when the parameter block is resident in HBM.  Reading it back would
// cost a stream synchronisation per call (and make the *_simulate_dev entry
// points blocking), so the host never learns it: gr4j_scan_x4 leaves
// {max ceil(x4), number of sets without ordinates} in the first ints of the
// workspace, the host enqueues the kernel of EVERY tier, and each kernel
// decodes the plan in its first instructions and returns at once unless it
// is the selected one (three empty launches, ~10 us each at a million sets).
// A parameter block the kernels cannot run (ceil(x4) < 1, NaN, or
// x4 > RR_GR4J_MAX_X4) selects no tier: nothing is written, and
// rr_gr4j_plan_status() reports it.
// plan[0] = max ceil(x4), plan[1] = #bad sets, plan[2] = longest uh1 the
// unit-hydrograph scratch behind this launch's workspace can hold (0: none)
#define GR4J_PLAN_INTS 4
// LDS-tier launches are sized for the longest hydrographs the tier holds
#define GR4J_LDS_N1CAP ((int)RR_GR4J_MAX_X4)
#define GR4J_LDS_N2CAP (2 * GR4J_LDS_N1CAP + 1)
#define GR4J_LDS_BYTES \
    ((size_t)2 * (GR4J_LDS_N1CAP + GR4J_LDS_N2CAP) * RR_BLOCK * sizeof(double))

// tier for a plan: 3 / 5 / 10 = UhRegs<3/5/10>, GR4J_TIER_LDS, GR4J_TIER_MEM,
// -1 = none (error)
static __host__ __device__ __forceinline__ int gr4j_plan_tier(int max_n1,
                                                              int bad,
                                                              int force_lds,
                                                              int mem_cap)
{
    if (bad > 0 || max_n1 < 1) return -1;
    if (max_n1 > (int)RR_GR4J_MAX_X4)
        return max_n1 <= mem_cap ? GR4J_TIER_MEM : -1;
    if (force_lds || max_n1 > 10) return GR4J_TIER_LDS;
    return max_n1 <= 3 ? 3 : (max_n1 <= 5 ? 5 : 10);
}

// true if this kernel instantiation (UH) is the one the plan selects; the
// indexed tiers also get their capacities (the launch's longest hydrographs)
// REF_BEHIND: the launch enqueues a one-lane reference kernel behind this one
// (false: the kernels for more than RR_CEMANEIGE_MAX_LAYERS layers)
template <class UH, bool REF_BEHIND = true>
__device__ __forceinline__ bool gr4j_plan_selects(const int *__restrict__ plan,
                                                  int force_lds, int &n1cap,
                                                  int &n2cap)
{
    const int mx = plan[0], bad = plan[1];        // wave-uniform scalar loads
    const int mem_cap = plan[2];
    n1cap = mx;
    n2cap = 2 * mx + 1;
    const int tier = gr4j_plan_tier(mx, bad, force_lds, mem_cap);
    // plan[3] != 0: a forcing value that is not civil (gr4j_reference.h) --
    // the one-lane reference kernel behind the fast ones then computes EVERY
    // set with x4 <= RR_GR4J_MAX_X4 anyway, so the fast kernels of those
    // tiers do not run at all (they used to, and their output was then
    // overwritten).  Only a launch of the HBM-scratch tier keeps its fast
    // kernel: the reference kernel's private hydrographs end at x4 = 20.
    if (REF_BEHIND && plan[3] != 0 && tier != GR4J_TIER_MEM) return false;
    return tier == UH::TIER;
}

// ... refined per WAVE (round 5): every tier's kernel runs over the whole grid
// anyway, so a wave need not take the tier of the launch's largest x4 -- it
// takes the smallest register tier that holds ITS 64 sets' hydrographs, and
// returns from every other tier's kernel.  A set's bits do not depend on the
// tier that ran it (tests/test_gpu_parity.py
// test_a_sets_bits_do_not_depend_on_its_launch), so nothing changes but the
// time: under bounds that reach x4 = 10 (the hysteresis couplings' defaults)
// the sets of a wave are as random as the population and most waves still
// need the widest tier -- but a score-only sweep can take its sets in any
// order, and with the sets sorted by ceil(x4) (gr4j_tier_sort_async below)
// a fifth of the waves run in the 3-register tier and another quarter in the
// 5-register one.  Launches whose plan selects the HBM-scratch tier, no tier
// at all, or that are pinned to LDS stay whole (their slabs are sized by the
// launch's largest x4).  `x4`: the lane's own (tail lanes: the last set's).
template <class UH>
__device__ __forceinline__ bool gr4j_wave_selects(const int *__restrict__ plan,
                                                  int force_lds, double x4,
                                                  int &n1cap, int &n2cap)
{
    const int mx = plan[0], bad = plan[1];        // wave-uniform scalar loads
    const int mem_cap = plan[2];
    n1cap = mx;
    n2cap = 2 * mx + 1;
    const int launch = gr4j_plan_tier(mx, bad, force_lds, mem_cap);
    if (plan[3] != 0 && launch != GR4J_TIER_MEM) return false;  // (see above)
    if (launch < 0 || launch == GR4J_TIER_MEM || force_lds || launch == 3)
        return launch == UH::TIER;
    const int n1 = gr4j_num_uh1(x4);
    int wave = 3;
    if (RR_LANES(n1 > 3) != 0) wave = 5;
    if (RR_LANES(n1 > 5) != 0) wave = 10;
    if (RR_LANES(n1 > 10) != 0) wave = GR4J_TIER_LDS;
    return wave == UH::TIER;
}

// Hydrograph state of the lane, whichever tier: `lds` the workgroup's dynamic
// LDS, `uh_mem` the launch's unit-hydrograph scratch (single-wave workgroups:
// wave = blockIdx.x).
template <class UH>
__device__ __forceinline__ void gr4j_uh_init(UH &uh, double *lds,
                                             double *uh_mem, int n1cap,
                                             int n2cap, double x4)
{
    if constexpr (std::is_same<UH, UhLds>::value)
        uh.init(lds, n1cap, n2cap, x4);
    else if constexpr (std::is_same<UH, UhMem>::value)
        uh.init(uh_mem + (size_t)blockIdx.x * gr4j_slab_doubles(n1cap, n2cap),
                n1cap, n2cap, x4);
    else
        uh.init(x4);
}

// Calls f(UH{}) for every tier (the launch loop of the GR4J-family entries).
template <class F>
static inline void gr4j_for_each_tier(F &&f)
{
    f(UhRegs<3>{});
    f(UhRegs<5>{});
    f(UhRegs<10>{});
    f(UhLds{});
    f(UhMem{});
}

// ... and for the indexed tiers only, which between them run any x4 (the
// kernels for more than RR_CEMANEIGE_MAX_LAYERS layers, launched with
// force_lds = 1: a rare path that is built for simplicity, not for the
// register tiers' speed)
template <class F>
static inline void gr4j_for_each_indexed_tier(F &&f)
{
    f(UhLds{});
    f(UhMem{});
}

// The transcendental calls of the daily step (reference: 1 tanh + 3 pow) are
// evaluated with fastmath.h instead of OCML's general tanh / pow (165 / 224
// VALU instructions each):
//   np.tanh(.)            -> fast_tanh_parts, <= ~2.5 ulp, its quotient
//                            merged with the store update's (gr4j_step);
//   (1 + v**4)**(-0.25)   -> inv_fourth_root, ~1 ulp (Newton on y^-4 = b);
//   (r/x3)**3.5           -> x*x*x*sqrt(x), <= ~2.5 ulp; for finite x >= 0
//                            the root is fast_sqrt_core (8 instructions,
//                            <= 0.6 ulp; arguments below 2^-500 are raised
//                            to it -- their cube is 0 anyway), any other x
//                            takes the wave through the IEEE sqrt: 0 -> 0,
//                            inf -> inf, x < 0 -> NaN (as pow for a negative
//                            base and a non-integer exponent), NaN -> NaN.
// libm's own pow/tanh are faithful to ~1 ulp; these few-ulp differences are
// far inside the 1e-10 parity tolerance (observed ~1e-13 on 30-year runs).
// (1 + v**4)**(-0.25): b is a positive normal number (>= 1) unless v was
// non-finite or v**4 overflowed; one class test instead of the clamp of
// fastmath.h's inv_fourth_root (3 instructions).  +inf gives 0 (callers use
// 1 - result), NaN propagates.
// (GUARD_BY_VOTE = false, the branch-free clamped form, and FAST_ROOT = false
// below are measurement switches: until round 3 the UhRegs<10> kernels ran
// them -- 2-3 % faster at their two waves per SIMD -- and a set's last bits
// then depended on the tier its launch happened to run.  Every tier now runs
// the voted forms.)
// Percolation root as a polynomial (fastmath.h inv_fourth_root_1p_small):
// where its eight coefficients live.  Plain GR4J kernels with the hydrographs
// in 3+7 or 5+11 registers have VGPRs to spare (and no SGPRs: 104 of 106
// taken): VGPR pairs; everything else -- the LDS tier, held to 80 VGPRs, and
// the fused snow kernels, short of both -- fetches them from constant memory
// at the point of use.  A/B in profiles/README.md.
// Where a step's polynomial constants live (fastmath.h fast_tanh_parts):
// the plain GR4J kernels keep the tanh's in SGPRs; the fused snow kernels are
// short of SGPRs and fetch them from constant memory at the point of use;
// their small-sweep variants (at most two waves per SIMD) hold them in VGPRs.
enum { GR4J_CONSTS_SGPR = 0, GR4J_CONSTS_VGPR = 1, GR4J_CONSTS_JIT = 2,
       // ... and the exponential tanh instead of the rational one: the ice
       // kernels (snownext_kernels.h), which spill already and lose 6-9 % to the
       // approximant's live ranges (ice 86.4 -> 92.0 ms, hysteresis + ice
       // 165 -> 181) where every other kernel gains 3-11 %
       GR4J_CONSTS_JIT_EXP = 3 };

// (every tier: a set's bits must not depend on which tier its launch runs)
template <class UH>
constexpr bool RR_R4_POLY_ENABLED = true;
template <class UH, int CONSTS>
constexpr int gr4j_r4_consts()
{
    return (CONSTS == GR4J_CONSTS_VGPR ||
            (CONSTS == GR4J_CONSTS_SGPR &&
             (std::is_same<UH, UhRegs<3>>::value ||
              std::is_same<UH, UhRegs<5>>::value))) ? 1 : 2;
}

template <bool GUARD_BY_VOTE = true, class V = CarefulVotes>
__device__ __forceinline__ double gr4j_inv_fourth_root(double b,
                                                       V &&votes = V())
{
    if constexpr (!GUARD_BY_VOTE) return inv_fourth_root(b);
    double y = inv_fourth_root_core3(b);
    if (RR_VOTE(votes, lanes_of_class(b, 0x100))) {
        asm volatile("");                   // keep this a branch
        y = (b < __builtin_inf()) ? y : ((b != b) ? b : 0.0);
    }
    return y;
}

// FAST_ROOT = false keeps the compiler's IEEE sqrt inline (measurement switch,
// see above).
template <bool FAST_ROOT = true, class V = CarefulVotes>
__device__ __forceinline__ double pow_3_5(double x, V &&votes = V())
{
    if constexpr (!FAST_ROOT) return x * x * x * sqrt(x);
    // +0, positive subnormal or positive normal: one class test
    const lanemask_t ok = lanes_of_class(x, 0x1c0);
    // (v_max_f64 written out: from C++ hipcc quiets the operand first with a
    // v_max x, x of its own; a NaN x gives 2^-500 here, the slow path below
    // replaces the root for it anyway)
    double xr;
    asm("v_max_f64 %0, %1, %2" : "=v"(xr) : "v"(x), "s"(0x1p-500));
    double root = fast_sqrt_core(xr);
    if (RR_VOTE(votes, ok)) {
        // (the empty asm keeps this a branch: hipcc would otherwise evaluate
        // the IEEE sqrt for every wave and select)
        asm volatile("");
        const double exact = sqrt(x);
        root = (x >= 0.0 && x < __builtin_inf()) ? root : exact;
    }
    return x * x * x * root;
}

// The production store's gain (wet day, eq. 3, gr4j_model.py:95-96) or loss
// (dry day, eq. 4, :107-108) is c*th / (1 + k*th), th = tanh(net/x1), with
__device__ __forceinline__ void gr4j_store_coefficients(bool wet, double s,
                                                        double x1, double sx,
                                                        double &c, double &k)
{
    if (wet) {
        c = x1 * __builtin_fma(-sx, sx, 1.0);
        k = sx;
    } else {
        c = s * (2 - sx);
        k = 1 - sx;
    }
}

// ... evaluated the reference's way (IEEE quotient s/x1, two divisions) for
// the lanes gr4j_step_net's vote sends here (out-of-line variant, used by the
// UhRegs<10> kernels).
__device__ __attribute__((noinline)) double gr4j_store_change_reference(
    bool wet, double s, double x1, double E, double D)
{
    double c, k;
    gr4j_store_coefficients(wet, s, x1, s / x1, c, k);
    const double th = E / D;
    return c * th / (1 + k * th);
}

// One day of GR4J (gr4j_model.py:86-154).  s, r: production / routing store
// (in/out).  Returns the simulated discharge of the day.
// `wet` / `net`: net rainfall or net evapotranspiration (:89-111) -- which
// branch applies and the net amount.  Both branches of the reference share
// one shape: a tanh of the net amount over x1, one quotient; only the branch
// that applies is evaluated.  (The plain GR4J kernel gets wet/net from its
// pre-pass, wave-uniform; the coupled kernels compute them per lane.)
// `net_m`: lanes whose net is a valid numerator of the 3-FMA quotient
// (gr4j_num_mask; the plain GR4J kernel's pre-pass knows it per day).
// CONSTS: GR4J_CONSTS_* above.
// MID: called once in the middle of the day, after the last of the step's
// constant-table loads (tanh, percolation polynomial) -- where the fused snow
// kernels request the next day's forcing record (cemaneige.hip).
struct Gr4jNoHook {
    __device__ __forceinline__ void operator()() const {}
};
// The day is cut where its data flow is feed-forward: the PRODUCTION half
// (net rainfall, production store, percolation, :89-123) only ever hands the
// routed amount p_r to the ROUTING half (hydrographs, exchange, routing
// store, discharge, :126-154), which never feeds anything back.  (Rounds 3-5
// carried wave-specialised kernels that ran the halves in different waves of
// a workgroup; they lost and were removed in round 6.)  The kernels call the
// halves back to back through gr4j_step_net.
// `wet` as a Gr4jUniformWet (the plain GR4J kernel, whose pre-pass decides it
// once per day for every set -- the forcing is shared): the day record's 0 / 1
// in a scalar register.  The branch's two outcomes are then formed with that
// word's masks -- the sign of the store's change flipped by one v_xor_b32, the
// excess kept or zeroed by two v_and_b32 -- instead of three v_cndmask_b32 on
// a lane mask that is all ones or all zeros (VOP3: twice the issue time of
// the VOP2 forms; profiles/ubench/valu_cost.hip).  The same bits.
struct Gr4jUniformWet { int word; };
// gr4j_step (the coupled kernels' day) can look for such days itself.
// Measured and left off: hysteresis 147.1 -> 145.8 ms, ice melt 87.0 -> 88.7
// (profiles/r04_uniform_wet_ab.txt).
// ANY_FORCING: the kernel has no one-lane reference kernel behind it that
// would redo a launch with a non-finite forcing value (the kernels for more
// than RR_CEMANEIGE_MAX_LAYERS layers; the HBM-scratch tier, UhMem, always):
// a dry day's excess is then the reference's literal 0 (gr4j_model.py:104,
// :123) by per-lane select, not net - frac times a factor 0.0 -- which is NaN
// for etp = +inf, where the reference's p_r = perc is finite.
template <class UH, int CONSTS_ = GR4J_CONSTS_SGPR, bool ANY_FORCING_ = false,
          class MID = Gr4jNoHook, class V = CarefulVotes, class W = bool>
__device__ __forceinline__ double gr4j_production(const Gr4jPar &P, double &s,
                                                  double net, W wet_arg,
                                                  lanemask_t net_m,
                                                  MID &&mid = MID(),
                                                  V &&votes = V())
{
    constexpr bool uniform_wet = std::is_same<W, Gr4jUniformWet>::value;
    constexpr bool any_forcing =
        ANY_FORCING_ || std::is_same<UH, UhMem>::value;
    bool wet;
    int wet_word = 0;
    if constexpr (uniform_wet) {
        wet_word = wet_arg.word;
        wet = wet_word != 0;
    } else {
        wet = wet_arg;
    }
    constexpr bool rational_tanh =
        CONSTS_ != GR4J_CONSTS_JIT_EXP;
    constexpr int CONSTS =
        CONSTS_ == GR4J_CONSTS_JIT_EXP ? (int)GR4J_CONSTS_JIT : CONSTS_;
    // tanh(net/x1) = E / D (fastmath.h: E = expm1(2a)/2, D = E + 1); its
    // quotient is folded into the store update's own:
    //     c*th / (1 + k*th) == c*E / (D + k*E),
    // one division per day instead of two.
    // RR_GR4J_TANH_RATIONAL: E / D is the [9/8] Pade approximant of tanh
    // (fastmath.h fast_tanh_rational_parts: ten instructions instead of 28)
    // for |net/x1| <= 1 -- net rainfall below the store's capacity, every day
    // of a sane run --; a lane beyond that (or NaN) is voted out with the
    // rest below and takes the exponential form.
    const double a_th = gr4j_div_m(net, net_m, P.inv_x1, P.x1_m, votes);
    double E, D;
    lanemask_t a_small = ~0ull;
    if constexpr (rational_tanh) {
        fast_tanh_rational_parts<CONSTS>(a_th, E, D);
        a_small = RR_LANES(fabs(a_th) <= FP_TANHR_AMAX);
    } else {
        fast_tanh_parts<CONSTS>(a_th, E, D);
    }
    // One vote covers the quotient s/x1 and the folded form: with
    // 0 <= s < 2^196 and |x1| in [2^-100, 2^100] (invdiv.h) the quotient is
    // within 1.5 ulp of s/x1 and |s/x1| <= 2^296, so c*E and k*E (E <= 1.2e17,
    // |c| <= 2^692) stay finite
    // -- they are D times the reference's own c*th, k*th and would otherwise
    // overflow before those do.  The folded quotient itself is taken with
    // fastmath.h's 6-instruction division (the IEEE sequence is 11, and the
    // form is a few-ulp restatement already), which needs a denominator
    // away from zero: D + k*E >= 1 whenever 0 <= s <= x1, anything else is
    // voted out.  Any other lane sends the wave through the reference's own
    // sequence (IEEE quotient, two divisions).
    const double sx = inv_mul_core(s, P.inv_x1);
    double c, k, den;
    if constexpr (uniform_wet) {
        // (the denominator inside the day's own arm of the wave-uniform
        // branch: joined behind it, k = sx / k = 1 - sx costs the wet arm a
        // v_mov_b64 a day)
        if (wet) {
            c = P.x1 * __builtin_fma(-sx, sx, 1.0);
            den = FP_FMA_CV(sx, E, D);        // (three-address: no copy)
            asm("" : "+v"(den));    // (or hipcc joins the two again)
        } else {
            c = s * (2 - sx);
            den = FP_FMA_CV(1 - sx, E, D);
            asm("" : "+v"(den));
        }
    } else {
        gr4j_store_coefficients(wet, s, P.x1, sx, c, k);
        den = __builtin_fma(k, E, D);
    }
    (void)k;
    const lanemask_t fast = gr4j_num_lanes(s) & P.x1_m & a_small &
                            RR_LANES(fabs(den) >= 0x1p-100);
    double frac = fast_div_core(c * E, den);
    if (RR_VOTE(votes, fast)) {
        // (a lane keeps ITS tanh whatever sends the wave here: the
        // approximant inside its range, the exponential form outside)
        bool small = true;
        double Es = E, Ds = D;
        if constexpr (rational_tanh) {
            asm volatile("");                       // keep this a branch
            // (the argument taken again -- the same product -- rather than
            // kept alive across the day for a path that never runs)
            const double a_again =
                gr4j_div_m(net, net_m, P.inv_x1, P.x1_m);
            double E2, D2;
            fast_tanh_parts<CONSTS>(a_again, E2, D2);
            small = fabs(a_again) <= FP_TANHR_AMAX;
            Es = small ? E : E2;
            Ds = small ? D : D2;
        }
        double exact;
        if constexpr (std::is_same<UH, UhRegs<10>>::value) {
            // measured: out of line is 2 % faster in these kernels, 1-2 %
            // slower in the others
            exact = gr4j_store_change_reference(wet, s, P.x1, Es, Ds);
        } else {
            asm volatile("");                       // keep this a branch
            double ce, ke;
            gr4j_store_coefficients(wet, s, P.x1, s / P.x1, ce, ke);
            const double th = Es / Ds;
            exact = ce * th / (1 + ke * th);
        }
        const bool ok = gr4j_num_ok(s) && P.inv_x1.ok && small &&
                        fabs(den) >= 0x1p-100;
        frac = ok ? frac : exact;
    }
    // s - e_s + p_s (:114) and p_n - p_s (:123) with the branch's zeros
    // dropped (x - 0 and x + 0 are x)
    double sn, excess;
    // (uniform_wet: the day's sign and its "keep the excess" as wave-uniform
    // factors +-1.0 and 1.0 / 0.0 inside two FMAs -- fma(frac, +-1, s) is
    // s +- frac and fma(e, 1 | 0, perc) is perc + e | perc, each with the one
    // rounding of the sum it replaces, so the bits are the per-lane form's --
    // where round 4 flipped frac's sign bit and masked e with integer
    // instructions: 2 vector instructions instead of 6, GR4J 97.9 -> 94
    // per set-day.  e is a number: a launch with a forcing value that is not
    // civil is redone by the reference kernel behind this one -- where there
    // is none, any_forcing selects the excess per lane instead.)
    double keep_excess = 0.0;
    if constexpr (any_forcing) {
        const double sgn = __hiloint2double(
            wet ? 0x3ff00000 : (int)0xbff00000, 0);
        keep_excess = 1.0;
        sn = __builtin_fma(frac, sgn, s);
        excess = wet ? net - frac : 0.0;
    } else if constexpr (uniform_wet) {
        const int flip = (int)((unsigned)(wet_word ^ 1) << 31);
        const int keep = -wet_word;                 // all ones on a wet day
        const double sgn = __hiloint2double(0x3ff00000 | flip, 0);
        keep_excess = __hiloint2double(0x3ff00000 & keep, 0);
        sn = __builtin_fma(frac, sgn, s);
        excess = net - frac;
    } else {
        // (per lane: the same two factors, their high words ONE v_cndmask_b32
        // each -- where s + frac / s - frac and net - frac / 0 were formed
        // both and selected, two 64-bit selects of two instructions each)
        const double sgn = __hiloint2double(
            wet ? 0x3ff00000 : (int)0xbff00000, 0);
        keep_excess = __hiloint2double(wet ? 0x3ff00000 : 0, 0);
        sn = __builtin_fma(frac, sgn, s);
        excess = net - frac;
    }
    // percolation (:117); **4 is two squarings
    const double v = gr4j_div(4.0 / 9.0 * sn, P.inv_x1, P.x1_m, votes);
    const double v2 = v * v;
    constexpr bool by_vote = true;    // (the same forms in every tier)
    // v <= 4/9 while the store is within its capacity: the root of 1 + v**4
    // is then a degree-7 polynomial (fastmath.h); a wave with a lane beyond
    // that takes the general form for those lanes.  The polynomial's argument
    // is (1 + v**4) - 1, not v**4: the reference rounds 1 + v**4 before it
    // takes the power, and for a nearly empty store that rounding decides
    // between a percolation of exactly 0 and one of 1e-16 S -- nothing in a
    // sane run, but the difference between r == 0 and r > 0 for a routing
    // store with a negative x3, whose exchange term is 0 in one case and NaN
    // in the other (found by the fuzz soak).  The subtraction is exact.
    const double b1 = __builtin_fma(v2, v2, 1.0);
    double root;
    if constexpr (RR_R4_POLY_ENABLED<UH>) {
        const double u = b1 - 1;
        root = inv_fourth_root_1p_small<gr4j_r4_consts<UH, CONSTS>()>(u);
        const lanemask_t small = RR_LANES(u <= FP_R4_UMAX);
        if (RR_VOTE(votes, small)) {
            asm volatile("");                       // keep this a branch
            const double general = gr4j_inv_fourth_root<by_vote>(b1);
            root = (u <= FP_R4_UMAX) ? root : general;
        }
    } else {
        root = gr4j_inv_fourth_root<by_vote>(b1, votes);
    }
    // :117, :120: the store keeps sn * root and percolates the rest
    const double kept = sn * root;
    const double perc = sn - kept;
    mid();
    s = kept;
    if constexpr (any_forcing) return perc + excess;            // p_r, :123
    return __builtin_fma(excess, keep_excess, perc);
}

// `in` -> `out`: the hydrograph slots' two generations (UhRegs::Slots), or
// the same object.
template <class UH, class V = CarefulVotes>
__device__ __forceinline__ double gr4j_routing(const Gr4jPar &P, double &r,
                                               UH &uh,
                                               const typename UH::Slots &in,
                                               typename UH::Slots &out,
                                               double p_r, V &&votes = V())
{
    constexpr bool by_vote = true;    // (the same forms in every tier)
    const double p_r_uh1 = p_r, p_r_uh2 = p_r;  // (the ordinates carry the split)

    double head1, head2;
    if constexpr (uh_is_indexed<UH>)
        uh.route(p_r_uh1, p_r_uh2, head1, head2);               // :130-136
    else
        uh.route(in, out, p_r_uh1, p_r_uh2, head1, head2, votes);

    // (the exchange term x2 * (r/x3)**3.5 goes into the two sums that take
    // it as a fused multiply-add, and so do the squares under the roots:
    // one rounding where the reference has two)
    const double p35 =
        pow_3_5<by_vote>(gr4j_div(r, P.inv_x3, P.x3_m, votes), votes); // :139
    double rn = nb_max(0.0, __builtin_fma(P.x2, p35, r + head1)); // :142
    const double w = gr4j_div(rn, P.inv_x3, P.x3_m, votes);
    const double w2 = w * w;
    // :145, :148: the store keeps rn y, y = (1 + w**4)**-0.25, and gives
    // the rest
    const double kept = rn * gr4j_inv_fourth_root<by_vote>(
                                 __builtin_fma(w2, w2, 1.0), votes);
    const double q_r = rn - kept;
    rn = kept;
    const double q_d = nb_max(0.0, __builtin_fma(P.x2, p35, head2)); // :151
    r = rn;
    return q_r + q_d;                                           // :154
}
template <class UH>
__device__ __forceinline__ double gr4j_routing(const Gr4jPar &P, double &r,
                                               UH &uh, double p_r)
{
    if constexpr (uh_is_indexed<UH>) {
        typename UH::Slots none;
        return gr4j_routing<UH>(P, r, uh, none, none, p_r);
    } else {
        return gr4j_routing<UH>(P, r, uh, uh.z, uh.z, p_r);
    }
}

template <class UH, int CONSTS = GR4J_CONSTS_SGPR, bool ANY_FORCING = false,
          class MID = Gr4jNoHook>
__device__ __forceinline__ double gr4j_step_net(const Gr4jPar &P, double &s,
                                                double &r, UH &uh, double net,
                                                bool wet, lanemask_t net_m,
                                                MID &&mid = MID())
{
    const double p_r = gr4j_production<UH, CONSTS, ANY_FORCING>(
        P, s, net, wet, net_m, static_cast<MID &&>(mid));
    return gr4j_routing<UH>(P, r, uh, p_r);
}

template <class UH, int CONSTS = GR4J_CONSTS_SGPR, bool ANY_FORCING = false,
          class MID = Gr4jNoHook>
__device__ __forceinline__ double gr4j_step(const Gr4jPar &P, double &s,
                                            double &r, UH &uh, double prec,
                                            double etp, MID &&mid = MID())
{
    const bool wet = prec >= etp;                               // :89
    // prec - etp on a wet day, etp - prec on a dry one (:90, :102): the two
    // differences are each other's negatives exactly, so both are the
    // magnitude of one (a NaN stays a NaN) -- one subtraction where the
    // per-lane selection of two cost four instructions more
    const double net = __builtin_fabs(prec - etp);
    const lanemask_t net_m = gr4j_num_lanes(net);
    return gr4j_step_net<UH, CONSTS, ANY_FORCING>(
        P, s, r, uh, net, wet, net_m, static_cast<MID &&>(mid));
}
