// common.h -- shared helpers for the gfx950 kernels and the C-ABI layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>

#include "../../include/rrhip.h"

// One parameter set per lane.  64 = one wavefront per workgroup: the time
// loop needs no workgroup-level cooperation (the shared forcing arrives
// through the scalar cache), so single-wave workgroups give the
// dispatcher the finest granule to spread over 256 CUs x 4 SIMDs.
#define RR_BLOCK 64
static inline size_t rr_align256_(size_t x) { return (x + 255) & ~(size_t)255; }

// numba's max(a, b) / min(a, b): select(b > a, b, a) / select(b < a, b, a)
// (numba/cpython/builtins.py do_minmax) -- so max(0, NaN) == 0.
__device__ __forceinline__ double nb_max(double a, double b) {
    return (b > a) ? b : a;
}
__device__ __forceinline__ double nb_min(double a, double b) {
    return (b < a) ? b : a;
}
// The hardware's own minimum / maximum (IEEE minNum / maxNum: a NaN operand is
// dropped, -0 < +0), written out because from C++ hipcc first quiets the
// operands with a v_max x, x of its own.  They replace nb_min / nb_max only
// where a kernel has established that the operands cannot be NaN or -0.
__device__ __forceinline__ double rr_hw_min(double a, double b) {
    double d;
    asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ double rr_hw_max(double a, double b) {
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// ---- division by a per-lane loop invariant ---------------------------------
// a / b where b is fixed for the lane's whole time loop (FC, PWP, x1, x3 ...).
// hipcc expands every fp64 `/` into ~12 VALU instructions (v_div_scale x2,
// v_rcp_f64, 4 Newton FMAs, mul, fma, v_div_fmas, v_div_fixup), most of which
// only refine 1/b again.  With rb = RN(1/b) computed ONCE by a true division,
//     q0 = RN(a * rb);  r = RN(a - b * q0) (exact, FMA);  q = RN(q0 + r * rb)
// is the correctly rounded quotient RN(a / b) (Markstein's theorem: rb is
// within 1/2 ulp of 1/b and q0 within 1 ulp of a/b) as long as nothing
// overflows or underflows -- 3 instructions.  The guard keeps |b| in
// [2^-100, 2^100] and |a| in [2^-900, 2^900]; any other operand (zero, inf,
// NaN, subnormal, huge) takes the ordinary `/` for the wave, so the result is
// bit-identical to `a / b` for EVERY input (tests/test_fastmath_cpu.py runs
// 10^8 random and adversarial pairs on the host; the GPU self-test entry
// rrdbg_divide_by_invariant does the same on the device).
#include "invdiv.h"

// Wave votes as lane-mask arithmetic.  hipcc compiles __any()/__all() (and
// the ballot of any bool that is not directly a comparison) into a
// mask -> VGPR -> v_cmp round trip, 2 VALU instructions per vote.  The ballot
// of a comparison is just the SGPR pair that comparison wrote, so votes over
// combined conditions are built from per-comparison masks with scalar
// and/or/not, which cost no VALU issue slot at all.
typedef unsigned long long lanemask_t;
#define RR_LANES(cmp) ((lanemask_t)__builtin_amdgcn_ballot_w64(cmp))
__device__ __forceinline__ lanemask_t rr_exec() { return RR_LANES(true); }
// "some lane of the wave is outside mask m" for the votes whose answer is no
// in any sane run: marked unlikely so that hipcc lays the slow block out of
// line and the fast path falls through (a taken branch per vote per day
// otherwise: the wave's instruction buffer is refilled every time).
#define RR_ANY_OUTSIDE(m) __builtin_expect((rr_exec() & ~(m)) != 0, 0)

// What a vote does with its answer.  Every fast form in these kernels is
// guarded by "is some lane of the wave outside the form's domain?"
//   CarefulVotes     decide at once: the wave branches into the reference's
//                    own operation for those lanes (two scalar instructions
//                    and a branch per vote);
//   OptimisticVotes  note the lanes in a mask (one scalar AND) and go on with
//                    the fast form everywhere; the caller asks any() ONCE at
//                    the end of the day and, if so, redoes that day from its
//                    untouched start state with CarefulVotes.  No sane run
//                    ever redoes a day, and a sweep of a few waves per SIMD
//                    -- where every scalar instruction and branch costs the
//                    wave an issue turn that nobody else fills -- runs a
//                    branch-free day (gr4j.hip, cemaneige.hip).
// Either way a lane inside the domain gets the fast form's value and a lane
// outside it the reference operation's, so results are bit-identical.
// (the "unlikely" has to sit in the `if` itself -- RR_VOTE / RR_VOTES_FAILED --
// : hipcc consumes __builtin_expect before it inlines, so one inside these
// members would never reach the caller's branch and the slow blocks would
// be laid out in line again)
#define RR_VOTE(votes, m) __builtin_expect((votes).outside(m), 0)
#define RR_VOTES_FAILED(votes) __builtin_expect((votes).any(), 0)
struct CarefulVotes {
    static constexpr bool optimistic = false;
    __device__ __forceinline__ bool outside(lanemask_t m) const
    {
        return (rr_exec() & ~m) != 0;
    }
};
struct OptimisticVotes {
    static constexpr bool optimistic = true;
    lanemask_t good = ~0ull;
    __device__ __forceinline__ bool outside(lanemask_t m)
    {
        good &= m;
        return false;
    }
    __device__ __forceinline__ bool any() const
    {
        // (a wave-uniform value by construction -- every mask is a ballot --
        // but not always to the compiler's divergence analysis, which then
        // builds a per-lane branch around the redo block: said explicitly)
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)good);
        const unsigned hi =
            __builtin_amdgcn_readfirstlane((unsigned)(good >> 32));
        const lanemask_t g = ((lanemask_t)hi << 32) | lo;
        return (rr_exec() & ~g) != 0;
    }
};

// Class tests written directly into an SGPR pair with one v_cmp_class_f64
// (spelled as an integer or class test in C++ they go through the VGPR round
// trip above).  Class bits: 3 -normal, 4 -subnormal, 5 -0, 6 +0, 7 +subnormal,
// 8 +normal, 9 +inf.
__device__ __forceinline__ lanemask_t lanes_of_class(double a, int classes) {
    lanemask_t m;
    asm("v_cmp_class_f64 %0, %1, %2" : "=s"(m) : "v"(a), "s"(classes));
    return m;
}
__device__ __forceinline__ lanemask_t lanes_plus_zero(double a) {
    lanemask_t m;
    asm("v_cmp_class_f64 %0, %1, 0x40" : "=s"(m) : "v"(a));   // inline constant
    return m;
}
__device__ __forceinline__ lanemask_t lanes_finite(double a) {
    return lanes_of_class(a, 0x1f8);
}

// (A hardware finding the kernels are written around: a v_cndmask_b32 in its
// 4-byte VOP2 encoding, which hipcc prefers whenever the mask sits in VCC,
// costs 17 cycles of SIMD time per wave64 instruction on gfx950 unless VCC was
// written by the instruction right before it, against 4.3 for the VOP3 form
// with the mask in VCC or any SGPR pair -- profiles/ubench/valu_cost.hip, rows
// sel_vop2_vcc / sel_vop3_*.  Hence selects next to their compares, FMA
// factors and exec-masked moves instead of masks kept in VCC.)

// lanes whose a suits inv_div_core as a numerator: |a| in [2^-900, 2^900], or
// +0 (invdiv.h: inv_div_numerator_ok0).  A caller may tighten the upper
// bound (`hi` <= 2^900).
__device__ __forceinline__ lanemask_t inv_div_numerator_mask0(
    double a, double hi = 0x1p900) {
    return (RR_LANES(fabs(a) >= 0x1p-900) | lanes_plus_zero(a)) &
           RR_LANES(fabs(a) <= hi);
}

// a / d.b, bit-identical to `/` for EVERY input, with the vote done on lane
// masks: a_ok = inv_div_numerator_mask0(a) or any mask that implies it
// (shared by all quotients of one numerator), d_ok = RR_LANES(d.ok), hoisted
// out of the time loop by the caller.  If any active lane is outside the fast form's domain the whole
// wave evaluates the IEEE division and those lanes take it.
template <class V = CarefulVotes>
__device__ __forceinline__ double div_by_invariant_m(double a, lanemask_t a_ok,
                                                     const InvDivisor &d,
                                                     lanemask_t d_ok,
                                                     double hi = 0x1p900,
                                                     V &&votes = V()) {
    double q = inv_div_core(a, d);
    if (RR_VOTE(votes, a_ok & d_ok)) {
        const bool ok = inv_div_numerator_ok0(a) && fabs(a) <= hi && d.ok;
        const double exact = a / d.b;
        q = ok ? q : exact;        // ok lanes: both values are RN(a / b)
    }
    return q;
}

// a / d.b in its faithful form (invdiv.h inv_mul_core: a * RN(1 / b), within
// 1.5 ulp for every numerator): the only thing to vote on is the divisor, a
// loop invariant -- no compare per day.  Lanes whose divisor is outside
// [2^-100, 2^100] take the IEEE division.
template <class V = CarefulVotes>
__device__ __forceinline__ double mul_by_inverse_m(double a,
                                                   const InvDivisor &d,
                                                   lanemask_t d_ok,
                                                   V &&votes = V()) {
    double q = inv_mul_core(a, d);
    if (RR_VOTE(votes, d_ok)) {
        const double exact = a / d.b;
        q = d.ok ? q : exact;
    }
    return q;
}

// ---- one output row of a wave: 64 adjacent columns ---------------------------
// Every output is [T][ld] row-major and a wave owns 64 adjacent columns, so a
// day's store is (wave-uniform row base) + (lane * 8 bytes).  Issued as a raw
// buffer store -- descriptor built from the scalar row base, the lane offset
// the only vector operand -- the address arithmetic of the time loop is all
// scalar (a flat pointer per lane costs a 64-bit VALU add per day), and the
// descriptor's size drops the columns past N in hardware, so the tail wave
// needs no exec masking either.  `bytes` = 8 * min(64, N - first column).
//
// Every output element is written once and never read by the sweep: the
// stores are NON-TEMPORAL (streamed past the L2's write-back lines).  With
// plain stores the million-set sweeps that write qsim are a third slower
// (HBV-Edu 19.2 -> 29.5 ms, 125k sets 2.78 -> 4.3: profiles/
// r04_streaming_stores.txt).
// cache-policy bits of the row stores (gfx94x / gfx950: 1 = sc0, 2 = nt,
// 16 = sc1).  nt + sc1 measured against nt alone: HBV-Edu headline 19.55 ->
// 19.18 ms, 125k sets 2.86 -> 2.82; nt + sc0 no different from nt
// (profiles/r04_streaming_stores.txt).
#ifndef RR_OUT_AUX
#define RR_OUT_AUX 18
#endif
__device__ __forceinline__ void rr_out(double *p, double v)
{
    __builtin_nontemporal_store(v, p);
}
// two adjacent columns as one 16-byte store (p 16-byte aligned)
typedef double rr_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rr_out2(double *p, double a, double b)
{
    rr_v2d v;
    v.x = a;
    v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<rr_v2d *>(p));
}
typedef int rr_v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rr_store_row(double *row_base, unsigned bytes,
                                             int lane_byte_off, double v,
                                             bool nontemporal = true)
{
    // word 3: DATA_FORMAT = 32-bit, raw addressing (the value the compiler's
    // own buffer intrinsics use on gfx90a / gfx94x / gfx950)
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)row_base, (short)0, (int)bytes, 0x00020000);
    rr_v2i d;
    d.x = __double2loint(v);
    d.y = __double2hiint(v);
    if (nontemporal)
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, lane_byte_off, 0, RR_OUT_AUX);
    else
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, lane_byte_off, 0, 0);
}
// The same with a wave-uniform byte offset in the instruction's SOFFSET field:
// a time loop unrolled k-fold keeps ONE row base per trip and addresses its
// k rows through loop-invariant offsets ld * 8, 2 ld * 8 ... -- no scalar
// address arithmetic per day.  On gfx950 the range check of a raw buffer
// covers soffset + the lane's offset (measured: with num_records = `bytes`
// every store of a row behind the first was dropped), so the descriptor's
// size is soffset + bytes: lanes at or beyond `bytes` are still dropped.
// soffset + bytes < 2^32: the caller's business.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "rr_store_row_at relies on gfx950's raw-buffer range check (SOFFSET included): build with ARCH=gfx950"
#endif
__device__ __forceinline__ void rr_store_row_at(double *row_base, unsigned bytes,
                                                int lane_byte_off,
                                                unsigned soffset, double v,
                                                bool nontemporal = true)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)row_base, (short)0, (int)(soffset + bytes), 0x00020000);
    rr_v2i d;
    d.x = __double2loint(v);
    d.y = __double2hiint(v);
    if (nontemporal)
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, lane_byte_off,
                                              (int)soffset, RR_OUT_AUX);
    else
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, lane_byte_off,
                                              (int)soffset, 0);
}
__device__ __forceinline__ unsigned rr_row_bytes(int64_t first, int64_t N)
{
    const int64_t rem = N - first;
    return rem >= RR_BLOCK ? 8u * RR_BLOCK : (rem > 0 ? (unsigned)rem * 8u : 0u);
}

// ---- the day records prefetched into the XCD's L2 -----------------------------
// The time loops read their day records through the scalar cache, whose
// misses go to the L2 of the wave's XCD -- and the records have just been
// written by a pre-pass, on whatever XCD its blocks ran: the first touch of a
// 64-byte line in an XCD goes out to the Infinity Cache / HBM, which takes
// longer than the day or two the prefetching loops ask ahead.  A sweep of
// several waves per SIMD never notices (another wave issues); with ONE wave on
// a SIMD every wave of the XCD sits behind the one that leads (HBV-Edu,
// 65,536 sets: 2.24 -> 1.51 ms, profiles/r05_hbv_warm_ab.txt).  So the waves
// of an XCD (workgroups are dealt round-robin: XCD = linear id % 8) share out
// the lines among themselves when they start, one ordinary load per lane, at
// most `max_loads` per wave, interleaved so that a partial cover is an even
// one: a few microseconds once, and the time loop's scalar loads are L2 hits
// from then on.  A prefetch only: nothing depends on which XCD a wave really
// runs on.  Written as asm, load and wait in one statement, no memory
// clobber: as C++ -- or with the clobber -- hipcc no longer takes global
// memory for unwritten, and a kernel that reads its records through the
// global address space gets them by vector loads, in VGPRs.
__device__ __forceinline__ void rr_warm_l2(const void *base, int64_t nbytes,
                                           int max_loads = 4)
{
    const int64_t nlines = (nbytes + 63) / 64;
    const int64_t wg = blockIdx.x + (int64_t)blockIdx.y * gridDim.x;
    const int64_t R = ((int64_t)gridDim.x * gridDim.y + 7) / 8;
    const int64_t r = wg >> 3;
    for (int m = 0; m < max_loads; ++m) {
        const int64_t line = ((int64_t)m * RR_BLOCK + threadIdx.x) * R + r;
        if (line < nlines) {
            const char *ptr = (const char *)base + line * 64;
            double dummy;
            asm volatile("global_load_dwordx2 %0, %1, off\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(dummy) : "v"(ptr));
        }
    }
}

// ---- the time axis in pieces ------------------------------------------------
// A million-set sweep is 15,625 waves of equal duration on 1,024 SIMDs: 15.26
// per SIMD, so 265 SIMDs run a sixteenth wave while the others idle -- the
// kernel takes 16 wave slots for 15.26 slots of work -- and the dispatcher's
// wave placement is not even either.  The tiled kernels cut every wave's 30
// years into PIECES: an ITEM is piece p of job j (the 64 sets of a wave),
// numbered piece-major (item = p * jobs + j), and the model states travel from
// piece to piece through a small HBM scratch [nstate][jobs * 64], handed over
// with a release / acquire pair at agent scope (the next piece may run on
// another XCD) and a per-job flag.  Items are a quarter as long, slots refill
// as they free, and the SIMDs finish within a fraction of a wave of each
// other (HBV-Edu 1M sets: 27.1 -> 25.3 ms, scores 21.8 -> 19.6; GR4J scores
// 46.8 -> 44.4).
// Which item a wave works on is decided by a TICKET drawn from one atomic
// counter when the wave starts to run (rr_tile_ticket) -- never by its
// workgroup index.  An item waits only for the item `jobs` tickets before it,
// and a ticket exists only because a wave that is already running drew it:
// whatever order the dispatcher starts workgroups in, whatever else shares the
// GPU, the smallest unfinished ticket can always proceed, so the launch
// cannot deadlock (rounds 1-3 took the item from blockIdx and relied on every
// XCD dispatching workgroups in increasing order, with a poll limit that
// turned a violation into a failed launch).  HBV-Edu runs the same scheme
// with PERSISTENT waves (as many as are resident at once, each drawing
// tickets until none is left: hbvedu.hip); GR4J and Cemaneige launch one
// single-wave workgroup per item, each drawing exactly one ticket -- no loop
// around the kernel body, whose register allocation did not survive one.
// Results are bit-identical to the untiled loops: the same operations in the
// same order.
struct RrTiles {
    int *queue;        // [0]: ticket counter; [1 + job]: pieces of the job done
    double *state;     // hand-over scratch
    int pieces;        // 0 / 1: untiled
    int warm;          // != 0: the waves prefetch the day records into their
                       // XCD's L2 when they start (rr_warm_l2; rides along
                       // here because these kernels take the struct anyway)
};
// sweeps whose waves prefetch their day records (rr_warm_l2): one wave per
// SIMD at most, or nothing but scores written (measured for HBV-Edu,
// hbvedu.hip hbv_launch; the GR4J family: profiles/r05_warm_family_ab.txt)
int64_t rr_option(int option);
int rr_simd_count();
static inline int rr_warm_choice(int64_t waves, int64_t simds, bool score_only)
{
    const int64_t opt = rr_option(RR_OPT_WARM_RECORDS);
    if (opt >= 0) return opt != 0;
    return (waves <= simds || score_only) ? 1 : 0;
}
// days [b, e) of piece `piece` of the days [t0, t1); piece lengths are
// multiples of `even` (2 for the loops that run two days per trip)
__device__ __forceinline__ void rr_tile_range(int t0, int t1, int pieces,
                                              int piece, int even, int &b,
                                              int &e)
{
    int len = (t1 - t0 + pieces - 1) / pieces;
    len = (len + even - 1) / even * even;
    b = t0 + piece * len;
    if (b > t1) b = t1;
    e = (b + len < t1) ? b + len : t1;
}
// this wave's item: the next ticket of the launch (wave-uniform; queue[0],
// zeroed by the host before the launch)
__device__ __forceinline__ int rr_tile_ticket(const RrTiles &q)
{
    int item = 0;
    if ((threadIdx.x & (RR_BLOCK - 1)) == 0)
        item = __hip_atomic_fetch_add(q.queue, 1, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(item);
}
// wait until piece - 1 of this job has published its states (an item with a
// smaller ticket, held by a wave that is running or done: the wait ends)
__device__ __forceinline__ void rr_tile_wait(const RrTiles &q, int job,
                                             int piece)
{
    int *flag = q.queue + 1 + job;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT) < piece)
        __builtin_amdgcn_s_sleep(8);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// after the states of `piece` have been stored
__device__ __forceinline__ void rr_tile_publish(const RrTiles &q, int job,
                                                int piece)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if ((threadIdx.x & (RR_BLOCK - 1)) == 0)
        __hip_atomic_store(q.queue + 1 + job, piece + 1, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
static inline size_t rr_tile_queue_bytes(int64_t N)
{
    return rr_align256_((size_t)((N > 0 ? N : 1) / RR_BLOCK + 2) * sizeof(int));
}
static inline size_t rr_tile_bytes(int64_t N, int nstate)
{
    const size_t jobs = (size_t)((N > 0 ? N : 1) + RR_BLOCK - 1) / RR_BLOCK;
    return rr_tile_queue_bytes(N) +
           rr_align256_((size_t)nstate * jobs * RR_BLOCK * sizeof(double));
}

// ---- error plumbing (host) ------------------------------------------------
void rr_set_error(const char *fmt, ...);

// Current value of a measurement / test option (rrhip.h RR_OPT_*): a relaxed
// atomic load, nothing else -- this is what the launch paths consult.
int64_t rr_option(int option);

#define RR_HIP(call)                                                        \
    do {                                                                    \
        hipError_t _e = (call);                                             \
        if (_e != hipSuccess) {                                             \
            rr_set_error("%s failed: %s (%s:%d)", #call,                    \
                         hipGetErrorString(_e), __FILE__, __LINE__);        \
            return RR_E_HIP;                                                \
        }                                                                   \
    } while (0)

static inline int64_t rr_ceil_div(int64_t a, int64_t b) {
    return (a + b - 1) / b;
}
static inline size_t rr_align256(size_t x) { return (x + 255) & ~(size_t)255; }

// SIMDs of the current device (4 per compute unit of what the process sees --
// 1024 on a whole MI355X, fewer in a partitioned mode): the kernel-variant
// choices by sweep size count waves per SIMD with it.  Cached per device.
int rr_simd_count();

// Common argument checks of the *_simulate_dev entry points.
int rr_check_common(const char *who, int64_t T, int64_t N, int64_t ld,
                    const void *params, const void *qobs, const void *sse);

// Storage outputs are returned WITH the discharge, as the reference's
// return_storage does (hbvedu.py:191 and its siblings), never alone: the
// kernels are built for the five combinations of outputs that leaves.
int rr_check_outputs(const char *who, const void *qsim, bool any_storage);

// Turns the three output flags (discharge, storages, sums of squares) into
// compile-time template arguments: f(std::bool_constant<q>, <s>, <e>) for
// {q}, {q, s}, {q, e}, {q, s, e} and {e}.  Storages without the discharge are
// refused before (rr_check_outputs); with nothing to write there is nothing
// to launch.
template <class F>
static inline void rr_dispatch3(bool q, bool s, bool e, F &&f)
{
    const std::true_type Y{};
    const std::false_type N{};
    if (q) {
        if (s) { if (e) f(Y, Y, Y); else f(Y, Y, N); }
        else   { if (e) f(Y, N, Y); else f(Y, N, N); }
    } else if (!s && e) {
        f(N, N, Y);
    }
}
