// issue_mix.hip -- what a wave's scalar instructions, branches and waits cost
// next to its vector work when FEW waves share a SIMD.
//
// The ensemble kernels' days are ~75 % fp64 vector instructions, ~20 % scalar
// (lane-mask votes, loop and address arithmetic) and a few branches.  With
// five waves on a SIMD the scalar part is hidden behind other waves' vector
// instructions; a sweep of one GPU's shard of a strong-scaled job has two.
// One workgroup of 4 W waves on ONE CU (W per SIMD), every wave running the
// same loop, timed with s_memtime; printed: cycles per ITERATION per wave
// slot, i.e. (slowest wave's cycles) / (W * iterations).
//
//   hipcc --offload-arch=gfx950 -O2 -o issue_mix issue_mix.hip && ./issue_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),  \
              "+v"(a6), "+v"(a7) : "v"(b), "v"(c)                           \
            : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", \
              "s27", "s28", "s29"

#define V(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n"
#define D "v_fma_f64 %0, %0, %8, %9\n"          /* dependent chain */
#define S "s_and_b64 s[20:21], s[22:23], s[24:25]\n"
#define S2 "s_or_b64 s[26:27], s[22:23], s[24:25]\n"
#define NOP "s_nop 0\n"
#define BR(n) "s_cmp_eq_u64 s[28:29], 1\n s_cbranch_scc1 1f\n 1:\n"
#define VC "v_cmp_lt_f64 s[20:21], %1, %8\n"

#define KERNEL(NAME, BODY)                                                  \
    __global__ void NAME(unsigned long long *out, double b, double c,       \
                         int iters)                                         \
    {                                                                       \
        double a0 = 1.0 + 1e-3 * threadIdx.x, a1 = a0 + 1, a2 = a0 + 2,     \
               a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,          \
               a7 = a0 + 7;                                                 \
        asm volatile("s_mov_b64 s[22:23], -1\n s_mov_b64 s[24:25], 0\n"     \
                     "s_mov_b64 s[28:29], 0\n" ::: "s22", "s23", "s24",     \
                     "s25", "s28", "s29");                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();         \
        for (int it = 0; it < iters; ++it) asm volatile(BODY OPS);          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();         \
        const double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;             \
        if (s == 12345.678) out[4096] = 1;                                  \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;       \
    }

// 16 independent vector instructions
KERNEL(v16, V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7) V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7))
// ... + 8 scalar instructions, interleaved one after every second
KERNEL(v16_s8_mixed, V(0) V(1) S V(2) V(3) S2 V(4) V(5) S V(6) V(7) S2 V(0) V(1) S V(2) V(3) S2 V(4) V(5) S V(6) V(7) S2)
// ... the same eight in one group
KERNEL(v16_s8_grouped, V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7) V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7) S S2 S S2 S S2 S S2)
// ... + 16 scalar
KERNEL(v16_s16_mixed, V(0) S V(1) S2 V(2) S V(3) S2 V(4) S V(5) S2 V(6) S V(7) S2 V(0) S V(1) S2 V(2) S V(3) S2 V(4) S V(5) S2 V(6) S V(7) S2)
// 16 vector + 4 votes as the careful kernels spell them: compare into an
// SGPR pair, s_andn2 with exec, s_cmp, branch (not taken)
KERNEL(v16_vote4, V(0) V(1) V(2) VC "s_andn2_b64 s[26:27], exec, s[20:21]\n s_cmp_lg_u64 s[26:27], 0\n s_cbranch_scc0 1f\n 1:\n" V(3) V(4) V(5) V(6) VC "s_andn2_b64 s[26:27], exec, s[20:21]\n s_cmp_lg_u64 s[26:27], 0\n s_cbranch_scc0 2f\n 2:\n" V(7) V(0) V(1) V(2) VC "s_andn2_b64 s[26:27], exec, s[20:21]\n s_cmp_lg_u64 s[26:27], 0\n s_cbranch_scc0 3f\n 3:\n" V(3) V(4) V(5) V(6) VC "s_andn2_b64 s[26:27], exec, s[20:21]\n s_cmp_lg_u64 s[26:27], 0\n s_cbranch_scc0 4f\n 4:\n")
// ... and as the optimistic kernels do: compare, one s_and into the mask
KERNEL(v16_optvote4, V(0) V(1) V(2) VC "s_and_b64 s[22:23], s[22:23], s[20:21]\n" V(3) V(4) V(5) V(6) VC "s_and_b64 s[22:23], s[22:23], s[20:21]\n" V(7) V(0) V(1) V(2) VC "s_and_b64 s[22:23], s[22:23], s[20:21]\n" V(3) V(4) V(5) V(6) VC "s_and_b64 s[22:23], s[22:23], s[20:21]\n")
// 16 vector + 8 s_nop
KERNEL(v16_nop8, V(0) V(1) NOP V(2) V(3) NOP V(4) V(5) NOP V(6) V(7) NOP V(0) V(1) NOP V(2) V(3) NOP V(4) V(5) NOP V(6) V(7) NOP)
// 16 vector + 4 not-taken branches on a scalar condition
KERNEL(v16_br4, V(0) V(1) V(2) V(3) "s_cmp_eq_u64 s[28:29], 1\n s_cbranch_scc1 1f\n 1:\n" V(4) V(5) V(6) V(7) "s_cmp_eq_u64 s[28:29], 1\n s_cbranch_scc1 2f\n 2:\n" V(0) V(1) V(2) V(3) "s_cmp_eq_u64 s[28:29], 1\n s_cbranch_scc1 3f\n 3:\n" V(4) V(5) V(6) V(7) "s_cmp_eq_u64 s[28:29], 1\n s_cbranch_scc1 4f\n 4:\n")
// 16 vector + 4 TAKEN branches
KERNEL(v16_brtaken4, V(0) V(1) V(2) V(3) "s_cmp_eq_u64 s[28:29], 0\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n" V(4) V(5) V(6) V(7) "s_cmp_eq_u64 s[28:29], 0\n s_cbranch_scc1 2f\n s_nop 0\n 2:\n" V(0) V(1) V(2) V(3) "s_cmp_eq_u64 s[28:29], 0\n s_cbranch_scc1 3f\n s_nop 0\n 3:\n" V(4) V(5) V(6) V(7) "s_cmp_eq_u64 s[28:29], 0\n s_cbranch_scc1 4f\n s_nop 0\n 4:\n")
// the same mixes with ONE dependent vector chain (what a model's day is)
KERNEL(d16, D D D D D D D D D D D D D D D D)
KERNEL(d16_s8_mixed, D D S D D S2 D D S D D S2 D D S D D S2 D D S D D S2)
KERNEL(d16_s16_mixed, D S D S2 D S D S2 D S D S2 D S D S2 D S D S2 D S D S2 D S D S2 D S D S2)

typedef void (*kern_t)(unsigned long long *, double, double, int);
struct Entry { const char *name; kern_t k; int v, other; };
#define E(NAME, V_, O_) {#NAME, NAME, V_, O_}
static const Entry entries[] = {
    E(v16, 16, 0), E(v16_s8_mixed, 16, 8), E(v16_s8_grouped, 16, 8),
    E(v16_s16_mixed, 16, 16), E(v16_vote4, 20, 12), E(v16_optvote4, 20, 4),
    E(v16_nop8, 16, 8), E(v16_br4, 16, 8), E(v16_brtaken4, 16, 8),
    E(d16, 16, 0), E(d16_s8_mixed, 16, 8), E(d16_s16_mixed, 16, 16)};

static double run(kern_t k, int w, int iters, unsigned long long *d_out)
{
    const int nw = 4 * w;
    unsigned long long h[64];
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64 * nw), 0, 0, d_out, 1.0000001,
                           1e-9, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d_out, 8 * nw, hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int i = 0; i < nw; ++i) mx = h[i] > mx ? h[i] : mx;
        const double c = (double)mx / ((double)w * iters);
        best = c < best ? c : best;
    }
    return best;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    unsigned long long *d_out;
    (void)hipMalloc(&d_out, 8 * 8192);
    (void)hipMemset(d_out, 0, 8 * 8192);
    printf("# cycles per iteration per wave slot (slowest wave / (W * iters)); "
           "W waves per SIMD on one CU; vec / other = instructions per "
           "iteration\n");
    printf("%-18s %4s %5s %8s %8s %8s %8s %8s\n", "loop body", "vec", "other",
           "W=1", "W=2", "W=3", "W=4", "W=5");
    for (const Entry &e : entries) {
        printf("%-18s %4d %5d", e.name, e.v, e.other);
        for (int w = 1; w <= 5; ++w)
            printf(" %8.1f", run(e.k, w, iters, d_out));
        printf("\n");
    }
    (void)hipFree(d_out);
    return 0;
}
