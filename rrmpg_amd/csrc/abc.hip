// abc.hip -- ABC-model ensemble kernel for gfx950.
//
// Replaces run_abcmodel (reference: rrmpg/models/abcmodel_model.py:15-60)
// and the per-set Python loop in ABCModel.simulate (reference:
// rrmpg/models/abcmodel.py:168-186).
//
// ABC is the one model of the family that is truly HBM-bound: 6 flops per
// 8 B (qsim) or 16 B (qsim + storage) written per model-timestep.  So the
// kernel is built around the store stream: each lane carries TWO adjacent
// parameter sets, so every output row is written with 16-byte
// global_store_dwordx4 (1 KiB contiguous per wave per day); the single
// storage state per set stays in a register; the shared precipitation is
// read through the scalar cache, eight days per s_load.
#include "common.h"

// Q/S/E: write qsim / write storage / accumulate the fused squared error.
template <bool Q, bool S, bool E>
__global__ __launch_bounds__(RR_BLOCK) void abc_kernel_x2(
    const double *__restrict__ prec, int64_t T, double initial_state,
    const double *__restrict__ params, int64_t N, double *__restrict__ qsim,
    double *__restrict__ storage, int64_t ld, const double *__restrict__ qobs,
    double *__restrict__ sse)
{
    const int64_t i0 = 2 * ((int64_t)blockIdx.x * RR_BLOCK + threadIdx.x);
    const bool act0 = i0 < N, act1 = i0 + 1 < N;
    const double *p0 = params + (act0 ? i0 : N - 1) * 3;
    const double *p1 = params + (act1 ? i0 + 1 : N - 1) * 3;
    const double a0 = p0[0], b0 = p0[1], c0 = p0[2];
    const double a1 = p1[0], b1 = p1[1], c1 = p1[2];
    // loop invariants, evaluated exactly as the reference writes them
    const double k0 = 1 - a0 - b0, k1 = 1 - a1 - b1;   // abcmodel_model.py:56
    const double m0 = 1 - c0, m1 = 1 - c1;             // :59

    double s0 = initial_state, s1 = initial_state;
    double e0 = 0.0, e1 = 0.0;
    int64_t off = i0;

    auto store = [&](double *base, double v0, double v1) {
        if (act1) {
            *reinterpret_cast<double2 *>(base + off) = make_double2(v0, v1);
        } else if (act0) {
            base[off] = v0;
        }
    };

    // t = 0 (abcmodel_model.py:46-50)
    if (Q) store(qsim, 0.0, 0.0);
    if (S) store(storage, s0, s1);
    if (E) {
        const double d = qobs[0] - 0.0;
        e0 = d * d;
        e1 = d * d;
    }
#pragma unroll 8
    for (int64_t t = 1; t < T; ++t) {
        const double pr = prec[t];   // wave-uniform -> scalar load
        off += ld;
        const double q0 = k0 * pr + c0 * s0;           // :56
        const double q1 = k1 * pr + c1 * s1;
        s0 = m0 * s0 + a0 * pr;                        // :59
        s1 = m1 * s1 + a1 * pr;
        if (Q) store(qsim, q0, q1);
        if (S) store(storage, s0, s1);
        if (E) {
            const double ob = qobs[t];
            const double d0 = ob - q0, d1 = ob - q1;
            e0 += d0 * d0;
            e1 += d1 * d1;
        }
    }
    if (E) {
        if (act0) sse[i0] = e0;
        if (act1) sse[i0 + 1] = e1;
    }
}

// Fallback when the 16-byte store shape does not apply (odd ld or a
// misaligned output pointer): one set per lane, 8-byte stores.
template <bool Q, bool S, bool E>
__global__ __launch_bounds__(RR_BLOCK) void abc_kernel_x1(
    const double *__restrict__ prec, int64_t T, double initial_state,
    const double *__restrict__ params, int64_t N, double *__restrict__ qsim,
    double *__restrict__ storage, int64_t ld, const double *__restrict__ qobs,
    double *__restrict__ sse)
{
    const int64_t i = (int64_t)blockIdx.x * RR_BLOCK + threadIdx.x;
    const bool active = i < N;
    const double *p = params + (active ? i : N - 1) * 3;
    const double a = p[0], b = p[1], c = p[2];
    const double k = 1 - a - b, m = 1 - c;
    double s = initial_state, e = 0.0;
    int64_t off = i;
    if (active) {
        if (Q) qsim[off] = 0.0;
        if (S) storage[off] = s;
    }
    if (E) {
        const double d = qobs[0] - 0.0;
        e = d * d;
    }
#pragma unroll 8
    for (int64_t t = 1; t < T; ++t) {
        const double pr = prec[t];
        off += ld;
        const double q = k * pr + c * s;
        s = m * s + a * pr;
        if (active) {
            if (Q) qsim[off] = q;
            if (S) storage[off] = s;
        }
        if (E) {
            const double d = qobs[t] - q;
            e += d * d;
        }
    }
    if (E && active) sse[i] = e;
}

extern "C" size_t rr_abc_workspace_bytes(int64_t T, int64_t N)
{
    (void)T; (void)N;
    return 256;   // none needed; a non-zero size keeps callers uniform
}

extern "C" int rr_abc_simulate_dev(const double *prec, int64_t T,
                                   double initial_state, const double *params,
                                   int64_t N, double *qsim, double *storage,
                                   int64_t ld, const double *qobs, double *sse,
                                   void *workspace, size_t workspace_bytes,
                                   void *stream)
{
    (void)workspace; (void)workspace_bytes;
    int rc = rr_check_common("rr_abc_simulate_dev", T, N, ld, params, qobs,
                             sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (!prec) {
        rr_set_error("rr_abc_simulate_dev: prec is NULL");
        return RR_E_NULL;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool q = qsim != nullptr, s = storage != nullptr, e = qobs && sse;
    const bool wide = (ld % 2 == 0) && (((uintptr_t)qsim) % 16 == 0) &&
                      (((uintptr_t)storage) % 16 == 0);
    const dim3 block(RR_BLOCK);
    const dim3 grid((unsigned)rr_ceil_div(wide ? rr_ceil_div(N, 2) : N,
                                          RR_BLOCK));
#define ABC_GO(K, Q, S, E)                                                   \
    hipLaunchKernelGGL((K<Q, S, E>), grid, block, 0, st, prec, T,            \
                       initial_state, params, N, qsim, storage, ld, qobs, sse)
#define ABC_DISPATCH(K)                                                      \
    do {                                                                     \
        if (q) {                                                             \
            if (s) { if (e) ABC_GO(K, true, true, true); else ABC_GO(K, true, true, false); } \
            else   { if (e) ABC_GO(K, true, false, true); else ABC_GO(K, true, false, false); } \
        } else {                                                             \
            if (s) { if (e) ABC_GO(K, false, true, true); else ABC_GO(K, false, true, false); } \
            else   { if (e) ABC_GO(K, false, false, true); else ABC_GO(K, false, false, false); } \
        }                                                                    \
    } while (0)
    if (wide) ABC_DISPATCH(abc_kernel_x2);
    else ABC_DISPATCH(abc_kernel_x1);
#undef ABC_DISPATCH
#undef ABC_GO
    RR_HIP(hipGetLastError());
    return RR_OK;
}
