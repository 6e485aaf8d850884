// abc.hip -- ABC-model ensemble kernel for gfx950.
//
// Replaces run_abcmodel (reference: rrmpg/models/abcmodel_model.py:15-60)
// and the per-set Python loop in ABCModel.simulate (reference:
// rrmpg/models/abcmodel.py:168-186).
//
// ABC is the one model of the family that is truly HBM-bound: 6 flops per
// 8 B (qsim) or 16 B (qsim + storage) written per model-timestep.  So the
// kernel is built around the store stream: each lane carries two adjacent
// parameter sets, so every output row is written with 16-byte
// global_store_dwordx4 (1 KiB contiguous per wave per day); the single
// storage state per set stays in a register; the shared precipitation is
// read through the scalar cache.
#include "common.h"

// Q/S/E: write qsim / write storage / accumulate the fused squared error.
// SPL = parameter sets per lane (2 is used): each lane owns SPL adjacent columns,
// so a wave writes SPL * 512 contiguous bytes per output row with 16-byte
// stores; more bytes in flight per wave is what a pure store stream needs.
template <int SPL, bool Q, bool S, bool E>
__global__ __launch_bounds__(RR_BLOCK) void abc_kernel_wide(
    const double *__restrict__ prec, int64_t T, double initial_state,
    const double *__restrict__ params, int64_t N, double *__restrict__ qsim,
    double *__restrict__ storage, int64_t ld, const double *__restrict__ qobs,
    double *__restrict__ sse)
{
    const int64_t i0 = SPL * ((int64_t)blockIdx.x * RR_BLOCK + threadIdx.x);
    bool act[SPL];
    double a[SPL], c[SPL], k[SPL], m[SPL], st[SPL], err[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        act[j] = i0 + j < N;
        const double *p = params + (act[j] ? i0 + j : N - 1) * 3;
        a[j] = p[0];
        c[j] = p[2];
        // loop invariants, evaluated exactly as the reference writes them
        k[j] = 1 - p[0] - p[1];                  // abcmodel_model.py:56
        m[j] = 1 - p[2];                         // :59
        st[j] = initial_state;
        err[j] = 0.0;
    }
    int64_t off = i0;

    // pairs (j, j+1) go out as one 16-byte store when both columns exist
    auto store = [&](double *base, const double (&v)[SPL]) {
#pragma unroll
        for (int j = 0; j < SPL; j += 2) {
            if (act[j + 1])
                rr_out2(base + off + j, v[j], v[j + 1]);
            else if (act[j])
                rr_out(&base[off + j], v[j]);
        }
    };

    // t = 0 (abcmodel_model.py:46-50)
    {
        double zero[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) zero[j] = 0.0;
        if (Q) store(qsim, zero);
        if (S) store(storage, st);
        if (E) {
            const double d = qobs[0] - 0.0;
#pragma unroll
            for (int j = 0; j < SPL; ++j) err[j] = d * d;
        }
    }
#pragma unroll 8
    for (int64_t t = 1; t < T; ++t) {
        const double pr = prec[t];   // wave-uniform -> scalar load
        off += ld;
        double q[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            q[j] = k[j] * pr + c[j] * st[j];     // :56
            st[j] = m[j] * st[j] + a[j] * pr;    // :59
        }
        if (Q) store(qsim, q);
        if (S) store(storage, st);
        if (E) {
            const double ob = qobs[t];
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const double d = ob - q[j];
                err[j] += d * d;
            }
        }
    }
    if (E) {
#pragma unroll
        for (int j = 0; j < SPL; ++j)
            if (act[j]) sse[i0 + j] = err[j];
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
        if (Q) rr_out(&qsim[off], 0.0);
        if (S) rr_out(&storage[off], s);
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
            if (Q) rr_out(&qsim[off], q);
            if (S) rr_out(&storage[off], s);
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
    if ((rc = rr_check_outputs("rr_abc_simulate_dev", qsim,
                               storage != nullptr)) != RR_OK)
        return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool q = qsim != nullptr, s = storage != nullptr, e = qobs && sse;
    const bool wide = (ld % 2 == 0) && (((uintptr_t)qsim) % 16 == 0) &&
                      (((uintptr_t)storage) % 16 == 0);
    // two sets per lane (measured: four sets per lane is slower, 16.8 vs
    // 13.9 ms at 1M sets -- fewer waves hide less store latency)
    const int spl = wide ? 2 : 1;
    const dim3 block(RR_BLOCK);
    const dim3 grid((unsigned)rr_ceil_div(rr_ceil_div(N, spl), RR_BLOCK));
    rr_dispatch3(q, s, e, [&](auto Q, auto S, auto E) {
        if (spl == 2)
            abc_kernel_wide<2, Q.value, S.value, E.value>
                <<<grid, block, 0, st>>>(prec, T, initial_state, params, N,
                                         qsim, storage, ld, qobs, sse);
        else
            abc_kernel_x1<Q.value, S.value, E.value>
                <<<grid, block, 0, st>>>(prec, T, initial_state, params, N,
                                         qsim, storage, ld, qobs, sse);
    });
    RR_HIP(hipGetLastError());
    return RR_OK;
}
