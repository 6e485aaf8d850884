// scores.hip -- per-set skill scores from a discharge array resident in HBM.
//
// The Monte-Carlo driver of the reference scores every simulated column with
// a Python loop over calc_mse (reference: rrmpg/tools/monte_carlo.py:66-71),
// and its metrics module offers NSE, RMSE, KGE, alpha/beta and Pearson r
// (reference: rrmpg/utils/metrics.py:29-299), each re-validating and copying
// both series.  All of those scores are functions of four per-column sums
// (and of sums over the observations alone, taken on the host), so one pass
// over qsim[T][ld] produces them for every parameter set at once:
//     sums[i] = { sum q, sum q^2, sum q*obs, sum (obs - q)^2 }
// (rr_column_sums_shifted_dev: the first three about a shift c, i.e. of q - c
// and obs - c, which keeps variance / covariance free of cancellation)
// One lane per column, rows streamed top to bottom: each wave reads 512
// contiguous bytes per row -- a pure HBM-bandwidth kernel (8 B per
// model-timestep read, nothing written but 32 B per set).  Accumulation is
// in time order (numpy's pairwise sums differ by ~1e-16 relative).  As in the
// reference's metric functions, a NaN in either series makes that column's
// scores NaN: there is no pairwise-finite filtering (callers with gaps in the
// observations mask them before, as they would for rrmpg.utils.metrics).
#include "common.h"

__global__ __launch_bounds__(256) void column_sums_kernel(
    const double *__restrict__ qsim, int64_t ld, const double *__restrict__ obs,
    int64_t T, int64_t N, double shift, double *__restrict__ sums)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    double s_q = 0.0, s_qq = 0.0, s_qo = 0.0, s_dd = 0.0;
    const double *p = qsim + i;
#pragma unroll 8
    for (int64_t t = 0; t < T; ++t) {
        const double q = __builtin_nontemporal_load(p);
        const double o = obs[t];            // wave-uniform -> scalar load
        const double d = o - q;
        // (moments about `shift`, not about 0: with shift = mean(obs) the
        // variance and covariance the host derives from them do not cancel
        // for a series that is large and nearly constant)
        const double qc = q - shift, oc = o - shift;
        s_q += qc;
        s_qq += qc * qc;
        s_qo += qc * oc;
        s_dd += d * d;
        p += ld;
    }
    double *out = sums + i * 4;
    out[0] = s_q; out[1] = s_qq; out[2] = s_qo; out[3] = s_dd;
}

extern "C" int rr_column_sums_dev(const double *qsim, int64_t ld,
                                  const double *obs, int64_t T, int64_t N,
                                  double *sums, void *stream)
{
    return rr_column_sums_shifted_dev(qsim, ld, obs, T, N, 0.0, sums, stream);
}

extern "C" int rr_column_sums_shifted_dev(const double *qsim, int64_t ld,
                                          const double *obs, int64_t T,
                                          int64_t N, double shift,
                                          double *sums, void *stream)
{
    int rc = rr_check_common("rr_column_sums_dev", T, N, ld, qsim, obs, sums);
    if (rc != RR_OK) return rc;
    if (N == 0) return RR_OK;
    if (!obs || !sums) {
        rr_set_error("rr_column_sums_dev: obs and sums are required");
        return RR_E_NULL;
    }
    hipLaunchKernelGGL(column_sums_kernel,
                       dim3((unsigned)rr_ceil_div(N, 256)), dim3(256), 0,
                       (hipStream_t)stream, qsim, ld, obs, T, N, shift,
                       sums);
    RR_HIP(hipGetLastError());
    return RR_OK;
}

extern "C" int rr_set_device(int device)
{
    RR_HIP(hipSetDevice(device));
    return RR_OK;
}

extern "C" int rr_get_device(void)
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return d;
}
