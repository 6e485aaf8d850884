// sample.hip -- Monte-Carlo parameter sets drawn in HBM.
//
// The reference draws the sets on the host, one numpy.random.uniform(size=N)
// call per parameter in _param_list order (reference:
// rrmpg/models/basemodel.py:83-91; the ABC model draws a, then c, then one
// b ~ U(0, 1 - a[i]) per set, rrmpg/models/abcmodel.py:85-103), and every
// sweep then starts with an N x k x 8 B upload.  Here the block
// params[N][k] is filled where it will be read.
//
// The stream is numpy's own counter-based generator, so a host program can
// reproduce every set bit for bit:
//     rng = numpy.random.Generator(numpy.random.Philox(key=key))
//     for each parameter, in draw order:  rng.uniform(lo, hi, size=n_total)
// Philox4x64-10 (Salmon et al., SC'11) keyed with {key, 0}; numpy bumps the
// 256-bit counter BEFORE producing a block, so the stream's element e is word
// e % 4 of the block with counter e / 4 + 1, and uniform() maps a 64-bit
// word w to lo + (hi - lo) * ((w >> 11) * 2^-53), multiply and add rounded
// separately.  Being counter based, any rank can draw its own shard
// [n0, n0 + n) of a global n_total-set population without communication.
//
// (Documented deviation from the reference: its get_random_params uses the
// legacy global MT19937 stream; rrmpg_amd.models.*.get_random_params keeps
// that for seeded reproducibility of the reference's results, this entry
// point is the HBM-resident alternative.)
#include "common.h"
#include "../../include/rrhip.h"

namespace {

struct SamplePlan {
    double lo[RR_SAMPLE_MAX_PARAMS];
    double hi[RR_SAMPLE_MAX_PARAMS];
    int pos[RR_SAMPLE_MAX_PARAMS];   // rank of parameter j in the draw order
    int k;
    int hi_one_minus;                // parameter whose hi is 1 - params[.][0]
};

__device__ inline void philox_round(uint64_t (&c)[4], uint64_t k0, uint64_t k1)
{
    const uint64_t m0 = 0xD2E7470EE14C6C93ull, m1 = 0xCA5A826395121157ull;
    const uint64_t hi0 = __umul64hi(m0, c[0]), lo0 = m0 * c[0];
    const uint64_t hi1 = __umul64hi(m1, c[2]), lo1 = m1 * c[2];
    const uint64_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

// word (e % 4) of block (e / 4 + 1) of the Philox4x64-10 stream with key
// {key, 0}
__device__ inline uint64_t philox_word(uint64_t key, uint64_t e)
{
    uint64_t c[4] = {(e >> 2) + 1, 0, 0, 0};
    uint64_t k0 = key, k1 = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r > 0) {
            k0 += 0x9E3779B97F4A7C15ull;
            k1 += 0xBB67AE8584CAA73Bull;
        }
        philox_round(c, k0, k1);
    }
    const unsigned w = (unsigned)(e & 3);
    return w == 0 ? c[0] : w == 1 ? c[1] : w == 2 ? c[2] : c[3];
}

__global__ __launch_bounds__(256) void sample_kernel(
    SamplePlan plan, uint64_t key, int64_t n_total, int64_t n0, int64_t n,
    double *__restrict__ params)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double first = 0.0;
    for (int j = 0; j < plan.k; ++j) {
        const uint64_t e = (uint64_t)plan.pos[j] * (uint64_t)n_total +
                           (uint64_t)(n0 + i);
        const double u = (double)(philox_word(key, e) >> 11) *
                         (1.0 / 9007199254740992.0);
        const double lo = plan.lo[j];
        const double hi = j == plan.hi_one_minus ? 1 - first : plan.hi[j];
        const double v = lo + (hi - lo) * u;
        if (j == 0) first = v;
        params[i * plan.k + j] = v;
    }
}

}   // namespace

extern "C" int rr_sample_params_dev(uint64_t key, int k, const double *lo,
                                    const double *hi, const int *draw_pos,
                                    int hi_one_minus_first, int64_t n_total,
                                    int64_t n0, int64_t n, double *params,
                                    void *stream)
{
    if (k < 1 || k > RR_SAMPLE_MAX_PARAMS || n < 0 || n0 < 0 ||
        n0 + n > n_total || hi_one_minus_first >= k) {
        rr_set_error("rr_sample_params_dev: bad sizes (k=%d, n0=%lld, n=%lld, "
                     "n_total=%lld)", k, (long long)n0, (long long)n,
                     (long long)n_total);
        return RR_E_SIZE;
    }
    if (!lo || !hi || (n > 0 && !params)) {
        rr_set_error("rr_sample_params_dev: NULL argument");
        return RR_E_NULL;
    }
    SamplePlan plan;
    plan.k = k;
    plan.hi_one_minus = hi_one_minus_first > 0 ? hi_one_minus_first : -1;
    unsigned seen = 0;
    for (int j = 0; j < k; ++j) {
        plan.lo[j] = lo[j];
        plan.hi[j] = hi[j];
        plan.pos[j] = draw_pos ? draw_pos[j] : j;
        if (plan.pos[j] < 0 || plan.pos[j] >= k ||
            (seen & (1u << plan.pos[j]))) {
            rr_set_error("rr_sample_params_dev: draw_pos is not a "
                         "permutation of 0..k-1");
            return RR_E_PARAM;
        }
        seen |= 1u << plan.pos[j];
    }
    if (n == 0) return RR_OK;
    sample_kernel<<<dim3((unsigned)rr_ceil_div(n, 256)), dim3(256), 0,
                    (hipStream_t)stream>>>(plan, key, n_total, n0, n, params);
    RR_HIP(hipGetLastError());
    return RR_OK;
}
