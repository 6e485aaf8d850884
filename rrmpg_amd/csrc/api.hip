// api.hip -- error plumbing, device queries and the host-pointer entry points
// (rr_<model>_simulate) of the C-ABI declared in include/rrhip.h.
//
// The host-pointer family is the drop-in for the reference's seam, where
// Model.simulate() hands numpy arrays to run_* and gets numpy arrays back
// (reference: rrmpg/models/hbvedu.py:190-214).  It owns no state between
// calls: it uploads the (small, shared) forcing and the parameter block,
// sweeps the parameter-set axis in column blocks sized to the free HBM, runs
// the *_simulate_dev entry point on each block and copies each [T][nc] device
// slab into the caller's [T][N] array with a pitched copy.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void rr_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *rr_last_error(void) { return g_err; }
extern "C" int rr_version(void) { return 100; }

// ---- measurement / test options (rrhip.h RR_OPT_*) ---------------------------
static std::atomic<int64_t> g_options[RR_OPT_COUNT_] = {
    {0}, {-1} /* HBV variant: heuristic */, {0}, {0}};

int64_t rr_option(int option)
{
    return g_options[option].load(std::memory_order_relaxed);
}

extern "C" int rr_debug_set_option(int option, int64_t value)
{
    bool ok = false;
    switch (option) {
    case RR_OPT_HBV_VARIANT: ok = value >= -1 && value <= 2; break;
    case RR_OPT_GR4J_FORCE_LDS: ok = value == 0 || value == 1; break;
    case RR_OPT_MAX_BLOCK_COLS: ok = value >= 0; break;
    default: break;
    }
    if (!ok) {
        rr_set_error("rr_debug_set_option: option %d does not take %lld",
                     option, (long long)value);
        return RR_E_PARAM;
    }
    g_options[option].store(value, std::memory_order_relaxed);
    return RR_OK;
}

extern "C" int64_t rr_debug_get_option(int option)
{
    if (option < 1 || option >= RR_OPT_COUNT_) return INT64_MIN;
    return rr_option(option);
}

extern "C" int rr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int rr_check_common(const char *who, int64_t T, int64_t N, int64_t ld,
                    const void *params, const void *qobs, const void *sse)
{
    g_err[0] = 0;
    if (T < 0 || N < 0) {
        rr_set_error("%s: negative size (T=%lld, N=%lld)", who, (long long)T,
                     (long long)N);
        return RR_E_SIZE;
    }
    if (ld < N) {
        rr_set_error("%s: ld=%lld < N=%lld", who, (long long)ld, (long long)N);
        return RR_E_SIZE;
    }
    if (N > 0 && !params) {
        rr_set_error("%s: params is NULL", who);
        return RR_E_NULL;
    }
    if ((qobs == nullptr) != (sse == nullptr)) {
        rr_set_error("%s: qobs and sse must be given together", who);
        return RR_E_NULL;
    }
    return RR_OK;
}

// --------------------------------------------------------------------------
// host-pointer family
// --------------------------------------------------------------------------
namespace {

struct DevBuf {
    void *p = nullptr;
    int alloc(size_t bytes)
    {
        if (bytes == 0) bytes = 8;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            p = nullptr;
            rr_set_error("hipMalloc(%zu) failed: %s", bytes,
                         hipGetErrorString(e));
            return RR_E_HIP;
        }
        return RR_OK;
    }
    int upload(const void *src, size_t bytes)
    {
        int rc = alloc(bytes);
        if (rc != RR_OK) return rc;
        if (bytes) RR_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return RR_OK;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <class T> T *as() const { return (T *)p; }
};

int require_device()
{
    if (rr_device_count() < 1) {
        rr_set_error("no HIP device visible: librrhip has no CPU path");
        return RR_E_NODEVICE;
    }
    return RR_OK;
}

// One host output array [T][rows_per_t][N] mirrored by a device slab
// [T][rows_per_t][nc].
struct OutSpec {
    double *host;          // may be NULL (not requested)
    int64_t rows_per_t;    // 1, or L for the [T][L][N] storages
};

// Column-block width so that all requested slabs fit in a fraction of the
// free device memory.
int64_t pick_block(int64_t T, int64_t N, const std::vector<OutSpec> &outs)
{
    size_t per_col = 0;
    for (const OutSpec &o : outs)
        if (o.host) per_col += (size_t)T * (size_t)o.rows_per_t * 8;
    if (per_col == 0) return N;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
    size_t budget = free_b / 2;
    int64_t nc = (int64_t)(budget / per_col);
    nc = (nc / 256) * 256;
    if (nc < 256) nc = 256;
    // test hook: force a small column block to exercise the pitched gather
    const int64_t cap = rr_option(RR_OPT_MAX_BLOCK_COLS);
    if (cap > 0 && cap < nc) nc = cap;
    return nc < N ? nc : N;
}

// First touch of a freshly allocated host array is what bounds the gather:
// D2H into never-touched pageable memory runs at ~13 GB/s (page faults) against
// ~50 GB/s into touched pages (profiles/README.md).  So while the first kernel
// runs, a few host threads fault the output pages in (writing the zeros the
// caller's np.zeros would have produced lazily anyway; every byte is
// overwritten by the gather afterwards).
void prefault(const std::vector<OutSpec> &outs, int64_t T, int64_t N)
{
    unsigned nthreads = std::thread::hardware_concurrency();
    if (nthreads > 16) nthreads = 16;
    if (nthreads < 1) nthreads = 1;
    const size_t page = 4096;
    for (const OutSpec &o : outs) {
        if (!o.host) continue;
        const size_t bytes = (size_t)T * (size_t)o.rows_per_t * (size_t)N * 8;
        if (bytes < ((size_t)64 << 20)) continue;      // small: not worth it
        char *base = (char *)o.host;
        std::vector<std::thread> pool;
        const size_t chunk = (bytes / nthreads + page) & ~(page - 1);
        for (unsigned k = 0; k < nthreads; ++k) {
            const size_t lo = (size_t)k * chunk;
            if (lo >= bytes) break;
            const size_t hi = (lo + chunk < bytes) ? lo + chunk : bytes;
            pool.emplace_back([base, lo, hi]() {
                for (size_t p = lo; p < hi; p += page)
                    *(volatile char *)(base + p) = 0;
            });
        }
        for (std::thread &t : pool) t.join();
    }
}

// Runs `launch(i0, nc, slabs, d_sse)` for every column block and gathers the
// slabs into the host arrays.
template <class Launch>
int sweep_blocks(int64_t T, int64_t N, const std::vector<OutSpec> &outs,
                 double *sse_host, Launch launch)
{
    const int64_t nc_max = pick_block(T, N, outs);
    std::vector<DevBuf> slabs(outs.size());
    for (size_t k = 0; k < outs.size(); ++k)
        if (outs[k].host) {
            int rc = slabs[k].alloc((size_t)T * outs[k].rows_per_t * nc_max * 8);
            if (rc != RR_OK) return rc;
        }
    DevBuf d_sse;
    if (sse_host) {
        int rc = d_sse.alloc((size_t)nc_max * 8);
        if (rc != RR_OK) return rc;
    }
    std::vector<double *> ptrs(outs.size());
    for (size_t k = 0; k < outs.size(); ++k) ptrs[k] = slabs[k].as<double>();

    for (int64_t i0 = 0; i0 < N; i0 += nc_max) {
        const int64_t nc = (N - i0 < nc_max) ? (N - i0) : nc_max;
        int rc = launch(i0, nc, ptrs.data(), d_sse.as<double>());
        if (rc != RR_OK) return rc;
        if (i0 == 0) prefault(outs, T, N);      // overlaps the first kernel
        RR_HIP(hipStreamSynchronize(nullptr));
        for (size_t k = 0; k < outs.size(); ++k) {
            if (!outs[k].host) continue;
            RR_HIP(hipMemcpy2D(outs[k].host + i0, (size_t)N * 8, ptrs[k],
                               (size_t)nc * 8, (size_t)nc * 8,
                               (size_t)T * outs[k].rows_per_t,
                               hipMemcpyDeviceToHost));
        }
        if (sse_host)
            RR_HIP(hipMemcpy(sse_host + i0, d_sse.p, (size_t)nc * 8,
                             hipMemcpyDeviceToHost));
    }
    return RR_OK;
}

}  // namespace

extern "C" int rr_abc_simulate(const double *prec, int64_t T,
                               double initial_state, const double *params,
                               int64_t N, double *qsim, double *storage,
                               const double *qobs, double *sse)
{
    int rc = rr_check_common("rr_abc_simulate", T, N, N, params, qobs, sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (!prec) { rr_set_error("rr_abc_simulate: prec is NULL"); return RR_E_NULL; }
    if ((rc = require_device()) != RR_OK) return rc;
    DevBuf d_prec, d_par, d_qobs, ws;
    if ((rc = d_prec.upload(prec, (size_t)T * 8)) != RR_OK) return rc;
    if ((rc = d_par.upload(params, (size_t)N * 3 * 8)) != RR_OK) return rc;
    if (qobs && (rc = d_qobs.upload(qobs, (size_t)T * 8)) != RR_OK) return rc;
    const size_t wsb = rr_abc_workspace_bytes(T, N);
    if ((rc = ws.alloc(wsb)) != RR_OK) return rc;
    std::vector<OutSpec> outs = {{qsim, 1}, {storage, 1}};
    return sweep_blocks(T, N, outs, sse,
        [&](int64_t i0, int64_t nc, double **o, double *d_sse) {
            return rr_abc_simulate_dev(
                d_prec.as<double>(), T, initial_state,
                d_par.as<double>() + i0 * 3, nc, o[0], o[1], nc,
                qobs ? d_qobs.as<double>() : nullptr, qobs ? d_sse : nullptr,
                ws.p, wsb, nullptr);
        });
}

extern "C" int rr_hbvedu_simulate(
    const double *temp, const double *prec, const int8_t *month,
    const double *PE_m, const double *T_m, int64_t T, double snow_init,
    double soil_init, double s1_init, double s2_init, const double *params,
    int64_t N, double *qsim, double *snow, double *soil, double *s1,
    double *s2, const double *qobs, double *sse)
{
    int rc = rr_check_common("rr_hbvedu_simulate", T, N, N, params, qobs, sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (!temp || !prec || !month || !PE_m || !T_m) {
        rr_set_error("rr_hbvedu_simulate: NULL forcing pointer");
        return RR_E_NULL;
    }
    if ((rc = require_device()) != RR_OK) return rc;
    DevBuf d_temp, d_prec, d_month, d_pe, d_tm, d_par, d_qobs, ws;
    if ((rc = d_temp.upload(temp, (size_t)T * 8)) != RR_OK) return rc;
    if ((rc = d_prec.upload(prec, (size_t)T * 8)) != RR_OK) return rc;
    if ((rc = d_month.upload(month, (size_t)T)) != RR_OK) return rc;
    if ((rc = d_pe.upload(PE_m, 12 * 8)) != RR_OK) return rc;
    if ((rc = d_tm.upload(T_m, 12 * 8)) != RR_OK) return rc;
    if ((rc = d_par.upload(params, (size_t)N * 11 * 8)) != RR_OK) return rc;
    if (qobs && (rc = d_qobs.upload(qobs, (size_t)T * 8)) != RR_OK) return rc;
    const size_t wsb = rr_hbvedu_workspace_bytes(T, N);
    if ((rc = ws.alloc(wsb)) != RR_OK) return rc;
    std::vector<OutSpec> outs = {{qsim, 1}, {snow, 1}, {soil, 1}, {s1, 1},
                                 {s2, 1}};
    return sweep_blocks(T, N, outs, sse,
        [&](int64_t i0, int64_t nc, double **o, double *d_sse) {
            return rr_hbvedu_simulate_dev(
                d_temp.as<double>(), d_prec.as<double>(),
                d_month.as<int8_t>(), d_pe.as<double>(), d_tm.as<double>(), T,
                snow_init, soil_init, s1_init, s2_init,
                d_par.as<double>() + i0 * 11, nc, o[0], o[1], o[2], o[3],
                o[4], nc, qobs ? d_qobs.as<double>() : nullptr,
                qobs ? d_sse : nullptr, ws.p, wsb, nullptr);
        });
}

extern "C" int rr_gr4j_simulate(const double *prec, const double *etp,
                                int64_t T, double s_init, double r_init,
                                const double *params, int64_t N, double *qsim,
                                double *s_store, double *r_store,
                                const double *qobs, double *sse)
{
    int rc = rr_check_common("rr_gr4j_simulate", T, N, N, params, qobs, sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (!prec || !etp) {
        rr_set_error("rr_gr4j_simulate: NULL forcing pointer");
        return RR_E_NULL;
    }
    if ((rc = require_device()) != RR_OK) return rc;
    DevBuf d_prec, d_etp, d_par, d_qobs, ws;
    if ((rc = d_prec.upload(prec, (size_t)T * 8)) != RR_OK) return rc;
    if ((rc = d_etp.upload(etp, (size_t)T * 8)) != RR_OK) return rc;
    if ((rc = d_par.upload(params, (size_t)N * 4 * 8)) != RR_OK) return rc;
    if (qobs && (rc = d_qobs.upload(qobs, (size_t)T * 8)) != RR_OK) return rc;
    const size_t wsb = rr_gr4j_workspace_bytes(T, N);
    if ((rc = ws.alloc(wsb)) != RR_OK) return rc;
    std::vector<OutSpec> outs = {{qsim, 1}, {s_store, 1}, {r_store, 1}};
    return sweep_blocks(T, N, outs, sse,
        [&](int64_t i0, int64_t nc, double **o, double *d_sse) {
            int r = rr_gr4j_simulate_dev(
                d_prec.as<double>(), d_etp.as<double>(), T, s_init, r_init,
                d_par.as<double>() + i0 * 4, nc, o[0], o[1], o[2], nc,
                qobs ? d_qobs.as<double>() : nullptr, qobs ? d_sse : nullptr,
                ws.p, wsb, nullptr);
            return r != RR_OK ? r : rr_gr4j_plan_status(ws.p, nullptr);
        });
}

extern "C" int rr_cemaneige_simulate(
    const double *prec, const double *mean_temp, const double *frac_solid_prec,
    int64_t T, int64_t L, double snow_pack_init, double thermal_state_init,
    const double *params, int64_t N, double *outflow, double *G, double *eTG,
    const double *qobs, double *sse)
{
    int rc = rr_check_common("rr_cemaneige_simulate", T, N, N, params, qobs,
                             sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (L < 1) { rr_set_error("rr_cemaneige_simulate: L < 1"); return RR_E_PARAM; }
    if (!prec || !mean_temp || !frac_solid_prec) {
        rr_set_error("rr_cemaneige_simulate: NULL forcing pointer");
        return RR_E_NULL;
    }
    if ((rc = require_device()) != RR_OK) return rc;
    DevBuf d_prec, d_temp, d_frac, d_par, d_qobs, ws;
    const size_t tl = (size_t)T * (size_t)L * 8;
    if ((rc = d_prec.upload(prec, tl)) != RR_OK) return rc;
    if ((rc = d_temp.upload(mean_temp, tl)) != RR_OK) return rc;
    if ((rc = d_frac.upload(frac_solid_prec, tl)) != RR_OK) return rc;
    if ((rc = d_par.upload(params, (size_t)N * 2 * 8)) != RR_OK) return rc;
    if (qobs && (rc = d_qobs.upload(qobs, (size_t)T * 8)) != RR_OK) return rc;
    const size_t wsb = rr_cemaneige_workspace_bytes(T, L, N);
    if ((rc = ws.alloc(wsb)) != RR_OK) return rc;
    std::vector<OutSpec> outs = {{outflow, 1}, {G, L}, {eTG, L}};
    return sweep_blocks(T, N, outs, sse,
        [&](int64_t i0, int64_t nc, double **o, double *d_sse) {
            return rr_cemaneige_simulate_dev(
                d_prec.as<double>(), d_temp.as<double>(), d_frac.as<double>(),
                T, L, snow_pack_init, thermal_state_init,
                d_par.as<double>() + i0 * 2, nc, o[0], o[1], o[2], nc,
                qobs ? d_qobs.as<double>() : nullptr, qobs ? d_sse : nullptr,
                ws.p, wsb, nullptr);
        });
}

extern "C" int rr_cemaneigegr4j_simulate(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_solid_prec, int64_t T, int64_t L,
    double snow_pack_init, double thermal_state_init, double s_init,
    double r_init, const double *params, int64_t N, double *qsim, double *G,
    double *eTG, double *s_store, double *r_store, const double *qobs,
    double *sse)
{
    int rc = rr_check_common("rr_cemaneigegr4j_simulate", T, N, N, params,
                             qobs, sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (L < 1) { rr_set_error("rr_cemaneigegr4j_simulate: L < 1"); return RR_E_PARAM; }
    if (!prec || !mean_temp || !etp || !frac_solid_prec) {
        rr_set_error("rr_cemaneigegr4j_simulate: NULL forcing pointer");
        return RR_E_NULL;
    }
    if ((rc = require_device()) != RR_OK) return rc;
    DevBuf d_prec, d_temp, d_etp, d_frac, d_par, d_qobs, ws;
    const size_t tl = (size_t)T * (size_t)L * 8;
    if ((rc = d_prec.upload(prec, tl)) != RR_OK) return rc;
    if ((rc = d_temp.upload(mean_temp, tl)) != RR_OK) return rc;
    if ((rc = d_etp.upload(etp, (size_t)T * 8)) != RR_OK) return rc;
    if ((rc = d_frac.upload(frac_solid_prec, tl)) != RR_OK) return rc;
    if ((rc = d_par.upload(params, (size_t)N * 6 * 8)) != RR_OK) return rc;
    if (qobs && (rc = d_qobs.upload(qobs, (size_t)T * 8)) != RR_OK) return rc;
    const size_t wsb = rr_cemaneigegr4j_workspace_bytes(T, L, N);
    if ((rc = ws.alloc(wsb)) != RR_OK) return rc;
    std::vector<OutSpec> outs = {{qsim, 1}, {G, L}, {eTG, L}, {s_store, 1},
                                 {r_store, 1}};
    return sweep_blocks(T, N, outs, sse,
        [&](int64_t i0, int64_t nc, double **o, double *d_sse) {
            int r = rr_cemaneigegr4j_simulate_dev(
                d_prec.as<double>(), d_temp.as<double>(), d_etp.as<double>(),
                d_frac.as<double>(), T, L, snow_pack_init, thermal_state_init,
                s_init, r_init, d_par.as<double>() + i0 * 6, nc, o[0], o[1],
                o[2], o[3], o[4], nc, qobs ? d_qobs.as<double>() : nullptr,
                qobs ? d_sse : nullptr, ws.p, wsb, nullptr);
            return r != RR_OK ? r : rr_gr4j_plan_status(ws.p, nullptr);
        });
}

// ---- device self-test hook (not part of include/rrhip.h) -------------------
// out[i] = div_by_invariant_m(a[i], b[i]) and ref[i] = a[i] / b[i], host arrays.
__global__ void dbg_div_kernel(const double *a, const double *b, double *out,
                               double *ref, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double av = (i < n) ? a[i] : 1.0, bv = (i < n) ? b[i] : 1.0;
    const InvDivisor d = make_inv_divisor(bv);
    const double q = div_by_invariant_m(av, inv_div_numerator_mask0(av), d,
                                        RR_LANES(d.ok));
    if (i < n) {
        out[i] = q;
        ref[i] = av / bv;
    }
}

extern "C" int rrdbg_divide_by_invariant(const double *a, const double *b,
                                         double *out, double *ref, int64_t n)
{
    int rc = require_device();
    if (rc != RR_OK) return rc;
    DevBuf da, db, dout, dref;
    if ((rc = da.upload(a, (size_t)n * 8)) != RR_OK) return rc;
    if ((rc = db.upload(b, (size_t)n * 8)) != RR_OK) return rc;
    if ((rc = dout.alloc((size_t)n * 8)) != RR_OK) return rc;
    if ((rc = dref.alloc((size_t)n * 8)) != RR_OK) return rc;
    hipLaunchKernelGGL(dbg_div_kernel, dim3((unsigned)rr_ceil_div(n, 256)),
                       dim3(256), 0, nullptr, da.as<double>(), db.as<double>(),
                       dout.as<double>(), dref.as<double>(), n);
    RR_HIP(hipMemcpy(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    RR_HIP(hipMemcpy(ref, dref.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return RR_OK;
}

// ---- next tier: hysteresis / ice-melt couplings (host pointers) ------------
namespace {

enum SnowVariant { HYST = 1, ICE = 2 };

int snow_gr4j_host(const char *who, int variant, const double *prec,
                   const double *mean_temp, const double *etp,
                   const double *frac_ice, const double *frac_solid_prec,
                   int64_t T, int64_t L, double snow_pack_init,
                   double thermal_state_init, double sca_init, double s_init,
                   double r_init, const double *params, int64_t N,
                   double *qsim, double *G, double *eTG, double *s_store,
                   double *r_store, double *sca, double *icemelt,
                   double *snowmelt, const double *qobs, double *sse)
{
    const bool hyst = variant & HYST, ice = variant & ICE;
    const int npar = 6 + (hyst ? 2 : 0) + (ice ? 1 : 0);
    int rc = rr_check_common(who, T, N, N, params, qobs, sse);
    if (rc != RR_OK) return rc;
    if (T == 0 || N == 0) return RR_OK;
    if (L < 1) { rr_set_error("%s: L < 1", who); return RR_E_PARAM; }
    if (!prec || !mean_temp || !etp || !frac_solid_prec || (ice && !frac_ice)) {
        rr_set_error("%s: NULL forcing pointer", who);
        return RR_E_NULL;
    }
    if ((rc = require_device()) != RR_OK) return rc;
    DevBuf d_prec, d_temp, d_etp, d_frac, d_fice, d_par, d_qobs, ws;
    const size_t tl = (size_t)T * (size_t)L * 8;
    if ((rc = d_prec.upload(prec, tl)) != RR_OK) return rc;
    if ((rc = d_temp.upload(mean_temp, tl)) != RR_OK) return rc;
    if ((rc = d_etp.upload(etp, (size_t)T * 8)) != RR_OK) return rc;
    if ((rc = d_frac.upload(frac_solid_prec, tl)) != RR_OK) return rc;
    if (ice && (rc = d_fice.upload(frac_ice, (size_t)L * 8)) != RR_OK) return rc;
    if ((rc = d_par.upload(params, (size_t)N * npar * 8)) != RR_OK) return rc;
    if (qobs && (rc = d_qobs.upload(qobs, (size_t)T * 8)) != RR_OK) return rc;
    const size_t wsb = rr_snowgr4j_workspace_bytes(T, L, N);
    if ((rc = ws.alloc(wsb)) != RR_OK) return rc;
    std::vector<OutSpec> outs = {{qsim, 1}, {G, L}, {eTG, L}, {s_store, 1},
                                 {r_store, 1}, {sca, L}, {icemelt, 1},
                                 {snowmelt, 1}};
    return sweep_blocks(T, N, outs, sse,
        [&](int64_t i0, int64_t nc, double **o, double *d_sse) {
            const double *p = d_par.as<double>() + i0 * npar;
            const double *qo = qobs ? d_qobs.as<double>() : nullptr;
            double *so = qobs ? d_sse : nullptr;
            int r;
            if (hyst && ice)
                r = rr_cemaneigehystgr4jice_simulate_dev(
                    d_prec.as<double>(), d_temp.as<double>(),
                    d_etp.as<double>(), d_fice.as<double>(),
                    d_frac.as<double>(), T, L, snow_pack_init,
                    thermal_state_init, sca_init, s_init, r_init, p, nc, o[0],
                    o[1], o[2], o[3], o[4], o[5], o[6], o[7], nc, qo, so, ws.p,
                    wsb, nullptr);
            else if (hyst)
                r = rr_cemaneigehystgr4j_simulate_dev(
                    d_prec.as<double>(), d_temp.as<double>(),
                    d_etp.as<double>(), d_frac.as<double>(), T, L,
                    snow_pack_init, thermal_state_init, sca_init, s_init,
                    r_init, p, nc, o[0], o[1], o[2], o[3], o[4], o[5], nc, qo,
                    so, ws.p, wsb, nullptr);
            else
                r = rr_cemaneigegr4jice_simulate_dev(
                d_prec.as<double>(), d_temp.as<double>(), d_etp.as<double>(),
                d_fice.as<double>(), d_frac.as<double>(), T, L, snow_pack_init,
                thermal_state_init, s_init, r_init, p, nc, o[0], o[1], o[2],
                o[3], o[4], o[6], nc, qo, so, ws.p, wsb, nullptr);
            return r != RR_OK ? r : rr_gr4j_plan_status(ws.p, nullptr);
        });
}

}  // namespace

extern "C" int rr_cemaneigehystgr4j_simulate(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_solid_prec, int64_t T, int64_t L,
    double snow_pack_init, double thermal_state_init, double sca_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *sca, const double *qobs, double *sse)
{
    return snow_gr4j_host("rr_cemaneigehystgr4j_simulate", HYST, prec,
                          mean_temp, etp, nullptr, frac_solid_prec, T, L,
                          snow_pack_init, thermal_state_init, sca_init, s_init,
                          r_init, params, N, qsim, G, eTG, s_store, r_store,
                          sca, nullptr, nullptr, qobs, sse);
}

extern "C" int rr_cemaneigegr4jice_simulate(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *icemelt, const double *qobs, double *sse)
{
    return snow_gr4j_host("rr_cemaneigegr4jice_simulate", ICE, prec, mean_temp,
                          etp, frac_ice, frac_solid_prec, T, L, snow_pack_init,
                          thermal_state_init, 0.0, s_init, r_init, params, N,
                          qsim, G, eTG, s_store, r_store, nullptr, icemelt,
                          nullptr, qobs, sse);
}

extern "C" int rr_cemaneigehystgr4jice_simulate(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double sca_init, double s_init, double r_init, const double *params,
    int64_t N, double *qsim, double *G, double *eTG, double *s_store,
    double *r_store, double *sca, double *icemelt, double *snowmelt,
    const double *qobs, double *sse)
{
    return snow_gr4j_host("rr_cemaneigehystgr4jice_simulate", HYST | ICE, prec,
                          mean_temp, etp, frac_ice, frac_solid_prec, T, L,
                          snow_pack_init, thermal_state_init, sca_init, s_init,
                          r_init, params, N, qsim, G, eTG, s_store, r_store,
                          sca, icemelt, snowmelt, qobs, sse);
}
