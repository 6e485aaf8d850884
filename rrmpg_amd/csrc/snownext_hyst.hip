// snownext_hyst.hip -- C-ABI entry point of CemaneigeHystGR4J (kernels and
// launch logic: snownext_kernels.h).  A translation unit of its own: its
// kernels wait, at two or three waves per SIMD, for the dependent chains of
// the hysteresis routine, and LLVM's iterative-ilp scheduling strategy
// (Makefile: -mllvm -amdgpu-sched-strategy=iterative-ilp for this file only)
// orders them 3 % better than the default max-occupancy one -- 1M sets 133.6
// -> 129.8 ms, 125k 19.6 -> 18.9, the same bits -- while the ice-melt
// couplings lose 2 % or gain nothing with it and the fused CemaneigeGR4J
// kernel loses 3 % (profiles/r06_sched_strategy_ab.txt).
#include "snownext_kernels.h"

extern "C" int rr_cemaneigehystgr4j_simulate_dev(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_solid_prec, int64_t T, int64_t L,
    double snow_pack_init, double thermal_state_init, double sca_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *sca, int64_t ld, const double *qobs, double *sse, void *workspace,
    size_t workspace_bytes, void *stream)
{
    return snow_gr4j_dev<true, false>(
        "rr_cemaneigehystgr4j_simulate_dev", prec, mean_temp, etp, nullptr,
        frac_solid_prec, T, L, snow_pack_init, thermal_state_init, sca_init,
        s_init, r_init, params, N, qsim, G, eTG, s_store, r_store, sca,
        nullptr, nullptr, ld, qobs, sse, workspace, workspace_bytes, stream);
}

