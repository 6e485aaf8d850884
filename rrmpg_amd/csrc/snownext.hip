// snownext.hip -- C-ABI entry points of the ice-melt couplings
// (CemaneigeGR4JIce, CemaneigeHystGR4JIce) and the family's workspace sizes;
// kernels and launch logic: snownext_kernels.h.  The hysteresis coupling's
// entry point is a translation unit of its own (snownext_hyst.hip) because it
// is built with another instruction-scheduling strategy.
#include "snownext_kernels.h"

extern "C" size_t rr_snowgr4j_workspace_bytes(int64_t T, int64_t L, int64_t N)
{
    return cema_ws_bytes(T, L, true, N, 4, RR_SNOWNEXT_REG_LAYERS);   // N only matters for L > 5
}

extern "C" size_t rr_snowgr4j_workspace_bytes_x4(int64_t T, int64_t L,
                                                 int64_t N, double max_x4)
{
    return cema_ws_bytes(T, L, true, N, 4, RR_SNOWNEXT_REG_LAYERS) +
           rr_gr4j_uh_scratch_bytes(N, max_x4);
}

extern "C" int rr_cemaneigegr4jice_simulate_dev(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *icemelt, int64_t ld, const double *qobs, double *sse,
    void *workspace, size_t workspace_bytes, void *stream)
{
    return snow_gr4j_dev<false, true>(
        "rr_cemaneigegr4jice_simulate_dev", prec, mean_temp, etp, frac_ice,
        frac_solid_prec, T, L, snow_pack_init, thermal_state_init, 0.0,
        s_init, r_init, params, N, qsim, G, eTG, s_store, r_store, nullptr,
        icemelt, nullptr, ld, qobs, sse, workspace, workspace_bytes, stream);
}

extern "C" int rr_cemaneigehystgr4jice_simulate_dev(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double sca_init, double s_init, double r_init, const double *params,
    int64_t N, double *qsim, double *G, double *eTG, double *s_store,
    double *r_store, double *sca, double *icemelt, double *snowmelt,
    int64_t ld, const double *qobs, double *sse, void *workspace,
    size_t workspace_bytes, void *stream)
{
    return snow_gr4j_dev<true, true>(
        "rr_cemaneigehystgr4jice_simulate_dev", prec, mean_temp, etp,
        frac_ice, frac_solid_prec, T, L, snow_pack_init, thermal_state_init,
        sca_init, s_init, r_init, params, N, qsim, G, eTG, s_store, r_store,
        sca, icemelt, snowmelt, ld, qobs, sse, workspace, workspace_bytes,
        stream);
}
