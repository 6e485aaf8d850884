/*
 * rrhip.h -- C-ABI of librrhip.so, the MI355X (gfx950) ensemble
 * rainfall-runoff engine.
 *
 * The reference (kratzert/RRMPG) has no FFI: its seam is the Python call
 *     run_<model>(*forcing, *inits, params_record) -> tuple of [T] arrays
 * made once per parameter set from Model.simulate()'s Python loop and from
 * _loss() (reference: rrmpg/models/hbvedu.py:199-209 and :310-346, likewise
 * abcmodel.py:168-186, gr4j.py:162-183, cemaneige.py:218-245,
 * cemaneigegr4j.py:238-273).  Each entry point below replaces that loop AND
 * the run_* function under it with ONE batched call over N parameter sets:
 * argument order and meaning follow run_*'s signature, followed by the
 * parameter block, the outputs, and the optional fused error metric.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every buffer is owned by the caller and
 *     borrowed for the duration of the call;
 *   - `params` is the reference's structured-dtype buffer unchanged: packed
 *     all-float64 records in _param_list order, i.e. a row-major
 *     double[N][k] (reference: hbvedu.py:63-66, abcmodel.py:53-55,
 *     gr4j.py:57-60, cemaneige.py:64-65, cemaneigegr4j.py:67-72);
 *   - 2-D outputs are [T][ld] row-major with the parameter-set axis
 *     contiguous (the reference's qsim[T, N], hbvedu.py:191); 3-D storages
 *     are [T][L][ld] (reference: cemaneige.py:219-224).  ld >= N.  For the
 *     device entry points, make ld a multiple of 16 doubles (128 B) and the
 *     output pointers 128-byte aligned when N is large: a wave stores 512
 *     contiguous bytes of a row, and rows that start off a 64-byte boundary
 *     (dense ld = N with N % 8 != 0) cost a million-set sweep up to 40 %
 *     (profiles/r04_row_pitch.txt).  Correctness does not depend on it;
 *   - any output pointer may be NULL = "do not materialise it"
 *     (return_storage=False in the reference); storage outputs come WITH
 *     the discharge, as return_storage=True returns them, or not at all
 *     (storages without qsim: RR_E_NULL), and qsim may be NULL when only
 *     the sums below are wanted;
 *   - qobs/sse: if both non-NULL the kernel also accumulates, in time order,
 *     sse[i] = sum_t (qobs[t] - qsim[t, i])^2  -- the numerator of
 *     calc_mse / calc_nse (reference: rrmpg/utils/metrics.py:131, :72) so a
 *     Monte-Carlo sweep (rrmpg/tools/monte_carlo.py:66-71) need not move
 *     qsim at all;
 *   - return value: RR_OK or a negative RR_E_* code; rr_last_error() gives
 *     the text.  Numerical trouble is not an error: NaN propagates as in
 *     the reference.  Nothing here falls back to a CPU path.
 *
 * Numerics: fp64 throughout.  ABC, HBV-Edu's snow pack and every thermal
 * state are bit-identical to the reference's arithmetic; everything else is
 * within 1e-10 relative of it (measured over 30 years: 2e-14 HBV-Edu, 4e-13
 * the GR4J family, 1e-14 Cemaneige) -- power, tanh and roots are not libm's,
 * quotients by per-set constants are one multiply by the rounded reciprocal,
 * a product that feeds a sum is one FMA (each form has
 * a build flag that restores the reference's own sequence).  Sets outside
 * a generous box of meaningful parameters and states (recession factors
 * outside [0, 1], negative or huge Beta, 1e+-200, infinities, NaN; in the
 * GR4J family x1..x3, the initial stores and every snow parameter / initial
 * snow state beyond 1e6 in magnitude or not a number; forcing that is not a
 * number of at most 1e6) are computed with the reference's own statement
 * sequence in a second kernel behind the fast one, so their infinities and
 * NaN appear where and as the reference produces them, day by day.  The one
 * population that keeps the fast forms regardless is GR4J-family sets with
 * x4 > 20 days (the reference path holds 20-day unit hydrographs in
 * registers) and Cemaneige stacks of more than 5 layers.
 *
 * Two families:
 *   rr_<model>_simulate      host pointers; synchronous; the library moves
 *                            data to the GPU, runs, and copies results back
 *                            (drop-in for the reference's host-array seam).
 *   rr_<model>_simulate_dev  device pointers on the current HIP device;
 *                            enqueued on `stream` (a hipStream_t, may be
 *                            NULL = default stream) and asynchronous;
 *                            needs a caller-provided device workspace of
 *                            rr_<model>_workspace_bytes() bytes.
 */
#ifndef RRHIP_H
#define RRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_OK          0
#define RR_E_NULL     -1  /* a required pointer is NULL                     */
#define RR_E_SIZE     -2  /* negative or inconsistent size / ld < N / more
                           * than 2e9 timesteps (days are counted in 32 bits)
                           * / HBV-Edu: ld beyond 2^27 - 1 columns (a trip's
                           * rows are addressed through 32-bit offsets)       */
#define RR_E_HIP      -3  /* a HIP runtime call failed                      */
#define RR_E_PARAM    -4  /* parameter value the kernels cannot represent   */
#define RR_E_NODEVICE -5  /* no usable gfx950 device                        */
#define RR_E_WORKSPACE -6 /* workspace missing or too small                 */

/* Cemaneige and its couplings: up to this many elevation layers -- the
 * model's own five equal-area zones -- keep their snow states in registers;
 * more layers run through an HBM scratch (slower, same results). */
#define RR_CEMANEIGE_MAX_LAYERS 5
/* (the hysteresis / ice-melt couplings' name for it) */
#define RR_SNOWNEXT_REG_LAYERS RR_CEMANEIGE_MAX_LAYERS
/* GR4J: largest x4 whose unit hydrographs (ceil(x4) ordinates for UH1,
 * ceil(2*x4+1) for UH2) live on chip -- registers up to 10, LDS up to this.
 * Longer ones run too, as in the reference, from a scratch in HBM behind the
 * workspace: size it with the rr_*_workspace_bytes_x4 queries. */
#define RR_GR4J_MAX_X4 20.0

int rr_version(void);
/* Number of visible HIP devices (0 if none / no driver). */
int rr_device_count(void);
/* Text of the last error on the calling thread ("" if none). */
const char *rr_last_error(void);
/* Select / query the HIP device the calling thread's later calls run on (one
 * process per GPU: call once with the local rank).  rr_get_device: -1 if no
 * device. */
int rr_set_device(int device);
int rr_get_device(void);

/* The host-pointer family keeps, per device, its streams and its small device
 * buffers (inputs, parameter block, workspace, output slabs up to 64 MiB
 * each) between calls, and does not upload an input again whose bytes have
 * not changed since the previous call -- Model.fit() makes thousands of
 * one-set calls with the same forcing.  Large results additionally leave a
 * 256-MiB pinned staging ring behind.  This returns all of that memory (the
 * next call allocates again). */
int rr_release_cached_memory(void);

/* ---- the job's one collective ----------------------
 * A sweep sharded over several processes / GPUs (contiguous blocks of sets,
 * forcing replicated, no exchange on the data path: the loop the reference
 * runs over its sets, rrmpg/tools/monte_carlo.py:61-71, cut into blocks)
 * exchanges ONE thing: the per-set scores, 8 bytes a set, all-gathered so
 * that every rank holds the whole sweep's.  In Python that is
 * rrmpg_amd.sharding.allgather_scores over torch.distributed; these entry
 * points are the same exchange over RCCL / xGMI for a binder without Python
 * or torch.  librccl is opened at first use: librrhip.so itself does not
 * link against it.
 *   rr_shard_bounds     rank's block [first, stop) of n_total sets: sizes
 *                       differ by at most one, the first n_total % world
 *                       ranks hold the longer ones (sharding.shard_bounds)
 *   rr_comm_unique_id   rank 0: RR_COMM_ID_BYTES bytes to hand to every rank
 *                       out of band (ncclGetUniqueId)
 *   rr_comm_init        every rank, its GPU selected (rr_set_device /
 *                       hipSetDevice) beforehand; collective, blocks until
 *                       all ranks have called (ncclCommInitRank)
 *   rr_allgather_metric local: this rank's n_local device doubles; all:
 *                       n_total device doubles, block r of rank r in set
 *                       order (local may be all + first: in place).  Equal
 *                       blocks (n_total % world == 0): one ncclAllGather;
 *                       ragged blocks: one group of broadcasts, rank r the
 *                       root of block r -- no padding, nothing
 *                       allocated.  Enqueued on `stream` (hipStream_t),
 *                       returns at once like the *_simulate_dev family.
 *   rr_comm_destroy
 * One process that drives several GPUs itself (the clique of SURVEY.md 8e's
 * sketch) uses instead of the id exchange:
 *   rr_comm_init_all    comms_out[ndev]: the rank-j communicator lives on
 *                       devices[j] (NULL: devices 0..ndev-1; no device twice)
 *                       (ncclCommInitAll); each is destroyed with
 *                       rr_comm_destroy
 *   rr_comm_group_start / rr_comm_group_end   bracket the ndev
 *                       rr_allgather_metric calls the one thread issues, one
 *                       per communicator, each with its device's buffers and
 *                       stream (ncclGroupStart / ncclGroupEnd)                */
#define RR_COMM_ID_BYTES 128
int rr_shard_bounds(int64_t n_total, int world, int rank, int64_t *first,
                    int64_t *stop);
int rr_comm_unique_id(void *id_out);
int rr_comm_init(void **comm_out, int world, int rank, const void *id);
int rr_allgather_metric(void *comm, const double *local, int64_t n_local,
                        double *all, int64_t n_total, void *stream);
int rr_comm_destroy(void *comm);
int rr_comm_init_all(void **comms_out, int ndev, const int *devices);
int rr_comm_group_start(void);
int rr_comm_group_end(void);

/* Options: which GPUs a host-pointer call spreads over, and which kernel
 * variant / blocking a call uses where the library's own choice (by sweep
 * size) is to be overridden.  Every option is an integer indexed by RR_OPT_*.
 * Three ways to set one, consulted in this order:
 *   1. per call: rr_<model>_simulate_opt(..., const rr_call_options *opt) --
 *      the form a product caller uses (monte_carlo(gpus=...), sharding.sweep);
 *      it holds for that call only, on the calling thread and the shard
 *      threads the call starts.  Re-entrant: concurrent calls from several
 *      threads, each with its own options, do not see each other;
 *   2. per thread: rr_thread_options(opt) -- standing options of the calling
 *      thread for its later calls (the *_simulate_dev family has no per-call
 *      argument); NULL clears them;
 *   3. process-wide: rr_debug_set_option -- MEASUREMENT AND TEST HOOKS ONLY
 *      (A/B timing of kernel variants, parity tests that must reach a variant
 *      the size heuristics would not pick).  No product path sets them.
 * No environment variable is consulted anywhere in the library.
 * rr_debug_set_option / rr_call_options_set return RR_E_PARAM for an unknown
 * option or value; rr_debug_get_option returns the process-wide value (or
 * INT64_MIN for an unknown option). */
#define RR_OPT_HBV_VARIANT     1 /* -1 by sweep size (default); 0 the plain
                                  * loop (one scalar record load at the top of
                                  * each day; time-tiled for many waves); 3
                                  * three records rotating, a day's record
                                  * asked for two days ahead                 */
#define RR_OPT_GR4J_FORCE_LDS  2 /* 0 (default) / 1: unit hydrographs in LDS
                                  * even where a register tier would do      */
#define RR_OPT_MAX_BLOCK_COLS  3 /* 0 (default) = sized to free HBM; > 0 caps
                                  * the host-pointer family's column blocks  */
#define RR_OPT_GATHER_THREADS  4 /* 0 (default) = 12 (or fewer cores); > 0:
                                  * host threads scattering the staging ring
                                  * of a large gather into the caller's array */
#define RR_OPT_FUSED_VARIANT   5 /* CemaneigeGR4J kernel: 0 by sweep size
                                  * (default: 3 for at most two waves per
                                  * SIMD, else 1); 1 the many-waves kernel; 2
                                  * the small-sweep kernel (constants and melt
                                  * thresholds in VGPRs) wherever it exists; 3
                                  * the small-sweep kernel with an optimistic
                                  * GR4J half (votes noted, the half redone if
                                  * one failed)                              */
#define RR_OPT_GR4J_VARIANT    6 /* GR4J kernel: 0 the library's choice
                                  * (default: the optimistic kernel --
                                  * branch-free day, redone if a vote failed
                                  * -- where it exists); 1 gr4j_kernel in
                                  * every tier, every vote decided on the spot */
#define RR_OPT_HOST_SHARDS     7 /* the host-pointer family (rr_<model>_simulate
                                  * [_opt]) over several GPUs inside ONE call
                                  * -- the `ndev` of the call: 0 / 1
                                  * (default) the current device only; S > 1
                                  * cuts the N sets into S contiguous shards,
                                  * shard j on device (current + j) % count,
                                  * each from its own host thread, filling its
                                  * columns of the caller's [T][N] arrays; -1
                                  * one shard per visible device.  No
                                  * collective: sets are independent, the
                                  * forcing is uploaded to every device.      */
#define RR_OPT_TIME_TILES      8 /* HBV-Edu, GR4J, Cemaneige: the time axis of
                                  * every wave's sets in pieces, the states
                                  * handed from piece to piece through HBM,
                                  * which evens out the last round of
                                  * equal-length waves.  Work
                                  * items are handed out by an atomic ticket
                                  * counter (HBV-Edu: to persistent waves), so
                                  * an item only ever waits for one a running
                                  * wave already holds.  -1 by sweep size
                                  * (default), 0 never, k > 1 pieces          */
#define RR_OPT_WARM_RECORDS    9 /* every time-loop kernel: the waves read a
                                  * share of the day records with ordinary
                                  * loads when they start, so that the loop's
                                  * scalar loads hit the L2 of their XCD
                                  * (DESIGN.md 3.2).  -1 by sweep size and
                                  * mode (default), 0 never, 1 always         */
#define RR_OPT_COUNT_          10
int rr_debug_set_option(int option, int64_t value);
int64_t rr_debug_get_option(int option);

#define RR_OPT_UNSET INT64_MIN   /* "the library's choice" in rr_call_options */
typedef struct rr_call_options {
    size_t struct_bytes;         /* sizeof(rr_call_options), set by _init     */
    int64_t value[16];           /* indexed by RR_OPT_*                       */
} rr_call_options;
/* All options RR_OPT_UNSET. */
void rr_call_options_init(rr_call_options *opt);
/* opt->value[option] = value after the range check of rr_debug_set_option
 * (RR_OPT_UNSET is always accepted). */
int rr_call_options_set(rr_call_options *opt, int option, int64_t value);
/* Standing options of the calling thread (copied; NULL clears). */
int rr_thread_options(const rr_call_options *opt);

/* ---- ABC model -------------------------------------------------------
 * replaces run_abcmodel(prec, initial_state, params)
 * (reference: rrmpg/models/abcmodel_model.py:15-60); params = {a, b, c}. */
size_t rr_abc_workspace_bytes(int64_t T, int64_t N);
int rr_abc_simulate_dev(const double *prec, int64_t T, double initial_state,
                        const double *params, int64_t N,
                        double *qsim, double *storage, int64_t ld,
                        const double *qobs, double *sse,
                        void *workspace, size_t workspace_bytes, void *stream);
int rr_abc_simulate(const double *prec, int64_t T, double initial_state,
                    const double *params, int64_t N,
                    double *qsim, double *storage,
                    const double *qobs, double *sse);
/* the same with per-call options (NULL: none) */
int rr_abc_simulate_opt(const double *prec, int64_t T, double initial_state,
                    const double *params, int64_t N,
                    double *qsim, double *storage,
                    const double *qobs, double *sse,
    const rr_call_options *opt);

/* ---- HBV-Edu ---------------------------------------------------------
 * replaces run_hbvedu(temp, prec, month, PE_m, T_m, snow_init, soil_init,
 *                     s1_init, s2_init, params)
 * (reference: rrmpg/models/hbvedu_model.py:15-129); month holds 0..11
 * (already decremented, hbvedu.py:164); PE_m, T_m have 12 entries;
 * params = {T_t, DD, FC, Beta, C, PWP, K_0, K_1, K_2, K_p, L}. */
size_t rr_hbvedu_workspace_bytes(int64_t T, int64_t N);
int rr_hbvedu_simulate_dev(const double *temp, const double *prec,
                           const int8_t *month, const double *PE_m,
                           const double *T_m, int64_t T,
                           double snow_init, double soil_init, double s1_init,
                           double s2_init, const double *params, int64_t N,
                           double *qsim, double *snow, double *soil,
                           double *s1, double *s2, int64_t ld,
                           const double *qobs, double *sse,
                           void *workspace, size_t workspace_bytes,
                           void *stream);
int rr_hbvedu_simulate(const double *temp, const double *prec,
                       const int8_t *month, const double *PE_m,
                       const double *T_m, int64_t T,
                       double snow_init, double soil_init, double s1_init,
                       double s2_init, const double *params, int64_t N,
                       double *qsim, double *snow, double *soil, double *s1,
                       double *s2, const double *qobs, double *sse);
/* the same with per-call options (NULL: none) */
int rr_hbvedu_simulate_opt(const double *temp, const double *prec,
                       const int8_t *month, const double *PE_m,
                       const double *T_m, int64_t T,
                       double snow_init, double soil_init, double s1_init,
                       double s2_init, const double *params, int64_t N,
                       double *qsim, double *snow, double *soil, double *s1,
                       double *s2, const double *qobs, double *sse,
    const rr_call_options *opt);

/* Multi-catchment HBV-Edu (BASELINE.json configs[4]): C independent
 * catchments, each with its own forcing and its own N parameter sets, in ONE
 * launch -- replaces the user's outer loop `for basin: model.simulate(...)`
 * around hbvedu.py:82-214, whose per-basin sweeps (10k sets = 157 waves) are
 * far too small to fill 1024 SIMDs one at a time.  Device pointers only; all
 * arrays catchment-major: temp/prec/month [C][T], PE_m/T_m [C][12],
 * inits [C][4] = {snow, soil, s1, s2}, params [C][N][11], outputs
 * [C][T][ld], qobs [C][T], sse [C][N]. */
size_t rr_hbvedu_catchments_workspace_bytes(int64_t T, int64_t C, int64_t N);
int rr_hbvedu_simulate_catchments_dev(const double *temp, const double *prec,
                                      const int8_t *month, const double *PE_m,
                                      const double *T_m, int64_t T, int64_t C,
                                      const double *inits,
                                      const double *params, int64_t N,
                                      double *qsim, double *snow, double *soil,
                                      double *s1, double *s2, int64_t ld,
                                      const double *qobs, double *sse,
                                      void *workspace, size_t workspace_bytes,
                                      void *stream);

/* ---- GR4J ------------------------------------------------------------
 * replaces run_gr4j(prec, etp, s_init, r_init, params)
 * (reference: rrmpg/models/gr4j_model.py:15-192); params = {x1,x2,x3,x4};
 * s_init / r_init are fractions of x1 / x3 (gr4j_model.py:64-65); out[k] is
 * the state after day k (the reference's artificial step 0 is dropped,
 * gr4j_model.py:157).  RR_E_PARAM if any x4 gives no ordinates
 * (ceil(x4) < 1: the reference raises IndexError).  Any other x4 runs, as in
 * the reference (gr4j_model.py:68-79): up to RR_GR4J_MAX_X4 with the
 * workspace of rr_gr4j_workspace_bytes, beyond it with the larger one of
 * rr_gr4j_workspace_bytes_x4(T, N, largest x4 of the block) -- the bytes
 * behind the base workspace are the unit-hydrograph scratch (N * (6*ceil(x4)
 * + 2) * 8 B); a block whose x4 exceeds what its workspace holds writes
 * nothing and reports RR_E_PARAM through rr_gr4j_plan_status.  The
 * host-pointer family sizes the scratch itself (x4 <= 1e5).  Results do not
 * depend on which storage a launch uses: every tier runs the same
 * arithmetic. */
size_t rr_gr4j_workspace_bytes(int64_t T, int64_t N);
size_t rr_gr4j_workspace_bytes_x4(int64_t T, int64_t N, double max_x4);
int rr_gr4j_simulate_dev(const double *prec, const double *etp, int64_t T,
                         double s_init, double r_init,
                         const double *params, int64_t N,
                         double *qsim, double *s_store, double *r_store,
                         int64_t ld, const double *qobs, double *sse,
                         void *workspace, size_t workspace_bytes,
                         void *stream);
/* Deferred parameter check of the GR4J family.  Every *_simulate_dev entry
 * whose model contains GR4J (rr_gr4j_, rr_cemaneigegr4j_, rr_cemaneigehyst
 * gr4j_, rr_cemaneigegr4jice_, rr_cemaneigehystgr4jice_simulate_dev) is
 * fully asynchronous: the scan of x4 that picks the unit-hydrograph storage
 * runs on the GPU and is never read back by the call.  A parameter block
 * with a set the kernels cannot run (ceil(x4) < 1 or NaN: the reference
 * raises IndexError; x4 beyond what the workspace's unit-hydrograph scratch
 * holds, see rr_gr4j_workspace_bytes_x4) makes that sweep write NOTHING;
 * this function, given the workspace of the call, waits for `stream` and
 * returns RR_E_PARAM (with rr_last_error() text) or RR_OK.  The host-pointer
 * family calls it itself and returns the error directly. */
int rr_gr4j_plan_status(const void *workspace, void *stream);
int rr_gr4j_simulate(const double *prec, const double *etp, int64_t T,
                     double s_init, double r_init,
                     const double *params, int64_t N,
                     double *qsim, double *s_store, double *r_store,
                     const double *qobs, double *sse);
/* the same with per-call options (NULL: none) */
int rr_gr4j_simulate_opt(const double *prec, const double *etp, int64_t T,
                     double s_init, double r_init,
                     const double *params, int64_t N,
                     double *qsim, double *s_store, double *r_store,
                     const double *qobs, double *sse,
    const rr_call_options *opt);

/* ---- Cemaneige snow routine -------------------------------------------
 * replaces run_cemaneige(prec, mean_temp, frac_solid_prec, snow_pack_init,
 *                        thermal_state_init, params)
 * (reference: rrmpg/models/cemaneige_model.py:15-126); the three forcing
 * arrays are [T][L] row-major; params = {CTG, Kf}; G, eTG are [T][L][ld].
 * L >= 1.  qobs/sse compare against `outflow`. */
size_t rr_cemaneige_workspace_bytes(int64_t T, int64_t L, int64_t N);
int rr_cemaneige_simulate_dev(const double *prec, const double *mean_temp,
                              const double *frac_solid_prec, int64_t T,
                              int64_t L, double snow_pack_init,
                              double thermal_state_init,
                              const double *params, int64_t N,
                              double *outflow, double *G, double *eTG,
                              int64_t ld, const double *qobs, double *sse,
                              void *workspace, size_t workspace_bytes,
                              void *stream);
int rr_cemaneige_simulate(const double *prec, const double *mean_temp,
                          const double *frac_solid_prec, int64_t T, int64_t L,
                          double snow_pack_init, double thermal_state_init,
                          const double *params, int64_t N,
                          double *outflow, double *G, double *eTG,
                          const double *qobs, double *sse);
/* the same with per-call options (NULL: none) */
int rr_cemaneige_simulate_opt(const double *prec, const double *mean_temp,
                          const double *frac_solid_prec, int64_t T, int64_t L,
                          double snow_pack_init, double thermal_state_init,
                          const double *params, int64_t N,
                          double *outflow, double *G, double *eTG,
                          const double *qobs, double *sse,
    const rr_call_options *opt);

/* Forcing preprocessing of the Cemaneige family on the device -- replaces
 * extrapolate_precipitation, extrapolate_temperature and
 * calculate_solid_fraction (reference: rrmpg/models/cemaneige_utils.py:
 * 100-158, 160-207, 15-98): station series [T] (device) -> the [T][L] layer
 * arrays the *_simulate_dev entries take (device), same fp64 operations.
 * altitudes: host, [L].  prec_factor: host, [L], or NULL = computed here as
 * exp((z - z_station) * 0.0004) with the C library's exp (what numba calls;
 * pass numpy's values to match a numpy caller bit for bit).  A model without
 * elevation layers is L = 1 with altitudes[0] = met_station_height. */
size_t rr_cemaneige_layers_workspace_bytes(int64_t L);
int rr_cemaneige_layers_dev(const double *prec, const double *mean_temp,
                            const double *min_temp, const double *max_temp,
                            int64_t T, const double *altitudes, int64_t L,
                            double met_station_height,
                            const double *prec_factor, double *layer_prec,
                            double *layer_mean_temp, double *frac_solid_prec,
                            void *workspace, size_t workspace_bytes,
                            void *stream);

/* ---- Cemaneige + GR4J coupled ------------------------------------------
 * replaces run_cemaneigegr4j(prec, mean_temp, etp, frac_solid_prec,
 *                            snow_pack_init, thermal_state_init, s_init,
 *                            r_init, params)
 * (reference: rrmpg/models/cemaneigegr4j_model.py:16-63);
 * params = {CTG, Kf, x1, x2, x3, x4}.  One fused pass: the snow routine's
 * outflow feeds GR4J's precipitation in registers, no [T] intermediate. */
size_t rr_cemaneigegr4j_workspace_bytes(int64_t T, int64_t L, int64_t N);
size_t rr_cemaneigegr4j_workspace_bytes_x4(int64_t T, int64_t L, int64_t N,
                                           double max_x4);
int rr_cemaneigegr4j_simulate_dev(const double *prec, const double *mean_temp,
                                  const double *etp,
                                  const double *frac_solid_prec, int64_t T,
                                  int64_t L, double snow_pack_init,
                                  double thermal_state_init, double s_init,
                                  double r_init, const double *params,
                                  int64_t N, double *qsim, double *G,
                                  double *eTG, double *s_store,
                                  double *r_store, int64_t ld,
                                  const double *qobs, double *sse,
                                  void *workspace, size_t workspace_bytes,
                                  void *stream);
int rr_cemaneigegr4j_simulate(const double *prec, const double *mean_temp,
                              const double *etp, const double *frac_solid_prec,
                              int64_t T, int64_t L, double snow_pack_init,
                              double thermal_state_init, double s_init,
                              double r_init, const double *params, int64_t N,
                              double *qsim, double *G, double *eTG,
                              double *s_store, double *r_store,
                              const double *qobs, double *sse);
/* the same with per-call options (NULL: none) */
int rr_cemaneigegr4j_simulate_opt(const double *prec, const double *mean_temp,
                              const double *etp, const double *frac_solid_prec,
                              int64_t T, int64_t L, double snow_pack_init,
                              double thermal_state_init, double s_init,
                              double r_init, const double *params, int64_t N,
                              double *qsim, double *G, double *eTG,
                              double *s_store, double *r_store,
                              const double *qobs, double *sse,
    const rr_call_options *opt);

/* ---- per-set skill scores from a resident discharge array ---------------
 * One pass over qsim[T][ld] (device) against obs[T] (device) gives, for each
 * of the N columns, sums[i] = {sum q, sum q^2, sum q*obs, sum (obs-q)^2}
 * (sums: device double[N][4]), from which MSE, RMSE, NSE, KGE, alpha, beta
 * and Pearson r follow on the host -- replaces the per-column Python loop
 * over calc_mse (reference: rrmpg/tools/monte_carlo.py:66-71) and the
 * per-call array copies of rrmpg/utils/metrics.py:29-299. */
int rr_column_sums_dev(const double *qsim, int64_t ld, const double *obs,
                       int64_t T, int64_t N, double *sums, void *stream);
/* The same with the first three sums taken about `shift`:
 *   { sum (q-c), sum (q-c)^2, sum (q-c)*(obs-c), sum (obs-q)^2 },  c = shift.
 * With c = mean(obs) the variance / covariance the host derives for KGE,
 * alpha and r (np.std / pearsonr in rrmpg/utils/metrics.py:139-299, which are
 * two-pass) do not cancel for large, nearly constant series. */
int rr_column_sums_shifted_dev(const double *qsim, int64_t ld,
                               const double *obs, int64_t T, int64_t N,
                               double shift, double *sums, void *stream);

/* ---- Monte-Carlo parameter sets drawn in HBM ------------------------------
 * Fills params[n][k] (device, the AoS block every rr_*_simulate_dev takes)
 * with uniform draws -- replaces BaseModel.get_random_params (reference:
 * rrmpg/models/basemodel.py:68-91; ABC's rule b ~ U(0, 1 - a):
 * rrmpg/models/abcmodel.py:70-103) plus the upload of its result.
 * The stream is numpy's Philox bit generator, reproducible on a host as
 *     rng = numpy.random.Generator(numpy.random.Philox(key=key))
 *     for j in draw order: column_j = rng.uniform(lo[j], hi[j], size=n_total)
 * and this call writes rows n0 .. n0+n-1 of that n_total-row population (so
 * ranks draw their shards of one population independently).  lo, hi
 * (host, [k]) are the bounds in parameter order; draw_pos (host, [k], or
 * NULL = parameter order) is each parameter's rank in the draw order;
 * hi_one_minus_first = j > 0 makes parameter j's upper bound 1 - params[i][0]
 * (ABC: j = 1), 0 disables it.  (The reference's sampler uses numpy's legacy
 * global MT19937 stream; that remains available, unchanged, as
 * Model.get_random_params on the host.) */
#define RR_SAMPLE_MAX_PARAMS 16
int rr_sample_params_dev(uint64_t key, int k, const double *lo,
                         const double *hi, const int *draw_pos,
                         int hi_one_minus_first, int64_t n_total, int64_t n0,
                         int64_t n, double *params, void *stream);

/* ==== next tier: SWE-SCA hysteresis snow routine, ice melt, couplings ====
 * The reference's CemaneigeHystGR4J, CemaneigeGR4JIce, CemaneigeHystGR4JIce
 * (SURVEY.md section 8f N1).  Same conventions as above; any L >= 1 (more
 * than RR_SNOWNEXT_REG_LAYERS layers run from an HBM state scratch that is
 * part of the workspace, so size it with the N of the call);
 * `frac_ice` is [L]; sca is [T][L][ld]; icemelt, snowmelt are [T][ld]
 * (snowmelt = the snow routine's layer-mean outflow before the ice melt is
 * added).  All variants share one workspace size. */
size_t rr_snowgr4j_workspace_bytes(int64_t T, int64_t L, int64_t N);
size_t rr_snowgr4j_workspace_bytes_x4(int64_t T, int64_t L, int64_t N,
                                      double max_x4);

/* replaces run_cemaneigehystgr4j(prec, mean_temp, etp, frac_solid_prec,
 *     snow_pack_init, thermal_state_init, sca_init, s_init, r_init, params)
 * (reference: rrmpg/models/cemaneigehystgr4j_model.py:17-79 over
 * cemaneigehyst_model.py:5-166); params = {CTG,Kf,Thacc,Rsp,x1,x2,x3,x4}.
 * The reference's `rain` output is parameter independent (prec - prec*frac)
 * and is rebuilt on the host. */
int rr_cemaneigehystgr4j_simulate_dev(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_solid_prec, int64_t T, int64_t L,
    double snow_pack_init, double thermal_state_init, double sca_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *sca, int64_t ld, const double *qobs, double *sse, void *workspace,
    size_t workspace_bytes, void *stream);
int rr_cemaneigehystgr4j_simulate(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_solid_prec, int64_t T, int64_t L,
    double snow_pack_init, double thermal_state_init, double sca_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *sca, const double *qobs, double *sse);
/* the same with per-call options (NULL: none) */
int rr_cemaneigehystgr4j_simulate_opt(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_solid_prec, int64_t T, int64_t L,
    double snow_pack_init, double thermal_state_init, double sca_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *sca, const double *qobs, double *sse,
    const rr_call_options *opt);

/* replaces run_cemaneigegr4jice(prec, mean_temp, etp, frac_ice,
 *     frac_solid_prec, snow_pack_init, thermal_state_init, s_init, r_init,
 *     params)
 * (reference: rrmpg/models/cemaneigegr4jice_model.py:20-93 over
 * icemelt_model.py:15-65); params = {CTG,Kf,x1,x2,x3,x4,DDF}. */
int rr_cemaneigegr4jice_simulate_dev(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *icemelt, int64_t ld, const double *qobs, double *sse,
    void *workspace, size_t workspace_bytes, void *stream);
int rr_cemaneigegr4jice_simulate(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *icemelt, const double *qobs, double *sse);
/* the same with per-call options (NULL: none) */
int rr_cemaneigegr4jice_simulate_opt(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double s_init, double r_init, const double *params, int64_t N,
    double *qsim, double *G, double *eTG, double *s_store, double *r_store,
    double *icemelt, const double *qobs, double *sse,
    const rr_call_options *opt);

/* replaces run_cemaneigehystgr4jice(prec, mean_temp, etp, frac_ice,
 *     frac_solid_prec, snow_pack_init, thermal_state_init, sca_init, s_init,
 *     r_init, params)
 * (reference: rrmpg/models/cemaneigehystgr4jice_model.py:22-104);
 * params = {CTG,Kf,Thacc,Rsp,x1,x2,x3,x4,DDF}. */
int rr_cemaneigehystgr4jice_simulate_dev(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double sca_init, double s_init, double r_init, const double *params,
    int64_t N, double *qsim, double *G, double *eTG, double *s_store,
    double *r_store, double *sca, double *icemelt, double *snowmelt,
    int64_t ld, const double *qobs, double *sse, void *workspace,
    size_t workspace_bytes, void *stream);
int rr_cemaneigehystgr4jice_simulate(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double sca_init, double s_init, double r_init, const double *params,
    int64_t N, double *qsim, double *G, double *eTG, double *s_store,
    double *r_store, double *sca, double *icemelt, double *snowmelt,
    const double *qobs, double *sse);
/* the same with per-call options (NULL: none) */
int rr_cemaneigehystgr4jice_simulate_opt(
    const double *prec, const double *mean_temp, const double *etp,
    const double *frac_ice, const double *frac_solid_prec, int64_t T,
    int64_t L, double snow_pack_init, double thermal_state_init,
    double sca_init, double s_init, double r_init, const double *params,
    int64_t N, double *qsim, double *G, double *eTG, double *s_store,
    double *r_store, double *sca, double *icemelt, double *snowmelt,
    const double *qobs, double *sse,
    const rr_call_options *opt);

#ifdef __cplusplus
}
#endif
#endif /* RRHIP_H */
