/* phoskin.h -- C ABI of libphoskin_hip.so: the MI355X (gfx950) batched stiff-ODE engine for
 * PhosKinTime's per-protein parameter-estimation hot path.
 *
 * The reference (bibymaths/phoskintime) has no FFI: the seam is a set of plain Python callables
 * (SURVEY.md section 8b).  Every entry point below names the reference callable(s) it replaces.
 * Signatures use only plain pointers and sizes (no torch / numpy types).  Unless a function name
 * ends in `_host`, every array pointer is a DEVICE pointer (HBM) and the call is asynchronous on
 * the context's stream; `_host` variants take host pointers, stage through HBM and synchronise.
 *
 * Conventions
 *   model      0 = distmod (models/distmod.py), 1 = succmod (models/succmod.py), 2 = randmod (models/randmod.py)
 *   state      y = [R, P, X_1..X_m], m = n_sites (dist/succ) or 2^n_sites - 1 (rand); S = 2 + m.  Up to 64 states a replica is a lane
 *              group of a wavefront; beyond that (distmod / succmod n_sites 63..1276, randmod n_sites 7..20) one workgroup owns a
 *              replica: distmod / succmod keep the default LRP12 with exact structured solves; randmod n_sites = 7 keeps LRP12 too, the
 *              128 x 128 inverse spread over the registers of the workgroup (csrc/pk_rand_dense.hpp); randmod n_sites >= 8 integrates with
 *              the order-4 additive method ARK4(3)6L[2]SA on the n-cube at 0.02 x the requested tolerances (csrc/pk_wide.hpp)
 *   theta      [A, B, C, D, S_1..S_n, D_1..D_m], P = 4 + n + m      (reference unpack_params, distmod.py:68-91,
 *              succmod.py:94-112, randmod.py:88-119); batched as a row-major [B, P] f64 matrix
 *   return     0 = ok; < 0 = argument / runtime error (see pk_last_error); never throws or aborts.
 *   status[b]  per-replica bit flags: PK_ST_NONFINITE | PK_ST_MAXSTEPS | PK_ST_HMIN.  A flagged replica's
 *              remaining output rows are NaN (the reference only warns: odeint returns garbage and callers
 *              test np.isfinite, e.g. global_model/optproblem.py:125-133).
 */
#ifndef PHOSKIN_H
#define PHOSKIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PK_VERSION 200

enum { PK_MODEL_DIST = 0, PK_MODEL_SUCC = 1, PK_MODEL_RAND = 2 };

/* Integrators (all adaptive ones: max-norm local error control, steps land exactly on every t[k]).
 * LRP12 is the default: the three per-protein models are affine in y (constant Jacobian), which the LRP methods exploit. */
enum {
  PK_METHOD_RODAS4 = 0, /* 6-stage L-stable Rosenbrock 4(3) (Hairer-Wanner RODAS), analytic Jacobian, 1 factorisation / step */
  PK_METHOD_BDF2   = 1, /* variable-step BDF2 (BDF1 start), analytic Jacobian, 1 factorisation / step (BASELINE config 3) */
  PK_METHOD_RK4    = 2, /* classical explicit RK4, fixed step h <= rk4_h (BASELINE config 2); stability-bound when stiff */
  PK_METHOD_LRP8   = 3, /* L-stable restricted-Pade one-step method for affine systems: 8 resolvent solves, 1 rhs and
                           1 factorisation per step, order 7 with an embedded order-6 estimate */
  PK_METHOD_LRP12  = 5, /* the same construction with 12 solves: order 11, embedded order 10, gamma = 0.16 (L-stable; |R(iy)| <= 1 + 4.5e-9);
                           about half the steps of LRP8 at equal accuracy.  Default. */
  PK_METHOD_DP5    = 4, /* pk_network_simulate_batch only: the reference's opt-in explicit integrator, step for step -- Dormand-Prince
                           5(4), PI controller, dt in [1e-6, 1], bucket-edge landing, Hermite output (global_model/solvers.py:293-758) */
  PK_METHOD_ARK436 = 6, /* pk_network_simulate_batch only: ARK4(3)6L[2]SA (Kennedy-Carpenter) as a linearly implicit additive method, implicit
                           on the per-protein block Jacobian: order 4 for any Jacobian approximation, 5 block solves per step.  The network
                           default wherever the one-thread-per-protein layout applies (topologies 0 / 1 / 4, <= 8 sites, N <= 256); on
                           request also for the combinatorial topology (<= 3 sites), where the order-3 kernel is the faster one     */
  PK_METHOD_ROS34PW2 = 7 /* pk_network_simulate_batch only: the round-1 integrator, Rosenbrock-W of order 3 (4 block solves per step); every
                           network / topology; the default where ARK436 does not apply                                                       */
};

/* Linear solver for the implicit stage equations (g I - J) x = r. */
enum {
  PK_LINSOLVE_AUTO       = 0, /* fastest available: resolvent methods (LRP12 / LRP8 / RODAS4) -> the throughput kernels (distmod: 4-16 lanes per
                                 replica; randmod: bit-mask rows; small systems at large B: one lane per replica); else structured / dense */
  PK_LINSOLVE_DENSE      = 1, /* dense: in-register Gauss-Jordan to the explicit inverse, one matrix row per lane; every solve is a mat-vec (any model) */
  PK_LINSOLVE_STRUCTURED = 2  /* arrow (distmod) / tridiagonal (succmod) elimination; randmod falls back to dense */
};

/* Scalar Morris outputs, sensitivity/analysis.py:90-176 (_compute_Y; Y_METRIC, config/constants.py:104). */
enum {
  PK_METRIC_TOTAL_SIGNAL = 0, PK_METRIC_MEAN_ACTIVITY = 1, PK_METRIC_VARIANCE = 2,
  PK_METRIC_DYNAMICS = 3, PK_METRIC_L2_NORM = 4
};

enum { PK_ST_OK = 0, PK_ST_NONFINITE = 1, PK_ST_MAXSTEPS = 2, PK_ST_HMIN = 4 };

enum {
  PK_OK = 0, PK_ERR_ARG = -1, PK_ERR_UNSUPPORTED = -2, PK_ERR_HIP = -3, PK_ERR_NOMEM = -4
};

typedef struct pk_solver_opts {
  int32_t method;       /* PK_METHOD_*                                                        */
  int32_t linsolve;     /* PK_LINSOLVE_*                                                      */
  double  rtol, atol;   /* local error tolerances (max-norm); defaults 1e-6 / 1e-8 (chosen for LRP12: worst error 0.05 of the
                           parity band; RODAS4 / LRP8 / BDF2 need 1e-7 / 1e-9 for the same margin) */
  double  h0;           /* first step; 0 = automatic                                          */
  double  rk4_h;        /* RK4 only: largest fixed step (each output interval is split evenly) */
  int32_t max_steps;    /* per replica, accepted + rejected; default 100000                   */
  int32_t clip_nonneg;  /* np.clip(sol, 0, None) as in distmod.py:112-113 (default 1)         */
  int32_t normalize;    /* NORMALIZE_MODEL_OUTPUT: sol *= 1 / y0 (distmod.py:116-122)         */
  int32_t stage_form;   /* RODAS4: 0 = resolvent form (1 rhs + 6 solves / step; exact for the affine per-protein models),
                           1 = classical 6-stage Rosenbrock form (same method; kept for nonlinear right-hand sides) */
  int32_t kernel;       /* PK_KERNEL_*: which kernel family runs a small system.  AUTO picks by batch size (thread-per-replica above a
                           measured threshold), so the SAME replica can take different roundings / step sequences in batches of
                           different size -- a sharded run that must reproduce the single-GPU bits pins GROUP or TPR on every rank. */
  int32_t err_norm;     /* PK_NORM_*: how the local error of a step is measured against rtol / atol (pk_network_simulate_batch; the per-protein
                           kernels always use the max norm)                                                                          */
} pk_solver_opts;

enum {
  PK_NORM_DEFAULT = 0,  /* = PK_NORM_MAX                                                                                              */
  PK_NORM_MAX     = 1,  /* max_i |e_i| / (atol + rtol |y_i|): every component inside its tolerance (1.4-1.7x the steps of RMS at S ~ 550)   */
  PK_NORM_RMS     = 2   /* sqrt(mean_i (e_i / (atol + rtol |y_i|))^2): ODEPACK's vnorm, i.e. what the reference's LSODA controls with the
                           same rtol / atol (scipy.integrate.odeint, global_model/simulate.py:69-79).  Opt-in: 1.4-1.7x fewer steps, accuracy
                           of the reference run's own class on the fixtures but outside the parity band on some populations   */
};

enum {
  PK_KERNEL_AUTO  = 0,  /* by batch size (default)                                                             */
  PK_KERNEL_GROUP = 1,  /* lane-group kernels (several lanes per replica) at every batch size                  */
  PK_KERNEL_TPR   = 2   /* thread-per-replica kernels wherever one exists for (model, n_sites), else lane-group */
};

typedef struct pk_ctx pk_ctx;

int         pk_version(void);
/* One context per GPU / per thread.  Owns a HIP stream, two grow-only HBM arenas (staging of the `_host` entry points; per-replica
 * scratch of the largest random-model systems) and a page-locked host buffer for small `_host` calls -- no hipMalloc / hipFree per call. */
pk_ctx*     pk_create(int device_id);          /* NULL on failure: reason in pk_create_error() */
const char* pk_create_error(void);          /* thread-local; empty string after a successful pk_create */
void        pk_destroy(pk_ctx*);
const char* pk_last_error(pk_ctx*);            /* valid until the next call on this context */
/* Launch on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream; NULL is HIP's default
 * stream).  pk_use_own_stream switches back to the context's private non-blocking stream. */
int         pk_set_stream(pk_ctx*, void* hip_stream);
int         pk_use_own_stream(pk_ctx*);
int         pk_synchronize(pk_ctx*);
void        pk_default_opts(pk_solver_opts*);
/* {staging allocations so far, staging bytes, scratch allocations, scratch bytes, page-locked allocations, page-locked bytes}: serial
 * callers (the reference calls solve_ode once per parameter vector: paramest/normest.py:55, paramest/core.py:111,141,154) allocate once. */
int         pk_workspace_stats(pk_ctx*, int64_t out[6]);

/* Shapes (pure host arithmetic, usable without a GPU). */
int pk_protein_n_states(int model, int n_sites);           /* S, or PK_ERR_* */
int pk_protein_n_params(int model, int n_sites);           /* P */
int pk_protein_flat_len(int model, int n_sites, int T);    /* (T-5) + T + n_sites*T  (distmod.py:125-134) */

/* (state conventions, [r3]: randmod n_sites = 7, 8 integrate with the default LRP12 on EXACT solves -- the odd-popcount block of I - q J is
 *  diagonal, its even Schur complement is inverted in the registers of one workgroup, csrc/pk_rand_parity.hpp; n_sites >= 9: the additive
 *  method on the n-cube, csrc/pk_wide.hpp.)
 *
 * Replaces, for a whole batch of parameter vectors, models.solve_ode(params, init_cond, num_psites, t)
 *   -> (sol, flat)   [models/__init__.py:12; distmod.py:93-134, succmod.py:114-152, randmod.py:249-305]
 * and, fused behind it, sensitivity.analysis._compute_Y (sensitivity/analysis.py:90-176).
 *   theta [B,P]; y0 [S] (shared) or [B,S]; t [T] with t[0] the initial time, strictly increasing;
 *   sol [B,T,S] | NULL; flat [B,F] | NULL; metric [B] | NULL; status [B] | NULL; n_steps [B,2] | NULL
 *   (accepted, rejected).  sol is clipped / normalised per opts exactly like the reference's return value. */
int pk_solve_protein_batch(pk_ctx*, int model, int n_sites, int64_t B,
                           const double* theta, const double* y0, int y0_is_batched,
                           const double* t, int T, const pk_solver_opts* opts,
                           double* sol, double* flat, double* metric, int metric_id,
                           int32_t* status, int32_t* n_steps);

/* flat(theta) AND its parameter Jacobian d flat / d theta from ONE integration (forward sensitivities: S' = J S + df/dtheta solved with the
 * factors of the state solve, differentiated stage by stage: csrc/pk_sens.hpp).  Replaces the 1 + P calls of models.solve_ode per Jacobian
 * that scipy.optimize.curve_fit's '2-point' differences make under paramest/normest.py:167-326 and paramest/toggle.py.
 *   flat [B,F]; dflat [B,F,P] (row-major: the P derivatives of one flat entry are contiguous); status / n_steps as above.
 * The derivative follows flat's own post-processing: 0 where the value was clipped at 0, scaled by 1 / y0 under opts->normalize.
 * Tangents are held to the same rtol / atol as the states (maximum norm over all columns).  Method LRP12 only.
 * Sizes: pk_protein_sens_available(model, n_sites) != 0 -- distmod / succmod n_sites <= 62 (up to 14: one column per lane, csrc/pk_sens.hpp;
 * beyond: rows across the lanes, eight columns per lane, chunked columns, csrc/pk_sens_rows.hpp), randmod n_sites <= 7 (6 and 7: the
 * parity-eliminated inverse in registers serving eight columns per workgroup, csrc/pk_rand_sens.hpp); PK_ERR_UNSUPPORTED beyond (callers
 * difference pk_solve_protein_batch there, as phoskintime_amd.paramest.fit_rows_batch does).  Kernels that cut the columns of a replica
 * into chunks integrate the state once per chunk with its own step-size control: every column is within rtol / atol of the exact
 * derivative, the chunks are not bit-coupled; `status` is the OR over the chunks. */
int pk_protein_sens_available(int model, int n_sites);
int pk_solve_protein_sens_batch(pk_ctx*, int model, int n_sites, int64_t B,
                                const double* theta, const double* y0, int y0_is_batched,
                                const double* t, int T, const pk_solver_opts* opts,
                                double* flat, double* dflat, int32_t* status, int32_t* n_steps);

/* Replaces models.{distmod,succmod}.ode_core / models.randmod.ode_system (distmod.py:7-65, succmod.py:9-90,
 * randmod.py:122-247) evaluated for a batch: y [B,S] -> dydt [B,S]. */
int pk_rhs_protein_batch(pk_ctx*, int model, int n_sites, int64_t B,
                         const double* theta, const double* y, double* dydt);

/* Analytic Jacobian d f_i / d y_j, row-major J [B,S,S] (the reference has none for this path: LSODA
 * finite-differences ode_core; row-major as global_model/simulate.py:75 col_deriv=False). */
int pk_jacobian_protein_batch(pk_ctx*, int model, int n_sites, int64_t B,
                              const double* theta, double* J);

/* Steady state y* of dy/dt = J(theta) y + b(theta) = 0 for B parameter vectors: y_ss [B,S] (device), status [B] (or NULL;
 * PK_ST_NONFINITE + a NaN row when J is singular, e.g. no degradation).  Generalises steady.initial_condition(num_psites)
 * (steady/initdist.py:9-50, initsucc.py:9-55, initrand.py:9-77: SLSQP on the steady-state equations with every rate fixed to 1,
 * called once per protein at paramest/core.py:83) to per-replica theta.  Beyond 64 states one workgroup owns a replica: closed form (distmod),
 * cyclic reduction (succmod), symmetric Gauss-Seidel on the n-cube (randmod, n_sites <= 12; PK_ST_MAXSTEPS + NaN row if it does not converge). */
int pk_steady_state_protein_batch(pk_ctx*, int model, int n_sites, int64_t B, const double* theta, double* y_ss, int32_t* status);

/* Morris screening without a host round trip (reference: SALib morris.sample / morris.analyze as called at
 * sensitivity/analysis.py:221-265 and global_model/sensitivity.py:210-277; SALib itself is absent from the reference tree).
 * The N x D random draws -- base[r,i] on the level grid of the unit cube, sign[r,i] = +-1, rank[r,i] = position of coordinate i in
 * trajectory r's random order -- come from the host (KB); the N (D + 1) x D sample matrix is built in HBM:
 *   X[(r (D+1) + s), i] = lb_i + clip(base + sign * delta * [rank < s], 0, 1) * (ub_i - lb_i),       delta = p / (2 (p - 1)).
 * rank[r, :] must be a permutation of 0..D-1 (not checked on the device).  All pointers are device pointers. */
int pk_morris_build_batch(pk_ctx*, int64_t N, int D, double delta, const double* base, const double* sign, const int32_t* rank,
                          const double* lb, const double* ub, double* X);
/* Elementary effects EE [N,D] from the per-row outputs Y [N (D+1)] of the same design: (Y[r, rank+1] - Y[r, rank]) / (sign * delta). */
int pk_morris_effects_batch(pk_ctx*, int64_t N, int D, double delta, const double* sign, const int32_t* rank, const double* Y, double* EE);

/* Replaces config.config.score_fit(params, target, prediction, alpha, beta, gamma, delta, mu) (config/config.py:176-226) for B
 * candidates: theta [B,P], target [N] (shared), pred [B,N] (e.g. the `flat` output) -> out [B].  weights = {alpha (rmse), beta (mae),
 * gamma (var), delta (mse), mu (l2)} as a HOST pointer, NULL = all 1 (config/constants.py:77-83). */
int pk_score_fit_batch(pk_ctx*, int64_t B, const double* theta, int P, const double* target, const double* pred, int N,
                       const double* weights, double* out);

/* Host-pointer conveniences (stage through HBM, synchronise before returning). */
int pk_solve_protein_batch_host(pk_ctx*, int model, int n_sites, int64_t B,
                                const double* theta, const double* y0, int y0_is_batched,
                                const double* t, int T, const pk_solver_opts* opts,
                                double* sol, double* flat, double* metric, int metric_id,
                                int32_t* status, int32_t* n_steps);
int pk_solve_protein_sens_batch_host(pk_ctx*, int model, int n_sites, int64_t B,
                                     const double* theta, const double* y0, int y0_is_batched,
                                     const double* t, int T, const pk_solver_opts* opts,
                                     double* flat, double* dflat, int32_t* status, int32_t* n_steps);
int pk_rhs_protein_batch_host(pk_ctx*, int model, int n_sites, int64_t B,
                              const double* theta, const double* y, double* dydt);
int pk_jacobian_protein_batch_host(pk_ctx*, int model, int n_sites, int64_t B,
                                   const double* theta, double* J);

/* ------------------------------------------------------------------------------------------------------------------
 * Network (global_model) path -- SURVEY.md section 8 rows a9-a16, a21, a23.
 * pk_network_desc mirrors what global_model.network.System.odeint_args() packs (network.py:443-526) minus the per-candidate
 * parameters; all pointers are HOST pointers and are copied to HBM by pk_network_create.
 * A candidate is one row of x [B, n_var], n_var = n_K + 5 N + total_sites + 1, laid out as the optimiser's decision vector
 * (global_model/params.py:60-96):  [c_k | A_i | B_i | C_i | D_i | Dp_i | E_i | tf_scale];  x_is_raw != 0 applies the softplus of
 * params.unpack_params (params.py:106-132, utils.py:229-241) on load.
 * State layout: per protein i at offset_y[i]: models 0/1/4 [R, P, site_1..site_ns]; model 2 [R, state_0 .. state_{2^ns - 1}]. */
typedef struct pk_network_desc {
  int32_t model;                 /* 0 distributive, 1 sequential, 2 combinatorial, 4 saturating (global_model/config.py:59-61) */
  int32_t N, n_K, total_sites, n_grid;
  const int32_t *offset_y, *offset_s, *n_sites;                        /* [N] */
  const int32_t *W_indptr, *W_indices; const double* W_data;           /* CSR, total_sites x n_K  (kinase -> site) */
  const int32_t *TF_indptr, *TF_indices; const double* TF_data;        /* CSR, N x N              (TF -> target)   */
  const double*  tf_deg;                                               /* [N] */
  const int32_t* driver_map;                                           /* [N] kinase index driving protein i, or -1 */
  const double*  kin_grid;                                             /* [n_grid] bucket edges of the kinase input */
  const double*  kin_Kmat;                                             /* [n_K, n_grid] row-major */
} pk_network_desc;
typedef struct pk_net pk_net;

pk_net* pk_network_create(pk_ctx*, const pk_network_desc*);            /* NULL on error: see pk_last_error */
void    pk_network_destroy(pk_net*);
int     pk_network_n_states(const pk_net*);
int     pk_network_n_var(const pk_net*);

/* Replaces global_model.jacspeedup.rhs_odeint(y, t, *args) (jacspeedup.py:392-394 -> rhs_nb_* :176-388) for B candidates:
 * x [B,n_var]; y [S] or [B,S]; t [1] or [B] (device pointers); dydt [B,S]. */
int pk_network_rhs_batch(pk_ctx*, pk_net*, int64_t B, const double* x, int x_is_raw, const double* y, int y_is_batched,
                         const double* t, int t_is_batched, double* dydt);
/* Analytic Jacobian, row-major [B,S,S]; stands where the reference uses the finite-difference fd_jacobian_odeint
 * (jacspeedup.py:398-588) as odeint's Dfun. */
int pk_network_jacobian_batch(pk_ctx*, pk_net*, int64_t B, const double* x, int x_is_raw, const double* y, int y_is_batched,
                              const double* t, int t_is_batched, double* J);
/* Replaces global_model.simulate.simulate_odeint(sys, t_eval, rtol, atol, mxstep) -> Y[T,S] (simulate.py:34-80) for B candidates
 * of one network: x [B,n_var] and y0 ([S] or [B,S]) are device pointers, t is a HOST pointer to T strictly increasing times
 * (t[0] = initial time), Y [B,T,S] device.  Integrator: linearly implicit with the per-protein diagonal blocks of the analytic Jacobian
 * (DESIGN.md): PK_METHOD_ARK436 (order 4) where its kernel applies, else PK_METHOD_ROS34PW2 (order 3); either can be requested by name,
 * every other opts->method value means "the default"; PK_METHOD_DP5 selects the reference's explicit RK45 (jacspeedup.solve_custom,
 * jacspeedup.py:31-64; h0 = dt_init, 0 -> 0.05; max_steps <= 0 -> 2 000 000).  opts->rtol / atol / h0 / max_steps / err_norm are honoured.
 * All four topologies; combinatorial blocks (2) up to 3 sites per protein run out of registers, larger ones (<= 16 sites) in the LDS kernel. */
int pk_network_simulate_batch(pk_ctx*, pk_net*, int64_t B, const double* x, int x_is_raw, const double* y0, int y0_is_batched,
                              const double* t_host, int T, const pk_solver_opts* opts, double* Y, int32_t* status, int32_t* n_steps);
/* The integrator pk_network_simulate_batch will run for `opts` (NULL = defaults) on this network: PK_METHOD_DP5, PK_METHOD_ARK436 or
 * PK_METHOD_ROS34PW2; PK_ERR_UNSUPPORTED when ARK436 was requested and its kernel does not fit (N, sites per protein, 160 KB of LDS).  Pure
 * host arithmetic.  Host layers derive integrator-dependent defaults (tolerances) from this answer instead of re-stating the rule. */
int pk_network_resolve_method(const pk_net*, const pk_solver_opts* opts);
/* global_model.params.unpack_params (softplus of the raw decision vectors): x_raw [B,n_var] -> x_phys [B,n_var]. */
int pk_network_unpack_batch(pk_ctx*, pk_net*, int64_t B, const double* x_raw, double* x_phys);

/* Discrete Frechet distance between observed and predicted curves for B candidates x n_series series -- replaces
 * frechet.distance.frechet_distance (frechet/distance.py:9-56) inside the Pareto pick loop of global_model/runner.py:780-841.
 * Series s compares the observed points (obs_t, obs_v)[obs_ptr[s] .. obs_ptr[s+1]) with the predicted points
 * (pred_t[k], pred[b, pred_idx[k]]) for k in [pred_ptr[s], pred_ptr[s+1]); points are 2-D (time, value), both lists sorted by time by the
 * caller; pred [B, n_obs] is e.g. the output of pk_network_observables_batch.  max_points = longest curve (<= 32; checked here because
 * the kernel keeps the DP row in registers).  out [B, n_series]; a series with an empty side gives 0.  All pointers device pointers. */
int pk_frechet_batch(pk_ctx*, int64_t B, int n_series, const int32_t* obs_ptr, const double* obs_t, const double* obs_v,
                     const int32_t* pred_ptr, const double* pred_t, const int32_t* pred_idx, const double* pred, int n_obs,
                     int max_points, double* out);

/* Fused objective of one optimiser candidate after the simulation -- replaces global_model.lossfn.LOSS_FN (lossfn.py:114-382; all eight
 * LOSS_MODEs) and the assembly in GlobalODE_MOO._evaluate (optproblem.py:99-160).  pk_loss_data mirrors cache.prepare_fast_loss_data's
 * arrays (HOST pointers, copied to HBM and index-checked by pk_network_loss_create for a time grid of T points).
 *   Y [B,T,S] device; x [B,n_var] + defaults [n_var] (device; both optional: the prior penalty on A,B,C,D,E);
 *   lambdas = {protein, rna, phospho, prior} (host); status [B] from the simulation (optional: flagged => fail_value);
 *   loss_sums [B,3] raw weighted sums (LOSS_FN's return value) and / or F [B,3] the three objectives. */
typedef struct pk_loss_data {
  int32_t n_prot, n_rna, n_pho;
  const int32_t *p_prot, *t_prot; const double *obs_prot, *w_prot;
  const int32_t *p_rna, *t_rna;   const double *obs_rna, *w_rna;
  const int32_t *p_pho, *s_pho, *t_pho; const double *obs_pho, *w_pho;
  int32_t prot_base_idx, rna_base_idx, pho_base_idx;
} pk_loss_data;
typedef struct pk_loss pk_loss;
pk_loss* pk_network_loss_create(pk_ctx*, pk_net*, const pk_loss_data*, int T);
void     pk_network_loss_destroy(pk_loss*);
int      pk_network_objective_batch(pk_ctx*, pk_net*, pk_loss*, int64_t B, const double* Y, int T, int loss_mode,
                                    const double* x, int x_is_raw, const double* defaults, const double* lambdas, double fail_value,
                                    const int32_t* status, double* loss_sums, double* F);

/* [r3] simulate + objective in ONE launch (SURVEY fused op (i): optproblem.py:99-160 calls simulate_odeint -> LOSS_FN -> prior per candidate):
 * the additive integrator scores the loss handle's observations at its output times out of registers and writes F [B,3] / loss_sums [B,3];
 * the trajectory Y [B,T,S] is written only if Y != NULL.  Results equal pk_network_simulate_batch followed by pk_network_objective_batch
 * up to the order of the sums.  Takes: topologies 0 / 4 on the default integrator (PK_METHOD_LRP12 = "default" or PK_METHOD_ARK436), loss
 * data whose three baselines are time index 0 and that observe no (state, time) twice.  Otherwise PK_ERR_UNSUPPORTED (the message says
 * which): call the two functions above instead.  x, y0, defaults, Y, status, n_steps, loss_sums, F: DEVICE pointers; t, lambdas: HOST. */
int      pk_network_simulate_objective_batch(pk_ctx*, pk_net*, pk_loss*, int64_t B, const double* x, int x_is_raw, const double* y0,
                                             int y0_is_batched, const double* t_host, int T, const pk_solver_opts* opts, int loss_mode,
                                             const double* defaults, const double* lambdas, double fail_value, double* Y, int32_t* status,
                                             int32_t* n_steps, double* loss_sums, double* F);

/* global_model.lossfn.LOSS_FN with its own positional argument list (lossfn.py:114-121; :386 picks loss_function_comb when MODEL == 2):
 *   LOSS_FN(Y, p_prot, t_prot, obs_prot, w_prot, p_rna, t_rna, obs_rna, w_rna, p_pho, s_pho, t_pho, obs_pho, w_pho, prot_map,
 *           prot_base_idx, rna_base_idx, pho_base_idx) -> (loss_p, loss_r, loss_ph)          [called at optproblem.py:137-145]
 * for B trajectories and WITHOUT a network handle: the state offsets come from prot_map [N,2] int32 = (block start, n_sites) -- for the
 * combinatorial topology (block start, n_states = 2^n_sites).  HOST pointers throughout (the reference passes numpy arrays): indices are
 * checked on the host, arrays staged to HBM, one launch, synchronised.  Y [B,T,S]; loss_sums [B,3]; loss_mode = LOSS_MODE 0..7. */
int pk_loss_fn_batch_host(pk_ctx*, int combinatorial, int loss_mode, int64_t B, const double* Y, int T, int S, const pk_loss_data*,
                          const int32_t* prot_map, int N, double* loss_sums);

/* Array form of the pred_fc columns of global_model.simulate.simulate_and_measure (simulate.py:119-202): fold changes for the index
 * lists of a pk_loss (protein | rna | phospho, obs / w ignored), floor eps (the reference uses 1e-12 there): pred [B, n_prot+n_rna+n_pho]. */
int pk_network_observables_batch(pk_ctx*, pk_net*, pk_loss*, int64_t B, const double* Y, int T, double eps, double* pred);

/* Timing hook for bench.py: runs `iters` back-to-back launches of pk_solve_protein_batch on the context's
 * stream between two hipEvents and returns the mean kernel time per launch in milliseconds (< 0 on error). */
double pk_time_solve_protein_batch(pk_ctx*, int iters, int model, int n_sites, int64_t B,
                                   const double* theta, const double* y0, int y0_is_batched,
                                   const double* t, int T, const pk_solver_opts* opts,
                                   double* sol, double* flat, double* metric, int metric_id,
                                   int32_t* status, int32_t* n_steps);

/* Machine peaks measured on the context's GPU, for bench.py's roofline fractions (the spec-sheet values are printed beside them):
 * sustained HBM copy rate in GB/s (read + written bytes; `bytes` >= 1 MiB per buffer, choose it well beyond the 256 MiB Infinity Cache)
 * and sustained FP64 vector FMA rate in TFLOP/s.  Both allocate and free their own buffers; < 0 on error. */
double pk_measure_hbm_gbs(pk_ctx*, int64_t bytes, int iters);
/* one-directional rates, bytes moved per second: mode 0 = read-only, 1 = write-only (a copy pays the bus turnarounds between the two) */
double pk_measure_hbm_stream_gbs(pk_ctx*, int64_t bytes, int iters, int mode);
double pk_measure_fp64_fma_tflops(pk_ctx*, int iters);

/* ---- multi-GPU for an embedder without torch.distributed: one process per GPU, each rank integrates ITS rows with the
 * batch entry points above and gathers the per-row results with ONE all-gather (RCCL over xGMI, on the context's stream) -- candidates /
 * replicas never move.  Rank 0 makes an id (pk_comm_unique_id), the embedder ships its PK_COMM_ID_BYTES to every rank by its own means
 * (MPI, a file, a socket), every rank calls pk_comm_init(ctx, id, rank, world); pk_allgather_f64(ctx, send, count, recv): send [count],
 * recv [world * count] DEVICE pointers, rank r's block at recv + r * count.  Interleave rows as phoskintime_amd/distributed.py does
 * (rank = position mod world in decreasing order of a cost proxy) to balance step counts.  RCCL is bound at run time: PK_ERR_UNSUPPORTED
 * when librccl cannot be loaded.  pk_destroy frees the communicator too. */
#define PK_COMM_ID_BYTES 128
int pk_comm_unique_id(pk_ctx*, char* id_out /* [PK_COMM_ID_BYTES] */);
int pk_comm_init(pk_ctx*, const char* id /* [PK_COMM_ID_BYTES] */, int rank, int world);
int pk_comm_rank(pk_ctx*);
int pk_comm_world(pk_ctx*);
int pk_allgather_f64(pk_ctx*, const double* send, int64_t count, double* recv);
int pk_comm_destroy(pk_ctx*);

#ifdef __cplusplus
}
#endif
#endif /* PHOSKIN_H */
