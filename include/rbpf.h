/*
 * rbpf.h -- C ABI of the MI355X-native Rao-Blackwellized particle filter / smoother.
 *
 * This is the drop-in boundary for the hot path of manonkok/Rao-Blackwellized-SLAM-smoothing.
 * The reference has no FFI layer of its own (it is pure MATLAB); the boundary *is* the three
 * MATLAB function signatures
 *     src/particleFilter.m:1-3, src/particleSmoother.m:1-2,
 *     src/particleSmootherInformationForm.m:1-2
 * and their callback contracts.  A MEX gateway (matlab/rbpf_mex.cpp) binds
 * exactly the entry points declared here; on machines without MATLAB the same ABI is driven by
 * the Python ctypes host mirror in rao-blackwellized-slam-smoothing_amd/.
 *
 * Conventions
 *   - All host matrices are MATLAB column-major fp64, caller-owned, never mutated.
 *   - Function handles cannot cross the boundary; the dynModel / measModel / dynResNorm closures
 *     of the example runners are described by an `rbpf_model` descriptor (family id + constants).
 *   - MATLAB's global RNG stream is replaced by an `rbpf_rng` block: replay buffers (seed-exact
 *     parity runs) or a (seed) pair for the device Philox4x32-10 generator (throughput runs).
 *   - Every function returns an rbpf_status; RBPF_OK == 0.  No global state; a context is bound
 *     to the HIP device that was current when it was created and owns one stream.
 *   - Particle / time indices in trace outputs are 0-based.
 */
#ifndef RBPF_H_
#define RBPF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBPF_ABI_VERSION 9   /* 9 (r05): rbpf_options starts with `struct_size` (checked by every entry point that takes options), lost
                              * `family_products` (r04's family GEMM was measured slower and removed), gained `info_rebuild`, and
                              * chol_refresh = 0 now means "automatic"; rbpf_probe_family_pht is gone, rbpf_chol_refresh_resolve is new */

typedef enum {
  RBPF_OK = 0,
  RBPF_ERR_INVALID_ARG = 1,
  RBPF_ERR_UNSUPPORTED = 2,     /* model family / option not implemented on the device path        */
  RBPF_ERR_HIP = 3,             /* a HIP runtime call failed (rbpf_last_error has the text)         */
  RBPF_ERR_NO_DEVICE = 4,       /* no gfx950 device visible: the product path has NO CPU fallback   */
  RBPF_ERR_OUT_OF_MEMORY = 5,
  RBPF_ERR_CHOL_FAILED = 6,     /* second Cholesky failure: MATLAB would throw                      *
                                 * (particleFilter.m:147, particleSmootherInformationForm.m:228-231) */
  RBPF_ERR_STATE = 7,           /* call sequence error (e.g. advance past N_T)                      */
  RBPF_ERR_CALLBACK = 8         /* a host callback (model handle / on_step hook) returned non-zero   */
} rbpf_status;

/* Model families = the closures defined in the reference's example runners. */
typedef enum {
  /* examples/slam-dense-mag/run_dense3D_magfield.m: dynModel :301-308, measModel :265-279,
   * dynResNorm :202-203.  nNonLin=7 (pos3+quat4), ny=3, nw=6, n_odo=7, nLin=m+3.             */
  RBPF_MODEL_DENSE_MAG_6D = 1,
  /* examples/slam-dense-radio/run_dense2D_withHeading.m: dynModel :75-76, dynResNorm :77,
   * measModel :168.  nNonLin=3 (x,y,heading), ny=1, nw=1, n_odo=3, nLin=m.                    */
  RBPF_MODEL_DENSE_RADIO_2DH = 2,
  /* examples/slam-sparse-visual: dynModel pfslam.m:81 (xn + dx' + sqrt(dt*Q)*randn, element-wise sqrt), measModel
   * pfslam.m:82 -> measurement.m:32-84 (1-D pinhole camera, point landmarks), dynResNorm = [] (psslam.m:119).
   * sparseFeatures = true: per-particle EKF linearisation, NaN in y = not observed (particleFilter.m:127-137,
   * 165-181).  nNonLin=3 (x,y,heading), ny = m_basis landmarks, nw=3, n_odo=3, nLin = 2*m_basis.              */
  RBPF_MODEL_SPARSE_VISUAL_2D = 3,
  /* Arbitrary dense model (any dynModel / measModel handles, sparseFeatures = false): the caller evaluates the handles
   * itself -- dynModel per particle (particleFilter.m:108), measModel on the whole batch (:124) -- and hands states and
   * Jacobians to rbpf_filter_step_external; everything else of the step (weights, normalisation, resampling, Kalman
   * update, ancestry) runs on the device.  n_y must be 1 or 3.  The slow path behind unrecognised handles.         */
  RBPF_MODEL_GENERIC_DENSE = 4
} rbpf_model_kind;

/* Host callbacks of the generic family = the reference's handle contracts, batched over columns so that a binding
 * makes one call per step (the MEX gateway: one mexCallMATLAB each).  All matrices column-major.
 * A non-zero return aborts the run with RBPF_ERR_CALLBACK.
 *   dyn_model    xn_new(:,j) = dynModel(xn_anc(:,j), odometry(t,:), dt(t), Q(:,:,t)) for j = 0..n_cols-1, in this order
 *                (the handle draws its own random numbers: particleFilter.m:108, particleSmoother.m:134-136); t is the
 *                0-based row of odometry (= MATLAB's t-1)
 *   meas_model   dy = measModel(xn [n_nonlin x n_cols]) -> [n_cols x n_y x n_lin] ([n_cols x n_lin] memory for
 *                n_y = 1), particleFilter.m:124, particleSmoother.m:120
 *   dyn_res_norm e_dyn(:,j) = dynResNorm(xnk_t, xn(:,j), odometry(t,:), dt(t), Q(:,:,t))' [n_w x n_cols]
 *                (particleSmoother.m:178-180); NULL = isempty(dynResNorm): the additive default (:175-177) on the device */
typedef struct {
  int (*dyn_model)(void* user, int32_t t, int32_t n_cols, const double* xn_anc, double* xn_new);
  int (*meas_model)(void* user, int32_t n_cols, const double* xn, double* dy);
  int (*dyn_res_norm)(void* user, int32_t t, int32_t n_cols, const double* xnk_t, const double* xn, double* e_dyn);
  void* user;
} rbpf_callbacks;

typedef struct {
  int32_t kind;          /* rbpf_model_kind                                                     */
  int32_t m_basis;       /* number of basis functions m (tools/domain_cartesian_dx.m:43)        */
  int32_t dim;           /* input dimension of the basis (3 for dense-mag, 2 for dense-radio)   */
  int32_t use_dyn_res_norm; /* 1: model's dynResNorm handle; 0: additive default                *
                             * (particleSmoother.m:175-177, isempty(dynResNorm))                */
  const int32_t* NN;     /* [m x dim] column-major index table NN (domain_cartesian_dx.m:36-43); NULL for   *
                          * RBPF_MODEL_SPARSE_VISUAL_2D                                                       */
  double L[3];           /* domain half-widths (domain_cartesian_dx.m:27-29)                    */
  double cam[3];         /* RBPF_MODEL_SPARSE_VISUAL_2D: f, fp, fw (load_data.m:58-60)          */
  const rbpf_callbacks* callbacks; /* RBPF_MODEL_GENERIC_DENSE: the handles (one-shot entry points and            *
                                    * rbpf_filter_advance call them every step); NULL: the caller drives the       *
                                    * filter itself with rbpf_filter_ancestors / rbpf_filter_step_external          */
} rbpf_model;

typedef struct {
  int32_t N_P;           /* particles                                                           */
  int32_t N_T;           /* time steps = size(y,1)                                              */
  int32_t n_nonlin;      /* size(x0_nonLin,1)                                                   */
  int32_t n_lin;         /* size(x0_lin,1)                                                      */
  int32_t n_y;           /* size(y,2)                                                           */
  int32_t n_w;           /* size(Q,1)                                                           */
  int32_t n_odo;         /* size(odometry,2)                                                    */
  int32_t x0_lin_cols;   /* 1 or N_P          (particleFilter.m:60-64)                          */
  int32_t q_pages;       /* size(Q,3): 1 or >= N_T-1   (particleFilter.m:75-77)                 */
  int32_t dt_len;        /* 1 or >= N_T-1              (particleFilter.m:80-82)                 */
  const double* odometry;/* [>=N_T-1 x n_odo], leading dimension odo_ld                         */
  int32_t odo_ld;
  const double* y;       /* [N_T x n_y]                                                         */
  const double* x0_nonlin;/* [n_nonlin]                                                         */
  const double* x0_lin;  /* [n_lin x x0_lin_cols]                                               */
  const double* P0_lin;  /* [n_lin x n_lin]                                                     */
  const double* Q;       /* [n_w x n_w x q_pages]                                               */
  const double* R;       /* [n_y x n_y]                                                         */
  const double* dt;      /* [dt_len]                                                            */
} rbpf_problem;

typedef enum { RBPF_RNG_REPLAY = 0, RBPF_RNG_PHILOX = 1 } rbpf_rng_mode;

typedef struct {
  int32_t mode;          /* rbpf_rng_mode                                                       */
  int32_t n_iter;        /* pages available in the replay buffers (1 for the filter, N_K)       */
  /* Replay buffers (RBPF_RNG_REPLAY), drawn by the caller in the reference's call order:
   *   U [N_P x (N_T-1) x n_iter]      the `rand` of tools/sample.m:31 for slot i at step t
   *                                   (for smoother iterations k>1 slot N_P-1 holds the single
   *                                   rand of particleSmoother.m:241)
   *   Z [n_w x N_P x (N_T-1) x n_iter] the randn's consumed by dynModel for slot i at step t
   *   Ufin [n_iter]                   the rand of `ak = sample(w)` (particleSmoother.m:346)    */
  const double* U;
  const double* Z;
  const double* Ufin;
  uint64_t seed;         /* RBPF_RNG_PHILOX: key of the counter-based device generator          */
} rbpf_rng;

typedef struct rbpf_ctx rbpf_ctx;      /* opaque filter / smoother context (device-resident state) */

/* Per-step hook = the reference's makePlots call sites: after every time step of the filter
 * (src/particleFilter.m:215-217: makePlots(xn,xl_max,P_max,traj_max,yhattraj,xn_traj,traj_mean,xl,P)) and after every
 * iteration of the smoothers (src/particleSmoother.m:360-362: makePlots(xnk,xlk,k,XNK,XLK,PK)).  Filter: `ctx` is
 * valid inside the callback for rbpf_filter_finish(ctx, &partial), which returns any of the outputs / final_* banks
 * as of step t.  Smoothers: iteration k's pages of the caller's XNK / XLK / PK buffers are complete when it runs.
 * A non-zero return aborts with RBPF_ERR_CALLBACK.                                                                */
typedef struct {
  rbpf_ctx* ctx;         /* filter only (NULL for the smoothers)                                */
  int32_t t;             /* filter: 0-based time step just finished; smoother: 0-based iteration */
  int32_t is_smoother;
} rbpf_view;
typedef int (*rbpf_on_step_fn)(const rbpf_view* view, void* user);

typedef struct {
  int32_t struct_size;   /* = sizeof(rbpf_options) of the caller's build (rbpf_abi_sizeof(3) at run time): an entry  *
                          * point handed options of another size returns RBPF_ERR_INVALID_ARG instead of reading past the caller's     *
                          * struct.  0 is accepted from callers that zero the struct and fill fields by name (C designated             *
                          * initialisers) -- they were compiled against this header by construction                                    */
  int32_t keep_history;  /* 1: keep xn history + ancestor table (needed for traj_sample_iwmax,  *
                          *    xn_traj and every smoother); 0: ping-pong only                   */
  int32_t trace;         /* 1: record per-step logw / w / ancestor indices (tests)              */
  int32_t fix_p_mean;    /* 0: reproduce quirk Q3 (particleFilter.m:228-230 overwrites P_mean);  *
                          * 1: accumulate it over the particles (the evident intent)            */
  int32_t lazy_depth;    /* filter and information-form smoother: C >= 2 keeps up to C pending rank-n_y downdates on the    *
                          * fly and rewrites the stored covariances every C-th step only (C-1 read-only steps in         *
                          * between); 0/1: rewrite every step.  Results agree to rounding (same algebra).  max 4 (filter; *
                          * 8 on symmetric storage, storage = 2) / 3 (information form, also in the sharded smoother); ignored  *
                          * by the covariance-form smoother                                                                  */
  double jitter;         /* <=0: reference default (1e-3 filter :89, 1e-2 smoothers :70)        */
  int32_t inplace;       /* filter (and, on request only, the information-form smoother) with lazy_depth >= 2: keep ONE covariance bank and rewrite it in place at    *
                          * every flush (the first child of a stored matrix overwrites it after its siblings    *
                          * were written to dead slots) instead of ping-pong banks -- halves the memory, same    *
                          * results bit for bit.  0: automatic (when two banks do not fit the device), 1: on,    *
                          * -1: off.  On block-lower storage (eight / sixteen tile rows) the single bank keeps   *
                          * the shared flush: one writer per parent with children, the first writer of a stored  *
                          * matrix overwrites it in place after its readers (results to rounding, as two banks). */
  int32_t storage;       /* dense-mag filter (also sharded): 0 = the covariance banks hold fp64 (the reference's precision); 1 = fp32     *
                          * STORAGE of the banks (BASELINE.json configs[4]): half the HBM traffic and memory, all     *
                          * arithmetic and every other state stay fp64.  Results then agree with the fp64 run to     *
                          * ~1e-6 relative per step (not to 1e-9) and resampling indices may differ.                 *
                          * 2 = fp64, SYMMETRIC storage: particleFilter.m:198 keeps P_i symmetric up to rounding, so only the   *
                          * lower block triangle (64 x 64 tiles) + the border rows are kept -- 0.5625 n^2 elements at nLin =   *
                          * 515, i.e. 0.56 x the HBM traffic and memory of storage 0; read-only steps of lazy_depth apply the  *
                          * pending sets as P H' - KS (K' H') instead of element-wise.  Same algebra: results within 1e-9 of  *
                          * storage 0 (P(r,c) and P(c,r), which differ by rounding in the reference's plain form, are one     *
                          * stored value).  Dense families with n_y = 3 and 515 <= n_lin <= 639 (BASELINE.json configs[2]) or      *
                          * 259 <= n_lin <= 383: filter and both smoothers, single-GPU and sharded; 1024 <= n_lin <= 1151       *
                          * (sixteen tile rows, BASELINE.json configs[4]'s basis size): the filter; dense-radio (n_y = 1) with   *
                          * n_lin = 128 (two tile rows: 0.75 x the bytes): filter and both smoothers; RBPF_ERR_UNSUPPORTED       *
                          * elsewhere.                                                                                          *
                          * 3 = fp32 tiles of the lower block triangle (storage 1's rounding on storage 2's layout: 0.28 x the   *
                          * bytes of storage 0): the filter with 515 <= n_lin <= 639 or 1027 <= n_lin <= 1151, lazy_depth <= 4.  */
  int32_t chol_variant;  /* smoothers: kernel of the ancestor-weight factorisation (particleSmoother.m:221,                   *
                          * particleSmootherInformationForm.m:228).  0: by matrix size (default); 16 / 64 / 648 / 644 / 1 /             *
                          * (a non-zero value with chol_refresh = 0 selects the from-scratch factorisation, chol_refresh = 1)           *
                          * 10-14 as the `variant` of rbpf_chol_weights.  Same arithmetic, results to rounding (tests).                */
  int32_t chol_refresh;  /* information-form smoother, the ancestor-weight factorisation chol(Imat_i + ImatAddt) of                    *
                          * particleSmootherInformationForm.m:224-236.                                                                 *
                          * K > 1: CARRY the factor along every lineage -- per step n_y rank-1 updates (the particle's own           *
                          * H' R^-1 H, :334) and n_y rank-1 downdates (the reference trajectory's term leaving ImatAddt, :194-201)    *
                          * of the ancestor's factor, O(n^2) instead of n^3 / 3, the forward solve carried as an augmented row --     *
                          * and refactorise every K-th step from Imat rebuilt along the state history (recognised families; the       *
                          * information matrices are then only materialised at those steps).  Same algebra, different arithmetic:     *
                          * every ancestor index and trajectory draw identical, ancestor probabilities within 2e-9 absolute, outputs *
                          * within 1e-9 of the from-scratch factorisation (measured 8.8e-10 over T = 3000 at nLin = 515;             *
                          * tests/test_gpu_chol_carry.py, test_gpu_r05_parity.py).  nLin <= 575, n_y = 1 or 3; also in *
                          * the sharded smoother (rbpf_shard_smoother_refresh_*).  K >= N_T - 1 never refactorises after the first    *
                          * step: no information matrix is stored (or, sharded, exchanged) at all -- see info_rebuild; 256 < K <       *
                          * N_T - 1 (single device only) rebuilds the matrices from the origin at every refresh likewise, since a        *
                          * window that long would need N_P x K x n_y x n_lin doubles of Jacobians.                                      *
                          * 1: factorise from scratch at every step, the reference's own arithmetic.                                  *
                          * 0 (default): AUTOMATIC = 32 where the carried factors apply and pay (recognised dense family, 128 <=      *
                          * nLin <= 575), 1 elsewhere; rbpf_chol_refresh_resolve tells which.                                        *
                          * Failure behaviour of K > 1: the reference's retry of a failed chol (:228-231) is unusable as written      *
                          * (quirk Q4), so a failed factorisation is an error either way; a carried DOWNDATE that loses definiteness  *
                          * sets status bit 2, the particle's ancestor log-weight becomes NaN and the run ends with                    *
                          * RBPF_ERR_CHOL_FAILED when the iteration's flags are checked.  Use chol_refresh = 1 where near-singular     *
                          * Imat + ImatAddt are expected.                                                                              */
  int32_t exchange_capacity; /* sharded sessions: particle records one rank can send / receive per time step (buffers are  *
                          * sized from it, identically on every rank; received records persist for lazy_depth steps).       *
                          * > 0: a hard limit -- a step that needs more fails on EVERY rank with RBPF_ERR_OUT_OF_MEMORY before   *
                          * any collective is issued (the plan is replicated).  0: start at min(N_local, max(256, N_local /    *
                          * 16)) and GROW on demand: every rank reaches the same verdict from the replicated plan and enlarges *
                          * its buffers by the same rule without communicating (rbpf_shard_views_get then returns the new      *
                          * pointers / capacities; received records of the running lazy cycle are kept).  < 0: start at        *
                          * |exchange_capacity| and grow likewise.                                                              */
  rbpf_on_step_fn on_step; /* NULL: no hook                                                       */
  void* on_step_user;
  int32_t n_devices;     /* rbpf_particle_filter / rbpf_particle_smoother(info_form = 1) only.  0 / 1 with device_ids == NULL: one  *
                          * GPU (the current device).  W > 1: the N_P particles are sharded over W GPUs of this node INSIDE the   *
                          * library -- one host thread per device runs the sharded step loop (rbpf_shard_* below) and the library  *
                          * issues the collectives itself over RCCL (ncclCommInitAll; all-gather of the forward bank + grouped     *
                          * send / recv of the migrating particle records, on each context's stream), so a single host process   *
                          * (a MATLAB session behind the MEX gateway: particleSmootherInformationForm.m is one call) reaches all  *
                          * GPUs.  N_P must be a multiple of W; recognised dense model families; results equal the single-GPU run *
                          * (bit for bit without lazy_depth, 1e-9 with it), every reference output incl. xn_traj; no traces / final banks. */
  const int32_t* device_ids; /* [n_devices] HIP device of every rank, NULL = 0 .. n_devices-1.  A device named more than once makes *
                          * its ranks share that GPU over a host-staged transport (no RCCL) -- how a one-GPU machine exercises the  *
                          * multi-rank loop; n_devices = 1 with device_ids set runs the loop with a world of one.                  */
  int32_t info_rebuild;  /* information-form smoother with carried factors (chol_refresh = K > 1), single device: 0 = the information    *
                          * matrices are materialised at every refresh (two banks of N_P matrices; a refresh walks back K generations).   *
                          * 1 = NONE is stored: every refresh rebuilds them from the initial matrix along the WHOLE ancestral path --    *
                          * Imat_i = Imat0 + sum of H' R^-1 H over the path, :334's terms in another order of summation -- chunk by chunk  *
                          * (4096 particles x 32 generations at a time).  2.3 MB per particle less at nLin = 515: with inplace = 1 the     *
                          * state is 3.6 MB per particle and the metric's N_P = 65 536 fits one 288 GB GPU.  The cost of a refresh grows   *
                          * with t: choose K in the hundreds (K >= N_T - 1 never refreshes and implies this mode).  Same tolerance as      *
                          * chol_refresh (tests/test_gpu_chol_carry.py, test_gpu_r05_parity.py).                                          */
} rbpf_options;

/* Outputs of particleFilter (src/particleFilter.m:1,26-34).  NULL pointers are skipped. */
typedef struct {
  double* traj_max;          /* [n_nonlin x N_T]                                                */
  double* traj_mean;         /* [n_nonlin x N_T]                                                */
  double* xl_max;            /* [n_lin]                                                         */
  double* xl_mean;           /* [n_lin]                                                         */
  double* P_max;             /* [n_lin x n_lin]                                                 */
  double* P_mean;            /* [n_lin x n_lin]                                                 */
  double* traj_sample_iwmax; /* [n_nonlin x N_T]   (needs keep_history)                         */
  double* xn_traj;           /* [n_nonlin x N_P x N_T] (needs keep_history)                     */
  /* extras (not reference outputs; used by the parity tests) */
  double* trace_logw;        /* [N_P x N_T]        (needs trace)                                */
  double* trace_w;           /* [N_P x N_T]        (needs trace)                                */
  int32_t* trace_ai;         /* [N_P x N_T] 0-based, column 0 unused (needs trace)              */
  double* final_xn;          /* [n_nonlin x N_P]                                                */
  double* final_xl;          /* [n_lin x N_P]                                                   */
  double* final_P;           /* [n_lin x n_lin x N_P]                                           */
  int32_t* iw_max;           /* [1] 0-based index of the maximum-weight particle at t = N_T     */
} rbpf_filter_out;

/* Outputs of particleSmoother / particleSmootherInformationForm (particleSmoother.m:1,27-30). */
typedef struct {
  double* XNK;               /* [n_nonlin x N_T x N_K]                                          */
  double* XLK;               /* [n_lin x N_K]                                                   */
  double* PK;                /* [n_lin x n_lin x N_K]                                           */
  /* extras for the parity tests (need trace) */
  double* trace_logw;        /* [N_P x N_T x N_K]                                               */
  double* trace_w;           /* [N_P x N_T x N_K]                                               */
  int32_t* trace_ai;         /* [N_P x N_T x N_K] 0-based                                       */
  double* trace_paNt;        /* [N_P x N_T x N_K] ancestor probabilities AI(:,t) (:240)         */
  int32_t* trace_ak;         /* [N_K] 0-based                                                   */
} rbpf_smoother_out;

/* Timing of the dominant kernel (the fused resample-gather + weight + Kalman-update stream kernel),
 * measured with HIP events on the context's own stream.                                           */
typedef struct {
  double stream_kernel_ms;   /* sum of launch durations since the last reset                     */
  int64_t stream_kernel_launches;
  double algorithmic_bytes_per_launch; /* N_P * (2 n^2 + 2 n + 2 nNonLin) * 8         */
  double scheduled_bytes_per_launch;   /* mean over the timed launches of the bytes this run's schedule has to move:      *
                                        * per particle one read of the stored covariance, one write of it only when the  *
                                        * launch rewrites it (every lazy_depth-th step), the pending rank-n_y factor sets *
                                        * it applies (2 n n_y each) and writes (one), the mean and the non-linear state   *
                                        * in and out.  <= algorithmic; the roofline fraction is quoted on this figure     */
} rbpf_timing;

/* ---- library ---------------------------------------------------------------------------------- */
int rbpf_abi_version(void);
/* sizeof of the library's own view of an interface struct -- a binding in another language (ctypes, MEX, a MATLAB loadlibrary
 * prototype file) compares its mirror against it once at load time.  which: 0 rbpf_model, 1 rbpf_problem, 2 rbpf_rng, 3 rbpf_options,
 * 4 rbpf_filter_out, 5 rbpf_smoother_out, 6 rbpf_timing, 7 rbpf_callbacks, 8 rbpf_view; -1 for any other value.               */
int rbpf_abi_sizeof(int32_t which);
const char* rbpf_status_string(int status);
/* Thread-local text of the last error raised on this thread (HIP error string, argument name). */
const char* rbpf_last_error(void);
/* Number of visible gfx950 devices (0 when none; never touches a device).                        */
int rbpf_device_count(void);

/* ---- one-shot entry points: what the MEX gateway binds ----------------------------------------- */
/* Replaces src/particleFilter.m:1-3 for the recognised model families (dense branch).             */
int rbpf_particle_filter(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng,
                         const rbpf_options* opt, rbpf_filter_out* out);
/* Replaces src/particleSmoother.m:1-2 (info_form = 0) and
 * src/particleSmootherInformationForm.m:1-2 (info_form = 1).                                      */
int rbpf_particle_smoother(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng,
                           const rbpf_options* opt, int32_t N_K, int32_t info_form,
                           rbpf_smoother_out* out);

/* ---- resident-state API (what bench.py times: inputs already in HBM) --------------------------- */
/* Uploads model, problem and RNG block to the current device and allocates the particle banks.    */
int rbpf_filter_create(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng,
                       const rbpf_options* opt, rbpf_ctx** ctx);
/* Bytes of device memory a context for this problem needs (no device access).                     */
int rbpf_filter_workspace_bytes(const rbpf_model* model, const rbpf_problem* prob,
                                const rbpf_options* opt, size_t* bytes);
/* Enqueue `n_steps` time steps (particleFilter.m:100-218) on the context stream; asynchronous.    */
int rbpf_filter_advance(rbpf_ctx* ctx, int32_t n_steps);
/* Rewind to t = 0 (re-initialises weights / states: particleFilter.m:52-67); asynchronous.        */
int rbpf_filter_reset(rbpf_ctx* ctx);
/* Block until the stream is idle; reports a deferred RBPF_ERR_CHOL_FAILED.                        */
int rbpf_sync(rbpf_ctx* ctx);
/* Final extraction (particleFilter.m:220-233) + download of the requested outputs.                */
int rbpf_filter_finish(rbpf_ctx* ctx, rbpf_filter_out* out);
/* Current step index (number of steps processed).                                                 */
int rbpf_filter_tell(const rbpf_ctx* ctx, int32_t* t);
/* The covariance-bank schedule the context chose (rbpf_options.inplace = 0 is automatic): banks = 1 (single bank
 * rewritten in place at every flush) or 2 (ping-pong banks); shared_flush = 1 when a flush step writes one matrix
 * per parent for all its children (two banks, block-lower storage).  What particleFilter.m:112-113's gather became. */
int rbpf_filter_schedule(const rbpf_ctx* ctx, int32_t* banks, int32_t* shared_flush);
/* Generic model family (RBPF_MODEL_GENERIC_DENSE), one time step at a time:
 *   t = 0:  rbpf_filter_step_external(ctx, xn0, measModel(xn0))
 *   t > 0:  rbpf_filter_ancestors(ctx, ai, xn_prev); xn(:,i) = dynModel(xn_prev(:,ai(i)+1), ...); 
 *           rbpf_filter_step_external(ctx, xn, measModel(xn))
 * ai [N_P] 0-based ancestors of the step about to run (drawn by the previous step), xn_prev [n_nonlin x N_P] the
 * states of the last step; xn_new [n_nonlin x N_P]; dy [N_P x n_y x n_lin] exactly as measModel returns it
 * ([N_P x n_lin] memory for n_y = 1).                                                                       */
int rbpf_filter_ancestors(rbpf_ctx* ctx, int32_t* ai, double* xn_prev);
int rbpf_filter_step_external(rbpf_ctx* ctx, const double* xn_new, const double* dy);
/* Enable (1) / disable (0) per-launch HIP-event timing of the stream kernel; read / reset it.     */
int rbpf_timing_enable(rbpf_ctx* ctx, int32_t on);
int rbpf_timing_read(rbpf_ctx* ctx, rbpf_timing* out, int32_t reset);
int rbpf_destroy(rbpf_ctx* ctx);

/* ---- particle-sharded filter (one process per GPU) ----------------------------------
 * The global filter has N = world * N_P logical slots.  A logical slot keeps its identity (RNG stream,
 * position in every output) but its particle may live on any rank: children are computed on the rank
 * that already holds their ancestor's map state ("owner computes") and only the load-imbalance excess
 * migrates.  Every rank always holds exactly N_P particles in physical slots 0..N_P-1.  The collectives
 * stay outside the library (torch.distributed over RCCL in multigpu.py); the library exposes the device
 * buffers and the kernels between them.  Per time step t >= 1 the host side does
 *   all_gather(fwd_local -> fwd_gather)                 log-weights + non-linear states, physical order
 *   rbpf_shard_normalise_search(ctx, perm, ai_host)     permute to logical order; global w / cumsum /
 *                                                       ancestors -- identical on every rank
 *   (host) placement of the new generation + exchange plan from the replicated ancestor vector
 *   rbpf_shard_pack(ctx, idx, count)                    ancestors another rank needs -> send records
 *   all_to_all_single(send_rec -> recv_rec)
 *   rbpf_shard_step(ctx, anc_bank, slot_ids)            fused step kernel for my N_P physical slots
 * and after the last step one more gather + rbpf_shard_normalise_search(ctx, perm, NULL).            */
typedef struct {
  int32_t rank, world, N_local, N_global, n_nonlin;
  size_t record_doubles;       /* doubles per exchanged particle record [Pt | Pb | F | xl]            */
  size_t recv_capacity;        /* records the receive buffer can hold                                */
  size_t send_capacity;        /* records the send buffer can hold                                   */
  double* fwd_local;           /* [fwd_rows][N_local]: rows xn, then logw (step kernel output); the sharded smoother
                                * appends one row, the measurement part of my particles' ancestor log-weights        */
  double* fwd_gather;          /* [world][fwd_rows][N_local] all_gather target                        */
  double* send_rec;            /* [send_capacity][record_doubles]                                     */
  double* recv_rec;            /* [recv_capacity][record_doubles]                                     */
  int32_t fwd_rows;            /* n_nonlin + 1 (filter) / n_nonlin + 2 (smoother)                     */
} rbpf_shard_views;

int rbpf_shard_create(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng,
                      const rbpf_options* opt, int32_t rank, int32_t world, rbpf_ctx** ctx);
int rbpf_shard_views_get(rbpf_ctx* ctx, rbpf_shard_views* out);
/* After the gather: bring the forward bank into logical order (phys_of_logical_host [N_global]:
 * rank*N_local + physical index of each logical slot; NULL = identity), normalise the global weights,
 * cumsum, trajectory summaries of the step just finished; if ai_host != NULL also draw the ancestors of
 * ALL N_global logical slots (0-based logical ids; the same vector on every rank) into ai_host.
 * Synchronises the stream.                                                                          */
int rbpf_shard_normalise_search(rbpf_ctx* ctx, const int32_t* phys_of_logical_host, int32_t* ai_host);
/* Gather `count` local particles (physical indices idx_host, in send order) into send_rec.           */
int rbpf_shard_pack(rbpf_ctx* ctx, const int32_t* idx_host, int32_t count);
/* One fused step for my N_local physical slots.  slot_ids_host [N_local]: logical id of each physical
 * slot of the NEW generation; anc_bank_host [N_local]: where its ancestor's map state is (< N_local:
 * physical slot of the old local bank, >= N_local: N_local + record index in recv_rec).  Both NULL at
 * t = 0 (identity placement: logical slot rank*N_local + p at physical slot p).                      */
int rbpf_shard_step(rbpf_ctx* ctx, const int32_t* anc_bank_host, const int32_t* slot_ids_host);
/* Device-side planner (the production path): placement of the new generation + exchange plan from the
  * ancestors drawn by the last rbpf_shard_normalise_search, entirely on the device.  counts_host
 * [2*world+2] receives the records to send to / receive from every rank, the number of migrating
 * children, and the first record of recv_rec this step's exchange must write to (records persist across
 * the read-only steps of a lazy cycle).  Afterwards rbpf_shard_pack(ctx, NULL, n_send), rbpf_shard_step(ctx, NULL, NULL) and
 * rbpf_shard_normalise_search(ctx, NULL, ...) use the device plan.                                      */
int rbpf_shard_plan(rbpf_ctx* ctx, int64_t* counts_host);
/* rbpf_shard_normalise_search(ctx, NULL, draw) + rbpf_shard_plan(ctx, counts) in one call (what the production loop
 * uses): the ancestors are not copied to the host and the stream is synchronised once.                        */
int rbpf_shard_normalise_plan(rbpf_ctx* ctx, int64_t* counts_host);
/* Test hook: this rank's view of the current device plan.                                             */
int rbpf_shard_plan_read(rbpf_ctx* ctx, int32_t* slot_ids, int32_t* anc_bank, int32_t* send_idx,
                         int32_t n_send, int32_t* new_gid);
/* traj_max / traj_mean [n_nonlin x N_T] of the steps normalised so far (identical on every rank).     */
int rbpf_shard_trajectories(rbpf_ctx* ctx, double* traj_max, double* traj_mean);
/* xn_traj [n_nonlin x N_global x t] of the steps finished so far (particleFilter.m:117-118: every logical slot's path traced
 * back through the ancestor table; needs keep_history).  The state history and the ancestor table are replicated (they come with
 * the all-gather of the forward bank), so any one rank returns the whole array.                                              */
int rbpf_shard_xn_traj(rbpf_ctx* ctx, double* xn_traj);
/* The HIP stream a context enqueues on (hipStream_t).  A caller that issues its collectives on this stream
 * (torch.cuda.ExternalStream) needs no host synchronisation between the library calls and the collectives.      */
int rbpf_stream_get(rbpf_ctx* ctx, void** hip_stream);
/* on = 1: rbpf_shard_pack / rbpf_shard_step / rbpf_shard_smoother_step return without synchronising the stream (the caller's
 * collectives are ordered behind them on the same stream); on = 0 (default): they synchronise (host / gloo transports).    */
int rbpf_shard_set_async(rbpf_ctx* ctx, int32_t on);
/* Final extraction of the sharded filter (particleFilter.m:220-233), after the last step was gathered and normalised.
 * phase 0: iw_max (logical index, identical on every rank); xl_max [n], P_max [n x n] written by the rank that holds that
 *          particle and zero-filled elsewhere; xl_mean [n] = this rank's share of sum_i w_i xl_i; traj_sample_iwmax
 *          [n_nonlin x N_T] (needs keep_history; identical on every rank).  The caller sum-reduces xl_max, P_max, xl_mean.
 * phase 1: xl_mean is INPUT (the reduced mean); P_mean [n x n] = quirk Q3's last-particle term w_N (P_N + (xl_mean - xl_N)
 *          (xl_mean - xl_N)') written by the rank that holds logical slot N - 1, zero elsewhere (the caller sum-reduces it).
 * NULL outputs are skipped.                                                                                          */
int rbpf_shard_finish(rbpf_ctx* ctx, int32_t phase, double* xl_max, double* P_max, double* xl_mean, double* P_mean,
                      double* traj_sample_iwmax, int32_t* iw_max);
/* Test hook: replace the ancestors drawn for the next step (ai [N_global], logical ids) before rbpf_shard_plan.        */
int rbpf_shard_set_ancestors(rbpf_ctx* ctx, const int32_t* ai);

/* ---- particle-sharded information-form smoother ---------------------------------
 * particleSmootherInformationForm.m with the N = world * N_P particles of every CPF-AS iteration sharded like the
 * filter above.  The extra information-form state (ivec, Imat, halfLogDetP) travels in the particle records.
 * The measurement part of the ancestor weights of the reference trajectory (:205-236: one n x n factorisation per particle, or
 * one sweep over its carried factor) is computed on the rank that holds each particle, BEFORE the all_gather of the forward
 * bank, whose extra row carries it along (one collective per step instead of two); log w and the dynResNorm part are added
 * for all N particles from the gathered bank, identically on every rank, then normalised and sampled.  Results equal the
 * single-GPU rbpf_particle_smoother(info_form = 1) with N particles bit for bit.  Per iteration k the host side does
 *   rbpf_shard_smoother_begin(ctx, k)
 *   t = 0: rbpf_shard_smoother_step(ctx)
 *   t > 0: k > 0 (no refresh due): rbpf_shard_smoother_anc_weights(ctx)
 *          all_gather(fwd_local -> fwd_gather); rbpf_shard_smoother_normalise(ctx, 1)
 *          k > 0: rbpf_shard_smoother_anc_sample(ctx, 0)
 *                 (refresh steps of the carried factors: the refresh sequence below, all_gather(anc_local -> anc_gather),
 *                  rbpf_shard_smoother_anc_sample(ctx, 1))
 *          rbpf_shard_plan; rbpf_shard_pack; all_to_all_single(send_rec -> recv_rec); rbpf_shard_smoother_step(ctx)
 *   all_gather; rbpf_shard_smoother_normalise(ctx, 0); rbpf_shard_smoother_end(ctx, ...)                       */
typedef struct {
  double* anc_local;           /* [N_local] measurement part of my particles' ancestor log-weights (physical order): the
                                * last row of fwd_local                                                    */
  double* anc_gather;          /* [world][N_local] all_gather target of anc_local on its own (refresh steps) */
  /* carried factors (chol_refresh > 1): base matrices exchanged at a refresh, [refresh_capacity][matrix_doubles]; NULL / 0
   * otherwise                                                                                                */
  double* refresh_send;
  double* refresh_recv;
  int64_t refresh_capacity;
  int64_t matrix_doubles;      /* n_lin * n_lin                                                            */
} rbpf_shard_smoother_views;

int rbpf_shard_smoother_create(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng,
                               const rbpf_options* opt, int32_t N_K, int32_t rank, int32_t world, rbpf_ctx** ctx);
int rbpf_shard_smoother_views_get(rbpf_ctx* ctx, rbpf_shard_smoother_views* out);
/* Start of iteration k (0-based): rewind; k > 0: H along the reference trajectory + suffix sums (:120,:132-146). */
int rbpf_shard_smoother_begin(rbpf_ctx* ctx, int32_t k);
/* Global weights of the finished step; want_draw: ancestors of the ordinary slots of the next one (:160-166).   */
int rbpf_shard_smoother_normalise(rbpf_ctx* ctx, int32_t want_draw);
/* k > 0, t > 0, before the gather: measurement part of my particles' ancestor log-weights -> anc_local (:205-236). */
int rbpf_shard_smoother_anc_weights(rbpf_ctx* ctx);
/* After gather + normalise: add log w and the dynResNorm part (:175-182,232) for all N particles, normalise (:243-245) and
 * draw ai(N_P) (:248).  separate_gather: 0 = the measurement parts came with the forward bank, 1 = they are in anc_gather. */
int rbpf_shard_smoother_anc_sample(rbpf_ctx* ctx, int32_t separate_gather);
/* Carried ancestor-weight factors in the sharded smoother (rbpf_options.chol_refresh = K > 1, lazy_depth as usual): between
 * refreshes rbpf_shard_smoother_anc_weights runs one up/down-date sweep per particle over its ancestor's factor, which migrates
 * inside the particle records in place of Imat.  At the steps t = 1 and (t - 1) % K == 0 of an iteration k > 0 the factors are
 * refreshed INSTEAD of rbpf_shard_smoother_anc_weights:
 *   rbpf_shard_smoother_refresh_begin(ctx, owner_now, base_loc)   both [N_global] int32, identical on every rank:
 *        owner_now[j] = rank * N_local + slot of logical slot j's particle, base_loc[j] = the same for the ancestor its
 *        information matrix is rebuilt from (-1: the common initial matrix, nothing to fetch)
 *   the host mirror derives the fetch plan from the two tables (multigpu.plan_refresh: unique (destination, matrix) pairs),
 *   rbpf_shard_smoother_refresh_pack(ctx, slots, count); all_to_all_single(refresh_send -> refresh_recv);
 *   rbpf_shard_smoother_refresh_end(ctx, base_index, n_recv)      base_index [N_local]: bank slot or N_local + position in
 *        refresh_recv
 * (after the gather and normalise of the finished step: the walk reads the state history), then all_gather(anc_local ->
 * anc_gather) and rbpf_shard_smoother_anc_sample(ctx, 1).                                           */
 /* (the plan: rbpf_plan_refresh below -- the one implementation both multi-GPU drivers use) */
int rbpf_shard_smoother_refresh_begin(rbpf_ctx* ctx, int32_t* owner_now, int32_t* base_loc);
/* Fetch plan of one refresh for `rank`, from the two replicated tables (host arithmetic only, no device access): rank q sends
 * rank r each matrix some particle on r needs, once (siblings share it), ordered by (destination, source, slot).  send_slots
 * [<= send_capacity] bank slots to pack, concatenated per destination (*n_send of them); send_counts / recv_counts [world] of
 * this rank; send_totals / recv_totals [world]: every rank's totals (capacity check, identical on all ranks); base_index
 * [n_local]: a slot of the own bank, or n_local + position in refresh_recv.  Used by the in-library driver (rbpf_options.
 * n_devices) and by multigpu.ShardedSmootherSession alike; multigpu.plan_refresh (numpy) is its specification in the tests. */
int rbpf_plan_refresh(const int32_t* owner_now, const int32_t* base_loc, int32_t N_global, int32_t n_local, int32_t world,
                      int32_t rank, int32_t* send_slots, int32_t send_capacity, int32_t* n_send, int64_t* send_counts,
                      int64_t* recv_counts, int64_t* send_totals, int64_t* recv_totals, int32_t* base_index);
/* Before refresh_pack, when the plan's largest per-rank total exceeds rbpf_shard_smoother_views.refresh_capacity: every rank
 * (the plan is replicated) enlarges its refresh buffers to hold `count` matrices -- rbpf_shard_smoother_views_get then returns the
 * new pointers / capacity -- unless rbpf_options.exchange_capacity > 0 made the capacity a hard limit (RBPF_ERR_OUT_OF_MEMORY).  */
int rbpf_shard_smoother_refresh_reserve(rbpf_ctx* ctx, int64_t count);
int rbpf_shard_smoother_refresh_pack(rbpf_ctx* ctx, const int32_t* slots, int32_t count);
int rbpf_shard_smoother_refresh_end(rbpf_ctx* ctx, const int32_t* base_index, int32_t n_recv);
/* One information-form time step of my particles (:256-335), using the plan of rbpf_shard_plan for t > 0.        */
int rbpf_shard_smoother_step(rbpf_ctx* ctx);
/* End of iteration (:346-354): ak = sample(w), new reference trajectory -> XNK_k [n_nonlin x N_T] (every rank);
 * XLK_k [n_lin], PK_k [n_lin x n_lin] are filled by the rank that holds particle ak (*owner_rank), zero elsewhere. */
int rbpf_shard_smoother_end(rbpf_ctx* ctx, double* XNK_k, double* XLK_k, double* PK_k, int32_t* ak,
                            int32_t* owner_rank);

/* ---- helper kernels exposed for parity tests (a5-a8, a19 of SURVEY 8a) ------------------------- */
/* The uniforms / normals the Philox generator hands to slot i at step t of iteration k, in replay
 * layout (U [N_P x (N_T-1)], Z [n_w x N_P x (N_T-1)]), so a replay run can reproduce a Philox run. */
int rbpf_philox_fill(uint64_t seed, int32_t k_iter, int32_t N_P, int32_t N_T, int32_t n_w,
                     double* U, double* Z, double* Ufin);
/* measModel of the family evaluated on the device: xn [n_nonlin x Npred] -> dy stored as
 * [n_y x n_lin x Npred] (i.e. H_i contiguous per particle).  run_dense3D_magfield.m:265-279.      */
int rbpf_meas_model(const rbpf_model* model, int32_t n_nonlin, int32_t n_pred, const double* xn,
                    double* dy);
/* dynModel with injected normals: xn [n_nonlin x Np], odo [n_odo], cholQ from (dt,Q),
 * z [n_w x Np] -> xn_next [n_nonlin x Np].  run_dense3D_magfield.m:301-308.                       */
int rbpf_dyn_model(const rbpf_model* model, int32_t n_nonlin, int32_t n_w, int32_t n_odo,
                   int32_t n_p, const double* xn, const double* odo, double dt, const double* Q,
                   const double* z, double* xn_next);
/* dynResNorm: eDyn [n_w x Np] for reference state xnk_t against xn [n_nonlin x Np]
 * (run_dense3D_magfield.m:202-203; default additive form when model->use_dyn_res_norm == 0).      */
int rbpf_dyn_res_norm(const rbpf_model* model, int32_t n_nonlin, int32_t n_w, int32_t n_odo,
                      int32_t n_p, const double* xnk_t, const double* xn, const double* odo,
                      double dt, const double* Q, double* e_dyn);
/* tools/sample.m:30-32 applied to n_draws uniforms: ind[j] = sum(cumsum(w) < u[j]) (0-based,
 * clamped to N-1).                                                                                */
int rbpf_sample(int32_t N, const double* w, int32_t n_draws, const double* u, int32_t* ind);
/* tools/JacobianPhi3D.m:29-64: x [3 x Np] -> J [3 x 3 x m x Np].                                  */
int rbpf_jacobian_phi3d(const rbpf_model* model, int32_t n_p, const double* x,
                        const double* lower, const double* upper, double* J);
/* particleSmoother.m:221-229 for a batch of matrices: cS = chol(S,'lower') (one retry with S + jitter*I),
 * v = cS \ e, logw[b] = -sum(log(diag(cS))) - v'v/2 - M/2*log(2*pi).  S [batch][M x M] column-major (lower
 * triangle read), e [batch][M].  variant 0: automatic kernel choice; 16: the 16-column kernel; 64 / 648 / 644: the
 * 64-column kernel (waves by size / 8 / 4); 1: the register-resident kernel in its default shape (64 <= M <= 143, information
 * form only; two waves per matrix); 11 / 12 / 14: the same with one / two / four waves per matrix (14: the r02 kernel); 10: one
 * wave per matrix, left-looking on register tiles (the next block column's loads in flight during a column's work).
 * variant + 1000: the information-form loaders and expression of particleSmootherInformationForm.m:224-236 with
 * ImatAddt = ivecAddt = 0, i.e. logw[b] = -sum(log(diag(cI))) + v'v/2 and no retry.  reps >= 1 repeats the launch;
 * *ms (may be NULL) = mean kernel time (HIP events).  *status (may be NULL): bit 1 set when a factorisation failed
 * (its logw is NaN).                                                                               */
int rbpf_chol_weights(int32_t M, int32_t batch, const double* S, const double* e, double jitter,
                      int32_t variant, int32_t reps, double* logw, int32_t* status, double* ms);
/* What rbpf_options.chol_refresh = `requested` means for an information-form smoother of this model family and size: the K in
 * use (> 1: carried factors refreshed every K-th step; 1: from-scratch factorisation every step).  The multi-rank drivers need it
 * to issue the refresh exchange at the right steps.                                                                          */
int32_t rbpf_chol_refresh_resolve(int32_t model_kind, int32_t n_lin, int32_t n_y, int32_t requested);
/* The sweep kernel of the carried factors (rbpf_options.chol_refresh) on its own, for the kernel-level parity test and the
 * bench: ONE augmented factor L [(n+1) x (n+1)] column-major lower = [chol(A) 0; z' *], z = chol(A) \ b, replicated `batch`
 * times; U, V [d x n] column-major (row a = update / downdate vector a), eta [d] = the entry both vectors carry in the augmented
 * row.  Result (copy 0): L_out (same layout) = the factor of A + U'U - V'V with the row (L_out \ (b + U' eta - V' eta))', and
 * *logw = -sum(log(diag(L_out))) + z_out' z_out / 2 (particleSmootherInformationForm.m:234-236 without the qf / halfLogDetP
 * terms).  d = 1 or 3, n <= 575.  *status bit 1: a downdate lost definiteness.  reps / *ms as in rbpf_chol_weights.        */
int rbpf_chol_sweep_probe(int32_t n, int32_t d, int32_t batch, const double* L, const double* U, const double* V,
                          const double* eta, int32_t reps, double* L_out, double* logw, int32_t* status, double* ms);
/* Quaternion helpers of tools/ on the device, batched over n columns, for the parity tests:
 *   op 0  expq, scalar branch   tools/expq.m:22-31  (flip when q0 < 0)          in [3 x n]      -> out [4 x n]
 *   op 1  expq, batched branch  tools/expq.m:33-37  (flip when q0 <= 0)         in [3 x n]      -> out [4 x n]
 *   op 2  logq, scalar branch   tools/logq.m:25-30  (flip when q0 < 0)          in [4 x n]      -> out [3 x n]
 *   op 3  logq, batched branch  tools/logq.m:32-35  (flip when q0 <= 0)         in [4 x n]      -> out [3 x n]
 *   op 4  qLeft                 tools/qLeft.m:30-40                              in [4 x n]      -> out [4 x 4 x n]
 *   op 5  qRight                tools/qRight.m:29-39                             in [4 x n]      -> out [4 x 4 x n]
 *   op 6  qInv                  tools/qInv.m:27-31                               in [4 x n]      -> out [4 x n]
 *   op 7  quat2rmat             tools/quat2rmat.m:27-40                          in [4 x n]      -> out [3 x 3 x n]
 *   op 8  mcross                tools/mcross.m:33-42                             in [3 x n]      -> out [3 x 3 x n]
 * q0 > 1 by rounding is clamped before acos (quirk Q7).                                                             */
int rbpf_quat_helpers(int32_t op, int32_t n, const double* in, double* out);
/* The wave-level reduction of the symmetric-storage step kernel (rbpf_step_sym.hip: v_permlane32_swap / v_permlane16_swap
 * folds + DPP row rotations) on its own, for the kernel-level test: in [4][64] (four values per lane of one wave64) -> out [4],
 * out[v] = sum over the 64 lanes of in[v][.] in the kernel's fixed association order.                                    */
int rbpf_probe_wave_reduce(const double* in, double* out);

#ifdef __cplusplus
}
#endif
#endif /* RBPF_H_ */
