/*
 * ORACLE -- TEST / BASELINE INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C restatement of src/particleFilter.m (dense branch), src/particleSmoother.m and
 * src/particleSmootherInformationForm.m (dense branches) for the two dense model families,
 * used (a) as a second, independent oracle next to oracle/rbpf_oracle.py and (b) as the
 * "CPU restatement, not MATLAB" baseline that bench.py times on the GPU box's host cores
 * (cpu_baseline.kind = "port").  It keeps the reference's algorithmic structure -- including the
 * O(N) cumsum inside every sample() call (tools/sample.m:30-32 via particleFilter.m:106), the full
 * gather copies xl = xl(:,ai), P = P(:,:,ai) (:112-113), the eager history permutation (:118) and
 * the two separate per-particle loops that each recompute S and chol(S) (:126-151, :164-204) --
 * with OpenMP over the particle loops (the reference is serial; threads used are reported).
 *
 * PARITY STATUS: parity unpinned by the reference itself (no MATLAB/Octave, no reference goldens);
 * checked against oracle/rbpf_oracle.py in tests/test_oracle_c.py.
 *
 * It takes the same structs as include/rbpf.h so the same marshalling drives both.
 * All file:line citations are relative to /root/reference/.
 */
#include "../include/rbpf.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Working precision.  The default build computes in IEEE fp64 like the reference (MATLAB doubles).  -DRBPF_ORACLE_LONG_DOUBLE
 * builds the ARBITER: the same statements in x87 extended precision (64-bit significand, 2^-11 of fp64's rounding error) on the
 * same fp64 inputs, its results rounded to fp64 once on the way out.  It answers what two fp64 evaluations cannot settle between
 * themselves: which of them is closer to the exact result of the reference's formulas (tests/golden/make_arbiter_fixture.py). */
#ifdef RBPF_ORACLE_LONG_DOUBLE
typedef long double real;
#define RM(f) f##l
#define R_PI 3.14159265358979323846264338327950288L
#else
typedef double real;
#define RM(f) f
#define R_PI M_PI
#endif

/* fp64 API buffers <-> working precision */
static void cp_in(real* dst, const double* src, size_t count) { for (size_t q = 0; q < count; ++q) dst[q] = (real)src[q]; }
static void cp_out(double* dst, const real* src, size_t count) { for (size_t q = 0; q < count; ++q) dst[q] = (double)src[q]; }

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* tools/sample.m:30-32: wc = cumsum(w); ind = sum(wc < u) + 1  (0-based here) */
static int sample_ref(const real* w, int N, real u, real* wc) {
  real run = 0.0;
  int cnt = 0;
  for (int j = 0; j < N; ++j) { run += w[j]; wc[j] = run; }
  for (int j = 0; j < N; ++j) cnt += (wc[j] < u);
  return cnt;
}

/* chol(A,'lower') for small/medium n, column-major; returns 0 on success */
static int chol_lower(const real* A, int n, real* L) {
  memset(L, 0, sizeof(real) * (size_t)n * n);
  for (int j = 0; j < n; ++j) {
    real s = A[j + (size_t)n * j];
    for (int k = 0; k < j; ++k) s -= L[j + (size_t)n * k] * L[j + (size_t)n * k];
    if (!(s > 0.0)) return j + 1;
    const real ljj = RM(sqrt)(s);
    L[j + (size_t)n * j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      real v = A[i + (size_t)n * j];
      for (int k = 0; k < j; ++k) v -= L[i + (size_t)n * k] * L[j + (size_t)n * k];
      L[i + (size_t)n * j] = v / ljj;
    }
  }
  return 0;
}

/* tools/qLeft.m:30-35 applied to p */
static void qleft_mul(const real* q, const real* p, real* r) {
  r[0] = q[0] * p[0] - q[1] * p[1] - q[2] * p[2] - q[3] * p[3];
  r[1] = q[1] * p[0] + q[0] * p[1] - q[3] * p[2] + q[2] * p[3];
  r[2] = q[2] * p[0] + q[3] * p[1] + q[0] * p[2] - q[1] * p[3];
  r[3] = q[3] * p[0] - q[2] * p[1] + q[1] * p[2] + q[0] * p[3];
}

/* tools/expq.m:22-31 */
static void expq(const real* phi, real* eq) {
  const real mag = RM(sqrt)(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
  const real den = mag + (mag == 0.0 ? 1.0 : 0.0);
  eq[0] = RM(cos)(mag);
  for (int k = 0; k < 3; ++k) eq[1 + k] = phi[k] / den * RM(sin)(mag);
  if (eq[0] < 0.0) for (int k = 0; k < 4; ++k) eq[k] = -eq[k];
}

/* tools/quat2rmat.m:27-33, R[row*3+col] */
static void quat2rmat(const real* q, real* R) {
  const real q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  R[0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3; R[1] = 2 * q1 * q2 - 2 * q0 * q3; R[2] = 2 * q1 * q3 + 2 * q0 * q2;
  R[3] = 2 * q1 * q2 + 2 * q0 * q3; R[4] = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3; R[5] = 2 * q2 * q3 - 2 * q0 * q1;
  R[6] = 2 * q1 * q3 - 2 * q0 * q2; R[7] = 2 * q2 * q3 + 2 * q0 * q1; R[8] = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
}

typedef struct {
  int kind, m, dim, nN, n, d, nw, nodo;
  const int32_t* NN;
  real L[3];
} omodel;

/* dynModel closures: run_dense3D_magfield.m:301-308 / run_dense2D_withHeading.m:75-76 */
static int dyn_model(const omodel* M, const real* x, const real* odo, real dt, const double* Q, const double* z,
                     real* xp) {
  if (M->kind == RBPF_MODEL_DENSE_MAG_6D) {
    real A[9], Lp[9], La[9], phi[3], eq[4], dq[4];
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) A[r + 3 * c] = dt * Q[r + 6 * c];
    if (chol_lower(A, 3, Lp)) return 1;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) A[r + 3 * c] = dt * Q[(3 + r) + 6 * (3 + c)];
    if (chol_lower(A, 3, La)) return 1;
    for (int r = 0; r < 3; ++r) {
      real s = 0.0;
      for (int c = 0; c < 3; ++c) s += Lp[r + 3 * c] * z[c];
      xp[r] = x[r] + odo[r] + s;
      real a = 0.0;
      for (int c = 0; c < 3; ++c) a += La[r + 3 * c] * z[3 + c];
      phi[r] = a;
    }
    expq(phi, eq);
    qleft_mul(&odo[3], eq, dq);
    qleft_mul(&x[3], dq, &xp[3]);
  } else {
    const real c = RM(cos)(x[2]), s = RM(sin)(x[2]);
    if (!(dt * Q[0] > 0.0)) return 1;
    xp[0] = x[0] + (c * odo[0] + s * odo[1]);
    xp[1] = x[1] + (-s * odo[0] + c * odo[1]);
    xp[2] = x[2] + odo[2] + RM(sqrt)(dt * Q[0]) * z[0];
  }
  return 0;
}

/* measModel closures -> H [d x n] column-major.  run_dense3D_magfield.m:265-279 with
 * tools/domain_cartesian_dx.m:142-170; run_dense2D_withHeading.m:168 with :84-93. */
static void meas_model(const omodel* M, const real* x, real* H) {
  const int m = M->m, d = M->d;
  if (M->kind == RBPF_MODEL_DENSE_MAG_6D) {
    real R[9];
    quat2rmat(&x[3], R);
    for (int c = 0; c < M->n; ++c) {
      real g[3];
      if (c < 3) { g[0] = (c == 0); g[1] = (c == 1); g[2] = (c == 2); }
      else {
        const int j = c - 3;
        for (int di = 0; di < 3; ++di) {
          real v = 1.0;
          for (int a = 0; a < 3; ++a) {
            const real La = M->L[a];
            const real nn = (real)M->NN[j + (size_t)m * a];
            const real arg = R_PI * nn * (x[a] + La) / (2.0 * La);
            if (a == di) v = v * R_PI * nn / (2.0 * La * RM(sqrt)(La)) * RM(cos)(arg);
            else v = v * 1.0 / RM(sqrt)(La) * RM(sin)(arg);
          }
          g[di] = v;
        }
      }
      for (int k = 0; k < 3; ++k) H[k + (size_t)d * c] = R[0 + k] * g[0] + R[3 + k] * g[1] + R[6 + k] * g[2];
    }
  } else {
    for (int c = 0; c < M->n; ++c) {
      real v = 1.0;
      for (int a = 0; a < 2; ++a) {
        const real La = M->L[a];
        const real nn = (real)M->NN[c + (size_t)m * a];
        v = v * 1.0 / RM(sqrt)(La) * RM(sin)(R_PI * nn * (x[a] + La) / (2.0 * La));
      }
      H[c] = v;
    }
  }
}

/* e = y - H xl ; SS = H P H' + R ; cS = chol (jitter retry) -- particleFilter.m:139-148.
 * HP is scratch [d x n].  Returns 0 ok, 1 failed twice. */
static int innovation(int n, int d, const real* H, const real* P, const real* xl, const real* y, const double* R,
                      real jitter, real* HP, real* e, real* SS, real* cS) {
  for (int k = 0; k < d; ++k) {
    real s = 0.0;
    for (int c = 0; c < n; ++c) s += H[k + (size_t)d * c] * xl[c];
    e[k] = y[k] - s;
  }
  /* HP = H * P  (d x n) */
  for (int c = 0; c < n; ++c) {
    const real* Pc = P + (size_t)n * c;
    for (int k = 0; k < d; ++k) {
      real s = 0.0;
      for (int r = 0; r < n; ++r) s += H[k + (size_t)d * r] * Pc[r];
      HP[k + (size_t)d * c] = s;
    }
  }
  for (int b = 0; b < d; ++b)
    for (int a = 0; a < d; ++a) {
      real s = 0.0;
      for (int c = 0; c < n; ++c) s += HP[a + (size_t)d * c] * H[b + (size_t)d * c];
      SS[a + d * b] = s + R[a + d * b];
    }
  if (chol_lower(SS, d, cS)) {
    real SJ[64];
    for (int q = 0; q < d * d; ++q) SJ[q] = SS[q];
    for (int q = 0; q < d; ++q) SJ[q + d * q] += jitter;
    if (chol_lower(SJ, d, cS)) return 1;
  }
  return 0;
}

int rbpf_oracle_particle_filter(const rbpf_model* model, const rbpf_problem* p, const rbpf_rng* rng,
                                const rbpf_options* opt, rbpf_filter_out* out, int n_threads, double* loop_seconds) {
  if (!model || !p || !rng || !out || rng->mode != RBPF_RNG_REPLAY) return RBPF_ERR_INVALID_ARG;
  omodel M;
  M.kind = model->kind; M.m = model->m_basis; M.dim = model->dim; M.NN = model->NN;
  M.nN = p->n_nonlin; M.n = p->n_lin; M.d = p->n_y; M.nw = p->n_w; M.nodo = p->n_odo;
  for (int a = 0; a < 3; ++a) M.L[a] = model->L[a];
  const int N = p->N_P, T = p->N_T, nN = M.nN, n = M.n, d = M.d, nw = M.nw;
  if (d > 8 || nN > 8 || nw > 8) return RBPF_ERR_UNSUPPORTED;
  const int keep_hist = (opt && opt->keep_history) || out->xn_traj || out->traj_sample_iwmax;
  const real jitter = (opt && opt->jitter > 0) ? opt->jitter : 1e-3;            /* :89 */
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#else
  (void)n_threads;
#endif
  const size_t nn2 = (size_t)n * n;
  real* w = malloc(sizeof(real) * N), *logw = malloc(sizeof(real) * N);
  real* xn = malloc(sizeof(real) * nN * N), *xn_ = malloc(sizeof(real) * nN * N);
  real* xl = malloc(sizeof(real) * (size_t)n * N), *xl2 = malloc(sizeof(real) * (size_t)n * N);
  real* P = malloc(sizeof(real) * nn2 * N), *P2 = malloc(sizeof(real) * nn2 * N);
  real* H = malloc(sizeof(real) * (size_t)d * n * N);
  int* ai = calloc(N, sizeof(int));
  real* hist = keep_hist ? calloc((size_t)nN * N * T, sizeof(real)) : NULL;
  real* hist2 = keep_hist ? malloc(sizeof(real) * (size_t)nN * N) : NULL;
  int nthr = 1;
#ifdef _OPENMP
  nthr = omp_get_max_threads();
#endif
  real* wc_all = malloc(sizeof(real) * (size_t)N * nthr);
  real* scratch = malloc(sizeof(real) * ((size_t)d * n * 2 + nn2) * nthr);   /* HP, M (n x d), K S (n x d) reuse */
  int status = RBPF_OK, iw_max = 0;
  if (!w || !logw || !xn || !xn_ || !xl || !xl2 || !P || !P2 || !H || !ai || !wc_all || !scratch || (keep_hist && (!hist || !hist2))) {
    status = RBPF_ERR_OUT_OF_MEMORY; goto done;
  }
  for (int i = 0; i < N; ++i) {                                                     /* :55-67 */
    w[i] = 1.0 / N; logw[i] = RM(log)(w[i]);
    for (int c = 0; c < nN; ++c) xn[c + (size_t)nN * i] = p->x0_nonlin[c];
    cp_in(xl + (size_t)n * i, p->x0_lin + (size_t)n * (p->x0_lin_cols > 1 ? i : 0), (size_t)n);
    cp_in(P + nn2 * i, p->P0_lin, nn2);
  }
  if (keep_hist) memcpy(hist, xn, sizeof(real) * nN * N);                        /* :96 */
  if (out->traj_max) for (size_t q = 0; q < (size_t)nN * T; ++q) out->traj_max[q] = NAN;
  if (out->traj_mean) for (size_t q = 0; q < (size_t)nN * T; ++q) out->traj_mean[q] = NAN;

  const real t0 = now_s();
  for (int t = 0; t < T; ++t) {                                                     /* :100 */
    const real* yt = NULL;
    real ybuf[8];
    for (int k = 0; k < d; ++k) ybuf[k] = p->y[t + (size_t)T * k];
    yt = ybuf;
    if (t != 0) {
      memcpy(xn_, xn, sizeof(real) * nN * N);                                     /* :102 */
      const real dtt = p->dt[p->dt_len > 1 ? t - 1 : 0];
      const double* Qt = p->Q + (size_t)(p->q_pages > 1 ? t - 1 : 0) * nw * nw;
      real odo[8];
      for (int k = 0; k < M.nodo; ++k) odo[k] = p->odometry[(t - 1) + (size_t)p->odo_ld * k];
      const double* U = rng->U + (size_t)(t - 1) * N;
      const double* Z = rng->Z + (size_t)(t - 1) * N * nw;
      int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
      for (int i = 0; i < N; ++i) {                                                 /* :104-109 */
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        int a = sample_ref(w, N, U[i], wc_all + (size_t)N * tid);
        if (a >= N) a = N - 1;   /* the reference would index out of range here (MATLAB error) */
        ai[i] = a;
        bad |= dyn_model(&M, xn_ + (size_t)nN * a, odo, dtt, Qt, Z + (size_t)nw * i, xn + (size_t)nN * i);
      }
      if (bad) { status = RBPF_ERR_CHOL_FAILED; goto done; }
#pragma omp parallel for schedule(static)
      for (int i = 0; i < N; ++i) {                                                 /* :112-113 */
        memcpy(xl2 + (size_t)n * i, xl + (size_t)n * ai[i], sizeof(real) * n);
        memcpy(P2 + nn2 * i, P + nn2 * ai[i], sizeof(real) * nn2);
      }
      { real* tmp = xl; xl = xl2; xl2 = tmp; tmp = P; P = P2; P2 = tmp; }
      if (keep_hist) {                                                              /* :117-118 */
        memcpy(hist + (size_t)nN * N * t, xn, sizeof(real) * nN * N);
        for (int s = 0; s < t; ++s) {
          real* hs = hist + (size_t)nN * N * s;
          for (int i = 0; i < N; ++i) memcpy(hist2 + (size_t)nN * i, hs + (size_t)nN * ai[i], sizeof(real) * nN);
          memcpy(hs, hist2, sizeof(real) * nN * N);
        }
      }
    }
    /* dy = measModel(xn)  :124 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) meas_model(&M, xn + (size_t)nN * i, H + (size_t)d * n * i);
    /* importance weights :126-151 */
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int i = 0; i < N; ++i) {
      int tid = 0;
#ifdef _OPENMP
      tid = omp_get_thread_num();
#endif
      real* HP = scratch + ((size_t)d * n * 2 + nn2) * tid;
      real e[8], SS[64], cS[64], v[8];
      if (innovation(n, d, H + (size_t)d * n * i, P + nn2 * i, xl + (size_t)n * i, yt, p->R, jitter, HP, e, SS, cS)) { bad |= 1; continue; }
      real sl = 0.0, vv = 0.0;
      for (int a = 0; a < d; ++a) {                                                 /* v = cS\e */
        real s = e[a];
        for (int k = 0; k < a; ++k) s -= cS[a + d * k] * v[k];
        v[a] = s / cS[a + d * a];
        sl += RM(log)(cS[a + d * a]); vv += v[a] * v[a];
      }
      logw[i] = -sl - 0.5 * vv - 0.5 * d * RM(log)(2 * R_PI);                           /* :150 */
    }
    if (bad) { status = RBPF_ERR_CHOL_FAILED; goto done; }
    {                                                                               /* :154-161 */
      real c = -INFINITY, s = 0.0;
      for (int i = 0; i < N; ++i) if (logw[i] > c) c = logw[i];
      for (int i = 0; i < N; ++i) s += RM(exp)(logw[i] - c);
      const real lse = c + RM(log)(s);
      real best = -1.0;
      for (int i = 0; i < N; ++i) { w[i] = RM(exp)(logw[i] - lse); if (w[i] > best) { best = w[i]; iw_max = i; } }
      for (int k = 0; k < nN; ++k) {
        real mean = 0.0;
        for (int i = 0; i < N; ++i) mean += xn[k + (size_t)nN * i] * w[i];
        if (out->traj_mean) out->traj_mean[k + (size_t)nN * t] = mean;
        if (out->traj_max) out->traj_max[k + (size_t)nN * t] = xn[k + (size_t)nN * iw_max];
      }
    }
    if (out->trace_logw) cp_out(out->trace_logw + (size_t)N * t, logw, (size_t)(N));
    if (out->trace_w) cp_out(out->trace_w + (size_t)N * t, w, (size_t)(N));
    if (out->trace_ai) for (int i = 0; i < N; ++i) out->trace_ai[i + (size_t)N * t] = ai[i];
    /* Kalman update :164-204 */
    bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int i = 0; i < N; ++i) {
      int tid = 0;
#ifdef _OPENMP
      tid = omp_get_thread_num();
#endif
      real* HP = scratch + ((size_t)d * n * 2 + nn2) * tid;
      real* Mm = HP + (size_t)d * n;       /* (dyt'/cS')/cS : n x d */
      real* K = Mm + (size_t)d * n;        /* n x d, then KS n x d reuses the tail */
      real* Hi = H + (size_t)d * n * i;
      real* Pi = P + nn2 * i;
      real* xli = xl + (size_t)n * i;
      real e[8], SS[64], cS[64];
      if (innovation(n, d, Hi, Pi, xli, yt, p->R, jitter, HP, e, SS, cS)) { bad |= 1; continue; }
      for (int r = 0; r < n; ++r) {                                                 /* row r of dyt' = H(:,r)' */
        real u[8], kk[8];
        for (int a = 0; a < d; ++a) {                                               /* / cS' */
          real s = Hi[a + (size_t)d * r];
          for (int k = 0; k < a; ++k) s -= cS[a + d * k] * u[k];
          u[a] = s / cS[a + d * a];
        }
        for (int a = d - 1; a >= 0; --a) {                                          /* / cS */
          real s = u[a];
          for (int k = a + 1; k < d; ++k) s -= cS[k + d * a] * kk[k];
          kk[a] = s / cS[a + d * a];
        }
        for (int a = 0; a < d; ++a) Mm[r + (size_t)n * a] = kk[a];
      }
      for (int a = 0; a < d; ++a)                                                   /* K = P * M  :194 */
        for (int r = 0; r < n; ++r) K[r + (size_t)n * a] = 0.0;
      for (int a = 0; a < d; ++a)
        for (int c = 0; c < n; ++c) {
          const real mv = Mm[c + (size_t)n * a];
          const real* Pc = Pi + (size_t)n * c;
          for (int r = 0; r < n; ++r) K[r + (size_t)n * a] += Pc[r] * mv;
        }
      for (int r = 0; r < n; ++r) {                                                 /* xl += K e  :197 */
        real s = 0.0;
        for (int a = 0; a < d; ++a) s += K[r + (size_t)n * a] * e[a];
        xli[r] += s;
      }
      real* KS = K + (size_t)n * d;                                               /* K*SS */
      for (int b = 0; b < d; ++b)
        for (int r = 0; r < n; ++r) {
          real s = 0.0;
          for (int a = 0; a < d; ++a) s += K[r + (size_t)n * a] * SS[a + d * b];
          KS[r + (size_t)n * b] = s;
        }
      for (int c = 0; c < n; ++c) {                                                 /* P -= (K*SS)*K'  :198 */
        real* Pc = Pi + (size_t)n * c;
        for (int r = 0; r < n; ++r) {
          real s = 0.0;
          for (int a = 0; a < d; ++a) s += KS[r + (size_t)n * a] * K[c + (size_t)n * a];
          Pc[r] -= s;
        }
      }
    }
    if (bad) { status = RBPF_ERR_CHOL_FAILED; goto done; }
  }
  if (loop_seconds) *loop_seconds = now_s() - t0;

  /* final extraction :220-233 */
  if (out->iw_max) *out->iw_max = iw_max;
  if (out->xl_max) cp_out(out->xl_max, xl + (size_t)n * iw_max, (size_t)(n));
  if (out->P_max) cp_out(out->P_max, P + nn2 * iw_max, (size_t)(nn2));
  if (out->xl_mean || out->P_mean) {
    real* xm = malloc(sizeof(real) * n);
    for (int r = 0; r < n; ++r) { real s = 0.0; for (int i = 0; i < N; ++i) s += xl[r + (size_t)n * i] * w[i]; xm[r] = s; }
    if (out->xl_mean) cp_out(out->xl_mean, xm, (size_t)(n));
    if (out->P_mean) {                                                              /* quirk Q3: '=' in :229 */
      const int i = N - 1;
      for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r)
          out->P_mean[r + (size_t)n * c] = w[i] * (P[nn2 * i + r + (size_t)n * c] + (xm[r] - xl[r + (size_t)n * i]) * (xm[c] - xl[c + (size_t)n * i]));
    }
    free(xm);
  }
  if (out->traj_sample_iwmax && hist)
    for (int t = 0; t < T; ++t) cp_out(out->traj_sample_iwmax + (size_t)nN * t, hist + (size_t)nN * N * t + (size_t)nN * iw_max, (size_t)(nN));
  if (out->xn_traj && hist) cp_out(out->xn_traj, hist, (size_t)((size_t)nN * N * T));
  if (out->final_xn) cp_out(out->final_xn, xn, (size_t)(nN * N));
  if (out->final_xl) cp_out(out->final_xl, xl, (size_t)((size_t)n * N));
  if (out->final_P) cp_out(out->final_P, P, (size_t)(nn2 * N));
done:
  free(w); free(logw); free(xn); free(xn_); free(xl); free(xl2); free(P); free(P2); free(H); free(ai);
  free(hist); free(hist2); free(wc_all); free(scratch);
  return status;
}

/* ================================================================================================================
 * The two conditional particle smoothers (CPF-AS), dense branch: src/particleSmoother.m:88-366 (covariance-form ancestor
 * weights) and src/particleSmootherInformationForm.m:98-362 (information-form ancestor weights).  Written from the .m
 * files, not from oracle/rbpf_oracle.py: explicit *_pred arrays, eager history permutation, a fresh cumsum inside every
 * sample() call, the stacked future Jacobian dy(t:T) materialised per step.  Randomness is replayed:
 *   U [N_P x (N_T-1) x N_K]      rand of sample() for slot i at step t (k > 1: slot N_P's single rand of :241 / :248)
 *   Z [n_w x N_P x (N_T-1) x N_K] randn of dynModel
 *   Ufin [N_K]                   rand of ak = sample(w) (:346)
 * ================================================================================================================ */
static void logq(const real* qin, real* lq) {                 /* tools/logq.m:25-31, q0 > 1 clamped (quirk Q7) */
  real q[4] = {qin[0], qin[1], qin[2], qin[3]};
  if (q[0] < 0.0) for (int k = 0; k < 4; ++k) q[k] = -q[k];
  const real na = RM(acos)(q[0] > 1.0 ? 1.0 : q[0]);
  const real den = RM(sin)(na) + (na == 0.0 ? 1.0 : 0.0);
  for (int k = 0; k < 3; ++k) lq[k] = na * q[1 + k] / den;
}

/* eDyn [1 x nw] = r' / chol(dt*Q,'lower')  (right division by the LOWER factor: x*L = r', solved from the last column) */
static int dyn_res_norm(const omodel* M, int use_handle, const real* xnk, const real* xni, const real* odo, real dt,
                        const double* Q, real* eDyn) {
  const int nw = M->nw;
  real r[8], A[64], L[64];
  if (use_handle && M->kind == RBPF_MODEL_DENSE_MAG_6D) {         /* run_dense3D_magfield.m:202-203 */
    for (int c = 0; c < 3; ++c) r[c] = xnk[c] - xni[c] - odo[c];
    const real dqi[4] = {odo[3], -odo[4], -odo[5], -odo[6]};    /* qInv(dx(4:7)) */
    const real xqi[4] = {xni[3], -xni[4], -xni[5], -xni[6]};    /* qInv(xni(iQuat)) */
    real t1[4], t2[4];
    qleft_mul(dqi, xqi, t1);                                      /* qLeft(qInv(dx)) * qInv(xni) */
    qleft_mul(t1, &xnk[3], t2);                                   /* qLeft(...) * xnk(iQuat) */
    logq(t2, &r[3]);
  } else if (use_handle) {                                        /* run_dense2D_withHeading.m:77 */
    r[0] = xnk[2] - xni[2] - odo[2];
  } else {                                                        /* particleSmoother.m:176-177 */
    for (int c = 0; c < nw; ++c) r[c] = xnk[c] - xni[c] - odo[c];
  }
  for (int q = 0; q < nw * nw; ++q) A[q] = dt * Q[q];
  if (chol_lower(A, nw, L)) return 1;
  for (int q = nw - 1; q >= 0; --q) {
    real s = r[q];
    for (int k = q + 1; k < nw; ++k) s -= L[k + nw * q] * eDyn[k];
    eDyn[q] = s / L[q + nw * q];
  }
  return 0;
}

/* in-place lower Cholesky of the n x n column-major matrix A (only the lower triangle is read); 0 ok */
static int chol_inplace(real* A, int n) {
  for (int j = 0; j < n; ++j) {
    real s = A[j + (size_t)n * j];
    for (int k = 0; k < j; ++k) s -= A[j + (size_t)n * k] * A[j + (size_t)n * k];
    if (!(s > 0.0)) return j + 1;
    const real ljj = RM(sqrt)(s);
    A[j + (size_t)n * j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      real v = A[i + (size_t)n * j];
      for (int k = 0; k < j; ++k) v -= A[i + (size_t)n * k] * A[j + (size_t)n * k];
      A[i + (size_t)n * j] = v / ljj;
    }
  }
  return 0;
}

/* One Kalman / weight step of particle i shared by both smoothers (particleSmoother.m:266-294,305-340).
 * Returns logw; updates xl, P in place when `update`. */
static int smoother_meas(int n, int d, const real* Hi, real* Pi, real* xli, const real* yt, const double* R,
                         real jitter, real* work /* 3*d*n */, real* logw_out, int update, real* SS_out, real* cS_out) {
  real e[8], SS[64], cS[64], v[8];
  real* HP = work; real* Mm = work + (size_t)d * n; real* K = Mm + (size_t)d * n;
  if (innovation(n, d, Hi, Pi, xli, yt, R, jitter, HP, e, SS, cS)) return 1;
  if (logw_out) {
    real sl = 0.0, vv = 0.0;
    for (int a = 0; a < d; ++a) {
      real s = e[a];
      for (int k = 0; k < a; ++k) s -= cS[a + d * k] * v[k];
      v[a] = s / cS[a + d * a];
      sl += RM(log)(cS[a + d * a]); vv += v[a] * v[a];
    }
    *logw_out = -sl - 0.5 * vv - 0.5 * d * RM(log)(2 * R_PI);
  }
  if (SS_out) memcpy(SS_out, SS, sizeof(real) * d * d);
  if (cS_out) memcpy(cS_out, cS, sizeof(real) * d * d);
  if (!update) return 0;
  for (int r = 0; r < n; ++r) {                                   /* (dyi'/cS')/cS */
    real u[8], kk[8];
    for (int a = 0; a < d; ++a) { real s = Hi[a + (size_t)d * r]; for (int k = 0; k < a; ++k) s -= cS[a + d * k] * u[k]; u[a] = s / cS[a + d * a]; }
    for (int a = d - 1; a >= 0; --a) { real s = u[a]; for (int k = a + 1; k < d; ++k) s -= cS[k + d * a] * kk[k]; kk[a] = s / cS[a + d * a]; }
    for (int a = 0; a < d; ++a) Mm[r + (size_t)n * a] = kk[a];
  }
  for (int a = 0; a < d; ++a) for (int r = 0; r < n; ++r) K[r + (size_t)n * a] = 0.0;
  for (int a = 0; a < d; ++a)
    for (int c = 0; c < n; ++c) {
      const real mv = Mm[c + (size_t)n * a];
      const real* Pc = Pi + (size_t)n * c;
      for (int r = 0; r < n; ++r) K[r + (size_t)n * a] += Pc[r] * mv;
    }
  for (int r = 0; r < n; ++r) { real s = 0.0; for (int a = 0; a < d; ++a) s += K[r + (size_t)n * a] * e[a]; xli[r] += s; }
  real* KS = HP;                                                /* K*SS (HP is free now) */
  for (int b = 0; b < d; ++b)
    for (int r = 0; r < n; ++r) { real s = 0.0; for (int a = 0; a < d; ++a) s += K[r + (size_t)n * a] * SS[a + d * b]; KS[r + (size_t)n * b] = s; }
  for (int c = 0; c < n; ++c) {
    real* Pc = Pi + (size_t)n * c;
    for (int r = 0; r < n; ++r) { real s = 0.0; for (int a = 0; a < d; ++a) s += KS[r + (size_t)n * a] * K[c + (size_t)n * a]; Pc[r] -= s; }
  }
  return 0;
}

int rbpf_oracle_particle_smoother(const rbpf_model* model, const rbpf_problem* p, const rbpf_rng* rng, const rbpf_options* opt,
                                  int N_K, int info_form, rbpf_smoother_out* out, int n_threads, double* loop_seconds) {
  if (!model || !p || !rng || !out || rng->mode != RBPF_RNG_REPLAY || !rng->Ufin || N_K < 1 || rng->n_iter < N_K) return RBPF_ERR_INVALID_ARG;
  omodel M;
  M.kind = model->kind; M.m = model->m_basis; M.dim = model->dim; M.NN = model->NN;
  M.nN = p->n_nonlin; M.n = p->n_lin; M.d = p->n_y; M.nw = p->n_w; M.nodo = p->n_odo;
  for (int a = 0; a < 3; ++a) M.L[a] = model->L[a];
  const int N = p->N_P, T = p->N_T, nN = M.nN, n = M.n, d = M.d, nw = M.nw;
  if (d > 3 || nN > 8 || nw > 8) return RBPF_ERR_UNSUPPORTED;
  if (!model->use_dyn_res_norm && nw != nN) return RBPF_ERR_INVALID_ARG;
  if (info_form && p->x0_lin_cols != 1) return RBPF_ERR_INVALID_ARG;                    /* quirk Q5 */
  const real jitter = (opt && opt->jitter > 0) ? opt->jitter : 1e-2;                  /* particleSmoother.m:70 */
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
  const int nthr = omp_get_max_threads();
#else
  (void)n_threads;
  const int nthr = 1;
#endif
  const size_t nn2 = (size_t)n * n;
  const int Mmax = d * T;
  int status = RBPF_OK;
  real* w = malloc(sizeof(real) * N), *logw = malloc(sizeof(real) * N), *paNtLog = malloc(sizeof(real) * N), *paNt = malloc(sizeof(real) * N);
  real* xn = malloc(sizeof(real) * nN * N), *xn_pred = malloc(sizeof(real) * nN * N);
  real* xl = malloc(sizeof(real) * (size_t)n * N), *xl_pred = malloc(sizeof(real) * (size_t)n * N);
  real* P = malloc(sizeof(real) * nn2 * N), *P_pred = malloc(sizeof(real) * nn2 * N);
  real* ivec = info_form ? malloc(sizeof(real) * (size_t)n * N) : NULL, *ivec_pred = info_form ? malloc(sizeof(real) * (size_t)n * N) : NULL;
  real* Imat = info_form ? malloc(sizeof(real) * nn2 * N) : NULL, *Imat_pred = info_form ? malloc(sizeof(real) * nn2 * N) : NULL;
  real* hld = malloc(sizeof(real) * N), *hld2 = malloc(sizeof(real) * N);
  real* H = malloc(sizeof(real) * (size_t)d * n * N);
  real* xn_traj = calloc((size_t)nN * N * T, sizeof(real)), *trj2 = malloc(sizeof(real) * (size_t)nN * N);
  real* xnk = calloc((size_t)nN * T, sizeof(real));
  real* dy_xnk = malloc(sizeof(real) * (size_t)d * n * T);                           /* H along the reference trajectory: [T][d x n] */
  real* ImatAddt = info_form ? calloc(nn2, sizeof(real)) : NULL, *ivecAddt = info_form ? calloc(n, sizeof(real)) : NULL;
  int* ai = calloc(N, sizeof(int));
  real* wc_all = malloc(sizeof(real) * (size_t)N * nthr);
  /* per-thread scratch: 3*d*n for the Kalman step; covariance form: G [Mmax x n], S [Mmax x Mmax], e [Mmax]; info form: A [n x n], v [n], Pi [n] */
  const size_t per = (size_t)3 * d * n + (info_form ? nn2 + 2 * (size_t)n : (size_t)Mmax * n + (size_t)Mmax * Mmax + Mmax);
  real* scratch = malloc(sizeof(real) * per * nthr);
  real Rinv[9], halfLogDetR = 0.0;
  {
    real Lr[9], yv[3], xv[3];
    real Rr[9];
    cp_in(Rr, p->R, (size_t)d * d);
    if (chol_lower(Rr, d, Lr)) { status = RBPF_ERR_CHOL_FAILED; goto done; }
    for (int j = 0; j < d; ++j) halfLogDetR += RM(log)(Lr[j + d * j]);
    for (int col = 0; col < d; ++col) {
      for (int i = 0; i < d; ++i) { real v = (i == col); for (int k = 0; k < i; ++k) v -= Lr[i + d * k] * yv[k]; yv[i] = v / Lr[i + d * i]; }
      for (int i = d - 1; i >= 0; --i) { real v = yv[i]; for (int k = i + 1; k < d; ++k) v -= Lr[k + d * i] * xv[k]; xv[i] = v / Lr[i + d * i]; }
      for (int i = 0; i < d; ++i) Rinv[i + d * col] = xv[i];
    }
  }
  if (!w || !logw || !paNtLog || !paNt || !xn || !xn_pred || !xl || !xl_pred || !P || !P_pred || !hld || !hld2 || !H || !xn_traj || !trj2 ||
      !xnk || !dy_xnk || !ai || !wc_all || !scratch || (info_form && (!ivec || !ivec_pred || !Imat || !Imat_pred || !ImatAddt || !ivecAddt))) {
    status = RBPF_ERR_OUT_OF_MEMORY; goto done;
  }
  const real t_start = now_s();
  for (int k = 0; k < N_K; ++k) {
    const double* Uk = rng->U + (size_t)k * N * (T - 1);
    const double* Zk = rng->Z + (size_t)k * N * (T - 1) * nw;
    /* initialisation :92-118 */
    real hld0 = 0.0;
    for (int r = 0; r < n; ++r) hld0 += RM(log)(RM(sqrt)(p->P0_lin[r + (size_t)n * r]));
    for (int i = 0; i < N; ++i) {
      w[i] = 1.0 / N; logw[i] = RM(log)(w[i]);
      for (int c = 0; c < nN; ++c) xn[c + (size_t)nN * i] = p->x0_nonlin[c];
      cp_in(xl + (size_t)n * i, p->x0_lin + (size_t)n * (p->x0_lin_cols > 1 ? i : 0), (size_t)n);
      cp_in(P + nn2 * i, p->P0_lin, nn2);
      if (info_form) {                                                                  /* :110-115, quirk Q5 */
        memset(Imat + nn2 * i, 0, sizeof(real) * nn2);
        for (int r = 0; r < n; ++r) {
          const real pd = p->P0_lin[r + (size_t)n * r];
          Imat[nn2 * i + r + (size_t)n * r] = 1.0 / pd;
          ivec[(size_t)n * i + r] = (1.0 / pd) * p->x0_lin[r];
        }
        hld[i] = hld0;
      }
    }
    if (k > 0) {
      for (int c = 0; c < nN; ++c) xn[c + (size_t)nN * (N - 1)] = xnk[c];               /* :103-107 */
      for (int t = 0; t < T; ++t) memcpy(xn_traj + (size_t)nN * N * t + (size_t)nN * (N - 1), xnk + (size_t)nN * t, sizeof(real) * nN);   /* :112-114 */
      for (int t = 0; t < T; ++t) meas_model(&M, xnk + (size_t)nN * t, dy_xnk + (size_t)d * n * t);   /* :119-121 */
      if (info_form) {                                                                  /* :132-146, jj = 1..T in order */
        memset(ImatAddt, 0, sizeof(real) * nn2); memset(ivecAddt, 0, sizeof(real) * n);
        for (int jj = 0; jj < T; ++jj) {
          const real* Hj = dy_xnk + (size_t)d * n * jj;
          real Riy[3];
          for (int a = 0; a < d; ++a) { real s = 0.0; for (int b = 0; b < d; ++b) s += Rinv[a + d * b] * p->y[jj + (size_t)T * b]; Riy[a] = s; }
          for (int c = 0; c < n; ++c) {
            real RiH[3];
            for (int a = 0; a < d; ++a) { real s = 0.0; for (int b = 0; b < d; ++b) s += Rinv[a + d * b] * Hj[b + (size_t)d * c]; RiH[a] = s; }
            for (int r = 0; r < n; ++r) { real s = 0.0; for (int a = 0; a < d; ++a) s += Hj[a + (size_t)d * r] * RiH[a]; ImatAddt[r + (size_t)n * c] += s; }
          }
          for (int r = 0; r < n; ++r) { real s = 0.0; for (int a = 0; a < d; ++a) s += Hj[a + (size_t)d * r] * Riy[a]; ivecAddt[r] += s; }
        }
      }
    }
    memcpy(xn_traj, xn, sizeof(real) * nN * N);                                       /* :117 */
    for (int t = 0; t < T; ++t) {
      real yt[3];
      for (int a = 0; a < d; ++a) yt[a] = p->y[t + (size_t)T * a];
      if (t != 0) {
        const real dtt = p->dt[p->dt_len > 1 ? t - 1 : 0];
        const double* Qt = p->Q + (size_t)(p->q_pages > 1 ? t - 1 : 0) * nw * nw;
        real odo[8];
        for (int q = 0; q < M.nodo; ++q) odo[q] = p->odometry[(t - 1) + (size_t)p->odo_ld * q];
        const double* U = Uk + (size_t)(t - 1) * N;
        const double* Z = Zk + (size_t)(t - 1) * N * nw;
        const int n_ord = (k == 0) ? N : N - 1;                                          /* :132-137, :149-152 */
        int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
        for (int i = 0; i < n_ord; ++i) {
          int tid = 0;
#ifdef _OPENMP
          tid = omp_get_thread_num();
#endif
          int a = sample_ref(w, N, U[i], wc_all + (size_t)N * tid);
          if (a >= N) a = N - 1;
          ai[i] = a;
          bad |= dyn_model(&M, xn + (size_t)nN * a, odo, dtt, Qt, Z + (size_t)nw * i, xn_pred + (size_t)nN * i);
        }
        if (bad) { status = RBPF_ERR_CHOL_FAILED; goto done; }
        if (k > 0) {                                                                     /* :156-245 */
          if (info_form) {                                                               /* the (t-1) term leaves the sums :194-201 */
            const real* Hj = dy_xnk + (size_t)d * n * (t - 1);
            real Riy[3];
            for (int a = 0; a < d; ++a) { real s = 0.0; for (int b = 0; b < d; ++b) s += Rinv[a + d * b] * p->y[(t - 1) + (size_t)T * b]; Riy[a] = s; }
            for (int c = 0; c < n; ++c) {
              real RiH[3];
              for (int a = 0; a < d; ++a) { real s = 0.0; for (int b = 0; b < d; ++b) s += Rinv[a + d * b] * Hj[b + (size_t)d * c]; RiH[a] = s; }
              for (int r = 0; r < n; ++r) { real s = 0.0; for (int a = 0; a < d; ++a) s += Hj[a + (size_t)d * r] * RiH[a]; ImatAddt[r + (size_t)n * c] -= s; }
            }
            for (int r = 0; r < n; ++r) { real s = 0.0; for (int a = 0; a < d; ++a) s += Hj[a + (size_t)d * r] * Riy[a]; ivecAddt[r] -= s; }
          }
          const real* xnkt = xnk + (size_t)nN * t;
          const int Mt = d * (T - t);                                                    /* ny*(N_T-t+1) in 1-based t */
          bad = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : bad)
          for (int i = 0; i < N; ++i) {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            real* wk = scratch + per * tid + (size_t)3 * d * n;
            real eDyn[8], logwDyn = 0.0, logwMeas;
            if (dyn_res_norm(&M, model->use_dyn_res_norm, xnkt, xn + (size_t)nN * i, odo, dtt, Qt, eDyn)) { bad |= 1; continue; }
            for (int q = 0; q < nw; ++q) logwDyn += eDyn[q] * eDyn[q];
            logwDyn *= -0.5;                                                             /* :182 */
            const real* Pi = P + nn2 * i;
            const real* xli = xl + (size_t)n * i;
            if (!info_form) {                                                            /* :191-229 */
              real* G = wk; real* S = G + (size_t)Mt * n; real* e = S + (size_t)Mt * Mt;
              /* dy = rows (tt, a) = H_{xnk(:,tt)}(a,:), tt = t..T-1 : G = dy * P_i */
              for (int q = 0; q < Mt; ++q) {
                const real* Hrow = dy_xnk + (size_t)d * n * (t + q / d) + (q % d);     /* element (a, c) at Hrow[d*c] */
                for (int c = 0; c < n; ++c) {
                  const real* Pc = Pi + (size_t)n * c;
                  real s = 0.0;
                  for (int r = 0; r < n; ++r) s += Hrow[(size_t)d * r] * Pc[r];
                  G[q + (size_t)Mt * c] = s;
                }
                real sx = 0.0;
                for (int c = 0; c < n; ++c) sx += Hrow[(size_t)d * c] * xli[c];
                e[q] = p->y[(t + q / d) + (size_t)T * (q % d)] - sx;                      /* :192-193 */
              }
              for (int q2 = 0; q2 < Mt; ++q2) {
                const real* Hrow2 = dy_xnk + (size_t)d * n * (t + q2 / d) + (q2 % d);
                for (int q = q2; q < Mt; ++q) {
                  real s = 0.0;
                  for (int c = 0; c < n; ++c) s += G[q + (size_t)Mt * c] * Hrow2[(size_t)d * c];
                  if (q / d == q2 / d) s += p->R[(q % d) + d * (q2 % d)];                 /* kron(eye, R) */
                  S[q + (size_t)Mt * q2] = s;
                }
              }
              /* keep a copy of the lower triangle for the jitter retry: factor in place, on failure rebuild */
              int fl = chol_inplace(S, Mt);
              if (fl) {                                                                  /* :221-224: chol(SS + jitter*I) */
                for (int q2 = 0; q2 < Mt; ++q2) {
                  const real* Hrow2 = dy_xnk + (size_t)d * n * (t + q2 / d) + (q2 % d);
                  for (int q = q2; q < Mt; ++q) {
                    real s = 0.0;
                    for (int c = 0; c < n; ++c) s += G[q + (size_t)Mt * c] * Hrow2[(size_t)d * c];
                    if (q / d == q2 / d) s += p->R[(q % d) + d * (q2 % d)];
                    if (q == q2) s += jitter;
                    S[q + (size_t)Mt * q2] = s;
                  }
                }
                if (chol_inplace(S, Mt)) { bad |= 1; continue; }
              }
              real sl = 0.0, vv = 0.0;
              for (int q = 0; q < Mt; ++q) {                                             /* v = cS \ e */
                real s = e[q];
                for (int c = 0; c < q; ++c) s -= S[q + (size_t)Mt * c] * e[c];
                e[q] = s / S[q + (size_t)Mt * q];
                sl += RM(log)(S[q + (size_t)Mt * q]); vv += e[q] * e[q];
              }
              logwMeas = -sl - 0.5 * vv - (real)Mt / 2.0 * RM(log)(2 * R_PI);              /* :229 */
            } else {                                                                     /* InformationForm.m:224-236 */
              real* A = wk; real* v = A + nn2; real* Pv = v + n;
              const real* Ii = Imat + nn2 * i; const real* iv = ivec + (size_t)n * i;
              for (size_t q = 0; q < nn2; ++q) A[q] = Ii[q] + ImatAddt[q];
              for (int r = 0; r < n; ++r) v[r] = iv[r] + ivecAddt[r];
              if (chol_inplace(A, n)) { bad |= 1; continue; }                            /* quirk Q4: the reference's retry is unusable */
              real sl = 0.0, vv = 0.0;
              for (int q = 0; q < n; ++q) {
                real s = v[q];
                for (int c = 0; c < q; ++c) s -= A[q + (size_t)n * c] * v[c];
                v[q] = s / A[q + (size_t)n * q];
                sl += RM(log)(A[q + (size_t)n * q]); vv += v[q] * v[q];
              }
              real qf = 0.0;                                                           /* ivec' * P * ivec */
              for (int r = 0; r < n; ++r) Pv[r] = 0.0;
              for (int c = 0; c < n; ++c) { const real* Pc = Pi + (size_t)n * c; const real x = iv[c]; for (int r = 0; r < n; ++r) Pv[r] += Pc[r] * x; }
              for (int r = 0; r < n; ++r) qf += iv[r] * Pv[r];
              logwMeas = -0.5 * qf - hld[i] - sl + 0.5 * vv;
            }
            paNtLog[i] = RM(log)(w[i]) + logwDyn + logwMeas;                                 /* :232 */
          }
          if (bad) { status = RBPF_ERR_CHOL_FAILED; goto done; }
          real c = -INFINITY, s = 0.0;                                                 /* :236-238 */
          for (int i = 0; i < N; ++i) if (paNtLog[i] > c) c = paNtLog[i];
          for (int i = 0; i < N; ++i) s += RM(exp)(paNtLog[i] - c);
          const real lse = c + RM(log)(s);
          for (int i = 0; i < N; ++i) paNt[i] = RM(exp)(paNtLog[i] - lse);
          if (out->trace_paNt) cp_out(out->trace_paNt + ((size_t)k * T + t) * N, paNt, (size_t)(N));
          int a = sample_ref(paNt, N, U[N - 1], wc_all);                                 /* :241 */
          if (a >= N) a = N - 1;
          ai[N - 1] = a;
          memcpy(xn_pred + (size_t)nN * (N - 1), xnkt, sizeof(real) * nN);             /* :242 */
        }
        /* gather of the linear states :140-141, :243-244 (and ivec / Imat / halfLogDetP for the information form) */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) {
          memcpy(xl_pred + (size_t)n * i, xl + (size_t)n * ai[i], sizeof(real) * n);
          memcpy(P_pred + nn2 * i, P + nn2 * ai[i], sizeof(real) * nn2);
          if (info_form) {
            memcpy(ivec_pred + (size_t)n * i, ivec + (size_t)n * ai[i], sizeof(real) * n);
            memcpy(Imat_pred + nn2 * i, Imat + nn2 * ai[i], sizeof(real) * nn2);
            hld2[i] = hld[ai[i]];
          }
        }
        { real* tp; tp = xn; xn = xn_pred; xn_pred = tp; tp = xl; xl = xl_pred; xl_pred = tp; tp = P; P = P_pred; P_pred = tp; }
        if (info_form) { real* tp; tp = ivec; ivec = ivec_pred; ivec_pred = tp; tp = Imat; Imat = Imat_pred; Imat_pred = tp; tp = hld; hld = hld2; hld2 = tp; }
        memcpy(xn_traj + (size_t)nN * N * t, xn, sizeof(real) * nN * N);               /* :256-257 */
        for (int s2 = 0; s2 < t; ++s2) {
          real* hs = xn_traj + (size_t)nN * N * s2;
          for (int i = 0; i < N; ++i) memcpy(trj2 + (size_t)nN * i, hs + (size_t)nN * ai[i], sizeof(real) * nN);
          memcpy(hs, trj2, sizeof(real) * nN * N);
        }
      }
      /* importance weights (step 12) and update (step 13) */
#pragma omp parallel for schedule(static)
      for (int i = 0; i < N; ++i) meas_model(&M, xn + (size_t)nN * i, H + (size_t)d * n * i);
      int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
      for (int i = 0; i < N; ++i) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        real* work = scratch + per * tid;
        const real* Hi = H + (size_t)d * n * i;
        real* Pi = P + nn2 * i; real* xli = xl + (size_t)n * i;
        if (!info_form) {
          bad |= smoother_meas(n, d, Hi, Pi, xli, yt, p->R, jitter, work, &logw[i], 0, NULL, NULL);
        } else {                                                                        /* InformationForm.m:279-305 */
          real SS[9], cS[9], lw_dummy;
          real* iv = ivec + (size_t)n * i;
          real* Pv = work + (size_t)3 * d * n + nn2;                                   /* [n] */
          real* ivp = Pv + n;                                                          /* ivecPlus [n] */
          real qa = 0.0, qb = 0.0;
          for (int r = 0; r < n; ++r) Pv[r] = 0.0;
          for (int c = 0; c < n; ++c) { const real* Pc = Pi + (size_t)n * c; const real x = iv[c]; for (int r = 0; r < n; ++r) Pv[r] += Pc[r] * x; }
          for (int r = 0; r < n; ++r) qa += iv[r] * Pv[r];                               /* ivec' P ivec with the PRIOR P */
          if (smoother_meas(n, d, Hi, Pi, xli, yt, p->R, jitter, work, &lw_dummy, 0, SS, cS)) { bad |= 1; continue; }
          real Riy[3];
          for (int a = 0; a < d; ++a) { real s = 0.0; for (int b = 0; b < d; ++b) s += Rinv[a + d * b] * yt[b]; Riy[a] = s; }
          for (int r = 0; r < n; ++r) { real s = iv[r]; for (int a = 0; a < d; ++a) s += Hi[a + (size_t)d * r] * Riy[a]; ivp[r] = s; }   /* :292 */
          /* K = P*((dyi'/cS')/cS) (:293); Pplus = P - K*SS*K' (:294), formed explicitly as the reference does */
          real* K = work + (size_t)2 * d * n;                                          /* n x d */
          real* Mm = work + (size_t)d * n;
          real* KS = work;                                                             /* n x d */
          real* Pplus = work + (size_t)3 * d * n;                                      /* n x n */
          for (int r = 0; r < n; ++r) {
            real u[3] = {0.0, 0.0, 0.0}, kk[3] = {0.0, 0.0, 0.0};
            for (int a = 0; a < d; ++a) { real s = Hi[a + (size_t)d * r]; for (int q = 0; q < a; ++q) s -= cS[a + d * q] * u[q]; u[a] = s / cS[a + d * a]; }
            for (int a = d - 1; a >= 0; --a) { real s = u[a]; for (int q = a + 1; q < d; ++q) s -= cS[q + d * a] * kk[q]; kk[a] = s / cS[a + d * a]; }
            for (int a = 0; a < d; ++a) Mm[r + (size_t)n * a] = kk[a];
          }
          for (int a = 0; a < d; ++a) for (int r = 0; r < n; ++r) K[r + (size_t)n * a] = 0.0;
          for (int a = 0; a < d; ++a)
            for (int c = 0; c < n; ++c) { const real mv = Mm[c + (size_t)n * a]; const real* Pc = Pi + (size_t)n * c; for (int r = 0; r < n; ++r) K[r + (size_t)n * a] += Pc[r] * mv; }
          for (int b = 0; b < d; ++b)
            for (int r = 0; r < n; ++r) { real s = 0.0; for (int a = 0; a < d; ++a) s += K[r + (size_t)n * a] * SS[a + d * b]; KS[r + (size_t)n * b] = s; }
          for (int c = 0; c < n; ++c) {
            const real* Pc = Pi + (size_t)n * c; real* Qc = Pplus + (size_t)n * c;
            for (int r = 0; r < n; ++r) { real s = 0.0; for (int a = 0; a < d; ++a) s += KS[r + (size_t)n * a] * K[c + (size_t)n * a]; Qc[r] = Pc[r] - s; }
          }
          for (int r = 0; r < n; ++r) Pv[r] = 0.0;
          for (int c = 0; c < n; ++c) { const real* Qc = Pplus + (size_t)n * c; const real x = ivp[c]; for (int r = 0; r < n; ++r) Pv[r] += Qc[r] * x; }
          for (int r = 0; r < n; ++r) qb += ivp[r] * Pv[r];                              /* ivecPlus' * Pplus * ivecPlus (:303) */
          real sl = 0.0;
          for (int a = 0; a < d; ++a) sl += RM(log)(cS[a + d * a]);
          const real hldp = -sl + halfLogDetR + hld[i];                                /* :298 */
          real yRy = 0.0;
          for (int a = 0; a < d; ++a) yRy += yt[a] * Riy[a];
          logw[i] = -0.5 * qa - hld[i] + hldp + 0.5 * qb - 0.5 * yRy - (0.5 * d * RM(log)(2 * R_PI) + halfLogDetR);   /* :301-304 */
          hld2[i] = hldp;
        }
      }
      if (bad) { status = RBPF_ERR_CHOL_FAILED; goto done; }
      if (info_form) { real* tp = hld; hld = hld2; hld2 = tp; }                         /* :308 */
      {
        real c = -INFINITY, s = 0.0;
        for (int i = 0; i < N; ++i) if (logw[i] > c) c = logw[i];
        for (int i = 0; i < N; ++i) s += RM(exp)(logw[i] - c);
        const real lse = c + RM(log)(s);
        for (int i = 0; i < N; ++i) w[i] = RM(exp)(logw[i] - lse);
      }
      if (out->trace_logw) cp_out(out->trace_logw + ((size_t)k * T + t) * N, logw, (size_t)(N));
      if (out->trace_w) cp_out(out->trace_w + ((size_t)k * T + t) * N, w, (size_t)(N));
      if (out->trace_ai) for (int i = 0; i < N; ++i) out->trace_ai[((size_t)k * T + t) * N + i] = ai[i];
      bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
      for (int i = 0; i < N; ++i) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        real* work = scratch + per * tid;
        const real* Hi = H + (size_t)d * n * i;
        bad |= smoother_meas(n, d, Hi, P + nn2 * i, xl + (size_t)n * i, yt, p->R, jitter, work, NULL, 1, NULL, NULL);
        if (info_form) {                                                                /* :333-334 */
          real* iv = ivec + (size_t)n * i; real* Ii = Imat + nn2 * i;
          real Riy[3];
          for (int a = 0; a < d; ++a) { real s = 0.0; for (int b = 0; b < d; ++b) s += Rinv[a + d * b] * yt[b]; Riy[a] = s; }
          for (int r = 0; r < n; ++r) { real s = 0.0; for (int a = 0; a < d; ++a) s += Hi[a + (size_t)d * r] * Riy[a]; iv[r] += s; }
          for (int c = 0; c < n; ++c) {
            real RiH[3];
            for (int a = 0; a < d; ++a) { real s = 0.0; for (int b = 0; b < d; ++b) s += Rinv[a + d * b] * Hi[b + (size_t)d * c]; RiH[a] = s; }
            for (int r = 0; r < n; ++r) { real s = 0.0; for (int a = 0; a < d; ++a) s += Hi[a + (size_t)d * r] * RiH[a]; Ii[r + (size_t)n * c] += s; }
          }
        }
      }
      if (bad) { status = RBPF_ERR_CHOL_FAILED; goto done; }
    }
    /* ak = sample(w); xnk = xn_traj(:,ak,:) :346-354 */
    int ak = sample_ref(w, N, rng->Ufin[k], wc_all);
    if (ak >= N) ak = N - 1;
    for (int t = 0; t < T; ++t) memcpy(xnk + (size_t)nN * t, xn_traj + (size_t)nN * N * t + (size_t)nN * ak, sizeof(real) * nN);
    if (out->XNK) cp_out(out->XNK + (size_t)k * nN * T, xnk, (size_t)(nN * T));
    if (out->XLK) cp_out(out->XLK + (size_t)k * n, xl + (size_t)n * ak, (size_t)(n));
    if (out->PK) cp_out(out->PK + (size_t)k * nn2, P + nn2 * ak, (size_t)(nn2));
    if (out->trace_ak) out->trace_ak[k] = ak;
  }
  if (loop_seconds) *loop_seconds = now_s() - t_start;
done:
  free(w); free(logw); free(paNtLog); free(paNt); free(xn); free(xn_pred); free(xl); free(xl_pred); free(P); free(P_pred);
  free(ivec); free(ivec_pred); free(Imat); free(Imat_pred); free(hld); free(hld2); free(H); free(xn_traj); free(trj2); free(xnk);
  free(dy_xnk); free(ImatAddt); free(ivecAddt); free(ai); free(wc_all); free(scratch);
  return status;
}

int rbpf_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
