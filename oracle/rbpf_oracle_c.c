/*
 * ORACLE -- TEST / BASELINE INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C restatement of src/particleFilter.m (dense branch) for the two dense model families,
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

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* tools/sample.m:30-32: wc = cumsum(w); ind = sum(wc < u) + 1  (0-based here) */
static int sample_ref(const double* w, int N, double u, double* wc) {
  double run = 0.0;
  int cnt = 0;
  for (int j = 0; j < N; ++j) { run += w[j]; wc[j] = run; }
  for (int j = 0; j < N; ++j) cnt += (wc[j] < u);
  return cnt;
}

/* chol(A,'lower') for small/medium n, column-major; returns 0 on success */
static int chol_lower(const double* A, int n, double* L) {
  memset(L, 0, sizeof(double) * (size_t)n * n);
  for (int j = 0; j < n; ++j) {
    double s = A[j + (size_t)n * j];
    for (int k = 0; k < j; ++k) s -= L[j + (size_t)n * k] * L[j + (size_t)n * k];
    if (!(s > 0.0)) return j + 1;
    const double ljj = sqrt(s);
    L[j + (size_t)n * j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double v = A[i + (size_t)n * j];
      for (int k = 0; k < j; ++k) v -= L[i + (size_t)n * k] * L[j + (size_t)n * k];
      L[i + (size_t)n * j] = v / ljj;
    }
  }
  return 0;
}

/* tools/qLeft.m:30-35 applied to p */
static void qleft_mul(const double* q, const double* p, double* r) {
  r[0] = q[0] * p[0] - q[1] * p[1] - q[2] * p[2] - q[3] * p[3];
  r[1] = q[1] * p[0] + q[0] * p[1] - q[3] * p[2] + q[2] * p[3];
  r[2] = q[2] * p[0] + q[3] * p[1] + q[0] * p[2] - q[1] * p[3];
  r[3] = q[3] * p[0] - q[2] * p[1] + q[1] * p[2] + q[0] * p[3];
}

/* tools/expq.m:22-31 */
static void expq(const double* phi, double* eq) {
  const double mag = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
  const double den = mag + (mag == 0.0 ? 1.0 : 0.0);
  eq[0] = cos(mag);
  for (int k = 0; k < 3; ++k) eq[1 + k] = phi[k] / den * sin(mag);
  if (eq[0] < 0.0) for (int k = 0; k < 4; ++k) eq[k] = -eq[k];
}

/* tools/quat2rmat.m:27-33, R[row*3+col] */
static void quat2rmat(const double* q, double* R) {
  const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  R[0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3; R[1] = 2 * q1 * q2 - 2 * q0 * q3; R[2] = 2 * q1 * q3 + 2 * q0 * q2;
  R[3] = 2 * q1 * q2 + 2 * q0 * q3; R[4] = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3; R[5] = 2 * q2 * q3 - 2 * q0 * q1;
  R[6] = 2 * q1 * q3 - 2 * q0 * q2; R[7] = 2 * q2 * q3 + 2 * q0 * q1; R[8] = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
}

typedef struct {
  int kind, m, dim, nN, n, d, nw, nodo;
  const int32_t* NN;
  double L[3];
} omodel;

/* dynModel closures: run_dense3D_magfield.m:301-308 / run_dense2D_withHeading.m:75-76 */
static int dyn_model(const omodel* M, const double* x, const double* odo, double dt, const double* Q, const double* z,
                     double* xp) {
  if (M->kind == RBPF_MODEL_DENSE_MAG_6D) {
    double A[9], Lp[9], La[9], phi[3], eq[4], dq[4];
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) A[r + 3 * c] = dt * Q[r + 6 * c];
    if (chol_lower(A, 3, Lp)) return 1;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) A[r + 3 * c] = dt * Q[(3 + r) + 6 * (3 + c)];
    if (chol_lower(A, 3, La)) return 1;
    for (int r = 0; r < 3; ++r) {
      double s = 0.0;
      for (int c = 0; c < 3; ++c) s += Lp[r + 3 * c] * z[c];
      xp[r] = x[r] + odo[r] + s;
      double a = 0.0;
      for (int c = 0; c < 3; ++c) a += La[r + 3 * c] * z[3 + c];
      phi[r] = a;
    }
    expq(phi, eq);
    qleft_mul(&odo[3], eq, dq);
    qleft_mul(&x[3], dq, &xp[3]);
  } else {
    const double c = cos(x[2]), s = sin(x[2]);
    if (!(dt * Q[0] > 0.0)) return 1;
    xp[0] = x[0] + (c * odo[0] + s * odo[1]);
    xp[1] = x[1] + (-s * odo[0] + c * odo[1]);
    xp[2] = x[2] + odo[2] + sqrt(dt * Q[0]) * z[0];
  }
  return 0;
}

/* measModel closures -> H [d x n] column-major.  run_dense3D_magfield.m:265-279 with
 * tools/domain_cartesian_dx.m:142-170; run_dense2D_withHeading.m:168 with :84-93. */
static void meas_model(const omodel* M, const double* x, double* H) {
  const int m = M->m, d = M->d;
  if (M->kind == RBPF_MODEL_DENSE_MAG_6D) {
    double R[9];
    quat2rmat(&x[3], R);
    for (int c = 0; c < M->n; ++c) {
      double g[3];
      if (c < 3) { g[0] = (c == 0); g[1] = (c == 1); g[2] = (c == 2); }
      else {
        const int j = c - 3;
        for (int di = 0; di < 3; ++di) {
          double v = 1.0;
          for (int a = 0; a < 3; ++a) {
            const double La = M->L[a];
            const double nn = (double)M->NN[j + (size_t)m * a];
            const double arg = M_PI * nn * (x[a] + La) / (2.0 * La);
            if (a == di) v = v * M_PI * nn / (2.0 * La * sqrt(La)) * cos(arg);
            else v = v * 1.0 / sqrt(La) * sin(arg);
          }
          g[di] = v;
        }
      }
      for (int k = 0; k < 3; ++k) H[k + (size_t)d * c] = R[0 + k] * g[0] + R[3 + k] * g[1] + R[6 + k] * g[2];
    }
  } else {
    for (int c = 0; c < M->n; ++c) {
      double v = 1.0;
      for (int a = 0; a < 2; ++a) {
        const double La = M->L[a];
        const double nn = (double)M->NN[c + (size_t)m * a];
        v = v * 1.0 / sqrt(La) * sin(M_PI * nn * (x[a] + La) / (2.0 * La));
      }
      H[c] = v;
    }
  }
}

/* e = y - H xl ; SS = H P H' + R ; cS = chol (jitter retry) -- particleFilter.m:139-148.
 * HP is scratch [d x n].  Returns 0 ok, 1 failed twice. */
static int innovation(int n, int d, const double* H, const double* P, const double* xl, const double* y, const double* R,
                      double jitter, double* HP, double* e, double* SS, double* cS) {
  for (int k = 0; k < d; ++k) {
    double s = 0.0;
    for (int c = 0; c < n; ++c) s += H[k + (size_t)d * c] * xl[c];
    e[k] = y[k] - s;
  }
  /* HP = H * P  (d x n) */
  for (int c = 0; c < n; ++c) {
    const double* Pc = P + (size_t)n * c;
    for (int k = 0; k < d; ++k) {
      double s = 0.0;
      for (int r = 0; r < n; ++r) s += H[k + (size_t)d * r] * Pc[r];
      HP[k + (size_t)d * c] = s;
    }
  }
  for (int b = 0; b < d; ++b)
    for (int a = 0; a < d; ++a) {
      double s = 0.0;
      for (int c = 0; c < n; ++c) s += HP[a + (size_t)d * c] * H[b + (size_t)d * c];
      SS[a + d * b] = s + R[a + d * b];
    }
  if (chol_lower(SS, d, cS)) {
    double SJ[64];
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
  const double jitter = (opt && opt->jitter > 0) ? opt->jitter : 1e-3;            /* :89 */
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#else
  (void)n_threads;
#endif
  const size_t nn2 = (size_t)n * n;
  double* w = malloc(sizeof(double) * N), *logw = malloc(sizeof(double) * N);
  double* xn = malloc(sizeof(double) * nN * N), *xn_ = malloc(sizeof(double) * nN * N);
  double* xl = malloc(sizeof(double) * (size_t)n * N), *xl2 = malloc(sizeof(double) * (size_t)n * N);
  double* P = malloc(sizeof(double) * nn2 * N), *P2 = malloc(sizeof(double) * nn2 * N);
  double* H = malloc(sizeof(double) * (size_t)d * n * N);
  int* ai = calloc(N, sizeof(int));
  double* hist = keep_hist ? calloc((size_t)nN * N * T, sizeof(double)) : NULL;
  double* hist2 = keep_hist ? malloc(sizeof(double) * (size_t)nN * N) : NULL;
  int nthr = 1;
#ifdef _OPENMP
  nthr = omp_get_max_threads();
#endif
  double* wc_all = malloc(sizeof(double) * (size_t)N * nthr);
  double* scratch = malloc(sizeof(double) * ((size_t)d * n * 2 + nn2) * nthr);   /* HP, M (n x d), K S (n x d) reuse */
  int status = RBPF_OK, iw_max = 0;
  if (!w || !logw || !xn || !xn_ || !xl || !xl2 || !P || !P2 || !H || !ai || !wc_all || !scratch || (keep_hist && (!hist || !hist2))) {
    status = RBPF_ERR_OUT_OF_MEMORY; goto done;
  }
  for (int i = 0; i < N; ++i) {                                                     /* :55-67 */
    w[i] = 1.0 / N; logw[i] = log(w[i]);
    for (int c = 0; c < nN; ++c) xn[c + (size_t)nN * i] = p->x0_nonlin[c];
    memcpy(xl + (size_t)n * i, p->x0_lin + (size_t)n * (p->x0_lin_cols > 1 ? i : 0), sizeof(double) * n);
    memcpy(P + nn2 * i, p->P0_lin, sizeof(double) * nn2);
  }
  if (keep_hist) memcpy(hist, xn, sizeof(double) * nN * N);                        /* :96 */
  if (out->traj_max) for (size_t q = 0; q < (size_t)nN * T; ++q) out->traj_max[q] = NAN;
  if (out->traj_mean) for (size_t q = 0; q < (size_t)nN * T; ++q) out->traj_mean[q] = NAN;

  const double t0 = now_s();
  for (int t = 0; t < T; ++t) {                                                     /* :100 */
    const double* yt = NULL;
    double ybuf[8];
    for (int k = 0; k < d; ++k) ybuf[k] = p->y[t + (size_t)T * k];
    yt = ybuf;
    if (t != 0) {
      memcpy(xn_, xn, sizeof(double) * nN * N);                                     /* :102 */
      const double dtt = p->dt[p->dt_len > 1 ? t - 1 : 0];
      const double* Qt = p->Q + (size_t)(p->q_pages > 1 ? t - 1 : 0) * nw * nw;
      double odo[8];
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
        memcpy(xl2 + (size_t)n * i, xl + (size_t)n * ai[i], sizeof(double) * n);
        memcpy(P2 + nn2 * i, P + nn2 * ai[i], sizeof(double) * nn2);
      }
      { double* tmp = xl; xl = xl2; xl2 = tmp; tmp = P; P = P2; P2 = tmp; }
      if (keep_hist) {                                                              /* :117-118 */
        memcpy(hist + (size_t)nN * N * t, xn, sizeof(double) * nN * N);
        for (int s = 0; s < t; ++s) {
          double* hs = hist + (size_t)nN * N * s;
          for (int i = 0; i < N; ++i) memcpy(hist2 + (size_t)nN * i, hs + (size_t)nN * ai[i], sizeof(double) * nN);
          memcpy(hs, hist2, sizeof(double) * nN * N);
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
      double* HP = scratch + ((size_t)d * n * 2 + nn2) * tid;
      double e[8], SS[64], cS[64], v[8];
      if (innovation(n, d, H + (size_t)d * n * i, P + nn2 * i, xl + (size_t)n * i, yt, p->R, jitter, HP, e, SS, cS)) { bad |= 1; continue; }
      double sl = 0.0, vv = 0.0;
      for (int a = 0; a < d; ++a) {                                                 /* v = cS\e */
        double s = e[a];
        for (int k = 0; k < a; ++k) s -= cS[a + d * k] * v[k];
        v[a] = s / cS[a + d * a];
        sl += log(cS[a + d * a]); vv += v[a] * v[a];
      }
      logw[i] = -sl - 0.5 * vv - 0.5 * d * log(2 * M_PI);                           /* :150 */
    }
    if (bad) { status = RBPF_ERR_CHOL_FAILED; goto done; }
    {                                                                               /* :154-161 */
      double c = -INFINITY, s = 0.0;
      for (int i = 0; i < N; ++i) if (logw[i] > c) c = logw[i];
      for (int i = 0; i < N; ++i) s += exp(logw[i] - c);
      const double lse = c + log(s);
      double best = -1.0;
      for (int i = 0; i < N; ++i) { w[i] = exp(logw[i] - lse); if (w[i] > best) { best = w[i]; iw_max = i; } }
      for (int k = 0; k < nN; ++k) {
        double mean = 0.0;
        for (int i = 0; i < N; ++i) mean += xn[k + (size_t)nN * i] * w[i];
        if (out->traj_mean) out->traj_mean[k + (size_t)nN * t] = mean;
        if (out->traj_max) out->traj_max[k + (size_t)nN * t] = xn[k + (size_t)nN * iw_max];
      }
    }
    if (out->trace_logw) memcpy(out->trace_logw + (size_t)N * t, logw, sizeof(double) * N);
    if (out->trace_w) memcpy(out->trace_w + (size_t)N * t, w, sizeof(double) * N);
    if (out->trace_ai) for (int i = 0; i < N; ++i) out->trace_ai[i + (size_t)N * t] = ai[i];
    /* Kalman update :164-204 */
    bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int i = 0; i < N; ++i) {
      int tid = 0;
#ifdef _OPENMP
      tid = omp_get_thread_num();
#endif
      double* HP = scratch + ((size_t)d * n * 2 + nn2) * tid;
      double* Mm = HP + (size_t)d * n;       /* (dyt'/cS')/cS : n x d */
      double* K = Mm + (size_t)d * n;        /* n x d, then KS n x d reuses the tail */
      double* Hi = H + (size_t)d * n * i;
      double* Pi = P + nn2 * i;
      double* xli = xl + (size_t)n * i;
      double e[8], SS[64], cS[64];
      if (innovation(n, d, Hi, Pi, xli, yt, p->R, jitter, HP, e, SS, cS)) { bad |= 1; continue; }
      for (int r = 0; r < n; ++r) {                                                 /* row r of dyt' = H(:,r)' */
        double u[8], kk[8];
        for (int a = 0; a < d; ++a) {                                               /* / cS' */
          double s = Hi[a + (size_t)d * r];
          for (int k = 0; k < a; ++k) s -= cS[a + d * k] * u[k];
          u[a] = s / cS[a + d * a];
        }
        for (int a = d - 1; a >= 0; --a) {                                          /* / cS */
          double s = u[a];
          for (int k = a + 1; k < d; ++k) s -= cS[k + d * a] * kk[k];
          kk[a] = s / cS[a + d * a];
        }
        for (int a = 0; a < d; ++a) Mm[r + (size_t)n * a] = kk[a];
      }
      for (int a = 0; a < d; ++a)                                                   /* K = P * M  :194 */
        for (int r = 0; r < n; ++r) K[r + (size_t)n * a] = 0.0;
      for (int a = 0; a < d; ++a)
        for (int c = 0; c < n; ++c) {
          const double mv = Mm[c + (size_t)n * a];
          const double* Pc = Pi + (size_t)n * c;
          for (int r = 0; r < n; ++r) K[r + (size_t)n * a] += Pc[r] * mv;
        }
      for (int r = 0; r < n; ++r) {                                                 /* xl += K e  :197 */
        double s = 0.0;
        for (int a = 0; a < d; ++a) s += K[r + (size_t)n * a] * e[a];
        xli[r] += s;
      }
      double* KS = K + (size_t)n * d;                                               /* K*SS */
      for (int b = 0; b < d; ++b)
        for (int r = 0; r < n; ++r) {
          double s = 0.0;
          for (int a = 0; a < d; ++a) s += K[r + (size_t)n * a] * SS[a + d * b];
          KS[r + (size_t)n * b] = s;
        }
      for (int c = 0; c < n; ++c) {                                                 /* P -= (K*SS)*K'  :198 */
        double* Pc = Pi + (size_t)n * c;
        for (int r = 0; r < n; ++r) {
          double s = 0.0;
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
  if (out->xl_max) memcpy(out->xl_max, xl + (size_t)n * iw_max, sizeof(double) * n);
  if (out->P_max) memcpy(out->P_max, P + nn2 * iw_max, sizeof(double) * nn2);
  if (out->xl_mean || out->P_mean) {
    double* xm = malloc(sizeof(double) * n);
    for (int r = 0; r < n; ++r) { double s = 0.0; for (int i = 0; i < N; ++i) s += xl[r + (size_t)n * i] * w[i]; xm[r] = s; }
    if (out->xl_mean) memcpy(out->xl_mean, xm, sizeof(double) * n);
    if (out->P_mean) {                                                              /* quirk Q3: '=' in :229 */
      const int i = N - 1;
      for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r)
          out->P_mean[r + (size_t)n * c] = w[i] * (P[nn2 * i + r + (size_t)n * c] + (xm[r] - xl[r + (size_t)n * i]) * (xm[c] - xl[c + (size_t)n * i]));
    }
    free(xm);
  }
  if (out->traj_sample_iwmax && hist)
    for (int t = 0; t < T; ++t) memcpy(out->traj_sample_iwmax + (size_t)nN * t, hist + (size_t)nN * N * t + (size_t)nN * iw_max, sizeof(double) * nN);
  if (out->xn_traj && hist) memcpy(out->xn_traj, hist, sizeof(double) * (size_t)nN * N * T);
  if (out->final_xn) memcpy(out->final_xn, xn, sizeof(double) * nN * N);
  if (out->final_xl) memcpy(out->final_xl, xl, sizeof(double) * (size_t)n * N);
  if (out->final_P) memcpy(out->final_P, P, sizeof(double) * nn2 * N);
done:
  free(w); free(logw); free(xn); free(xn_); free(xl); free(xl2); free(P); free(P2); free(H); free(ai);
  free(hist); free(hist2); free(wc_all); free(scratch);
  return status;
}

int rbpf_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
