// C-ABI host side of the MI355X-native RBPF (see include/rbpf.h).  Everything here is plumbing
// around the kernels of rbpf_kernels.hip / rbpf_smoother.hip: argument validation, HBM-resident
// state, per-step launch sequence, final extraction.  There is NO CPU compute fallback: without a
// visible gfx950 device every entry point that needs one returns RBPF_ERR_NO_DEVICE.
#include "../../include/rbpf.h"
#include "rbpf_internal.hpp"
#include "rbpf_ctx.hpp"
#include "rbpf_sparse.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rbpf {

thread_local std::string g_last_error;

void set_error(const std::string& s) { g_last_error = s; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  g_last_error = buf;
  // the runtime keeps the failure as its thread's "last error" until somebody reads it: left there, the hipGetLastError() after the next
  // kernel launch of ANOTHER call would report this one again (seen as "out of memory" in a filter created after a refused smoother)
  (void)hipGetLastError();
  return (e == hipErrorOutOfMemory) ? RBPF_ERR_OUT_OF_MEMORY : RBPF_ERR_HIP;
}

static bool chol_lower_host(const double* A, int n, int lda, double* Lc, int ldl) {
  for (int j = 0; j < n; ++j) {
    double s = A[j + (size_t)lda * j];
    for (int k = 0; k < j; ++k) s -= Lc[j + (size_t)ldl * k] * Lc[j + (size_t)ldl * k];
    if (!(s > 0.0)) return false;
    const double ljj = std::sqrt(s);
    Lc[j + (size_t)ldl * j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double v = A[i + (size_t)lda * j];
      for (int k = 0; k < j; ++k) v -= Lc[i + (size_t)ldl * k] * Lc[j + (size_t)ldl * k];
      Lc[i + (size_t)ldl * j] = v / ljj;
    }
  }
  return true;
}

// chol(dt*Q) factors per step.  For dense-mag dynModel uses the two diagonal 3x3 blocks separately
// (run_dense3D_magfield.m:304-305) while dynResNorm uses the full 6x6 factor (:203).
static int build_chol_factors(const rbpf_model* model, const rbpf_problem* p, std::vector<double>& blk,
                              std::vector<double>& full, int& pages, bool* full_ok = nullptr) {
  if (full_ok) *full_ok = true;
  const int nw = p->n_w;
  const bool varying = (p->q_pages > 1) || (p->dt_len > 1);
  pages = varying ? std::max(p->N_T - 1, 1) : 1;
  blk.assign((size_t)pages * nw * nw, 0.0);
  full.assign((size_t)pages * nw * nw, 0.0);
  std::vector<double> A((size_t)nw * nw);
  for (int t = 0; t < pages; ++t) {
    const double dt = p->dt[p->dt_len > 1 ? t : 0];
    const double* Q = p->Q + (size_t)(p->q_pages > 1 ? t : 0) * nw * nw;
    for (int q = 0; q < nw * nw; ++q) A[q] = dt * Q[q];
    double* Lb = &blk[(size_t)t * nw * nw];
    double* Lf = &full[(size_t)t * nw * nw];
    bool ok = true;
    if (model->kind == RBPF_MODEL_GENERIC_DENSE) {
      // dynModel runs on the host: Q is its business.  Only the additive default of an empty dynResNorm
      // (particleSmoother.m:175-177) needs chol(dt*Q,'lower'); a failure is reported when a smoother asks for it.
      if (!chol_lower_host(A.data(), nw, nw, Lf, nw) && full_ok) *full_ok = false;
      continue;
    }
    if (model->kind == RBPF_MODEL_DENSE_MAG_6D) {
      ok = chol_lower_host(A.data(), 3, nw, Lb, nw) && chol_lower_host(A.data() + 3 + 3 * nw, 3, nw, Lb + 3 + 3 * nw, nw);
    } else if (model->kind == RBPF_MODEL_SPARSE_VISUAL_2D) {
      for (int q = 0; q < nw * nw; ++q) Lb[q] = std::sqrt(A[q]);         // sqrt(dt*Q), element-wise (pfslam.m:81)
    } else {
      ok = chol_lower_host(A.data(), nw, nw, Lb, nw);
    }
    ok = ok && chol_lower_host(A.data(), nw, nw, Lf, nw);
    if (!ok) {
      set_error("chol(dt*Q,'lower') failed: process noise covariance must be positive definite");
      return RBPF_ERR_CHOL_FAILED;
    }
  }
  return RBPF_OK;
}

// inv(R) and 0.5*log(det(R)) for the information-form recursions (particleSmootherInformationForm.m:292,298,304), d <= 3.
// Only the information-form smoother needs them; the filter accepts any R whose innovation covariance passes chol
// (with the jitter retry), so a failure here only leaves NaNs behind.
static void invert_R_small(const double* R, int d, double* Rinv, double& halfLogDetR) {
  double Lr[9] = {0};
  const bool pd = R && d <= 3 && chol_lower_host(R, d, d, Lr, d);
  halfLogDetR = pd ? 0.0 : std::nan("");
  for (int q = 0; q < d * d && q < 9; ++q) Rinv[q] = std::nan("");
  for (int j = 0; pd && j < d; ++j) halfLogDetR += std::log(Lr[j + d * j]);
  for (int col = 0; pd && col < d; ++col) {
    double y[3], x[3];
    for (int i = 0; i < d; ++i) { double v = (i == col); for (int k = 0; k < i; ++k) v -= Lr[i + d * k] * y[k]; y[i] = v / Lr[i + d * i]; }
    for (int i = d - 1; i >= 0; --i) { double v = y[i]; for (int k = i + 1; k < d; ++k) v -= Lr[k + d * i] * x[k]; x[i] = v / Lr[i + d * i]; }
    for (int i = 0; i < d; ++i) Rinv[i + d * col] = x[i];
  }
}

int fill_model_dev(const rbpf_model* model, int nN, int n, int d, int nw, int nodo, const double* R, double jitter,
                   ModelDev& M, std::vector<int>& nn_axis_major) {
  if (!model || (!model->NN && model->kind != RBPF_MODEL_SPARSE_VISUAL_2D && model->kind != RBPF_MODEL_GENERIC_DENSE)) { set_error("model / model->NN is NULL"); return RBPF_ERR_INVALID_ARG; }
  std::memset(&M, 0, sizeof(M));
  M.kind = model->kind;
  M.m = model->m_basis;
  M.dim = model->dim;
  M.nN = nN; M.n = n; M.d = d; M.nw = nw; M.nodo = nodo;
  M.use_dyn_res_norm = model->use_dyn_res_norm;
  if (model->kind == RBPF_MODEL_DENSE_MAG_6D) {
    if (model->dim != 3 || nN != 7 || d != 3 || nw != 6 || nodo != 7 || n != model->m_basis + 3) {
      set_error("dense-mag-6D expects dim=3, nNonLin=7, ny=3, nw=6, n_odo=7, nLin=m+3");
      return RBPF_ERR_INVALID_ARG;
    }
  } else if (model->kind == RBPF_MODEL_DENSE_RADIO_2DH) {
    if (model->dim != 2 || nN != 3 || d != 1 || nw != 1 || nodo != 3 || n != model->m_basis) {
      set_error("dense-radio-2D+heading expects dim=2, nNonLin=3, ny=1, nw=1, n_odo=3, nLin=m");
      return RBPF_ERR_INVALID_ARG;
    }
  } else if (model->kind == RBPF_MODEL_GENERIC_DENSE) {
    if ((d != 1 && d != 3) || nN < 1 || nN > 8 || nw < 1 || nw > 8 || n < 1) {
      set_error("generic dense family: n_y must be 1 or 3, n_nonlin and n_w <= 8"); return RBPF_ERR_UNSUPPORTED;
    }
    M.m = n; M.dim = 0; M.ktot = 0;
    for (int q = 0; q < d * d; ++q) M.R[q] = R ? R[q] : 0.0;
    invert_R_small(R, d, M.Rinv, M.halfLogDetR);
    M.jitter = jitter;
    M.logconst = -0.5 * d * std::log(2.0 * 3.14159265358979323846);
    nn_axis_major.clear();
    return RBPF_OK;
  } else if (model->kind == RBPF_MODEL_SPARSE_VISUAL_2D) {
    if (nN != 3 || nw != 3 || nodo != 3 || d != model->m_basis || n != 2 * model->m_basis || d < 1 || d > 32) {
      set_error("sparse-visual-2D expects nNonLin=3, nw=3, n_odo=3, ny = landmarks <= 32, nLin = 2*landmarks");
      return RBPF_ERR_INVALID_ARG;
    }
    for (int q = 0; q < 3; ++q) M.cam[q] = model->cam[q];
    if (!(M.cam[0] != 0.0)) { set_error("sparse-visual-2D: focal length cam[0] must be non-zero"); return RBPF_ERR_INVALID_ARG; }
    M.dim = 2;
    M.jitter = jitter;
    M.logconst = -0.5 * d * std::log(2.0 * 3.14159265358979323846);
    nn_axis_major.clear();
    return RBPF_OK;                       // R lives in device memory (ModelDev::Rdev, set by ctx_create)
  } else {
    set_error("unknown model family");
    return RBPF_ERR_UNSUPPORTED;
  }
  nn_axis_major.resize((size_t)M.m * M.dim);
  M.ktot = 0;
  for (int a = 0; a < M.dim; ++a) {
    int km = 0;
    for (int j = 0; j < M.m; ++j) {
      const int v = model->NN[j + (size_t)M.m * a];
      if (v < 1) { set_error("NN entries must be >= 1"); return RBPF_ERR_INVALID_ARG; }
      nn_axis_major[(size_t)a * M.m + j] = v;
      km = std::max(km, v);
    }
    M.kmax[a] = km;
    M.ktot += km;
    M.L[a] = model->L[a];
    if (!(M.L[a] > 0)) { set_error("domain half-widths L must be positive"); return RBPF_ERR_INVALID_ARG; }
  }
  for (int q = 0; q < d * d; ++q) M.R[q] = R ? R[q] : 0.0;
  if (d > 3) { set_error("n_y <= 3 supported by the dense families"); return RBPF_ERR_UNSUPPORTED; }
  if (R) invert_R_small(R, d, M.Rinv, M.halfLogDetR);
  M.jitter = jitter;
  M.logconst = -0.5 * d * std::log(2.0 * 3.14159265358979323846);
  return RBPF_OK;
}

static int validate_problem(const rbpf_problem* p) {
  if (!p) { set_error("problem is NULL"); return RBPF_ERR_INVALID_ARG; }
  if (p->N_P < 1 || p->N_T < 1) { set_error("N_P and N_T must be >= 1"); return RBPF_ERR_INVALID_ARG; }
  if (!p->y || !p->x0_nonlin || !p->x0_lin || !p->P0_lin || !p->Q || !p->R || !p->dt) {
    set_error("a required problem array is NULL"); return RBPF_ERR_INVALID_ARG;
  }
  if (p->N_T > 1 && !p->odometry) { set_error("odometry is NULL"); return RBPF_ERR_INVALID_ARG; }
  if (p->x0_lin_cols != 1 && p->x0_lin_cols != p->N_P) { set_error("x0_lin must be n x 1 or n x N_P"); return RBPF_ERR_INVALID_ARG; }
  if (p->q_pages != 1 && p->q_pages < p->N_T - 1) { set_error("Q must have 1 or >= N_T-1 pages"); return RBPF_ERR_INVALID_ARG; }
  if (p->dt_len != 1 && p->dt_len < p->N_T - 1) { set_error("dt must have 1 or >= N_T-1 entries"); return RBPF_ERR_INVALID_ARG; }
  if (p->n_nonlin > 8 || p->n_w > 8) { set_error("n_nonlin, n_w <= 8 supported"); return RBPF_ERR_UNSUPPORTED; }
  // the multi-workgroup resample pipeline stages one block total per 1024 particles in a 1024-entry LDS array
  if (p->N_P > kMaxParticles) { set_error("N_P above 1048576 is not supported (resample pipeline: 1024 blocks of 1024)"); return RBPF_ERR_UNSUPPORTED; }
  return RBPF_OK;
}

template <typename T>
static int dmalloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) return RBPF_OK;
  hipError_t e = hipMalloc((void**)p, count * sizeof(T));
  if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
  return RBPF_OK;
}

#define RB_TRY(x) do { int _s = (x); if (_s != RBPF_OK) return _s; } while (0)


static bool have_device() {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

static size_t bank_bytes(const Layout& L, int d, int N) {
  return ((size_t)N * (L.szT + L.szB) + (size_t)N * 2 * d * L.ldx + (size_t)N * L.ldx) * sizeof(double);
}

int ctx_create(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
               bool smoother, int N_K, rbpf_ctx** out, const CreateExtras* ex) {
  if (!out) { set_error("ctx out pointer is NULL"); return RBPF_ERR_INVALID_ARG; }
  *out = nullptr;
  RB_TRY(validate_problem(prob));
  if (!rng) { set_error("rng is NULL"); return RBPF_ERR_INVALID_ARG; }
  RB_TRY(options_ok(opt));
  if (!have_device()) { set_error("no HIP device visible: the RBPF product path has no CPU fallback"); return RBPF_ERR_NO_DEVICE; }
  rbpf_ctx* c = new rbpf_ctx();
  std::unique_ptr<rbpf_ctx, void (*)(rbpf_ctx*)> guard(c, [](rbpf_ctx* p) { ctx_free(p); });
  if (opt) c->opt = *opt; else { std::memset(&c->opt, 0, sizeof(c->opt)); c->opt.keep_history = 1; }
  if (smoother) c->opt.keep_history = 1;
  const double jitter = (c->opt.jitter > 0) ? c->opt.jitter : (smoother ? 1e-2 : 1e-3);   // quirk Q2
  std::vector<int> nn;
  RB_TRY(fill_model_dev(model, prob->n_nonlin, prob->n_lin, prob->n_y, prob->n_w, prob->n_odo, prob->R, jitter, c->mdl, nn));
  c->lay = make_layout(prob->n_lin, prob->n_y);
  c->lay_low = make_layout_low_regs(prob->n_lin, prob->n_y);
  const bool sparse = model->kind == RBPF_MODEL_SPARSE_VISUAL_2D;
  c->fp32 = c->opt.storage == 1 || c->opt.storage == 3;
  if (c->opt.storage < 0 || c->opt.storage > 3) { set_error("options.storage must be 0 (fp64), 1 (fp32), 2 (fp64, symmetric) or 3 (fp32, symmetric)"); return RBPF_ERR_INVALID_ARG; }
  if (c->opt.storage == 2 || c->opt.storage == 3) {
    // symmetric storage (lower block triangle, rbpf_step_sym.hip): the filter of the ny = 3 dense families at the
    // sizes its wave decomposition takes (eight 64-row tile rows: 512 <= nLin - nb < 640)
    if (sparse || !sym_supported(prob->n_lin, prob->n_y)) {
      set_error("symmetric storage (options.storage = 2 / 3): dense filter / smoothers (single-GPU or sharded) with ny = 3 and nLin in 259..383 or 515..639, dense filter with nLin in 1027..1151, dense-radio (ny = 1) with nLin = 128 only"); return RBPF_ERR_UNSUPPORTED;
    }
    c->lay = make_layout_sym(prob->n_lin, prob->n_y, c->fp32 ? 1 : 0);
    c->lay_low = c->lay;
    if (c->lay.CH64 == 16 && smoother) { set_error("symmetric storage at sixteen tile rows (nLin >= 1027): the filter only"); return RBPF_ERR_UNSUPPORTED; }
    if (c->fp32 && c->lay.CH64 == 4) { set_error("fp32 tiles (options.storage = 3): nLin in 515..639 or 1027..1151"); return RBPF_ERR_UNSUPPORTED; }
  }
  if (c->fp32 && (smoother || sparse || prob->n_y != 3)) {
    set_error("fp32 storage of the covariance banks: dense-mag filter only"); return RBPF_ERR_UNSUPPORTED;
  }
  {
    const int cv = c->opt.chol_variant;
    if (cv != 0 && cv != 1 && cv != 10 && cv != 11 && cv != 12 && cv != 14 && cv != 16 && cv != 64 && cv != 648 && cv != 644 && cv != 649 && cv != 128) { set_error("options.chol_variant must be 0, 1, 10, 11, 12, 14, 16, 64, 648, 644, 649 or 128"); return RBPF_ERR_INVALID_ARG; }
  }
  if (model->kind == RBPF_MODEL_GENERIC_DENSE) {
    if (ex) { set_error("generic (host-callback) models are not sharded"); return RBPF_ERR_UNSUPPORTED; }
    if (model->callbacks) {
      c->cb = *model->callbacks; c->has_cb = true;
      if (!c->cb.dyn_model || !c->cb.meas_model) { set_error("generic model: callbacks need dyn_model and meas_model"); return RBPF_ERR_INVALID_ARG; }
    } else if (smoother) { set_error("generic (host-callback) smoothers need rbpf_model.callbacks"); return RBPF_ERR_INVALID_ARG; }
    RB_TRY(dmalloc(&c->d_xn_ext, (size_t)prob->n_nonlin * prob->N_P));
    RB_TRY(dmalloc(&c->d_H_ext, (size_t)prob->N_P * prob->n_y * c->lay.ldx));
    c->h_odo.assign((size_t)std::max(prob->N_T - 1, 1) * prob->n_odo, 0.0);
    for (int t = 0; t < prob->N_T - 1; ++t) for (int k = 0; k < prob->n_odo; ++k) c->h_odo[(size_t)t * prob->n_odo + k] = prob->odometry[t + (size_t)prob->odo_ld * k];
    c->h_Q.assign(prob->Q, prob->Q + (size_t)prob->n_w * prob->n_w * prob->q_pages);
    c->h_dt.assign(prob->dt, prob->dt + prob->dt_len);
    c->q_pages = prob->q_pages; c->dt_len = prob->dt_len;
  }
  if (sparse) {
    if (c->lay.mc != 0 || sparse_step_lds_bytes(prob->n_lin, prob->n_y) > 150 * 1024) { set_error("sparse-visual-2D supports nLin <= 96"); return RBPF_ERR_UNSUPPORTED; }
    if (ex) { set_error("the sparseFeatures branch is not sharded"); return RBPF_ERR_UNSUPPORTED; }
  } else if (step_lds_bytes(c->mdl, c->lay, smoother ? 1 : 0) > 160 * 1024) { set_error("nLin too large for the LDS plan of the step kernel"); return RBPF_ERR_UNSUPPORTED; }
  c->N = prob->N_P; c->T = prob->N_T; c->smoother = smoother; c->N_K = smoother ? N_K : 1;
  c->bank_cap = (size_t)prob->N_P + (ex ? ex->bank_extra : 0);
  c->rng_slots = (ex && ex->rng_slots) ? ex->rng_slots : (size_t)prob->N_P;
  c->rng_mode = rng->mode; c->seed = rng->seed;
  const int N = c->N, T = c->T, nN = c->mdl.nN, n = c->mdl.n, d = c->mdl.d, nw = c->mdl.nw, nodo = c->mdl.nodo;
  const Layout& L = c->lay;
  HIPCHK(hipGetDevice(&c->device));
  if (const char* cuf = tuning_env("RBPF_CU_OF_32")) {   // diagnostic builds only: run on K of every 32 CUs (is a kernel bound inside the CU or by HBM?)
    uint32_t mask[8];
    const int k = atoi(cuf);
    for (int w = 0; w < 8; ++w) mask[w] = (k >= 32) ? 0xffffffffu : ((1u << k) - 1u);
    HIPCHK(hipExtStreamCreateWithCUMask(&c->stream, 8, mask));
  } else {
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  }

  // ---- model / problem constants ----
  RB_TRY(dmalloc(&c->d_NN, nn.size()));
  if (!nn.empty()) HIPCHK(hipMemcpy(c->d_NN, nn.data(), nn.size() * sizeof(int), hipMemcpyHostToDevice));
  c->mdl.NN = c->d_NN;
  if (sparse) {
    RB_TRY(dmalloc(&c->d_R, (size_t)d * d));
    HIPCHK(hipMemcpy(c->d_R, prob->R, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice));
    c->mdl.Rdev = c->d_R;
  }
  {
    std::vector<double> yt((size_t)T * d), od((size_t)std::max(T - 1, 1) * nodo, 0.0);
    for (int t = 0; t < T; ++t) for (int k = 0; k < d; ++k) yt[(size_t)t * d + k] = prob->y[t + (size_t)T * k];
    for (int t = 0; t < T - 1; ++t) for (int k = 0; k < nodo; ++k) od[(size_t)t * nodo + k] = prob->odometry[t + (size_t)prob->odo_ld * k];
    RB_TRY(dmalloc(&c->d_y, yt.size()));
    RB_TRY(dmalloc(&c->d_odo, od.size()));
    HIPCHK(hipMemcpy(c->d_y, yt.data(), yt.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_odo, od.data(), od.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> blk, full;
    RB_TRY(build_chol_factors(model, prob, blk, full, c->chol_pages, &c->cholQfull_ok));
    RB_TRY(dmalloc(&c->d_cholQ, blk.size()));
    RB_TRY(dmalloc(&c->d_cholQfull, full.size()));
    HIPCHK(hipMemcpy(c->d_cholQ, blk.data(), blk.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_cholQfull, full.data(), full.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  // initial state images
  {
    std::vector<double> x0l((size_t)prob->x0_lin_cols * L.ldx, 0.0);
    for (int j = 0; j < prob->x0_lin_cols; ++j)
      for (int r = 0; r < n; ++r) x0l[(size_t)j * L.ldx + r] = prob->x0_lin[r + (size_t)n * j];
    c->x0_lin_cols = prob->x0_lin_cols;
    RB_TRY(dmalloc(&c->d_x0l, x0l.size()));
    HIPCHK(hipMemcpy(c->d_x0l, x0l.data(), x0l.size() * sizeof(double), hipMemcpyHostToDevice));
    c->h_x0n.assign(prob->x0_nonlin, prob->x0_nonlin + nN);
    c->h_P0.assign(prob->P0_lin, prob->P0_lin + (size_t)n * n);
    c->h_x0l.assign(prob->x0_lin, prob->x0_lin + (size_t)n * prob->x0_lin_cols);
    c->h_R.assign(prob->R, prob->R + (size_t)d * d);
    c->h_y.assign(prob->y, prob->y + (size_t)T * d);
    double* tmp = nullptr;
    RB_TRY(dmalloc(&tmp, (size_t)n * n));
    hipError_t e = hipMemcpy(tmp, prob->P0_lin, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      int s1 = dmalloc(&c->d_P0t, std::max<size_t>(L.szT, 1));
      int s2 = dmalloc(&c->d_P0b, std::max<size_t>(L.szB, 1));
      if (s1 != RBPF_OK || s2 != RBPF_OK) { hipFree(tmp); return s1 != RBPF_OK ? s1 : s2; }
      e = launch_pack_P(L, tmp, 0, c->d_P0t, c->d_P0b, 1, c->stream, c->fp32 ? 1 : 0);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    hipFree(tmp);
    HIPCHK(e);
  }
  // ---- RNG block ----
  if (rng->mode == RBPF_RNG_REPLAY) {
    const int iters = c->N_K;
    if (T > 1 && (!rng->U || !rng->Z)) { set_error("replay RNG needs U and Z"); return RBPF_ERR_INVALID_ARG; }
    if (rng->n_iter < iters) { set_error("replay RNG has fewer pages than iterations"); return RBPF_ERR_INVALID_ARG; }
    if (smoother && !rng->Ufin) { set_error("replay RNG needs Ufin for the smoother"); return RBPF_ERR_INVALID_ARG; }
    const size_t nu = c->rng_slots * std::max(T - 1, 0) * iters;
    RB_TRY(dmalloc(&c->d_U, nu));
    RB_TRY(dmalloc(&c->d_Z, nu * nw));
    if (nu) {
      HIPCHK(hipMemcpy(c->d_U, rng->U, nu * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(c->d_Z, rng->Z, nu * nw * sizeof(double), hipMemcpyHostToDevice));
    }
    if (rng->Ufin) c->h_Ufin.assign(rng->Ufin, rng->Ufin + iters);
  } else if (rng->mode != RBPF_RNG_PHILOX) {
    set_error("unknown rng mode"); return RBPF_ERR_INVALID_ARG;
  }
  // ---- particle banks ----
  // multi-step lazy update: the filter (up to kMaxSets pending sets) and the information-form smoother (up to 3: its step
  // kernel carries two more right-hand sides); the covariance-form smoother switches it off (smoother_run)
  c->lazy_depth = (!sparse && c->opt.lazy_depth >= 2) ? std::min(c->opt.lazy_depth, smoother ? 3 : ((c->lay.sym && c->lay.CH64 == 8 && !c->fp32) ? (int)kMaxSets : (int)kMaxSetsFull)) : 1;
  {
    // one covariance bank rewritten in place instead of ping-pong banks: on request, or when two do not fit
    // (the information-form smoother on request only -- smoother_run refuses it for the covariance form, whose every step reads and
    //  rewrites the matrices)
    const bool can = !ex && c->lazy_depth >= 2 && (!smoother || c->opt.inplace > 0);
    if (c->opt.inplace > 0 && !can) { set_error("inplace=1 needs lazy_depth >= 2 and an unsharded session"); return RBPF_ERR_UNSUPPORTED; }
    c->inplace = c->opt.inplace > 0;
    if (c->opt.inplace == 0 && can) {
      size_t fr = 0, tot = 0;
      HIPCHK(hipMemGetInfo(&fr, &tot));
      const size_t two = 2 * c->bank_cap * (L.szT + L.szB) * (c->fp32 ? sizeof(float) : sizeof(double));
      c->inplace = (double)two > 0.85 * (double)fr;
    }
  }
  for (int b = 0; b < 2; ++b) {
    if (b == 1 && c->inplace) { c->Pt[1] = c->Pt[0]; c->Pb[1] = c->Pb[0]; }
    else {
      // (counts in doubles; a float bank needs half of them)
      RB_TRY(dmalloc(&c->Pt[b], c->fp32 ? (c->bank_cap * L.szT + 1) / 2 : c->bank_cap * L.szT));
      RB_TRY(dmalloc(&c->Pb[b], c->fp32 ? (c->bank_cap * L.szB + 1) / 2 : c->bank_cap * L.szB));
    }
    RB_TRY(dmalloc(&c->F[b], c->bank_cap * 2 * d * L.ldx));
    RB_TRY(dmalloc(&c->xl[b], c->bank_cap * L.ldx));
    HIPCHK(hipMemsetAsync(c->F[b], 0, c->bank_cap * 2 * d * L.ldx * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(c->xl[b], 0, c->bank_cap * L.ldx * sizeof(double), c->stream));
  }
  if (c->inplace) RB_TRY(dmalloc(&c->d_ip, (size_t)8 * N));
  c->share_inplace = c->inplace && c->lazy_depth >= 2 && !ex && c->lay.sym && (c->lay.CH64 == 8 || c->lay.CH64 == 16);
  if (const char* e = tuning_env("RBPF_SHARE_INPLACE")) c->share_inplace = c->share_inplace && atoi(e) != 0;   // diagnostic builds: the per-child in-place flush
  if (c->share_inplace) {
    RB_TRY(dmalloc(&c->d_share_writers, 1));
    HIPCHK(hipMemset(c->d_share_writers, 0, sizeof(unsigned long long)));
  }
  // shared flush: with ping-pong banks the children of one parent store ONE copy of their (identical) flushed matrix
  c->share_flush = c->lazy_depth >= 2 && !c->inplace && !ex && c->lay.sym && (c->lay.CH64 == 8 || c->lay.CH64 == 16);   // (smoothers: the information form)
  if (const size_t sd = sym_strip_doubles(c->lay, d)) {     // sixteen tile rows: the step kernel's column strips live in global memory
    RB_TRY(dmalloc(&c->d_strip_ws, (size_t)N * sd));
    c->strip_ws_stride = sd;
  }
  if (c->share_flush) {
    RB_TRY(dmalloc(&c->d_share, (size_t)3 * N));
    RB_TRY(dmalloc(&c->d_share_writers, 1));
    HIPCHK(hipMemset(c->d_share_writers, 0, sizeof(unsigned long long)));
  }
  if (c->lazy_depth >= 2) {
    if (L.CH < 1 || L.CPL < 1 || L.CPL > 2) { set_error("lazy_depth >= 2 needs 128 <= nLin with at most two row chunks per wave"); return RBPF_ERR_UNSUPPORTED; }
    if (!L.sym && step_lds_bytes(c->mdl, c->lay, smoother ? 1 : 0, c->lazy_depth) > 160 * 1024) { set_error("lazy_depth too large for the LDS plan"); return RBPF_ERR_UNSUPPORTED; }
    if (L.sym && step_sym_lds_bytes(c->mdl, c->lay, c->lazy_depth, 1, smoother ? 1 : 0) > 160 * 1024) { set_error("lazy_depth too large for the LDS plan"); return RBPF_ERR_UNSUPPORTED; }
    for (int b = 0; b <= c->lazy_depth; ++b) {            // entry N of every bank stays zero (fresh lineages)
      RB_TRY(dmalloc(&c->Fb[b], (size_t)(N + 1) * 2 * d * L.ldx));
      HIPCHK(hipMemsetAsync(c->Fb[b], 0, (size_t)(N + 1) * 2 * d * L.ldx * sizeof(double), c->stream));
    }
    for (int b = 0; b < 2; ++b) {
      RB_TRY(dmalloc(&c->fidx[b], (size_t)(c->lazy_depth + 1) * N));
      RB_TRY(dmalloc(&c->base[b], (size_t)N));
    }
  }
  c->hist_slabs = c->opt.keep_history ? T : 2;
  RB_TRY(dmalloc(&c->X, (size_t)c->hist_slabs * nN * N));
  RB_TRY(dmalloc(&c->A, (size_t)(c->opt.keep_history ? T : 1) * N));
  HIPCHK(hipMemsetAsync(c->A, 0, (size_t)(c->opt.keep_history ? T : 1) * N * sizeof(int), c->stream));
  const size_t tr = c->opt.trace ? (size_t)T : 1;
  RB_TRY(dmalloc(&c->logw, tr * N));
  RB_TRY(dmalloc(&c->w, tr * N));
  RB_TRY(dmalloc(&c->wc, (size_t)N));
  RB_TRY(dmalloc(&c->traj_max, (size_t)T * nN));
  RB_TRY(dmalloc(&c->traj_mean, (size_t)T * nN));
  RB_TRY(dmalloc(&c->d_scal, 64));
  RB_TRY(dmalloc(&c->d_rs, resample_scratch_doubles(std::max<size_t>((size_t)N, c->rng_slots))));
  RB_TRY(dmalloc(&c->d_unext, (size_t)N));
  RB_TRY(dmalloc(&c->d_pre_i, (size_t)N * kPreInts));
  RB_TRY(dmalloc(&c->d_pre_d, (size_t)N * kPreDoubles));
  RB_TRY(dmalloc(&c->d_order, (size_t)N));
  RB_TRY(dmalloc(&c->d_counts, (size_t)2 * N + 128));
  RB_TRY(dmalloc(&c->d_flags, 16));
  HIPCHK(hipMemsetAsync(c->d_flags, 0, 16 * sizeof(int), c->stream));
  RB_TRY(ctx_reset(c));
  HIPCHK(hipStreamSynchronize(c->stream));
  guard.release();
  *out = c;
  return RBPF_OK;
}

// fill X[0] with x0 (particleFilter.m:59) and rewind
__global__ void fill_x0_kernel(int N, int nN, const double x0_0, const double x0_1, const double x0_2, const double x0_3,
                               const double x0_4, const double x0_5, const double x0_6, const double x0_7, double* X0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double x0[8] = {x0_0, x0_1, x0_2, x0_3, x0_4, x0_5, x0_6, x0_7};
  for (int c = 0; c < nN; ++c) X0[(size_t)c * N + i] = x0[c];
}

int ctx_reset(rbpf_ctx* c) {
  double x0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < c->mdl.nN; ++k) x0[k] = c->h_x0n[k];
  hipLaunchKernelGGL(fill_x0_kernel, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, c->N, c->mdl.nN, x0[0], x0[1],
                     x0[2], x0[3], x0[4], x0[5], x0[6], x0[7], c->X);
  HIPCHK(hipGetLastError());
  c->t = 0;
  c->cur = 0;
  c->xcur = 0;
  c->tcur = 0;
  c->ready_step = -1;
  c->drawn_step = -1;
  c->order_step = -1;
  if (c->mdl.kind == RBPF_MODEL_GENERIC_DENSE) {              // particleFilter.m:59: xn = repmat(x0_nonLin, 1, N_P)
    c->h_xn.resize((size_t)c->mdl.nN * c->N);
    for (int i = 0; i < c->N; ++i) for (int q = 0; q < c->mdl.nN; ++q) c->h_xn[q + (size_t)c->mdl.nN * i] = c->h_x0n[q];
  }
  return RBPF_OK;
}

void ctx_free(rbpf_ctx* c) {
  if (!c) return;
  if (c->stream) hipStreamSynchronize(c->stream);
  for (auto& ev : c->events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
  hipFree(c->d_R); hipFree(c->d_xn_ext); hipFree(c->d_H_ext);
  hipFree(c->d_NN); hipFree(c->d_y); hipFree(c->d_odo); hipFree(c->d_cholQ); hipFree(c->d_cholQfull);
  hipFree(c->d_x0l); hipFree(c->d_P0t); hipFree(c->d_P0b); hipFree(c->d_U); hipFree(c->d_Z);
  if (c->inplace) { c->Pt[1] = nullptr; c->Pb[1] = nullptr; }     // aliases of bank 0
  hipFree(c->d_strip_ws);
  hipFree(c->d_ip); hipFree(c->d_share); hipFree(c->d_share_writers); hipFree(c->d_distinct_mark); hipFree(c->d_distinct_counter);
  for (int b = 0; b < 2; ++b) { hipFree(c->Pt[b]); hipFree(c->Pb[b]); hipFree(c->F[b]); hipFree(c->xl[b]); }
  hipFree(c->X); hipFree(c->A); hipFree(c->logw); hipFree(c->w); hipFree(c->wc);
  for (int b = 0; b <= kMaxSets; ++b) hipFree(c->Fb[b]);
  for (int b = 0; b < 2; ++b) { hipFree(c->fidx[b]); hipFree(c->base[b]); }
  hipFree(c->traj_max); hipFree(c->traj_mean); hipFree(c->d_scal); hipFree(c->d_flags); hipFree(c->d_order); hipFree(c->d_counts); hipFree(c->d_pre_i); hipFree(c->d_pre_d); hipFree(c->d_unext); hipFree(c->d_rs);
  smoother_free(c);
  shard_free(c);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

// Ancestors of the ordinary slots [0, n_draw) of the step about to run (particleFilter.m:106, particleSmoother.m:134):
// the weights of step t-1 were scanned in parallel, so draws within the rounding bound of a bin edge are flagged and
// resolved with the strict left-to-right cumsum (tools/sample.m:30) only then.
static int ctx_draw_ancestors(rbpf_ctx* c, int k_iter, int n_draw) {
  const int t = c->t, N = c->N;
  const bool hist = c->opt.keep_history != 0;
  int* A_t = c->A + (hist ? (size_t)t * N : 0);
  const size_t rng_page = (size_t)k_iter * N * std::max(c->T - 1, 0);
  SearchArgs s;
  s.N = N; s.n_draw = n_draw; s.t = t; s.wc = c->wc; s.rng_mode = c->rng_mode; s.k_iter = k_iter;
  s.U = c->d_U ? c->d_U + rng_page + (size_t)(t - 1) * N : nullptr;
  s.seed = c->seed; s.ai = A_t; s.overflow = c->d_flags + 1; s.slot0 = 0; s.u_is_scalar = 0;
  s.approx = 1; s.ambiguous = c->d_flags + 4; s.w = c->w + (c->opt.trace ? (size_t)(t - 1) * N : 0); s.wc_exact = c->wc;
  HIPCHK(launch_search(s, c->stream));
  HIPCHK(launch_resample_fixup(s, c->stream));
  return RBPF_OK;
}

// ---- generic family driven through rbpf_model.callbacks ---------------------------------------------------------
// The reference's handle contracts (particleFilter.m:104-109,124; particleSmoother.m:132-137,149-152): dynModel once per
// ordinary slot in slot order on the ancestor's state, measModel once on the whole batch.  The host keeps the states of
// the last step (it produced them), so only the ancestor indices come down and states + Jacobians go up.
int generic_draw_propagate(rbpf_ctx* c, int k_iter, int n_draw) {
  const int t = c->t, N = c->N, nN = c->mdl.nN;
  if (!c->has_cb) { set_error("generic model without callbacks"); return RBPF_ERR_STATE; }
  c->h_xn_new.assign((size_t)nN * N, 0.0);
  if (t == 0) { c->h_xn_new = c->h_xn; return RBPF_OK; }
  const bool hist = c->opt.keep_history != 0;
  int* A_t = c->A + (hist ? (size_t)t * N : 0);
  if (c->ready_step != t && c->drawn_step != t && n_draw > 0) { RB_TRY(ctx_draw_ancestors(c, k_iter, n_draw)); c->drawn_step = t; }
  c->h_ai.resize(N);
  HIPCHK(hipMemcpyAsync(c->h_ai.data(), A_t, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  std::vector<double> anc((size_t)nN * std::max(n_draw, 1));
  for (int i = 0; i < n_draw; ++i) {
    const int a = std::min(std::max(c->h_ai[i], 0), N - 1);
    for (int q = 0; q < nN; ++q) anc[q + (size_t)nN * i] = c->h_xn[q + (size_t)nN * a];
  }
  if (n_draw > 0 && c->cb.dyn_model(c->cb.user, t - 1, n_draw, anc.data(), c->h_xn_new.data()) != 0) {
    set_error("the dynModel callback failed"); return RBPF_ERR_CALLBACK;
  }
  return RBPF_OK;
}

// xref_host != null: slot N-1 is the conditioned reference trajectory (particleSmoother.m:242).
int generic_finish_inputs(rbpf_ctx* c, const double* xref_host) {
  const int N = c->N, nN = c->mdl.nN, d = c->mdl.d, n = c->mdl.n, ldx = c->lay.ldx;
  if (xref_host) for (int q = 0; q < nN; ++q) c->h_xn_new[q + (size_t)nN * (N - 1)] = xref_host[q];
  c->h_dy.assign((size_t)N * d * n, 0.0);
  if (c->cb.meas_model(c->cb.user, N, c->h_xn_new.data(), c->h_dy.data()) != 0) { set_error("the measModel callback failed"); return RBPF_ERR_CALLBACK; }
  std::vector<double> soa((size_t)nN * N), H((size_t)N * d * ldx, 0.0);
  for (int i = 0; i < N; ++i)
    for (int q = 0; q < nN; ++q) soa[(size_t)q * N + i] = c->h_xn_new[q + (size_t)nN * i];
  for (int cc = 0; cc < n; ++cc)                     // dy(i, k, cc) at i + N*(k + d*cc)  ->  H[(i*d + k)*ldx + cc]
    for (int k = 0; k < d; ++k)
      for (int i = 0; i < N; ++i) H[((size_t)i * d + k) * ldx + cc] = c->h_dy[(size_t)i + (size_t)N * (k + (size_t)d * cc)];
  HIPCHK(hipMemcpyAsync(c->d_xn_ext, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_H_ext, H.data(), H.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));          // the host vectors above go out of scope
  c->ext_xn = c->d_xn_ext; c->ext_H = c->d_H_ext;
  c->h_xn = c->h_xn_new;
  return RBPF_OK;
}

// Bytes a timed launch of the step kernel has to move (rbpf_timing.scheduled_bytes_per_launch): the stored covariance in
// (and out when it is rewritten), the pending factor sets it applies and the one it produces, the means and the states in
// and out (+ ivec in / out and H out for the information form).
// Timed launches: let propagate_kernel count the distinct stored matrices this step reads (`keys`: range of the slots).
int ctx_arm_distinct(rbpf_ctx* c, StepArgs& a, size_t keys) {
  a.distinct_mark = nullptr; a.distinct_counter = nullptr; a.distinct_tag = 0;
  if (!c->timing_on || a.t == 0) return RBPF_OK;
  if (c->distinct_keys < keys) {
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(c->d_distinct_mark); c->d_distinct_mark = nullptr;
    HIPCHK(hipMalloc(&c->d_distinct_mark, keys * sizeof(int)));
    HIPCHK(hipMemset(c->d_distinct_mark, 0, keys * sizeof(int)));
    c->distinct_keys = keys;
  }
  if (!c->d_distinct_counter) {
    HIPCHK(hipMalloc(&c->d_distinct_counter, sizeof(unsigned long long)));
    HIPCHK(hipMemset(c->d_distinct_counter, 0, sizeof(unsigned long long)));
  }
  a.distinct_mark = c->d_distinct_mark; a.distinct_counter = c->d_distinct_counter; a.distinct_tag = ++c->distinct_epoch;
  c->distinct_nominal += a.N;
  return RBPF_OK;
}

void ctx_account_launch(rbpf_ctx* c, const StepArgs& a) {
  const double sP = c->fp32 ? 4.0 : 8.0, nn = (double)c->mdl.n, d = (double)c->mdl.d, nN = (double)c->mdl.nN;
  // stored elements of one covariance: n^2, or the lower block triangle + border rows of the symmetric layout
  const double stored = c->lay.sym ? (double)(c->lay.szT + c->lay.szB) : nn * nn;
  const double per = stored * sP * ((a.t > 0 ? 1.0 : 0.0) + (a.write_base ? 1.0 : 0.0))
                   + 8.0 * (2.0 * nn * d * (a.n_sets + 1) + 2.0 * nn + 2.0 * nN + (a.info ? 2.0 * nn + d * nn : 0.0));
  c->sched_bytes += per * (double)a.N;
}

int ctx_call_on_step(rbpf_ctx* c, int t, bool is_smoother) {
  if (!c->opt.on_step) return RBPF_OK;
  rbpf_view v;
  v.ctx = is_smoother ? nullptr : c; v.t = t; v.is_smoother = is_smoother ? 1 : 0;
  if (c->opt.on_step(&v, c->opt.on_step_user) != 0) { set_error("the on_step hook returned non-zero"); return RBPF_ERR_CALLBACK; }
  return RBPF_OK;
}

// One time step of particleFilter.m:100-218 / particleSmoother.m:124-341 (iteration k_iter).
// xref != nullptr: slot N-1 is the conditioned reference trajectory (its ancestor index has
// already been written to A_t[N-1] by the ancestor-sampling kernels).
int ctx_step(rbpf_ctx* c, int k_iter, const double* xref_t, int n_draw, const InfoStep* info) {
  const int t = c->t, N = c->N, nN = c->mdl.nN, d = c->mdl.d, nw = c->mdl.nw;
  if (t >= c->T) { set_error("advance past N_T"); return RBPF_ERR_STATE; }
  const Layout& L = c->lay;
  const bool hist = c->opt.keep_history != 0;
  int* A_t = c->A + (hist ? (size_t)t * N : 0);
  double* X_new = c->X + (size_t)(hist ? t : (t & 1)) * nN * N;
  const double* X_old = (t == 0) ? X_new : c->X + (size_t)(hist ? t - 1 : ((t - 1) & 1)) * nN * N;
  const size_t tr = c->opt.trace ? (size_t)t * N : 0;
  const size_t rng_page = (size_t)k_iter * N * std::max(c->T - 1, 0);

  const bool pre_drawn = (c->ready_step == t);       // ancestors + order came from the fused kernel of step t-1
  if (t > 0 && n_draw > 0 && !pre_drawn && c->drawn_step != t) RB_TRY(ctx_draw_ancestors(c, k_iter, n_draw));
  if (c->sort_steps && t > 0 && !pre_drawn && c->mdl.kind != RBPF_MODEL_SPARSE_VISUAL_2D) {
    // smoothers: every ancestor of this step is known now (slot N-1's came from the ancestor weights); process the children
    // in the order of the stored matrix they read, so that siblings share it through the caches as in the filter
    const int* remap = (c->lazy_depth >= 2) ? c->base[c->tcur] : nullptr;
    HIPCHK(launch_order(N, N, A_t, c->d_order, c->d_counts, c->stream, remap));
    c->order_step = t;
  }
  if (c->mdl.kind == RBPF_MODEL_SPARSE_VISUAL_2D) {
    // sparseFeatures branch: one small kernel does gather, dynModel, EKF weight and update (rbpf_sparse.hip);
    // the update is applied at once, so the pending-factor banks stay zero
    if (info) { set_error("This code has only been implemented for dense features"); return RBPF_ERR_UNSUPPORTED; }
    const int ob = c->cur, nb = (t == 0) ? 0 : (c->cur ^ 1);
    SparseStepArgs sa;
    sa.N = N; sa.t = t; sa.n = c->mdl.n; sa.d = d; sa.nN = nN; sa.nw = nw; sa.ldx = L.ldx; sa.ldb = L.ldb; sa.propagate = (t > 0);
    sa.szB = L.szB; sa.f = c->mdl.cam[0]; sa.fp = c->mdl.cam[1];
    sa.ai = (t > 0) ? A_t : nullptr;
    sa.xn_old = X_old; sa.xn_old_stride = (size_t)N; sa.xn_new = X_new; sa.xn_new_stride = (size_t)N;
    if (t == 0) {
      sa.xl_old = c->d_x0l; sa.xl_old_stride = (c->x0_lin_cols > 1) ? (size_t)L.ldx : 0;
      sa.Pb_old = c->d_P0b; sa.Pb_old_stride = 0;
    } else {
      sa.xl_old = c->xl[ob]; sa.xl_old_stride = (size_t)L.ldx;
      sa.Pb_old = c->Pb[ob]; sa.Pb_old_stride = L.szB;
    }
    sa.xl_new = c->xl[nb]; sa.Pb_new = c->Pb[nb];
    sa.y = c->d_y + (size_t)t * d; sa.R = c->d_R;
    sa.odo = c->d_odo + (size_t)(t > 0 ? t - 1 : 0) * c->mdl.nodo;
    sa.Ssqrt = c->d_cholQ + (size_t)((c->chol_pages > 1 && t > 0) ? t - 1 : 0) * nw * nw;
    sa.rng_mode = c->rng_mode; sa.k_iter = k_iter; sa.seed = c->seed;
    sa.Z = (c->d_Z && t > 0) ? c->d_Z + (rng_page + (size_t)(t - 1) * N) * nw : nullptr;
    sa.xref = xref_t; sa.jitter = c->mdl.jitter; sa.logw = c->logw + tr; sa.status = c->d_flags;
    HIPCHK(launch_sparse_step(sa, c->stream));
    NormArgs nm;
    nm.N = N; nm.nN = nN; nm.t = t; nm.logw = c->logw + tr; nm.w = c->w + tr; nm.wc = c->wc; nm.xn = X_new;
    nm.traj_max = c->traj_max + (size_t)t * nN; nm.traj_mean = c->traj_mean + (size_t)t * nN;
    nm.iw_max = c->d_flags + 2; nm.lse_out = nullptr;
    nm.parallel_scan = 1;
    if (N > kSingleWgResampleMaxN) HIPCHK(launch_resample_pipeline(nm, nullptr, nullptr, nullptr, nullptr, c->d_rs, c->stream));
    else HIPCHK(launch_normalise_scan(nm, c->stream));
    c->cur = nb; c->xcur = nb;
    c->t = t + 1;
    return RBPF_OK;
  }
  StepArgs a;
  a.mdl = c->mdl; a.lay = L; a.N = N; a.t = t; a.propagate = (t > 0);
  a.ai = (t > 0) ? A_t : nullptr;
  a.ai_bank = nullptr; a.slot_offset = 0; a.xn_old_stride = (size_t)N; a.xn_new_stride = (size_t)N;
  a.order = ((pre_drawn || c->order_step == t) && t > 0) ? c->d_order : nullptr;
  a.zero_set_idx = N;
  a.slot_ids = nullptr; a.n_bank_local = 0; a.rec = nullptr; a.rec_stride = 0; a.rec_off_B = a.rec_off_F = a.rec_off_X = 0;
  a.rec_off_I = a.rec_off_hld = 0;
  {
    static const int no_order = tuning_env("RBPF_NO_ORDER") ? 1 : 0;      // tuning / debugging only
    static const int dbg_order = tuning_env("RBPF_DEBUG_ORDER") ? 1 : 0;
    if (no_order) a.order = nullptr;
#ifdef RBPF_STAMPS
    if (t >= 100 && t <= 103) {
      unsigned long long ks[8];
      HIPCHK(hipStreamSynchronize(c->stream));
      HIPCHK(hipMemcpy(ks, c->d_counts + 2 * N + 32, sizeof(ks), hipMemcpyDeviceToHost));
      fprintf(stderr, "[rbpf kstamps] step %d (prev launch): A %.1f B %.1f C %.1f D(stream) %.1f comb %.1f E %.1f F %.1f us\n", t - 1, (ks[1]-ks[0])*0.01, (ks[2]-ks[1])*0.01, (ks[3]-ks[2])*0.01, (ks[4]-ks[3])*0.01, 0.0, (ks[5]-ks[4])*0.01, (ks[6]-ks[5])*0.01);
    }
    if (t == 100) {
      unsigned long long st[8];
      HIPCHK(hipStreamSynchronize(c->stream));
      HIPCHK(hipMemcpy(st, c->d_counts + 2 * N, sizeof(st), hipMemcpyDeviceToHost));
      fprintf(stderr, "[rbpf stamps] normalise %.1f us, search %.1f us, order %.1f us\n", (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01);
      fprintf(stderr, "[rbpf stamps] shader clock during the kernel: %.0f MHz\n", (double)(st[7] - st[4]) / ((st[3] - st[0]) * 0.01));
    }
#endif
    if (dbg_order && a.order && (t == 5 || t == 50)) {
      std::vector<int> ho(N), ha(N);
      HIPCHK(hipStreamSynchronize(c->stream));
      HIPCHK(hipMemcpy(ho.data(), c->d_order, (size_t)N * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(ha.data(), A_t, (size_t)N * 4, hipMemcpyDeviceToHost));
      long bad = 0; std::vector<char> seen(N, 0); long dup = 0;
      for (int b = 0; b < N; ++b) { if (ho[b] < 0 || ho[b] >= N || seen[ho[b]]) ++dup; else seen[ho[b]] = 1; }
      for (int b = 1; b < N; ++b) if (ha[ho[b]] < ha[ho[b - 1]]) ++bad;
      fprintf(stderr, "[rbpf debug] t=%d order: %ld inversions, %ld duplicates/out-of-range; first anc %d %d %d %d\n", t, bad, dup, ha[ho[0]], ha[ho[1]], ha[ho[2]], ha[ho[3]]);
    }
  }
  a.xn_old = X_old; a.xn_new = X_new;
  const bool lazy = c->lazy_depth >= 2;
  const int ob = c->cur;
  int nb = (t == 0) ? 0 : (c->cur ^ 1);          // bank the stored covariances are written to (if at all)
  const int xo = c->xcur, xn = (t == 0) ? 0 : (c->xcur ^ 1);
  const int told = c->tcur, tnew = c->tcur ^ 1;
  bool flush = true;
  for (int q = 0; q < kMaxSets; ++q) { a.fset[q] = nullptr; a.fset_idx_old[q] = nullptr; a.fset_idx_new[q] = nullptr; }
  a.fself_idx_new = nullptr; a.base_old = nullptr; a.base_new = nullptr;
  a.dst_slot = nullptr; a.phase_of = nullptr; a.phase = -1; a.share_flush = 0;
  a.fp32 = c->fp32 ? 1 : 0;
  a.strip_ws = c->d_strip_ws; a.strip_ws_stride = c->strip_ws_stride;
  a.xn_ext = c->ext_xn; a.H_ext = c->ext_H;
  if (c->mdl.kind == RBPF_MODEL_GENERIC_DENSE && (!a.xn_ext || !a.H_ext)) {
    set_error("generic (host-callback) model: advance with rbpf_filter_step_external"); return RBPF_ERR_STATE;
  }
  a.n_sets = (t > 0) ? 1 : 0; a.write_base = 1;
  if (lazy) {
    // multi-step lazy update: sets produced at steps t-ell .. t-1 are pending; every C-th step rewrites the matrices
    const int C = c->lazy_depth, B = C + 1;
    const int ell = (t == 0) ? 0 : ((t - 1) % C) + 1;
    flush = (t == 0) || (ell == C);
    a.n_sets = ell; a.write_base = flush ? 1 : 0;
    if (ell >= 3) a.lay = c->lay_low;
    for (int q = 0; q < ell; ++q) {
      const int bank = (t - ell + q) % B;
      a.fset[q] = c->Fb[bank];
      a.fset_idx_old[q] = c->fidx[told] + (size_t)bank * N;
      a.fset_idx_new[q] = c->fidx[tnew] + (size_t)bank * N;
    }
    a.fself_idx_new = c->fidx[tnew] + (size_t)(t % B) * N;
    a.base_old = (t > 0) ? c->base[told] : nullptr;
    a.base_new = c->base[tnew];
    if (!flush) nb = ob;
  }
  if (t == 0) {
    a.xl_old = c->d_x0l; a.xl_old_stride = (c->x0_lin_cols > 1) ? (size_t)L.ldx : 0;
    a.F_old = nullptr;
    a.Pt_old = c->d_P0t; a.Pb_old = c->d_P0b; a.Pt_old_stride = 0; a.Pb_old_stride = 0;
  } else {
    a.xl_old = c->xl[xo]; a.xl_old_stride = (size_t)L.ldx;
    a.F_old = lazy ? nullptr : c->F[ob];
    a.Pt_old = c->Pt[ob]; a.Pb_old = c->Pb[ob]; a.Pt_old_stride = L.szT; a.Pb_old_stride = L.szB;
  }
  a.xl_new = c->xl[xn]; a.F_new = lazy ? c->Fb[t % (c->lazy_depth + 1)] : c->F[nb];
  a.Pt_new = c->Pt[nb]; a.Pb_new = c->Pb[nb];
  a.logw = c->logw + tr;
  a.rng_mode = c->rng_mode; a.k_iter = k_iter; a.seed = c->seed;
  a.Z = (c->d_Z && t > 0) ? c->d_Z + (rng_page + (size_t)(t - 1) * N) * nw : nullptr;
  a.odo = c->d_odo + (size_t)(t > 0 ? t - 1 : 0) * c->mdl.nodo;
  a.cholQ = c->d_cholQ + (size_t)((c->chol_pages > 1 && t > 0) ? t - 1 : 0) * nw * nw;
  a.y = c->d_y + (size_t)t * d;
  a.xref = xref_t; a.xref_gslot = N - 1;
  a.status = c->d_flags;
  a.stamps = reinterpret_cast<unsigned long long*>(c->d_counts + 2 * N + 32);
  a.pre_i = c->d_pre_i; a.pre_d = c->d_pre_d;
  // fused fast path with the device generator: the uniforms of step t+1 are produced here, in parallel
  a.u_next = (c->fuse_resample && c->rng_mode == RBPF_RNG_PHILOX && t + 1 < c->T) ? c->d_unext : nullptr;
  a.info = info ? 1 : 0;
  a.ivec_old = nullptr; a.ivec_old_stride = 0; a.ivec_new = nullptr; a.hld_old = nullptr; a.hld_old_stride = 0;
  a.hld_new = nullptr; a.qf_new = nullptr; a.Hb_new = nullptr;
  if (info) {
    a.ivec_old = info->ivec_old; a.ivec_old_stride = info->ivec_old_stride; a.ivec_new = info->ivec_new;
    a.hld_old = info->hld_old; a.hld_old_stride = info->hld_old_stride; a.hld_new = info->hld_new;
    a.qf_new = info->qf_new; a.Hb_new = info->Hb_new;
  }
  const bool two_phase = c->inplace && lazy && flush && t > 0;
  const bool sip = two_phase && c->share_inplace && a.n_sets >= 1 && a.n_sets <= (info ? 3 : 7);
  if (sip) {
    // single bank, shared flush: one writer per parent with children; the first writer of a stored matrix overwrites it in place
    // after everything else has read it (launch_share_inplace_plan)
    if (!a.order) { set_error("in-place flush without a processing order"); return RBPF_ERR_STATE; }
    int* dst = c->d_ip; int* ph = c->d_ip + N;
    HIPCHK(launch_share_inplace_plan(N, a.order, A_t, c->base[told], dst, ph, c->d_ip + 2 * (size_t)N, c->timing_on ? c->d_share_writers : nullptr, c->stream));
    a.dst_slot = dst; a.phase_of = ph; a.share_flush = 1;
  } else if (two_phase) {
    // single-bank flush: siblings move to dead entries first (launch 0), then the first child of every stored
    // matrix overwrites it (launch 1); the plan comes from the ancestor-sorted order of the fused resample kernel
    if (!a.order) { set_error("in-place flush without a processing order"); return RBPF_ERR_STATE; }
    int* dst = c->d_ip; int* ph = c->d_ip + N;
    HIPCHK(launch_inplace_plan(N, a.order, A_t, c->base[told], dst, ph, c->d_ip + 2 * (size_t)N, c->stream));
    a.dst_slot = dst; a.phase_of = ph;
  }
  const bool share = c->share_flush && lazy && flush && t > 0 && !two_phase && a.n_sets >= 1 && a.n_sets <= (info ? 3 : 7);
  if (share) {
    // the children of one parent flush to ONE entry (launch_share_plan): the smallest child stores it, its siblings only read
    int* lead = c->d_share; int* dst = c->d_share + N; int* ph = c->d_share + 2 * (size_t)N;
    HIPCHK(launch_share_plan(N, N, A_t, lead, dst, ph, c->timing_on ? c->d_share_writers : nullptr, c->stream));
    a.dst_slot = dst; a.phase_of = ph; a.share_flush = 1;
  }
  RB_TRY(ctx_arm_distinct(c, a, (size_t)N + 1));
  HIPCHK(launch_propagate(a, c->stream));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->timing_on) {
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, c->stream));
  }
  if (sip) {
    StepArgs rd = a;
    rd.phase = 0; rd.write_base = 0; HIPCHK(launch_step(rd, c->stream));   // readers over the old matrices
    a.phase = 1; HIPCHK(launch_step(a, c->stream));                        // writers into entries nobody refers to
    a.phase = 2; HIPCHK(launch_step(a, c->stream));                        // first writers, in place
    if (c->timing_on) c->share_flush_particles += N;
  } else if (two_phase) {
    a.phase = 0; HIPCHK(launch_step(a, c->stream));
    a.phase = 1; HIPCHK(launch_step(a, c->stream));
  } else if (share) {
    // Writers (the flush variant, one per parent with children) first, then their read-only siblings (the same pending sets; they
    // point at their writer's new entry), one after the other.  (Measured r05 with diagnostic builds of commit 31144f4: both side by side on
    // two streams 11.39 against 11.24 ms per step; one step of a STAGGERED flush -- a quarter of the families flushing beside three
    // quarters of read-only particles -- 12.2 ms whether serial or concurrent, against the lock-step schedule's 11.66: DESIGN.md 9.)
    a.phase = 1; HIPCHK(launch_step(a, c->stream));                 // writers: the flush variant
    StepArgs rd = a;
    rd.phase = 0; rd.write_base = 0;                                // readers: the read-only variant with the same pending sets
    HIPCHK(launch_step(rd, c->stream));
    if (c->timing_on) c->share_flush_particles += N;
  } else {
    HIPCHK(launch_step(a, c->stream));
  }
  if (c->timing_on) {
    HIPCHK(hipEventRecord(e1, c->stream)); c->events.emplace_back(e0, e1);
    ctx_account_launch(c, a);
  }

  NormArgs nm;
  nm.N = N; nm.nN = nN; nm.t = t; nm.logw = c->logw + tr; nm.w = c->w + tr; nm.wc = c->wc; nm.xn = X_new;
  nm.traj_max = c->traj_max + (size_t)t * nN; nm.traj_mean = c->traj_mean + (size_t)t * nN;
  nm.iw_max = c->d_flags + 2; nm.lse_out = nullptr;
  nm.parallel_scan = 1;
  if (c->fuse_resample && t + 1 < c->T) {
    // filter fast path: one single-workgroup kernel normalises step t and draws step t+1's ancestors and
    // their ancestor-sorted processing order
    int* A_next = c->A + (hist ? (size_t)(t + 1) * N : 0);
    SearchArgs s;
    s.N = N; s.n_draw = N; s.t = t + 1; s.wc = c->wc; s.rng_mode = c->rng_mode; s.k_iter = k_iter;
    s.U = c->d_U ? c->d_U + rng_page + (size_t)t * N : nullptr;
    if (a.u_next) { s.rng_mode = RBPF_RNG_REPLAY; s.U = c->d_unext; }     // pre-drawn by propagate_kernel (same Philox values)
    s.seed = c->seed; s.ai = A_next; s.overflow = c->d_flags + 1; s.slot0 = 0; s.u_is_scalar = 0;
    s.approx = 1; s.ambiguous = c->d_flags + 4; s.w = c->w + tr; s.wc_exact = c->wc;
    // sort key of the next step: the slot of the stored matrix each ancestor's lineage refers to
    if (N > kSingleWgResampleMaxN)
      HIPCHK(launch_resample_pipeline(nm, &s, c->d_order, c->d_counts, lazy ? c->base[tnew] : nullptr, c->d_rs, c->stream));
    else
      HIPCHK(launch_normalise_resample(nm, s, c->d_order, c->d_counts, c->stream, lazy ? c->base[tnew] : nullptr));
    c->ready_step = t + 1;
  } else if (N > kSingleWgResampleMaxN) {
    HIPCHK(launch_resample_pipeline(nm, nullptr, nullptr, nullptr, nullptr, c->d_rs, c->stream));
  } else {
    HIPCHK(launch_normalise_scan(nm, c->stream));
  }
  c->cur = nb;
  c->xcur = lazy ? xn : nb;
  if (lazy) c->tcur = tnew;
  c->t = t + 1;
  return RBPF_OK;
}

// Flushed covariances of `count` particles (index: device list or null = 0..count-1) in MATLAB layout.
int ctx_unpack(rbpf_ctx* c, const int* d_index, int count, double* d_out) {
  const Layout& L = c->lay;
  const int d = c->mdl.d, N = c->N;
  if (c->lazy_depth < 2) {
    HIPCHK(launch_unpack_P(L, d, c->Pt[c->cur], c->Pb[c->cur], c->t > 0 ? c->F[c->cur] : nullptr, d_index, count, d_out, c->stream,
                           c->fp32 ? 1 : 0));
    return RBPF_OK;
  }
  const int C = c->lazy_depth, B = C + 1, t = c->t;          // state after step t-1
  const int ell = (t == 0) ? 0 : ((t - 1) % C) + 1;
  const double* fset[kMaxSets]; const int* fidx[kMaxSets];
  for (int q = 0; q < ell; ++q) {
    const int bank = (t - ell + q) % B;
    fset[q] = c->Fb[bank]; fidx[q] = c->fidx[c->tcur] + (size_t)bank * N;
  }
  HIPCHK(launch_unpack_P_sets(L, d, c->Pt[c->cur], c->Pb[c->cur], ell, fset, fidx, c->base[c->tcur], d_index, count, d_out, c->stream,
                              c->fp32 ? 1 : 0));
  return RBPF_OK;
}

// fix_p_mean = 1: P_mean = sum_i w(i) * (P(:,:,i) + (xl_mean - xl(:,i)) * (xl_mean - xl(:,i))'), i.e. particleFilter.m:228-230
// with "+=" instead of the reference's "=" (quirk Q3).  One thread per matrix element, particles added in index order.
__global__ void p_mean_accum_kernel(int n, int ldx, int i0, int count, const double* __restrict__ dP,
                                    const double* __restrict__ xl, const double* __restrict__ xl_mean,
                                    const double* __restrict__ w, double* __restrict__ acc) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (size_t)n * n) return;
  const int r = (int)(q % n), cc = (int)(q / n);
  double s = acc[q];
  for (int j = 0; j < count; ++j) {
    const int i = i0 + j;
    const double dr = xl_mean[r] - xl[(size_t)i * ldx + r], dc = xl_mean[cc] - xl[(size_t)i * ldx + cc];
    s += w[i] * (dP[(size_t)j * n * n + q] + dr * dc);
  }
  acc[q] = s;
}

__global__ void iota_kernel(int count, int first, int* out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < count) out[j] = first + j;
}

static int accumulate_p_mean(rbpf_ctx* c, const double* w_last, const std::vector<double>& xl_mean_h, double* P_mean_host) {
  const int N = c->N, n = c->mdl.n, ldx = c->lay.ldx, cur = c->xcur;
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)N, ((size_t)64 << 20) / ((size_t)n * n * sizeof(double))));
  double *dP = nullptr, *dacc = nullptr, *dmean = nullptr; int* didx = nullptr;
  int st = dmalloc(&dP, (size_t)chunk * n * n);
  if (st == RBPF_OK) st = dmalloc(&dacc, (size_t)n * n);
  if (st == RBPF_OK) st = dmalloc(&dmean, (size_t)n);
  if (st == RBPF_OK) st = dmalloc(&didx, (size_t)chunk);
  hipError_t e = hipSuccess;
  if (st == RBPF_OK) {
    e = hipMemsetAsync(dacc, 0, (size_t)n * n * sizeof(double), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dmean, xl_mean_h.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream);
    for (int i0 = 0; i0 < N && e == hipSuccess && st == RBPF_OK; i0 += chunk) {
      const int count = std::min(chunk, N - i0);
      hipLaunchKernelGGL(iota_kernel, dim3((count + 255) / 256), dim3(256), 0, c->stream, count, i0, didx);
      st = ctx_unpack(c, didx, count, dP);
      if (st != RBPF_OK) break;
      hipLaunchKernelGGL(p_mean_accum_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, c->stream, n, ldx, i0,
                         count, dP, c->xl[cur], dmean, w_last, dacc);
      e = hipGetLastError();
    }
    if (e == hipSuccess && st == RBPF_OK) e = hipMemcpyAsync(P_mean_host, dacc, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && st == RBPF_OK) e = hipStreamSynchronize(c->stream);
  }
  hipFree(dP); hipFree(dacc); hipFree(dmean); hipFree(didx);
  if (st != RBPF_OK) return st;
  HIPCHK(e);
  return RBPF_OK;
}

int ctx_check_flags(rbpf_ctx* c) {
  int flags[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(flags, c->d_flags, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->overflow_draws = flags[1];
  if (flags[0] & 1) { set_error("Cholesky of the innovation covariance failed twice (matrix must be positive definite)"); return RBPF_ERR_CHOL_FAILED; }
  if (flags[0] & 2) { set_error("Cholesky in the ancestor-weight computation failed (matrix must be positive definite)"); return RBPF_ERR_CHOL_FAILED; }
  return RBPF_OK;
}

}  // namespace rbpf

using namespace rbpf;

extern "C" {

int rbpf_abi_version(void) { return RBPF_ABI_VERSION; }

int rbpf_abi_sizeof(int32_t which) {
  switch (which) {
    case 0: return (int)sizeof(rbpf_model);
    case 1: return (int)sizeof(rbpf_problem);
    case 2: return (int)sizeof(rbpf_rng);
    case 3: return (int)sizeof(rbpf_options);
    case 4: return (int)sizeof(rbpf_filter_out);
    case 5: return (int)sizeof(rbpf_smoother_out);
    case 6: return (int)sizeof(rbpf_timing);
    case 7: return (int)sizeof(rbpf_callbacks);
    case 8: return (int)sizeof(rbpf_view);
    default: return -1;
  }
}

const char* rbpf_status_string(int s) {
  switch (s) {
    case RBPF_OK: return "ok";
    case RBPF_ERR_INVALID_ARG: return "invalid argument";
    case RBPF_ERR_UNSUPPORTED: return "unsupported on the device path";
    case RBPF_ERR_HIP: return "HIP runtime error";
    case RBPF_ERR_NO_DEVICE: return "no gfx950 device (no CPU fallback)";
    case RBPF_ERR_OUT_OF_MEMORY: return "out of device memory";
    case RBPF_ERR_CHOL_FAILED: return "matrix must be positive definite";
    case RBPF_ERR_STATE: return "invalid call sequence";
    default: return "unknown status";
  }
}

const char* rbpf_last_error(void) { return g_last_error.c_str(); }

int rbpf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rbpf_filter_workspace_bytes(const rbpf_model* model, const rbpf_problem* p, const rbpf_options* opt, size_t* bytes) {
  if (!model || !p || !bytes) { set_error("NULL argument"); return RBPF_ERR_INVALID_ARG; }
  RB_TRY(options_ok(opt));
  const Layout L = (opt && (opt->storage == 2 || opt->storage == 3) && sym_supported(p->n_lin, p->n_y)) ? make_layout_sym(p->n_lin, p->n_y, opt->storage == 3 ? 1 : 0) : make_layout(p->n_lin, p->n_y);
  const bool f32 = opt && (opt->storage == 1 || opt->storage == 3);
  const bool hist = !opt || opt->keep_history;
  const bool trace = opt && opt->trace;
  size_t b = 2 * bank_bytes(L, p->n_y, p->N_P);
  if (f32) b -= (size_t)p->N_P * (L.szT + L.szB) * sizeof(double);  // float banks: half of two double banks
  if (opt && opt->inplace > 0) b -= (size_t)p->N_P * (L.szT + L.szB) * (f32 ? sizeof(float) : sizeof(double));   // one bank
  b += (size_t)p->N_P * sym_strip_doubles(L, p->n_y) * sizeof(double);
  b += (size_t)(hist ? p->N_T : 2) * p->n_nonlin * p->N_P * sizeof(double);
  b += (size_t)(hist ? p->N_T : 1) * p->N_P * sizeof(int);
  b += (size_t)(trace ? 2 * p->N_T : 2) * p->N_P * sizeof(double) + (size_t)p->N_P * sizeof(double);
  b += (L.szT + L.szB) * sizeof(double);
  *bytes = b;
  return RBPF_OK;
}

int rbpf_filter_create(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
                       rbpf_ctx** ctx) {
  return ctx_create(model, prob, rng, opt, false, 1, ctx);
}

int rbpf_filter_advance(rbpf_ctx* c, int32_t n_steps) {
  if (!c) { set_error("ctx is NULL"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  c->fuse_resample = true;
  for (int s = 0; s < n_steps; ++s) {
    if (c->has_cb) {                                   // generic family: evaluate the handles of this step on the host
      RB_TRY(generic_draw_propagate(c, 0, c->N));
      RB_TRY(generic_finish_inputs(c, nullptr));
    }
    const int st = ctx_step(c, 0, nullptr, c->N, nullptr);
    if (c->has_cb) { c->ext_xn = nullptr; c->ext_H = nullptr; }
    RB_TRY(st);
    RB_TRY(ctx_call_on_step(c, c->t - 1, false));      // particleFilter.m:215-217
  }
  return RBPF_OK;
}

int rbpf_filter_ancestors(rbpf_ctx* c, int32_t* ai, double* xn_prev) {
  if (!c || !ai || !xn_prev) { set_error("NULL argument"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  const int t = c->t, N = c->N, nN = c->mdl.nN;
  if (t < 1 || t >= c->T || c->ready_step != t) { set_error("no ancestors drawn for the next step (run a step first)"); return RBPF_ERR_STATE; }
  const bool hist = c->opt.keep_history != 0;
  HIPCHK(hipMemcpyAsync(ai, c->A + (hist ? (size_t)t * N : 0), (size_t)N * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  std::vector<double> soa((size_t)nN * N);
  HIPCHK(hipMemcpyAsync(soa.data(), c->X + (size_t)(hist ? t - 1 : ((t - 1) & 1)) * nN * N, soa.size() * sizeof(double),
                        hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int i = 0; i < N; ++i)
    for (int q = 0; q < nN; ++q) xn_prev[q + (size_t)nN * i] = soa[(size_t)q * N + i];
  return RBPF_OK;
}

int rbpf_filter_step_external(rbpf_ctx* c, const double* xn_new, const double* dy) {
  if (!c || !xn_new || !dy) { set_error("NULL argument"); return RBPF_ERR_INVALID_ARG; }
  if (!c->d_xn_ext || !c->d_H_ext) { set_error("not a generic-model context"); return RBPF_ERR_STATE; }
  HIPCHK(hipSetDevice(c->device));
  const int N = c->N, nN = c->mdl.nN, d = c->mdl.d, n = c->mdl.n, ldx = c->lay.ldx;
  std::vector<double> soa((size_t)nN * N), H((size_t)N * d * ldx, 0.0);
  for (int i = 0; i < N; ++i)
    for (int q = 0; q < nN; ++q) soa[(size_t)q * N + i] = xn_new[q + (size_t)nN * i];
  for (int cc = 0; cc < n; ++cc)                     // dy(i, k, cc) at i + N*(k + d*cc)  ->  H[(i*d + k)*ldx + cc]
    for (int k = 0; k < d; ++k)
      for (int i = 0; i < N; ++i) H[((size_t)i * d + k) * ldx + cc] = dy[(size_t)i + (size_t)N * (k + (size_t)d * cc)];
  HIPCHK(hipMemcpyAsync(c->d_xn_ext, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_H_ext, H.data(), H.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  c->ext_xn = c->d_xn_ext; c->ext_H = c->d_H_ext;
  c->fuse_resample = true;
  const int st = ctx_step(c, 0, nullptr, c->N, nullptr);
  c->ext_xn = nullptr; c->ext_H = nullptr;
  if (st != RBPF_OK) return st;
  HIPCHK(hipStreamSynchronize(c->stream));          // the host vectors above go out of scope
  return ctx_call_on_step(c, c->t - 1, false);
}

int rbpf_filter_reset(rbpf_ctx* c) {
  if (!c) { set_error("ctx is NULL"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  return ctx_reset(c);
}

int rbpf_sync(rbpf_ctx* c) {
  if (!c) { set_error("ctx is NULL"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  return ctx_check_flags(c);
}

int rbpf_filter_tell(const rbpf_ctx* c, int32_t* t) {
  if (!c || !t) { set_error("NULL argument"); return RBPF_ERR_INVALID_ARG; }
  *t = c->t;
  return RBPF_OK;
}

int rbpf_filter_schedule(const rbpf_ctx* c, int32_t* banks, int32_t* shared_flush) {
  if (!c || !banks || !shared_flush) { set_error("NULL argument"); return RBPF_ERR_INVALID_ARG; }
  *banks = c->inplace ? 1 : 2;
  *shared_flush = (c->share_flush || c->share_inplace) ? 1 : 0;
  return RBPF_OK;
}

int rbpf_timing_enable(rbpf_ctx* c, int32_t on) {
  if (!c) { set_error("ctx is NULL"); return RBPF_ERR_INVALID_ARG; }
  c->timing_on = on != 0;
  return RBPF_OK;
}

int rbpf_timing_read(rbpf_ctx* c, rbpf_timing* out, int32_t reset) {
  if (!c || !out) { set_error("NULL argument"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  double ms = 0.0;
  for (auto& ev : c->events) {
    float f = 0.f;
    HIPCHK(hipEventElapsedTime(&f, ev.first, ev.second));
    ms += f;
  }
  out->stream_kernel_ms = ms;
  out->stream_kernel_launches = (int64_t)c->events.size();
  const double n = c->mdl.n, nN = c->mdl.nN;
  out->algorithmic_bytes_per_launch = (double)c->N * (2.0 * n * n + 2.0 * n + 2.0 * nN) * (c->fp32 ? 4.0 : 8.0);   // SURVEY 8d, s = 4 | 8
  double sched = c->sched_bytes;
  if ((c->share_flush || c->share_inplace) && c->share_flush_particles > 0) {
    // shared flushes: only the writers stored a matrix (the accounting above charged every particle of a flush step with one)
    unsigned long long wr = 0;
    HIPCHK(hipMemcpy(&wr, c->d_share_writers, sizeof(wr), hipMemcpyDeviceToHost));
    const double stored = c->lay.sym ? (double)(c->lay.szT + c->lay.szB) : n * n;
    sched -= ((double)c->share_flush_particles - (double)wr) * stored * (c->fp32 ? 4.0 : 8.0);
  }
  if (c->d_distinct_counter && c->distinct_nominal > 0) {
    // particles that share a stored matrix (siblings, cousins) read it once from memory: charge the distinct matrices
    unsigned long long dn = 0;
    HIPCHK(hipMemcpy(&dn, c->d_distinct_counter, sizeof(dn), hipMemcpyDeviceToHost));
    const double stored = c->lay.sym ? (double)(c->lay.szT + c->lay.szB) : n * n;
    sched -= ((double)c->distinct_nominal - (double)dn) * stored * (c->fp32 ? 4.0 : 8.0);
  }
  out->scheduled_bytes_per_launch = c->events.empty() ? 0.0 : sched / (double)c->events.size();
  if (reset) {
    for (auto& ev : c->events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    c->events.clear();
    c->sched_bytes = 0.0;
    c->share_flush_particles = 0;
    if (c->d_share_writers) HIPCHK(hipMemset(c->d_share_writers, 0, sizeof(unsigned long long)));
    c->distinct_nominal = 0;
    if (c->d_distinct_counter) HIPCHK(hipMemset(c->d_distinct_counter, 0, sizeof(unsigned long long)));
  }
  return RBPF_OK;
}

int rbpf_destroy(rbpf_ctx* c) {
  if (!c) return RBPF_OK;
  hipSetDevice(c->device);
  ctx_free(c);
  return RBPF_OK;
}

// Final extraction: particleFilter.m:220-233
int rbpf_filter_finish(rbpf_ctx* c, rbpf_filter_out* o) {
  if (!c || !o) { set_error("NULL argument"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  if (c->t < 1) { set_error("finish before any step"); return RBPF_ERR_STATE; }
  RB_TRY(ctx_check_flags(c));
  const int N = c->N, T = c->T, nN = c->mdl.nN, n = c->mdl.n;
  const int Tdone = c->t;
  const Layout& L = c->lay;
  const int cur = c->xcur;          // bank of the means (== covariance bank unless the lazy update is on)
  int iw = 0;
  HIPCHK(hipMemcpy(&iw, c->d_flags + 2, sizeof(int), hipMemcpyDeviceToHost));
  if (o->iw_max) *o->iw_max = iw;
  const size_t tr_last = c->opt.trace ? (size_t)(Tdone - 1) * N : 0;
  const double* w_last = c->w + tr_last;
  // traj_max / traj_mean are kept [T][nN] on the device == MATLAB [nN x T] column-major
  if (o->traj_max) {
    for (size_t q = 0; q < (size_t)nN * T; ++q) o->traj_max[q] = NAN;      // particleFilter.m:92
    HIPCHK(hipMemcpy(o->traj_max, c->traj_max, (size_t)Tdone * nN * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (o->traj_mean) {
    for (size_t q = 0; q < (size_t)nN * T; ++q) o->traj_mean[q] = NAN;
    HIPCHK(hipMemcpy(o->traj_mean, c->traj_mean, (size_t)Tdone * nN * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (o->xl_max) HIPCHK(hipMemcpy(o->xl_max, c->xl[cur] + (size_t)iw * L.ldx, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<double> xl_mean;
  if (o->xl_mean || o->P_mean) {
    double* dm = nullptr;
    RB_TRY(dmalloc(&dm, (size_t)n));
    hipError_t e = launch_weighted_mean_xl(N, n, L.ldx, c->xl[cur], w_last, dm, c->stream);
    xl_mean.resize(n);
    if (e == hipSuccess) e = hipMemcpyAsync(xl_mean.data(), dm, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(dm);
    HIPCHK(e);
    if (o->xl_mean) std::memcpy(o->xl_mean, xl_mean.data(), (size_t)n * sizeof(double));
  }
  if (o->P_max || o->P_mean) {
    double* dP = nullptr; int* didx = nullptr;
    RB_TRY(dmalloc(&dP, (size_t)n * n));
    int s2 = dmalloc(&didx, 1);
    if (s2 != RBPF_OK) { hipFree(dP); return s2; }
    hipError_t e = hipSuccess;
    if (o->P_max) {
      e = hipMemcpy(didx, &iw, sizeof(int), hipMemcpyHostToDevice);
      if (e == hipSuccess && ctx_unpack(c, didx, 1, dP) != RBPF_OK) e = hipErrorUnknown;
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e == hipSuccess) e = hipMemcpy(o->P_max, dP, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess && o->P_mean) {
      if (c->opt.fix_p_mean) {
        hipFree(dP); hipFree(didx);
        return accumulate_p_mean(c, w_last, xl_mean, o->P_mean);   // consciously fixed quirk Q3 (option, not the default)
      }
      // quirk Q3 (particleFilter.m:228-230): P_mean = w(N)*(P(:,:,N) + (xl_mean-xl(:,N))*(...)')
      const int last = N - 1;
      std::vector<double> Pl((size_t)n * n), xll(n);
      double wl = 0.0;
      e = hipMemcpy(didx, &last, sizeof(int), hipMemcpyHostToDevice);
      if (e == hipSuccess && ctx_unpack(c, didx, 1, dP) != RBPF_OK) e = hipErrorUnknown;
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e == hipSuccess) e = hipMemcpy(Pl.data(), dP, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(xll.data(), c->xl[cur] + (size_t)last * L.ldx, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(&wl, w_last + last, sizeof(double), hipMemcpyDeviceToHost);
      if (e == hipSuccess)
        for (int cc = 0; cc < n; ++cc)
          for (int r = 0; r < n; ++r)
            o->P_mean[r + (size_t)n * cc] = wl * (Pl[r + (size_t)n * cc] + (xl_mean[r] - xll[r]) * (xl_mean[cc] - xll[cc]));
    }
    hipFree(dP); hipFree(didx);
    HIPCHK(e);
  }
  if (o->traj_sample_iwmax || o->xn_traj) {
    if (!c->opt.keep_history) { set_error("traj_sample_iwmax / xn_traj need keep_history=1"); return RBPF_ERR_STATE; }
    if (o->traj_sample_iwmax) {
      double* dout = nullptr; int* didx = nullptr;
      RB_TRY(dmalloc(&dout, (size_t)nN * Tdone));
      int s2 = dmalloc(&didx, 1);
      if (s2 != RBPF_OK) { hipFree(dout); return s2; }
      hipError_t e = hipMemcpy(didx, &iw, sizeof(int), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = launch_backtrace(N, nN, Tdone, c->X, c->A, didx, 1, dout, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e == hipSuccess) e = hipMemcpy(o->traj_sample_iwmax, dout, (size_t)nN * Tdone * sizeof(double), hipMemcpyDeviceToHost);
      hipFree(dout); hipFree(didx);
      HIPCHK(e);
    }
    if (o->xn_traj) {
      double* dout = nullptr;
      RB_TRY(dmalloc(&dout, (size_t)nN * N * Tdone));
      hipError_t e = launch_backtrace(N, nN, Tdone, c->X, c->A, nullptr, N, dout, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e == hipSuccess) e = hipMemcpy(o->xn_traj, dout, (size_t)nN * N * Tdone * sizeof(double), hipMemcpyDeviceToHost);
      hipFree(dout);
      HIPCHK(e);
    }
  }
  if (o->trace_logw || o->trace_w || o->trace_ai) {
    if (!c->opt.trace) { set_error("trace outputs need options.trace=1"); return RBPF_ERR_STATE; }
    if (o->trace_logw) HIPCHK(hipMemcpy(o->trace_logw, c->logw, (size_t)Tdone * N * sizeof(double), hipMemcpyDeviceToHost));
    if (o->trace_w) HIPCHK(hipMemcpy(o->trace_w, c->w, (size_t)Tdone * N * sizeof(double), hipMemcpyDeviceToHost));
    if (o->trace_ai) {
      if (!c->opt.keep_history) { set_error("trace_ai needs keep_history=1"); return RBPF_ERR_STATE; }
      HIPCHK(hipMemcpy(o->trace_ai, c->A, (size_t)Tdone * N * sizeof(int), hipMemcpyDeviceToHost));
    }
  }
  if (o->final_xn) {
    double* dout = nullptr;
    RB_TRY(dmalloc(&dout, (size_t)nN * N));
    const double* Xl = c->X + (size_t)(c->opt.keep_history ? Tdone - 1 : ((Tdone - 1) & 1)) * nN * N;
    hipError_t e = launch_transpose_soa(N, nN, Xl, dout, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(o->final_xn, dout, (size_t)nN * N * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(dout);
    HIPCHK(e);
  }
  if (o->final_xl) {
    double* dout = nullptr;
    RB_TRY(dmalloc(&dout, (size_t)n * N));
    hipError_t e = launch_gather_xl(N, n, L.ldx, c->xl[cur], dout, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(o->final_xl, dout, (size_t)n * N * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(dout);
    HIPCHK(e);
  }
  if (o->final_P) {
    double* dout = nullptr;
    RB_TRY(dmalloc(&dout, (size_t)n * n * N));
    hipError_t e = (ctx_unpack(c, nullptr, N, dout) == RBPF_OK) ? hipSuccess : hipErrorUnknown;
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(o->final_P, dout, (size_t)n * n * N * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(dout);
    HIPCHK(e);
  }
  return RBPF_OK;
}

int rbpf_particle_filter(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
                         rbpf_filter_out* out) {
  RB_TRY(options_ok(opt));
  if (wants_multi(opt)) return multi_particle_filter(model, prob, rng, opt, out);      // sharded over several GPUs in this process
  rbpf_ctx* c = nullptr;
  int s = rbpf_filter_create(model, prob, rng, opt, &c);
  if (s != RBPF_OK) return s;
  s = rbpf_filter_advance(c, prob->N_T);
  if (s == RBPF_OK) s = rbpf_filter_finish(c, out);
  rbpf_destroy(c);
  return s;
}

// ---- helper kernels for parity tests ----------------------------------------------------------------
int rbpf_philox_fill(uint64_t seed, int32_t k_iter, int32_t N, int32_t T, int32_t nw, double* U, double* Z, double* Ufin) {
  if (!have_device()) { set_error("no HIP device"); return RBPF_ERR_NO_DEVICE; }
  if (nw > 8 || N < 1 || T < 1) { set_error("bad sizes"); return RBPF_ERR_INVALID_ARG; }
  const size_t nu = (size_t)N * (T - 1);
  double *dU = nullptr, *dZ = nullptr, *dF = nullptr;
  RB_TRY(dmalloc(&dU, std::max<size_t>(nu, 1)));
  int s = dmalloc(&dZ, std::max<size_t>(nu * nw, 1));
  if (s == RBPF_OK) s = dmalloc(&dF, 1);
  hipError_t e = hipSuccess;
  if (s == RBPF_OK) {
    e = launch_philox_fill(seed, k_iter, N, T, nw, dU, dZ, dF, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && U && nu) e = hipMemcpy(U, dU, nu * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && Z && nu) e = hipMemcpy(Z, dZ, nu * nw * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && Ufin) e = hipMemcpy(Ufin, dF, sizeof(double), hipMemcpyDeviceToHost);
  }
  hipFree(dU); hipFree(dZ); hipFree(dF);
  if (s != RBPF_OK) return s;
  HIPCHK(e);
  return RBPF_OK;
}

}  // extern "C"

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  int alloc(size_t bytes) {
    hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 8));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
    return RBPF_OK;
  }
  template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

static int model_for_helpers(const rbpf_model* model, int nN, int nw, int nodo, ModelDev& M, DevBuf& nnbuf) {
  int n = 0, d = 0;
  if (!model) { set_error("model is NULL"); return RBPF_ERR_INVALID_ARG; }
  if (model->kind == RBPF_MODEL_DENSE_MAG_6D) { n = model->m_basis + 3; d = 3; }
  else if (model->kind == RBPF_MODEL_DENSE_RADIO_2DH) { n = model->m_basis; d = 1; }
  else { set_error("unknown model family"); return RBPF_ERR_UNSUPPORTED; }
  std::vector<int> nn;
  RB_TRY(fill_model_dev(model, nN, n, d, nw, nodo, nullptr, 0.0, M, nn));
  RB_TRY(nnbuf.alloc(nn.size() * sizeof(int)));
  HIPCHK(hipMemcpy(nnbuf.p, nn.data(), nn.size() * sizeof(int), hipMemcpyHostToDevice));
  M.NN = nnbuf.as<int>();
  return RBPF_OK;
}

static void default_dims(const rbpf_model* model, int& nN, int& nw, int& nodo) {
  if (model && model->kind == RBPF_MODEL_DENSE_MAG_6D) { nN = 7; nw = 6; nodo = 7; }
  else { nN = 3; nw = 1; nodo = 3; }
}

extern "C" {

int rbpf_meas_model(const rbpf_model* model, int32_t n_nonlin, int32_t n_pred, const double* xn, double* dy) {
  if (!have_device()) { set_error("no HIP device"); return RBPF_ERR_NO_DEVICE; }
  if (!xn || !dy || n_pred < 1) { set_error("bad argument"); return RBPF_ERR_INVALID_ARG; }
  int nN, nw, nodo; default_dims(model, nN, nw, nodo);
  if (n_nonlin != nN) { set_error("n_nonlin does not match the model family"); return RBPF_ERR_INVALID_ARG; }
  ModelDev M; DevBuf nnb, dx, dd;
  RB_TRY(model_for_helpers(model, nN, nw, nodo, M, nnb));
  RB_TRY(dx.alloc((size_t)nN * n_pred * sizeof(double)));
  RB_TRY(dd.alloc((size_t)M.d * M.n * n_pred * sizeof(double)));
  HIPCHK(hipMemcpy(dx.p, xn, (size_t)nN * n_pred * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(launch_meas_model(M, n_pred, dx.as<double>(), dd.as<double>(), 0));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(dy, dd.p, (size_t)M.d * M.n * n_pred * sizeof(double), hipMemcpyDeviceToHost));
  return RBPF_OK;
}

static int chol_for_helper(const rbpf_model* model, int nw, double dt, const double* Q, std::vector<double>& blk,
                           std::vector<double>& full) {
  rbpf_problem p;
  std::memset(&p, 0, sizeof(p));
  p.N_T = 2; p.n_w = nw; p.q_pages = 1; p.dt_len = 1; p.Q = Q; p.dt = &dt;
  int pages = 0;
  return build_chol_factors(model, &p, blk, full, pages);
}

int rbpf_dyn_model(const rbpf_model* model, int32_t n_nonlin, int32_t n_w, int32_t n_odo, int32_t n_p, const double* xn,
                   const double* odo, double dt, const double* Q, const double* z, double* xn_next) {
  if (!have_device()) { set_error("no HIP device"); return RBPF_ERR_NO_DEVICE; }
  if (!xn || !odo || !Q || !z || !xn_next || n_p < 1) { set_error("bad argument"); return RBPF_ERR_INVALID_ARG; }
  int nN, nw, nodo; default_dims(model, nN, nw, nodo);
  if (n_nonlin != nN || n_w != nw || n_odo != nodo) { set_error("dims do not match the model family"); return RBPF_ERR_INVALID_ARG; }
  ModelDev M; DevBuf nnb, dx, dodo, dL, dz, dout;
  RB_TRY(model_for_helpers(model, nN, nw, nodo, M, nnb));
  std::vector<double> blk, full;
  RB_TRY(chol_for_helper(model, nw, dt, Q, blk, full));
  RB_TRY(dx.alloc((size_t)nN * n_p * 8)); RB_TRY(dodo.alloc((size_t)nodo * 8)); RB_TRY(dL.alloc((size_t)nw * nw * 8));
  RB_TRY(dz.alloc((size_t)nw * n_p * 8)); RB_TRY(dout.alloc((size_t)nN * n_p * 8));
  HIPCHK(hipMemcpy(dx.p, xn, (size_t)nN * n_p * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dodo.p, odo, (size_t)nodo * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dL.p, blk.data(), (size_t)nw * nw * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dz.p, z, (size_t)nw * n_p * 8, hipMemcpyHostToDevice));
  HIPCHK(launch_dyn_model(M, n_p, dx.as<double>(), dodo.as<double>(), dL.as<double>(), dz.as<double>(), dout.as<double>(), 0));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(xn_next, dout.p, (size_t)nN * n_p * 8, hipMemcpyDeviceToHost));
  return RBPF_OK;
}

int rbpf_dyn_res_norm(const rbpf_model* model, int32_t n_nonlin, int32_t n_w, int32_t n_odo, int32_t n_p,
                      const double* xnk_t, const double* xn, const double* odo, double dt, const double* Q, double* e_dyn) {
  if (!have_device()) { set_error("no HIP device"); return RBPF_ERR_NO_DEVICE; }
  if (!xnk_t || !xn || !odo || !Q || !e_dyn || n_p < 1) { set_error("bad argument"); return RBPF_ERR_INVALID_ARG; }
  int nN, nw, nodo; default_dims(model, nN, nw, nodo);
  if (n_nonlin != nN || n_odo != nodo) { set_error("dims do not match the model family"); return RBPF_ERR_INVALID_ARG; }
  if (!model->use_dyn_res_norm) nw = nN;    // additive default: residual over all non-linear states
  if (n_w != nw) { set_error("n_w does not match"); return RBPF_ERR_INVALID_ARG; }
  ModelDev M; DevBuf nnb, dk, dx, dodo, dL, dout;
  RB_TRY(model_for_helpers(model, nN, model->use_dyn_res_norm ? nw : (model->kind == RBPF_MODEL_DENSE_MAG_6D ? 6 : 1), nodo, M, nnb));
  M.nw = nw;
  std::vector<double> Lf((size_t)nw * nw, 0.0), A((size_t)nw * nw);
  for (int q = 0; q < nw * nw; ++q) A[q] = dt * Q[q];
  if (!chol_lower_host(A.data(), nw, nw, Lf.data(), nw)) { set_error("chol(dt*Q) failed"); return RBPF_ERR_CHOL_FAILED; }
  RB_TRY(dk.alloc((size_t)nN * 8)); RB_TRY(dx.alloc((size_t)nN * n_p * 8)); RB_TRY(dodo.alloc((size_t)nodo * 8));
  RB_TRY(dL.alloc((size_t)nw * nw * 8)); RB_TRY(dout.alloc((size_t)nw * n_p * 8));
  HIPCHK(hipMemcpy(dk.p, xnk_t, (size_t)nN * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dx.p, xn, (size_t)nN * n_p * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dodo.p, odo, (size_t)nodo * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dL.p, Lf.data(), (size_t)nw * nw * 8, hipMemcpyHostToDevice));
  HIPCHK(launch_dyn_res_norm(M, n_p, dk.as<double>(), dx.as<double>(), dodo.as<double>(), dL.as<double>(), dout.as<double>(), 0));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(e_dyn, dout.p, (size_t)nw * n_p * 8, hipMemcpyDeviceToHost));
  return RBPF_OK;
}

int rbpf_sample(int32_t N, const double* w, int32_t n_draws, const double* u, int32_t* ind) {
  if (!have_device()) { set_error("no HIP device"); return RBPF_ERR_NO_DEVICE; }
  if (!w || !u || !ind || N < 1 || n_draws < 1) { set_error("bad argument"); return RBPF_ERR_INVALID_ARG; }
  // reuse normalise_scan's cumsum by feeding log(w): instead run the dedicated path: upload w as
  // already-normalised weights through logw = log(w) would re-normalise; so scan directly.
  DevBuf dw, dwc, du, di, dlog, dx, dflag;
  RB_TRY(dw.alloc((size_t)N * 8)); RB_TRY(dwc.alloc((size_t)N * 8)); RB_TRY(du.alloc((size_t)n_draws * 8));
  RB_TRY(di.alloc((size_t)n_draws * 4)); RB_TRY(dflag.alloc(16));
  HIPCHK(hipMemcpy(dw.p, w, (size_t)N * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(du.p, u, (size_t)n_draws * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dflag.p, 0, 16));
  HIPCHK(launch_cumsum(N, dw.as<double>(), dwc.as<double>(), 0));
  SearchArgs s;
  s.N = N; s.n_draw = n_draws; s.t = 0; s.wc = dwc.as<double>(); s.rng_mode = 0; s.k_iter = 0; s.U = du.as<double>();
  s.seed = 0; s.ai = di.as<int>(); s.overflow = dflag.as<int>();
  HIPCHK(launch_search(s, 0));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(ind, di.p, (size_t)n_draws * 4, hipMemcpyDeviceToHost));
  return RBPF_OK;
}

int rbpf_quat_helpers(int32_t op, int32_t n, const double* in, double* out) {
  if (!have_device()) { set_error("no HIP device"); return RBPF_ERR_NO_DEVICE; }
  if (op < 0 || op > 8 || n < 1 || !in || !out) { set_error("bad argument"); return RBPF_ERR_INVALID_ARG; }
  static const int nin_of[9] = {3, 3, 4, 4, 4, 4, 4, 4, 3}, nout_of[9] = {4, 4, 3, 3, 16, 16, 4, 9, 9};
  DevBuf di, dout;
  RB_TRY(di.alloc((size_t)n * nin_of[op] * 8)); RB_TRY(dout.alloc((size_t)n * nout_of[op] * 8));
  HIPCHK(hipMemcpy(di.p, in, (size_t)n * nin_of[op] * 8, hipMemcpyHostToDevice));
  HIPCHK(launch_quat_helpers(op, n, di.as<double>(), dout.as<double>(), 0));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, dout.p, (size_t)n * nout_of[op] * 8, hipMemcpyDeviceToHost));
  return RBPF_OK;
}

int rbpf_probe_wave_reduce(const double* in, double* out) {
  if (!have_device()) { set_error("no HIP device"); return RBPF_ERR_NO_DEVICE; }
  if (!in || !out) { set_error("bad argument"); return RBPF_ERR_INVALID_ARG; }
  DevBuf di, dout;
  RB_TRY(di.alloc(256 * 8)); RB_TRY(dout.alloc(4 * 8));
  HIPCHK(hipMemcpy(di.p, in, 256 * 8, hipMemcpyHostToDevice));
  HIPCHK(launch_probe_wave_reduce(di.as<double>(), dout.as<double>(), 0));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, dout.p, 4 * 8, hipMemcpyDeviceToHost));
  return RBPF_OK;
}

int rbpf_jacobian_phi3d(const rbpf_model* model, int32_t n_p, const double* x, const double* lower, const double* upper, double* J) {
  if (!have_device()) { set_error("no HIP device"); return RBPF_ERR_NO_DEVICE; }
  if (!model || model->kind != RBPF_MODEL_DENSE_MAG_6D || !x || !lower || !upper || !J || n_p < 1) { set_error("bad argument"); return RBPF_ERR_INVALID_ARG; }
  ModelDev M; DevBuf nnb, dx, dlo, dup, dj;
  RB_TRY(model_for_helpers(model, 7, 6, 7, M, nnb));
  RB_TRY(dx.alloc((size_t)3 * n_p * 8)); RB_TRY(dlo.alloc(24)); RB_TRY(dup.alloc(24)); RB_TRY(dj.alloc((size_t)9 * M.m * n_p * 8));
  HIPCHK(hipMemcpy(dx.p, x, (size_t)3 * n_p * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dlo.p, lower, 24, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dup.p, upper, 24, hipMemcpyHostToDevice));
  HIPCHK(launch_jacobian_phi3d(M, n_p, dx.as<double>(), dlo.as<double>(), dup.as<double>(), dj.as<double>(), 0));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(J, dj.p, (size_t)9 * M.m * n_p * 8, hipMemcpyDeviceToHost));
  return RBPF_OK;
}

}  // extern "C"
