// MEX gateway: binds the C ABI of include/rbpf.h for MATLAB (build: see INTEGRATION.md).
//
// MATLAB and its mex.h exist neither in the build image nor on the GPU box; the gateway is compiled and EXECUTED in the test
// suite against a test double of mex.h (tests/mexdouble/, tests/test_gpu_mex_gateway.py), which drives mexFunction the way
// MATLAB does.  It only marshals mxArrays into the rbpf_* structs and MATLAB handles into rbpf_callbacks; all arithmetic is
// in librbpf_hip.so.
//
//   [traj_max,traj_mean,xl_max,xl_mean,P_max,P_mean,traj_sample_iwmax,xn_traj] = ...
//       rbpf_mex('filter', desc, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, dt, rng, makePlots)
//   [XNK,XLK,PK] = rbpf_mex('smoother', desc, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, N_K, dt, rng, info_form, makePlots)
//   xn_next = rbpf_mex('dynModel', desc, xn [nN x Np], dx [1 x n_odo], dt, Q, z [nw x Np])      run_dense3D_magfield.m:301-308
//   dy      = rbpf_mex('measModel', desc, xn [nN x Np])     -> [Np x ny x nLin] ([Np x nLin] for ny = 1)        :265-279
//   eDyn    = rbpf_mex('dynResNorm', desc, xnk [nN], xni [nN x Np], dx, dt, Q) -> [Np x nw] (row per particle)  :202-203
//
// desc : struct made by matlab/rbpf_recognise.m / rbpf_model.m
//          kind 1 dense-mag-6D | 2 dense-radio-2D+heading : NN (m x dim), L (1 x dim), use_dyn_res_norm
//          kind 3 sparse-visual                            : nLand, cam [f fp fw]
//          kind 4 generic (any handles)                    : dynModel, measModel, dynResNorm (function handles; [] = the
//                                                            additive default of particleSmoother.m:175-177)
// rng  : struct, mode 'replay' (U, Z, Ufin drawn by the wrapper with MATLAB's own rand / randn) or 'philox' (seed)
// makePlots : [] or the reference's plot handle (particleFilter.m:215-217 / particleSmoother.m:360-362), called through the
//             library's on_step hook
#include "mex.h"
#include "../include/rbpf.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Gateway {                         // everything the callbacks need; lives on mexFunction's stack
  const rbpf_problem* prob = nullptr;
  mxArray* h_dyn = nullptr;              // function handles (borrowed from desc)
  mxArray* h_meas = nullptr;
  mxArray* h_drn = nullptr;
  mxArray* h_plots = nullptr;
  std::string error;                     // first MATLAB error raised inside a callback
  // smoother hook: the output arrays (pages beyond the finished iteration are still NaN, as in the reference)
  mxArray *XNK = nullptr, *XLK = nullptr, *PK = nullptr;
  int N_K = 0;
};

void fail(int status) { mexErrMsgIdAndTxt("rbpf:status", "%s (%s)", rbpf_last_error(), rbpf_status_string(status)); }

std::string trap_text(mxArray* me) {
  char buf[1024] = "MATLAB error inside a callback";
  if (me && mxIsChar(me)) mxGetString(me, buf, sizeof(buf));
#ifndef RBPF_TEST_MEX_H_
  else if (me) { mxArray* msg = mxGetProperty(me, 0, "message"); if (msg) { mxGetString(msg, buf, sizeof(buf)); mxDestroyArray(msg); } }
#endif
  if (me) mxDestroyArray(me);
  return buf;
}

// call name(args...) -> one output; returns nullptr and records the error text on failure
mxArray* call1(Gateway* g, const char* name, int nrhs, mxArray** prhs) {
  mxArray* out = nullptr;
  mxArray* me = mexCallMATLABWithTrap(1, &out, nrhs, prhs, name);
  if (me) { if (g->error.empty()) g->error = trap_text(me); else mxDestroyArray(me); return nullptr; }
  return out;
}

mxArray* row_of(const double* base, int ld, int row, int n) {                 // odometry(t,:) as 1 x n
  mxArray* a = mxCreateDoubleMatrix(1, n, mxREAL);
  for (int k = 0; k < n; ++k) mxGetPr(a)[k] = base[row + (size_t)ld * k];
  return a;
}

mxArray* q_page(const rbpf_problem* p, int t) {                               // Q(:,:,t)
  mxArray* a = mxCreateDoubleMatrix(p->n_w, p->n_w, mxREAL);
  std::memcpy(mxGetPr(a), p->Q + (size_t)(p->q_pages > 1 ? t : 0) * p->n_w * p->n_w, sizeof(double) * p->n_w * p->n_w);
  return a;
}

// rbpf_callbacks.dyn_model: one mexCallMATLAB per step -- rbpf_batch_dyn.m loops xn_new(:,j) = dynModel(xn(:,j),dx,dt,Q) in
// MATLAB, in slot order, so the handle draws its random numbers from MATLAB's global stream as in the reference
int cb_dyn(void* user, int32_t t, int32_t n_cols, const double* xn_anc, double* xn_new) {
  Gateway* g = static_cast<Gateway*>(user);
  const rbpf_problem* p = g->prob;
  mxArray* args[5];
  args[0] = g->h_dyn;
  args[1] = mxCreateDoubleMatrix(p->n_nonlin, n_cols, mxREAL);
  std::memcpy(mxGetPr(args[1]), xn_anc, sizeof(double) * p->n_nonlin * n_cols);
  args[2] = row_of(p->odometry, p->odo_ld, t, p->n_odo);
  args[3] = mxCreateDoubleScalar(p->dt[p->dt_len > 1 ? t : 0]);
  args[4] = q_page(p, t);
  mxArray* out = call1(g, "rbpf_batch_dyn", 5, args);
  for (int q = 1; q < 5; ++q) mxDestroyArray(args[q]);
  if (!out) return 1;
  int rc = 0;
  if (!mxIsDouble(out) || mxGetNumberOfElements(out) != (size_t)p->n_nonlin * n_cols) { g->error = "dynModel returned the wrong number of elements"; rc = 1; }
  else std::memcpy(xn_new, mxGetPr(out), sizeof(double) * p->n_nonlin * n_cols);
  mxDestroyArray(out);
  return rc;
}

int cb_meas(void* user, int32_t n_cols, const double* xn, double* dy) {
  Gateway* g = static_cast<Gateway*>(user);
  const rbpf_problem* p = g->prob;
  mxArray* args[2];
  args[0] = g->h_meas;
  args[1] = mxCreateDoubleMatrix(p->n_nonlin, n_cols, mxREAL);
  std::memcpy(mxGetPr(args[1]), xn, sizeof(double) * p->n_nonlin * n_cols);
  mxArray* out = call1(g, "feval", 2, args);                                  // dy = measModel(xn), particleFilter.m:124
  mxDestroyArray(args[1]);
  if (!out) return 1;
  int rc = 0;
  const size_t want = (size_t)n_cols * p->n_y * p->n_lin;
  if (!mxIsDouble(out) || mxGetNumberOfElements(out) != want) { g->error = "measModel must return [Npred x ny x nLin] ([Npred x nLin] for ny = 1)"; rc = 1; }
  else std::memcpy(dy, mxGetPr(out), sizeof(double) * want);
  mxDestroyArray(out);
  return rc;
}

int cb_drn(void* user, int32_t t, int32_t n_cols, const double* xnk_t, const double* xn, double* e_dyn) {
  Gateway* g = static_cast<Gateway*>(user);
  const rbpf_problem* p = g->prob;
  mxArray* args[6];
  args[0] = g->h_drn;
  args[1] = mxCreateDoubleMatrix(p->n_nonlin, 1, mxREAL);
  std::memcpy(mxGetPr(args[1]), xnk_t, sizeof(double) * p->n_nonlin);
  args[2] = mxCreateDoubleMatrix(p->n_nonlin, n_cols, mxREAL);
  std::memcpy(mxGetPr(args[2]), xn, sizeof(double) * p->n_nonlin * n_cols);
  args[3] = row_of(p->odometry, p->odo_ld, t, p->n_odo);
  args[4] = mxCreateDoubleScalar(p->dt[p->dt_len > 1 ? t : 0]);
  args[5] = q_page(p, t);
  mxArray* out = call1(g, "rbpf_batch_drn", 6, args);                         // [nw x n_cols]
  for (int q = 1; q < 6; ++q) mxDestroyArray(args[q]);
  if (!out) return 1;
  int rc = 0;
  if (!mxIsDouble(out) || mxGetNumberOfElements(out) != (size_t)p->n_w * n_cols) { g->error = "dynResNorm must return size(Q,1) values"; rc = 1; }
  else std::memcpy(e_dyn, mxGetPr(out), sizeof(double) * p->n_w * n_cols);
  mxDestroyArray(out);
  return rc;
}

// on_step hook -> makePlots
int cb_on_step(const rbpf_view* v, void* user) {
  Gateway* g = static_cast<Gateway*>(user);
  const rbpf_problem* p = g->prob;
  if (!g->h_plots) return 0;
  if (v->is_smoother) {                                                       // particleSmoother.m:360-362
    const int k = v->t;
    const mwSize nN = p->n_nonlin, n = p->n_lin, T = p->N_T;
    mxArray* args[7];
    args[0] = g->h_plots;
    args[1] = mxCreateDoubleMatrix(nN, T, mxREAL);
    std::memcpy(mxGetPr(args[1]), mxGetPr(g->XNK) + (size_t)k * nN * T, sizeof(double) * nN * T);
    args[2] = mxCreateDoubleMatrix(n, 1, mxREAL);
    if (g->XLK) std::memcpy(mxGetPr(args[2]), mxGetPr(g->XLK) + (size_t)k * n, sizeof(double) * n);
    args[3] = mxCreateDoubleScalar((double)(k + 1));
    args[4] = g->XNK; args[5] = g->XLK; args[6] = g->PK;
    mxArray* me = mexCallMATLABWithTrap(0, nullptr, 7, args, "feval");
    for (int q = 1; q < 4; ++q) mxDestroyArray(args[q]);
    if (me) { if (g->error.empty()) g->error = trap_text(me); else mxDestroyArray(me); return 1; }
    return 0;
  }
  // particleFilter.m:215-217: makePlots(xn,xl_max,P_max,traj_max,yhattraj,xn_traj,traj_mean,xl,P)
  const mwSize nN = p->n_nonlin, n = p->n_lin, N = p->N_P, T = p->N_T, Td = v->t + 1;
  const mwSize d3[3] = {nN, N, T}, dP[3] = {n, n, N}, dh[3] = {nN, N, Td};
  mxArray* xn = mxCreateDoubleMatrix(nN, N, mxREAL);
  mxArray* xl_max = mxCreateDoubleMatrix(n, 1, mxREAL);
  mxArray* P_max = mxCreateDoubleMatrix(n, n, mxREAL);
  mxArray* traj_max = mxCreateDoubleMatrix(nN, T, mxREAL);
  mxArray* yhattraj = mxCreateDoubleMatrix(p->n_y, T, mxREAL);
  mxArray* xn_traj = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
  mxArray* traj_mean = mxCreateDoubleMatrix(nN, T, mxREAL);
  mxArray* xl = mxCreateDoubleMatrix(n, N, mxREAL);
  mxArray* P = mxCreateNumericArray(3, dP, mxDOUBLE_CLASS, mxREAL);
  mxArray* hist = mxCreateNumericArray(3, dh, mxDOUBLE_CLASS, mxREAL);
  for (size_t q = 0; q < (size_t)p->n_y * T; ++q) mxGetPr(yhattraj)[q] = mxGetNaN();   // particleFilter.m:94: never filled
  rbpf_filter_out o;
  std::memset(&o, 0, sizeof(o));
  o.final_xn = mxGetPr(xn); o.xl_max = mxGetPr(xl_max); o.P_max = mxGetPr(P_max); o.traj_max = mxGetPr(traj_max);
  o.traj_mean = mxGetPr(traj_mean); o.final_xl = mxGetPr(xl); o.final_P = mxGetPr(P); o.xn_traj = mxGetPr(hist);
  const int st = rbpf_filter_finish(v->ctx, &o);
  std::memcpy(mxGetPr(xn_traj), mxGetPr(hist), sizeof(double) * nN * N * Td);           // later pages stay zero (:91)
  mxArray* args[10] = {g->h_plots, xn, xl_max, P_max, traj_max, yhattraj, xn_traj, traj_mean, xl, P};
  mxArray* me = (st == RBPF_OK) ? mexCallMATLABWithTrap(0, nullptr, 10, args, "feval") : nullptr;
  for (int q = 1; q < 10; ++q) mxDestroyArray(args[q]);
  mxDestroyArray(hist);
  if (st != RBPF_OK) { g->error = rbpf_last_error(); return 1; }
  if (me) { if (g->error.empty()) g->error = trap_text(me); else mxDestroyArray(me); return 1; }
  return 0;
}

bool is_handle(const mxArray* a) { return a && !mxIsEmpty(a) && mxIsClass(a, "function_handle"); }

rbpf_model model_from(const mxArray* d, std::vector<int32_t>& nn, Gateway& g, rbpf_callbacks& cb) {
  rbpf_model m;
  std::memset(&m, 0, sizeof(m));
  if (!d || !mxIsStruct(d) || !mxGetField(d, 0, "kind")) mexErrMsgIdAndTxt("rbpf:usage", "desc must be a struct with a field 'kind'");
  m.kind = (int32_t)mxGetScalar(mxGetField(d, 0, "kind"));
  if (m.kind == RBPF_MODEL_SPARSE_VISUAL_2D) {                 // examples/slam-sparse-visual: camera f, fp, fw
    m.m_basis = (int32_t)mxGetScalar(mxGetField(d, 0, "nLand"));
    m.dim = 2;
    const double* cam = mxGetPr(mxGetField(d, 0, "cam"));
    for (int a = 0; a < 3; ++a) m.cam[a] = cam[a];
    m.use_dyn_res_norm = 0;                                    // psslam.m passes dynResNorm = []
    return m;
  }
  if (m.kind == RBPF_MODEL_GENERIC_DENSE) {                    // arbitrary handles: evaluated in MATLAB through the callbacks
    g.h_dyn = mxGetField(d, 0, "dynModel");
    g.h_meas = mxGetField(d, 0, "measModel");
    mxArray* drn = mxGetField(d, 0, "dynResNorm");
    g.h_drn = is_handle(drn) ? drn : nullptr;
    if (!is_handle(g.h_dyn) || !is_handle(g.h_meas)) mexErrMsgIdAndTxt("rbpf:usage", "generic desc needs function handles dynModel and measModel");
    cb.dyn_model = cb_dyn; cb.meas_model = cb_meas; cb.dyn_res_norm = g.h_drn ? cb_drn : nullptr; cb.user = &g;
    m.callbacks = &cb;
    m.use_dyn_res_norm = g.h_drn ? 1 : 0;
    return m;
  }
  const mxArray* NN = mxGetField(d, 0, "NN");
  const mxArray* L = mxGetField(d, 0, "L");
  if (!NN || !L) mexErrMsgIdAndTxt("rbpf:usage", "desc needs NN and L");
  m.m_basis = (int32_t)mxGetM(NN);
  m.dim = (int32_t)mxGetN(NN);
  nn.resize((size_t)m.m_basis * m.dim);
  if (mxIsInt32(NN)) std::memcpy(nn.data(), mxGetData(NN), nn.size() * sizeof(int32_t));
  else { const double* p = mxGetPr(NN); for (size_t q = 0; q < nn.size(); ++q) nn[q] = (int32_t)p[q]; }
  m.NN = nn.data();
  if ((int)mxGetNumberOfElements(L) < m.dim || m.dim > 3) mexErrMsgIdAndTxt("rbpf:usage", "L must have one half-width per basis dimension");
  for (int a = 0; a < m.dim; ++a) m.L[a] = mxGetPr(L)[a];
  const mxArray* u = mxGetField(d, 0, "use_dyn_res_norm");
  m.use_dyn_res_norm = u ? (int32_t)mxGetScalar(u) : 1;
  return m;
}

rbpf_problem problem_from(const mxArray* odo, const mxArray* y, const mxArray* x0n, const mxArray* x0l,
                          const mxArray* P0, const mxArray* Q, const mxArray* R, const mxArray* NP, const mxArray* dt) {
  rbpf_problem p;
  std::memset(&p, 0, sizeof(p));
  p.N_P = (int32_t)mxGetScalar(NP);
  p.N_T = (int32_t)mxGetM(y);
  p.n_y = (int32_t)mxGetN(y);
  p.n_nonlin = (int32_t)mxGetNumberOfElements(x0n);
  p.n_lin = (int32_t)mxGetM(x0l);
  p.x0_lin_cols = (int32_t)mxGetN(x0l);
  p.n_w = (int32_t)mxGetM(Q);
  const mwSize* qd = mxGetDimensions(Q);
  p.q_pages = mxGetNumberOfDimensions(Q) > 2 ? (int32_t)qd[2] : 1;
  p.dt_len = (int32_t)mxGetNumberOfElements(dt);
  p.n_odo = (int32_t)mxGetN(odo);
  p.odo_ld = (int32_t)mxGetM(odo);
  p.odometry = mxGetPr(odo); p.y = mxGetPr(y); p.x0_nonlin = mxGetPr(x0n); p.x0_lin = mxGetPr(x0l);
  p.P0_lin = mxGetPr(P0); p.Q = mxGetPr(Q); p.R = mxGetPr(R); p.dt = mxGetPr(dt);
  return p;
}

rbpf_rng rng_from(const mxArray* r, int n_iter) {
  rbpf_rng g;
  std::memset(&g, 0, sizeof(g));
  char mode[16] = {0};
  if (!r || !mxIsStruct(r) || !mxGetField(r, 0, "mode")) mexErrMsgIdAndTxt("rbpf:usage", "rng must be a struct with a field 'mode'");
  mxGetString(mxGetField(r, 0, "mode"), mode, sizeof(mode));
  g.n_iter = n_iter;
  if (std::string(mode) == "replay") {
    // U [N_P x (N_T-1) x n_iter], Z [n_w x N_P x (N_T-1) x n_iter], drawn with MATLAB's own rand / randn in
    // the reference's interleaved order by the wrapper (matlab/particleFilter.m)
    g.mode = RBPF_RNG_REPLAY;
    g.U = mxGetPr(mxGetField(r, 0, "U"));
    const mxArray* z = mxGetField(r, 0, "Z");
    g.Z = z ? mxGetPr(z) : nullptr;
    const mxArray* uf = mxGetField(r, 0, "Ufin");
    g.Ufin = uf ? mxGetPr(uf) : nullptr;
  } else {
    g.mode = RBPF_RNG_PHILOX;
    g.seed = (uint64_t)mxGetScalar(mxGetField(r, 0, "seed"));
  }
  return g;
}

void raise(Gateway& g, int st) {
  if (st == RBPF_ERR_CALLBACK && !g.error.empty()) mexErrMsgIdAndTxt("rbpf:callback", "%s", g.error.c_str());
  fail(st);
}

void fill_nan(mxArray* a) { double* p = mxGetPr(a); const size_t n = mxGetNumberOfElements(a); for (size_t q = 0; q < n; ++q) p[q] = mxGetNaN(); }

}  // namespace

// Options that change the schedule or the arithmetic, not the interface (include/rbpf.h `rbpf_options`): they cannot travel in
// the reference's signatures, so they are session state of the gateway, set once by  rbpf_mex('options', struct(...))  (see
// matlab/rbpf_options.m) and applied to every later filter / smoother call.  All zero = the reference's behaviour.
struct SessionOptions {
  int lazy_depth = 0, chol_refresh = 0, chol_variant = 0, storage = 0, inplace = 0, fix_p_mean = 0, n_devices = 0, rng_mode = 0, info_rebuild = 0;
  double jitter = 0.0, rng_seed = 0.0;
  std::vector<int32_t> device_ids;                 // [n_devices] HIP device of every rank (empty: 0 .. n_devices-1)
};
SessionOptions g_session;

void session_field(const mxArray* s, const char* name, int& v) {
  if (const mxArray* f = mxGetField(s, 0, name)) { if (!mxIsEmpty(f)) v = (int)mxGetScalar(f); }
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("rbpf:usage", "first argument must be a command string");
  {
    // the gateway was compiled against include/rbpf.h; the library it is linked / loaded with must be the same ABI (a stale
    // librbpf_hip.so would otherwise read option fields at other offsets)
    static bool abi_checked = false;
    if (!abi_checked) {
      if (rbpf_abi_version() != RBPF_ABI_VERSION || rbpf_abi_sizeof(3) != (int)sizeof(rbpf_options) || rbpf_abi_sizeof(0) != (int)sizeof(rbpf_model) ||
          rbpf_abi_sizeof(1) != (int)sizeof(rbpf_problem) || rbpf_abi_sizeof(2) != (int)sizeof(rbpf_rng))
        mexErrMsgIdAndTxt("rbpf:abi", "librbpf_hip.so and this MEX gateway were built against different versions of include/rbpf.h: rebuild both");
      abi_checked = true;
    }
  }
  char cmdbuf[24] = {0};
  mxGetString(prhs[0], cmdbuf, sizeof(cmdbuf));
  const std::string cmd(cmdbuf);
  std::vector<int32_t> nn;
  Gateway g;
  rbpf_callbacks cb;
  std::memset(&cb, 0, sizeof(cb));
  rbpf_options opt;
  std::memset(&opt, 0, sizeof(opt));
  opt.keep_history = 1;
  opt.lazy_depth = g_session.lazy_depth; opt.chol_refresh = g_session.chol_refresh; opt.chol_variant = g_session.chol_variant;
  opt.storage = g_session.storage; opt.inplace = g_session.inplace; opt.fix_p_mean = g_session.fix_p_mean; opt.jitter = g_session.jitter;
  opt.n_devices = g_session.n_devices;                       // > 1: the library shards the particles over that many GPUs itself (RCCL)
  opt.device_ids = g_session.device_ids.empty() ? nullptr : g_session.device_ids.data();
  opt.struct_size = (int32_t)sizeof(opt);
  opt.info_rebuild = g_session.info_rebuild;
  if (cmd == "options") {
    if (nrhs > 2 || (nrhs == 2 && !mxIsStruct(prhs[1]))) mexErrMsgIdAndTxt("rbpf:usage", "options expects one struct (or nothing: query)");
    if (nrhs == 2) {
      SessionOptions o;                                       // fields that are absent go back to zero
      session_field(prhs[1], "lazy_depth", o.lazy_depth); session_field(prhs[1], "chol_refresh", o.chol_refresh);
      session_field(prhs[1], "chol_variant", o.chol_variant); session_field(prhs[1], "storage", o.storage);
      session_field(prhs[1], "inplace", o.inplace); session_field(prhs[1], "fix_p_mean", o.fix_p_mean);
      session_field(prhs[1], "n_devices", o.n_devices); session_field(prhs[1], "rng_mode", o.rng_mode);
      session_field(prhs[1], "info_rebuild", o.info_rebuild);
      if (const mxArray* f = mxGetField(prhs[1], 0, "rng_seed")) { if (!mxIsEmpty(f)) o.rng_seed = mxGetScalar(f); }
      if (const mxArray* f = mxGetField(prhs[1], 0, "jitter")) { if (!mxIsEmpty(f)) o.jitter = mxGetScalar(f); }
      if (const mxArray* f = mxGetField(prhs[1], 0, "device_ids")) {
        if (!mxIsEmpty(f)) {
          if (!mxIsDouble(f) || (int)mxGetNumberOfElements(f) != o.n_devices) mexErrMsgIdAndTxt("rbpf:usage", "options: device_ids must hold n_devices device numbers");
          const double* dv = mxGetPr(f);
          for (int q = 0; q < o.n_devices; ++q) {
            if (dv[q] < 0 || dv[q] != (double)(int)dv[q]) mexErrMsgIdAndTxt("rbpf:usage", "options: device_ids must be non-negative integers (0-based HIP devices)");
            o.device_ids.push_back((int32_t)dv[q]);
          }
        }
      }
      if (o.lazy_depth < 0 || o.chol_refresh < 0 || o.storage < 0 || o.storage > 3 || o.n_devices < 0 || o.rng_mode < 0 || o.rng_mode > 2 || o.rng_seed < 0) mexErrMsgIdAndTxt("rbpf:usage", "options: value out of range");
      g_session = o;
    }
    // rng_mode / rng_seed are consumed by the .m wrappers (matlab/rbpf_rngblock.m): 0 = MATLAB's stream in the reference's
    // interleaved order (seed-exact), 1 = MATLAB's stream, vectorised draws, 2 = the device Philox generator keyed by rng_seed
    const char* names[] = {"lazy_depth", "chol_refresh", "chol_variant", "storage", "inplace", "fix_p_mean", "jitter", "n_devices", "rng_mode", "rng_seed", "info_rebuild", "device_ids"};
    plhs[0] = mxCreateStructMatrix(1, 1, 12, names);
    const double vals[] = {(double)g_session.lazy_depth, (double)g_session.chol_refresh, (double)g_session.chol_variant,
                           (double)g_session.storage, (double)g_session.inplace, (double)g_session.fix_p_mean, g_session.jitter,
                           (double)g_session.n_devices, (double)g_session.rng_mode, g_session.rng_seed, (double)g_session.info_rebuild};
    for (int q = 0; q < 11; ++q) mxSetField(plhs[0], 0, names[q], mxCreateDoubleScalar(vals[q]));
    mxArray* ids = mxCreateDoubleMatrix(1, g_session.device_ids.size(), mxREAL);
    for (size_t q = 0; q < g_session.device_ids.size(); ++q) mxGetPr(ids)[q] = (double)g_session.device_ids[q];
    mxSetField(plhs[0], 0, "device_ids", ids);
  } else if (cmd == "filter") {
    if (nrhs != 12 && nrhs != 13) mexErrMsgIdAndTxt("rbpf:usage", "filter expects 11 or 12 arguments after the command");
    rbpf_model m = model_from(prhs[1], nn, g, cb);
    rbpf_problem p = problem_from(prhs[2], prhs[3], prhs[4], prhs[5], prhs[6], prhs[7], prhs[8], prhs[9], prhs[10]);
    rbpf_rng r = rng_from(prhs[11], 1);
    std::vector<double> zeroZ;
    if (m.kind == RBPF_MODEL_GENERIC_DENSE && r.mode == RBPF_RNG_REPLAY && !r.Z) {      // the handle draws its own randn
      zeroZ.assign((size_t)p.n_w * p.N_P * (p.N_T > 1 ? p.N_T - 1 : 0) + 1, 0.0);
      r.Z = zeroZ.data();
    }
    g.prob = &p;
    if (nrhs == 13 && is_handle(prhs[12])) {
      if (opt.n_devices > 1 || opt.device_ids) mexErrMsgIdAndTxt("rbpf:unsupported", "particleFilter with rbpf_options('n_devices', W): the makePlots hook needs the particle cloud of every step on the host and is not served by the sharded filter -- pass makePlots = [] or reset n_devices");
      g.h_plots = const_cast<mxArray*>(prhs[12]); opt.on_step = cb_on_step; opt.on_step_user = &g;
    }
    const mwSize nN = p.n_nonlin, n = p.n_lin, N = p.N_P, T = p.N_T;
    rbpf_filter_out o;
    std::memset(&o, 0, sizeof(o));
    const mwSize d3[3] = {nN, N, T};
    plhs[0] = mxCreateDoubleMatrix(nN, T, mxREAL); o.traj_max = mxGetPr(plhs[0]);
    if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(nN, T, mxREAL); o.traj_mean = mxGetPr(plhs[1]); }
    if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix(n, 1, mxREAL); o.xl_max = mxGetPr(plhs[2]); }
    if (nlhs > 3) { plhs[3] = mxCreateDoubleMatrix(n, 1, mxREAL); o.xl_mean = mxGetPr(plhs[3]); }
    if (nlhs > 4) { plhs[4] = mxCreateDoubleMatrix(n, n, mxREAL); o.P_max = mxGetPr(plhs[4]); }
    if (nlhs > 5) { plhs[5] = mxCreateDoubleMatrix(n, n, mxREAL); o.P_mean = mxGetPr(plhs[5]); }
    if (nlhs > 6) { plhs[6] = mxCreateDoubleMatrix(nN, T, mxREAL); o.traj_sample_iwmax = mxGetPr(plhs[6]); }
    if (nlhs > 7) { plhs[7] = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL); o.xn_traj = mxGetPr(plhs[7]); }
    const int st = rbpf_particle_filter(&m, &p, &r, &opt, &o);
    if (st != RBPF_OK) raise(g, st);
  } else if (cmd == "smoother") {
    if (nrhs != 14 && nrhs != 15) mexErrMsgIdAndTxt("rbpf:usage", "smoother expects 13 or 14 arguments after the command");
    rbpf_model m = model_from(prhs[1], nn, g, cb);
    rbpf_problem p = problem_from(prhs[2], prhs[3], prhs[4], prhs[5], prhs[6], prhs[7], prhs[8], prhs[9], prhs[11]);
    const int N_K = (int)mxGetScalar(prhs[10]);
    rbpf_rng r = rng_from(prhs[12], N_K);
    std::vector<double> zeroZ;
    if (m.kind == RBPF_MODEL_GENERIC_DENSE && r.mode == RBPF_RNG_REPLAY && !r.Z) {
      zeroZ.assign((size_t)p.n_w * p.N_P * (p.N_T > 1 ? p.N_T - 1 : 0) * N_K + 1, 0.0);
      r.Z = zeroZ.data();
    }
    const int info_form = (int)mxGetScalar(prhs[13]);
    g.prob = &p;
    const mwSize nN = p.n_nonlin, n = p.n_lin, T = p.N_T;
    rbpf_smoother_out o;
    std::memset(&o, 0, sizeof(o));
    const mwSize d1[3] = {nN, T, (mwSize)N_K}, d3[3] = {n, n, (mwSize)N_K};
    const bool plots = nrhs == 15 && is_handle(prhs[14]);
    plhs[0] = mxCreateNumericArray(3, d1, mxDOUBLE_CLASS, mxREAL); o.XNK = mxGetPr(plhs[0]); fill_nan(plhs[0]);   // XNK = nan(...) :81
    mxArray *xlk = nullptr, *pk = nullptr;
    if (nlhs > 1 || plots) { xlk = mxCreateDoubleMatrix(n, N_K, mxREAL); o.XLK = mxGetPr(xlk); fill_nan(xlk); }
    if (nlhs > 2 || plots) { pk = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL); o.PK = mxGetPr(pk); fill_nan(pk); }
    if (nlhs > 1) plhs[1] = xlk;
    if (nlhs > 2) plhs[2] = pk;
    if (plots) {
      g.h_plots = const_cast<mxArray*>(prhs[14]); g.XNK = plhs[0]; g.XLK = xlk; g.PK = pk; g.N_K = N_K;
      opt.on_step = cb_on_step; opt.on_step_user = &g;
    }
    const int st = rbpf_particle_smoother(&m, &p, &r, &opt, N_K, info_form, &o);
    if (nlhs <= 1 && xlk) mxDestroyArray(xlk);
    if (nlhs <= 2 && pk) mxDestroyArray(pk);
    if (st != RBPF_OK) raise(g, st);
  } else if (cmd == "dynModel" || cmd == "measModel" || cmd == "dynResNorm") {
    // the closures of a recognised family evaluated on the device (what the handles of rbpf_model.m call)
    if (nrhs < 3) mexErrMsgIdAndTxt("rbpf:usage", "%s: too few arguments", cmdbuf);
    rbpf_model m = model_from(prhs[1], nn, g, cb);
    if (m.kind != RBPF_MODEL_DENSE_MAG_6D && m.kind != RBPF_MODEL_DENSE_RADIO_2DH) mexErrMsgIdAndTxt("rbpf:unsupported", "%s: dense families only", cmdbuf);
    const int nN = m.kind == RBPF_MODEL_DENSE_MAG_6D ? 7 : 3, nw = m.kind == RBPF_MODEL_DENSE_MAG_6D ? 6 : 1, nodo = nN;
    const int ny = m.kind == RBPF_MODEL_DENSE_MAG_6D ? 3 : 1, n = m.kind == RBPF_MODEL_DENSE_MAG_6D ? m.m_basis + 3 : m.m_basis;
    int st = RBPF_OK;
    if (cmd == "measModel") {
      const mwSize np = mxGetNumberOfElements(prhs[2]) / nN;
      std::vector<double> dy((size_t)ny * n * np);                            // [ny x nLin x Npred]
      st = rbpf_meas_model(&m, nN, (int32_t)np, mxGetPr(prhs[2]), dy.data());
      if (st != RBPF_OK) fail(st);
      const mwSize d3[3] = {np, (mwSize)ny, (mwSize)n};
      plhs[0] = ny == 1 ? mxCreateDoubleMatrix(np, n, mxREAL) : mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
      double* out = mxGetPr(plhs[0]);                                         // dy(i,k,c), run_dense3D_magfield.m:274-278
      for (mwSize i = 0; i < np; ++i) for (int c = 0; c < n; ++c) for (int k = 0; k < ny; ++k)
        out[i + np * (k + (size_t)ny * c)] = dy[k + (size_t)ny * (c + (size_t)n * i)];
    } else if (cmd == "dynModel") {
      if (nrhs != 7) mexErrMsgIdAndTxt("rbpf:usage", "dynModel: desc, xn, dx, dt, Q, z");
      const mwSize np = mxGetNumberOfElements(prhs[2]) / nN;
      if (mxGetNumberOfElements(prhs[6]) != (size_t)nw * np) mexErrMsgIdAndTxt("rbpf:usage", "dynModel: z must be [nw x Np]");
      plhs[0] = mxCreateDoubleMatrix(nN, np, mxREAL);
      st = rbpf_dyn_model(&m, nN, nw, nodo, (int32_t)np, mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetScalar(prhs[4]), mxGetPr(prhs[5]),
                          mxGetPr(prhs[6]), mxGetPr(plhs[0]));
      if (st != RBPF_OK) fail(st);
    } else {
      if (nrhs != 7) mexErrMsgIdAndTxt("rbpf:usage", "dynResNorm: desc, xnk, xni, dx, dt, Q");
      const mwSize np = mxGetNumberOfElements(prhs[3]) / nN;
      std::vector<double> e((size_t)nw * np);
      st = rbpf_dyn_res_norm(&m, nN, nw, nodo, (int32_t)np, mxGetPr(prhs[2]), mxGetPr(prhs[3]), mxGetPr(prhs[4]), mxGetScalar(prhs[5]),
                             mxGetPr(prhs[6]), e.data());
      if (st != RBPF_OK) fail(st);
      plhs[0] = mxCreateDoubleMatrix(np, nw, mxREAL);                         // a 1 x nw row per particle
      for (mwSize i = 0; i < np; ++i) for (int q = 0; q < nw; ++q) mxGetPr(plhs[0])[i + np * q] = e[q + (size_t)nw * i];
    }
  } else if (cmd == "version") {
    plhs[0] = mxCreateDoubleScalar((double)rbpf_abi_version());
  } else {
    mexErrMsgIdAndTxt("rbpf:usage", "unknown command");
  }
}
