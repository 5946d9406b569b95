// MEX gateway: binds the C ABI of include/rbpf.h for MATLAB (build: see INTEGRATION.md).
// NOT compiled in this repository's CI: neither MATLAB nor mex.h exists in the build image or on the
// GPU box.  It only marshals mxArrays into the rbpf_* structs; all arithmetic is in librbpf_hip.so.
//
//   [traj_max,traj_mean,xl_max,xl_mean,P_max,P_mean,traj_sample_iwmax,xn_traj] = ...
//       rbpf_mex('filter', desc, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, dt, rng)
//   [XNK,XLK,PK] = rbpf_mex('smoother', desc, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, N_K, dt, rng, info_form)
//
// desc : struct with fields kind (1 dense-mag-6D, 2 dense-radio-2D+heading), NN (m x dim, int32), L (1 x dim),
//        use_dyn_res_norm (logical)
// rng  : struct with fields mode ('replay'|'philox'), U, Z, Ufin (replay) or seed (philox)
#include "mex.h"
#include "../include/rbpf.h"

#include <cstring>
#include <string>
#include <vector>

static void fail(int status) {
  mexErrMsgIdAndTxt("rbpf:status", "%s (%s)", rbpf_last_error(), rbpf_status_string(status));
}

static rbpf_model model_from(const mxArray* d, std::vector<int32_t>& nn) {
  rbpf_model m;
  std::memset(&m, 0, sizeof(m));
  m.kind = (int32_t)mxGetScalar(mxGetField(d, 0, "kind"));
  if (m.kind == RBPF_MODEL_SPARSE_VISUAL_2D) {                 // examples/slam-sparse-visual: camera f, fp, fw
    m.m_basis = (int32_t)mxGetScalar(mxGetField(d, 0, "nLand"));
    m.dim = 2;
    const double* cam = mxGetPr(mxGetField(d, 0, "cam"));
    for (int a = 0; a < 3; ++a) m.cam[a] = cam[a];
    m.use_dyn_res_norm = 0;                                    // psslam.m passes dynResNorm = []
    return m;
  }
  const mxArray* NN = mxGetField(d, 0, "NN");
  m.m_basis = (int32_t)mxGetM(NN);
  m.dim = (int32_t)mxGetN(NN);
  nn.resize((size_t)m.m_basis * m.dim);
  if (mxIsInt32(NN)) std::memcpy(nn.data(), mxGetData(NN), nn.size() * sizeof(int32_t));
  else { const double* p = mxGetPr(NN); for (size_t q = 0; q < nn.size(); ++q) nn[q] = (int32_t)p[q]; }
  m.NN = nn.data();
  const double* L = mxGetPr(mxGetField(d, 0, "L"));
  for (int a = 0; a < m.dim; ++a) m.L[a] = L[a];
  const mxArray* u = mxGetField(d, 0, "use_dyn_res_norm");
  m.use_dyn_res_norm = u ? (int32_t)mxGetScalar(u) : 1;
  return m;
}

static rbpf_problem problem_from(const mxArray* odo, const mxArray* y, const mxArray* x0n, const mxArray* x0l,
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

static rbpf_rng rng_from(const mxArray* r, int n_iter) {
  rbpf_rng g;
  std::memset(&g, 0, sizeof(g));
  char mode[16] = {0};
  mxGetString(mxGetField(r, 0, "mode"), mode, sizeof(mode));
  g.n_iter = n_iter;
  if (std::string(mode) == "replay") {
    // U [N_P x (N_T-1) x n_iter], Z [n_w x N_P x (N_T-1) x n_iter], drawn with MATLAB's own rand / randn in
    // the reference's interleaved order by the wrapper (matlab/particleFilter.m)
    g.mode = RBPF_RNG_REPLAY;
    g.U = mxGetPr(mxGetField(r, 0, "U"));
    g.Z = mxGetPr(mxGetField(r, 0, "Z"));
    const mxArray* uf = mxGetField(r, 0, "Ufin");
    g.Ufin = uf ? mxGetPr(uf) : nullptr;
  } else {
    g.mode = RBPF_RNG_PHILOX;
    g.seed = (uint64_t)mxGetScalar(mxGetField(r, 0, "seed"));
  }
  return g;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("rbpf:usage", "first argument must be 'filter' or 'smoother'");
  char cmd[16] = {0};
  mxGetString(prhs[0], cmd, sizeof(cmd));
  std::vector<int32_t> nn;
  rbpf_options opt;
  std::memset(&opt, 0, sizeof(opt));
  opt.keep_history = 1;
  if (std::string(cmd) == "filter") {
    if (nrhs != 12) mexErrMsgIdAndTxt("rbpf:usage", "filter expects 11 arguments after the command");
    rbpf_model m = model_from(prhs[1], nn);
    rbpf_problem p = problem_from(prhs[2], prhs[3], prhs[4], prhs[5], prhs[6], prhs[7], prhs[8], prhs[9], prhs[10]);
    rbpf_rng g = rng_from(prhs[11], 1);
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
    const int st = rbpf_particle_filter(&m, &p, &g, &opt, &o);
    if (st != RBPF_OK) fail(st);
  } else if (std::string(cmd) == "smoother") {
    if (nrhs != 14) mexErrMsgIdAndTxt("rbpf:usage", "smoother expects 13 arguments after the command");
    rbpf_model m = model_from(prhs[1], nn);
    rbpf_problem p = problem_from(prhs[2], prhs[3], prhs[4], prhs[5], prhs[6], prhs[7], prhs[8], prhs[9], prhs[11]);
    const int N_K = (int)mxGetScalar(prhs[10]);
    rbpf_rng g = rng_from(prhs[12], N_K);
    const int info_form = (int)mxGetScalar(prhs[13]);
    const mwSize nN = p.n_nonlin, n = p.n_lin, T = p.N_T;
    rbpf_smoother_out o;
    std::memset(&o, 0, sizeof(o));
    const mwSize d1[3] = {nN, T, (mwSize)N_K}, d3[3] = {n, n, (mwSize)N_K};
    plhs[0] = mxCreateNumericArray(3, d1, mxDOUBLE_CLASS, mxREAL); o.XNK = mxGetPr(plhs[0]);
    if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(n, N_K, mxREAL); o.XLK = mxGetPr(plhs[1]); }
    if (nlhs > 2) { plhs[2] = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL); o.PK = mxGetPr(plhs[2]); }
    const int st = rbpf_particle_smoother(&m, &p, &g, &opt, N_K, info_form, &o);
    if (st != RBPF_OK) fail(st);
  } else {
    mexErrMsgIdAndTxt("rbpf:usage", "unknown command");
  }
}
