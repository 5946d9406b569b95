// Drives matlab/rbpf_mex.cpp's mexFunction through the mex.h test double the way MATLAB would (tests only).
//
//   gateway_driver <dir> [--no-device]
//
// <dir> holds the problem written by tests/test_gpu_mex_gateway.py (meta.txt + raw little-endian arrays); every scenario's
// outputs are written back as <dir>/<scenario>_<name>.f64 (column-major) and compared there with the Python ctypes path.
#include "mexdouble.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

using mexdouble::Callable;
using mexdouble::MatlabError;

namespace {

std::string g_dir;
struct Arr { std::vector<mwSize> dims; std::vector<double> v; };
std::map<std::string, Arr> g_in;

void load_inputs() {
  std::ifstream meta(g_dir + "/meta.txt");
  std::string line;
  while (std::getline(meta, line)) {
    std::istringstream ss(line);
    std::string name; int nd;
    if (!(ss >> name >> nd)) continue;
    Arr a;
    size_t n = 1;
    for (int k = 0; k < nd; ++k) { mwSize d; ss >> d; a.dims.push_back(d); n *= d; }
    a.v.resize(n);
    std::ifstream f(g_dir + "/" + name + ".f64", std::ios::binary);
    f.read(reinterpret_cast<char*>(a.v.data()), (std::streamsize)(n * sizeof(double)));
    if (!f) throw std::runtime_error("cannot read " + name);
    g_in[name] = a;
  }
}

mxArray* mx_of(const std::string& name) {
  const Arr& a = g_in.at(name);
  mxArray* m = mxCreateNumericArray(a.dims.size(), a.dims.data(), mxDOUBLE_CLASS, mxREAL);
  std::memcpy(mxGetPr(m), a.v.data(), a.v.size() * sizeof(double));
  return m;
}

void save(const std::string& scenario, const std::string& name, const mxArray* a) {
  std::ofstream f(g_dir + "/" + scenario + "_" + name + ".f64", std::ios::binary);
  f.write(reinterpret_cast<const char*>(mxGetPr(a)), (std::streamsize)(mxGetNumberOfElements(a) * sizeof(double)));
  std::ofstream d(g_dir + "/" + scenario + "_" + name + ".dims");
  for (mwSize k = 0; k < mxGetNumberOfDimensions(a); ++k) d << mxGetDimensions(a)[k] << " ";
}

mxArray* family_desc() {                                   // what rbpf_recognise / rbpf_model build for a dense family
  const char* names[] = {"kind", "NN", "L", "use_dyn_res_norm"};
  mxArray* d = mxCreateStructMatrix(1, 1, 4, names);
  mxSetField(d, 0, "kind", mxCreateDoubleScalar(g_in.at("kind").v[0]));
  const Arr& nn = g_in.at("NN");
  mxArray* NN = mxCreateNumericMatrix(nn.dims[0], nn.dims[1], mxINT32_CLASS, mxREAL);
  for (size_t q = 0; q < nn.v.size(); ++q) static_cast<int32_t*>(mxGetData(NN))[q] = (int32_t)nn.v[q];
  mxSetField(d, 0, "NN", NN);
  mxSetField(d, 0, "L", mx_of("L"));
  mxSetField(d, 0, "use_dyn_res_norm", mxCreateLogicalScalar(true));
  return d;
}

mxArray* rng_block(bool with_z, bool with_ufin) {
  const char* names[] = {"mode", "U"};
  mxArray* r = mxCreateStructMatrix(1, 1, 2, names);
  mxSetField(r, 0, "mode", mxCreateString("replay"));
  mxSetField(r, 0, "U", mx_of(with_ufin ? "U_s" : "U_f"));
  if (with_z) mxSetField(r, 0, "Z", mx_of(with_ufin ? "Z_s" : "Z_f"));
  if (with_ufin) mxSetField(r, 0, "Ufin", mx_of("Ufin"));
  return r;
}

// gateway helper commands as "MATLAB calls" (what the handles of rbpf_model.m do)
mxArray* gateway1(std::vector<mxArray*> args) {
  mxArray* out = nullptr;
  std::vector<const mxArray*> c(args.begin(), args.end());
  mexFunction(1, &out, (int)c.size(), c.data());
  return out;
}

struct Counters { int dyn = 0, meas = 0, drn = 0, plots = 0; bool plot_shapes_ok = true; };

// handles "unknown to the library": they evaluate the radio / mag closures through the gateway's helper commands, with the
// normals of the replay buffer injected in the reference's call order (k, t, slot)
struct Handles {
  mxArray *dyn, *meas, *drn;
  Counters* c;
};

Handles make_handles(Counters* cnt, bool smoother, int N, int T, int nw, int N_K) {
  Handles h;
  h.c = cnt;
  const std::string zname = smoother ? "Z_s" : "Z_f";
  h.dyn = mexdouble::make_function_handle([=](int, mxArray** plhs, int nrhs, mxArray** prhs) {
    if (nrhs != 4) throw MatlabError("test:dyn", "dynModel expects (xn, dx, dt, Q)");
    // call order (particleSmoother.m:132-137,149-152): iteration k, step t, slots 0..N-1 (k = 0) or 0..N-2 (k > 0)
    int idx = cnt->dyn++, k = 0, t = 0, i = 0;
    if (!smoother) { t = idx / N; i = idx % N; }
    else {
      const int first = (T - 1) * N;
      if (idx < first) { t = idx / N; i = idx % N; }
      else { idx -= first; k = 1 + idx / ((T - 1) * (N - 1)); idx %= (T - 1) * (N - 1); t = idx / (N - 1); i = idx % (N - 1); }
    }
    if (k >= N_K || t >= T - 1) throw MatlabError("test:dyn", "dynModel called too often");
    const Arr& Z = g_in.at(zname);                                    // [nw x N x (T-1) (x N_K)]
    mxArray* z = mxCreateDoubleMatrix(nw, 1, mxREAL);
    for (int q = 0; q < nw; ++q) mxGetPr(z)[q] = Z.v[q + (size_t)nw * (i + (size_t)N * (t + (size_t)(T - 1) * k))];
    mxArray* desc = family_desc();
    mxArray* cmd = mxCreateString("dynModel");
    plhs[0] = gateway1({cmd, desc, prhs[0], prhs[1], prhs[2], prhs[3], z});
    mxDestroyArray(z); mxDestroyArray(desc); mxDestroyArray(cmd);
  }, "@(xn,dx,dt,Q)...");
  h.meas = mexdouble::make_function_handle([=](int, mxArray** plhs, int nrhs, mxArray** prhs) {
    if (nrhs != 1) throw MatlabError("test:meas", "measModel expects (xn)");
    ++cnt->meas;
    mxArray* desc = family_desc();
    mxArray* cmd = mxCreateString("measModel");
    plhs[0] = gateway1({cmd, desc, prhs[0]});
    mxDestroyArray(desc); mxDestroyArray(cmd);
  }, "@(xn)...");
  h.drn = mexdouble::make_function_handle([=](int, mxArray** plhs, int nrhs, mxArray** prhs) {
    if (nrhs != 5) throw MatlabError("test:drn", "dynResNorm expects (xnk, xni, dx, dt, Q)");
    ++cnt->drn;
    mxArray* desc = family_desc();
    mxArray* cmd = mxCreateString("dynResNorm");
    plhs[0] = gateway1({cmd, desc, prhs[0], prhs[1], prhs[2], prhs[3], prhs[4]});
    mxDestroyArray(desc); mxDestroyArray(cmd);
  }, "@(xnk,xni,dx,dt,Q)...");
  return h;
}

// the build's helper .m files (matlab/rbpf_batch_dyn.m, rbpf_batch_drn.m), restated for the double's "MATLAB path"
void register_batch_helpers() {
  mexdouble::register_function("rbpf_batch_dyn", [](int, mxArray** plhs, int nrhs, mxArray** prhs) {
    if (nrhs != 5) throw MatlabError("test:batch", "rbpf_batch_dyn(dynModel, xn, dx, dt, Q)");
    const mwSize nN = mxGetM(prhs[1]), n = mxGetN(prhs[1]);
    plhs[0] = mxCreateDoubleMatrix(nN, n, mxREAL);
    for (mwSize j = 0; j < n; ++j) {
      mxArray* col = mxCreateDoubleMatrix(nN, 1, mxREAL);
      std::memcpy(mxGetPr(col), mxGetPr(prhs[1]) + nN * j, sizeof(double) * nN);
      mxArray* args[5] = {prhs[0], col, prhs[2], prhs[3], prhs[4]};
      mxArray* out = nullptr;
      mexCallMATLAB(1, &out, 5, args, "feval");
      std::memcpy(mxGetPr(plhs[0]) + nN * j, mxGetPr(out), sizeof(double) * nN);
      mxDestroyArray(out); mxDestroyArray(col);
    }
  });
  mexdouble::register_function("rbpf_batch_drn", [](int, mxArray** plhs, int nrhs, mxArray** prhs) {
    if (nrhs != 6) throw MatlabError("test:batch", "rbpf_batch_drn(dynResNorm, xnkt, xn, dx, dt, Q)");
    const mwSize nN = mxGetM(prhs[2]), n = mxGetN(prhs[2]), nw = mxGetM(prhs[5]);
    plhs[0] = mxCreateDoubleMatrix(nw, n, mxREAL);
    for (mwSize i = 0; i < n; ++i) {
      mxArray* col = mxCreateDoubleMatrix(nN, 1, mxREAL);
      std::memcpy(mxGetPr(col), mxGetPr(prhs[2]) + nN * i, sizeof(double) * nN);
      mxArray* args[6] = {prhs[0], prhs[1], col, prhs[3], prhs[4], prhs[5]};
      mxArray* out = nullptr;
      mexCallMATLAB(1, &out, 6, args, "feval");
      for (mwSize q = 0; q < nw; ++q) mxGetPr(plhs[0])[q + nw * i] = mxGetPr(out)[q];
      mxDestroyArray(out); mxDestroyArray(col);
    }
  });
}

struct Problem { mxArray *odo, *y, *x0n, *x0l, *P0, *Q, *R, *NP, *dt; };
Problem problem() { return {mx_of("odometry"), mx_of("y"), mx_of("x0_nonLin"), mx_of("x0_lin"), mx_of("P0_lin"), mx_of("Q"), mx_of("R"), mx_of("N_P"), mx_of("dt")}; }
void free_problem(Problem& p) { for (mxArray* a : {p.odo, p.y, p.x0n, p.x0l, p.P0, p.Q, p.R, p.NP, p.dt}) mxDestroyArray(a); }

void run_filter(const std::string& scenario, mxArray* desc, mxArray* rng, mxArray* plots) {
  Problem p = problem();
  mxArray* cmd = mxCreateString("filter");
  std::vector<const mxArray*> in = {cmd, desc, p.odo, p.y, p.x0n, p.x0l, p.P0, p.Q, p.R, p.NP, p.dt, rng};
  if (plots) in.push_back(plots);
  mxArray* out[8] = {nullptr};
  mexFunction(8, out, (int)in.size(), in.data());
  const char* names[8] = {"traj_max", "traj_mean", "xl_max", "xl_mean", "P_max", "P_mean", "traj_sample_iwmax", "xn_traj"};
  for (int k = 0; k < 8; ++k) { save(scenario, names[k], out[k]); mxDestroyArray(out[k]); }
  mxDestroyArray(cmd); free_problem(p);
}

void run_smoother(const std::string& scenario, mxArray* desc, mxArray* rng, int info_form, mxArray* plots) {
  Problem p = problem();
  mxArray* cmd = mxCreateString("smoother");
  mxArray* nk = mx_of("N_K");
  mxArray* inf = mxCreateDoubleScalar(info_form);
  std::vector<const mxArray*> in = {cmd, desc, p.odo, p.y, p.x0n, p.x0l, p.P0, p.Q, p.R, p.NP, nk, p.dt, rng, inf};
  if (plots) in.push_back(plots);
  mxArray* out[3] = {nullptr};
  mexFunction(3, out, (int)in.size(), in.data());
  const char* names[3] = {"XNK", "XLK", "PK"};
  for (int k = 0; k < 3; ++k) { save(scenario, names[k], out[k]); mxDestroyArray(out[k]); }
  mxDestroyArray(cmd); mxDestroyArray(nk); mxDestroyArray(inf); free_problem(p);
}

mxArray* generic_desc(const Handles& h, bool with_drn) {
  const char* names[] = {"kind", "dynModel", "measModel", "dynResNorm"};
  mxArray* d = mxCreateStructMatrix(1, 1, 4, names);
  mxSetField(d, 0, "kind", mxCreateDoubleScalar(4));
  mxSetField(d, 0, "dynModel", mxDuplicateArray(h.dyn));
  mxSetField(d, 0, "measModel", mxDuplicateArray(h.meas));
  mxSetField(d, 0, "dynResNorm", with_drn ? mxDuplicateArray(h.drn) : mxCreateDoubleMatrix(0, 0, mxREAL));
  return d;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: gateway_driver <dir> [--no-device]\n"); return 2; }
  g_dir = argv[1];
  const bool no_device = argc > 2 && !std::strcmp(argv[2], "--no-device");
  std::ofstream report(g_dir + "/report.txt");
  try {
    load_inputs();
    register_batch_helpers();
    const int N = (int)g_in.at("N_P").v[0], T = (int)g_in.at("y").dims[0], nw = (int)g_in.at("Q").dims[0], N_K = (int)g_in.at("N_K").v[0];
    const size_t base = mexdouble::live_arrays();
    {   // the 'version' command works without a device
      mxArray* cmd = mxCreateString("version");
      mxArray* v = gateway1({cmd});
      report << "version " << mxGetScalar(v) << "\n";
      mxDestroyArray(v); mxDestroyArray(cmd);
    }
    if (no_device) {
      // without a GPU every compute entry point must fail loudly as a MATLAB error, never fall back to the CPU
      mxArray* desc = family_desc(); mxArray* rng = rng_block(true, false);
      try { run_filter("nodev", desc, rng, nullptr); report << "nodevice_error MISSING\n"; }
      catch (const MatlabError& e) { report << "nodevice_error " << e.id << " | " << e.what() << "\n"; }
      mxDestroyArray(desc); mxDestroyArray(rng);
      return 0;
    }
    Counters cf;
    {   // 1. recognised family through the gateway, makePlots hook after every step
      mxArray* plots = mexdouble::make_function_handle([&](int, mxArray**, int nrhs, mxArray** prhs) {
        ++cf.plots;
        const mwSize nN = g_in.at("x0_nonLin").v.size(), n = g_in.at("x0_lin").dims[0];
        cf.plot_shapes_ok = cf.plot_shapes_ok && nrhs == 9 && mxGetM(prhs[0]) == nN && mxGetN(prhs[0]) == (mwSize)N && mxGetM(prhs[1]) == n &&
                            mxGetM(prhs[2]) == n && mxGetN(prhs[2]) == n && mxGetN(prhs[3]) == (mwSize)T && mxGetNumberOfDimensions(prhs[5]) == 3 &&
                            mxGetDimensions(prhs[5])[2] == (mwSize)T && mxGetNumberOfDimensions(prhs[8]) == 3 && mxGetDimensions(prhs[8])[2] == (mwSize)N &&
                            std::isnan(mxGetPr(prhs[4])[0]);
        if (cf.plots == T) save("filter_family", "plot_last_P", prhs[8]);
      }, "@makePlots");
      mxArray* desc = family_desc(); mxArray* rng = rng_block(true, false);
      run_filter("filter_family", desc, rng, plots);
      mxDestroyArray(desc); mxDestroyArray(rng); mxDestroyArray(plots);
      report << "filter_family_plots " << cf.plots << " shapes_ok " << cf.plot_shapes_ok << "\n";
    }
    {   // 2. the same closures as handles unknown to the library: generic family, callbacks through mexCallMATLAB
      Counters c;
      Handles h = make_handles(&c, false, N, T, nw, 1);
      mxArray* desc = generic_desc(h, false); mxArray* rng = rng_block(false, false);
      run_filter("filter_generic", desc, rng, nullptr);
      report << "filter_generic_calls dyn " << c.dyn << " meas " << c.meas << "\n";
      mxDestroyArray(desc); mxDestroyArray(rng); mxDestroyArray(h.dyn); mxDestroyArray(h.meas); mxDestroyArray(h.drn);
    }
    for (int info = 0; info < 2; ++info) {
      const std::string tag = info ? "info" : "cov";
      {   // 3. smoothers, recognised family, makePlots after every iteration
        int plots_n = 0; bool nan_ok = true;
        mxArray* plots = mexdouble::make_function_handle([&](int, mxArray**, int nrhs, mxArray** prhs) {
          ++plots_n;
          const int k = (int)mxGetScalar(prhs[2]);                               // 1-based iteration
          const mwSize page = mxGetM(prhs[3]) * (mxGetN(prhs[3]) / N_K);
          nan_ok = nan_ok && nrhs == 6 && k == plots_n && !std::isnan(mxGetPr(prhs[3])[(size_t)(k - 1) * page]) &&
                   (k == N_K || std::isnan(mxGetPr(prhs[3])[(size_t)k * page]));
        }, "@makePlotsSmoother");
        mxArray* desc = family_desc(); mxArray* rng = rng_block(true, true);
        run_smoother("smoother_family_" + tag, desc, rng, info, plots);
        mxDestroyArray(desc); mxDestroyArray(rng); mxDestroyArray(plots);
        report << "smoother_family_" << tag << "_plots " << plots_n << " nan_ok " << nan_ok << "\n";
      }
      {   // 4. smoothers with arbitrary handles incl. dynResNorm
        Counters c;
        Handles h = make_handles(&c, true, N, T, nw, N_K);
        mxArray* desc = generic_desc(h, true); mxArray* rng = rng_block(false, true);
        run_smoother("smoother_generic_" + tag, desc, rng, info, nullptr);
        report << "smoother_generic_" << tag << "_calls dyn " << c.dyn << " meas " << c.meas << " drn " << c.drn << "\n";
        mxDestroyArray(desc); mxDestroyArray(rng); mxDestroyArray(h.dyn); mxDestroyArray(h.meas); mxDestroyArray(h.drn);
      }
    }
    report << "leaked " << (mexdouble::live_arrays() - base) << "\n";      // scenarios 1-4: every temporary of the gateway was freed
    {   // 5. a MATLAB error inside a handle surfaces as a MATLAB error of the gateway, with its message
      Counters c;
      Handles h = make_handles(&c, false, N, T, nw, 1);
      mxDestroyArray(h.meas);
      h.meas = mexdouble::make_function_handle([](int, mxArray**, int, mxArray**) { throw MatlabError("user:boom", "Index exceeds matrix dimensions."); }, "@bad");
      mxArray* desc = generic_desc(h, false); mxArray* rng = rng_block(false, false);
      try { run_filter("bad", desc, rng, nullptr); report << "callback_error MISSING\n"; }
      catch (const MatlabError& e) { report << "callback_error " << e.id << " | " << e.what() << "\n"; }
      mxDestroyArray(desc); mxDestroyArray(rng); mxDestroyArray(h.dyn); mxDestroyArray(h.meas); mxDestroyArray(h.drn);
    }
    {   // 6. usage errors
      mxArray* cmd = mxCreateString("nonsense");
      try { gateway1({cmd}); report << "usage_error MISSING\n"; }
      catch (const MatlabError& e) { report << "usage_error " << e.id << "\n"; }
      mxDestroyArray(cmd);
    }
    {   // 7. session options (matlab/rbpf_options.m): set -> they apply to the next smoother call; an empty struct resets
      const char* names[] = {"chol_refresh"};
      mxArray* so = mxCreateStructMatrix(1, 1, 1, names);
      mxSetField(so, 0, "chol_refresh", mxCreateDoubleScalar(3));
      mxArray* cmd = mxCreateString("options");
      mxArray* cur = gateway1({cmd, so});
      report << "options_set " << mxGetScalar(mxGetField(cur, 0, "chol_refresh")) << " " << mxGetScalar(mxGetField(cur, 0, "lazy_depth")) << "\n";
      mxDestroyArray(cur);
      mxArray* desc = family_desc(); mxArray* rng = rng_block(true, true);
      run_smoother("smoother_options_info", desc, rng, 1, nullptr);
      mxDestroyArray(desc); mxDestroyArray(rng);
      mxArray* none = mxCreateStructMatrix(1, 1, 0, nullptr);
      cur = gateway1({cmd, none});
      report << "options_reset " << mxGetScalar(mxGetField(cur, 0, "chol_refresh")) << "\n";
      mxDestroyArray(cur);
      cur = gateway1({cmd});
      report << "options_query " << mxGetScalar(mxGetField(cur, 0, "chol_refresh")) << "\n";
      mxDestroyArray(cur); mxDestroyArray(cmd); mxDestroyArray(so); mxDestroyArray(none);
    }
    {   // 8. n_devices routes the next call to the in-library multi-device driver (rbpf_options.n_devices): on a one-GPU machine
        //    the second device does not exist, which surfaces as a MATLAB error; on a multi-GPU machine the smoother runs sharded
      const char* names[] = {"n_devices"};
      mxArray* so = mxCreateStructMatrix(1, 1, 1, names);
      mxSetField(so, 0, "n_devices", mxCreateDoubleScalar(2));
      mxArray* cmd = mxCreateString("options");
      mxArray* cur = gateway1({cmd, so});
      report << "options_n_devices " << mxGetScalar(mxGetField(cur, 0, "n_devices")) << "\n";
      mxDestroyArray(cur);
      mxArray* desc = family_desc(); mxArray* rng = rng_block(true, true);
      try { run_smoother("smoother_multi", desc, rng, 1, nullptr); report << "multi_route ran\n"; }
      catch (const MatlabError& e) { report << "multi_route " << e.id << "\n"; }
      mxDestroyArray(desc); mxDestroyArray(rng);
      mxArray* none = mxCreateStructMatrix(1, 1, 0, nullptr);
      cur = gateway1({cmd, none});
      mxDestroyArray(cur); mxDestroyArray(cmd); mxDestroyArray(so); mxDestroyArray(none);
    }
    {   // 9. the particleFilter route with n_devices = 2 on ONE GPU (device_ids = [0 0]: two ranks share it over the host-staged
        //    transport): matlab/particleFilter.m always asks for all 8 outputs, xn_traj included; a makePlots handle
        //    is refused with a message of its own
      const char* names[] = {"n_devices", "device_ids"};
      mxArray* so = mxCreateStructMatrix(1, 1, 2, names);
      mxSetField(so, 0, "n_devices", mxCreateDoubleScalar(2));
      mxArray* ids = mxCreateDoubleMatrix(1, 2, mxREAL);
      mxGetPr(ids)[0] = 0; mxGetPr(ids)[1] = 0;
      mxSetField(so, 0, "device_ids", ids);
      mxArray* cmd = mxCreateString("options");
      mxArray* cur = gateway1({cmd, so});
      report << "options_device_ids " << mxGetNumberOfElements(mxGetField(cur, 0, "device_ids")) << "\n";
      mxDestroyArray(cur);
      mxArray* desc = family_desc(); mxArray* rng = rng_block(true, true); mxArray* rngf = rng_block(true, false);
      try { run_filter("filter_multi", desc, rngf, nullptr); report << "multi_filter ran\n"; }
      catch (const MatlabError& e) { report << "multi_filter " << e.id << " " << e.what() << "\n"; }
      try { run_smoother("smoother_multi2", desc, rng, 1, nullptr); report << "multi_smoother ran\n"; }
      catch (const MatlabError& e) { report << "multi_smoother " << e.id << " " << e.what() << "\n"; }
      mxArray* plots = mexdouble::make_function_handle([&](int, mxArray**, int, mxArray**) {}, "@makePlots");
      try { run_filter("bad_multi", desc, rngf, plots); report << "multi_plots_error MISSING\n"; }
      catch (const MatlabError& e) { report << "multi_plots_error " << e.id << "\n"; }
      mxDestroyArray(plots); mxDestroyArray(desc); mxDestroyArray(rng); mxDestroyArray(rngf);
      mxArray* none = mxCreateStructMatrix(1, 1, 0, nullptr);
      cur = gateway1({cmd, none});
      mxDestroyArray(cur); mxDestroyArray(cmd); mxDestroyArray(so); mxDestroyArray(none);
    }
  } catch (const std::exception& e) {
    report << "DRIVER_FAILED " << e.what() << "\n";
    fprintf(stderr, "gateway_driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
