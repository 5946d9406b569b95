// Device-resident context shared by the filter and smoother host code.
#pragma once
#include "../../include/rbpf.h"
#include "rbpf_internal.hpp"

#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace rbpf {

void set_error(const std::string& s);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define HIPCHK(expr)                                                          \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) return ::rbpf::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

struct SmootherState;   // rbpf_smoother.hip
struct ShardState;      // rbpf_shard.hip

// extra capacity requested by the sharded filter
struct CreateExtras {
  size_t bank_extra = 0;   // particles appended to every bank (recv region of remote ancestors)
  size_t rng_slots = 0;    // slots per step in the replay buffers (global particle count); 0: N_P
};

// information-form per-step buffers handed to the step kernel
struct InfoStep {
  const double* ivec_old; size_t ivec_old_stride; double* ivec_new;
  const double* hld_old; size_t hld_old_stride; double* hld_new;
  double* qf_new; double* Hb_new;
};

}  // namespace rbpf

struct rbpf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  rbpf::ModelDev mdl;
  rbpf::Layout lay;
  rbpf::Layout lay_low;     // same HBM layout, 2 x 2 wave decomposition (lazy variants with >= 3 pending sets)
  rbpf_options opt;
  int N = 0, T = 0;
  bool smoother = false;
  int N_K = 1;
  int rng_mode = 0;
  unsigned long long seed = 0;
  int chol_pages = 1;
  int x0_lin_cols = 1;
  // constants
  int* d_NN = nullptr;
  double *d_y = nullptr, *d_odo = nullptr, *d_cholQ = nullptr, *d_cholQfull = nullptr;
  double *d_x0l = nullptr, *d_P0t = nullptr, *d_P0b = nullptr;
  double* d_R = nullptr;    // sparse-visual family: R [d x d]
  double *d_U = nullptr, *d_Z = nullptr;
  std::vector<double> h_x0n, h_P0, h_x0l, h_R, h_y, h_Ufin;
  // particle banks (ping-pong)
  double* Pt[2] = {nullptr, nullptr};
  double* Pb[2] = {nullptr, nullptr};
  double* F[2] = {nullptr, nullptr};
  double* xl[2] = {nullptr, nullptr};
  int cur = 0;              // bank holding the stored covariances
  int xcur = 0;             // bank holding the means (== cur unless the multi-step lazy update is on)
  // multi-step lazy update (filter fast path)
  int lazy_depth = 1;
  double* Fb[rbpf::kMaxSets + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // factor-set banks
  int* fidx[2] = {nullptr, nullptr};   // [lazy_depth+1][N] entry tables, ping-pong
  int* base[2] = {nullptr, nullptr};   // [N] stored-matrix slot of every lineage, ping-pong
  int tcur = 0;
  // generic model family: host-evaluated states / Jacobians of the step about to run (consumed by ctx_step)
  double *d_xn_ext = nullptr, *d_H_ext = nullptr;
  const double *ext_xn = nullptr, *ext_H = nullptr;
  rbpf_callbacks cb = {nullptr, nullptr, nullptr, nullptr};   // the model's handles (copied at create; all null: caller-driven)
  bool has_cb = false;
  std::vector<double> h_xn;      // generic family: states of the last step [nN x N] column-major (host evaluates dynModel)
  std::vector<double> h_xn_new, h_dy;   // scratch of the callbacks
  std::vector<int> h_ai;
  std::vector<double> h_odo, h_Q, h_dt;  // host copies for the callbacks' per-step arguments
  int q_pages = 1, dt_len = 1;
  bool cholQfull_ok = true;      // chol(dt*Q,'lower') of the full matrix exists (default additive dynResNorm needs it)
  int drawn_step = -1;           // step whose ordinary ancestors were already drawn by generic_draw_propagate
  bool fp32 = false;        // the covariance banks hold float (rbpf_options.storage = 1)
  bool inplace = false;     // single covariance bank, rewritten in place at every flush (rbpf_options.inplace)
  int* d_ip = nullptr;      // [8][N] in-place flush plan: destination entry, phase, scratch
  bool share_inplace = false;   // single bank + block-lower storage at eight / sixteen tile rows: the children of one parent store ONE flushed matrix (launch_share_inplace_plan)
  // shared flush (filter, ping-pong banks, symmetric storage at eight tile rows): one child per parent stores the flushed matrix
  bool share_flush = false;
  int* d_share = nullptr;                       // [3][N]: smallest child per parent, destination entry, phase (1 = writer)
  double* d_strip_ws = nullptr; size_t strip_ws_stride = 0;   // symmetric storage at sixteen tile rows: the step kernel's column strips ([N][stride])
  unsigned long long* d_share_writers = nullptr;   // writers of the timed shared flushes (device counter)
  long long share_flush_particles = 0;          // particles of the timed shared flushes (N per flush step)
  // timed launches: reads of stored matrices counted per particle (nominal) and per DISTINCT matrix (device counter)
  int* d_distinct_mark = nullptr; size_t distinct_keys = 0; unsigned long long* d_distinct_counter = nullptr;
  long long distinct_nominal = 0; int distinct_epoch = 0;
  // history
  int hist_slabs = 2;
  double* X = nullptr;      // [slabs][nN][N]
  int* A = nullptr;         // [T or 1][N]
  double *logw = nullptr, *w = nullptr, *wc = nullptr;
  double *traj_max = nullptr, *traj_mean = nullptr;
  double* d_scal = nullptr;
  int* d_flags = nullptr;   // [0] status bits, [1] clamped draws, [2] iw_max, [3..] scratch
  int t = 0;
  int overflow_draws = 0;
  bool timing_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  double sched_bytes = 0.0;  // bytes the timed launches had to move (rbpf_timing.scheduled_bytes_per_launch)
  rbpf::SmootherState* sm = nullptr;
  rbpf::ShardState* sh = nullptr;
  double* d_rs = nullptr;    // scratch of the multi-workgroup resample pipeline
  double* d_unext = nullptr; // [N] Philox resampling uniforms of the next step (written by propagate_kernel)
  int* d_pre_i = nullptr;   // [N][kPreInts]  per-workgroup descriptors of the step kernel
  double* d_pre_d = nullptr; // [N][kPreDoubles]
  int* d_order = nullptr;   // [N] processing order of the next step (ancestor-sorted)
  int* d_counts = nullptr;  // [2N] counting-sort scratch
  int ready_step = -1;      // step whose ancestors (and order) were already drawn by the fused resample kernel
  bool fuse_resample = false;
  bool sort_steps = false;  // smoothers: process every step in ancestor order too (the filter's fused resample kernel does it there)
  int order_step = -1;      // step d_order was computed for by ctx_step (smoothers)
  size_t bank_cap = 0;      // particles per bank incl. the recv region
  size_t rng_slots = 0;     // slots per step in d_U / d_Z
};

namespace rbpf {

int fill_model_dev(const rbpf_model* model, int nN, int n, int d, int nw, int nodo, const double* R, double jitter,
                   ModelDev& M, std::vector<int>& nn_axis_major);
int ctx_create(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
               bool smoother, int N_K, rbpf_ctx** out, const CreateExtras* ex = nullptr);
int ctx_reset(rbpf_ctx* c);
void ctx_free(rbpf_ctx* c);
int ctx_step(rbpf_ctx* c, int k_iter, const double* xref_t, int n_draw, const InfoStep* info);
int ctx_check_flags(rbpf_ctx* c);
// generic family with callbacks (rbpf_model.callbacks): ordinary ancestors of the step about to run + dynModel on the
// host for slots [0, n_draw); then (generic_finish_inputs) the reference slot, measModel and the upload
int generic_draw_propagate(rbpf_ctx* c, int k_iter, int n_draw);
int generic_finish_inputs(rbpf_ctx* c, const double* xref_host);
int ctx_call_on_step(rbpf_ctx* c, int t, bool is_smoother);
void ctx_account_launch(rbpf_ctx* c, const StepArgs& a);
int ctx_arm_distinct(rbpf_ctx* c, StepArgs& a, size_t keys);
int ctx_unpack(rbpf_ctx* c, const int* d_index, int count, double* d_out);
void smoother_free(rbpf_ctx* c);
// in-library multi-device driver (rbpf_multi.hip): rbpf_options.n_devices
// rbpf_options.struct_size (ABI 9): a caller compiled against another layout is refused before any field is trusted
inline int options_ok(const rbpf_options* o) {
  if (!o || o->struct_size == 0 || o->struct_size == (int32_t)sizeof(rbpf_options)) return RBPF_OK;
  set_error("rbpf_options.struct_size = " + std::to_string(o->struct_size) + ", this library's rbpf_options has " + std::to_string(sizeof(rbpf_options)) +
            " bytes (ABI " + std::to_string(RBPF_ABI_VERSION) + "): rebuild the binding against include/rbpf.h");
  return RBPF_ERR_INVALID_ARG;
}
// the K rbpf_options.chol_refresh stands for (rbpf_chol_refresh_resolve; rbpf_smoother.hip)
int resolve_chol_refresh(int model_kind, int n, int d, int requested);
// an explicit kernel choice (chol_variant) names the kernel of the from-scratch factorisation: with chol_refresh left at 0
// (automatic) it selects that factorisation at every step
inline int effective_chol_refresh(const rbpf_options& o) { return (o.chol_refresh == 0 && o.chol_variant != 0) ? 1 : o.chol_refresh; }
inline bool wants_multi(const rbpf_options* o) { return o && (o->n_devices > 1 || (o->n_devices == 1 && o->device_ids)); }
int multi_particle_filter(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
                          rbpf_filter_out* out);
int multi_particle_smoother(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
                            int32_t N_K, int32_t info_form, rbpf_smoother_out* out);
void shard_free(rbpf_ctx* c);

}  // namespace rbpf
