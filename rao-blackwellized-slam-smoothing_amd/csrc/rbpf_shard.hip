// Particle-sharded forward filter: kernels and buffers between the collectives.
// The collectives themselves (all_gather of the forward bank, all_to_all of remote ancestors) are
// issued by the host mirror (multigpu.py) through torch.distributed / RCCL on the pointers exposed
// here.  Every rank normalises the *global* weight vector and draws the *global* ancestor vector with
// the same kernels as the single-GPU path, so a W-rank run equals the single-GPU run with
// N = W * N_local bit for bit.
#include "../../include/rbpf.h"
#include "rbpf_internal.hpp"
#include "rbpf_ctx.hpp"

#include <cstring>
#include <vector>

namespace rbpf {

struct ShardState {
  int rank = 0, world = 1, Nloc = 0, Nglob = 0;
  size_t recv_cap = 0, send_cap = 0;
  double* logw_gather = nullptr;   // [world][Nloc]
  double* xn_gather = nullptr;     // [world][nN][Nloc]
  double* xn_glob = nullptr;       // SoA [nN][Nglob] of the step just gathered
  double* w_glob = nullptr;        // [Nglob]
  double* wc_glob = nullptr;       // [Nglob]
  int* ai_glob = nullptr;          // [Nglob]
  int* ai_bank = nullptr;          // [Nloc]
  int* pack_idx = nullptr;         // [send_cap]
  double *send_Pt = nullptr, *send_Pb = nullptr, *send_F = nullptr, *send_xl = nullptr;
  int t_norm = 0;                  // steps normalised so far
};

void shard_free(rbpf_ctx* c) {
  ShardState* s = c->sh;
  if (!s) return;
  hipFree(s->logw_gather); hipFree(s->xn_gather); hipFree(s->xn_glob); hipFree(s->w_glob); hipFree(s->wc_glob);
  hipFree(s->ai_glob); hipFree(s->ai_bank); hipFree(s->pack_idx);
  hipFree(s->send_Pt); hipFree(s->send_Pb); hipFree(s->send_F); hipFree(s->send_xl);
  delete s;
  c->sh = nullptr;
}

}  // namespace rbpf

using namespace rbpf;

#define RB_TRY(x) do { int _s = (x); if (_s != RBPF_OK) return _s; } while (0)

template <typename T>
static int dmalloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc((void**)p, count * sizeof(T));
  if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
  return RBPF_OK;
}

extern "C" {

int rbpf_shard_create(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
                      int32_t rank, int32_t world, rbpf_ctx** out) {
  if (!prob || !out || world < 1 || rank < 0 || rank >= world) { set_error("bad shard arguments"); return RBPF_ERR_INVALID_ARG; }
  if (prob->x0_lin_cols != 1) { set_error("sharded filter: x0_lin must be nLin x 1"); return RBPF_ERR_UNSUPPORTED; }
  const size_t Nloc = (size_t)prob->N_P;
  CreateExtras ex;
  ex.bank_extra = (world > 1) ? Nloc : 0;          // a slot has one ancestor: at most N_local remote ones
  ex.rng_slots = Nloc * world;
  rbpf_options o;
  if (opt) o = *opt; else std::memset(&o, 0, sizeof(o));
  o.keep_history = 0;
  o.trace = 0;
  rbpf_ctx* c = nullptr;
  RB_TRY(ctx_create(model, prob, rng, &o, false, 1, &c, &ex));
  ShardState* s = new ShardState();
  c->sh = s;
  s->rank = rank; s->world = world; s->Nloc = (int)Nloc; s->Nglob = (int)(Nloc * world);
  s->recv_cap = ex.bank_extra;
  // worst case one rank's particles are wanted by every other rank: (world-1)*N_local copies.
  // Start with 2*N_local and let the host side chunk if a step ever needs more (it reports the need).
  const Layout& L = c->lay;
  const int nN = c->mdl.nN, d = c->mdl.d;
  if (world > 1) {
    // worst case every other rank wants all of this rank's particles: (world-1)*N_local copies.
    // Take that when it fits in half of the free memory, otherwise 2*N_local (the host side reports
    // a step that needs more instead of corrupting memory).
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    const size_t per = (L.szT + L.szB + (size_t)2 * d * L.ldx + L.ldx) * sizeof(double);
    const size_t fit = per ? (free_b / 2) / per : 0;
    s->send_cap = std::min(Nloc * (size_t)(world - 1), std::max(Nloc * 2, fit));
  } else {
    s->send_cap = 0;
  }
  int st = RBPF_OK;
  auto A = [&](int r) { if (st == RBPF_OK) st = r; };
  A(dmalloc(&s->logw_gather, (size_t)s->Nglob));
  A(dmalloc(&s->xn_gather, (size_t)s->Nglob * nN));
  A(dmalloc(&s->xn_glob, (size_t)s->Nglob * nN));
  A(dmalloc(&s->w_glob, (size_t)s->Nglob));
  A(dmalloc(&s->wc_glob, (size_t)s->Nglob));
  A(dmalloc(&s->ai_glob, (size_t)s->Nglob));
  A(dmalloc(&s->ai_bank, Nloc));
  A(dmalloc(&s->pack_idx, s->send_cap));
  A(dmalloc(&s->send_Pt, s->send_cap * L.szT));
  A(dmalloc(&s->send_Pb, s->send_cap * L.szB));
  A(dmalloc(&s->send_F, s->send_cap * 2 * d * L.ldx));
  A(dmalloc(&s->send_xl, s->send_cap * L.ldx));
  if (st != RBPF_OK) { ctx_free(c); return st; }
  *out = c;
  return RBPF_OK;
}

int rbpf_shard_views_get(rbpf_ctx* c, rbpf_shard_views* v) {
  if (!c || !c->sh || !v) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  ShardState* s = c->sh;
  const Layout& L = c->lay;
  const int d = c->mdl.d;
  v->rank = s->rank; v->world = s->world; v->N_local = s->Nloc; v->N_global = s->Nglob;
  v->szT = L.szT; v->szB = L.szB; v->szF = (size_t)2 * d * L.ldx; v->szX = (size_t)L.ldx;
  v->recv_capacity = s->recv_cap; v->send_capacity = s->send_cap;
  v->logw_local = c->logw; v->xn_local = c->X;
  v->logw_gather = s->logw_gather; v->xn_gather = s->xn_gather;
  v->send_Pt = s->send_Pt; v->send_Pb = s->send_Pb; v->send_F = s->send_F; v->send_xl = s->send_xl;
  const int ob = c->cur;            // the bank the NEXT step reads from
  const size_t N = (size_t)s->Nloc;
  v->recv_Pt = c->Pt[ob] + N * L.szT; v->recv_Pb = c->Pb[ob] + N * L.szB;
  v->recv_F = c->F[ob] + N * v->szF; v->recv_xl = c->xl[ob] + N * v->szX;
  return RBPF_OK;
}

int rbpf_shard_normalise_search(rbpf_ctx* c, int32_t* ai_host) {
  if (!c || !c->sh) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  ShardState* s = c->sh;
  const int nN = c->mdl.nN, N = s->Nglob;
  const int t_done = s->t_norm;                // index of the step whose weights are being normalised
  if (t_done >= c->T || t_done >= c->t) { set_error("normalise without a finished step"); return RBPF_ERR_STATE; }
  HIPCHK(launch_unblock_soa(s->world, nN, s->Nloc, s->xn_gather, s->xn_glob, c->stream));
  NormArgs nm;
  nm.N = N; nm.nN = nN; nm.t = t_done; nm.logw = s->logw_gather; nm.w = s->w_glob; nm.wc = s->wc_glob; nm.xn = s->xn_glob;
  nm.traj_max = c->traj_max + (size_t)t_done * nN; nm.traj_mean = c->traj_mean + (size_t)t_done * nN;
  nm.iw_max = c->d_flags + 2; nm.lse_out = nullptr;
  nm.parallel_scan = 1;
  HIPCHK(launch_normalise_scan(nm, c->stream));
  s->t_norm = t_done + 1;
  if (ai_host) {
    const int t = c->t;                        // the step about to run
    SearchArgs sa;
    sa.N = N; sa.n_draw = N; sa.t = t; sa.wc = s->wc_glob; sa.rng_mode = c->rng_mode; sa.k_iter = 0; sa.slot0 = 0;
    sa.u_is_scalar = 0;
    sa.U = c->d_U ? c->d_U + (size_t)(t - 1) * N : nullptr;
    sa.seed = c->seed; sa.ai = s->ai_glob; sa.overflow = c->d_flags + 1;
    sa.approx = 1; sa.ambiguous = c->d_flags + 4; sa.w = s->w_glob; sa.wc_exact = s->wc_glob;
    HIPCHK(launch_search(sa, c->stream));
    HIPCHK(launch_resample_fixup(sa, c->stream));
    HIPCHK(hipMemcpyAsync(ai_host, s->ai_glob, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return RBPF_OK;
}

int rbpf_shard_pack(rbpf_ctx* c, const int32_t* idx_host, int32_t count) {
  if (!c || !c->sh || count < 0) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  ShardState* s = c->sh;
  if ((size_t)count > s->send_cap) { set_error("send staging too small for this step's exchange"); return RBPF_ERR_OUT_OF_MEMORY; }
  if (count == 0) return RBPF_OK;
  HIPCHK(hipMemcpyAsync(s->pack_idx, idx_host, (size_t)count * sizeof(int), hipMemcpyHostToDevice, c->stream));
  const int ob = c->cur;
  HIPCHK(launch_pack_bank(c->lay, c->mdl.d, s->pack_idx, count, c->Pt[ob], c->Pb[ob], c->F[ob], c->xl[ob], s->send_Pt,
                          s->send_Pb, s->send_F, s->send_xl, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));     // the collective runs on another stream / library
  return RBPF_OK;
}

int rbpf_shard_step(rbpf_ctx* c, const int32_t* anc_bank_host) {
  if (!c || !c->sh) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  ShardState* s = c->sh;
  const int t = c->t, N = s->Nloc, nN = c->mdl.nN, d = c->mdl.d, nw = c->mdl.nw;
  if (t >= c->T) { set_error("advance past N_T"); return RBPF_ERR_STATE; }
  if ((t > 0) != (anc_bank_host != nullptr)) { set_error("anc_bank must be NULL exactly at t = 0"); return RBPF_ERR_INVALID_ARG; }
  const Layout& L = c->lay;
  StepArgs a;
  std::memset(&a, 0, sizeof(a));
  a.mdl = c->mdl; a.lay = L; a.N = N; a.t = t; a.propagate = (t > 0);
  a.slot_offset = s->rank * N;
  a.xn_new = c->X; a.xn_new_stride = (size_t)N;
  const int ob = c->cur, nb = (t == 0) ? 0 : (c->cur ^ 1);
  if (t == 0) {
    a.ai = nullptr; a.ai_bank = nullptr;
    a.xn_old = c->X; a.xn_old_stride = (size_t)N;           // filled with x0 by ctx_reset
    a.xl_old = c->d_x0l; a.xl_old_stride = 0; a.F_old = nullptr;
    a.Pt_old = c->d_P0t; a.Pb_old = c->d_P0b; a.Pt_old_stride = 0; a.Pb_old_stride = 0;
  } else {
    HIPCHK(hipMemcpyAsync(s->ai_bank, anc_bank_host, (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->stream));
    a.ai = s->ai_glob + (size_t)s->rank * N;                // global ancestor ids of my slots
    a.ai_bank = s->ai_bank;
    HIPCHK(launch_order(N, 2 * N, s->ai_bank, c->d_order, c->d_counts, c->stream));
    a.order = c->d_order;
    a.xn_old = s->xn_glob; a.xn_old_stride = (size_t)s->Nglob;
    a.xl_old = c->xl[ob]; a.xl_old_stride = (size_t)L.ldx; a.F_old = c->F[ob];
    a.Pt_old = c->Pt[ob]; a.Pb_old = c->Pb[ob]; a.Pt_old_stride = L.szT; a.Pb_old_stride = L.szB;
  }
  a.xl_new = c->xl[nb]; a.F_new = c->F[nb]; a.Pt_new = c->Pt[nb]; a.Pb_new = c->Pb[nb];
  a.logw = c->logw;
  a.rng_mode = c->rng_mode; a.k_iter = 0; a.seed = c->seed;
  a.Z = (c->d_Z && t > 0) ? c->d_Z + (size_t)(t - 1) * s->Nglob * nw : nullptr;
  a.odo = c->d_odo + (size_t)(t > 0 ? t - 1 : 0) * c->mdl.nodo;
  a.cholQ = c->d_cholQ + (size_t)((c->chol_pages > 1 && t > 0) ? t - 1 : 0) * nw * nw;
  a.y = c->d_y + (size_t)t * d;
  a.xref = nullptr; a.status = c->d_flags; a.info = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->timing_on) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, c->stream)); }
  HIPCHK(launch_step(a, c->stream));
  if (c->timing_on) { HIPCHK(hipEventRecord(e1, c->stream)); c->events.emplace_back(e0, e1); }
  HIPCHK(hipStreamSynchronize(c->stream));     // logw_local / xn_local feed the next collective
  c->cur = nb;
  c->t = t + 1;
  return RBPF_OK;
}

int rbpf_shard_trajectories(rbpf_ctx* c, double* traj_max, double* traj_mean) {
  if (!c || !c->sh) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  RB_TRY(ctx_check_flags(c));
  const size_t cnt = (size_t)c->sh->t_norm * c->mdl.nN;
  if (traj_max) HIPCHK(hipMemcpy(traj_max, c->traj_max, cnt * sizeof(double), hipMemcpyDeviceToHost));
  if (traj_mean) HIPCHK(hipMemcpy(traj_mean, c->traj_mean, cnt * sizeof(double), hipMemcpyDeviceToHost));
  return RBPF_OK;
}

}  // extern "C"
