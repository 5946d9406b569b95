// Particle-sharded forward filter: kernels and buffers between the collectives.
// The collectives themselves (one all_gather of the forward bank, one all_to_all of particle records)
// are issued by the host mirror (multigpu.py) through torch.distributed / RCCL on the pointers exposed
// here.  Every rank normalises the *global* weight vector and draws the *global* ancestor vector with
// the same kernels as the single-GPU path, and logical slots keep their RNG streams wherever their
// particle lives, so a W-rank run equals the single-GPU run with N = W * N_local bit for bit.
#include "../../include/rbpf.h"
#include "rbpf_internal.hpp"
#include "rbpf_ctx.hpp"
#include "rbpf_plan.hpp"
#include "rbpf_shard_state.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace rbpf {

void shard_free(rbpf_ctx* c) {
  ShardState* s = c->sh;
  if (!s) return;
  hipFree(s->fwd_local); hipFree(s->fwd_gather); hipFree(s->logw_glob); hipFree(s->xn_glob); hipFree(s->w_glob);
  hipFree(s->wc_glob); hipFree(s->ai_glob); hipFree(s->perm); hipFree(s->ai_bank); hipFree(s->slot_ids);
  hipFree(s->pack_idx); hipFree(s->send_rec); hipFree(s->recv_rec); hipFree(s->d_share_lead); hipFree(s->d_share);
  hipFree(s->pb.key); hipFree(s->pb.counts); hipFree(s->pb.offsets); hipFree(s->pb.fill); hipFree(s->pb.tmp);
  hipFree(s->pb.order); hipFree(s->pb.mv_child); hipFree(s->pb.mv_src); hipFree(s->pb.mv_q); hipFree(s->pb.pref);
  hipFree(s->pb.slot_ids); hipFree(s->pb.anc_bank); hipFree(s->pb.send_idx); hipFree(s->pb.scalars); hipFree(s->pb.counts_dev);
  hipFree(s->gid_buf[0]); hipFree(s->gid_buf[1]);
  hipFree(s->Xhist); hipFree(s->Ahist); hipFree(s->anc_gather); hipFree(s->anc_glob);      // anc_local is a row of fwd_local
  hipFree(s->anc_w); hipFree(s->anc_wc); hipFree(s->w_local); hipFree(s->ident_bank);
  if (s->counts_pin) hipHostFree(s->counts_pin);
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

int rbpf::shard_create_impl(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
                            int32_t rank, int32_t world, bool smoother, int N_K, rbpf_ctx** out) {
  if (!prob || !out || world < 1 || rank < 0 || rank >= world) { set_error("bad shard arguments"); return RBPF_ERR_INVALID_ARG; }
  RB_TRY(options_ok(opt));
  if (prob->x0_lin_cols != 1) { set_error("sharded filter: x0_lin must be nLin x 1"); return RBPF_ERR_UNSUPPORTED; }
  const size_t Nloc = (size_t)prob->N_P;
  if (world > kMaxWorld) { set_error("world size above 64 is not supported"); return RBPF_ERR_UNSUPPORTED; }
  if (prob->N_P < 1 || Nloc * (size_t)world > (size_t)kMaxParticles) {
    set_error("world * N_local above 1048576 particles is not supported (resample pipeline: 1024 blocks of 1024)"); return RBPF_ERR_UNSUPPORTED;
  }
  CreateExtras ex;
  ex.bank_extra = 0;
  ex.rng_slots = Nloc * world;
  rbpf_options o;
  if (opt) o = *opt; else std::memset(&o, 0, sizeof(o));
  o.keep_history = 0;
  o.trace = 0;
  rbpf_ctx* c = nullptr;
  RB_TRY(ctx_create(model, prob, rng, &o, smoother, N_K, &c, &ex));
  ShardState* s = new ShardState();
  c->sh = s;
  s->rank = rank; s->world = world; s->Nloc = (int)Nloc; s->Nglob = (int)(Nloc * world);
  const Layout& L = c->lay;
  const int nN = c->mdl.nN, d = c->mdl.d;
  // covariance blocks in their stored type (float: half the doubles), then the fp64 factors and mean
  const size_t per = c->fp32 ? 2 : 1;
  s->recsz = (L.szT + L.szB) / per + (size_t)2 * d * L.ldx + L.ldx;
  s->recsz += s->recsz & 1;
  s->recsz_base = s->recsz;
  s->smoother = smoother;
  if (smoother) {
    // information part of a record: [ivec ldx | halfLogDetP, pad | pending H d*ldx | Imat n*n], 16-byte aligned; with carried
    // factors (chol_refresh > 1) the matrix part is the ancestor-weight factor in sweep layout instead of Imat
    const size_t n = (size_t)c->mdl.n;
    s->rec_off_I = s->recsz_base;
    s->rec_off_hld = s->rec_off_I + L.ldx;
    s->rec_off_Hb = s->rec_off_hld + 2;
    s->rec_off_Imat = s->rec_off_Hb + (size_t)d * L.ldx;
    s->recsz = s->rec_off_Imat + smoother_record_matrix_doubles((int)n, (int)d, resolve_chol_refresh(c->mdl.kind, (int)n, (int)d, effective_chol_refresh(o)));
    s->recsz += s->recsz & 1;
  }
  if (world > 1) {
    // Exchange buffers, sized identically on every rank (a deterministic function of the problem and the options, never
    // of the free memory, so that every rank reaches the same verdict about a step's exchange): `step_cap` records may
    // leave / enter a rank per time step; received records stay alive until the next flush of the lazy update.  With
    // owner-computes placement a step moves the load imbalance only -- and one record per (destination, ancestor) pair,
    // however many children it has -- so the default is an eighth of the local particles (measured need at N_local = 4096: 1 %,
    // tools/sharded_rehearsal.py; at N_local = 32 768, nLin = 515, lazy_depth 4 the buffers then take 44 GB next to 139 GB of banks).
    // (r03: the buffers grow on demand -- rbpf_shard_plan -- so the default starts at a sixteenth: four times the measured need.)
    const size_t dflt = std::min(Nloc, std::max<size_t>(256, Nloc / 16));
    const size_t asked = (size_t)(o.exchange_capacity > 0 ? o.exchange_capacity : -(long long)o.exchange_capacity);
    s->step_cap = asked > 0 ? std::min<size_t>(asked, Nloc * (size_t)(world - 1)) : dflt;
    s->send_cap = s->step_cap;
    s->recv_cap = std::min(s->step_cap, Nloc) * (size_t)std::max(c->lazy_depth, 1);
  }
  s->rec_used_all.assign((size_t)world, 0);
  int st = RBPF_OK;
  auto A = [&](int r) { if (st == RBPF_OK) st = r; };
  s->fwd_rows = nN + 1 + (smoother ? 1 : 0);
  A(dmalloc(&s->fwd_local, Nloc * (size_t)s->fwd_rows));
  A(dmalloc(&s->fwd_gather, (size_t)s->Nglob * s->fwd_rows));
  A(dmalloc(&s->logw_glob, (size_t)s->Nglob));
  A(dmalloc(&s->xn_glob, (size_t)s->Nglob * nN));
  A(dmalloc(&s->w_glob, (size_t)s->Nglob));
  A(dmalloc(&s->wc_glob, (size_t)s->Nglob));
  A(dmalloc(&s->ai_glob, (size_t)s->Nglob));
  A(dmalloc(&s->perm, (size_t)s->Nglob));
  A(dmalloc(&s->ai_bank, Nloc));
  A(dmalloc(&s->slot_ids, Nloc));
  A(dmalloc(&s->pack_idx, s->send_cap));
  A(dmalloc(&s->send_rec, s->send_cap * s->recsz));
  A(dmalloc(&s->recv_rec, s->recv_cap * s->recsz));
  {
    const size_t Ng = (size_t)s->Nglob;
    A(dmalloc(&s->pb.key, Ng)); A(dmalloc(&s->pb.counts, Ng + 1)); A(dmalloc(&s->pb.offsets, Ng + 1));
    A(dmalloc(&s->pb.fill, Ng)); A(dmalloc(&s->pb.tmp, Ng)); A(dmalloc(&s->pb.order, Ng));
    A(dmalloc(&s->pb.mv_child, Ng)); A(dmalloc(&s->pb.mv_src, Ng)); A(dmalloc(&s->pb.mv_q, Ng)); A(dmalloc(&s->pb.pref, Ng));
    A(dmalloc(&s->pb.slot_ids, Nloc)); A(dmalloc(&s->pb.anc_bank, Nloc)); A(dmalloc(&s->pb.send_idx, Ng));
    A(dmalloc(&s->pb.scalars, 1)); A(dmalloc(&s->pb.counts_dev, (size_t)4 * world + 1));
    A(dmalloc(&s->gid_buf[0], Ng)); A(dmalloc(&s->gid_buf[1], Ng));
    if (st == RBPF_OK && hipHostMalloc((void**)&s->counts_pin, ((size_t)4 * world + 1) * sizeof(long long)) != hipSuccess) st = RBPF_ERR_OUT_OF_MEMORY;
  }
  if (smoother || (opt && opt->keep_history)) {
    // global history (replicated): states and ancestors of every step in logical order, for the trajectory draws
    const size_t Ng = (size_t)s->Nglob, T = (size_t)prob->N_T;
    A(dmalloc(&s->Xhist, T * nN * Ng)); A(dmalloc(&s->Ahist, T * Ng));
    if (st == RBPF_OK && hipMemset(s->Ahist, 0, T * Ng * sizeof(int)) != hipSuccess) st = RBPF_ERR_HIP;
  }
  if (smoother) {
    const size_t Ng = (size_t)s->Nglob;
    if (st == RBPF_OK) {
      s->anc_local = s->fwd_local + (size_t)(nN + 1) * Nloc;          // alias: the extra row of the forward bank
      if (hipMemset(s->anc_local, 0, Nloc * sizeof(double)) != hipSuccess) st = RBPF_ERR_HIP;
    }
    A(dmalloc(&s->anc_gather, Ng)); A(dmalloc(&s->anc_glob, Ng));
    A(dmalloc(&s->anc_w, Ng)); A(dmalloc(&s->anc_wc, Ng)); A(dmalloc(&s->ident_bank, Nloc));
  }
  A(dmalloc(&s->w_local, Nloc));
  if (st != RBPF_OK) { ctx_free(c); return st; }
  *out = c;
  return RBPF_OK;
}

extern "C" {

int rbpf_shard_create(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
                      int32_t rank, int32_t world, rbpf_ctx** out) {
  return shard_create_impl(model, prob, rng, opt, rank, world, false, 1, out);
}

int rbpf_shard_views_get(rbpf_ctx* c, rbpf_shard_views* v) {
  if (!c || !c->sh || !v) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  ShardState* s = c->sh;
  v->rank = s->rank; v->world = s->world; v->N_local = s->Nloc; v->N_global = s->Nglob; v->n_nonlin = c->mdl.nN;
  v->record_doubles = s->recsz; v->recv_capacity = s->recv_cap; v->send_capacity = s->send_cap;
  v->fwd_local = s->fwd_local; v->fwd_gather = s->fwd_gather; v->send_rec = s->send_rec; v->recv_rec = s->recv_rec;
  v->fwd_rows = s->fwd_rows;
  return RBPF_OK;
}

int rbpf_shard_normalise_search(rbpf_ctx* c, const int32_t* perm_host, int32_t* ai_host) {
  if (!c || !c->sh) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  return shard_normalise_impl(c, perm_host, ai_host, 0, c->sh->Nglob);
}

}  // extern "C"

// ai_host == kDrawOnly: draw the ancestors but leave them on the device (device planner: nobody reads them on the host)
static int32_t* const kDrawOnly = rbpf::draw_only_tag();

int rbpf::shard_normalise_impl(rbpf_ctx* c, const int32_t* perm_host, int32_t* ai_host, int k_iter, int n_draw) {
  HIPCHK(hipSetDevice(c->device));
  ShardState* s = c->sh;
  const int nN = c->mdl.nN, N = s->Nglob;
  const int t_done = s->t_norm;                // index of the step whose weights are being normalised
  if (t_done >= c->T || t_done >= c->t) { set_error("normalise without a finished step"); return RBPF_ERR_STATE; }
  const int* perm = s->placed ? s->cur_gid : nullptr;     // device-planned placement (identity before the first plan)
  if (perm_host) {
    HIPCHK(hipMemcpyAsync(s->perm, perm_host, (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->stream));
    perm = s->perm;
  }
  // (smoother: the extra row -- the measurement part of the ancestor log-weights -- comes along into anc_glob)
  HIPCHK(launch_permute_fwd(N, nN, s->world, s->Nloc, perm, s->fwd_gather, s->logw_glob, s->xn_glob, c->stream, s->fwd_rows,
                            s->smoother ? s->anc_glob : nullptr));
  NormArgs nm;
  nm.N = N; nm.nN = nN; nm.t = t_done; nm.logw = s->logw_glob; nm.w = s->w_glob; nm.wc = s->wc_glob; nm.xn = s->xn_glob;
  nm.traj_max = c->traj_max + (size_t)t_done * nN; nm.traj_mean = c->traj_mean + (size_t)t_done * nN;
  nm.iw_max = c->d_flags + 2; nm.lse_out = nullptr;
  nm.parallel_scan = 1;
  if (ai_host) {
    const int t = c->t;                        // the step about to run
    SearchArgs sa;
    sa.N = N; sa.n_draw = n_draw; sa.t = t; sa.wc = s->wc_glob; sa.rng_mode = c->rng_mode; sa.k_iter = k_iter; sa.slot0 = 0;
    sa.u_is_scalar = 0;
    sa.U = c->d_U ? c->d_U + ((size_t)k_iter * std::max(c->T - 1, 0) + (size_t)(t - 1)) * N : nullptr;
    sa.seed = c->seed; sa.ai = s->ai_glob; sa.overflow = c->d_flags + 1;
    sa.approx = 1; sa.ambiguous = c->d_flags + 4; sa.w = s->w_glob; sa.wc_exact = s->wc_glob;
    if (N > kSingleWgResampleMaxN) HIPCHK(launch_resample_pipeline(nm, &sa, nullptr, nullptr, nullptr, c->d_rs, c->stream));
    else HIPCHK(launch_normalise_resample(nm, sa, nullptr, nullptr, c->stream));
    if (ai_host != kDrawOnly) HIPCHK(hipMemcpyAsync(ai_host, s->ai_glob, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  } else if (N > kSingleWgResampleMaxN) {
    HIPCHK(launch_resample_pipeline(nm, nullptr, nullptr, nullptr, nullptr, c->d_rs, c->stream));
  } else {
    HIPCHK(launch_normalise_scan(nm, c->stream));
  }
  if (s->Xhist) {       // smoother: keep every step's states / ancestors (logical order) for the trajectory draw
    HIPCHK(hipMemcpyAsync(s->Xhist + (size_t)t_done * nN * N, s->xn_glob, (size_t)nN * N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (ai_host && c->t < c->T)
      HIPCHK(hipMemcpyAsync(s->Ahist + (size_t)c->t * N, s->ai_glob, (size_t)n_draw * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
  }
  s->t_norm = t_done + 1;
  if (ai_host != kDrawOnly) HIPCHK(hipStreamSynchronize(c->stream));
  return RBPF_OK;
}

extern "C" {

// rbpf_shard_normalise_search + rbpf_shard_plan in one call for the device planner: the ancestors stay on the device
// and the stream is synchronised once (by the plan's count read-back).
int rbpf_shard_normalise_plan(rbpf_ctx* c, int64_t* counts_host) {
  if (!c || !c->sh || !counts_host) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  if (c->sh->smoother) { set_error("smoother context: use the rbpf_shard_smoother_* sequence"); return RBPF_ERR_STATE; }
  RB_TRY(shard_normalise_impl(c, nullptr, kDrawOnly, 0, c->sh->Nglob));
  return rbpf_shard_plan(c, counts_host);
}

int rbpf_shard_pack(rbpf_ctx* c, const int32_t* idx_host, int32_t count) {
  if (!c || !c->sh || count < 0) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  ShardState* s = c->sh;
  if ((size_t)count > s->send_cap) { set_error("send buffer too small for this step's exchange"); return RBPF_ERR_OUT_OF_MEMORY; }
  if (count == 0) return RBPF_OK;
  const int* idx = s->pb.send_idx;                        // device plan
  if (idx_host) {
    HIPCHK(hipMemcpyAsync(s->pack_idx, idx_host, (size_t)count * sizeof(int), hipMemcpyHostToDevice, c->stream));
    idx = s->pack_idx;
  }
  const int ob = c->cur;
  if (c->lazy_depth >= 2) {
    // state after step t-1: ell pending sets per lineage; apply them while packing
    const int C = c->lazy_depth, B = C + 1, t = c->t, N = s->Nloc;
    const int ell = (t == 0) ? 0 : ((t - 1) % C) + 1;
    const double* fset[kMaxSets]; const int* fidx[kMaxSets];
    for (int q = 0; q < ell; ++q) { const int bank = (t - ell + q) % B; fset[q] = c->Fb[bank]; fidx[q] = c->fidx[c->tcur] + (size_t)bank * N; }
    HIPCHK(launch_pack_records_flushed(c->lay, c->mdl.d, idx, count, c->Pt[ob], c->Pb[ob], ell, fset, fidx, c->base[c->tcur], N,
                                       s->recv_rec, s->recsz, c->xl[c->xcur], s->send_rec, c->stream, c->fp32 ? 1 : 0));
  } else {
    HIPCHK(launch_pack_records(c->lay, c->mdl.d, idx, count, c->Pt[ob], c->Pb[ob], c->F[ob], c->xl[ob], s->send_rec,
                               c->stream, s->recsz, c->fp32 ? 1 : 0));
  }
  if (s->smoother) RB_TRY(shard_smoother_pack_info(c, idx, count));
  // the collective runs on another stream / library; and `idx_host` is caller memory read by an asynchronous copy: a host
  // plan always synchronises, whatever rbpf_shard_set_async says (only the device-plan path is sync-free)
  if (!s->async || idx_host) HIPCHK(hipStreamSynchronize(c->stream));
  return RBPF_OK;
}

int rbpf_shard_step(rbpf_ctx* c, const int32_t* anc_bank_host, const int32_t* slot_ids_host) {
  if (!c || !c->sh) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  if (c->sh->smoother) { set_error("smoother context: use rbpf_shard_smoother_step"); return RBPF_ERR_STATE; }
  return shard_step_impl(c, anc_bank_host, slot_ids_host, 0, nullptr, nullptr);
}

}  // extern "C"

int rbpf::shard_step_impl(rbpf_ctx* c, const int32_t* anc_bank_host, const int32_t* slot_ids_host, int k_iter,
                          const double* xref_t, const InfoStep* info) {
  HIPCHK(hipSetDevice(c->device));
  ShardState* s = c->sh;
  const int t = c->t, N = s->Nloc, nN = c->mdl.nN, d = c->mdl.d, nw = c->mdl.nw;
  if (t >= c->T) { set_error("advance past N_T"); return RBPF_ERR_STATE; }
  const bool dev_plan = (t > 0) && !anc_bank_host && !slot_ids_host && s->plan_ready;
  if (t > 0 && !dev_plan && (!anc_bank_host || !slot_ids_host)) {
    set_error("t > 0 needs anc_bank + slot_ids, or a device plan (rbpf_shard_plan)"); return RBPF_ERR_INVALID_ARG;
  }
  if (t == 0 && (anc_bank_host || slot_ids_host)) { set_error("anc_bank / slot_ids must be NULL at t = 0"); return RBPF_ERR_INVALID_ARG; }
  const Layout& L = c->lay;
  StepArgs a;
  std::memset(&a, 0, sizeof(a));
  a.mdl = c->mdl; a.lay = L; a.N = N; a.t = t; a.propagate = (t > 0);
  a.n_sets = (t > 0) ? 1 : 0; a.write_base = 1;          // one pending set, rewritten every step
  a.slot_offset = s->rank * N;
  a.xn_new = s->fwd_local; a.xn_new_stride = (size_t)N;
  a.logw = s->fwd_local + (size_t)nN * N;
  const bool lazy = c->lazy_depth >= 2;
  if (lazy && t > 0 && !dev_plan) { set_error("lazy_depth >= 2 in the sharded filter needs the device planner"); return RBPF_ERR_UNSUPPORTED; }
  const int ob = c->cur;
  int nb = (t == 0) ? 0 : (c->cur ^ 1);
  const int xo = c->xcur, xn = (t == 0) ? 0 : (c->xcur ^ 1);
  const int told = c->tcur, tnew = c->tcur ^ 1;
  bool flush = true;
  a.zero_set_idx = N;
  if (lazy) {
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
    a.xn_old = c->X; a.xn_old_stride = (size_t)N;           // filled with x0 by ctx_reset
    a.xl_old = c->d_x0l; a.xl_old_stride = 0; a.F_old = nullptr;
    a.Pt_old = c->d_P0t; a.Pb_old = c->d_P0b; a.Pt_old_stride = 0; a.Pb_old_stride = 0;
  } else {
    if (dev_plan) {
      a.slot_ids = s->pb.slot_ids;
      a.ai_bank = s->pb.anc_bank;
    } else {
      HIPCHK(hipMemcpyAsync(s->ai_bank, anc_bank_host, (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(s->slot_ids, slot_ids_host, (size_t)N * sizeof(int), hipMemcpyHostToDevice, c->stream));
      a.slot_ids = s->slot_ids;
      a.ai_bank = s->ai_bank;
    }
    a.ai = s->ai_glob;                                       // indexed by logical slot id
    a.xn_old = s->xn_glob; a.xn_old_stride = (size_t)s->Nglob;
    a.xl_old = c->xl[xo]; a.xl_old_stride = (size_t)L.ldx; a.F_old = lazy ? nullptr : c->F[ob];
    a.Pt_old = c->Pt[ob]; a.Pb_old = c->Pb[ob]; a.Pt_old_stride = L.szT; a.Pb_old_stride = L.szB;
    a.n_bank_local = N;
    a.rec = s->recv_rec; a.rec_stride = s->recsz;
    const size_t per = c->fp32 ? 2 : 1;                      // stored elements per double in the covariance blocks
    a.rec_off_B = L.szT / per; a.rec_off_F = (L.szT + L.szB) / per; a.rec_off_X = a.rec_off_F + (size_t)2 * d * L.ldx;
    a.rec_off_I = s->rec_off_I; a.rec_off_hld = s->rec_off_hld;
    // the host places the new generation in ancestor order, so physical order is already cache-friendly
  }
  a.xl_new = c->xl[xn]; a.F_new = lazy ? c->Fb[t % (c->lazy_depth + 1)] : c->F[nb];
  a.Pt_new = c->Pt[nb]; a.Pb_new = c->Pb[nb];
  a.fp32 = c->fp32 ? 1 : 0;
  a.strip_ws = c->d_strip_ws; a.strip_ws_stride = c->strip_ws_stride;
  a.rng_mode = c->rng_mode; a.k_iter = k_iter; a.seed = c->seed;
  a.Z = (c->d_Z && t > 0) ? c->d_Z + ((size_t)k_iter * std::max(c->T - 1, 0) + (size_t)(t - 1)) * s->Nglob * nw : nullptr;
  a.xref = xref_t; a.xref_gslot = s->Nglob - 1;
  a.info = info ? 1 : 0;
  if (info) {
    a.ivec_old = info->ivec_old; a.ivec_old_stride = info->ivec_old_stride; a.ivec_new = info->ivec_new;
    a.hld_old = info->hld_old; a.hld_old_stride = info->hld_old_stride; a.hld_new = info->hld_new;
    a.qf_new = info->qf_new; a.Hb_new = info->Hb_new;
  }
  a.odo = c->d_odo + (size_t)(t > 0 ? t - 1 : 0) * c->mdl.nodo;
  a.cholQ = c->d_cholQ + (size_t)((c->chol_pages > 1 && t > 0) ? t - 1 : 0) * nw * nw;
  a.y = c->d_y + (size_t)t * d;
  a.status = c->d_flags;
  a.stamps = nullptr;
  a.pre_i = c->d_pre_i; a.pre_d = c->d_pre_d; a.u_next = nullptr;
  a.phase = -1;                                            // (every slot; the shared flush below launches by phase)
  // shared flush (see ctx_step): the children of one parent -- one bank entry or one received record -- store ONE flushed matrix
  const bool share = lazy && flush && t > 0 && dev_plan && L.sym && (L.CH64 == 8 || L.CH64 == 16) && a.n_sets >= 1 && a.n_sets <= (info ? 3 : 7);
  if (share) {
    const size_t keys = (size_t)N + s->recv_cap;
    if (s->share_keys < keys) {
      HIPCHK(hipStreamSynchronize(c->stream));
      hipFree(s->d_share_lead); s->d_share_lead = nullptr;
      RB_TRY(dmalloc(&s->d_share_lead, keys));
      s->share_keys = keys;
    }
    if (!s->d_share) RB_TRY(dmalloc(&s->d_share, (size_t)2 * N));
    if (!c->d_share_writers) { RB_TRY(dmalloc(&c->d_share_writers, 1)); HIPCHK(hipMemset(c->d_share_writers, 0, sizeof(unsigned long long))); c->share_flush = true; }
    HIPCHK(launch_share_plan(N, (int)keys, a.ai_bank, s->d_share_lead, s->d_share, s->d_share + N, c->timing_on ? c->d_share_writers : nullptr, c->stream));
    a.dst_slot = s->d_share; a.phase_of = s->d_share + N; a.share_flush = 1;
  }
  RB_TRY(ctx_arm_distinct(c, a, (size_t)N + s->recv_cap + 1));
  HIPCHK(launch_propagate(a, c->stream));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->timing_on) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, c->stream)); }
  if (share) {
    a.phase = 1; HIPCHK(launch_step(a, c->stream));                 // writers: the flush variant
    StepArgs rd = a;
    rd.phase = 0; rd.write_base = 0;                                // readers: the read-only variant with the same pending sets
    HIPCHK(launch_step(rd, c->stream));
    if (c->timing_on) c->share_flush_particles += N;
  } else {
    HIPCHK(launch_step(a, c->stream));
  }
  if (c->timing_on) { HIPCHK(hipEventRecord(e1, c->stream)); c->events.emplace_back(e0, e1); ctx_account_launch(c, a); }
  // fwd_local feeds the next collective; host arrays (anc_bank / slot_ids) are caller memory behind asynchronous copies
  if (!s->async || anc_bank_host || slot_ids_host) HIPCHK(hipStreamSynchronize(c->stream));
  if (t > 0 && !dev_plan) s->host_planned = true;           // placement unknown to the library from here on (rbpf_shard_finish)
  if (dev_plan) {                                           // commit the placement of the new generation
    s->cur_gid = s->pb.new_gid;
    s->gid_cur ^= 1;
    s->placed = true;
    s->plan_ready = false;
  }
  if (lazy) {
    // records received for this step stay alive (imported lineages keep them as base) until the next flush
    s->rec_used = flush ? 0 : s->rec_used + s->plan_recv;
    for (int q = 0; q < s->world; ++q)
      s->rec_used_all[q] = flush ? 0 : s->rec_used_all[q] + (dev_plan ? (int)s->counts_pin[2 * s->world + 1 + q] : 0);
    s->plan_recv = 0;
    c->tcur = tnew;
  }
  c->cur = nb;
  c->xcur = lazy ? xn : nb;
  c->t = t + 1;
  return RBPF_OK;
}

extern "C" {

// Device-side placement + exchange plan for the step about to run, from the ancestors drawn by the last
// rbpf_shard_normalise_search.  counts_host [2*world+1]: records to send to / receive from each rank, then the
// number of migrating children.  Equivalent to multigpu.plan_generation + rank_view on the host.
int rbpf_shard_plan(rbpf_ctx* c, int64_t* counts_host) {
  if (!c || !c->sh || !counts_host) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  ShardState* s = c->sh;
  if (c->t < 1 || s->t_norm < c->t) { set_error("plan needs the ancestors of the next step (normalise_search first)"); return RBPF_ERR_STATE; }
  s->pb.new_gid = s->gid_buf[s->gid_cur ^ 1];
  const int rec_off = (c->lazy_depth >= 2) ? s->rec_used : 0;
  HIPCHK(plan_run(s->pb, s->Nglob, s->world, s->Nloc, s->rank, s->ai_glob, s->placed ? s->cur_gid : nullptr, rec_off, c->stream));
  HIPCHK(hipMemcpyAsync(s->counts_pin, s->pb.counts_dev, ((size_t)4 * s->world + 1) * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int q = 0; q < 2 * s->world + 1; ++q) counts_host[q] = (int64_t)s->counts_pin[q];
  counts_host[2 * s->world + 1] = rec_off;                 // first record of recv_rec the exchange may write
  s->plan_recv = 0;
  for (int q = 0; q < s->world; ++q) s->plan_recv += (int)s->counts_pin[s->world + q];
  // Capacity verdict, identical on every rank: the plan and the buffer sizes are replicated, so every rank checks every
  // rank's exchange BEFORE a collective is issued -- a step that does not fit is an error everywhere, not a hang.  With an
  // explicit exchange_capacity > 0 that is final (RBPF_ERR_OUT_OF_MEMORY on every rank); with the default (0) or a starting
  // capacity (< 0) every rank GROWS its buffers by the same deterministic rule (the verdict and the old sizes are replicated, so
  // no communication is needed) and the step goes on: a 3000-step run is not lost to one unusually skewed generation.
  if (s->world > 1) {
    long long need_send = 0, need_hold = 0; int worst = -1;
    for (int q = 0; q < s->world; ++q) {
      const long long rcv = s->counts_pin[2 * s->world + 1 + q], snd = s->counts_pin[3 * s->world + 1 + q];
      const long long alive = (c->lazy_depth >= 2) ? s->rec_used_all[q] : 0;
      if ((size_t)(alive + rcv) > s->recv_cap || (size_t)snd > s->send_cap) worst = q;
      need_send = std::max(need_send, snd); need_hold = std::max(need_hold, alive + rcv);
    }
    if (worst >= 0) {
      if (c->opt.exchange_capacity > 0) {
        const long long rcv = s->counts_pin[2 * s->world + 1 + worst], snd = s->counts_pin[3 * s->world + 1 + worst];
        const long long alive = (c->lazy_depth >= 2) ? s->rec_used_all[worst] : 0;
        char buf[256];
        snprintf(buf, sizeof(buf), "exchange of step %d does not fit: rank %d would send %lld and hold %lld received records "
                 "(capacity %zu / %zu); raise rbpf_options.exchange_capacity", c->t, worst, snd, alive + rcv, s->send_cap, s->recv_cap);
        set_error(buf);
        return RBPF_ERR_OUT_OF_MEMORY;
      }
      const size_t lz = (size_t)std::max(c->lazy_depth, 1);
      size_t cap = std::max<size_t>(s->step_cap, 1);
      while (cap < (size_t)need_send || std::min(cap, (size_t)s->Nloc) * lz < (size_t)need_hold) {
        if (cap >= (size_t)s->Nloc * (size_t)(s->world - 1)) break;
        cap *= 2;
      }
      cap = std::min(cap, (size_t)s->Nloc * (size_t)(s->world - 1));
      const size_t new_send = cap, new_recv = std::max(std::min(cap, (size_t)s->Nloc) * lz, (size_t)need_hold);
      double *ns = nullptr, *nr = nullptr; int* ni = nullptr;
      int st = dmalloc(&ns, new_send * s->recsz);
      if (st == RBPF_OK) st = dmalloc(&nr, new_recv * s->recsz);
      if (st == RBPF_OK) st = dmalloc(&ni, new_send);
      if (st != RBPF_OK) {
        hipFree(ns); hipFree(nr); hipFree(ni);
        set_error("exchange buffers could not be grown to " + std::to_string(new_send) + " / " + std::to_string(new_recv) + " records");
        return RBPF_ERR_OUT_OF_MEMORY;
      }
      // records received earlier in this lazy cycle stay alive: they move to the new buffer
      if (s->rec_used > 0) HIPCHK(hipMemcpyAsync(nr, s->recv_rec, (size_t)s->rec_used * s->recsz * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      hipFree(s->send_rec); hipFree(s->recv_rec); hipFree(s->pack_idx);
      s->send_rec = ns; s->recv_rec = nr; s->pack_idx = ni;
      s->step_cap = cap; s->send_cap = new_send; s->recv_cap = new_recv;
      ++s->regrown;
    }
  }
  s->plan_ready = true;
  return RBPF_OK;
}

// Test hook: this rank's view of the current device plan (slot_ids, anc_bank [N_local]; send_idx [n_send]).
int rbpf_shard_plan_read(rbpf_ctx* c, int32_t* slot_ids, int32_t* anc_bank, int32_t* send_idx, int32_t n_send, int32_t* new_gid) {
  if (!c || !c->sh) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  ShardState* s = c->sh;
  if (slot_ids) HIPCHK(hipMemcpy(slot_ids, s->pb.slot_ids, (size_t)s->Nloc * 4, hipMemcpyDeviceToHost));
  if (anc_bank) HIPCHK(hipMemcpy(anc_bank, s->pb.anc_bank, (size_t)s->Nloc * 4, hipMemcpyDeviceToHost));
  if (send_idx && n_send > 0) HIPCHK(hipMemcpy(send_idx, s->pb.send_idx, (size_t)n_send * 4, hipMemcpyDeviceToHost));
  if (new_gid) HIPCHK(hipMemcpy(new_gid, s->pb.new_gid, (size_t)s->Nglob * 4, hipMemcpyDeviceToHost));
  return RBPF_OK;
}

int rbpf_stream_get(rbpf_ctx* c, void** hip_stream) {
  if (!c || !hip_stream) { set_error("NULL argument"); return RBPF_ERR_INVALID_ARG; }
  *hip_stream = reinterpret_cast<void*>(c->stream);
  return RBPF_OK;
}

int rbpf_shard_set_async(rbpf_ctx* c, int32_t on) {
  if (!c || !c->sh) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  c->sh->async = on != 0;
  return RBPF_OK;
}

int rbpf_shard_set_ancestors(rbpf_ctx* c, const int32_t* ai) {
  if (!c || !c->sh || !ai) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  ShardState* s = c->sh;
  for (int i = 0; i < s->Nglob; ++i)
    if (ai[i] < 0 || ai[i] >= s->Nglob) { set_error("ancestor out of range"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipMemcpy(s->ai_glob, ai, (size_t)s->Nglob * sizeof(int), hipMemcpyHostToDevice));
  if (s->Ahist && c->t < c->T) HIPCHK(hipMemcpy(s->Ahist + (size_t)c->t * s->Nglob, ai, (size_t)s->Nglob * sizeof(int), hipMemcpyHostToDevice));
  return RBPF_OK;
}

}  // extern "C"

namespace rbpf {
// w_local[p] = w_glob[logical id of physical slot p]
__global__ void shard_w_local_kernel(int Nloc, int slot0, const int* __restrict__ slot_ids, const double* __restrict__ w_glob,
                                     double* __restrict__ w_local) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < Nloc) w_local[p] = w_glob[slot_ids ? slot_ids[p] : slot0 + p];
}
}  // namespace rbpf

// Covariance of local particle `idx` as of the last finished step in MATLAB layout (every pending downdate applied; the
// stored matrix of its lineage may sit in a received record), into the device buffer dP [n x n].
int rbpf::shard_unpack_particle(rbpf_ctx* c, int idx, double* dP) {
  ShardState* s = c->sh;
  const Layout& L = c->lay;
  const int d = c->mdl.d, N = s->Nloc;
  int* didx = nullptr;
  RB_TRY(dmalloc(&didx, 1));
  hipError_t e = hipMemcpyAsync(didx, &idx, sizeof(int), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && c->lazy_depth >= 2) {
    const int C = c->lazy_depth, B = C + 1, t = c->t;
    const int ell = (t == 0) ? 0 : ((t - 1) % C) + 1;
    const double* fset[kMaxSets]; const int* fidx[kMaxSets];
    for (int q = 0; q < ell; ++q) { const int bank = (t - ell + q) % B; fset[q] = c->Fb[bank]; fidx[q] = c->fidx[c->tcur] + (size_t)bank * N; }
    double* rec = nullptr;
    int st = dmalloc(&rec, s->recsz);
    if (st != RBPF_OK) { hipFree(didx); return st; }
    e = launch_pack_records_flushed(L, d, didx, 1, c->Pt[c->cur], c->Pb[c->cur], ell, fset, fidx, c->base[c->tcur], N, s->recv_rec,
                                    s->recsz, c->xl[c->xcur], rec, c->stream, c->fp32 ? 1 : 0);
    const size_t per = c->fp32 ? 2 : 1;
    if (e == hipSuccess) e = launch_unpack_P(L, d, rec, rec + L.szT / per, nullptr, nullptr, 1, dP, c->stream, c->fp32 ? 1 : 0);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(rec);
  } else if (e == hipSuccess) {
    e = launch_unpack_P(L, d, c->Pt[c->cur], c->Pb[c->cur], c->t > 0 ? c->F[c->cur] : nullptr, didx, 1, dP, c->stream, c->fp32 ? 1 : 0);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  hipFree(didx);
  HIPCHK(e);
  return RBPF_OK;
}

extern "C" {

int rbpf_shard_finish(rbpf_ctx* c, int32_t phase, double* xl_max, double* P_max, double* xl_mean, double* P_mean,
                      double* traj_sample_iwmax, int32_t* iw_max) {
  if (!c || !c->sh || (phase != 0 && phase != 1)) { set_error("not a shard context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  ShardState* s = c->sh;
  if (s->smoother) { set_error("smoother context: use rbpf_shard_smoother_end"); return RBPF_ERR_STATE; }
  if (c->t < 1 || s->t_norm != c->t) { set_error("finish needs the last finished step gathered and normalised"); return RBPF_ERR_STATE; }
  if (s->host_planned) {
    // owner / slot / weight of a logical particle come from the device planner's tables; after a step placed by a HOST plan
    // (rbpf_shard_step with anc_bank / slot_ids) they are unknown here and an identity guess would silently return the wrong
    // particles' xl_max / P_max / xl_mean / P_mean
    set_error("rbpf_shard_finish needs every step placed by the device planner (rbpf_shard_plan); this session took a host plan");
    return RBPF_ERR_STATE;
  }
  RB_TRY(ctx_check_flags(c));
  const int n = c->mdl.n, nN = c->mdl.nN, N = s->Nglob, Nloc = s->Nloc, Td = c->t;
  const Layout& L = c->lay;
  hipStream_t st = c->stream;
  // where a logical slot's particle lives now
  auto locate = [&](int logical, int& owner, int& idx) -> int {
    int gid = logical;
    if (s->placed) HIPCHK(hipMemcpy(&gid, s->cur_gid + logical, sizeof(int), hipMemcpyDeviceToHost));
    owner = gid / Nloc; idx = gid % Nloc;
    return RBPF_OK;
  };
  if (phase == 0) {
    int iw = 0;
    HIPCHK(hipMemcpy(&iw, c->d_flags + 2, sizeof(int), hipMemcpyDeviceToHost));
    if (iw_max) *iw_max = iw;
    int owner = 0, idx = 0;
    RB_TRY(locate(iw, owner, idx));
    if (xl_max) {
      std::memset(xl_max, 0, (size_t)n * sizeof(double));
      if (owner == s->rank) HIPCHK(hipMemcpy(xl_max, c->xl[c->xcur] + (size_t)idx * L.ldx, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (P_max) {
      std::memset(P_max, 0, (size_t)n * n * sizeof(double));
      if (owner == s->rank) {
        double* dP = nullptr;
        RB_TRY(dmalloc(&dP, (size_t)n * n));
        int rc = shard_unpack_particle(c, idx, dP);
        hipError_t e = (rc == RBPF_OK) ? hipMemcpy(P_max, dP, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost) : hipSuccess;
        hipFree(dP);
        if (rc != RBPF_OK) return rc;
        HIPCHK(e);
      }
    }
    if (xl_mean) {                       // my share of sum_i w_i xl_i (particleFilter.m:224)
      hipLaunchKernelGGL(shard_w_local_kernel, dim3((Nloc + 255) / 256), dim3(256), 0, st, Nloc, s->rank * Nloc,
                         s->placed ? s->pb.slot_ids : nullptr, s->w_glob, s->w_local);
      double* dm = nullptr;
      RB_TRY(dmalloc(&dm, (size_t)n));
      hipError_t e = launch_weighted_mean_xl(Nloc, n, L.ldx, c->xl[c->xcur], s->w_local, dm, st);
      if (e == hipSuccess) e = hipMemcpyAsync(xl_mean, dm, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      hipFree(dm);
      HIPCHK(e);
    }
    if (traj_sample_iwmax) {             // :233, back-trace through the replicated history
      if (!s->Xhist) { set_error("traj_sample_iwmax needs keep_history = 1"); return RBPF_ERR_STATE; }
      double* dout = nullptr; int* didx = nullptr;
      RB_TRY(dmalloc(&dout, (size_t)nN * Td));
      int s2 = dmalloc(&didx, 1);
      if (s2 != RBPF_OK) { hipFree(dout); return s2; }
      hipError_t e = hipMemcpy(didx, &iw, sizeof(int), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = launch_backtrace(N, nN, Td, s->Xhist, s->Ahist, didx, 1, dout, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e == hipSuccess) e = hipMemcpy(traj_sample_iwmax, dout, (size_t)nN * Td * sizeof(double), hipMemcpyDeviceToHost);
      hipFree(dout); hipFree(didx);
      HIPCHK(e);
    }
    return RBPF_OK;
  }
  // phase 1: quirk Q3 -- P_mean is the LAST particle's term only (particleFilter.m:228-230 assigns instead of accumulating)
  if (!P_mean || !xl_mean) { set_error("phase 1 needs xl_mean (input) and P_mean"); return RBPF_ERR_INVALID_ARG; }
  if (c->opt.fix_p_mean) { set_error("fix_p_mean = 1 is not available in the sharded session"); return RBPF_ERR_UNSUPPORTED; }
  std::memset(P_mean, 0, (size_t)n * n * sizeof(double));
  int owner = 0, idx = 0;
  RB_TRY(locate(N - 1, owner, idx));
  if (owner == s->rank) {
    double* dP = nullptr;
    RB_TRY(dmalloc(&dP, (size_t)n * n));
    std::vector<double> Pl((size_t)n * n), xll(n);
    double wl = 0.0;
    int rc = shard_unpack_particle(c, idx, dP);
    hipError_t e = (rc == RBPF_OK) ? hipMemcpy(Pl.data(), dP, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost) : hipSuccess;
    hipFree(dP);
    if (rc != RBPF_OK) return rc;
    if (e == hipSuccess) e = hipMemcpy(xll.data(), c->xl[c->xcur] + (size_t)idx * L.ldx, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&wl, s->w_glob + (N - 1), sizeof(double), hipMemcpyDeviceToHost);
    HIPCHK(e);
    for (int cc = 0; cc < n; ++cc)
      for (int r = 0; r < n; ++r)
        P_mean[r + (size_t)n * cc] = wl * (Pl[r + (size_t)n * cc] + (xl_mean[r] - xll[r]) * (xl_mean[cc] - xll[cc]));
  }
  return RBPF_OK;
}

int rbpf_shard_xn_traj(rbpf_ctx* c, double* xn_traj) {
  if (!c || !c->sh || !xn_traj) { set_error("not a shard context / NULL output"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  ShardState* s = c->sh;
  if (s->smoother) { set_error("smoother context: xn_traj is a particleFilter output"); return RBPF_ERR_STATE; }
  if (!s->Xhist) { set_error("xn_traj needs keep_history = 1"); return RBPF_ERR_STATE; }
  if (c->t < 1) { set_error("xn_traj needs at least one finished step"); return RBPF_ERR_STATE; }
  RB_TRY(ctx_check_flags(c));
  const int nN = c->mdl.nN, N = s->Nglob, Td = c->t;
  // The output is nN * N * T doubles (9.4 GB at N = 65 536, T = 3000, nN = 6) and the banks may have left little of the device: the
  // paths are traced in chunks through a scratch buffer of at most 256 MB (rows of the host array are nN * N doubles apart, a
  // chunk's are nN * cnt: one strided copy per chunk)
  const size_t budget = (size_t)256 << 20;
  const int chunk = (int)std::max<size_t>(64, std::min<size_t>((size_t)N, budget / ((size_t)nN * Td * sizeof(double))));
  double* dout = nullptr;
  RB_TRY(dmalloc(&dout, (size_t)nN * chunk * Td));
  hipError_t e = hipSuccess;
  for (int p0 = 0; p0 < N && e == hipSuccess; p0 += chunk) {
    const int cnt = std::min(chunk, N - p0);
    e = launch_backtrace(N, nN, Td, s->Xhist, s->Ahist, nullptr, cnt, dout, c->stream, p0);      // particleFilter.m:117-118
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy2D(xn_traj + (size_t)nN * p0, (size_t)nN * N * sizeof(double), dout, (size_t)nN * cnt * sizeof(double),
                                         (size_t)nN * cnt * sizeof(double), (size_t)Td, hipMemcpyDeviceToHost);
  }
  hipFree(dout);
  HIPCHK(e);
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
