// State of one rank of the particle-sharded filter / smoother (shared by rbpf_shard.hip and rbpf_smoother.hip).
#pragma once
#include "rbpf_ctx.hpp"
#include "rbpf_plan.hpp"

namespace rbpf {

struct ShardState {
  int rank = 0, world = 1, Nloc = 0, Nglob = 0;
  size_t recsz = 0, recv_cap = 0, send_cap = 0;
  int fwd_rows = 0;                // rows of the forward bank: nN + 1 (states, log-weight); + 1 in the smoother (anc_local)
  double* fwd_local = nullptr;     // [fwd_rows][Nloc]
  double* fwd_gather = nullptr;    // [world][fwd_rows][Nloc]
  double* logw_glob = nullptr;     // [Nglob] logical order
  double* xn_glob = nullptr;       // SoA [nN][Nglob] logical order
  double* w_glob = nullptr;        // [Nglob]
  double* wc_glob = nullptr;       // [Nglob]
  int* ai_glob = nullptr;          // [Nglob] ancestors by logical id
  int* perm = nullptr;             // [Nglob] phys_of_logical
  int* ai_bank = nullptr;          // [Nloc]
  int* slot_ids = nullptr;         // [Nloc]
  int* pack_idx = nullptr;         // [send_cap]
  double* send_rec = nullptr;
  double* recv_rec = nullptr;
  int t_norm = 0;                  // steps normalised so far
  // device-side planner state
  PlanBuffers pb{};
  int* cur_gid = nullptr;          // [Nglob] location of every logical slot's current particle (null: identity)
  int* gid_buf[2] = {nullptr, nullptr};
  int gid_cur = 0;
  bool placed = false;             // false until the first planned generation (identity placement)
  bool plan_ready = false;         // a device plan for the next step exists
  bool host_planned = false;       // some step was placed by a host plan: the device tables no longer locate the particles
  long long* counts_pin = nullptr; // pinned host copy of [send counts | recv counts | migrated]
  int last_send_total = 0;
  // multi-step lazy update: received records persist until the next flush (imported lineages use them as base)
  std::vector<int> rec_used_all;   // records alive in every rank's receive buffer (replicated bookkeeping)
  size_t step_cap = 0;             // records one rank may send / receive per step (identical on every rank)
  // shared flush (rbpf_api.hip): smallest child per parent key (bank entry or received record), destination entry, phase
  int* d_share_lead = nullptr; size_t share_keys = 0;
  int* d_share = nullptr;          // [2][Nloc]
  int regrown = 0;                 // times the exchange buffers were grown (rbpf_shard_plan): callers re-read rbpf_shard_views
  bool async = false;              // no stream synchronisation at the end of pack / step (collectives on the same stream)
  int rec_used = 0;                // records currently alive in recv_rec
  int plan_recv = 0;               // records the pending plan will append
  // ---- sharded information-form smoother (rbpf_smoother.hip) ----
  bool smoother = false;
  int k_iter = 0;                  // current CPF-AS iteration
  size_t recsz_base = 0;           // doubles of the filter part [Pt | Pb | F | xl] of a record
  size_t rec_off_I = 0, rec_off_hld = 0, rec_off_Hb = 0, rec_off_Imat = 0;   // information part of a record
  double* Xhist = nullptr;         // [T][nN][Nglob] states of every step, logical order (replicated)
  int* Ahist = nullptr;            // [T][Nglob] ancestors of every step, logical ids (replicated)
  double* anc_local = nullptr;     // [Nloc] measurement part of the ancestor log-weights of my particles, physical order:
                                   // the LAST ROW of fwd_local, so that one all_gather moves it with the forward bank
  double* anc_gather = nullptr;    // [world][Nloc] all_gather target
  double* anc_glob = nullptr;      // [Nglob] logical order
  double* anc_w = nullptr;         // [Nglob] normalised ancestor probabilities (paNt)
  double* anc_wc = nullptr;        // [Nglob]
  double* w_local = nullptr;       // [Nloc] normalised weights of my particles, physical order
  int* ident_bank = nullptr;       // [Nloc] 0..Nloc-1
};

int shard_create_impl(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
                      int32_t rank, int32_t world, bool smoother, int N_K, rbpf_ctx** out);
// one sharded time step (rbpf_shard_step); k_iter / xref_t / info: CPF-AS iteration, state of the reference slot
// (logical id Nglob - 1) and information-form buffers of the sharded smoother (null for the filter)
int shard_step_impl(rbpf_ctx* c, const int32_t* anc_bank_host, const int32_t* slot_ids_host, int k_iter,
                    const double* xref_t, const InfoStep* info);
// ai_host value of shard_normalise_impl that draws the ancestors but leaves them on the device and does not synchronise
inline int32_t* draw_only_tag() { return reinterpret_cast<int32_t*>(static_cast<intptr_t>(-1)); }
// rbpf_shard_normalise_search with the draw count (N for the filter, N - 1 when slot N - 1 is the reference
// trajectory) and the iteration whose RNG page is used
int shard_normalise_impl(rbpf_ctx* c, const int32_t* perm_host, int32_t* ai_host, int k_iter, int n_draw);
// information part of the send records (ivec, halfLogDetP, pending H, Imat) of `count` local particles
int shard_smoother_pack_info(rbpf_ctx* c, const int* d_idx, int count);
int shard_unpack_particle(rbpf_ctx* c, int idx, double* dP);
// doubles of the matrix part of a smoother record: Imat [n x n], or the carried factor in sweep layout (chol_refresh > 1)
size_t smoother_record_matrix_doubles(int n, int d, int chol_refresh);

}  // namespace rbpf
