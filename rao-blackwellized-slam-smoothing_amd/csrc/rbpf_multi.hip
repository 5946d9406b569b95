// In-library multi-device driver: ONE host process (e.g. a MATLAB session behind the MEX gateway) runs the particle-sharded
// filter / information-form smoother over several GPUs of a node (BASELINE.json north_star "host code stays MATLAB ...
// partitioned across the 8 GPUs of one node").
//
// The sharded algorithm itself is unchanged (rbpf_shard.hip, rbpf_smoother.hip: owner-computes placement, one all-gather of the
// forward bank + one all-to-all of the migrating particle records per step); what moves into the library is the step loop that
// multigpu.py drives from Python under torchrun: one C++ thread per device runs gather -> normalise + draw + plan -> pack ->
// exchange -> step on that device's context and stream, and the collectives are issued by the library itself --
//
//   * RCCL (ncclCommInitAll: one communicator per device of this process, ncclAllGather and grouped ncclSend / ncclRecv on
//     each context's own stream, so a step needs one host synchronisation: the plan's count read-back), loaded with dlopen at
//     the first multi-device call (the single-GPU path has no RCCL dependency), or
//   * a host-staged transport (pinned-free plain host buffers + a thread barrier) when the device list names a device more than
//     once -- several ranks sharing one GPU, which is how a one-GPU box exercises the world-2 loop (RCCL refuses duplicates).
//
// Selected by rbpf_options.n_devices (> 1, or == 1 with device_ids set: the loop with a world of one) in the ordinary one-shot
// entry points rbpf_particle_filter / rbpf_particle_smoother, so the MATLAB wrappers shard with an option and nothing else.
#include "rbpf_ctx.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace rbpf {

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string& why) {
    if (h) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) { why = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return false; }
#define RB_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(h, sym)); if (!field) { why = std::string("RCCL lacks ") + sym; return false; }
    RB_SYM(CommInitAll, "ncclCommInitAll") RB_SYM(CommDestroy, "ncclCommDestroy") RB_SYM(CommAbort, "ncclCommAbort") RB_SYM(AllGather, "ncclAllGather")
    RB_SYM(Send, "ncclSend") RB_SYM(Recv, "ncclRecv") RB_SYM(GroupStart, "ncclGroupStart") RB_SYM(GroupEnd, "ncclGroupEnd")
    RB_SYM(GetErrorString, "ncclGetErrorString")
#undef RB_SYM
    return true;
  }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

// reusable barrier for the host-staged transport; abort() releases everybody (a rank failed)
struct HostBarrier {
  std::mutex m; std::condition_variable cv; int n = 1, waiting = 0; unsigned gen = 0; bool aborted = false;
  bool wait() {
    std::unique_lock<std::mutex> lk(m);
    if (aborted) return false;
    const unsigned g = gen;
    if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); return true; }
    cv.wait(lk, [&] { return gen != g || aborted; });
    return !aborted;
  }
  void abort() { std::lock_guard<std::mutex> lk(m); aborted = true; cv.notify_all(); }
};

struct Multi {
  int W = 1;
  std::vector<int> devs;
  bool host_staged = false, smoother = false;
  int N_K = 1, Nloc = 0, Nglob = 0, T = 0, nN = 0, n = 0;
  std::vector<rbpf_ctx*> ctx;
  std::vector<rbpf_shard_views> v;
  std::vector<hipStream_t> stream;
  std::vector<ncclComm_t> comm;
  std::vector<rbpf_shard_smoother_views> sv;       // smoother only
  int chol_refresh = 0;
  // host staging (host_staged only)
  std::vector<double> h_fwd;                       // [W][fwd_rows * Nloc]
  std::vector<std::vector<double>> h_send;         // per rank: its send records of this step
  std::vector<std::vector<long long>> h_cnt;       // per rank: counts_host of this step
  HostBarrier bar;
  // per-rank status of the last parallel section
  std::vector<int> status; std::vector<std::string> msg;
  long long migrated = 0, sent_records = 0;
  // RCCL transport: a rank that fails between collectives aborts EVERY communicator of this process (ncclCommAbort makes the
  // collective kernels its peers wait behind give up), so that the peers' stream synchronisations return, their threads join and the
  // entry point reports the first failure instead of hanging.  The handles stay where they are (peers read comm[r] concurrently: no
  // write to the vector after creation); `comms_aborted` is checked before every collective and keeps the destructor from destroying
  // aborted communicators; the rank that failed FIRST is recorded by a compare-exchange, not recovered from message texts.
  std::mutex abort_m;
  std::atomic<bool> comms_aborted{false};
  std::atomic<int> first_failed{-1};
  int home_device = -1;                            // the caller's current device, restored when the call returns
  void abort_comms() {
    std::lock_guard<std::mutex> lk(abort_m);
    if (comms_aborted.load() || host_staged || !g_rccl.CommAbort) return;
    comms_aborted.store(true);
    for (auto c : comm) if (c) g_rccl.CommAbort(c);
  }
  Multi() { if (hipGetDevice(&home_device) != hipSuccess) home_device = -1; }
  ~Multi() {
    for (size_t r = 0; r < ctx.size(); ++r) {
      if (!ctx[r]) continue;
      hipSetDevice(devs[r]);
      rbpf_destroy(ctx[r]);
    }
    if (!comms_aborted.load()) for (auto c : comm) if (c && g_rccl.CommDestroy) g_rccl.CommDestroy(c);
    if (home_device >= 0) hipSetDevice(home_device);
  }
};

#define MT_TRY(expr) do { int _s = (expr); if (_s != RBPF_OK) return _s; } while (0)
#define MT_NCCL(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) { set_error(std::string(#expr " failed: ") + g_rccl.GetErrorString(_r)); return RBPF_ERR_HIP; } } while (0)

// every rank runs fn(rank) on its own thread with its device current; the first failing rank's status / message is returned
template <typename F>
int run_ranks(Multi& M, F fn) {
  M.status.assign(M.W, RBPF_OK); M.msg.assign(M.W, std::string());
  auto body = [&](int r) {
    int s = (hipSetDevice(M.devs[r]) == hipSuccess) ? RBPF_OK : RBPF_ERR_HIP;
    if (s == RBPF_OK) s = fn(r);
    M.status[r] = s;
    if (s != RBPF_OK) {
      int none = -1;
      M.first_failed.compare_exchange_strong(none, r);         // the first rank to get here failed on its own
      const char* e = rbpf_last_error(); M.msg[r] = e ? e : ""; M.bar.abort(); M.abort_comms();
    }
  };
  if (M.W == 1) body(0);
  else {
    std::vector<std::thread> th;
    for (int r = 0; r < M.W; ++r) th.emplace_back(body, r);
    for (auto& t : th) t.join();
  }
  int first = M.first_failed.load();               // the rank that failed on its own, not one released by its abort
  if (first < 0) for (int r = 0; r < M.W && first < 0; ++r) if (M.status[r] != RBPF_OK) first = r;
  M.first_failed.store(-1);
  if (first >= 0) { set_error("device " + std::to_string(M.devs[first]) + " (rank " + std::to_string(first) + "): " + M.msg[first]); return M.status[first]; }
  return RBPF_OK;
}

int gather_rows(Multi& M, int r, const double* local, double* gathered, size_t count) {
  if (!M.host_staged) {
    if (M.comms_aborted.load()) { set_error("another rank failed"); return RBPF_ERR_STATE; }
    MT_NCCL(g_rccl.AllGather(local, gathered, count, ncclDouble, M.comm[r], M.stream[r]));
    return RBPF_OK;
  }
  HIPCHK(hipStreamSynchronize(M.stream[r]));
  if (M.h_fwd.size() < (size_t)M.W * count) { set_error("host staging buffer too small"); return RBPF_ERR_STATE; }
  HIPCHK(hipMemcpy(M.h_fwd.data() + (size_t)r * count, local, count * sizeof(double), hipMemcpyDeviceToHost));
  if (!M.bar.wait()) { set_error("another rank failed"); return RBPF_ERR_STATE; }
  HIPCHK(hipMemcpy(gathered, M.h_fwd.data(), (size_t)M.W * count * sizeof(double), hipMemcpyHostToDevice));
  if (!M.bar.wait()) { set_error("another rank failed"); return RBPF_ERR_STATE; }
  return RBPF_OK;
}

int gather_fwd(Multi& M, int r) {
  return gather_rows(M, r, M.v[r].fwd_local, M.v[r].fwd_gather, (size_t)M.v[r].fwd_rows * M.Nloc);
}

// all-to-all of whole rows of `width` doubles: snd[q] rows go to rank q (concatenated in rank order in send), rcv[q] rows arrive
// from rank q at recv + recv_off rows.  Identical plans on every rank (replicated), so the host-staged variant only needs barriers.
int exchange_rows(Multi& M, int r, const double* send, double* recv, const long long* snd, const long long* rcv, size_t width,
                  size_t recv_off) {
  const int W = M.W;
  long long ns = 0;
  for (int q = 0; q < W; ++q) ns += snd[q];
  if (!M.host_staged) {
    if (M.comms_aborted.load()) { set_error("another rank failed"); return RBPF_ERR_STATE; }
    MT_NCCL(g_rccl.GroupStart());
    size_t so = 0, ro = recv_off;
    for (int q = 0; q < W; ++q) {
      if (snd[q] > 0) MT_NCCL(g_rccl.Send(send + so * width, (size_t)snd[q] * width, ncclDouble, q, M.comm[r], M.stream[r]));
      if (rcv[q] > 0) MT_NCCL(g_rccl.Recv(recv + ro * width, (size_t)rcv[q] * width, ncclDouble, q, M.comm[r], M.stream[r]));
      so += (size_t)snd[q]; ro += (size_t)rcv[q];
    }
    MT_NCCL(g_rccl.GroupEnd());
    return RBPF_OK;
  }
  HIPCHK(hipStreamSynchronize(M.stream[r]));
  M.h_cnt[r].assign(snd, snd + W);
  M.h_send[r].resize((size_t)ns * width);
  if (ns) HIPCHK(hipMemcpy(M.h_send[r].data(), send, (size_t)ns * width * sizeof(double), hipMemcpyDeviceToHost));
  if (!M.bar.wait()) { set_error("another rank failed"); return RBPF_ERR_STATE; }
  size_t ro = recv_off;
  for (int q = 0; q < W; ++q) {
    const long long want = rcv[q];
    if (want <= 0) continue;
    size_t so = 0;
    for (int p = 0; p < r; ++p) so += (size_t)M.h_cnt[q][p];            // rows rank q sends to ranks before me
    if (M.h_cnt[q][r] != want) { set_error("exchange plan mismatch between ranks"); return RBPF_ERR_STATE; }
    HIPCHK(hipMemcpy(recv + ro * width, M.h_send[q].data() + so * width, (size_t)want * width * sizeof(double), hipMemcpyHostToDevice));
    ro += (size_t)want;
  }
  if (!M.bar.wait()) { set_error("another rank failed"); return RBPF_ERR_STATE; }
  return RBPF_OK;
}

// all-to-all of the migrating particle records; cnt = counts_host of rbpf_shard_plan: [send to q | receive from q | migrated | first
// record of recv_rec to write]
int exchange(Multi& M, int r, const long long* cnt) {
  const int W = M.W;
  const size_t rs = M.v[r].record_doubles;
  long long ns = 0;
  for (int q = 0; q < W; ++q) ns += cnt[q];
  const long long recv_off = cnt[2 * W + 1];
  MT_TRY(rbpf_shard_views_get(M.ctx[r], &M.v[r]));             // the plan may have grown the record buffers (exchange_capacity <= 0)
  MT_TRY(rbpf_shard_pack(M.ctx[r], nullptr, (int32_t)ns));
  if (r == 0) { M.migrated += cnt[2 * W]; }
  return exchange_rows(M, r, M.v[r].send_rec, M.v[r].recv_rec, cnt, cnt + W, rs, (size_t)recv_off);
}

// Which base matrices cross ranks at a refresh of the carried factors (multigpu.plan_refresh, the numpy specification).  owner_now[j]
// = rank * N_local + slot of logical slot j's particle, base_loc[j] = the same for the matrix its information matrix is rebuilt
// from; both replicated, so every rank derives the same plan: rank q sends rank r each matrix some particle on r needs, once,
// ordered by (destination, source, slot).
struct RefreshPlan {
  std::vector<int32_t> send_slots, base_index;
  std::vector<long long> send_counts, recv_counts, send_totals, recv_totals;
};
RefreshPlan plan_refresh(const int32_t* owner_now, const int32_t* base_loc, size_t N, int n_local, int world, int rank) {
  RefreshPlan P;
  std::vector<long long> keys;
  keys.reserve(N);
  for (size_t j = 0; j < N; ++j) {
    const long long r = owner_now[j] / n_local, q = base_loc[j] / n_local, sl = base_loc[j] % n_local;
    if (r != q) keys.push_back((r * world + q) * n_local + sl);
  }
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  P.send_counts.assign(world, 0); P.recv_counts.assign(world, 0); P.send_totals.assign(world, 0); P.recv_totals.assign(world, 0);
  std::vector<long long> recv_keys;
  for (long long key : keys) {
    const int ur = (int)(key / ((long long)world * n_local)), uq = (int)((key / n_local) % world), us = (int)(key % n_local);
    P.send_totals[uq]++; P.recv_totals[ur]++;
    if (uq == rank) { P.send_slots.push_back(us); P.send_counts[ur]++; }
    if (ur == rank) { recv_keys.push_back(key); P.recv_counts[uq]++; }
  }
  P.base_index.assign(n_local, 0);
  for (size_t j = 0; j < N; ++j) {
    const long long r = owner_now[j] / n_local, p = owner_now[j] % n_local, q = base_loc[j] / n_local, sl = base_loc[j] % n_local;
    if (r != rank) continue;
    if (r == q) P.base_index[p] = (int32_t)sl;
    else {
      const long long key = (r * world + q) * n_local + sl;
      P.base_index[p] = n_local + (int32_t)(std::lower_bound(recv_keys.begin(), recv_keys.end(), key) - recv_keys.begin());
    }
  }
  return P;
}

// The refresh of the carried factors in place of rbpf_shard_smoother_anc_weights (multigpu.ShardedSmootherSession._refresh)
int refresh(Multi& M, int r) {
  std::vector<int32_t> own(M.Nglob), bl(M.Nglob);
  MT_TRY(rbpf_shard_smoother_refresh_begin(M.ctx[r], own.data(), bl.data()));
  if (bl[0] < 0) return rbpf_shard_smoother_refresh_end(M.ctx[r], nullptr, 0);     // first refresh of an iteration: the common initial matrix
  RefreshPlan P = plan_refresh(own.data(), bl.data(), own.size(), M.Nloc, M.W, r);
  long long worst = 0, ns = 0, nr = 0;
  for (int q = 0; q < M.W; ++q) { worst = std::max(worst, std::max(P.send_totals[q], P.recv_totals[q])); ns += P.send_counts[q]; nr += P.recv_counts[q]; }
  if (worst > (long long)M.sv[r].refresh_capacity) {                                // replicated plan: every rank reaches this verdict
    MT_TRY(rbpf_shard_smoother_refresh_reserve(M.ctx[r], worst));                  // grows by a replicated rule, or fails on every rank
    MT_TRY(rbpf_shard_smoother_views_get(M.ctx[r], &M.sv[r]));
  }
  MT_TRY(rbpf_shard_smoother_refresh_pack(M.ctx[r], ns ? P.send_slots.data() : nullptr, (int32_t)ns));
  MT_TRY(exchange_rows(M, r, M.sv[r].refresh_send, M.sv[r].refresh_recv, P.send_counts.data(), P.recv_counts.data(),
                       (size_t)M.sv[r].matrix_doubles, 0));
  return rbpf_shard_smoother_refresh_end(M.ctx[r], P.base_index.data(), (int32_t)nr);
}

int create(Multi& M, const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt, bool smoother,
           int N_K) {
  const int W = opt->n_devices;
  if (W < 1 || W > 64) { set_error("options.n_devices must be in 1..64"); return RBPF_ERR_INVALID_ARG; }
  if (prob->N_P % W) { set_error("N_P must be a multiple of options.n_devices"); return RBPF_ERR_INVALID_ARG; }
  if (model->kind == RBPF_MODEL_GENERIC_DENSE || model->kind == RBPF_MODEL_SPARSE_VISUAL_2D) {
    set_error("options.n_devices: the recognised dense model families only (host callbacks are not sharded)"); return RBPF_ERR_UNSUPPORTED;
  }
  if (opt->on_step && !smoother) { set_error("options.n_devices: the per-step hook (makePlots) is not available in the sharded filter"); return RBPF_ERR_UNSUPPORTED; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device visible"); return RBPF_ERR_NO_DEVICE; }
  M.W = W; M.smoother = smoother; M.N_K = N_K;
  M.devs.resize(W);
  for (int r = 0; r < W; ++r) {
    M.devs[r] = opt->device_ids ? opt->device_ids[r] : r;
    if (M.devs[r] < 0 || M.devs[r] >= ndev) { set_error("options.device_ids names a device that does not exist (" + std::to_string(ndev) + " visible)"); return RBPF_ERR_INVALID_ARG; }
    for (int q = 0; q < r; ++q) if (M.devs[q] == M.devs[r]) M.host_staged = true;     // ranks sharing a GPU: RCCL refuses that
  }
  M.Nloc = prob->N_P / W; M.Nglob = prob->N_P; M.T = prob->N_T; M.nN = prob->n_nonlin; M.n = prob->n_lin;
  M.ctx.assign(W, nullptr); M.v.resize(W); M.stream.assign(W, nullptr); M.comm.assign(W, nullptr); M.sv.resize(W);
  M.chol_refresh = smoother ? resolve_chol_refresh(model->kind, prob->n_lin, prob->n_y, effective_chol_refresh(*opt)) : 0;
  M.h_send.resize(W); M.h_cnt.resize(W); M.bar.n = W;
  if (!M.host_staged) {
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    std::string why;
    if (!g_rccl.load(why)) { set_error(why); return RBPF_ERR_UNSUPPORTED; }
    MT_NCCL(g_rccl.CommInitAll(M.comm.data(), W, M.devs.data()));
  }
  rbpf_options o = *opt;
  o.n_devices = 0; o.device_ids = nullptr; o.on_step = nullptr; o.on_step_user = nullptr;
  MT_TRY(run_ranks(M, [&](int r) -> int {
    rbpf_problem p = *prob;
    p.N_P = M.Nloc;
    if (prob->x0_lin_cols > 1) p.x0_lin = prob->x0_lin + (size_t)r * M.Nloc * prob->n_lin;      // this rank's columns of x0_lin
    if (smoother) MT_TRY(rbpf_shard_smoother_create(model, &p, rng, &o, N_K, r, W, &M.ctx[r]));
    else MT_TRY(rbpf_shard_create(model, &p, rng, &o, r, W, &M.ctx[r]));
    MT_TRY(rbpf_shard_views_get(M.ctx[r], &M.v[r]));
    if (smoother) MT_TRY(rbpf_shard_smoother_views_get(M.ctx[r], &M.sv[r]));
    void* sp = nullptr;
    MT_TRY(rbpf_stream_get(M.ctx[r], &sp));
    M.stream[r] = reinterpret_cast<hipStream_t>(sp);
    return rbpf_shard_set_async(M.ctx[r], M.host_staged ? 0 : 1);
  }));
  if (M.host_staged) M.h_fwd.assign((size_t)W * std::max<size_t>((size_t)M.v[0].fwd_rows * M.Nloc, (size_t)M.Nloc), 0.0);
  return RBPF_OK;
}

// n_steps time steps of the sharded filter on rank r (multigpu.ShardedFilterSession.advance, device planner)
int filter_steps(Multi& M, int r, int n_steps) {
  const int W = M.W;
  std::vector<long long> cnt(2 * W + 2, 0);
  for (int sidx = 0; sidx < n_steps; ++sidx) {
    int32_t t = 0;
    MT_TRY(rbpf_filter_tell(M.ctx[r], &t));
    if (t > 0) {
      MT_TRY(gather_fwd(M, r));
      MT_TRY(rbpf_shard_normalise_plan(M.ctx[r], reinterpret_cast<int64_t*>(cnt.data())));
      MT_TRY(exchange(M, r, cnt.data()));
    }
    MT_TRY(rbpf_shard_step(M.ctx[r], nullptr, nullptr));
  }
  return RBPF_OK;
}

}  // namespace

int multi_particle_filter(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
                          rbpf_filter_out* out) {
  if (!model || !prob || !rng || !opt || !out) { set_error("NULL argument"); return RBPF_ERR_INVALID_ARG; }
  if (out->trace_logw || out->trace_w || out->trace_ai || out->final_xn || out->final_xl || out->final_P) {
    set_error("options.n_devices: the traces and the final particle banks are not gathered from the sharded filter"); return RBPF_ERR_UNSUPPORTED;
  }
  if ((out->xn_traj || out->traj_sample_iwmax) && !opt->keep_history) { set_error("traj_sample_iwmax / xn_traj need keep_history=1"); return RBPF_ERR_STATE; }
  Multi M;
  MT_TRY(create(M, model, prob, rng, opt, false, 1));
  MT_TRY(run_ranks(M, [&](int r) { return filter_steps(M, r, M.T); }));
  const int n = M.n, nN = M.nN, T = M.T, W = M.W;
  // last step: gather + normalise (no draw), then the extraction of particleFilter.m:220-233 -- every rank fills what it holds
  std::vector<std::vector<double>> xl_max(W, std::vector<double>(n, 0.0)), xl_mean(W, std::vector<double>(n, 0.0)),
      P_max(W), P_mean(W);
  std::vector<int32_t> iw(W, 0);
  const bool need_mean = out->xl_mean || out->P_mean;
  MT_TRY(run_ranks(M, [&](int r) -> int {
    MT_TRY(gather_fwd(M, r));
    MT_TRY(rbpf_shard_normalise_search(M.ctx[r], nullptr, nullptr));
    if (out->P_max) P_max[r].assign((size_t)n * n, 0.0);
    MT_TRY(rbpf_shard_finish(M.ctx[r], 0, out->xl_max ? xl_max[r].data() : nullptr, out->P_max ? P_max[r].data() : nullptr,
                             need_mean ? xl_mean[r].data() : nullptr, nullptr, (r == 0) ? out->traj_sample_iwmax : nullptr, &iw[r]));
    if (r == 0) MT_TRY(rbpf_shard_trajectories(M.ctx[r], out->traj_max, out->traj_mean));
    if (r == 0 && out->xn_traj) MT_TRY(rbpf_shard_xn_traj(M.ctx[r], out->xn_traj));   // replicated history: one rank has it all
    return RBPF_OK;
  }));
  // the owner's rows (zeros elsewhere) / the ranks' shares of the weighted mean, summed in rank order
  std::vector<double> mean(n, 0.0);
  for (int r = 0; r < W; ++r) for (int q = 0; q < n; ++q) mean[q] += xl_mean[r][q];
  if (out->xl_max) { std::memset(out->xl_max, 0, (size_t)n * sizeof(double)); for (int r = 0; r < W; ++r) for (int q = 0; q < n; ++q) out->xl_max[q] += xl_max[r][q]; }
  if (out->P_max) { std::memset(out->P_max, 0, (size_t)n * n * sizeof(double)); for (int r = 0; r < W; ++r) for (size_t q = 0; q < (size_t)n * n; ++q) out->P_max[q] += P_max[r][q]; }
  if (out->xl_mean) std::memcpy(out->xl_mean, mean.data(), (size_t)n * sizeof(double));
  if (out->iw_max) *out->iw_max = iw[0];
  if (out->P_mean) {
    MT_TRY(run_ranks(M, [&](int r) -> int {
      P_mean[r].assign((size_t)n * n, 0.0);
      std::vector<double> xm(mean);
      return rbpf_shard_finish(M.ctx[r], 1, nullptr, nullptr, xm.data(), P_mean[r].data(), nullptr, nullptr);
    }));
    std::memset(out->P_mean, 0, (size_t)n * n * sizeof(double));
    for (int r = 0; r < W; ++r) for (size_t q = 0; q < (size_t)n * n; ++q) out->P_mean[q] += P_mean[r][q];
  }
  (void)nN; (void)T;
  return RBPF_OK;
}

int multi_particle_smoother(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
                            int32_t N_K, int32_t info_form, rbpf_smoother_out* out) {
  if (!model || !prob || !rng || !opt || !out) { set_error("NULL argument"); return RBPF_ERR_INVALID_ARG; }
  if (!info_form) { set_error("options.n_devices: the covariance-form smoother is not sharded (particleSmootherInformationForm is)"); return RBPF_ERR_UNSUPPORTED; }
  if (out->trace_logw || out->trace_w || out->trace_ai || out->trace_paNt) { set_error("options.n_devices: traces are not gathered from the sharded smoother"); return RBPF_ERR_UNSUPPORTED; }
  Multi M;
  MT_TRY(create(M, model, prob, rng, opt, true, N_K));
  const int n = M.n, nN = M.nN, T = M.T, W = M.W;
  for (int k = 0; k < N_K; ++k) {
    std::vector<std::vector<double>> xlk(W, std::vector<double>(n, 0.0)), pk(W, std::vector<double>((size_t)n * n, 0.0));
    std::vector<int32_t> ak(W, 0), owner(W, 0);
    double* XNK_k = out->XNK ? out->XNK + (size_t)k * nN * T : nullptr;
    std::vector<double> xnk_scratch((size_t)nN * T, 0.0);
    MT_TRY(run_ranks(M, [&](int r) -> int {
      // multigpu.ShardedSmootherSession.run, one CPF-AS iteration (particleSmootherInformationForm.m:98-362)
      std::vector<long long> cnt(2 * W + 2, 0);
      MT_TRY(rbpf_shard_smoother_begin(M.ctx[r], k));
      for (int t = 0; t < T; ++t) {
        if (t == 0) { MT_TRY(rbpf_shard_smoother_step(M.ctx[r])); continue; }
        const int K = M.chol_refresh;
        const bool rf = k > 0 && K > 1 && (t == 1 || (t - 1) % K == 0);    // refresh of the carried factors due at this step
        if (k > 0 && !rf) MT_TRY(rbpf_shard_smoother_anc_weights(M.ctx[r]));   // local factorisations / sweeps, before the gather that carries them
        MT_TRY(gather_fwd(M, r));
        MT_TRY(rbpf_shard_smoother_normalise(M.ctx[r], 1));
        if (k > 0) {
          if (rf) {                                                         // walks the state history: after gather + normalise, own all-gather
            MT_TRY(refresh(M, r));
            MT_TRY(gather_rows(M, r, M.sv[r].anc_local, M.sv[r].anc_gather, (size_t)M.Nloc));
          }
          MT_TRY(rbpf_shard_smoother_anc_sample(M.ctx[r], rf ? 1 : 0));
        }
        MT_TRY(rbpf_shard_plan(M.ctx[r], reinterpret_cast<int64_t*>(cnt.data())));
        MT_TRY(exchange(M, r, cnt.data()));
        MT_TRY(rbpf_shard_smoother_step(M.ctx[r]));
      }
      MT_TRY(gather_fwd(M, r));
      MT_TRY(rbpf_shard_smoother_normalise(M.ctx[r], 0));
      std::vector<double> xnk_r((size_t)nN * T, 0.0);
      MT_TRY(rbpf_shard_smoother_end(M.ctx[r], (r == 0) ? xnk_scratch.data() : xnk_r.data(), xlk[r].data(), pk[r].data(), &ak[r], &owner[r]));
      return RBPF_OK;
    }));
    if (XNK_k) std::memcpy(XNK_k, xnk_scratch.data(), (size_t)nN * T * sizeof(double));
    if (out->XLK) { double* d = out->XLK + (size_t)k * n; std::memset(d, 0, (size_t)n * sizeof(double)); for (int r = 0; r < W; ++r) for (int q = 0; q < n; ++q) d[q] += xlk[r][q]; }
    if (out->PK) { double* d = out->PK + (size_t)k * n * n; std::memset(d, 0, (size_t)n * n * sizeof(double)); for (int r = 0; r < W; ++r) for (size_t q = 0; q < (size_t)n * n; ++q) d[q] += pk[r][q]; }
    if (out->trace_ak) out->trace_ak[k] = ak[0];
    if (opt->on_step) {                                                     // makePlots(xnk,xlk,k,XNK,XLK,PK), particleSmoother.m:360-362
      rbpf_view vw{};
      vw.ctx = nullptr; vw.t = k; vw.is_smoother = 1;
      if (opt->on_step(&vw, opt->on_step_user) != 0) { set_error("the on_step hook returned non-zero"); return RBPF_ERR_CALLBACK; }
    }
  }
  return RBPF_OK;
}

}  // namespace rbpf

extern "C" int rbpf_plan_refresh(const int32_t* owner_now, const int32_t* base_loc, int32_t N_global, int32_t n_local, int32_t world,
                                 int32_t rank, int32_t* send_slots, int32_t send_capacity, int32_t* n_send, int64_t* send_counts,
                                 int64_t* recv_counts, int64_t* send_totals, int64_t* recv_totals, int32_t* base_index) {
  using namespace rbpf;
  if (!owner_now || !base_loc || !n_send || !send_counts || !recv_counts || !send_totals || !recv_totals || !base_index ||
      N_global < 1 || n_local < 1 || world < 1 || rank < 0 || rank >= world || N_global != n_local * world) {
    set_error("rbpf_plan_refresh: NULL argument or inconsistent sizes"); return RBPF_ERR_INVALID_ARG;
  }
  for (int32_t j = 0; j < N_global; ++j)
    if (owner_now[j] < 0 || owner_now[j] >= N_global || base_loc[j] < 0 || base_loc[j] >= N_global) {
      set_error("rbpf_plan_refresh: table entry out of range"); return RBPF_ERR_INVALID_ARG;
    }
  RefreshPlan P = plan_refresh(owner_now, base_loc, (size_t)N_global, n_local, world, rank);
  *n_send = (int32_t)P.send_slots.size();
  if ((int32_t)P.send_slots.size() > send_capacity || (P.send_slots.size() && !send_slots)) { set_error("rbpf_plan_refresh: send_slots too small"); return RBPF_ERR_OUT_OF_MEMORY; }
  std::copy(P.send_slots.begin(), P.send_slots.end(), send_slots);
  for (int q = 0; q < world; ++q) {
    send_counts[q] = P.send_counts[q]; recv_counts[q] = P.recv_counts[q]; send_totals[q] = P.send_totals[q]; recv_totals[q] = P.recv_totals[q];
  }
  std::copy(P.base_index.begin(), P.base_index.end(), base_index);
  return RBPF_OK;
}
