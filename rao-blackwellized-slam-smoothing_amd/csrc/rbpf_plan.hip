// Device-side placement + exchange plan of the particle-sharded filter (the host/numpy planner in
// multigpu.py::plan_generation / rank_view is the specification; tests compare the two).
//
// Inputs (replicated on every rank): ai[i] = ancestor (logical id) of logical slot i, cur_gid[i] = global
// physical id (rank*N_local + slot) of logical slot i's current particle.  "Owner computes": a child is
// placed on its ancestor's rank; the last `excess` children (in ancestor-slot order, ties by logical id)
// of every overloaded rank move to the ranks with room, in rank order.  Every choice is a deterministic
// function of the inputs, so all ranks derive the same plan without communicating.
#include "rbpf_internal.hpp"
#include "rbpf_plan.hpp"

namespace rbpf {

constexpr int kPlanThreads = 1024;

__global__ void plan_key_kernel(int N, const int* __restrict__ ai, const int* __restrict__ cur_gid, int* __restrict__ key,
                                int* __restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int k = cur_gid ? cur_gid[ai[i]] : ai[i];
  key[i] = k;
  atomicAdd(&counts[k], 1);
}

// exclusive scan of counts[0..N) -> offsets[0..N], then the per-rank scalars (one workgroup); also clears fill[0..N).
// Chunks of 8192 counts go through LDS so that every global access is coalesced (thread t of the scan owns the 8
// consecutive counts 8 t .. 8 t + 7 of a chunk; with thread-strided global loads the lone workgroup took 97 us at
// N = 65536, a fifth of the replicated normalise + plan phase of a sharded step).
constexpr int kPlanChunk = 8 * kPlanThreads;
__device__ inline int plan_lds_pos(int e) { return 9 * (e >> 3) + (e & 7); }          // 8 counts per thread, stride 9: no bank conflicts
__global__ __launch_bounds__(kPlanThreads) void plan_scan_kernel(int N, int world, int nl, const int* __restrict__ counts,
                                                                 int* __restrict__ offsets, int* __restrict__ fill,
                                                                 PlanScalars* __restrict__ ps) {
  __shared__ int sdat[9 * kPlanThreads];
  __shared__ int swave[kPlanThreads / 64];
  __shared__ int scarry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) scarry = 0;
  for (int base = 0; base < N; base += kPlanChunk) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = tid + kPlanThreads * k, idx = base + e;
      sdat[plan_lds_pos(e)] = (idx < N) ? counts[idx] : 0;
      if (idx < N) fill[idx] = 0;
    }
    __syncthreads();
    int v[8], tot = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = sdat[9 * tid + k]; tot += v[k]; }
    int inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    if (lane == 63) swave[wave] = inc;
    __syncthreads();
    int run = scarry + inc - tot;
    for (int w = 0; w < wave; ++w) run += swave[w];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sdat[9 * tid + k] = run; run += v[k]; }
    __syncthreads();
    if (tid == kPlanThreads - 1) scarry = run;                               // total up to the end of this chunk
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = tid + kPlanThreads * k, idx = base + e;
      if (idx < N) offsets[idx] = sdat[plan_lds_pos(e)];
    }
  }
  __syncthreads();
  if (tid == 0) offsets[N] = scarry;
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    int start = 0, mv = 0, imp = 0;
    for (int r = 0; r < world; ++r) {
      const int load = offsets[(r + 1) * nl] - offsets[r * nl];
      const int excess = max(load - nl, 0), deficit = max(nl - load, 0);
      ps->start[r] = start; ps->stay_cnt[r] = load - excess; ps->mv_off[r] = mv; ps->imp_start[r] = imp;
      start += load; mv += excess; imp += deficit;
    }
    ps->start[world] = start; ps->mv_off[world] = mv; ps->imp_start[world] = imp;
    ps->M = mv;
  }
}

__global__ void plan_scatter_kernel(int N, const int* __restrict__ key, const int* __restrict__ offsets,
                                    int* __restrict__ fill, int* __restrict__ tmp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int k = key[i];
  tmp[offsets[k] + atomicAdd(&fill[k], 1)] = i;
}

// deterministic order inside each sibling group: position = number of siblings with a smaller logical id
__global__ void plan_rank_kernel(int N, const int* __restrict__ key, const int* __restrict__ offsets,
                                 const int* __restrict__ counts, const int* __restrict__ tmp, int* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int k = key[i], s0 = offsets[k], cnt = counts[k];
  int r = 0;
  for (int j = 0; j < cnt; ++j) r += (tmp[s0 + j] < i) ? 1 : 0;
  order[s0 + r] = i;
}

__global__ void plan_place_kernel(int N, int world, int nl, int me, const int* __restrict__ key,
                                  const int* __restrict__ order, const PlanScalars* __restrict__ ps,
                                  int* __restrict__ new_gid, int* __restrict__ slot_ids, int* __restrict__ anc_bank,
                                  int* __restrict__ mv_child, int* __restrict__ mv_src, int* __restrict__ mv_q) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N) return;
  const int i = order[s], k = key[i], grp = k / nl;
  const int pos = s - ps->start[grp];
  if (pos < ps->stay_cnt[grp]) {
    new_gid[i] = grp * nl + pos;
    if (grp == me) { slot_ids[pos] = i; anc_bank[pos] = k - me * nl; }
  } else {
    const int j = ps->mv_off[grp] + (pos - ps->stay_cnt[grp]);
    int q = 0;
    while (q + 1 < world && ps->imp_start[q + 1] <= j) ++q;
    const int idx = ps->stay_cnt[q] + (j - ps->imp_start[q]);
    new_gid[i] = q * nl + idx;
    mv_child[j] = i; mv_src[j] = k; mv_q[j] = q;
    if (q == me) slot_ids[idx] = i;               // anc_bank of imported children: plan_finish_kernel
  }
}

// unique (destination, ancestor) pairs among the M moved children: flags + inclusive prefix, then the
// send / receive counts of this rank (one workgroup)
__global__ __launch_bounds__(kPlanThreads) void plan_pairs_kernel(int world, int me, const PlanScalars* __restrict__ ps,
                                                                  const int* __restrict__ mv_src, const int* __restrict__ mv_q,
                                                                  int* __restrict__ pref, long long* __restrict__ counts_out) {
  __shared__ int swave[kPlanThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = ps->M;
  const int S = (M + kPlanThreads - 1) / kPlanThreads;
  const int j0 = min(tid * S, M), j1 = min(j0 + S, M);
  int tot = 0;
  for (int j = j0; j < j1; ++j) tot += (j == 0 || mv_src[j] != mv_src[j - 1] || mv_q[j] != mv_q[j - 1]) ? 1 : 0;
  int inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) swave[wave] = inc;
  __syncthreads();
  int run = inc - tot;
  for (int w = 0; w < wave; ++w) run += swave[w];
  for (int j = j0; j < j1; ++j) {
    run += (j == 0 || mv_src[j] != mv_src[j - 1] || mv_q[j] != mv_q[j - 1]) ? 1 : 0;
    pref[j] = run;
  }
  __syncthreads();
  // counts: flags inside [lo, hi) = pref[hi-1] - pref[lo-1]
  if (tid < world) {
    const int r = tid;
    auto cnt = [&](int lo, int hi) -> long long {
      if (hi <= lo) return 0;
      return (long long)pref[hi - 1] - (lo > 0 ? (long long)pref[lo - 1] : 0);
    };
    // what I send to rank r: source range (me) intersected with destination range (r)
    const int slo = max(ps->mv_off[me], ps->imp_start[r]), shi = min(ps->mv_off[me + 1], ps->imp_start[r + 1]);
    counts_out[r] = cnt(slo, shi);
    // what I receive from rank r: destination range (me) intersected with source range (r)
    const int rlo = max(ps->imp_start[me], ps->mv_off[r]), rhi = min(ps->imp_start[me + 1], ps->mv_off[r + 1]);
    counts_out[world + r] = cnt(rlo, rhi);
    // totals of EVERY rank (the plan is replicated, so every rank can check every rank's buffers before the exchange):
    // records rank r receives / sends in this step
    counts_out[2 * world + 1 + r] = cnt(ps->imp_start[r], ps->imp_start[r + 1]);
    counts_out[3 * world + 1 + r] = cnt(ps->mv_off[r], ps->mv_off[r + 1]);
  }
  if (tid == 0) counts_out[2 * world] = M;
}

__global__ void plan_finish_kernel(int nl, int me, int rec_off, const PlanScalars* __restrict__ ps, const int* __restrict__ mv_child,
                                   const int* __restrict__ mv_src, const int* __restrict__ mv_q, const int* __restrict__ pref,
                                   const int* __restrict__ new_gid, int* __restrict__ anc_bank, int* __restrict__ send_idx) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ps->M) return;
  const bool first = (j == 0 || mv_src[j] != mv_src[j - 1] || mv_q[j] != mv_q[j - 1]);
  if (mv_q[j] == me) {
    const int lo = ps->imp_start[me];
    const int base = lo > 0 ? pref[lo - 1] : 0;
    anc_bank[new_gid[mv_child[j]] - me * nl] = nl + rec_off + (pref[j] - 1 - base);
  }
  if (first && mv_src[j] / nl == me) {
    const int lo = ps->mv_off[me];
    const int base = lo > 0 ? pref[lo - 1] : 0;
    send_idx[pref[j] - 1 - base] = mv_src[j] - me * nl;
  }
}

hipError_t plan_run(const PlanBuffers& b, int N, int world, int nl, int me, const int* ai, const int* cur_gid,
                    int rec_off, hipStream_t s) {
  hipError_t e;
  if ((e = hipMemsetAsync(b.counts, 0, (size_t)(N + 1) * sizeof(int), s)) != hipSuccess) return e;
  const int nb = (N + 255) / 256;
  hipLaunchKernelGGL(plan_key_kernel, dim3(nb), dim3(256), 0, s, N, ai, cur_gid, b.key, b.counts);
  hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(kPlanThreads), 0, s, N, world, nl, b.counts, b.offsets, b.fill, b.scalars);
  hipLaunchKernelGGL(plan_scatter_kernel, dim3(nb), dim3(256), 0, s, N, b.key, b.offsets, b.fill, b.tmp);
  hipLaunchKernelGGL(plan_rank_kernel, dim3(nb), dim3(256), 0, s, N, b.key, b.offsets, b.counts, b.tmp, b.order);
  hipLaunchKernelGGL(plan_place_kernel, dim3(nb), dim3(256), 0, s, N, world, nl, me, b.key, b.order, b.scalars, b.new_gid,
                     b.slot_ids, b.anc_bank, b.mv_child, b.mv_src, b.mv_q);
  hipLaunchKernelGGL(plan_pairs_kernel, dim3(1), dim3(kPlanThreads), 0, s, world, me, b.scalars, b.mv_src, b.mv_q, b.pref,
                     b.counts_dev);
  // M is only known on the device: launch enough threads for the worst case (every child moves)
  hipLaunchKernelGGL(plan_finish_kernel, dim3(nb), dim3(256), 0, s, nl, me, rec_off, b.scalars, b.mv_child, b.mv_src, b.mv_q, b.pref,
                     b.new_gid, b.anc_bank, b.send_idx);
  return hipGetLastError();
}

}  // namespace rbpf
