// Multi-workgroup weight normalisation + resampling for large particle counts (the single-workgroup
// normalise_resample_kernel of rbpf_kernels.hip is latency-bound: 75 us at N = 8192 but 1.3 ms at N = 65536).
// Same semantics (particleFilter.m:154-161, tools/sample.m:30-32): every reduction and the prefix sum use a
// fixed two-level order (1024-element blocks, then the block partials in index order), so results are
// deterministic and identical on every rank; the running sum is again only an approximation of the strict
// left-to-right cumsum, certified per draw by search_kernel and repaired by resample_fixup_kernel.
#include "rbpf_internal.hpp"
#include "rbpf_device.hpp"

namespace rbpf {

constexpr int kRB = 1024;     // elements (= threads) per block

__device__ inline double blk_sum(double v, double* sred) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sred[wave] = v;
  __syncthreads();
  double s = sred[0];
  for (int w = 1; w < kRB / 64; ++w) s += sred[w];
  return s;
}

__device__ inline double blk_max(double v, double* sred) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if (lane == 0) sred[wave] = v;
  __syncthreads();
  double s = sred[0];
  for (int w = 1; w < kRB / 64; ++w) s = fmax(s, sred[w]);
  return s;
}

__global__ __launch_bounds__(kRB) void rs_max_kernel(int N, const double* __restrict__ logw, double* __restrict__ pmax) {
  __shared__ double sred[16];
  const int i = blockIdx.x * kRB + threadIdx.x;
  const double m = blk_max(i < N ? logw[i] : -INFINITY, sred);
  if (threadIdx.x == 0) pmax[blockIdx.x] = m;
}

__device__ inline double global_max(const double* pmax, int B, double* sred) {
  double m = -INFINITY;
  for (int b = threadIdx.x; b < B; b += kRB) m = fmax(m, pmax[b]);
  return blk_max(m, sred);
}

__global__ __launch_bounds__(kRB) void rs_sumexp_kernel(int N, int B, const double* __restrict__ logw,
                                                        const double* __restrict__ pmax, double* __restrict__ psum) {
  __shared__ double sred[16];
  const double c = global_max(pmax, B, sred);
  const int i = blockIdx.x * kRB + threadIdx.x;
  const double s = blk_sum(i < N ? exp(logw[i] - c) : 0.0, sred);
  if (threadIdx.x == 0) psum[blockIdx.x] = s;
}

// w, per-block first-maximum / weighted state sums / local inclusive prefix of w
__global__ __launch_bounds__(kRB) void rs_weights_kernel(NormArgs a, int B, const double* __restrict__ pmax,
                                                         const double* __restrict__ psum, double* __restrict__ btot,
                                                         double* __restrict__ bparts /* [B][10] */) {
  __shared__ double sred[16];
  __shared__ int sidx[16];
  __shared__ double sbc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, N = a.N;
  const double c = global_max(pmax, B, sred);
  if (tid == 0) { double t = 0.0; for (int b = 0; b < B; ++b) t += psum[b]; sbc = c + log(t); }   // fixed order
  __syncthreads();
  const double lse = sbc;
  const int i = blockIdx.x * kRB + tid;
  const double wi = (i < N) ? exp(a.logw[i] - lse) : 0.0;
  if (i < N) a.w[i] = wi;
  // first maximum of the block
  double bw = (i < N) ? wi : -1.0;
  int bi = (i < N) ? i : 0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ow = __shfl_xor(bw, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ow > bw || (ow == bw && oi < bi)) { bw = ow; bi = oi; }
  }
  __syncthreads();
  if (lane == 0) { sred[wave] = bw; sidx[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    double gw = sred[0]; int gi = sidx[0];
    for (int w = 1; w < kRB / 64; ++w) if (sred[w] > gw || (sred[w] == gw && sidx[w] < gi)) { gw = sred[w]; gi = sidx[w]; }
    bparts[blockIdx.x * 10 + 0] = gw;
    bparts[blockIdx.x * 10 + 1] = (double)gi;
  }
  for (int cix = 0; cix < a.nN; ++cix) {
    const double s = blk_sum((i < N) ? a.xn[(size_t)cix * N + i] * wi : 0.0, sred);
    if (tid == 0) bparts[blockIdx.x * 10 + 2 + cix] = s;
  }
  // local inclusive prefix (wave scan + wave offsets in order)
  double inc = wi;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  __syncthreads();
  if (lane == 63) sred[wave] = inc;
  __syncthreads();
  double woff = 0.0;
  for (int w = 0; w < wave; ++w) woff += sred[w];
  const double lp = woff + inc;
  if (i < N) a.wc[i] = lp;
  if (tid == kRB - 1) btot[blockIdx.x] = lp;
}

__global__ __launch_bounds__(kRB) void rs_offsets_kernel(NormArgs a, int B, const double* __restrict__ btot,
                                                         const double* __restrict__ bparts) {
  __shared__ double sb[1024];                           // block totals (B <= 1024)
  __shared__ double sp[16][10];
  __shared__ double soff;
  const int tid = threadIdx.x, N = a.N;
  for (int b = tid; b < B; b += kRB) sb[b] = btot[b];   // parallel loads, then an in-order sum from LDS
  __syncthreads();
  if (tid == 0) { double t = 0.0; for (int b = 0; b < (int)blockIdx.x; ++b) t += sb[b]; soff = t; }
  __syncthreads();
  const int i = blockIdx.x * kRB + tid;
  if (i < N && blockIdx.x > 0) a.wc[i] = soff + a.wc[i];
  if (blockIdx.x == 0) {
    // global first maximum and weighted state sums from the per-block partials, in block order
    if (tid < 2 + a.nN) {
      const int q = tid;
      if (q >= 2) {
        double t = 0.0;
        for (int b = 0; b < B; ++b) t += bparts[b * 10 + q];
        sp[0][q] = t;
      }
    }
    if (tid == 0) {
      double gw = bparts[0]; int gi = (int)bparts[1];
      for (int b = 1; b < B; ++b) { const double w = bparts[b * 10]; const int ix = (int)bparts[b * 10 + 1]; if (w > gw || (w == gw && ix < gi)) { gw = w; gi = ix; } }
      if (gi < 0 || gi >= N) gi = 0;
      *a.iw_max = gi;
      sp[1][0] = (double)gi;
    }
    __syncthreads();
    if (tid >= 2 && tid < 2 + a.nN) {
      const int cix = tid - 2, gi = (int)sp[1][0];
      if (a.traj_mean) a.traj_mean[cix] = sp[0][tid];
      if (a.traj_max) a.traj_max[cix] = a.xn[(size_t)cix * N + gi];
    }
  }
}

// counting sort of the slots by (remapped) ancestor: histogram, single-workgroup scan, scatter
__global__ void rs_hist_kernel(int N, const int* __restrict__ key, const int* __restrict__ remap, int* __restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) atomicAdd(&counts[remap ? remap[key[i]] : key[i]], 1);
}

__global__ __launch_bounds__(kRB) void rs_scan_kernel(int N, int* __restrict__ counts) {
  __shared__ int swave[kRB / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = (N + kRB - 1) / kRB;
  const int j0 = min(tid * S, N), j1 = min(j0 + S, N);
  constexpr int NBATCH = 8;                            // loads in flight per thread (a lone workgroup is latency-bound)
  int tot = 0;
  for (int jb = j0; jb < j1; jb += NBATCH) {
    int v[NBATCH];
#pragma unroll
    for (int k = 0; k < NBATCH; ++k) v[k] = counts[min(jb + k, N - 1)];
#pragma unroll
    for (int k = 0; k < NBATCH; ++k) tot += (jb + k < j1) ? v[k] : 0;
  }
  int inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
  if (lane == 63) swave[wave] = inc;
  __syncthreads();
  int run = inc - tot;
  for (int w = 0; w < wave; ++w) run += swave[w];
  for (int jb = j0; jb < j1; jb += NBATCH) {
    int v[NBATCH];
#pragma unroll
    for (int k = 0; k < NBATCH; ++k) v[k] = counts[min(jb + k, N - 1)];
#pragma unroll
    for (int k = 0; k < NBATCH; ++k)
      if (jb + k < j1) { counts[jb + k] = run; run += v[k]; }
  }
}

__global__ void rs_scatter_kernel(int N, const int* __restrict__ key, const int* __restrict__ remap, int* __restrict__ counts,
                                  int* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) order[atomicAdd(&counts[remap ? remap[key[i]] : key[i]], 1)] = i;
}

// scratch: doubles [pmax B | psum B | btot B | bparts 10 B], B = ceil(N/1024)
size_t resample_scratch_doubles(int N) { return (size_t)13 * ((N + kRB - 1) / kRB) + 16; }

hipError_t launch_resample_pipeline(const NormArgs& nm, const SearchArgs* sa, int* order, int* counts, const int* remap,
                                    double* scratch, hipStream_t s) {
  const int N = nm.N, B = (N + kRB - 1) / kRB;
  double* pmax = scratch; double* psum = pmax + B; double* btot = psum + B; double* bparts = btot + B;
  hipLaunchKernelGGL(rs_max_kernel, dim3(B), dim3(kRB), 0, s, N, nm.logw, pmax);
  hipLaunchKernelGGL(rs_sumexp_kernel, dim3(B), dim3(kRB), 0, s, N, B, nm.logw, pmax, psum);
  hipLaunchKernelGGL(rs_weights_kernel, dim3(B), dim3(kRB), 0, s, nm, B, pmax, psum, btot, bparts);
  hipLaunchKernelGGL(rs_offsets_kernel, dim3(B), dim3(kRB), 0, s, nm, B, btot, bparts);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !sa) return e;
  SearchArgs q = *sa;
  q.scan_depth = B + 32;
  if ((e = launch_search(q, s)) != hipSuccess) return e;
  if (q.approx && (e = launch_resample_fixup(q, s)) != hipSuccess) return e;
  if (order) {
    if ((e = hipMemsetAsync(counts, 0, (size_t)N * sizeof(int), s)) != hipSuccess) return e;
    const int nb = (q.n_draw + 255) / 256;
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nb), dim3(256), 0, s, q.n_draw, q.ai, remap, counts);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(kRB), 0, s, N, counts);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(nb), dim3(256), 0, s, q.n_draw, q.ai, remap, counts, order);
    e = hipGetLastError();
  }
  return e;
}

// counting sort of n_slots slots by remap[key[i]] over `range` key values, multi-workgroup (histogram, scan, scatter);
// counts: >= range ints of scratch
hipError_t launch_order_large(int n_slots, int range, const int* key, const int* remap, int* order, int* counts, hipStream_t s) {
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)range * sizeof(int), s);
  if (e != hipSuccess) return e;
  const int nb = (n_slots + 255) / 256;
  hipLaunchKernelGGL(rs_hist_kernel, dim3(nb), dim3(256), 0, s, n_slots, key, remap, counts);
  hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(kRB), 0, s, range, counts);
  hipLaunchKernelGGL(rs_scatter_kernel, dim3(nb), dim3(256), 0, s, n_slots, key, remap, counts, order);
  return hipGetLastError();
}

// ---- plan of a single-bank ("in place") flush ---------------------------------------------------------------
// Every stored matrix with children keeps its bank entry for its first child (in the ancestor-sorted processing
// order); the other children take the entries no child refers to, the k-th of them the k-th free entry.
__global__ void ip_mark_kernel(int N, const int* __restrict__ ai, const int* __restrict__ base, int* __restrict__ has,
                               int* __restrict__ nzp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { const int s = base[ai[i]]; has[s] = 1; nzp[s] = 1; }
}

__global__ void ip_free_kernel(int N, const int* __restrict__ has, const int* __restrict__ nzp, int* __restrict__ freelist) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f < N && !has[f]) freelist[f - nzp[f]] = f;
}

__global__ void ip_assign_kernel(int N, const int* __restrict__ order, const int* __restrict__ ai, const int* __restrict__ base,
                                 const int* __restrict__ nzp, const int* __restrict__ freelist, int* __restrict__ dst,
                                 int* __restrict__ phase) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= N) return;
  const int i = order[b];
  const int s = base[ai[i]];
  const int prev = (b > 0) ? base[ai[order[b - 1]]] : -1;
  const bool first = (s != prev);
  // children before position b that are not the first of their matrix: b - (#matrices with children below s) - 1
  dst[i] = first ? s : freelist[b - nzp[s] - 1];
  phase[i] = first ? 1 : 0;
}

// ---- shared flush (ping-pong banks): the children of one parent have the SAME prior covariance at a flush step (their lineage's
// stored matrix with the same pending sets applied), so only one of them -- the child with the smallest slot, the "writer" -- runs
// the flush variant and stores it; its siblings run the read-only variant of the step over the same source and take the writer's
// new entry as their stored matrix.  lead[j] = smallest child of parent j;  dst[i] = lead[ai[i]];  phase[i] = 1 for writers.
__global__ void share_fill_kernel(int N, int* lead) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) lead[i] = 0x7fffffff;
}
__global__ void share_min_kernel(int N, const int* __restrict__ ai, int* lead) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) atomicMin(&lead[ai[i]], i);
}
__global__ void share_assign_kernel(int N, const int* __restrict__ ai, const int* __restrict__ lead, int* __restrict__ dst,
                                    int* __restrict__ phase, unsigned long long* writers) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool w = i < N && lead[ai[i]] == i;
  if (i < N) { dst[i] = lead[ai[i]]; phase[i] = w ? 1 : 0; }
  const unsigned long long m = __ballot(w);
  if (writers && (threadIdx.x & 63) == 0 && m) atomicAdd(writers, (unsigned long long)__popcll(m));
}

// nkeys: range of the parent keys ai[.] (N on one GPU; bank entries + received records in the sharded filter)
hipError_t launch_share_plan(int N, int nkeys, const int* ai, int* lead, int* dst, int* phase, unsigned long long* writers, hipStream_t s) {
  const int nb = (N + 255) / 256;
  hipLaunchKernelGGL(share_fill_kernel, dim3((nkeys + 255) / 256), dim3(256), 0, s, nkeys, lead);
  hipLaunchKernelGGL(share_min_kernel, dim3(nb), dim3(256), 0, s, N, ai, lead);
  hipLaunchKernelGGL(share_assign_kernel, dim3(nb), dim3(256), 0, s, N, ai, lead, dst, phase, writers);
  return hipGetLastError();
}

// ---- shared flush in ONE bank (r05): the two plans above combined.  The children of one parent store ONE flushed matrix (its writer:
// the smallest child); the first writer, in processing order, of every stored matrix with children overwrites that matrix in place
// (phase 2: after everything that still reads it), the other writers take the entries no child refers to (phase 1), every other
// child runs the read-only variant over the old matrices (phase 0) and points at its writer's entry.
__global__ void sip_fill_kernel(int N, int* lead, int* leadbase, int* has, int* nzp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { lead[i] = 0x7fffffff; leadbase[i] = 0x7fffffff; has[i] = 0; nzp[i] = 0; }
}
__global__ void sip_mark_kernel(int N, const int* __restrict__ ai, const int* __restrict__ base, int* lead, int* __restrict__ has, int* __restrict__ nzp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { atomicMin(&lead[ai[i]], i); const int s = base[ai[i]]; has[s] = 1; nzp[s] = 1; }
}
__global__ void sip_first_kernel(int N, const int* __restrict__ order, const int* __restrict__ ai, const int* __restrict__ base,
                                 const int* __restrict__ lead, int* leadbase) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= N) return;
  const int i = order[b];
  if (lead[ai[i]] == i) atomicMin(&leadbase[base[ai[i]]], b);
}
__global__ void sip_flag_kernel(int N, const int* __restrict__ order, const int* __restrict__ ai, const int* __restrict__ base,
                                const int* __restrict__ lead, const int* __restrict__ leadbase, int* __restrict__ flag) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= N) return;
  const int i = order[b];
  flag[b] = (lead[ai[i]] == i && leadbase[base[ai[i]]] != b) ? 1 : 0;            // a writer that is not the first of its matrix
}
__global__ void sip_writers_kernel(int N, const int* __restrict__ order, const int* __restrict__ ai, const int* __restrict__ base,
                                   const int* __restrict__ lead, const int* __restrict__ leadbase, const int* __restrict__ rank,
                                   const int* __restrict__ freelist, int* __restrict__ dst, int* __restrict__ phase, unsigned long long* writers) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  bool w = false;
  if (b < N) {
    const int i = order[b];
    w = lead[ai[i]] == i;
    if (w) {
      const int s = base[ai[i]];
      const bool first = leadbase[s] == b;
      dst[i] = first ? s : freelist[rank[b]];
      phase[i] = first ? 2 : 1;
    }
  }
  const unsigned long long m = __ballot(w);
  if (writers && (threadIdx.x & 63) == 0 && m) atomicAdd(writers, (unsigned long long)__popcll(m));
}
__global__ void sip_readers_kernel(int N, const int* __restrict__ ai, const int* __restrict__ lead, int* __restrict__ dst, int* __restrict__ phase) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N && lead[ai[i]] != i) { dst[i] = dst[lead[ai[i]]]; phase[i] = 0; }
}

// scratch: 6 N ints
hipError_t launch_share_inplace_plan(int N, const int* order, const int* ai, const int* base, int* dst, int* phase, int* scratch,
                                     unsigned long long* writers, hipStream_t s) {
  int* lead = scratch; int* leadbase = scratch + N; int* has = scratch + 2 * (size_t)N; int* nzp = scratch + 3 * (size_t)N;
  int* freelist = scratch + 4 * (size_t)N; int* flag = scratch + 5 * (size_t)N;
  const int nb = (N + 255) / 256;
  hipLaunchKernelGGL(sip_fill_kernel, dim3(nb), dim3(256), 0, s, N, lead, leadbase, has, nzp);
  hipLaunchKernelGGL(sip_mark_kernel, dim3(nb), dim3(256), 0, s, N, ai, base, lead, has, nzp);
  hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(kRB), 0, s, N, nzp);          // exclusive prefix of the flags
  hipLaunchKernelGGL(ip_free_kernel, dim3(nb), dim3(256), 0, s, N, has, nzp, freelist);
  hipLaunchKernelGGL(sip_first_kernel, dim3(nb), dim3(256), 0, s, N, order, ai, base, lead, leadbase);
  hipLaunchKernelGGL(sip_flag_kernel, dim3(nb), dim3(256), 0, s, N, order, ai, base, lead, leadbase, flag);
  hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(kRB), 0, s, N, flag);         // rank among the writers that need a free entry
  hipLaunchKernelGGL(sip_writers_kernel, dim3(nb), dim3(256), 0, s, N, order, ai, base, lead, leadbase, flag, freelist, dst, phase, writers);
  hipLaunchKernelGGL(sip_readers_kernel, dim3(nb), dim3(256), 0, s, N, ai, lead, dst, phase);
  return hipGetLastError();
}

hipError_t launch_inplace_plan(int N, const int* order, const int* ai, const int* base, int* dst, int* phase, int* scratch,
                               hipStream_t s) {
  int* has = scratch; int* nzp = scratch + N; int* freelist = scratch + 2 * (size_t)N;
  hipError_t e = hipMemsetAsync(scratch, 0, (size_t)2 * N * sizeof(int), s);
  if (e != hipSuccess) return e;
  const int nb = (N + 255) / 256;
  hipLaunchKernelGGL(ip_mark_kernel, dim3(nb), dim3(256), 0, s, N, ai, base, has, nzp);
  hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(kRB), 0, s, N, nzp);          // exclusive prefix of the flags
  hipLaunchKernelGGL(ip_free_kernel, dim3(nb), dim3(256), 0, s, N, has, nzp, freelist);
  hipLaunchKernelGGL(ip_assign_kernel, dim3(nb), dim3(256), 0, s, N, order, ai, base, nzp, freelist, dst, phase);
  return hipGetLastError();
}

}  // namespace rbpf
