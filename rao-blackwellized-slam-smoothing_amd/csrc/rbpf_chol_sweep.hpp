// Carried ancestor-weight factors (rbpf_options.chol_refresh = K > 1; included by rbpf_smoother.hip inside namespace rbpf).
//
// particleSmootherInformationForm.m:224-236 factorises  A_i = Imat_i + ImatAddt  from scratch for every particle at every
// time step (n^3/3 flop each).  Along a lineage that matrix changes by a rank-2*ny term per step:
//     Imat_child = Imat_anc + H_child' R^-1 H_child           (:334, the child's own update)
//     ImatAddt  -= H_ref'  R^-1 H_ref                          (:194-201, the reference trajectory's term leaves the sum)
// and the right-hand side  ivec + ivecAddt  by  H_child' R^-1 y - H_ref' R^-1 y.  With K > 1 the factor is therefore
// CARRIED: every particle keeps  L = chol(A)  augmented by the row  z' = (L \ b)'  and each step applies ny rank-1 updates
// and ny rank-1 downdates to its ancestor's factor (LINPACK dchud / dchdd recurrences, one sweep over the columns,
// O(n^2) flop and one read + one write of the factor), the augmented row carrying the forward solve along.  Every K-th step
// the factor is recomputed from the exactly carried Imat by the factorisation kernels above ("refresh"), which bounds the
// drift.  This changes the ARITHMETIC of the ancestor weights (not the algebra): the weights agree with the fresh
// factorisation to the tolerance stated in DESIGN.md / tests/test_gpu_chol_carry.py; it is therefore an option, off by
// default.
//
// Sweep layout of a factor (per particle): rows 0..n-1 and the augmented row n, padded to NS slots of 64 rows; column k
// keeps the slots k/64 .. NS-1 only (the rows above its diagonal block are zero), slot q at row 64 q + lane:
//     off(k) = 64 * (64 * (b NS - b (b - 1) / 2) + r (NS - b)),   b = k / 64, r = k % 64       [doubles]
// Compact tail: when the last slot holds at most 8 rows (nLin = 515: the 3 border rows + the augmented row; nLin = 128: the
// augmented row) it is STORED as 8 doubles (one 64-byte sector, lanes 0..7) instead of 64 -- 18 % / 33 % less traffic; in
// registers it stays a slot like the others:
//     off(k) = 64 * (64 * (b M - b (b - 1) / 2) + r (M - b)) + 8 k   for k <= 64 M (M = NS - 1),   off(64 M) + 8 (k - 64 M) beyond
// One wave64 owns one particle: the column and the 2 ny vectors live in registers (lane = row within a slot), the pivots are
// broadcast with v_readlane, so there is no LDS and no barrier; the next column's loads are issued before the rotations of
// the current one.
#pragma once

constexpr int kSweepMaxSlots = 9;                    // rows n + 1 <= 576: nLin <= 575 (dense-mag m = 512 is 515)

__host__ __device__ inline int sweep_slots(int n) { return (n + 1 + 63) >> 6; }
__host__ __device__ inline int sweep_tail_compact(int n) { const int t = (n + 1) & 63; return (t >= 1 && t <= 8 && n + 1 > 64) ? 1 : 0; }
__host__ __device__ inline size_t sweep_col_offset(int k, int NS, int tailc) {
  if (!tailc) {
    const size_t b = (size_t)(k >> 6), r = (size_t)(k & 63);
    return 64 * (64 * (b * NS - b * (b - 1) / 2) + r * (NS - b));
  }
  const int M = NS - 1;
  const int kk = k < 64 * M ? k : 64 * M;
  const size_t b = (size_t)(kk >> 6), r = (size_t)(kk & 63);
  return 64 * (64 * (b * M - b * (b - 1) / 2) + r * (M - b)) + 8 * (size_t)k;
}
__host__ __device__ inline size_t sweep_factor_doubles(int n) { return sweep_col_offset(n, sweep_slots(n), sweep_tail_compact(n)); }

struct SweepArgs {
  int n, d, ldx, NS, N;
  int tailc;                        // compact tail: the last slot is stored as 8 doubles
  int ref_slot;                     // particle that carries the reference trajectory (its update and downdate cancel: copy), or -1
  const double* Lold; double* Lnew; size_t stride;      // factor banks (sweep layout), doubles per particle
  const int* anc;                   // [N] bank entry of each particle's ancestor (null: identity)
  const int* order;                 // [N] processing order (ancestor-sorted: siblings read their ancestor's factor together) or null
  const double* Hb;                 // [N][d][ldx] H of each particle's last update
  const double* Href;               // [d][n] H along the reference trajectory at the step that leaves the suffix sums
  const double* W;                  // [d x d] column-major whitening factor: W' W = R^-1 (W = inv(chol(R,'lower')))
  const double* yt;                 // [d] measurement of that step
  const double* qf; const double* hld;    // [N] ivec' P ivec and halfLogDetP after the last update
  double* pant_log;                 // += logwMeas
  int* status;
  // particle-sharded smoother: ancestors with a bank entry >= n_bank read their factor from a received particle record, and
  // the reference trajectory is the particle whose LOGICAL slot (slot_ids[p]) is ref_logical
  const double* rec = nullptr; size_t rec_stride = 0, rec_off = 0; int n_bank = 0;
  const int* slot_ids = nullptr; int ref_logical = -1;
};

// fragment-order factor of the 64-column kernel (row-tile major: ((rt * 4 RT + kg) * 64 + kk * 16 + r)) -> sweep layout
__global__ void sweep_from_chol64_kernel(int n, int NS, int tailc, const double* __restrict__ Lfrag, size_t frag_stride,
                                         double* __restrict__ Lsw, size_t sw_stride) {
  const int p = blockIdx.x;
  const int RT = (n + 1 + 15) >> 4;
  const size_t KGS = (size_t)4 * RT;
  const double* src = Lfrag + (size_t)p * frag_stride;
  double* dst = Lsw + (size_t)p * sw_stride;
  for (int k = blockIdx.y; k < n; k += gridDim.y) {
    const int b = k >> 6;
    double* col = dst + sweep_col_offset(k, NS, tailc);
    for (int e = threadIdx.x; e < (NS - b) * 64; e += blockDim.x) {
      const int q = b + (e >> 6), ln = e & 63, row = 64 * q + ln;
      if (tailc && q == NS - 1 && ln >= 8) continue;                    // the compact tail keeps lanes 0..7 only
      double v = 0.0;
      if (row >= k && row <= n) v = src[((size_t)(row >> 4) * KGS + (size_t)(k >> 2)) * 64 + (size_t)(k & 3) * 16 + (row & 15)];
      col[(size_t)(q - b) * 64 + ln] = v;
    }
  }
}

// The D downdate vectors of a particle: in registers, or -- VLDS, the large sizes, where 2 D NS doubles of vectors plus the
// column leave one wave per SIMD -- in LDS (lane-private slices, [a][q][lane]; the pivot read is a broadcast).  With the
// downdate vectors out of the register file two waves fit a SIMD and overlap each other's load latency.
template <int D, int NS, bool VLDS>
struct SweepV {
  double r[VLDS ? 1 : D][VLDS ? 1 : NS];
  double* s;                                            // LDS slice of this wave (VLDS)
  int lane;
  __device__ __forceinline__ double get(int a, int q) const { if constexpr (VLDS) return s[(a * NS + q) * 64 + lane]; else return r[a][q]; }
  __device__ __forceinline__ void set(int a, int q, double x) { if constexpr (VLDS) s[(a * NS + q) * 64 + lane] = x; else r[a][q] = x; }
  __device__ __forceinline__ double pivot(int a, int q, int r_lane) const {
    if constexpr (VLDS) return s[(a * NS + q) * 64 + r_lane]; else return readlane_f64(r[a][q], r_lane);
  }
};

// The 2 D rank-1 rotations of column k (D updates with u_a, then D downdates with v_a), slots B.. of the column:
//   r_a = sqrt(r_{a-1}^2 +- x_a(k)^2),  c_a = r_a / r_{a-1},  s_a = x_a(k) / r_{a-1}     (r_0 = L_kk)
//   L_ik = (L_ik +- s_a x_a(i)) / c_a;  x_a(i) = c_a x_a(i) - s_a L_ik
// A rotation changes only its own vector and the column, so the pivots x_a(k) of all 2 D vectors are read first and the
// squared pivots t_a = L_kk^2 +- ... are accumulated directly; the 2 D + 1 square roots are then independent of each other
// and are taken in ONE pass, lane a working on t_a (v_rsq_f64 + Newton, 13 instructions instead of 13 (2 D + 1)), the results
// broadcast with v_readlane.  The division by c_a is deferred: the column is kept unscaled, L = sigma_a * Lt with
// sigma_a = r_0 / r_a (the product of the 1 / c so far), so that a rotation costs three operations per element
//   Lt_i += (+- x_a(k) / r_0) x_a(i)         [the factor s_a / sigma_a]
//   x_a(i) = c_a x_a(i) - (s_a sigma_{a+1}) Lt_i
// and one multiplication by sigma_{2D} at the end.  Returns false when a downdate loses definiteness.
template <int D, int NS, int B, bool VLDS>
__device__ __forceinline__ bool sweep_rotate_all(double (&col)[NS], double (&u)[D][NS], SweepV<D, NS, VLDS>& v, int r_lane, int lane,
                                                 double& Lkk) {
  double xk[2 * D], t[2 * D + 1];
#pragma unroll
  for (int a = 0; a < D; ++a) { xk[a] = readlane_f64(u[a][B], r_lane); xk[D + a] = v.pivot(a, B, r_lane); }
  t[0] = Lkk * Lkk;
#pragma unroll
  for (int a = 0; a < 2 * D; ++a) t[a + 1] = fma(a < D ? xk[a] : -xk[a], xk[a], t[a]);
  bool ok = true;
#pragma unroll
  for (int a = 0; a <= 2 * D; ++a) ok = ok && (t[a] > 0.0);
  if (!ok) return false;
  double rt[2 * D + 1], ri[2 * D + 1];                 // sqrt(t_a), 1 / sqrt(t_a): lane a takes the roots of t_a
  {
    double tv = t[0];
#pragma unroll
    for (int a = 1; a <= 2 * D; ++a) tv = (lane == a) ? t[a] : tv;
    double g, h;
    sqrt_rsqrt(tv, g, h);
#pragma unroll
    for (int a = 0; a <= 2 * D; ++a) { rt[a] = readlane_f64(g, a); ri[a] = readlane_f64(h, a); }
  }
#pragma unroll
  for (int a = 0; a < 2 * D; ++a) {
    // sigma_a = rt[0] ri[a]; s_a / sigma_a = xk ri[a] / (rt[0] ri[a]) = xk ri[0]; c_a = rt[a+1] ri[a];
    // s_a sigma_{a+1} = xk ri[a] rt[0] ri[a+1]
    const double c = rt[a + 1] * ri[a];
    const double f = (a < D ? xk[a] : -xk[a]) * ri[0];
    const double g = -(xk[a] * ri[a]) * (rt[0] * ri[a + 1]);
    if (a < D) {
#pragma unroll
      for (int q = B; q < NS; ++q) {
        const double lq = fma(f, u[a][q], col[q]);
        u[a][q] = fma(g, lq, c * u[a][q]);
        col[q] = lq;
      }
    } else {
#pragma unroll
      for (int q = B; q < NS; ++q) {
        const double vq = v.get(a - D, q);
        const double lq = fma(f, vq, col[q]);
        v.set(a - D, q, fma(g, lq, c * vq));
        col[q] = lq;
      }
    }
  }
  const double sig = rt[0] * ri[2 * D];
#pragma unroll
  for (int q = B; q < NS; ++q) col[q] *= sig;
  // rows above the pivot inside the diagonal slot are structurally zero
  col[B] = (lane >= r_lane) ? col[B] : 0.0;
  Lkk = rt[2 * D];
  return true;
}

template <int D, int NS, int B, bool VLDS>
__device__ __forceinline__ bool sweep_block(const SweepArgs& a, const double* __restrict__ src, double* __restrict__ dst,
                                            double (&u)[D][NS], SweepV<D, NS, VLDS>& v, int lane, bool plain_copy,
                                            double& sumlog, double& vv) {
  const int k0 = 64 * B, k1 = min(a.n, k0 + 64);
  if (k0 >= a.n) return true;
  // sum of log L_kk (wave-uniform): the mantissas are multiplied up and one logarithm is taken per 16 columns
  double mant = 1.0;
  int expo = 0;
  double col[NS], nxt[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) { col[q] = 0.0; nxt[q] = 0.0; }
  const bool tailc = a.tailc != 0;
  // slot q of a column sits at (q - B) * 64 + lane; with the compact tail the last slot is 8 doubles, lanes 0..7
#define RBPF_SW_LOAD(dst_, base_)                                                                          \
  _Pragma("unroll") for (int q = B; q < NS; ++q) {                                                         \
    if (q == NS - 1 && tailc) dst_[q] = (lane < 8) ? (base_)[(size_t)(q - B) * 64 + lane] : 0.0;            \
    else dst_[q] = (base_)[(size_t)(q - B) * 64 + lane];                                                   \
  }
  // (two columns requested ahead instead of one -- a register move per slot -- measured r05, commit 31144f4: 4.36-4.44 against 4.38-4.56 ms,
  //  noise: the sweep is bound by the memory system, not by the depth of its prefetch)
  {
    const double* c0 = src + sweep_col_offset(k0, NS, a.tailc);
    RBPF_SW_LOAD(nxt, c0)
  }
  for (int k = k0; k < k1; ++k) {
    const int r_lane = k - k0;
#pragma unroll
    for (int q = B; q < NS; ++q) col[q] = nxt[q];
    if (k + 1 < k1) {                                     // the next column of this block (same slot range)
      const double* cn = src + sweep_col_offset(k + 1, NS, a.tailc);
      RBPF_SW_LOAD(nxt, cn)
    }

    double Lkk = readlane_f64(col[B], r_lane);
    if (!plain_copy) {
      if (!sweep_rotate_all<D, NS, B, VLDS>(col, u, v, r_lane, lane, Lkk)) return false;
    }
    if (!(Lkk > 0.0)) return false;
    double* cd = dst + sweep_col_offset(k, NS, a.tailc);
#pragma unroll
    for (int q = B; q < NS; ++q) {
      if (q == NS - 1 && tailc) { if (lane < 8) __builtin_nontemporal_store(col[q], &cd[(size_t)(q - B) * 64 + lane]); }
      else __builtin_nontemporal_store(col[q], &cd[(size_t)(q - B) * 64 + lane]);
    }
    mant *= __builtin_amdgcn_frexp_mant(Lkk);
    expo += __builtin_amdgcn_frexp_exp(Lkk);
    if ((r_lane & 15) == 15) { sumlog += log(mant); mant = 1.0; }
    if (lane == (a.n & 63)) vv = fma(col[NS - 1], col[NS - 1], vv);    // z_k = L(n, k): the augmented row sits in the last slot
  }
#undef RBPF_SW_LOAD
  sumlog += log(mant) + 0.6931471805599453094 * (double)expo;
  return true;
}

template <int D, int NS, bool VLDS>
__global__ __launch_bounds__(256, VLDS ? 2 : 1) void chol_sweep_kernel(const SweepArgs a) {
  extern __shared__ double sweep_lds[];
  const int lane = threadIdx.x & 63;
  // (processing positions dealt to the XCDs in contiguous ranges: siblings, side by side in `order`, read their ancestor's factor
  //  out of one L2 -- xcd_position, rbpf_internal.hpp)
  const int pb = xcd_position((int)blockIdx.x, (int)gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pb >= a.N) return;                                 // wave-uniform
  const int p = a.order ? a.order[pb] : pb;
  const int src_p = a.anc ? a.anc[p] : p;
  const double* src = (a.rec && src_p >= a.n_bank) ? a.rec + (size_t)(src_p - a.n_bank) * a.rec_stride + a.rec_off
                                                   : a.Lold + (size_t)src_p * a.stride;
  double* dst = a.Lnew + (size_t)p * a.stride;
  const bool plain_copy = a.slot_ids ? (a.slot_ids[p] == a.ref_logical) : (p == a.ref_slot);
  // update vectors u_a = (W H_p)_a, downdate vectors v_a = (W H_ref)_a, both augmented by eta_a = (W y)_a in row n
  double u[D][NS];
  SweepV<D, NS, VLDS> v;
  v.lane = lane;
  v.s = VLDS ? sweep_lds + (size_t)(threadIdx.x >> 6) * D * NS * 64 : nullptr;
  {
    double Wm[D * D], eta[D];
#pragma unroll
    for (int q = 0; q < D * D; ++q) Wm[q] = a.W[q];
#pragma unroll
    for (int aa = 0; aa < D; ++aa) { double s = 0.0; for (int bb = 0; bb < D; ++bb) s = fma(Wm[aa + D * bb], a.yt[bb], s); eta[aa] = s; }
    const double* H = a.Hb + (size_t)p * D * a.ldx;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const int row = 64 * q + lane;
      const int rc = min(row, a.n - 1);
      double h[D], hr[D];
#pragma unroll
      for (int bb = 0; bb < D; ++bb) { h[bb] = H[(size_t)bb * a.ldx + rc]; hr[bb] = a.Href[(size_t)bb * a.n + rc]; }
#pragma unroll
      for (int aa = 0; aa < D; ++aa) {
        double su = 0.0, sv = 0.0;
#pragma unroll
        for (int bb = 0; bb < D; ++bb) { su = fma(Wm[aa + D * bb], h[bb], su); sv = fma(Wm[aa + D * bb], hr[bb], sv); }
        u[aa][q] = (row < a.n) ? su : (row == a.n ? eta[aa] : 0.0);
        v.set(aa, q, (row < a.n) ? sv : (row == a.n ? eta[aa] : 0.0));
      }
    }
  }
  double sumlog = 0.0, vv = 0.0;
  bool ok = true;
#define RBPF_SWEEP_BLOCK(B_) if constexpr (NS > B_) { if (ok) ok = sweep_block<D, NS, B_, VLDS>(a, src, dst, u, v, lane, plain_copy, sumlog, vv); }
  RBPF_SWEEP_BLOCK(0) RBPF_SWEEP_BLOCK(1) RBPF_SWEEP_BLOCK(2) RBPF_SWEEP_BLOCK(3) RBPF_SWEEP_BLOCK(4)
  RBPF_SWEEP_BLOCK(5) RBPF_SWEEP_BLOCK(6) RBPF_SWEEP_BLOCK(7) RBPF_SWEEP_BLOCK(8)
#undef RBPF_SWEEP_BLOCK
  vv = wave_sum(vv);                                     // (sumlog is wave-uniform)
  if (lane == 0) {
    if (ok) a.pant_log[p] += -0.5 * a.qf[p] - a.hld[p] - sumlog + 0.5 * vv;        // InformationForm.m:234-236
    else { atomicOr(a.status, 2); a.pant_log[p] = nan(""); }
  }
}

template <int D>
static hipError_t launch_chol_sweep_d(const SweepArgs& a, hipStream_t st) {
  const dim3 grid((a.N + 3) / 4), block(256);
  switch (a.NS) {
  // downdate vectors in LDS where D * NS is large (two waves per SIMD instead of one): 4 waves x D x NS x 512 B per workgroup
#define RBPF_SW(NS_) case NS_:                                                                                          \
    if constexpr (D * NS_ >= 21) hipLaunchKernelGGL((chol_sweep_kernel<D, NS_, true>), grid, block, (size_t)4 * D * NS_ * 64 * sizeof(double), st, a); \
    else hipLaunchKernelGGL((chol_sweep_kernel<D, NS_, false>), grid, block, 0, st, a);                                    \
    break;
    RBPF_SW(1) RBPF_SW(2) RBPF_SW(3) RBPF_SW(4) RBPF_SW(5) RBPF_SW(6) RBPF_SW(7) RBPF_SW(8) RBPF_SW(9)
#undef RBPF_SW
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

static hipError_t launch_chol_sweep(const SweepArgs& a, hipStream_t st) {
  if (a.d == 3) return launch_chol_sweep_d<3>(a, st);
  if (a.d == 1) return launch_chol_sweep_d<1>(a, st);
  return hipErrorInvalidValue;
}

static hipError_t launch_sweep_from_chol64(int n, int batch, const double* Lfrag, size_t frag_stride, double* Lsw, size_t sw_stride,
                                           hipStream_t st) {
  hipLaunchKernelGGL(sweep_from_chol64_kernel, dim3(batch, 8), dim3(256), 0, st, n, sweep_slots(n), sweep_tail_compact(n), Lfrag, frag_stride,
                     Lsw, sw_stride);
  return hipGetLastError();
}

// ---- refresh from the state history ------------------------------------------------------------------------------
// Between two refreshes nothing reads Imat, so it is not advanced at all: at a refresh step the information matrix of every
// particle is rebuilt from the last materialised generation t0 ("base") and the measurement Jacobians along its ancestral
// path,  Imat_i(t-1) = Imat_base(anc_{t0}(i)) + sum_{s = t0+1}^{t-1} H(x_{path_i(s), s})' R^-1 H(...),  the H recomputed from
// the state history (basis evaluation is cheap) and the sum formed as G' G on the matrix cores, G = [W H_s] stacked (block-lower
// tiles only, the base matrix added in the GEMM's epilogue).
// Exactly the terms :334 adds step by step, in another summation order.

// thread per particle: walk the ancestor table back from generation t_last to t0 + 1, collecting the states on the way
__global__ void sweep_path_kernel(int N, int nN, int Kp, int t_last, int t0, const int* __restrict__ A, const double* __restrict__ X,
                                  int* __restrict__ base_slot, double* __restrict__ Xp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int slot = i;
  for (int s = t_last; s > t0; --s) {
    const double* Xs = X + (size_t)s * nN * N;
    double* col = Xp + ((size_t)i * Kp + (s - t0 - 1)) * nN;
    for (int c = 0; c < nN; ++c) col[c] = Xs[(size_t)c * N + slot];
    if (s > 0) slot = A[(size_t)s * N + slot];          // ancestor in generation s - 1
  }
  base_slot[i] = slot;
}

// Refresh without stored information matrices (rbpf_options.info_rebuild): the whole ancestral path, in segments of S generations
// counted from the top (segment j = generations [max(0, hi_j - S), hi_j), hi_j = t_last + 1 - j S).  One walk notes where every
// particle's lineage stands at the top generation of each segment ...
__global__ void origin_marks_kernel(int N, int t_last, int S, const int* __restrict__ A, int* __restrict__ marks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int slot = i;
  for (int s = t_last; s >= 0; --s) {
    const int below_top = t_last - s;                        // generations under the top one
    if (below_top % S == 0) marks[(size_t)(below_top / S) * N + i] = slot;
    if (s > 0) slot = A[(size_t)s * N + slot];               // ancestor in generation s - 1
  }
}

// ... and a segment's states are then collected from its mark: particles p0 .. p0 + cnt - 1, generations lo .. hi - 1 ->
// Xp [(i - p0) * (hi - lo) + (s - lo)][nN]
__global__ void origin_segment_kernel(int N, int nN, int p0, int cnt, int lo, int hi, const int* __restrict__ A, const double* __restrict__ X,
                                      const int* __restrict__ mark, double* __restrict__ Xp) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= cnt) return;
  int slot = mark[p0 + q];                                   // the lineage at generation hi - 1
  for (int s = hi - 1; s >= lo; --s) {
    const double* Xs = X + (size_t)s * nN * N;
    double* col = Xp + ((size_t)q * (hi - lo) + (s - lo)) * nN;
    for (int c = 0; c < nN; ++c) col[c] = Xs[(size_t)c * N + slot];
    if (s > 0) slot = A[(size_t)s * N + slot];
  }
}

// Particle-sharded smoother: the same walk over the replicated global history (logical slot ids), for every logical slot j:
// owner_now[j] = where its particle lives now (rank * Nloc + physical slot), base_loc[j] = where the particle of its ancestor
// in generation t0 lived when that generation was materialised (-1: t0 < 0, the common initial matrix).  The path states are
// written for the particles of this rank only, at their physical slot.
__global__ void shard_path_kernel(int Nglob, int Nloc, int rank, int nN, int Kp, int t_last, int t0, const int* __restrict__ A,
                                  const double* __restrict__ X, const int* __restrict__ cur_gid, const int* __restrict__ base_gid,
                                  int* __restrict__ owner_now, int* __restrict__ base_loc, double* __restrict__ Xp) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Nglob) return;
  const int gid = cur_gid ? cur_gid[j] : j;
  const bool mine = (gid / Nloc) == rank;
  const int p = gid % Nloc;
  int slot = j;
  for (int s = t_last; s > t0; --s) {
    if (mine) {
      const double* Xs = X + (size_t)s * nN * Nglob;
      double* col = Xp + ((size_t)p * Kp + (s - t0 - 1)) * nN;
      for (int c = 0; c < nN; ++c) col[c] = Xs[(size_t)c * Nglob + slot];
    }
    if (s > 0) slot = A[(size_t)s * Nglob + slot];
  }
  owner_now[j] = gid;
  base_loc[j] = (t0 < 0) ? -1 : (base_gid ? base_gid[slot] : slot);
}

// rows of `count` matrices [n x n] out of a bank into a contiguous buffer (the base matrices other ranks asked for)
__global__ __launch_bounds__(256) void gather_matrices_kernel(size_t nn, const int* __restrict__ idx, const double* __restrict__ bank,
                                                              double* __restrict__ out) {
  const double* src = bank + (size_t)idx[blockIdx.x] * nn;
  double* dst = out + (size_t)blockIdx.x * nn;
  const size_t per = (nn + gridDim.y - 1) / gridDim.y, q0 = (size_t)blockIdx.y * per, q1 = q0 + per < nn ? q0 + per : nn;
  for (size_t q = q0 + threadIdx.x; q < q1; q += blockDim.x) dst[q] = src[q];
}

// G = W H in place: H [rows][d][n] (rows of H contiguous), W [d x d] column-major
__global__ void sweep_whiten_kernel(size_t rows, int d, int n, const double* __restrict__ W, double* __restrict__ H) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= rows * n) return;
  const size_t row = q / n; const int c = (int)(q % n);
  double* h = H + row * d * n + c;
  double v[3], o[3];
  for (int b = 0; b < d; ++b) v[b] = h[(size_t)b * n];
  for (int a = 0; a < d; ++a) { double s = 0.0; for (int b = 0; b < d; ++b) s = fma(W[a + d * b], v[b], s); o[a] = s; }
  for (int a = 0; a < d; ++a) h[(size_t)a * n] = o[a];
}

// Imat(:,:,i) of the new generation = ancestor's stored matrix + the particle's own last update (InformationForm.m:170,334),
// on its own (the steps between two refreshes: the factorisation kernels, which otherwise do this while loading, do not run).
// Covers the block-lower part the factorisation kernels read (rows >= 64 * (column / 64)).
__global__ __launch_bounds__(256) void imat_gather_kernel(int n, int d, int ldx, const double* __restrict__ Imat, long imat_stride,
                                                          const int* __restrict__ anc, const double* __restrict__ Hb,
                                                          const double* __restrict__ Rinv, double* __restrict__ ImatOut) {
  extern __shared__ double gsm[];
  const int p = blockIdx.x;
  double* Hs = gsm;                       // [d][n]
  double* RH = gsm + (size_t)d * n;       // [d][n]  R^-1 H
  const double* H = Hb + (size_t)p * d * ldx;
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    for (int aa = 0; aa < d; ++aa) {
      double t = 0.0;
      for (int bb = 0; bb < d; ++bb) t = fma(Rinv[aa + d * bb], H[(size_t)bb * ldx + i], t);
      Hs[aa * n + i] = H[(size_t)aa * ldx + i];
      RH[aa * n + i] = t;
    }
  __syncthreads();
  const double* src = Imat + (size_t)(anc ? anc[p] : p) * imat_stride;
  double* dst = ImatOut + (size_t)p * n * n;
  for (int c = blockIdx.y; c < n; c += gridDim.y) {
    const int r0 = (c >> 6) << 6;
    for (int r = r0 + threadIdx.x; r < n; r += blockDim.x) {
      double v = src[(size_t)r + (size_t)n * c];
      for (int aa = 0; aa < d; ++aa) v = fma(Hs[aa * n + r], RH[aa * n + c], v);
      __builtin_nontemporal_store(v, &dst[(size_t)r + (size_t)n * c]);
    }
  }
}

static hipError_t launch_imat_gather(int n, int d, int ldx, int batch, const double* Imat, long imat_stride, const int* anc,
                                     const double* Hb, const double* Rinv, double* ImatOut, hipStream_t st) {
  hipLaunchKernelGGL(imat_gather_kernel, dim3(batch, 4), dim3(256), (size_t)2 * d * n * sizeof(double), st, n, d, ldx, Imat, imat_stride,
                     anc, Hb, Rinv, ImatOut);
  return hipGetLastError();
}
