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
// One wave64 owns one particle: the column and the 2 ny vectors live in registers (lane = row within a slot), the pivots are
// broadcast with v_readlane, so there is no LDS and no barrier; the next column's loads are issued before the rotations of
// the current one.
#pragma once

constexpr int kSweepMaxSlots = 9;                    // rows n + 1 <= 576: nLin <= 575 (dense-mag m = 512 is 515)

__host__ __device__ inline size_t sweep_col_offset(int k, int NS) {
  const size_t b = (size_t)(k >> 6), r = (size_t)(k & 63);
  return 64 * (64 * (b * NS - b * (b - 1) / 2) + r * (NS - b));
}
__host__ __device__ inline int sweep_slots(int n) { return (n + 1 + 63) >> 6; }
__host__ __device__ inline size_t sweep_factor_doubles(int n) { return sweep_col_offset(n, sweep_slots(n)); }

struct SweepArgs {
  int n, d, ldx, NS, N;
  int ref_slot;                     // particle that carries the reference trajectory (its update and downdate cancel: copy), or -1
  const double* Lold; double* Lnew; size_t stride;      // factor banks (sweep layout), doubles per particle
  const int* anc;                   // [N] bank entry of each particle's ancestor (null: identity)
  const double* Hb;                 // [N][d][ldx] H of each particle's last update
  const double* Href;               // [d][n] H along the reference trajectory at the step that leaves the suffix sums
  const double* W;                  // [d x d] column-major whitening factor: W' W = R^-1 (W = inv(chol(R,'lower')))
  const double* yt;                 // [d] measurement of that step
  const double* qf; const double* hld;    // [N] ivec' P ivec and halfLogDetP after the last update
  double* pant_log;                 // += logwMeas
  int* status;
};

// fragment-order factor of the 64-column kernel (row-tile major: ((rt * 4 RT + kg) * 64 + kk * 16 + r)) -> sweep layout
__global__ void sweep_from_chol64_kernel(int n, int NS, const double* __restrict__ Lfrag, size_t frag_stride,
                                         double* __restrict__ Lsw, size_t sw_stride) {
  const int p = blockIdx.x;
  const int RT = (n + 1 + 15) >> 4;
  const size_t KGS = (size_t)4 * RT;
  const double* src = Lfrag + (size_t)p * frag_stride;
  double* dst = Lsw + (size_t)p * sw_stride;
  for (int k = blockIdx.y; k < n; k += gridDim.y) {
    const int b = k >> 6;
    double* col = dst + sweep_col_offset(k, NS);
    for (int e = threadIdx.x; e < (NS - b) * 64; e += blockDim.x) {
      const int row = 64 * b + e;
      double v = 0.0;
      if (row >= k && row <= n) v = src[((size_t)(row >> 4) * KGS + (size_t)(k >> 2)) * 64 + (size_t)(k & 3) * 16 + (row & 15)];
      col[e] = v;
    }
  }
}

// One rank-1 rotation of column k against vector x (SIGN = +1 update, -1 downdate), slots B.. of the column.
//   r = sqrt(Lkk^2 +- xk^2); c = r / Lkk; s = xk / Lkk;  L_ik = (L_ik +- s x_i) / c;  x_i = c x_i - s L_ik
// linv = 1 / Lkk on entry, 1 / r on exit (the next rotation's pivot is r).  Returns false when a downdate loses definiteness.
template <int NS, int B, int SIGN>
__device__ __forceinline__ bool sweep_rotate(double (&col)[NS], double (&x)[NS], int r_lane, int lane, double& Lkk, double& linv) {
  const double xk = readlane_f64(x[B], r_lane);
  const double t = fma((double)SIGN * xk, xk, Lkk * Lkk);
  if (!(t > 0.0)) return false;
  double rr, rinv;
  sqrt_rsqrt(t, rr, rinv);
  const double c = rr * linv, s = xk * linv, cinv = Lkk * rinv;
#pragma unroll
  for (int q = B; q < NS; ++q) {
    const double lq = fma((double)SIGN * s, x[q], col[q]) * cinv;
    x[q] = fma(c, x[q], -s * lq);
    col[q] = lq;
  }
  // rows above the pivot inside the diagonal slot are structurally zero
  col[B] = (lane >= r_lane) ? col[B] : 0.0;
  Lkk = rr; linv = rinv;
  return true;
}

template <int D, int NS, int B>
__device__ __forceinline__ bool sweep_block(const SweepArgs& a, const double* __restrict__ src, double* __restrict__ dst,
                                            double (&u)[D][NS], double (&v)[D][NS], int lane, bool plain_copy,
                                            double& sumlog, double& vv) {
  const int k0 = 64 * B, k1 = min(a.n, k0 + 64);
  if (k0 >= a.n) return true;
  double col[NS], nxt[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) { col[q] = 0.0; nxt[q] = 0.0; }
  {
    const double* c0 = src + sweep_col_offset(k0, NS);
#pragma unroll
    for (int q = B; q < NS; ++q) nxt[q] = c0[(size_t)(q - B) * 64 + lane];
  }
  for (int k = k0; k < k1; ++k) {
    const int r_lane = k - k0;
#pragma unroll
    for (int q = B; q < NS; ++q) col[q] = nxt[q];
    if (k + 1 < k1) {                                     // the next column of this block (same slot range)
      const double* cn = src + sweep_col_offset(k + 1, NS);
#pragma unroll
      for (int q = B; q < NS; ++q) nxt[q] = cn[(size_t)(q - B) * 64 + lane];
    }
    double Lkk = readlane_f64(col[B], r_lane);
    if (!plain_copy) {
      double linv = 1.0 / Lkk;
#pragma unroll
      for (int aa = 0; aa < D; ++aa)
        if (!sweep_rotate<NS, B, 1>(col, u[aa], r_lane, lane, Lkk, linv)) return false;
#pragma unroll
      for (int aa = 0; aa < D; ++aa)
        if (!sweep_rotate<NS, B, -1>(col, v[aa], r_lane, lane, Lkk, linv)) return false;
    }
    if (!(Lkk > 0.0)) return false;
    double* cd = dst + sweep_col_offset(k, NS);
#pragma unroll
    for (int q = B; q < NS; ++q) __builtin_nontemporal_store(col[q], &cd[(size_t)(q - B) * 64 + lane]);
    if (lane == r_lane) sumlog += log(Lkk);
    if (lane == (a.n & 63)) vv = fma(col[NS - 1], col[NS - 1], vv);    // z_k = L(n, k): the augmented row sits in the last slot
  }
  return true;
}

template <int D, int NS>
__global__ __launch_bounds__(256) void chol_sweep_kernel(const SweepArgs a) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= a.N) return;                                  // wave-uniform
  const int src_p = a.anc ? a.anc[p] : p;
  const double* src = a.Lold + (size_t)src_p * a.stride;
  double* dst = a.Lnew + (size_t)p * a.stride;
  const bool plain_copy = (p == a.ref_slot);
  // update vectors u_a = (W H_p)_a, downdate vectors v_a = (W H_ref)_a, both augmented by eta_a = (W y)_a in row n
  double u[D][NS], v[D][NS];
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
        v[aa][q] = (row < a.n) ? sv : (row == a.n ? eta[aa] : 0.0);
      }
    }
  }
  double sumlog = 0.0, vv = 0.0;
  bool ok = true;
#define RBPF_SWEEP_BLOCK(B_) if constexpr (NS > B_) { if (ok) ok = sweep_block<D, NS, B_>(a, src, dst, u, v, lane, plain_copy, sumlog, vv); }
  RBPF_SWEEP_BLOCK(0) RBPF_SWEEP_BLOCK(1) RBPF_SWEEP_BLOCK(2) RBPF_SWEEP_BLOCK(3) RBPF_SWEEP_BLOCK(4)
  RBPF_SWEEP_BLOCK(5) RBPF_SWEEP_BLOCK(6) RBPF_SWEEP_BLOCK(7) RBPF_SWEEP_BLOCK(8)
#undef RBPF_SWEEP_BLOCK
  sumlog = wave_sum(sumlog);
  vv = wave_sum(vv);
  if (lane == 0) {
    if (ok) a.pant_log[p] += -0.5 * a.qf[p] - a.hld[p] - sumlog + 0.5 * vv;        // InformationForm.m:234-236
    else { atomicOr(a.status, 2); a.pant_log[p] = nan(""); }
  }
}

template <int D>
static hipError_t launch_chol_sweep_d(const SweepArgs& a, hipStream_t st) {
  const dim3 grid((a.N + 3) / 4), block(256);
  switch (a.NS) {
#define RBPF_SW(NS_) case NS_: hipLaunchKernelGGL((chol_sweep_kernel<D, NS_>), grid, block, 0, st, a); break;
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
  hipLaunchKernelGGL(sweep_from_chol64_kernel, dim3(batch, 8), dim3(256), 0, st, n, sweep_slots(n), Lfrag, frag_stride, Lsw, sw_stride);
  return hipGetLastError();
}

// ---- refresh from the state history ------------------------------------------------------------------------------
// Between two refreshes nothing reads Imat, so it is not advanced at all: at a refresh step the information matrix of every
// particle is rebuilt from the last materialised generation t0 ("base") and the measurement Jacobians along its ancestral
// path,  Imat_i(t-1) = Imat_base(anc_{t0}(i)) + sum_{s = t0+1}^{t-1} H(x_{path_i(s), s})' R^-1 H(...),  the H recomputed from
// the state history (basis evaluation is cheap) and the sum formed as G' G on the matrix cores, G = [W H_s] stacked.
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

// Imat_new[i] += base[base_slot[i]] over the block-lower part the factorisation kernels read
__global__ __launch_bounds__(256) void sweep_add_base_kernel(int n, const double* __restrict__ base, long base_stride,
                                                             const int* __restrict__ base_slot, double* __restrict__ Imat) {
  const int p = blockIdx.x;
  const double* src = base + (size_t)(base_slot ? base_slot[p] : 0) * base_stride;
  double* dst = Imat + (size_t)p * n * n;
  for (int c = blockIdx.y; c < n; c += gridDim.y) {
    const int r0 = (c >> 6) << 6;
    for (int r = r0 + threadIdx.x; r < n; r += blockDim.x) dst[(size_t)r + (size_t)n * c] += src[(size_t)r + (size_t)n * c];
  }
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
