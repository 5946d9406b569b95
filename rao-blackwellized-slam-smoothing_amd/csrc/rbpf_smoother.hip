// Conditional particle filter with ancestor sampling (CPF-AS), iterated N_K times:
//   src/particleSmoother.m                 (covariance-form ancestor weights, :156-245)
//   src/particleSmootherInformationForm.m  (information-form weights, :194-254, :279-335)
// Every time step re-uses the filter's fused step kernel (rbpf_kernels.hip); this file adds the
// reference-trajectory machinery: ancestor weights of slot N_P, the extra information-form state
// (Imat, ivec, halfLogDetP) and the per-iteration trajectory draw.
#include "../../include/rbpf.h"
#include "rbpf_internal.hpp"
#include "rbpf_ctx.hpp"
#include "rbpf_device.hpp"
#include "rbpf_shard_state.hpp"
#include "rbpf_sparse.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace rbpf {

struct SmootherState {
  double* d_xnk = nullptr;      // [T][nN] reference trajectory of this iteration (row per step)
  double* d_dyref = nullptr;    // [T][d][n] H along the reference trajectory (:120)
  double* d_pant_log = nullptr; // [N]
  double* d_pant = nullptr;     // [N] (or [T][N] when tracing)
  double* d_wc2 = nullptr;      // [N]
  double* d_Pfull = nullptr;    // cov form: [N][n*n] flushed covariances
  double* d_G = nullptr;        // cov form: [N][Mmax*n]
  double* d_S = nullptr;        // [N][Mmax*Mmax] (cov) / unused (info)
  double* d_L = nullptr;        // [N][Mmax*Mmax] or [N][n*n] Cholesky factors
  double* d_e = nullptr;        // [N][Mmax]
  // information form
  double* d_Imat[2] = {nullptr, nullptr};   // [N][n*n] column-major, ping-pong
  double* d_Hb[2] = {nullptr, nullptr};     // [N][d][ldx] H used by the last update (pending Imat term)
  double* d_ivec[2] = {nullptr, nullptr};   // [N][ldx]
  double* d_hld[2] = {nullptr, nullptr};    // [N]
  double* d_qf[2] = {nullptr, nullptr};     // [N] ivec' P ivec after the last update
  double* d_ImatAdd = nullptr;  // [n*n]
  double* d_ivecAdd = nullptr;  // [n]
  double* d_Imat0 = nullptr;    // [n*n] diag(1./diag(P0))
  int* d_ak = nullptr;
  double* d_ivec0 = nullptr;    // [ldx] initial information vector (:110)
  double* d_hld0 = nullptr;     // [1]
  // carried ancestor-weight factors (rbpf_options.chol_refresh > 1)
  double* d_Lsw[2] = {nullptr, nullptr};    // [N][sweep_factor_doubles(n)] factor banks in sweep layout, ping-pong
  double* d_W = nullptr;        // [d x d] whitening factor: W' W = R^-1
  int sw_cur = 0;
  int refresh = 0;              // K (0 / 1: off)
  bool lazy_imat = false;       // Imat is rebuilt from the state history at the refreshes only (recognised families)
  int base_gen = -1;            // generation the Imat bank holds (-1: Imat0)
  int* d_base_slot = nullptr;   // [N]
  double* d_Xp = nullptr;       // [N * K][nN] states along the ancestral paths (d_G then holds [N * K][d][n] whitened Jacobians)
  int icur = 0;                 // ping-pong index of ivec / hld / qf / Hb
  int imat_cur = 0;
  bool imat_valid = false;      // false until the first gather of an iteration (Imat = Imat0)
  bool imat_packed = false;     // the banks, Imat0 and ImatAdd in packed block-lower storage (imat_packed_index)
  size_t imat_len = 0;          // doubles per information matrix of the banks: n * n, or imat_packed_doubles(n)
  size_t Mmax = 0;
  int l_chunk = 0;              // carried factors: matrices per launch of the refresh factorisation (d_L holds that many workspaces)
  bool refresh_free = false;    // no information matrix is stored (rbpf_options.info_rebuild, or chol_refresh >= N_T - 1): a refresh rebuilds them
                                // from Imat0 along the whole ancestral path, chunk by chunk -- the Imat "bank" is one chunk of l_chunk matrices
  int seg_len = 0;              // ... in segments of this many generations (d_G / d_Xp hold l_chunk x seg_len rows)
  int* d_marks = nullptr;       // [ceil(T / seg_len)][N] slot of every particle's ancestor at the top generation of each segment
  // sharded information-form smoother
  double* d_Rinv = nullptr;     // [d*d]
  std::vector<double> h_ivec0;  // [ldx]
  double hld0 = 0.0;
  // sharded smoother with carried factors: where every logical slot's particle lives now / lived when the Imat bank was
  // materialised, and the base matrices fetched from (packed for) other ranks at a refresh
  int* d_owner_now = nullptr;   // [Nglob] rank * Nloc + physical slot
  int* d_base_gid = nullptr;    // [Nglob] the same table of generation base_gen
  int* d_base_loc = nullptr;    // [Nglob] location of each logical slot's ancestor of generation base_gen at that time (-1: Imat0)
  double* d_rf_send = nullptr;  // [rf_cap][n*n]
  double* d_rf_recv = nullptr;  // [rf_cap][n*n]
  int* d_rf_idx = nullptr;      // [rf_cap]
  size_t rf_cap = 0;
  int rf_stage = 0;             // 1 between refresh_begin and refresh_end
  int rf_Kp = 0;
  bool sw_valid = false;        // the factor bank holds the current generation
};

void smoother_free(rbpf_ctx* c) {
  SmootherState* s = c->sm;
  if (!s) return;
  hipFree(s->d_xnk); hipFree(s->d_dyref); hipFree(s->d_pant_log); hipFree(s->d_pant); hipFree(s->d_wc2);
  hipFree(s->d_Pfull); hipFree(s->d_G); hipFree(s->d_S); hipFree(s->d_L); hipFree(s->d_e);
  for (int b = 0; b < 2; ++b) { hipFree(s->d_Imat[b]); hipFree(s->d_Hb[b]); hipFree(s->d_ivec[b]); hipFree(s->d_hld[b]); hipFree(s->d_qf[b]); }
  hipFree(s->d_ImatAdd); hipFree(s->d_ivecAdd); hipFree(s->d_Imat0); hipFree(s->d_ak); hipFree(s->d_ivec0); hipFree(s->d_hld0);
  hipFree(s->d_Rinv); hipFree(s->d_Lsw[0]); hipFree(s->d_Lsw[1]); hipFree(s->d_W);
  hipFree(s->d_base_slot); hipFree(s->d_Xp); hipFree(s->d_marks);
  hipFree(s->d_owner_now); hipFree(s->d_base_gid); hipFree(s->d_base_loc); hipFree(s->d_rf_send); hipFree(s->d_rf_recv); hipFree(s->d_rf_idx);
  delete s;
  c->sm = nullptr;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// paNtLog(i) = log(w(i)) + logwDyn  (particleSmoother.m:175-182,232); logwMeas is added later.
__global__ void anc_dyn_kernel(ModelDev M, int N, const double* __restrict__ xn_prev /*SoA [nN][N]*/,
                               const double* __restrict__ xnk_t, const double* __restrict__ odo,
                               const double* __restrict__ Lq, const double* __restrict__ w, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double xi[8], xk[8], ed[8];
  for (int c = 0; c < M.nN; ++c) { xi[c] = xn_prev[(size_t)c * N + i]; xk[c] = xnk_t[c]; }
  // same arithmetic as rbpf_kernels.hip::dyn_res_norm_dev (kept local: separate TU)
  double r[8];
  const int nw = M.nw;
  if (M.use_dyn_res_norm && M.kind == 1) {                       // run_dense3D_magfield.m:202-203
    for (int c = 0; c < 3; ++c) r[c] = xk[c] - xi[c] - odo[c];
    const double oqi[4] = {odo[3], -odo[4], -odo[5], -odo[6]};
    const double xqi[4] = {xi[3], -xi[4], -xi[5], -xi[6]};
    double t1[4], t2[4];
    qleft_mul(oqi, xqi, t1);
    qleft_mul(t1, &xk[3], t2);
    logq_dev(t2, &r[3]);
  } else if (M.use_dyn_res_norm && M.kind == 2) {                // run_dense2D_withHeading.m:77
    r[0] = xk[2] - xi[2] - odo[2];
  } else {                                                       // particleSmoother.m:176
    for (int c = 0; c < nw; ++c) r[c] = xk[c] - xi[c] - odo[c];
  }
  for (int q = nw - 1; q >= 0; --q) {                            // r' / Lq
    double s = r[q];
    for (int k = q + 1; k < nw; ++k) s -= Lq[k + nw * q] * ed[k];
    ed[q] = s / Lq[q + nw * q];
  }
  double ss = 0.0;
  for (int q = 0; q < nw; ++q) ss += ed[q] * ed[q];
  out[i] = log(w[i]) + (-0.5 * ss);
}

// Generic family: eDyn was evaluated by the dynResNorm handle on the host (particleSmoother.m:178-182): e_dyn [nw x N].
__global__ void anc_dyn_ext_kernel(int N, int nw, const double* __restrict__ e_dyn, const double* __restrict__ w,
                                   double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double ss = 0.0;
  for (int q = 0; q < nw; ++q) { const double e = e_dyn[(size_t)i * nw + q]; ss += e * e; }
  out[i] = log(w[i]) + (-0.5 * ss);
}

// Packed storage of the information matrices (r03; the default-arithmetic information-form smoother at nLin >= 176).  They are
// symmetric and only their block-lower half is ever read (the factorisation), so a matrix is stored as its row tiles of 16 rows,
// row tile rt holding its 4 (rt + 1) column groups of 4 columns, 64 values each at kk * 16 + r -- the fragment order of the factor
// storage, i.e. the lane order of an MFMA operand.  A 16 x 64 strip of the matrix is then 8 KB of consecutive memory (16 wave loads
// of 512 B) instead of 64 segments of 128 B at a stride of 8 n bytes, and a bank entry takes 128 RT (RT + 1) doubles
// (1.15 MB at nLin = 515) instead of n * n (2.12 MB).  Inside the diagonal tiles the positions above the diagonal exist but are
// never written and never read unmasked.
__host__ __device__ inline size_t imat_packed_row(int rt) { return (size_t)128 * rt * (rt + 1); }
__host__ __device__ inline size_t imat_packed_doubles(int n) { return imat_packed_row((n + 15) >> 4); }
__host__ __device__ inline size_t imat_packed_index(int i, int j) {                // i >= j
  return imat_packed_row(i >> 4) + (size_t)(j >> 2) * 64 + (size_t)((j & 3) * 16 + (i & 15));
}

// Generic strided fp64 GEMM on the matrix cores, batched over blockIdx.z:  C = A * B.
// Element (i,k) of A at A[i*rsA + k*csA] etc.  64 x 64 output tile per workgroup, 4 waves of 32 x 32
// (2 x 2 v_mfma_f64_16x16x4 tiles); 16-deep slices of A and B go through LDS k-major so that an MFMA operand
// (lane l: row/col l & 15, k = l >> 4) is a conflict-free read, whichever stride of the source is the unit one.
struct GemmArgs {
  int M, N, K;
  const double* A; long rsA, csA, bsA;
  const double* B; long rsB, csB, bsB;
  double* C; long rsC, csC, bsC;
  // optional (zero: off).  lower: only the 64 x 64 tiles on and below the diagonal of C are computed (symmetric products whose
  // reader takes the block-lower part).  add: C += add[add_idx[batch]] (same strides as C; entries >= add_nbank come from
  // add_rec [..][M*N]; add_idx null: entry 0) -- the refresh of the carried factors adds the base matrix in the epilogue.
  int lower;
  const double* add; long add_stride; const int* add_idx; const double* add_rec; int add_nbank;
  // packed (zero: off): C -- symmetric, M == N -- and the matrices `add` / `add_rec` point at are information matrices in packed
  // block-lower storage (imat_packed_index below; bsC, add_stride and the records' pitch = imat_packed_doubles(M)); rsC / csC unused
  int packed;
  int add_self;      // C += (this batch's C): accumulate into the output (add / add_idx unused)
};

typedef double gemm_v4d __attribute__((ext_vector_type(4)));
constexpr int kGemmTile = 64, kGemmK = 16;

__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  __shared__ double As[kGemmK][kGemmTile + 1];   // As[k][i]
  __shared__ double Bs[kGemmK][kGemmTile + 1];   // Bs[k][j]
  const int bz = blockIdx.z;
  const double* A = g.A + (size_t)bz * g.bsA;
  const double* B = g.B + (size_t)bz * g.bsB;
  double* C = g.C + (size_t)bz * g.bsC;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i0 = blockIdx.y * kGemmTile, j0 = blockIdx.x * kGemmTile;
  if (g.lower == 1 && blockIdx.y < blockIdx.x) return;      // tile above the diagonal (rows along blockIdx.y)
  if (g.lower == 2 && blockIdx.x < blockIdx.y) return;      // ... of the transposed problem launch_gemm set up
  const int wi = (wv >> 1) * 32, wj = (wv & 1) * 32;
  const bool a_kfast = (g.csA == 1), b_kfast = (g.rsB == 1);
  gemm_v4d acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = (gemm_v4d){0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < g.K; k0 += kGemmK) {
    double av[4], bv[4];
    int ak[4], ai[4], bk[4], bj[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int q = tid + 256 * e;
      ak[e] = a_kfast ? (q & 15) : (q >> 6);
      ai[e] = a_kfast ? (q >> 4) : (q & 63);
      bk[e] = b_kfast ? (q & 15) : (q >> 6);
      bj[e] = b_kfast ? (q >> 4) : (q & 63);
      const int ia = min(i0 + ai[e], g.M - 1), ka = min(k0 + ak[e], g.K - 1);
      const int jb = min(j0 + bj[e], g.N - 1), kb = min(k0 + bk[e], g.K - 1);
      av[e] = A[(size_t)ia * g.rsA + (size_t)ka * g.csA];
      bv[e] = B[(size_t)kb * g.rsB + (size_t)jb * g.csB];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      As[ak[e]][ai[e]] = (k0 + ak[e] < g.K) ? av[e] : 0.0;                 // rows/cols past the edge are never stored
      Bs[bk[e]][bj[e]] = (k0 + bk[e] < g.K) ? bv[e] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < kGemmK / 4; ++ks) {
      double a[2], b[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        a[x] = As[4 * ks + (lane >> 4)][wi + 16 * x + (lane & 15)];
        b[x] = Bs[4 * ks + (lane >> 4)][wj + 16 * x + (lane & 15)];
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
    }
    __syncthreads();
  }
  // result layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  const double* addp = nullptr;
  if (g.add_self) addp = C;
  else if (g.add) {
    const int e = g.add_idx ? g.add_idx[bz] : 0;
    const size_t rec_pitch = g.packed ? imat_packed_doubles(g.M) : (size_t)g.M * g.N;
    addp = (g.add_rec && e >= g.add_nbank) ? g.add_rec + (size_t)(e - g.add_nbank) * rec_pitch : g.add + (size_t)e * g.add_stride;
  }
  if (g.packed) {
    // the product is symmetric: this tile's element (a, b) is stored as row b, column a of the block-lower matrix -- the lanes
    // of a result register (b = lane & 15 fastest, then a & 3 = lane >> 4) are then 64 consecutive doubles of the packed layout
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ca = i0 + wi + 16 * x + (lane >> 4) + 4 * r, rb = j0 + wj + 16 * y + (lane & 15);   // column, row
          if (rb < g.M && ca < g.M && (ca >> 4) <= (rb >> 4)) {
            const size_t o = imat_packed_index(rb, ca);
            C[o] = addp ? acc[x][y][r] + addp[o] : acc[x][y][r];
          }
        }
    return;
  }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wi + 16 * x + (lane >> 4) + 4 * r, j = j0 + wj + 16 * y + (lane & 15);
        if (i < g.M && j < g.N) {
          const size_t o = (size_t)i * g.rsC + (size_t)j * g.csC;
          C[o] = addp ? acc[x][y][r] + addp[o] : acc[x][y][r];
        }
      }
}

static hipError_t launch_gemm(const GemmArgs& g0, int batch, hipStream_t s) {
  GemmArgs g = g0;
  if (g0.rsC == 1 && g0.csC != 1) {
    // the lanes of a result tile run along its column index: compute C' = B' * A' so that they run along the unit stride
    g.M = g0.N; g.N = g0.M;
    g.A = g0.B; g.rsA = g0.csB; g.csA = g0.rsB; g.bsA = g0.bsB;
    g.B = g0.A; g.rsB = g0.csA; g.csB = g0.rsA; g.bsB = g0.bsA;
    g.rsC = g0.csC; g.csC = g0.rsC;
    if (g0.lower) g.lower = 2;
  }
  dim3 grid((g.N + kGemmTile - 1) / kGemmTile, (g.M + kGemmTile - 1) / kGemmTile, batch);
  hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, s, g);
  return hipGetLastError();
}

// Batched dense Cholesky + forward solve, one workgroup per particle (left-looking, thread per row).
//   MODE 0 (covariance form, particleSmoother.m:191-229):
//     A = S_j + kron(I,R) ; rhs = ytf - dyf*xl_j ; jitter retry (:221-224);
//     logwMeas = -sum(log(diag(cS))) - .5 v'v - numel(e)/2 log(2pi)
//   MODE 1 (information form, particleSmootherInformationForm.m:224-236):
//     A = Imat_j (+ pending H'R^-1 H) + ImatAddt ; rhs = ivec_j + ivecAddt ; no usable retry (quirk Q4);
//     logwMeas = -.5 qf_j - halfLogDetP_j - sum(log(diag(cI))) + .5 v'v
struct CholArgs {
  int mode, Msz, d, n, ldx;
  double* Lbuf; long ldL;           // [batch][Msz*Msz] column-major factors
  // mode 0
  const double* S; const double* R; const double* yf; const double* dyf; const double* xl; double jitter;
  const double* rhs;                // mode 0, sparse branch: innovation given directly [batch][Msz]; R == null: S complete
  // mode 1
  const double* Imat; long imat_stride; const double* Hb; const double* Rinv; const double* ImatAdd; const double* ivec;
  const double* ivecAdd; const double* qf; const double* hld;
  // mode 1, fused gather (:170,:253,:334): the particle's information matrix is its ancestor's stored matrix (bank entry
  // imat_anc[p], or a received record when the index is >= n_bank_local) plus its own last update; the sum is stored
  // as the particle's matrix (ImatOut) while it is loaded for the factorisation
  const int* imat_anc; double* ImatOut;
  int n_bank_local; const double* rec; size_t rec_stride, rec_off_Imat;
  double* pant_log;                 // += logwMeas
  int* status;
  int variant;                      // host side only: rbpf_options.chol_variant (0: kernel by matrix size)
  int batch, l_slots;               // 64-column kernel: l_slots > 0 = persistent workgroups, Lbuf holds l_slots factor workspaces
  int imat_packed;                  // mode 1: Imat / ImatAdd / ImatOut / the records' matrices in packed storage (below)
  long imat_out_stride;             // doubles between two matrices of ImatOut (n * n, or imat_packed_doubles(n))
};

// Blocked left-looking Cholesky on the fp64 matrix cores, one workgroup (16 waves) per particle.
//
// The matrix is augmented with the right-hand side as row M ([A rhs; rhs' *]): row M of its factor is (cS \ rhs)',
// so the forward solve comes with the factorisation.  Rows are padded to RT = ceil((M+1)/16) tiles of 16.
//
// Factor storage ("fragment order"): for row tile rt and column group kg (4 columns) the 64 values
// L(16 rt + r, 4 kg + kk) sit contiguously at ((kg * RT + rt) * 64 + kk * 16 + r) -- exactly the lane order of the
// A/B operand of v_mfma_f64_16x16x4_f64 (lane l: row/col l & 15, k = l >> 4), so one operand is one 512 B load.
//
// Column block jt (16 columns):
//   1. every wave owns up to 4 row tiles rt >= jt and accumulates  W' = panel * tile'  over the columns k < 16 jt
//      (A operand = the fragment of tile jt, B operand = the fragment of its own tile; result layout
//      W'[c = (l >> 4) + 4 reg][r = l & 15]), then  V' = A' - W';
//   2. wave 0 factors the 16 x 16 diagonal tile in registers (lane = row, v_readlane broadcasts) and inverts it;
//   3. the other tiles need X = V * Ld^-T, i.e. X' = Ld^-1 * V': the accumulator registers ARE the B operand of
//      that product (k = 4 s + (l >> 4) is register s), so it is 4 more MFMAs per tile with no data movement;
//      X' leaves in fragment order with one coalesced store per register.
// The triangular solve uses the explicit inverse of the 16 x 16 diagonal tile (error ~ eps * cond of that tile).
constexpr int kCholThreads = 1024;               // at most 16 waves; M + 1 <= 1024 rows -> <= 64 row tiles, <= 4 per wave
typedef double v4d __attribute__((ext_vector_type(4)));
#ifdef RBPF_CHOL_STAMPS                          // tuning aid: per-phase clocks of workgroup 0, printed from the device
#define CSTAMP(k) do { __syncthreads(); if (tid == 0) { const long long now_ = clock64(); cst[k] += now_ - clast; clast = now_; } } while (0)
#else
#define CSTAMP(k) do { } while (0)
#endif

__device__ inline double readlane_f64(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// acc[s] -= panel(jt) * tile(rt[s])' over the column groups [0, nkg); nkg is a multiple of 4.
template <int NT>
__device__ inline void chol_panel_product(const double* __restrict__ Lt, int RT, int jt, const int (&rt)[4], int nkg,
                                          int lane, v4d (&acc)[4]) {
  const size_t gs = (size_t)RT * 64;             // stride between column groups
  const double* pa = Lt + (size_t)jt * 64 + lane;
  const double* pb[NT];
#pragma unroll
  for (int s = 0; s < NT; ++s) pb[s] = Lt + (size_t)rt[s] * 64 + lane;
  double a0[4], b0[NT][4], a1[4], b1[NT][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    a0[u] = -pa[gs * u];
#pragma unroll
    for (int s = 0; s < NT; ++s) b0[s][u] = pb[s][gs * u];
  }
  for (int kg = 0; kg < nkg; kg += 8) {
    const int k1 = min(kg + 4, nkg - 4), k2 = min(kg + 8, nkg - 4);       // clamped: loads stay in range, branch-free
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a1[u] = -pa[gs * (k1 + u)];
#pragma unroll
      for (int s = 0; s < NT; ++s) b1[s][u] = pb[s][gs * (k1 + u)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < NT; ++s) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[s][u], acc[s], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0[u] = -pa[gs * (k2 + u)];
#pragma unroll
      for (int s = 0; s < NT; ++s) b0[s][u] = pb[s][gs * (k2 + u)];
    }
    if (kg + 4 < nkg) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < NT; ++s) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[s][u], acc[s], 0, 0, 0);
    }
  }
}

// Elements (i, j0 + cg + 4 q), q = 0..3, of the augmented matrix [A rhs; rhs' *] (lower triangle, everything else
// 0).  All loads are unconditional (clamped indices) and issued together; Hs / RH hold the pending measurement
// term of the information form (H and R^-1 H of this particle) in LDS.
template <int MODE>
__device__ inline void chol_aug_elems(const CholArgs& a, int p, int i, int jb, int M, const double* rhs_s,
                                      const double* Hs, const double* RH, double jit, v4d& out) {
  const int ic = min(i, M - 1);
  double v[4];
  int j[4], jc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { j[q] = jb + 4 * q; jc[q] = min(j[q], M - 1); }
  if (MODE == 0) {
    const int* dv = reinterpret_cast<const int*>(Hs);                        // (index / d) << 3 | index % d
    const int di = dv[ic];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = a.S[(size_t)p * M * M + (size_t)ic + (size_t)M * jc[q]];
    if (a.R) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int dj = dv[jc[q]];
        const double rr = a.R[(di & 7) + a.d * (dj & 7)];                    // kron(eye, R)
        v[q] += ((di >> 3) == (dj >> 3)) ? rr : 0.0;
      }
    }
  } else {
    double ad[4];
    const double* src = a.Imat + (size_t)p * a.imat_stride;                  // p: the ancestor's entry (resolved by the caller)
    size_t idx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // packed storage: the block-lower half only -- an element above the diagonal (masked out below) reads its mirror image
      idx[q] = a.imat_packed ? imat_packed_index(max(ic, jc[q]), min(ic, jc[q])) : (size_t)ic + (size_t)a.n * jc[q];
      v[q] = src[idx[q]];
      ad[q] = a.ImatAdd[idx[q]];
    }
    if (Hs) {                                                                // + dyi'/R*dyi of the last update (:334)
      double sacc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int aa = 0; aa < a.d; ++aa) {
        const double h = Hs[aa * M + ic];
#pragma unroll
        for (int q = 0; q < 4; ++q) sacc[q] = fma(h, RH[aa * M + jc[q]], sacc[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += sacc[q];
    }
    if (a.ImatOut && i < M) {                                                // Imat(:,:,i) of the new generation
      double* dst = a.ImatOut + (size_t)p * a.imat_out_stride;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j[q] < M && (!a.imat_packed || i >= j[q])) __builtin_nontemporal_store(v[q], &dst[idx[q]]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += ad[q];                               // :225
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (i == j[q]) v[q] += jit;
    if (i == M) v[q] = rhs_s[jc[q]];
    out[q] = (j[q] < M && i <= M && i >= j[q]) ? v[q] : 0.0;
  }
}

// acc[s] = A'(tile rt[s], block jt) - panel(jt) * tile(rt[s])' for the NT tiles of this wave
template <int NT, int MODE>
__device__ inline void chol_block_front(const CholArgs& a, int p, const double* __restrict__ Lt, int RT, int jt,
                                        const int (&rt)[4], int M, const double* rhs_s, const double* Hs,
                                        const double* RH, double jit, int lane, v4d (&acc)[4]) {
#pragma unroll
  for (int s = 0; s < NT; ++s)
    chol_aug_elems<MODE>(a, p, 16 * rt[s] + (lane & 15), 16 * jt + (lane >> 4), M, rhs_s, Hs, RH, jit, acc[s]);
  if (jt > 0) chol_panel_product<NT>(Lt, RT, jt, rt, 4 * jt, lane, acc);
}

// sqrt(x) and 1/sqrt(x) from v_rsq_f64 + two coupled Newton steps (the library sqrt + divide pair is ~4x longer and
// sits on the serial path of the diagonal-tile factorisation)
__device__ inline void sqrt_rsqrt(double x, double& g, double& rinv) {
  const double y = __builtin_amdgcn_rsq(x);
  g = x * y;
  double h = 0.5 * y;
  double r = fma(-g, h, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-g, h, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  const double dd = fma(-g, g, x);
  g = fma(dd, h, g);
  rinv = h + h;
}

// Right-hand side (and, for the information form, the pending measurement term H, R^-1 H of this particle) into LDS.
__device__ inline void chol_prologue(const CholArgs& a, int p, int tid, int nthreads, int M, double* rhs_s, double* Hs,
                                     double* RH, bool pend) {
  for (int i = tid; i < M; i += nthreads) {
    double r;
    if (a.mode == 0 && a.rhs) {
      r = a.rhs[(size_t)p * M + i];                                         // particleSmoother.m:207-208 (sparse branch)
    } else if (a.mode == 0) {
      double s = 0.0;
      const double* dr = a.dyf + (size_t)i * a.n;
      const double* x = a.xl + (size_t)p * a.ldx;
      for (int c = 0; c < a.n; ++c) s = fma(dr[c], x[c], s);
      r = a.yf[i] - s;                                                      // particleSmoother.m:193
    } else {
      r = a.ivec[(size_t)p * a.ldx + i] + a.ivecAdd[i];                     // InformationForm.m:224
    }
    rhs_s[i] = r;
    if (a.mode == 0) reinterpret_cast<int*>(Hs)[i] = ((i / a.d) << 3) | (i % a.d);
    if (pend) {
      const double* H = a.Hb + (size_t)p * a.d * a.ldx;
      for (int aa = 0; aa < a.d; ++aa) {
        double t = 0.0;
        for (int bb = 0; bb < a.d; ++bb) t = fma(a.Rinv[aa + a.d * bb], H[(size_t)bb * a.ldx + i], t);
        Hs[aa * M + i] = H[(size_t)aa * a.ldx + i];
        RH[aa * M + i] = t;
      }
    }
  }
}

__device__ inline bool chol_diag_tile_frag(v4d& V, v4d& NI, int nvalid, int lane);   // rbpf_chol64.hpp

// W waves per workgroup (4, 8 or 16): the smallest that keeps <= 4 row tiles per wave.  A small matrix then leaves room
// for several workgroups per CU, whose single-wave diagonal-tile sections and barriers overlap (a 16-wave workgroup owns
// the whole register file: at n = 128 seven of its waves had no tile and every CU waited on one particle's serial chain).
// NTMAX: the most row tiles a wave can own for this matrix size, ceil(RT / W); variants above it are not compiled in
// (their operand buffers would only raise the register pressure of the whole kernel).
template <int W, int NTMAX>
__global__ __launch_bounds__(W * 64, 4) void chol_solve_kernel(CholArgs a_in) {
  constexpr int kThreadsW = W * 64;
  extern __shared__ double csm[];
  CholArgs a = a_in;
  const int p = blockIdx.x, tid = threadIdx.x, M = a.Msz;
  if (a.mode == 1) {
    // source of the stored information matrix: my own entry, my ancestor's entry, or a received record
    const int src = a.imat_anc ? a.imat_anc[p] : p;
    const bool remote = a.rec != nullptr && src >= a.n_bank_local;
    a.Imat = remote ? a.rec + (size_t)(src - a.n_bank_local) * a.rec_stride + a.rec_off_Imat
                    : a.Imat + (size_t)src * a.imat_stride;
    a.imat_stride = 0;
  }
  const int lane = tid & 63, wv = tid >> 6;
  const int RT = (M + 1 + 15) >> 4;
  double* Lt = a.Lbuf + (size_t)p * a.ldL;       // fragment order, (16 RT)^2 doubles
  double* Dg = csm;                               // [16][16] diagonal tile, Dg[c][r]
  double* LinvT = Dg + 256;                       // [16][16] LinvT[cc][c] = inv(Ld)(c, cc)
  double* red = LinvT + 256;                      // [32] reduction scratch
  double* rhs_s = red + 32;                       // [M]
  int& sfail = *reinterpret_cast<int*>(rhs_s + M);
  const bool pend = (a.mode == 1 && a.Hb != nullptr);
  double* Hs = (pend || a.mode == 0) ? rhs_s + M + 2 : nullptr;   // [d][M] H of this particle (mode 0: int index table)
  double* RH = pend ? Hs + (size_t)a.d * M : nullptr;   // [d][M]  R^-1 H
#ifdef RBPF_CHOL_STAMPS
  long long cst[6] = {0, 0, 0, 0, 0, 0}, clast = clock64();
#endif
  chol_prologue(a, p, tid, kThreadsW, M, rhs_s, Hs, RH, pend);
  const int cr = lane & 15, cg = lane >> 4;       // my row within a tile / my column group within a block
  double jit = 0.0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (tid == 0) sfail = 0;
    __syncthreads();
    for (int jt = 0; 16 * jt < M; ++jt) {
      const int j0 = 16 * jt;
      int rt[4], nt = 0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int t = jt + wv + W * s;
        rt[s] = min(t, RT - 1);
        nt += (t < RT) ? 1 : 0;
      }
      v4d acc[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[s] = (v4d){0.0, 0.0, 0.0, 0.0};
      const int fsel = nt ? nt + 4 * a.mode : 0;  // wave-uniform
#define RBPF_CF(NT_, MODE_) chol_block_front<(NT_ <= NTMAX ? NT_ : 1), MODE_>(a, p, Lt, RT, jt, rt, M, rhs_s, Hs, RH, jit, lane, acc)
      switch (fsel) {
        case 1: RBPF_CF(1, 0); break;
        case 2: if (NTMAX >= 2) RBPF_CF(2, 0); break;
        case 3: if (NTMAX >= 3) RBPF_CF(3, 0); break;
        case 4: if (NTMAX >= 4) RBPF_CF(4, 0); break;
        case 5: RBPF_CF(1, 1); break;
        case 6: if (NTMAX >= 2) RBPF_CF(2, 1); break;
        case 7: if (NTMAX >= 3) RBPF_CF(3, 1); break;
        case 8: if (NTMAX >= 4) RBPF_CF(4, 1); break;
        default: break;
      }
#undef RBPF_CF
      CSTAMP(0);
      // diagonal tile: wave 0, slot 0 — factorised in the accumulator's own layout (rbpf_chol64.hpp), 4.5 K clocks
      // instead of the 15 K of the former lane-per-row routine through LDS
      if (wv == 0) {
        v4d V = acc[0], NI;
        const bool bad = chol_diag_tile_frag(V, NI, M - j0, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          LinvT[64 * q + lane] = -NI[q];                                     // inv(Ld)(lane & 15, 4 q + (lane >> 4))
          Lt[((size_t)(4 * jt + q) * RT + jt) * 64 + lane] = V[q];
        }
        if (bad && lane == 0) sfail = 1;
      }
      __syncthreads();
      CSTAMP(2);
      if (sfail) break;
      // X' = inv(Ld) * V' for the tiles below the diagonal
      {
        double af[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) af[q] = LinvT[64 * q + lane];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (s < nt && !(wv == 0 && s == 0)) {
            v4d x = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) x = __builtin_amdgcn_mfma_f64_16x16x4f64(af[q], acc[s][q], x, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) Lt[((size_t)(4 * jt + q) * RT + rt[s]) * 64 + lane] = x[q];
          }
        }
      }
      __syncthreads();                            // the block's columns are visible to the next panel product
      CSTAMP(3);
    }
    __syncthreads();
    const int failed = sfail;
    __syncthreads();
    if (!failed) {
      // sum(log(diag(cS))) and v'v
      double sl = 0.0, vv = 0.0;
      for (int j = tid; j < M; j += kThreadsW) {
        const size_t col = (size_t)(j >> 2) * RT * 64 + (size_t)(j & 3) * 16;
        const double dj = Lt[col + (size_t)(j >> 4) * 64 + (j & 15)];
        const double vj = Lt[col + (size_t)(M >> 4) * 64 + (M & 15)];
        sl += log(dj);
        vv = fma(vj, vj, vv);
      }
      sl = wave_sum(sl); vv = wave_sum(vv);
      if (lane == 0) { red[wv] = sl; red[16 + wv] = vv; }
      __syncthreads();
      if (tid == 0) {
        sl = 0.0; vv = 0.0;
        for (int w = 0; w < W; ++w) { sl += red[w]; vv += red[16 + w]; }
        double lw;
        if (a.mode == 0) lw = -sl - 0.5 * vv - 0.5 * (double)M * 1.8378770664093453;     // log(2*pi)
        else lw = -0.5 * a.qf[p] - a.hld[p] - sl + 0.5 * vv;
        a.pant_log[p] += lw;
#ifdef RBPF_CHOL_STAMPS
        cst[4] = clock64() - clast;
        if (p == 0 && M >= 500) printf("chol M=%d mode=%d clocks: product %lld elem %lld diag %lld solve %lld tail %lld\n", M, a.mode, cst[0], cst[1], cst[2], cst[3], cst[4]);
#endif
      }
      return;
    }
    if (a.mode == 1 || attempt == 1) {
      if (tid == 0) { atomicOr(a.status, 2); a.pant_log[p] = nan(""); }
      return;
    }
    jit = a.jitter;                                                         // particleSmoother.m:223
  }
}

static size_t chol_lds_bytes(int M, int d) { return ((size_t)256 + 256 + 32 + M + 2 + (d ? 2 * (size_t)d * M : (size_t)M)) * sizeof(double); }
template <int W, int NTMAX>
static hipError_t launch_chol_w(const CholArgs& ca, int batch, size_t lds, hipStream_t st) {
  static std::atomic<uint64_t> attr{0};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&chol_solve_kernel<W, NTMAX>), 150 * 1024, attr)) return e;
  hipLaunchKernelGGL((chol_solve_kernel<W, NTMAX>), dim3(batch), dim3(W * 64), lds, st, ca);
  return hipGetLastError();
}

template <int W>
static hipError_t launch_chol_nt(const CholArgs& ca, int batch, size_t lds, int ntmax, hipStream_t st) {
  switch (ntmax) {
    case 1: return launch_chol_w<W, 1>(ca, batch, lds, st);
    case 2: return launch_chol_w<W, 2>(ca, batch, lds, st);
    case 3: return launch_chol_w<W, 3>(ca, batch, lds, st);
    default: return launch_chol_w<W, 4>(ca, batch, lds, st);
  }
}

#include "rbpf_chol64.hpp"
#include "rbpf_chol_small.hpp"
#include "rbpf_chol_sweep.hpp"

// batched ancestor-weight factorisation: d_lds = number of pending-update rows kept in LDS (mode 1: n_y, mode 0: 0).
// Matrices of more than 11 row tiles (n >= 176) take the 64-column kernel (rbpf_chol64.hpp; 4 waves and two workgroups per
// CU up to 27 row tiles, 8 waves above; measured crossovers, profiles/r01y_chol_bench.jsonl), information-form matrices of
// 5..9 row tiles (dense-radio n = 128) the register-resident kernel (rbpf_chol_small.hpp), the rest the 16-column kernel.  RBPF_CHOL64 = 0 / 1 forces one of them (tuning and tests).
static hipError_t launch_chol16(const CholArgs& ca, int batch, int d_lds, hipStream_t st, int w_force = 0) {
  const int RT = (ca.Msz + 1 + 15) >> 4;
  int W = (RT <= 16) ? 4 : (RT <= 32) ? 8 : 16;
  if ((w_force == 4 || w_force == 8 || w_force == 16) && (RT + w_force - 1) / w_force <= 4) W = w_force;
  const size_t lds = chol_lds_bytes(ca.Msz, d_lds);
  const int ntmax = (RT + W - 1) / W;
  if (W == 4) return launch_chol_nt<4>(ca, batch, lds, ntmax, st);
  if (W == 8) return launch_chol_nt<8>(ca, batch, lds, ntmax, st);
  return launch_chol_nt<16>(ca, batch, lds, ntmax, st);
}

// rbpf_options.chol_variant / the `variant` of rbpf_chol_weights: is the kernel usable for this matrix?
static bool chol_variant_ok(const CholArgs& ca, int d_lds, int variant) {
  const int RT = (ca.Msz + 1 + 15) >> 4;
  switch (variant) {
    case 0: case 16: return true;
    case 64: case 648: case 644: return chol64_lds_bytes(ca.Msz, d_lds) <= kC64MaxLds;
    case 1: case 10: case 11: case 12: case 14: return ca.mode == 1 && RT >= 5 && RT <= kCsMaxRT;
    default: return false;
  }
}

static hipError_t launch_chol(const CholArgs& ca, int batch, int d_lds, hipStream_t st) {
  const int RT = (ca.Msz + 1 + 15) >> 4;
  if (ca.variant != 0) {                                                      // explicit choice (option / test entry point)
    if (!chol_variant_ok(ca, d_lds, ca.variant)) return hipErrorInvalidValue;
    if (ca.variant == 1) return launch_chol_small(ca, batch, d_lds, st);          // register-resident kernel, default shape
    if (ca.variant == 11 || ca.variant == 12 || ca.variant == 14) return launch_chol_small(ca, batch, d_lds, st, ca.variant - 10);   // 1 / 2 / 4 waves per matrix
    if (ca.variant == 10) return launch_chol_small(ca, batch, d_lds, st, 10);                                                         // one wave, left-looking
    if (ca.variant == 16) return launch_chol16(ca, batch, d_lds, st);
    return launch_chol64(ca, batch, d_lds, st, ca.variant == 648 ? 8 : ca.variant == 644 ? 4 : 0);
  }
  // diagnostic builds (-DRBPF_TUNING) can override the choice from the environment
  static const int w_env = tuning_env("RBPF_CHOL_WAVES") ? atoi(tuning_env("RBPF_CHOL_WAVES")) : 0;      // force 4 / 8 / 16
  const char* v64 = tuning_env("RBPF_CHOL64");
  // (a 128-column kernel -- factor re-read once per 128 columns -- was built and measured in r04: 17.1 against 16.2 ms per launch in the
  //  smoother at N_P = 8192, n = 515; removed in r05, see commit 4711b84 and DESIGN_NOTEBOOK.md 9)
  if ((v64 ? atoi(v64) != 0 : RT > 11) && chol64_lds_bytes(ca.Msz, d_lds) <= kC64MaxLds) return launch_chol64(ca, batch, d_lds, st);
  const char* vsm = tuning_env("RBPF_CHOL_SMALL");                                // 0: keep the 16-column kernel for 5..9 row tiles
  if (!v64 && ca.mode == 1 && RT >= 5 && RT <= kCsMaxRT && !(vsm && atoi(vsm) == 0)) return launch_chol_small(ca, batch, d_lds, st);
  return launch_chol16(ca, batch, d_lds, st, w_env);
}

// Persistent mode of the 64-column kernel (rbpf_chol64.hpp, l_slots): one factor workspace per resident workgroup instead of one
// per particle.  MEASURED AND NOT KEPT (r03): the idea was that 256 reused workspaces (294 MB at n = 515) would stay in the 256 MB
// Infinity Cache, so that the factor's write + re-reads (4.1 of the 9.8 MB a particle moves) stop reaching HBM; in the smoother
// (N_P = 8192, n = 515) the launch got SLOWER, 16.1 -> 19.8 ms (the cache does not keep them, and the persistent loop loses the
// dispatcher's overlap of one workgroup's tail with the next one's prologue).  Compiled in only with -DRBPF_C64_PERSIST=1.
#ifndef RBPF_C64_PERSIST
#define RBPF_C64_PERSIST 0
#endif
static int chol64_workspace_slots(int M) {
#if RBPF_C64_PERSIST
  static int cus = 0;
  if (!cus) { int dev = 0; hipGetDevice(&dev); if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256; }
  const int RT = (M + 1 + 15) >> 4;
  return cus * ((RT > 27) ? 1 : 2);
#else
  (void)M;
  return 0;
#endif
}

static size_t chol_factor_doubles(int M) { const size_t mp = (size_t)16 * ((M + 1 + 15) / 16); return mp * mp; }

// [T][nN] row-per-step trajectory -> per-step pointer is d_xnk + t*nN.  Backtrace writes [nN x T]
// column-major (= [T][nN] row-per-step), so no transpose is needed.

// Sum over the reference trajectory (InformationForm.m:132-146), in the reference's order jj = 1..T,
// and the per-step decrement (:194-201).  sign = +1 accumulates steps [t0,t1), -1 subtracts them.
__global__ void info_addt_kernel(int n, int d, int t0, int t1, double sign, const double* __restrict__ dyref,
                                 const double* __restrict__ Rinv, const double* __restrict__ y,
                                 double* __restrict__ ImatAdd, double* __restrict__ ivecAdd, int packed) {
  size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (size_t)n * n) return;
  const int i = (int)(q % n), j = (int)(q / n);
  if (packed) {                                   // block-lower storage (imat_packed_index)
    if (i < j) return;
    q = imat_packed_index(i, j);
  }
  double acc = ImatAdd[q];
  double av = (j == 0) ? ivecAdd[i] : 0.0;
  for (int t = t0; t < t1; ++t) {
    const double* H = dyref + (size_t)t * d * n;
    double s = 0.0, sv = 0.0;
    for (int a = 0; a < d; ++a) {
      double tt = 0.0, ty = 0.0;
      for (int b = 0; b < d; ++b) { tt = fma(Rinv[a + d * b], H[(size_t)b * n + j], tt); ty = fma(Rinv[a + d * b], y[(size_t)t * d + b], ty); }
      s = fma(H[(size_t)a * n + i], tt, s);
      sv = fma(H[(size_t)a * n + i], ty, sv);
    }
    acc += sign * s;
    av += sign * sv;
  }
  ImatAdd[q] = acc;
  if (j == 0) ivecAdd[i] = av;
}

__global__ void gather_scalar_kernel(int N, const int* __restrict__ ai, const double* __restrict__ in, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) out[i] = in[ai ? ai[i] : i];
}

__global__ void fill_kernel(size_t count, double v, double* p) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < count) p[q] = v;
}

// sharded smoother: information part of the send records [ivec | halfLogDetP | pending H | Imat] of the particles idx[.]
__global__ __launch_bounds__(256) void pack_info_kernel(int n, int d, int ldx, const int* __restrict__ idx,
                                                        const double* __restrict__ ivec, const double* __restrict__ hld,
                                                        const double* __restrict__ Hb, const double* __restrict__ Imat,
                                                        size_t imat_stride, size_t imat_len, double* __restrict__ rec, size_t rec_stride,
                                                        size_t off_I, size_t off_hld, size_t off_Hb, size_t off_Imat) {
  const int p = blockIdx.x, src = idx[p];
  double* r = rec + (size_t)p * rec_stride;
  if (blockIdx.y == 0) {
    for (int q = threadIdx.x; q < ldx; q += blockDim.x) r[off_I + q] = ivec[(size_t)src * ldx + q];
    for (int q = threadIdx.x; q < d * ldx; q += blockDim.x) r[off_Hb + q] = Hb[(size_t)src * d * ldx + q];
    if (threadIdx.x == 0) { r[off_hld] = hld[src]; r[off_hld + 1] = 0.0; }
  }
  if (Imat) {
    const size_t nn = imat_len, per = (nn + gridDim.y - 1) / gridDim.y;
    const size_t q0 = (size_t)blockIdx.y * per, q1 = q0 + per < nn ? q0 + per : nn;
    const double* im = Imat + (size_t)src * imat_stride;
    for (size_t q = q0 + threadIdx.x; q < q1; q += blockDim.x) r[off_Imat + q] = im[q];
  }
}

}  // namespace rbpf

using namespace rbpf;

// Packed storage of the information matrices: whenever they go through the 64-column kernel (more than eleven row tiles, i.e.
// nLin >= 176).  With carried factors (refresh = K > 1) the banks are written by the G'G product of the refreshes only, whose
// epilogue writes the packed layout as well (r05) -- except for host-callback models (`lazy` false), whose matrices are advanced
// every step by the copy kernel on full squares.
static bool imat_storage_packed(int n, int d, int refresh, bool lazy) {
  return (refresh <= 1 || lazy) && ((n + 1 + 15) >> 4) > 11 && chol64_lds_bytes(n, d) <= kC64MaxLds;
}

// rbpf_options.chol_refresh -> the K in use.  0 = automatic: the carried factors (K = 32) for the recognised dense families from
// nLin = 128 on (dense-radio's 128, dense-mag's 259 / 515: the sizes they were validated and measured at; below that the
// factorisation is not what a step costs), the from-scratch factorisation elsewhere.
constexpr int kAutoCholRefresh = 32;
static bool chol_carry_supported(int kind, int n, int d) {
  return (kind == RBPF_MODEL_DENSE_MAG_6D || kind == RBPF_MODEL_DENSE_RADIO_2DH) && (d == 1 || d == 3) && sweep_slots(n) <= kSweepMaxSlots &&
         chol64_lds_bytes(n, d) <= kC64MaxLds;
}
int rbpf::resolve_chol_refresh(int kind, int n, int d, int requested) {
  if (requested > 1) return requested;
  if (requested == 1 || requested < 0) return 1;
  return (chol_carry_supported(kind, n, d) && n >= 128) ? kAutoCholRefresh : 1;
}

// host copy of Imat0 (column-major n x n) in the storage of the banks
static std::vector<double> imat_host_storage(const std::vector<double>& full, int n, bool packed) {
  if (!packed) return full;
  std::vector<double> pk(imat_packed_doubles(n), 0.0);
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) pk[imat_packed_index(i, j)] = full[(size_t)i + (size_t)n * j];
  return pk;
}

// information-form hooks (defined below)
static int info_begin_iteration(rbpf_ctx* c, const double* ivec0, double hld0, double qf0, double halfLogDetR, const double* d_Rinv);
static int info_fill_chol_args(rbpf_ctx* c, CholArgs& ca, const double* d_Rinv, const int* anc);
static int info_step(rbpf_ctx* c, int k, int t, const double* xref, int n_draw, const double* d_Rinv);

#define RB_TRY(x) do { int _s = (x); if (_s != RBPF_OK) return _s; } while (0)

template <typename T>
static int dmalloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc((void**)p, count * sizeof(T));
  if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
  return RBPF_OK;
}

// Normalise N ancestor log-probabilities (particleSmoother.m:236-238) and draw ONE index from them into ai[slot]
// (:241).  Up to 8192 entries: the single-workgroup kernel with the strict left-to-right running sum.  Above: the
// multi-workgroup pipeline (parallel prefix + certified search + exact fallback), same indices, ~0.15 ms instead of
// 0.8 ms at N = 65536 -- on 8 GPUs that serial, replicated kernel would otherwise dominate a sharded smoother step.
static int normalise_draw_one(rbpf_ctx* c, int N, int t, int k, const double* logw, double* w, double* wc, int slot,
                              const double* U, int* ai, hipStream_t st) {
  NormArgs nm;
  nm.N = N; nm.nN = 0; nm.t = t; nm.logw = logw; nm.w = w; nm.wc = wc; nm.xn = nullptr;
  nm.traj_max = nullptr; nm.traj_mean = nullptr; nm.iw_max = c->d_flags + 3; nm.lse_out = nullptr;
  SearchArgs sa;
  sa.N = N; sa.n_draw = 1; sa.t = t; sa.wc = wc; sa.rng_mode = c->rng_mode; sa.k_iter = k;
  sa.slot0 = slot; sa.u_is_scalar = 0; sa.U = U; sa.seed = c->seed; sa.ai = ai; sa.overflow = c->d_flags + 1;
  if (N > kSingleWgResampleMaxN) {
    nm.parallel_scan = 1;
    sa.approx = 1; sa.ambiguous = c->d_flags + 4; sa.w = w; sa.wc_exact = wc;
    HIPCHK(launch_resample_pipeline(nm, &sa, nullptr, nullptr, nullptr, c->d_rs, st));
  } else {
    HIPCHK(launch_normalise_scan(nm, st));
    HIPCHK(launch_search(sa, st));
  }
  return RBPF_OK;
}

// R^-1 and 0.5*log(det(R)) via Cholesky (d <= 8)
static int invert_R(const std::vector<double>& Rh, int d, std::vector<double>& Rinv, double& halfLogDetR) {
  halfLogDetR = 0.0;
  std::vector<double> Lr((size_t)d * d, 0.0);
  for (int j = 0; j < d; ++j) {
    double sdiag = Rh[j + (size_t)d * j];
    for (int k = 0; k < j; ++k) sdiag -= Lr[j + (size_t)d * k] * Lr[j + (size_t)d * k];
    if (!(sdiag > 0)) { set_error("R must be positive definite"); return RBPF_ERR_CHOL_FAILED; }
    Lr[j + (size_t)d * j] = std::sqrt(sdiag);
    for (int i = j + 1; i < d; ++i) {
      double v = Rh[i + (size_t)d * j];
      for (int k = 0; k < j; ++k) v -= Lr[i + (size_t)d * k] * Lr[j + (size_t)d * k];
      Lr[i + (size_t)d * j] = v / Lr[j + (size_t)d * j];
    }
    halfLogDetR += std::log(Lr[j + (size_t)d * j]);
  }
  for (int col = 0; col < d; ++col) {             // solve R x = e_col
    std::vector<double> y(d), x(d);
    for (int i = 0; i < d; ++i) { double v = (i == col); for (int k = 0; k < i; ++k) v -= Lr[i + (size_t)d * k] * y[k]; y[i] = v / Lr[i + (size_t)d * i]; }
    for (int i = d - 1; i >= 0; --i) { double v = y[i]; for (int k = i + 1; k < d; ++k) v -= Lr[k + (size_t)d * i] * x[k]; x[i] = v / Lr[i + (size_t)d * i]; }
    for (int i = 0; i < d; ++i) Rinv[i + (size_t)d * col] = x[i];
  }
  return RBPF_OK;
}

// information-form initial values (quirk Q5: diagonal of P0 only, particleSmootherInformationForm.m:110-115)
static void info_initial_values(rbpf_ctx* c, std::vector<double>& ivec0, std::vector<double>& Imat0, double& hld0) {
  const int n = c->mdl.n;
  hld0 = 0.0;
  for (int r = 0; r < n; ++r) {
    const double pd = c->h_P0[r + (size_t)n * r];
    Imat0[r + (size_t)n * r] = 1.0 / pd;
    ivec0[r] = (1.0 / pd) * c->h_x0l[r];
    hld0 += std::log(std::sqrt(pd));
  }
}

// W = inv(chol(R,'lower')): W' W = R^-1, so H' R^-1 H = sum_a (W H)_a' (W H)_a
static void whitening_factor(const std::vector<double>& Rh, int d, std::vector<double>& Wm) {
  std::vector<double> Lr((size_t)d * d, 0.0);
  Wm.assign((size_t)d * d, 0.0);
  for (int j = 0; j < d; ++j) {
    double sd = Rh[j + (size_t)d * j];
    for (int q = 0; q < j; ++q) sd -= Lr[j + (size_t)d * q] * Lr[j + (size_t)d * q];
    Lr[j + (size_t)d * j] = std::sqrt(sd);
    for (int i = j + 1; i < d; ++i) {
      double v = Rh[i + (size_t)d * j];
      for (int q = 0; q < j; ++q) v -= Lr[i + (size_t)d * q] * Lr[j + (size_t)d * q];
      Lr[i + (size_t)d * j] = v / Lr[j + (size_t)d * j];
    }
  }
  for (int col = 0; col < d; ++col)
    for (int i = 0; i < d; ++i) {
      double v = (i == col);
      for (int q = 0; q < i; ++q) v -= Lr[i + (size_t)d * q] * Wm[q + (size_t)d * col];
      Wm[i + (size_t)d * col] = v / Lr[i + (size_t)d * i];
    }
}

// Refresh of the carried factors, second half: factorise Imat + ImatAddt of the N particles whose information matrices the G'G product
// has just materialised (bank s->imat_cur) with the 64-column kernel, add logwMeas to pant_log, and store the factors in the sweep
// layout in the OTHER factor bank (the caller flips sw_cur).  Runs in chunks of s->l_chunk particles over one set of factor
// workspaces: same launches per particle, 2.2 MB per particle less memory at nLin = 515.
// chunk_imat != null (refresh-free smoother): the particles first .. first + N - 1 (N <= l_chunk) whose matrices sit in that chunk buffer.
static int refresh_factorise(rbpf_ctx* c, const double* d_Rinv, double* pant_log, int N, hipStream_t st, int first = 0,
                             const double* chunk_imat = nullptr) {
  SmootherState* s = c->sm;
  const int n = c->mdl.n, d = c->mdl.d;
  const size_t fd = chol_factor_doubles(n), sd = sweep_factor_doubles(n);
  for (int q0 = 0; q0 < N; q0 += s->l_chunk) {
    const int cnt = std::min(s->l_chunk, N - q0), p0 = first + q0;
    CholArgs ca;
    std::memset(&ca, 0, sizeof(ca));
    ca.d = d; ca.n = n; ca.ldx = c->lay.ldx; ca.status = c->d_flags; ca.pant_log = pant_log + p0;
    ca.mode = 1; ca.Msz = n; ca.Lbuf = s->d_L; ca.ldL = (long)fd;
    ca.Imat = chunk_imat ? chunk_imat + (size_t)q0 * s->imat_len : s->d_Imat[s->imat_cur] + (size_t)p0 * s->imat_len;
    ca.imat_stride = (long)s->imat_len; ca.imat_packed = s->imat_packed ? 1 : 0;
    ca.imat_anc = nullptr; ca.ImatOut = nullptr; ca.Hb = nullptr; ca.Rinv = d_Rinv; ca.ImatAdd = s->d_ImatAdd;
    ca.ivec = s->d_ivec[s->icur] + (size_t)p0 * c->lay.ldx; ca.ivecAdd = s->d_ivecAdd;
    ca.qf = s->d_qf[s->icur] + p0; ca.hld = s->d_hld[s->icur] + p0;
    ca.variant = 64;                                                   // the conversion below reads the 64-column kernel's layout
    HIPCHK(launch_chol(ca, cnt, d, st));
    HIPCHK(launch_sweep_from_chol64(n, cnt, s->d_L, fd, s->d_Lsw[s->sw_cur ^ 1] + (size_t)p0 * sd, sd, st));
  }
  return RBPF_OK;
}

static int smoother_run(rbpf_ctx* c, int N_K, int info_form, rbpf_smoother_out* out) {
  const int N = c->N, T = c->T, nN = c->mdl.nN, n = c->mdl.n, d = c->mdl.d, nw = c->mdl.nw;
  const Layout& L = c->lay;
  hipStream_t st = c->stream;
  SmootherState* s = new SmootherState();
  c->sm = s;
  if (!info_form && c->inplace) { set_error("inplace=1: the covariance-form smoother rewrites its covariances at every step (use the information form)"); return RBPF_ERR_UNSUPPORTED; }
  if (!info_form) c->lazy_depth = 1;          // the covariance form reads the flushed covariances of every particle every step
  c->sort_steps = true;
  const bool generic = c->mdl.kind == RBPF_MODEL_GENERIC_DENSE;
  const bool drn_cb = generic && c->cb.dyn_res_norm != nullptr;      // the handle; otherwise isempty(dynResNorm)
  if (generic) c->mdl.use_dyn_res_norm = 0;                          // device side: only the additive default exists
  if (!c->mdl.use_dyn_res_norm && !drn_cb && nw != nN) {
    set_error("isempty(dynResNorm): the additive default (particleSmoother.m:176) needs size(Q,1) == nNonLin");
    return RBPF_ERR_INVALID_ARG;
  }
  if (generic && !drn_cb && !c->cholQfull_ok) { set_error("isempty(dynResNorm): chol(dt*Q,'lower') failed (particleSmoother.m:177)"); return RBPF_ERR_CHOL_FAILED; }
  double* d_edyn = nullptr;
  struct EdynGuard { double** p; ~EdynGuard() { hipFree(*p); } } edyn_guard{&d_edyn};
  if (drn_cb) RB_TRY(dmalloc(&d_edyn, (size_t)N * nw));
  std::vector<double> h_edyn;
  const size_t Mmax = (size_t)d * T;
  s->Mmax = Mmax;
  RB_TRY(dmalloc(&s->d_xnk, (size_t)T * nN));
  RB_TRY(dmalloc(&s->d_dyref, (size_t)T * d * n));
  RB_TRY(dmalloc(&s->d_pant_log, (size_t)N));
  RB_TRY(dmalloc(&s->d_pant, (size_t)N * (c->opt.trace ? (size_t)T * N_K : 1)));
  {
    const size_t cnt = (size_t)N * (c->opt.trace ? (size_t)T * N_K : 1);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, cnt, (double)NAN, s->d_pant);
    HIPCHK(hipGetLastError());
  }
  RB_TRY(dmalloc(&s->d_wc2, (size_t)N));
  RB_TRY(dmalloc(&s->d_ak, 4));
  std::vector<double> Rinv((size_t)d * d), Rh(c->h_R);
  double halfLogDetR = 0.0;
  RB_TRY(invert_R(Rh, d, Rinv, halfLogDetR));
  double *d_R = nullptr, *d_Rinv = nullptr;
  RB_TRY(dmalloc(&d_R, (size_t)d * d));
  RB_TRY(dmalloc(&d_Rinv, (size_t)d * d));
  struct Guard { double* a; double* b; ~Guard() { hipFree(a); hipFree(b); } } guard{d_R, d_Rinv};
  HIPCHK(hipMemcpy(d_R, Rh.data(), (size_t)d * d * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_Rinv, Rinv.data(), (size_t)d * d * 8, hipMemcpyHostToDevice));

  const bool sparse = c->mdl.kind == RBPF_MODEL_SPARSE_VISUAL_2D;
  std::vector<int> pair_t, pair_j, pair_off((size_t)T + 1, 0);
  int *d_pair_t = nullptr, *d_pair_j = nullptr;
  struct PairGuard { int** a; int** b; ~PairGuard() { hipFree(*a); hipFree(*b); } } pair_guard{&d_pair_t, &d_pair_j};
  if (sparse) {
    if (info_form) { set_error("This code has only been implemented for dense features"); return RBPF_ERR_UNSUPPORTED; }   // InformationForm.m:77-80
    // observed (time, output) pairs, time-major: the future observations of step t are the suffix from pair_off[t]
    for (int ti = 0; ti < T; ++ti) {
      pair_off[ti] = (int)pair_t.size();
      for (int j = 0; j < d; ++j)
        if (!std::isnan(c->h_y[ti + (size_t)T * j])) { pair_t.push_back(ti); pair_j.push_back(j); }
    }
    pair_off[T] = (int)pair_t.size();
    const size_t Ms = (T > 1) ? (size_t)(pair_off[T] - pair_off[1]) : 0;
    if (Ms > 1023) { set_error("sparse smoother: more than 1023 future observations (particleSmoother.m:197-212 stacks them all)"); return RBPF_ERR_UNSUPPORTED; }
    s->Mmax = Ms;
    RB_TRY(dmalloc(&d_pair_t, pair_t.size()));
    RB_TRY(dmalloc(&d_pair_j, pair_j.size()));
    if (!pair_t.empty()) {
      HIPCHK(hipMemcpy(d_pair_t, pair_t.data(), pair_t.size() * sizeof(int), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d_pair_j, pair_j.data(), pair_j.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    RB_TRY(dmalloc(&s->d_S, (size_t)N * Ms * Ms));
    RB_TRY(dmalloc(&s->d_e, (size_t)N * Ms));
    RB_TRY(dmalloc(&s->d_L, (size_t)N * chol_factor_doubles((int)Ms)));
  } else if (!info_form) {
    RB_TRY(dmalloc(&s->d_Pfull, (size_t)N * n * n));
    RB_TRY(dmalloc(&s->d_G, (size_t)N * Mmax * n));
    RB_TRY(dmalloc(&s->d_S, (size_t)N * Mmax * Mmax));
    RB_TRY(dmalloc(&s->d_L, (size_t)N * chol_factor_doubles((int)Mmax)));
    if (Mmax > 1023) { set_error("covariance-form smoother supports ny*N_T <= 1023 (use the information form for long T)"); return RBPF_ERR_UNSUPPORTED; }
  } else {
    if (n > 1023) { set_error("information-form smoother supports nLin <= 1023"); return RBPF_ERR_UNSUPPORTED; }
    if (c->x0_lin_cols != 1) {
      // quirk Q5: the reference's repmat(x0_lin,1,N_P) (:109) only works for a single column
      set_error("particleSmootherInformationForm: x0_lin must be nLin x 1 (the reference repmat's it, :109)");
      return RBPF_ERR_INVALID_ARG;
    }
    {
      const int K = resolve_chol_refresh(c->mdl.kind, n, d, effective_chol_refresh(c->opt));
      s->refresh = K > 1 ? K : 0;
    }
    if (s->refresh && (sweep_slots(n) > kSweepMaxSlots || chol64_lds_bytes(n, d) > kC64MaxLds || (d != 1 && d != 3))) {
      set_error("chol_refresh > 1 supports nLin <= 575 and n_y = 1 or 3"); return RBPF_ERR_UNSUPPORTED;
    }
    s->lazy_imat = s->refresh && c->mdl.kind != RBPF_MODEL_GENERIC_DENSE;     // needs measModel on the device
    s->imat_packed = imat_storage_packed(n, d, s->refresh, s->lazy_imat);
    s->imat_len = s->imat_packed ? imat_packed_doubles(n) : (size_t)n * n;
    s->l_chunk = s->lazy_imat ? std::min(N, 4096) : N;
    // Refresh-free (r05): with K >= N_T - 1 the factors are never refactorised after t = 1 -- against the extended-precision arbiter
    // their ancestor probabilities are as close with K = 999 as with K = 32 or from scratch (6.8e-10 / 7.0e-10 / 6.7e-10 over T = 1000,
    // DESIGN.md 9) -- so no information matrix is ever read again: none is stored, the first factorisation runs chunk by chunk over
    // one buffer of l_chunk matrices.  3.6 MB of state per particle with one covariance bank (rbpf_options.inplace): the metric's full
    // N_P = 65 536 fits one 288 GB MI355X.
    if (c->opt.info_rebuild > 0 && !s->lazy_imat) {
      set_error("info_rebuild = 1 needs carried factors (chol_refresh != 1) and a recognised model family: the information matrices are rebuilt from measModel along the state history");
      return RBPF_ERR_UNSUPPORTED;
    }
    // (a window of more than 256 generations would need N_P x K x n_y x n doubles of Jacobians per refresh: such periods take the
    //  rebuild from the origin as well, whose scratch is one chunk x one segment)
    s->refresh_free = s->lazy_imat && (s->refresh >= T - 1 || c->opt.info_rebuild > 0 || s->refresh > 256);
    s->seg_len = 32;
    for (int b = 0; b < 2; ++b) {
      const size_t n_mat = s->refresh_free ? (b == 0 ? (size_t)s->l_chunk : (size_t)0) : (size_t)N;
      RB_TRY(dmalloc(&s->d_Imat[b], n_mat * s->imat_len));
      if (s->imat_packed && n_mat) HIPCHK(hipMemsetAsync(s->d_Imat[b], 0, n_mat * s->imat_len * 8, st));   // the never-written upper halves of the diagonal tiles
      RB_TRY(dmalloc(&s->d_Hb[b], (size_t)N * d * L.ldx));
      RB_TRY(dmalloc(&s->d_ivec[b], (size_t)N * L.ldx));
      RB_TRY(dmalloc(&s->d_hld[b], (size_t)N));
      RB_TRY(dmalloc(&s->d_qf[b], (size_t)N));
    }
    // factor workspaces of the 64-column kernel: one per particle -- or, with carried factors, per particle of a CHUNK of the
    // refresh (they are converted to the sweep layout chunk by chunk): 2.2 MB per particle at nLin = 515 that N_P = 32 768 has no room for
    RB_TRY(dmalloc(&s->d_L, (size_t)s->l_chunk * chol_factor_doubles(n)));
    RB_TRY(dmalloc(&s->d_ImatAdd, (size_t)n * n));
    RB_TRY(dmalloc(&s->d_ivecAdd, (size_t)n));
    RB_TRY(dmalloc(&s->d_Imat0, (size_t)n * n));
    if (s->refresh) {
      for (int b = 0; b < 2; ++b) RB_TRY(dmalloc(&s->d_Lsw[b], (size_t)N * sweep_factor_doubles(n)));
      RB_TRY(dmalloc(&s->d_W, (size_t)d * d));
      std::vector<double> Wm;
      whitening_factor(Rh, d, Wm);
      HIPCHK(hipMemcpy(s->d_W, Wm.data(), (size_t)d * d * 8, hipMemcpyHostToDevice));
      if (s->lazy_imat) {
        RB_TRY(dmalloc(&s->d_base_slot, (size_t)N));
        if (s->refresh_free) {                                             // a chunk of particles x a segment of generations at a time
          RB_TRY(dmalloc(&s->d_Xp, (size_t)s->l_chunk * s->seg_len * nN));
          RB_TRY(dmalloc(&s->d_G, (size_t)s->l_chunk * s->seg_len * d * n));
          RB_TRY(dmalloc(&s->d_marks, (size_t)((T + s->seg_len - 1) / s->seg_len) * N));
        } else {
          RB_TRY(dmalloc(&s->d_Xp, (size_t)N * s->refresh * nN));          // generations a refresh walks back: K
          RB_TRY(dmalloc(&s->d_G, (size_t)N * s->refresh * d * n));
        }
      }
    }
  }
  // information-form initial values (quirk Q5: diagonal of P0 only, :110-115)
  std::vector<double> ivec0(L.ldx, 0.0), Imat0((size_t)n * n, 0.0);
  double hld0 = 0.0, qf0 = 0.0;
  if (info_form) {
    info_initial_values(c, ivec0, Imat0, hld0);
    for (int r = 0; r < n; ++r) {                    // ivec0' * P0 * ivec0 (full P0, as :301 uses P)
      double sacc = 0.0;
      for (int cc = 0; cc < n; ++cc) sacc += c->h_P0[r + (size_t)n * cc] * ivec0[cc];
      qf0 += ivec0[r] * sacc;
    }
    const std::vector<double> Imat0s = imat_host_storage(Imat0, n, s->imat_packed);
    HIPCHK(hipMemcpy(s->d_Imat0, Imat0s.data(), Imat0s.size() * 8, hipMemcpyHostToDevice));
  }

  std::vector<double> xnk_h((size_t)nN * T);
  for (int k = 0; k < N_K; ++k) {
    RB_TRY(ctx_reset(c));
    if (info_form) {
      RB_TRY(info_begin_iteration(c, ivec0.data(), hld0, qf0, halfLogDetR, d_Rinv));
    }
    if (k > 0 && !sparse) {
      // dy_xnk = measModel(xnk)  (:120)
      if (generic) {
        std::vector<double> dyT((size_t)T * d * n), Href((size_t)T * d * n);
        if (c->cb.meas_model(c->cb.user, T, xnk_h.data(), dyT.data()) != 0) { set_error("the measModel callback failed"); return RBPF_ERR_CALLBACK; }
        for (int cc = 0; cc < n; ++cc)                   // dy(tt, a, cc) at tt + T*(a + d*cc) -> [T][d][n]
          for (int a = 0; a < d; ++a)
            for (int tt = 0; tt < T; ++tt) Href[((size_t)tt * d + a) * n + cc] = dyT[(size_t)tt + (size_t)T * (a + (size_t)d * cc)];
        HIPCHK(hipMemcpy(s->d_dyref, Href.data(), Href.size() * 8, hipMemcpyHostToDevice));
      } else
      HIPCHK(launch_meas_model(c->mdl, T, s->d_xnk, s->d_dyref, st, 1));
      if (info_form) {                                                       // :132-146
        HIPCHK(hipMemsetAsync(s->d_ImatAdd, 0, (size_t)n * n * 8, st));
        HIPCHK(hipMemsetAsync(s->d_ivecAdd, 0, (size_t)n * 8, st));
        hipLaunchKernelGGL(info_addt_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, st, n, d, 0, T,
                           1.0, s->d_dyref, d_Rinv, c->d_y, s->d_ImatAdd, s->d_ivecAdd, s->imat_packed ? 1 : 0);
        HIPCHK(hipGetLastError());
      }
    }
    for (int t = 0; t < T; ++t) {
      const double* xref = (k > 0) ? s->d_xnk + (size_t)t * nN : nullptr;
      int n_draw = N;
      if (k > 0 && t > 0) n_draw = N - 1;
      // generic family: ordinary ancestors + dynModel on the host first, in the reference's call order (:132-137)
      if (generic) RB_TRY(generic_draw_propagate(c, k, n_draw));
      if (k > 0 && t > 0) {
        // ---- ancestor weights of the reference trajectory (Steps 9-10 of Alg. 2) ----
        const double* X_prev = c->X + (size_t)(t - 1) * nN * N;
        const size_t tr_prev = c->opt.trace ? (size_t)(t - 1) * N : 0;
        const double* w_prev = c->w + tr_prev;
        const double* Lq = c->d_cholQfull + (size_t)((c->chol_pages > 1) ? t - 1 : 0) * nw * nw;
        if (drn_cb) {
          h_edyn.assign((size_t)N * nw, 0.0);
          if (c->cb.dyn_res_norm(c->cb.user, t - 1, N, xnk_h.data() + (size_t)t * nN, c->h_xn.data(), h_edyn.data()) != 0) {
            set_error("the dynResNorm callback failed"); return RBPF_ERR_CALLBACK;
          }
          HIPCHK(hipMemcpyAsync(d_edyn, h_edyn.data(), h_edyn.size() * 8, hipMemcpyHostToDevice, st));
          hipLaunchKernelGGL(anc_dyn_ext_kernel, dim3((N + 63) / 64), dim3(64), 0, st, N, nw, d_edyn, w_prev, s->d_pant_log);
          HIPCHK(hipGetLastError());
          HIPCHK(hipStreamSynchronize(st));
        } else {
        hipLaunchKernelGGL(anc_dyn_kernel, dim3((N + 63) / 64), dim3(64), 0, st, c->mdl, N, X_prev, xref,
                           c->d_odo + (size_t)(t - 1) * c->mdl.nodo, Lq, w_prev, s->d_pant_log);
        HIPCHK(hipGetLastError());
        }
        const int cur = c->cur;
        CholArgs ca;
        std::memset(&ca, 0, sizeof(ca));
        ca.d = d; ca.n = n; ca.ldx = L.ldx; ca.pant_log = s->d_pant_log; ca.status = c->d_flags;
        ca.variant = c->opt.chol_variant;
        bool skip_chol = false;
        if (sparse) {
          const int M = pair_off[T] - pair_off[t];
          skip_chol = (M == 0);                                            // no future observation: logwMeas = 0
          SparseAncArgs aa;
          aa.n = n; aa.d = d; aa.nN = nN; aa.ldx = L.ldx; aa.ldb = L.ldb; aa.M = M; aa.off = pair_off[t]; aa.szB = L.szB;
          aa.f = c->mdl.cam[0]; aa.fp = c->mdl.cam[1]; aa.pair_t = d_pair_t; aa.pair_j = d_pair_j; aa.xnk = s->d_xnk;
          aa.y = c->d_y; aa.xl = c->xl[cur]; aa.Pb = c->Pb[cur]; aa.R = c->d_R; aa.rhs = s->d_e; aa.S = s->d_S;
          HIPCHK(launch_sparse_anc(aa, N, st));
          ca.mode = 0; ca.Msz = M; ca.Lbuf = s->d_L; ca.ldL = (long)chol_factor_doubles(M);
          ca.S = s->d_S; ca.R = nullptr; ca.rhs = s->d_e; ca.jitter = c->mdl.jitter;
        } else if (!info_form) {
          const int M = d * (T - t);
          HIPCHK(launch_unpack_P(L, d, c->Pt[cur], c->Pb[cur], c->F[cur], nullptr, N, s->d_Pfull, st));
          const double* dyf = s->d_dyref + (size_t)t * d * n;              // [(T-t)*d x n] row-major (:163-166)
          GemmArgs g1{M, n, n, dyf, n, 1, 0, s->d_Pfull, 1, n, (long)((size_t)n * n), s->d_G, n, 1, (long)((size_t)M * n)};
          HIPCHK(launch_gemm(g1, N, st));                                  // G_j = dy * P_j
          GemmArgs g2{M, M, n, s->d_G, n, 1, (long)((size_t)M * n), dyf, 1, n, 0, s->d_S, 1, M, (long)((size_t)M * M)};
          HIPCHK(launch_gemm(g2, N, st));                                  // S_j = G_j * dy'
          ca.mode = 0; ca.Msz = M; ca.Lbuf = s->d_L; ca.ldL = (long)chol_factor_doubles(M);
          ca.S = s->d_S; ca.R = d_R; ca.yf = c->d_y + (size_t)t * d; ca.dyf = dyf; ca.xl = c->xl[cur];
          ca.jitter = c->mdl.jitter;
        } else {
          // the (t-1) term leaves the suffix sums (:194-201)
          hipLaunchKernelGGL(info_addt_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, st, n, d,
                             t - 1, t, -1.0, s->d_dyref, d_Rinv, c->d_y, s->d_ImatAdd, s->d_ivecAdd, s->imat_packed ? 1 : 0);
          HIPCHK(hipGetLastError());
          const bool carry = s->refresh > 1;
          const bool fresh = !carry || t == 1 || ((t - 1) % s->refresh) == 0;
          if (carry && !fresh) {
            // carried factors: one sweep per particle turns the ancestor's factor into this particle's and adds logwMeas
            // (rbpf_chol_sweep.hpp).  Host-callback models cannot rebuild Imat from the state history at a refresh, so for
            // them the exactly carried Imat is advanced every step by a copy kernel.
            if (!s->lazy_imat) {
              const int ni = s->imat_cur ^ 1;
              HIPCHK(launch_imat_gather(n, d, L.ldx, N, s->d_Imat[s->imat_cur], (long)((size_t)n * n), c->A + (size_t)(t - 1) * N,
                                        s->d_Hb[s->icur], d_Rinv, s->d_Imat[ni], st));
              s->imat_cur = ni;
            }
            SweepArgs sw;
            sw.n = n; sw.d = d; sw.ldx = L.ldx; sw.NS = sweep_slots(n); sw.tailc = sweep_tail_compact(n); sw.N = N; sw.ref_slot = N - 1;
            sw.Lold = s->d_Lsw[s->sw_cur]; sw.Lnew = s->d_Lsw[s->sw_cur ^ 1]; sw.stride = sweep_factor_doubles(n);
            sw.anc = c->A + (size_t)(t - 1) * N; sw.Hb = s->d_Hb[s->icur]; sw.Href = s->d_dyref + (size_t)(t - 1) * d * n;
            sw.W = s->d_W; sw.yt = c->d_y + (size_t)(t - 1) * d; sw.qf = s->d_qf[s->icur]; sw.hld = s->d_hld[s->icur];
            sw.pant_log = s->d_pant_log; sw.status = c->d_flags;
            sw.order = (c->order_step == t - 1) ? c->d_order : nullptr;     // generation t-1 in the order its step ran in
            // (measured r05, diagnostic build of commit 31144f4: the sweep on a second stream beside the step kernels of the same time
            //  step -- only slot N_P's ancestor depends on it -- 7.39 -> 7.15 ms per step: not worth splitting that slot out)
            HIPCHK(launch_chol_sweep(sw, st));
            s->sw_cur ^= 1;
            skip_chol = true;
          } else if (carry && s->lazy_imat) {
            // refresh: Imat of generation t-1 from the base generation + G'G along the ancestral paths, then factorise it
            if (s->refresh_free) {
              // No information matrix is stored: Imat_i(t-1) = Imat0 + sum over the WHOLE ancestral path of H' R^-1 H (:334's terms, in
              // another order of summation), accumulated segment by segment into a chunk buffer, then chol(Imat + ImatAddt) out of it
              // and the factors into the sweep layout.  Cost grows with t; the marks make every segment's walk O(seg_len).
              const int S = s->seg_len, n_gen = t, n_seg = (n_gen + S - 1) / S;        // generations 0 .. t-1; segment j = [hi_j - len_j, hi_j), hi_0 = t
              hipLaunchKernelGGL(origin_marks_kernel, dim3((N + 127) / 128), dim3(128), 0, st, N, t - 1, S, c->A, s->d_marks);
              HIPCHK(hipGetLastError());
              for (int p0 = 0; p0 < N; p0 += s->l_chunk) {
                const int cnt = std::min(s->l_chunk, N - p0);
                for (int j = 0; j < n_seg; ++j) {
                  const int hi = n_gen - j * S, lo = std::max(0, hi - S), len = hi - lo;
                  hipLaunchKernelGGL(origin_segment_kernel, dim3((cnt + 127) / 128), dim3(128), 0, st, N, nN, p0, cnt, lo, hi, c->A, c->X,
                                     s->d_marks + (size_t)j * N, s->d_Xp);
                  HIPCHK(hipGetLastError());
                  HIPCHK(launch_meas_model(c->mdl, cnt * len, s->d_Xp, s->d_G, st, 1));
                  const size_t rows = (size_t)cnt * len;
                  hipLaunchKernelGGL(sweep_whiten_kernel, dim3((unsigned)((rows * n + 255) / 256)), dim3(256), 0, st, rows, d, n, s->d_W, s->d_G);
                  HIPCHK(hipGetLastError());
                  const long Kd = (long)len * d;
                  GemmArgs gg{n, n, (int)Kd, s->d_G, 1, n, Kd * n, s->d_G, n, 1, Kd * n, s->d_Imat[0], 1, n, (long)s->imat_len};
                  gg.lower = 1; gg.packed = s->imat_packed ? 1 : 0;
                  if (j == 0) { gg.add = s->d_Imat0; gg.add_stride = 0L; gg.add_idx = nullptr; }
                  else gg.add_self = 1;
                  HIPCHK(launch_gemm(gg, cnt, st));
                }
                RB_TRY(refresh_factorise(c, d_Rinv, s->d_pant_log, cnt, st, p0, s->d_Imat[0]));
              }
              s->imat_valid = false; s->base_gen = t - 1;
              s->sw_cur ^= 1;
              skip_chol = true;
            } else {
            const int t0 = s->base_gen, Kp = t - 1 - t0;
            if (Kp < 1 || Kp > s->refresh) { set_error("internal: refresh window"); return RBPF_ERR_STATE; }
            const int ni = s->imat_valid ? (s->imat_cur ^ 1) : 0;
            hipLaunchKernelGGL(sweep_path_kernel, dim3((N + 127) / 128), dim3(128), 0, st, N, nN, Kp, t - 1, t0, c->A, c->X, s->d_base_slot, s->d_Xp);
            HIPCHK(hipGetLastError());
            HIPCHK(launch_meas_model(c->mdl, N * Kp, s->d_Xp, s->d_G, st, 1));
            const size_t rows = (size_t)N * Kp;
            hipLaunchKernelGGL(sweep_whiten_kernel, dim3((unsigned)((rows * n + 255) / 256)), dim3(256), 0, st, rows, d, n, s->d_W, s->d_G);
            HIPCHK(hipGetLastError());
            const long Kd = (long)Kp * d;
            GemmArgs gg{n, n, (int)Kd, s->d_G, 1, n, Kd * n, s->d_G, n, 1, Kd * n, s->d_Imat[ni], 1, n, (long)s->imat_len};
            gg.lower = 1;                                                      // the factorisation reads the block-lower part
            gg.packed = s->imat_packed ? 1 : 0;
            gg.add = t0 < 0 ? s->d_Imat0 : s->d_Imat[s->imat_cur];             // + the base matrix, in the epilogue
            gg.add_stride = t0 < 0 ? 0L : (long)s->imat_len;
            gg.add_idx = t0 < 0 ? (const int*)nullptr : s->d_base_slot;
            HIPCHK(launch_gemm(gg, N, st));                                   // Imat = base + G' G
            s->imat_cur = ni; s->imat_valid = true; s->base_gen = t - 1;
            RB_TRY(refresh_factorise(c, d_Rinv, s->d_pant_log, N, st));        // chol(Imat + ImatAddt), chunk by chunk -> sweep layout
            s->sw_cur ^= 1;
            skip_chol = true;
            }
          } else {
            RB_TRY(info_fill_chol_args(c, ca, d_Rinv, c->A + (size_t)(t - 1) * N));   // ancestors of the generation t-1
            if (carry) ca.variant = 64;            // the refresh reads the factor back in the 64-column kernel's layout
          }
        }
        if (!skip_chol) HIPCHK(launch_chol(ca, N, ca.mode == 1 ? d : 0, st));
        HIPCHK(hipGetLastError());
        if (info_form && s->refresh > 1 && !skip_chol) {                    // fresh factors -> sweep layout (host-callback models)
          HIPCHK(launch_sweep_from_chol64(n, N, s->d_L, chol_factor_doubles(n), s->d_Lsw[s->sw_cur ^ 1], sweep_factor_doubles(n), st));
          s->sw_cur ^= 1;
        }
        // normalise (:236-238), sample ai(N_P) (:241)
        RB_TRY(normalise_draw_one(c, N, t, k, s->d_pant_log, s->d_pant + (c->opt.trace ? ((size_t)k * T + t) * N : 0), s->d_wc2, N - 1,
                                  c->d_U ? c->d_U + ((size_t)k * (T - 1) + (t - 1)) * N : nullptr, c->A + (size_t)t * N, st));
      }
      if (generic) RB_TRY(generic_finish_inputs(c, k > 0 ? xnk_h.data() + (size_t)t * nN : nullptr));
      const int st_step = info_form ? info_step(c, k, t, xref, n_draw, d_Rinv) : ctx_step(c, k, xref, n_draw, nullptr);
      if (generic) { c->ext_xn = nullptr; c->ext_H = nullptr; }
      RB_TRY(st_step);
    }
    // ---- ak = sample(w); xnk = xn_traj(:,ak,:) (:346-354) ----
    {
      const size_t tr_last = c->opt.trace ? (size_t)(T - 1) * N : 0;
      double* d_uf = c->d_scal;
      if (c->rng_mode == RBPF_RNG_REPLAY) {
        HIPCHK(hipMemcpyAsync(d_uf, &c->h_Ufin[k], 8, hipMemcpyHostToDevice, st));
      }
      SearchArgs sa;
      sa.N = N; sa.n_draw = 1; sa.t = T; sa.wc = c->wc; sa.rng_mode = c->rng_mode; sa.k_iter = k; sa.slot0 = 0;
      sa.U = d_uf; sa.seed = c->seed; sa.ai = s->d_ak; sa.overflow = c->d_flags + 1;
      sa.u_is_scalar = 1;
      sa.approx = 1; sa.ambiguous = c->d_flags + 4; sa.w = c->w + (c->opt.trace ? (size_t)(T - 1) * N : 0); sa.wc_exact = c->wc;
      HIPCHK(launch_search(sa, st));
      HIPCHK(launch_resample_fixup(sa, st));
      (void)tr_last;
      HIPCHK(launch_backtrace(N, nN, T, c->X, c->A, s->d_ak, 1, s->d_xnk, st));
      int ak = 0;
      HIPCHK(hipMemcpyAsync(&ak, s->d_ak, 4, hipMemcpyDeviceToHost, st));
      RB_TRY(ctx_check_flags(c));
      const int cur = c->cur;
      HIPCHK(hipMemcpy(xnk_h.data(), s->d_xnk, (size_t)nN * T * 8, hipMemcpyDeviceToHost));
      if (out->XNK) std::memcpy(out->XNK + (size_t)k * nN * T, xnk_h.data(), (size_t)nN * T * 8);
      (void)cur;
      if (out->XLK) HIPCHK(hipMemcpy(out->XLK + (size_t)k * n, c->xl[c->xcur] + (size_t)ak * L.ldx, (size_t)n * 8, hipMemcpyDeviceToHost));
      if (out->PK) {
        double* dP = nullptr;
        RB_TRY(dmalloc(&dP, (size_t)n * n));
        hipError_t e = (ctx_unpack(c, s->d_ak, 1, dP) == RBPF_OK) ? hipSuccess : hipErrorUnknown;   // pending downdates applied
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipMemcpy(out->PK + (size_t)k * n * n, dP, (size_t)n * n * 8, hipMemcpyDeviceToHost);
        hipFree(dP);
        HIPCHK(e);
      }
      if (out->trace_ak) out->trace_ak[k] = ak;
      if (c->opt.trace) {
        if (out->trace_logw) HIPCHK(hipMemcpy(out->trace_logw + (size_t)k * N * T, c->logw, (size_t)N * T * 8, hipMemcpyDeviceToHost));
        if (out->trace_w) HIPCHK(hipMemcpy(out->trace_w + (size_t)k * N * T, c->w, (size_t)N * T * 8, hipMemcpyDeviceToHost));
        if (out->trace_ai) HIPCHK(hipMemcpy(out->trace_ai + (size_t)k * N * T, c->A, (size_t)N * T * 4, hipMemcpyDeviceToHost));
        if (out->trace_paNt && k > 0)
          HIPCHK(hipMemcpy(out->trace_paNt + (size_t)k * N * T, s->d_pant + (size_t)k * T * N, (size_t)N * T * 8, hipMemcpyDeviceToHost));
      }
      RB_TRY(ctx_call_on_step(c, k, true));              // particleSmoother.m:360-362
    }
  }
  return RBPF_OK;
}

static int info_begin_iteration(rbpf_ctx* c, const double* ivec0, double hld0, double, double, const double*) {
  SmootherState* s = c->sm;
  const Layout& L = c->lay;
  if (!s->d_ivec0) {
    RB_TRY(dmalloc(&s->d_ivec0, (size_t)L.ldx));
    RB_TRY(dmalloc(&s->d_hld0, 1));
  }
  HIPCHK(hipMemcpyAsync(s->d_ivec0, ivec0, (size_t)L.ldx * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(s->d_hld0, &hld0, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));       // host sources are stack / caller memory
  s->icur = 0;
  s->imat_cur = 0;
  s->imat_valid = false;
  s->base_gen = -1;
  return RBPF_OK;
}

// Arguments of the information-form ancestor-weight factorisation of the step about to run (t >= 1).  The kernel
// also forms the information matrices of the current generation, Imat(:,:,i) = Imat(:,:,ai(i)) + dyi'/R*dyi
// (:170,:253,:334), from the previous generation's bank (entry anc[i]; generation 0 starts from Imat0) and stores
// them into the other bank: the gather `Imat = Imat(:,:,ai)` costs one extra write instead of a copy kernel.
static int info_fill_chol_args(rbpf_ctx* c, CholArgs& ca, const double* d_Rinv, const int* anc) {
  SmootherState* s = c->sm;
  const int n = c->mdl.n;
  const int ni = s->imat_valid ? (s->imat_cur ^ 1) : 0;
  ca.mode = 1; ca.Msz = n; ca.Lbuf = s->d_L; ca.ldL = (long)chol_factor_doubles(n);
  ca.Imat = s->imat_valid ? s->d_Imat[s->imat_cur] : s->d_Imat0;
  ca.imat_stride = s->imat_valid ? (long)s->imat_len : 0;
  ca.imat_anc = s->imat_valid ? anc : nullptr;
  ca.ImatOut = s->d_Imat[ni]; ca.imat_out_stride = (long)s->imat_len; ca.imat_packed = s->imat_packed ? 1 : 0;
  ca.Hb = s->d_Hb[s->icur]; ca.Rinv = d_Rinv; ca.ImatAdd = s->d_ImatAdd; ca.ivec = s->d_ivec[s->icur];
  ca.ivecAdd = s->d_ivecAdd; ca.qf = s->d_qf[s->icur]; ca.hld = s->d_hld[s->icur];
  s->imat_cur = ni;                 // after the launch the new bank is the current one
  s->imat_valid = true;
  // nothing reads the factors of this launch afterwards unless they are carried on (chol_refresh): let the 64-column kernel
  // run persistent workgroups over one workspace slot each (rbpf_chol64.hpp)
  ca.l_slots = (s->refresh > 1) ? 0 : chol64_workspace_slots(n);
  return RBPF_OK;
}

// One information-form time step: the fused step kernel with the extra ivec / halfLogDetP state,
// followed (k>1) by the gather + pending update of the information matrices (:170,:186,:253,:334).
static int info_step(rbpf_ctx* c, int k, int t, const double* xref, int n_draw, const double* d_Rinv) {
  SmootherState* s = c->sm;
  const int N = c->N, n = c->mdl.n, d = c->mdl.d;
  const Layout& L = c->lay;
  const int oc = s->icur, nc = (t == 0) ? 0 : (oc ^ 1);
  InfoStep is;
  if (t == 0) { is.ivec_old = s->d_ivec0; is.ivec_old_stride = 0; is.hld_old = s->d_hld0; is.hld_old_stride = 0; }
  else { is.ivec_old = s->d_ivec[oc]; is.ivec_old_stride = (size_t)L.ldx; is.hld_old = s->d_hld[oc]; is.hld_old_stride = 1; }
  is.ivec_new = s->d_ivec[nc]; is.hld_new = s->d_hld[nc]; is.qf_new = s->d_qf[nc]; is.Hb_new = s->d_Hb[nc];
  RB_TRY(ctx_step(c, k, xref, n_draw, &is));
  (void)N; (void)n; (void)d; (void)d_Rinv;
  s->icur = nc;
  return RBPF_OK;
}

extern "C" int32_t rbpf_chol_refresh_resolve(int32_t model_kind, int32_t n_lin, int32_t n_y, int32_t requested) {
  return resolve_chol_refresh(model_kind, n_lin, n_y, requested);
}

extern "C" int rbpf_particle_smoother(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng,
                                      const rbpf_options* opt, int32_t N_K, int32_t info_form, rbpf_smoother_out* out) {
  if (!out || N_K < 1) { set_error("bad smoother arguments"); return RBPF_ERR_INVALID_ARG; }
  RB_TRY(options_ok(opt));
  if (wants_multi(opt)) return multi_particle_smoother(model, prob, rng, opt, N_K, info_form, out);   // sharded over several GPUs
  rbpf_ctx* c = nullptr;
  int st = ctx_create(model, prob, rng, opt, true, N_K, &c);
  if (st == RBPF_OK) st = smoother_run(c, N_K, info_form, out);
  if (st == RBPF_ERR_OUT_OF_MEMORY && info_form && prob) {
    // say what would fit: the information-form state without stored information matrices and with one covariance bank
    const double n = (double)prob->n_lin, mb = (8.0 * (0.57 * n * n + 2.0 * 0.52 * (n + 1) * (n + 1))) / 1e6;
    set_error(std::string(rbpf_last_error()) + " -- the information-form smoother did not fit this device with these options.  Its smallest "
              "footprint: storage = 2 (where supported), lazy_depth >= 2, inplace = 1 and chol_refresh >= N_T (or info_rebuild = 1): about " +
              std::to_string((int)(mb * 10) / 10.0).substr(0, 4) + " MB of state per particle at this nLin (include/rbpf.h, rbpf_options)");
  }
  if (!c) return st;
  if (st != RBPF_OK) { ctx_free(c); return st; }
  if (st == RBPF_OK) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) st = hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
  }
  ctx_free(c);
  return st;
}

// =============================================================================================================
// Particle-sharded information-form smoother: one process per GPU, collectives issued by the host
// mirror (multigpu.py) between these calls.  Every rank keeps N_local particles with their extra information-form
// state; the ancestor weights of the reference trajectory (particleSmootherInformationForm.m:205-247) are computed
// on the rank that holds each particle, all-gathered (N doubles), and normalised / sampled identically on every
// rank, so a W-rank run equals the single-GPU smoother with N = W * N_local particles bit for bit.
// =============================================================================================================
size_t rbpf::smoother_record_matrix_doubles(int n, int d, int chol_refresh) {      // chol_refresh: the resolved K
  return chol_refresh > 1 ? sweep_factor_doubles(n) : imat_storage_packed(n, d, chol_refresh, true) ? imat_packed_doubles(n) : (size_t)n * n;
}

int rbpf::shard_smoother_pack_info(rbpf_ctx* c, const int* d_idx, int count) {
  SmootherState* s = c->sm;
  ShardState* sh = c->sh;
  if (!s || !sh || count <= 0) return RBPF_OK;
  const int n = c->mdl.n, d = c->mdl.d;
  const Layout& L = c->lay;
  // iteration 1 never reads Imat (:157: k > 1 only); later the ancestor-weight factorisation of this step has just
  // stored the matrices of the current generation (own updates included), so the records carry them complete
  const bool with_imat = sh->k_iter > 0;
  const bool carry = s->refresh > 1;
  if (with_imat && !(carry ? s->sw_valid : s->imat_valid)) { set_error("pack before the ancestor weights of this step"); return RBPF_ERR_STATE; }
  // carried factors: the record takes the particle's factor (sweep layout) along instead of Imat, which is rebuilt at refreshes
  const double* im = !with_imat ? nullptr : (carry ? s->d_Lsw[s->sw_cur] : s->d_Imat[s->imat_cur]);
  const size_t stride = carry ? sweep_factor_doubles(n) : s->imat_len;
  hipLaunchKernelGGL(pack_info_kernel, dim3(count, with_imat ? 8 : 1), dim3(256), 0, c->stream, n, d, L.ldx, d_idx,
                     s->d_ivec[s->icur], s->d_hld[s->icur], s->d_Hb[s->icur], im, stride, stride, sh->send_rec, sh->recsz,
                     sh->rec_off_I, sh->rec_off_hld, sh->rec_off_Hb, sh->rec_off_Imat);
  HIPCHK(hipGetLastError());
  return RBPF_OK;
}

extern "C" {

int rbpf_shard_smoother_create(const rbpf_model* model, const rbpf_problem* prob, const rbpf_rng* rng, const rbpf_options* opt,
                               int32_t N_K, int32_t rank, int32_t world, rbpf_ctx** out) {
  if (N_K < 1 || !out) { set_error("bad smoother arguments"); return RBPF_ERR_INVALID_ARG; }
  rbpf_ctx* c = nullptr;
  RB_TRY(shard_create_impl(model, prob, rng, opt, rank, world, true, N_K, &c));
  std::unique_ptr<rbpf_ctx, void (*)(rbpf_ctx*)> guard(c, [](rbpf_ctx* p) { ctx_free(p); });
  const int N = c->N, T = c->T, nN = c->mdl.nN, n = c->mdl.n, d = c->mdl.d, nw = c->mdl.nw;
  const Layout& L = c->lay;
  if (!c->mdl.use_dyn_res_norm && nw != nN) {
    set_error("isempty(dynResNorm): the additive default (particleSmoother.m:176) needs size(Q,1) == nNonLin");
    return RBPF_ERR_INVALID_ARG;
  }
  if (n > 1023) { set_error("information-form smoother supports nLin <= 1023"); return RBPF_ERR_UNSUPPORTED; }
  SmootherState* s = new SmootherState();
  c->sm = s;
  {
    const int K = resolve_chol_refresh(c->mdl.kind, n, d, effective_chol_refresh(c->opt));
    s->refresh = K > 1 ? K : 0;
  }
  if (s->refresh && c->mdl.kind == RBPF_MODEL_GENERIC_DENSE) { set_error("chol_refresh > 1 in the sharded smoother needs measModel on the device"); return RBPF_ERR_UNSUPPORTED; }
  if (c->opt.info_rebuild > 0 && s->refresh < T - 1) { set_error("info_rebuild = 1 with refreshes is a single-device option (the sharded smoother exchanges stored information matrices at its refreshes; chol_refresh >= N_T - 1 never refreshes and stores none)"); return RBPF_ERR_UNSUPPORTED; }
  if (s->refresh > 256 && s->refresh < T - 1) { set_error("sharded smoother: chol_refresh between 257 and N_T - 2 is not supported (choose <= 256, or >= N_T - 1: never refresh)"); return RBPF_ERR_UNSUPPORTED; }
  s->lazy_imat = s->refresh > 1;
  // chol_refresh >= N_T - 1: the factors are never refactorised after t = 1 -- no information matrix is stored or exchanged, the first
  // factorisation runs chunk by chunk (as in the single-device smoother, smoother_run)
  s->refresh_free = s->lazy_imat && s->refresh >= T - 1;
  s->seg_len = 32;
  RB_TRY(dmalloc(&s->d_xnk, (size_t)T * nN));
  RB_TRY(dmalloc(&s->d_dyref, (size_t)T * d * n));
  RB_TRY(dmalloc(&s->d_ak, 4));
  s->imat_packed = imat_storage_packed(n, d, s->refresh, true);
  s->imat_len = s->imat_packed ? imat_packed_doubles(n) : (size_t)n * n;
  s->l_chunk = s->lazy_imat ? std::min(N, 4096) : N;
  for (int b = 0; b < 2; ++b) {
    const size_t n_mat = s->refresh_free ? (b == 0 ? (size_t)s->l_chunk : (size_t)0) : (size_t)N;
    RB_TRY(dmalloc(&s->d_Imat[b], n_mat * s->imat_len));
    if (s->imat_packed && n_mat) HIPCHK(hipMemset(s->d_Imat[b], 0, n_mat * s->imat_len * 8));
    RB_TRY(dmalloc(&s->d_Hb[b], (size_t)N * d * L.ldx));
    RB_TRY(dmalloc(&s->d_ivec[b], (size_t)N * L.ldx));
    RB_TRY(dmalloc(&s->d_hld[b], (size_t)N));
    RB_TRY(dmalloc(&s->d_qf[b], (size_t)N));
  }
  RB_TRY(dmalloc(&s->d_L, (size_t)s->l_chunk * chol_factor_doubles(n)));
  RB_TRY(dmalloc(&s->d_ImatAdd, (size_t)n * n));
  RB_TRY(dmalloc(&s->d_ivecAdd, (size_t)n));
  RB_TRY(dmalloc(&s->d_Imat0, (size_t)n * n));
  RB_TRY(dmalloc(&s->d_Rinv, (size_t)d * d));
  std::vector<double> Rinv((size_t)d * d), Imat0((size_t)n * n, 0.0);
  double halfLogDetR = 0.0;
  RB_TRY(invert_R(c->h_R, d, Rinv, halfLogDetR));
  s->h_ivec0.assign(L.ldx, 0.0);
  info_initial_values(c, s->h_ivec0, Imat0, s->hld0);
  HIPCHK(hipMemcpy(s->d_Rinv, Rinv.data(), (size_t)d * d * 8, hipMemcpyHostToDevice));
  {
    const std::vector<double> Imat0s = imat_host_storage(Imat0, n, s->imat_packed);
    HIPCHK(hipMemcpy(s->d_Imat0, Imat0s.data(), Imat0s.size() * 8, hipMemcpyHostToDevice));
  }
  if (s->refresh) {
    // carried ancestor-weight factors (rbpf_chol_sweep.hpp): factor banks, the buffers of the refresh from the state history,
    // and the exchange buffers for base matrices that sit on another rank
    if (sweep_slots(n) > kSweepMaxSlots || chol64_lds_bytes(n, d) > kC64MaxLds || (d != 1 && d != 3)) { set_error("chol_refresh > 1 supports nLin <= 575 and n_y = 1 or 3"); return RBPF_ERR_UNSUPPORTED; }
    ShardState* sh = c->sh;
    for (int b = 0; b < 2; ++b) RB_TRY(dmalloc(&s->d_Lsw[b], (size_t)N * sweep_factor_doubles(n)));
    RB_TRY(dmalloc(&s->d_W, (size_t)d * d));
    std::vector<double> Wm;
    whitening_factor(c->h_R, d, Wm);
    HIPCHK(hipMemcpy(s->d_W, Wm.data(), (size_t)d * d * 8, hipMemcpyHostToDevice));
    RB_TRY(dmalloc(&s->d_base_slot, (size_t)N));
    const size_t Kwin = s->refresh_free ? 1 : (size_t)s->refresh;        // generations a refresh walks back
    RB_TRY(dmalloc(&s->d_Xp, (size_t)N * Kwin * nN));
    RB_TRY(dmalloc(&s->d_G, (s->refresh_free ? (size_t)s->l_chunk : (size_t)N) * Kwin * d * n));
    RB_TRY(dmalloc(&s->d_owner_now, (size_t)sh->Nglob));
    RB_TRY(dmalloc(&s->d_base_gid, (size_t)sh->Nglob));
    RB_TRY(dmalloc(&s->d_base_loc, (size_t)sh->Nglob));
    // every particle needs one base matrix at most; half of the local particles importing theirs is far beyond what the
    // owner-computes placement produces between two refreshes (identical on every rank: a function of the options only)
    // (a starting value: rbpf_shard_smoother_refresh_reserve grows the buffers when a refresh needs more; exchange_capacity < 0 asks for
    //  a small start, which is how the tests reach the growth path)
    s->rf_cap = (world > 1) ? std::min<size_t>((size_t)N, std::max<size_t>(2 * sh->step_cap, c->opt.exchange_capacity < 0 ? 1 : 64)) : 1;
    RB_TRY(dmalloc(&s->d_rf_send, s->rf_cap * s->imat_len));
    RB_TRY(dmalloc(&s->d_rf_recv, s->rf_cap * s->imat_len));
    RB_TRY(dmalloc(&s->d_rf_idx, s->rf_cap));
  }
  guard.release();
  *out = c;
  return RBPF_OK;
}

int rbpf_shard_smoother_views_get(rbpf_ctx* c, rbpf_shard_smoother_views* v) {
  if (!c || !c->sh || !c->sm || !v) { set_error("not a sharded smoother context"); return RBPF_ERR_INVALID_ARG; }
  v->anc_local = c->sh->anc_local; v->anc_gather = c->sh->anc_gather;
  v->refresh_send = c->sm->d_rf_send; v->refresh_recv = c->sm->d_rf_recv;
  v->refresh_capacity = (int64_t)c->sm->rf_cap; v->matrix_doubles = (int64_t)c->sm->imat_len;
  return RBPF_OK;
}

// Start of CPF-AS iteration k (particleSmootherInformationForm.m:96-146): rewind, information-form initial values,
// and for k > 1 the measurement Jacobians along the reference trajectory with their suffix sums.
int rbpf_shard_smoother_begin(rbpf_ctx* c, int32_t k) {
  if (!c || !c->sh || !c->sm) { set_error("not a sharded smoother context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  SmootherState* s = c->sm;
  ShardState* sh = c->sh;
  if (k < 0 || k >= c->N_K) { set_error("iteration out of range"); return RBPF_ERR_INVALID_ARG; }
  const int T = c->T, n = c->mdl.n, d = c->mdl.d;
  RB_TRY(ctx_reset(c));
  sh->t_norm = 0; sh->placed = false; sh->plan_ready = false; sh->host_planned = false; sh->gid_cur = 0; sh->cur_gid = nullptr;
  sh->rec_used = 0; sh->plan_recv = 0; sh->k_iter = k;
  std::fill(sh->rec_used_all.begin(), sh->rec_used_all.end(), 0);
  RB_TRY(info_begin_iteration(c, s->h_ivec0.data(), s->hld0, 0.0, 0.0, s->d_Rinv));
  s->sw_valid = false; s->rf_stage = 0;
  if (k > 0) {
    HIPCHK(launch_meas_model(c->mdl, T, s->d_xnk, s->d_dyref, c->stream, 1));           // :120
    HIPCHK(hipMemsetAsync(s->d_ImatAdd, 0, (size_t)n * n * 8, c->stream));
    HIPCHK(hipMemsetAsync(s->d_ivecAdd, 0, (size_t)n * 8, c->stream));
    hipLaunchKernelGGL(info_addt_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, c->stream, n, d, 0, T,
                       1.0, s->d_dyref, s->d_Rinv, c->d_y, s->d_ImatAdd, s->d_ivecAdd, s->imat_packed ? 1 : 0);   // :132-146
    HIPCHK(hipGetLastError());
  }
  return RBPF_OK;
}

// After the all_gather of the forward bank: global weights of the finished step and (want_draw) the ancestors of the
// ordinary slots of the next one -- N - 1 of them when slot N - 1 carries the reference trajectory (:160-166).
int rbpf_shard_smoother_normalise(rbpf_ctx* c, int32_t want_draw) {
  if (!c || !c->sh || !c->sm) { set_error("not a sharded smoother context"); return RBPF_ERR_INVALID_ARG; }
  ShardState* sh = c->sh;
  const int n_draw = (sh->k_iter > 0) ? sh->Nglob - 1 : sh->Nglob;
  // the ancestors stay on the device (the device planner reads them there); the next host synchronisation is the plan's
  return shard_normalise_impl(c, nullptr, want_draw ? draw_only_tag() : nullptr, sh->k_iter, n_draw);
}

// k > 1, t > 1: the measurement part of my particles' ancestor log-weights for the step about to run (:205-236) -> anc_local
// [N_local] (physical order), BEFORE the all_gather of the forward bank, which carries it along.  Synchronises unless async.
// Start of the MEASUREMENT part of my particles' ancestor log-weights (logwMeas, :205-236) in anc_local -- the extra row of the
// forward bank, so that one all_gather carries it with the states and log-weights -- and the (t-1) term out of the suffix
// sums.  It needs nothing from other ranks; the dynResNorm part and log w are added for all N particles after the gather
// (rbpf_shard_smoother_anc_sample), in the reference's order of summation.
static int shard_anc_meas_begin(rbpf_ctx* c) {
  SmootherState* s = c->sm;
  ShardState* sh = c->sh;
  const int t = c->t, k = sh->k_iter, n = c->mdl.n, d = c->mdl.d;
  if (k < 1 || t < 1 || t >= c->T) { set_error("ancestor weights need k > 0 and 0 < t < N_T"); return RBPF_ERR_STATE; }
  hipStream_t st = c->stream;
  HIPCHK(hipMemsetAsync(sh->anc_local, 0, (size_t)sh->Nloc * sizeof(double), st));
  // the (t-1) term leaves the suffix sums (:194-201)
  hipLaunchKernelGGL(info_addt_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, st, n, d, t - 1, t, -1.0,
                     s->d_dyref, s->d_Rinv, c->d_y, s->d_ImatAdd, s->d_ivecAdd, s->imat_packed ? 1 : 0);
  HIPCHK(hipGetLastError());
  return RBPF_OK;
}

static bool shard_refresh_due(const SmootherState* s, int t) { return s->refresh > 1 && (t == 1 || ((t - 1) % s->refresh) == 0); }

int rbpf_shard_smoother_anc_weights(rbpf_ctx* c) {
  if (!c || !c->sh || !c->sm) { set_error("not a sharded smoother context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  SmootherState* s = c->sm;
  ShardState* sh = c->sh;
  const int t = c->t, N = sh->Nloc, n = c->mdl.n, d = c->mdl.d;
  hipStream_t st = c->stream;
  if (shard_refresh_due(s, t)) { set_error("carried factors: this step refreshes them (rbpf_shard_smoother_refresh_begin / _pack / _end)"); return RBPF_ERR_STATE; }
  RB_TRY(shard_anc_meas_begin(c));
  if (s->refresh > 1) {
    // carried factors: one sweep turns the ancestor's factor -- bank entry or received record -- into this particle's
    if (!s->sw_valid || !sh->placed) { set_error("carried factors: no factor bank for the previous generation"); return RBPF_ERR_STATE; }
    SweepArgs sw;
    sw.n = n; sw.d = d; sw.ldx = c->lay.ldx; sw.NS = sweep_slots(n); sw.tailc = sweep_tail_compact(n); sw.N = N; sw.ref_slot = -1;
    sw.Lold = s->d_Lsw[s->sw_cur]; sw.Lnew = s->d_Lsw[s->sw_cur ^ 1]; sw.stride = sweep_factor_doubles(n);
    sw.anc = sh->pb.anc_bank; sw.order = nullptr;             // the planner lays the generation out in ancestor order
    sw.Hb = s->d_Hb[s->icur]; sw.Href = s->d_dyref + (size_t)(t - 1) * d * n;
    sw.W = s->d_W; sw.yt = c->d_y + (size_t)(t - 1) * d; sw.qf = s->d_qf[s->icur]; sw.hld = s->d_hld[s->icur];
    sw.pant_log = sh->anc_local; sw.status = c->d_flags;
    sw.rec = sh->recv_rec; sw.rec_stride = sh->recsz; sw.rec_off = sh->rec_off_Imat; sw.n_bank = N;
    sw.slot_ids = sh->pb.slot_ids; sw.ref_logical = sh->Nglob - 1;
    HIPCHK(launch_chol_sweep(sw, st));
    s->sw_cur ^= 1;
  } else {
    CholArgs ca;
    std::memset(&ca, 0, sizeof(ca));
    ca.d = d; ca.n = n; ca.ldx = c->lay.ldx; ca.pant_log = sh->anc_local; ca.status = c->d_flags;
    ca.variant = c->opt.chol_variant;
    RB_TRY(info_fill_chol_args(c, ca, s->d_Rinv, sh->pb.anc_bank));     // plan of the step that made this generation
    ca.n_bank_local = N; ca.rec = sh->recv_rec; ca.rec_stride = sh->recsz; ca.rec_off_Imat = sh->rec_off_Imat;
    HIPCHK(launch_chol(ca, N, d, st));
    HIPCHK(hipGetLastError());
  }
  if (!sh->async) HIPCHK(hipStreamSynchronize(st));
  return RBPF_OK;
}

// ---- refresh of the carried factors (chol_refresh = K: the steps t = 1 and (t - 1) % K == 0 of an iteration k > 0) -------------
// The information matrix of every particle of generation t-1 is rebuilt from the last materialised generation ("base") and the
// measurement Jacobians along its ancestral path (rbpf_chol_sweep.hpp), then factorised.  The walk runs over the replicated
// global history; the base matrix of a lineage sits on the rank that held that ancestor when the base was materialised, so the
// host mirror fetches the ones on other ranks with one all_to_all between _pack and _end (unique per destination and matrix;
// the plan is a function of the two replicated tables this call returns, hence identical on every rank).
int rbpf_shard_smoother_refresh_begin(rbpf_ctx* c, int32_t* owner_now, int32_t* base_loc) {
  if (!c || !c->sh || !c->sm || !owner_now || !base_loc) { set_error("not a sharded smoother context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  SmootherState* s = c->sm;
  ShardState* sh = c->sh;
  const int t = c->t, nN = c->mdl.nN;
  if (!shard_refresh_due(s, t)) { set_error("no refresh is due at this step"); return RBPF_ERR_STATE; }
  if (sh->t_norm < t) { set_error("a refresh walks the state history: gather + normalise the finished step first"); return RBPF_ERR_STATE; }
  RB_TRY(shard_anc_meas_begin(c));
  const int t0 = s->base_gen, Kp = t - 1 - t0;
  if (Kp < 1 || Kp > s->refresh) { set_error("internal: refresh window"); return RBPF_ERR_STATE; }
  hipStream_t st = c->stream;
  hipLaunchKernelGGL(shard_path_kernel, dim3((sh->Nglob + 127) / 128), dim3(128), 0, st, sh->Nglob, sh->Nloc, sh->rank, nN, Kp, t - 1, t0,
                     sh->Ahist, sh->Xhist, sh->placed ? sh->cur_gid : (const int*)nullptr, s->d_base_gid, s->d_owner_now, s->d_base_loc, s->d_Xp);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(owner_now, s->d_owner_now, (size_t)sh->Nglob * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(base_loc, s->d_base_loc, (size_t)sh->Nglob * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  s->rf_stage = 1; s->rf_Kp = Kp;
  return RBPF_OK;
}

// The refresh buffers GROW on demand like the record buffers (rbpf_options.exchange_capacity <= 0): `count` = the largest number of
// matrices any rank sends or receives at this refresh -- a function of the replicated plan, so every rank calls this with the same
// value and enlarges by the same rule (at least `count`, at least twice the old capacity, at most (world - 1) x N_local: a matrix is packed once per destination) without communicating.
// Nothing in the buffers outlives a refresh.  exchange_capacity > 0 is a hard limit: RBPF_ERR_OUT_OF_MEMORY on every rank.
int rbpf_shard_smoother_refresh_reserve(rbpf_ctx* c, int64_t count) {
  if (!c || !c->sh || !c->sm || count < 0) { set_error("not a sharded smoother context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  SmootherState* s = c->sm;
  if ((size_t)count <= s->rf_cap) return RBPF_OK;
  // a rank RECEIVES one base matrix per particle at most, but SENDS a matrix once per destination rank that asked for it: the plan's
  // keys are unique per (destination, source, slot), so the send side can reach (W - 1) * N_local
  const size_t hard = (size_t)c->sh->Nloc * (size_t)std::max(1, c->sh->world - 1);
  if (c->opt.exchange_capacity > 0) {
    set_error("refresh of the carried factors moves up to " + std::to_string((long long)count) + " matrices per rank, above the capacity " +
              std::to_string((long long)s->rf_cap) + " fixed by exchange_capacity > 0 (a hard limit)");
    return RBPF_ERR_OUT_OF_MEMORY;
  }
  if ((size_t)count > hard) {
    set_error("refresh of the carried factors: " + std::to_string((long long)count) + " matrices per rank is more than any plan can need (" +
              std::to_string((long long)hard) + " = (world - 1) x N_local)");
    return RBPF_ERR_INVALID_ARG;
  }
  const size_t cap = std::min<size_t>(hard, std::max<size_t>((size_t)count, 2 * s->rf_cap));
  const size_t nn = s->imat_len;
  HIPCHK(hipStreamSynchronize(c->stream));
  double *ns_ = nullptr, *nr_ = nullptr; int* ni_ = nullptr;
  int rc = dmalloc(&ns_, cap * nn);
  if (rc == RBPF_OK) rc = dmalloc(&nr_, cap * nn);
  if (rc == RBPF_OK) rc = dmalloc(&ni_, cap);
  if (rc != RBPF_OK) { hipFree(ns_); hipFree(nr_); hipFree(ni_); return rc; }
  hipFree(s->d_rf_send); hipFree(s->d_rf_recv); hipFree(s->d_rf_idx);
  s->d_rf_send = ns_; s->d_rf_recv = nr_; s->d_rf_idx = ni_; s->rf_cap = cap;
  return RBPF_OK;
}

// Base matrices other ranks asked for: slots [count] of the materialised Imat bank -> refresh_send, in the order given.
int rbpf_shard_smoother_refresh_pack(rbpf_ctx* c, const int32_t* slots, int32_t count) {
  if (!c || !c->sh || !c->sm || count < 0 || (count > 0 && !slots)) { set_error("not a sharded smoother context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  SmootherState* s = c->sm;
  if (s->rf_stage != 1 || s->base_gen < 0) { set_error("refresh_pack outside a refresh with a materialised base"); return RBPF_ERR_STATE; }
  if ((size_t)count > s->rf_cap) { set_error("refresh exchange above its capacity"); return RBPF_ERR_OUT_OF_MEMORY; }
  if (count == 0) return RBPF_OK;
  for (int q = 0; q < count; ++q) if (slots[q] < 0 || slots[q] >= c->sh->Nloc) { set_error("refresh_pack: slot out of range"); return RBPF_ERR_INVALID_ARG; }
  const size_t nn = s->imat_len;
  HIPCHK(hipMemcpyAsync(s->d_rf_idx, slots, (size_t)count * sizeof(int), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(gather_matrices_kernel, dim3(count, 8), dim3(256), 0, c->stream, nn, s->d_rf_idx, s->d_Imat[s->imat_cur], s->d_rf_send);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));                 // `slots` is caller memory; the collective may run on another stream
  return RBPF_OK;
}

// base_index [N_local]: where each of my particles finds its base matrix -- a slot of my bank (< N_local) or N_local + the
// position of the fetched matrix in refresh_recv; ignored (may be NULL) while the base is the common initial matrix.
// n_recv: matrices in refresh_recv.
int rbpf_shard_smoother_refresh_end(rbpf_ctx* c, const int32_t* base_index, int32_t n_recv) {
  if (!c || !c->sh || !c->sm) { set_error("not a sharded smoother context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  SmootherState* s = c->sm;
  ShardState* sh = c->sh;
  if (s->rf_stage != 1) { set_error("refresh_end without refresh_begin"); return RBPF_ERR_STATE; }
  const int t = c->t, N = sh->Nloc, n = c->mdl.n, d = c->mdl.d, Kp = s->rf_Kp, t0 = s->base_gen;
  hipStream_t st = c->stream;
  if (t0 >= 0) {
    if (!base_index || n_recv < 0 || (size_t)n_recv > s->rf_cap) { set_error("refresh_end: base_index / n_recv"); return RBPF_ERR_INVALID_ARG; }
    for (int p = 0; p < N; ++p) if (base_index[p] < 0 || base_index[p] >= N + n_recv) { set_error("refresh_end: base_index out of range"); return RBPF_ERR_INVALID_ARG; }
    HIPCHK(hipMemcpyAsync(s->d_base_slot, base_index, (size_t)N * sizeof(int), hipMemcpyHostToDevice, st));
  }
  if (s->refresh_free) {
    // the first (and only) factorisation, chunk by chunk: Imat = Imat0 + H' R^-1 H of generation 0 into the chunk buffer,
    // chol(Imat + ImatAddt) out of it, factors into the sweep layout; nothing was fetched from other ranks
    if (t != 1 || t0 >= 0 || Kp != 1) { set_error("internal: refresh-free sharded smoother refreshed after t = 1"); return RBPF_ERR_STATE; }
    for (int p0 = 0; p0 < N; p0 += s->l_chunk) {
      const int cnt = std::min(s->l_chunk, N - p0);
      HIPCHK(launch_meas_model(c->mdl, cnt, s->d_Xp + (size_t)p0 * c->mdl.nN, s->d_G, st, 1));
      hipLaunchKernelGGL(sweep_whiten_kernel, dim3((unsigned)(((size_t)cnt * n + 255) / 256)), dim3(256), 0, st, (size_t)cnt, d, n, s->d_W, s->d_G);
      HIPCHK(hipGetLastError());
      GemmArgs g1{n, n, d, s->d_G, 1, n, (long)d * n, s->d_G, n, 1, (long)d * n, s->d_Imat[0], 1, n, (long)s->imat_len};
      g1.lower = 1; g1.packed = s->imat_packed ? 1 : 0;
      g1.add = s->d_Imat0; g1.add_stride = 0L; g1.add_idx = nullptr;
      HIPCHK(launch_gemm(g1, cnt, st));
      RB_TRY(refresh_factorise(c, s->d_Rinv, sh->anc_local, cnt, st, p0, s->d_Imat[0]));
    }
    s->imat_valid = false; s->base_gen = t - 1;
    HIPCHK(hipMemcpyAsync(s->d_base_gid, s->d_owner_now, (size_t)sh->Nglob * sizeof(int), hipMemcpyDeviceToDevice, st));
    s->sw_cur ^= 1; s->sw_valid = true; s->rf_stage = 0;
    HIPCHK(hipStreamSynchronize(st));
    return RBPF_OK;
  }
  const int ni = s->imat_valid ? (s->imat_cur ^ 1) : 0;
  HIPCHK(launch_meas_model(c->mdl, N * Kp, s->d_Xp, s->d_G, st, 1));
  const size_t rows = (size_t)N * Kp;
  hipLaunchKernelGGL(sweep_whiten_kernel, dim3((unsigned)((rows * n + 255) / 256)), dim3(256), 0, st, rows, d, n, s->d_W, s->d_G);
  HIPCHK(hipGetLastError());
  const long Kd = (long)Kp * d;
  GemmArgs gg{n, n, (int)Kd, s->d_G, 1, n, Kd * n, s->d_G, n, 1, Kd * n, s->d_Imat[ni], 1, n, (long)s->imat_len};
  gg.lower = 1;
  gg.packed = s->imat_packed ? 1 : 0;
  gg.add = t0 < 0 ? s->d_Imat0 : s->d_Imat[s->imat_cur];               // base matrix: own bank entry or a fetched one
  gg.add_stride = t0 < 0 ? 0L : (long)s->imat_len;
  gg.add_idx = t0 < 0 ? (const int*)nullptr : s->d_base_slot;
  gg.add_rec = t0 < 0 ? (const double*)nullptr : s->d_rf_recv; gg.add_nbank = N;
  HIPCHK(launch_gemm(gg, N, st));                                      // Imat = base + G' G
  s->imat_cur = ni; s->imat_valid = true; s->base_gen = t - 1;
  HIPCHK(hipMemcpyAsync(s->d_base_gid, s->d_owner_now, (size_t)sh->Nglob * sizeof(int), hipMemcpyDeviceToDevice, st));
  RB_TRY(refresh_factorise(c, s->d_Rinv, sh->anc_local, N, st));
  s->sw_cur ^= 1; s->sw_valid = true; s->rf_stage = 0;
  HIPCHK(hipStreamSynchronize(st));              // base_index is caller memory
  return RBPF_OK;
}

// anc[j] = (log w_j + logwDyn_j) + meas[j] in place over all N logical slots
__global__ void anc_combine_kernel(int N, const double* __restrict__ dyn, double* __restrict__ anc) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < N) anc[j] = dyn[j] + anc[j];
}

// After the gather and rbpf_shard_smoother_normalise: the complete ancestor log-weights of all N particles against the
// reference state of the step about to run (:175-182,232: log w + logwDyn, evaluated replicated from the gathered bank; +
// the gathered measurement parts), normalised (:243-245), and ai(N_P) drawn (:248).  separate_gather = 0: the measurement parts
// came with the forward bank (its extra row, already in logical order); 1: they sit in anc_gather (refresh steps: they are
// computed after the gather of the forward bank and all-gathered on their own).
int rbpf_shard_smoother_anc_sample(rbpf_ctx* c, int32_t separate_gather) {
  if (!c || !c->sh || !c->sm) { set_error("not a sharded smoother context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  SmootherState* s = c->sm;
  ShardState* sh = c->sh;
  const int t = c->t, k = sh->k_iter, N = sh->Nglob, nN = c->mdl.nN, nw = c->mdl.nw;
  if (k < 1 || t < 1 || t >= c->T || sh->t_norm < t) { set_error("ancestor sampling needs k > 0, t > 0 and the normalised weights of step t-1"); return RBPF_ERR_STATE; }
  hipStream_t st = c->stream;
  if (separate_gather)
    HIPCHK(launch_permute_fwd(N, 0, sh->world, sh->Nloc, sh->placed ? sh->cur_gid : nullptr, sh->anc_gather, sh->anc_glob, nullptr, st));
  const double* xref = s->d_xnk + (size_t)t * nN;
  const double* Lq = c->d_cholQfull + (size_t)((c->chol_pages > 1) ? t - 1 : 0) * nw * nw;
  hipLaunchKernelGGL(anc_dyn_kernel, dim3((N + 63) / 64), dim3(64), 0, st, c->mdl, N, sh->xn_glob, xref,
                     c->d_odo + (size_t)(t - 1) * c->mdl.nodo, Lq, sh->w_glob, sh->anc_w);      // anc_w: scratch until the normalisation
  hipLaunchKernelGGL(anc_combine_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, sh->anc_w, sh->anc_glob);
  HIPCHK(hipGetLastError());
  RB_TRY(normalise_draw_one(c, N, t, k, sh->anc_glob, sh->anc_w, sh->anc_wc, N - 1,
                            c->d_U ? c->d_U + ((size_t)k * (c->T - 1) + (t - 1)) * N : nullptr, sh->ai_glob, st));
  HIPCHK(hipMemcpyAsync(sh->Ahist + (size_t)t * N + (N - 1), sh->ai_glob + (N - 1), sizeof(int), hipMemcpyDeviceToDevice, st));
  if (!sh->async) HIPCHK(hipStreamSynchronize(st));
  return RBPF_OK;
}

// One information-form time step of my N_local particles (device plan of rbpf_shard_plan; nothing to plan at t = 0).
int rbpf_shard_smoother_step(rbpf_ctx* c) {
  if (!c || !c->sh || !c->sm) { set_error("not a sharded smoother context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  SmootherState* s = c->sm;
  ShardState* sh = c->sh;
  const int t = c->t, k = sh->k_iter, N = sh->Nloc, n = c->mdl.n, d = c->mdl.d, nN = c->mdl.nN;
  const Layout& L = c->lay;
  const double* xref = (k > 0) ? s->d_xnk + (size_t)t * nN : nullptr;
  const int oc = s->icur, nc = (t == 0) ? 0 : (oc ^ 1);
  InfoStep is;
  if (t == 0) { is.ivec_old = s->d_ivec0; is.ivec_old_stride = 0; is.hld_old = s->d_hld0; is.hld_old_stride = 0; }
  else { is.ivec_old = s->d_ivec[oc]; is.ivec_old_stride = (size_t)L.ldx; is.hld_old = s->d_hld[oc]; is.hld_old_stride = 1; }
  is.ivec_new = s->d_ivec[nc]; is.hld_new = s->d_hld[nc]; is.qf_new = s->d_qf[nc]; is.Hb_new = s->d_Hb[nc];
  RB_TRY(shard_step_impl(c, nullptr, nullptr, k, xref, &is));
  (void)N; (void)n; (void)d;
  s->icur = nc;
  return RBPF_OK;
}

// End of iteration k (:346-354), after the last step was gathered and normalised: ak = sample(w), the new reference
// trajectory (identical on every rank) into XNK_k [nN x T]; XLK_k [n] / PK_k [n x n] are written by the rank that
// holds particle ak (owner_rank) and zero-filled elsewhere.
int rbpf_shard_smoother_end(rbpf_ctx* c, double* XNK_k, double* XLK_k, double* PK_k, int32_t* ak_out, int32_t* owner_rank) {
  if (!c || !c->sh || !c->sm) { set_error("not a sharded smoother context"); return RBPF_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  SmootherState* s = c->sm;
  ShardState* sh = c->sh;
  const int T = c->T, k = sh->k_iter, N = sh->Nglob, nN = c->mdl.nN, n = c->mdl.n, d = c->mdl.d;
  const Layout& L = c->lay;
  hipStream_t st = c->stream;
  if (c->t != T || sh->t_norm != T) { set_error("iteration not finished (run all steps, then gather + normalise)"); return RBPF_ERR_STATE; }
  double* d_uf = c->d_scal;
  if (c->rng_mode == RBPF_RNG_REPLAY) HIPCHK(hipMemcpyAsync(d_uf, &c->h_Ufin[k], 8, hipMemcpyHostToDevice, st));
  SearchArgs sa;
  sa.N = N; sa.n_draw = 1; sa.t = T; sa.wc = sh->wc_glob; sa.rng_mode = c->rng_mode; sa.k_iter = k; sa.slot0 = 0;
  sa.U = d_uf; sa.seed = c->seed; sa.ai = s->d_ak; sa.overflow = c->d_flags + 1; sa.u_is_scalar = 1;
  sa.approx = 1; sa.ambiguous = c->d_flags + 4; sa.w = sh->w_glob; sa.wc_exact = sh->wc_glob;
  if (N > kSingleWgResampleMaxN) sa.scan_depth = (N + 1023) / 1024 + 32;
  HIPCHK(launch_search(sa, st));
  HIPCHK(launch_resample_fixup(sa, st));
  HIPCHK(launch_backtrace(N, nN, T, sh->Xhist, sh->Ahist, s->d_ak, 1, s->d_xnk, st));
  int ak = 0;
  HIPCHK(hipMemcpyAsync(&ak, s->d_ak, 4, hipMemcpyDeviceToHost, st));
  RB_TRY(ctx_check_flags(c));
  int gid = ak;
  if (sh->placed) HIPCHK(hipMemcpy(&gid, sh->cur_gid + ak, 4, hipMemcpyDeviceToHost));
  const int owner = gid / sh->Nloc, idx = gid % sh->Nloc;
  if (ak_out) *ak_out = ak;
  if (owner_rank) *owner_rank = owner;
  if (XNK_k) HIPCHK(hipMemcpy(XNK_k, s->d_xnk, (size_t)nN * T * 8, hipMemcpyDeviceToHost));
  if (XLK_k) {
    std::memset(XLK_k, 0, (size_t)n * 8);
    if (owner == sh->rank) HIPCHK(hipMemcpy(XLK_k, c->xl[c->xcur] + (size_t)idx * L.ldx, (size_t)n * 8, hipMemcpyDeviceToHost));
  }
  if (PK_k) {
    std::memset(PK_k, 0, (size_t)n * n * 8);
    if (owner == sh->rank) {
      double* dP = nullptr;
      RB_TRY(dmalloc(&dP, (size_t)n * n));
      int rc = shard_unpack_particle(c, idx, dP);                 // pending downdates applied, lineage base in a record or the bank
      hipError_t e = (rc == RBPF_OK) ? hipMemcpy(PK_k, dP, (size_t)n * n * 8, hipMemcpyDeviceToHost) : hipSuccess;
      hipFree(dP);
      if (rc != RBPF_OK) return rc;
      HIPCHK(e);
    }
  }
  return RBPF_OK;
}

// The batched ancestor-weight factorisation on its own (tests and tools/chol_bench.py): for every matrix of the batch
//   cS = chol(S,'lower') (retry with S + jitter I), v = cS \ e, logw = -sum(log(diag(cS))) - .5 v'v - M/2 log(2 pi)
// (particleSmoother.m:221-229).  variant: 0 automatic, 16 / 64 force the 16- / 64-column kernel.  reps > 1 repeats
// the launch and reports the mean kernel time in *ms (HIP events on the launch stream).
int rbpf_chol_weights(int32_t M, int32_t batch, const double* S, const double* e, double jitter, int32_t variant,
                      int32_t reps, double* logw, int32_t* status, double* ms) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device"); return RBPF_ERR_NO_DEVICE; }
  const bool info = variant >= 1000;                 // information-form expression and loaders (see rbpf.h)
  if (info) variant -= 1000;
  if (!S || !e || !logw || M < 1 || M > 1023 || batch < 1 || reps < 1 ||
      (variant != 0 && variant != 1 && variant != 10 && variant != 11 && variant != 12 && variant != 14 && variant != 16 && variant != 64 && variant != 648 && variant != 644) ||
      ((variant == 1 || variant == 10 || variant == 11 || variant == 12 || variant == 14) && (!info || ((M + 16) >> 4) > kCsMaxRT || ((M + 16) >> 4) < 5))) {
    set_error("bad argument"); return RBPF_ERR_INVALID_ARG;
  }
  double *dS = nullptr, *de = nullptr, *dL = nullptr, *dlw = nullptr; int* dst = nullptr;
  double* dzero_p = nullptr;
  auto cleanup = [&]() { hipFree(dS); hipFree(de); hipFree(dL); hipFree(dlw); hipFree(dst); hipFree(dzero_p); };
  int rc = dmalloc(&dS, (size_t)batch * M * M);
  if (rc == RBPF_OK) rc = dmalloc(&de, (size_t)batch * M);
  if (rc == RBPF_OK) rc = dmalloc(&dL, (size_t)batch * chol_factor_doubles(M));
  if (rc == RBPF_OK) rc = dmalloc(&dlw, (size_t)batch);
  if (rc == RBPF_OK) rc = dmalloc(&dst, 4);
  if (rc != RBPF_OK) { cleanup(); return rc; }
  hipError_t err = hipMemcpy(dS, S, (size_t)batch * M * M * 8, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(de, e, (size_t)batch * M * 8, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemset(dst, 0, 16);
  CholArgs ca;
  std::memset(&ca, 0, sizeof(ca));
  ca.Msz = M; ca.d = 1; ca.n = M; ca.ldx = M; ca.Lbuf = dL; ca.ldL = (long)chol_factor_doubles(M);
  ca.jitter = jitter; ca.pant_log = dlw; ca.status = dst;
  double* dzero = nullptr;                           // information form: ImatAddt = 0, ivecAddt = 0, qf = hld = 0
  if (!info) {
    ca.mode = 0; ca.S = dS; ca.R = nullptr; ca.rhs = de;
  } else {
    if (err == hipSuccess) err = hipMalloc(&dzero, ((size_t)M * M + M + batch) * 8);
    if (err == hipSuccess) err = hipMemset(dzero, 0, ((size_t)M * M + M + batch) * 8);
    ca.mode = 1; ca.Imat = dS; ca.imat_stride = (long)M * M; ca.ivec = de; ca.ImatAdd = dzero; ca.ivecAdd = dzero + (size_t)M * M;
    ca.qf = dzero + (size_t)M * M + M; ca.hld = ca.qf;
    dzero_p = dzero;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  float total = 0.f;
  for (int r = 0; r < reps && err == hipSuccess; ++r) {
    err = hipMemsetAsync(dlw, 0, (size_t)batch * 8, nullptr);
    if (err == hipSuccess) err = hipEventRecord(e0, nullptr);
    if (err == hipSuccess) {
      ca.variant = variant;
      err = launch_chol(ca, batch, variant == 1 ? 1 : 0, nullptr);
    }
    if (err == hipSuccess) err = hipEventRecord(e1, nullptr);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float t = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&t, e0, e1);
    total += t;
  }
  if (err == hipSuccess) err = hipMemcpy(logw, dlw, (size_t)batch * 8, hipMemcpyDeviceToHost);
#ifdef RBPF_TUNING
  if (const char* dump = tuning_env("RBPF_CHOL_DUMP")) {                      // diagnostic builds: the factor workspace of matrix 0
    std::vector<double> hl(chol_factor_doubles(M));
    if (err == hipSuccess) err = hipMemcpy(hl.data(), dL, hl.size() * 8, hipMemcpyDeviceToHost);
    if (FILE* f = fopen(dump, "wb")) { fwrite(hl.data(), 8, hl.size(), f); fclose(f); }
  }
#endif
  int flags[4] = {0, 0, 0, 0};
  if (err == hipSuccess) err = hipMemcpy(flags, dst, 16, hipMemcpyDeviceToHost);
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
  cleanup();
  if (err != hipSuccess) return hip_fail(err, "rbpf_chol_weights", __FILE__, __LINE__);
  if (status) *status = flags[0];
  if (ms) *ms = (double)total / reps;
  return RBPF_OK;
}

// The carried-factor sweep on its own (tests/test_gpu_chol.py, bench.py): ONE augmented factor, replicated `batch` times.
//   L [(n+1) x (n+1)] column-major, lower: [L 0; z' *] with L = chol(A), z = L \ b;  U [d x n] the update vectors (rows), V [d x n]
//   the downdate vectors, eta [d] the entry both carry in the augmented row  ->  L_out (same layout: the factor of
//   A + U'U - V'V and the row (L_out \ (b + U' eta - V' eta))'), logw = -sum log diag(L_out) + z_out' z_out / 2.
// reps > 1 repeats the launch; *ms is the mean kernel time (HIP events).  Every copy gives the same result; copy 0 is returned.
__global__ void sweep_probe_fill_kernel(size_t len, int batch, const double* __restrict__ src, double* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  const double v = src[i];
  for (int p = blockIdx.y; p < batch; p += gridDim.y) dst[(size_t)p * len + i] = v;
}

int rbpf_chol_sweep_probe(int32_t n, int32_t d, int32_t batch, const double* L, const double* U, const double* V, const double* eta,
                          int32_t reps, double* L_out, double* logw, int32_t* status, double* ms) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_error("no HIP device"); return RBPF_ERR_NO_DEVICE; }
  if (!L || !U || !V || !eta || n < 1 || sweep_slots(n) > kSweepMaxSlots || (d != 1 && d != 3) || batch < 1 || reps < 1) {
    set_error("bad argument (n <= 575, d = 1 or 3)"); return RBPF_ERR_INVALID_ARG;
  }
  const int NS = sweep_slots(n), tailc = sweep_tail_compact(n), ldx = (n + 1) & ~1, n1 = n + 1;
  const size_t fd = sweep_factor_doubles(n);
  // dense lower factor -> sweep layout (rows 0..n of the columns 0..n-1; the rest of every stored slot is zero)
  std::vector<double> hsw(fd, 0.0);
  for (int k = 0; k < n; ++k) {
    const int b = k >> 6;
    double* col = hsw.data() + sweep_col_offset(k, NS, tailc);
    for (int q = b; q < NS; ++q)
      for (int ln = 0; ln < ((tailc && q == NS - 1) ? 8 : 64); ++ln) {
        const int row = 64 * q + ln;
        col[(size_t)(q - b) * 64 + ln] = (row >= k && row <= n) ? L[(size_t)row + (size_t)n1 * k] : 0.0;
      }
  }
  std::vector<double> hH((size_t)d * ldx, 0.0), hHref((size_t)d * n), hW((size_t)d * d, 0.0);
  for (int a = 0; a < d; ++a) {
    for (int c = 0; c < n; ++c) { hH[(size_t)a * ldx + c] = U[a + (size_t)d * c]; hHref[(size_t)a * n + c] = V[a + (size_t)d * c]; }
    hW[a + (size_t)d * a] = 1.0;
  }
  double *d1 = nullptr, *dold = nullptr, *dnew = nullptr, *dH1 = nullptr, *dH = nullptr, *dHref = nullptr, *dW = nullptr, *dy = nullptr, *dz = nullptr, *dlw = nullptr;
  int* dst = nullptr;
  auto cleanup = [&]() { hipFree(d1); hipFree(dold); hipFree(dnew); hipFree(dH1); hipFree(dH); hipFree(dHref); hipFree(dW); hipFree(dy); hipFree(dz); hipFree(dlw); hipFree(dst); };
  int rc = dmalloc(&d1, fd);
  if (rc == RBPF_OK) rc = dmalloc(&dold, (size_t)batch * fd);
  if (rc == RBPF_OK) rc = dmalloc(&dnew, (size_t)batch * fd);
  if (rc == RBPF_OK) rc = dmalloc(&dH1, (size_t)d * ldx);
  if (rc == RBPF_OK) rc = dmalloc(&dH, (size_t)batch * d * ldx);
  if (rc == RBPF_OK) rc = dmalloc(&dHref, (size_t)d * n);
  if (rc == RBPF_OK) rc = dmalloc(&dW, (size_t)d * d);
  if (rc == RBPF_OK) rc = dmalloc(&dy, (size_t)d);
  if (rc == RBPF_OK) rc = dmalloc(&dz, (size_t)batch);
  if (rc == RBPF_OK) rc = dmalloc(&dlw, (size_t)batch);
  if (rc == RBPF_OK) rc = dmalloc(&dst, 4);
  if (rc != RBPF_OK) { cleanup(); return rc; }
  hipError_t err = hipMemcpy(d1, hsw.data(), fd * 8, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(dH1, hH.data(), hH.size() * 8, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(dHref, hHref.data(), hHref.size() * 8, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(dW, hW.data(), hW.size() * 8, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(dy, eta, (size_t)d * 8, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemset(dz, 0, (size_t)batch * 8);
  if (err == hipSuccess) err = hipMemset(dst, 0, 16);
  if (err == hipSuccess) {
    hipLaunchKernelGGL(sweep_probe_fill_kernel, dim3((unsigned)((fd + 255) / 256), 64), dim3(256), 0, nullptr, fd, batch, d1, dold);
    hipLaunchKernelGGL(sweep_probe_fill_kernel, dim3((unsigned)(((size_t)d * ldx + 255) / 256), 64), dim3(256), 0, nullptr, (size_t)d * ldx, batch, dH1, dH);
    err = hipGetLastError();
  }
  SweepArgs sw;
  sw.n = n; sw.d = d; sw.ldx = ldx; sw.NS = NS; sw.tailc = tailc; sw.N = batch; sw.ref_slot = -1;
  sw.Lold = dold; sw.Lnew = dnew; sw.stride = fd; sw.anc = nullptr; sw.order = nullptr;
  sw.Hb = dH; sw.Href = dHref; sw.W = dW; sw.yt = dy; sw.qf = dz; sw.hld = dz; sw.pant_log = dlw; sw.status = dst;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  float total = 0.f;
  for (int r = 0; r < reps && err == hipSuccess; ++r) {
    err = hipMemsetAsync(dlw, 0, (size_t)batch * 8, nullptr);
    if (err == hipSuccess) err = hipEventRecord(e0, nullptr);
    if (err == hipSuccess) err = launch_chol_sweep(sw, nullptr);
    if (err == hipSuccess) err = hipEventRecord(e1, nullptr);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float t = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&t, e0, e1);
    total += t;
  }
  if (err == hipSuccess) err = hipMemcpy(hsw.data(), dnew, fd * 8, hipMemcpyDeviceToHost);
  double lw = 0.0;
  if (err == hipSuccess) err = hipMemcpy(&lw, dlw, 8, hipMemcpyDeviceToHost);
  int flags[4] = {0, 0, 0, 0};
  if (err == hipSuccess) err = hipMemcpy(flags, dst, 16, hipMemcpyDeviceToHost);
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
  cleanup();
  if (err != hipSuccess) return hip_fail(err, "rbpf_chol_sweep_probe", __FILE__, __LINE__);
  if (L_out) {
    std::memset(L_out, 0, (size_t)n1 * n1 * 8);
    for (int k = 0; k < n; ++k) {
      const int b = k >> 6;
      const double* col = hsw.data() + sweep_col_offset(k, NS, tailc);
      for (int row = k; row <= n; ++row) {
        const int q = row >> 6, ln = row & 63;
        L_out[(size_t)row + (size_t)n1 * k] = col[(size_t)(q - b) * 64 + ln];
      }
    }
  }
  if (logw) *logw = lw;
  if (status) *status = flags[0];
  if (ms) *ms = (double)total / reps;
  return RBPF_OK;
}


}  // extern "C"
