// Ancestor-weight factorisation of SMALL matrices (5..9 row tiles of 16: n <= 143, e.g. dense-radio m = 128): the whole
// lower triangle lives in registers.  Included by rbpf_smoother.hip inside namespace rbpf after rbpf_chol64.hpp, whose
// 16 x 16 tile routine, MFMA helpers and argument block it shares.
//
//   particleSmoother.m:221-229                  cS = chol(S,'lower') (+ jitter retry), v = cS \ e, sum(log(diag(cS)))
//   particleSmootherInformationForm.m:224-236   cIend = chol(Imat_i + ImatAddt), v = cIend \ (ivec_i + ivecAddt)
//
// Why a third kernel.  At n = 128 the 16-column kernel is bound by HBM traffic, not by latency: per particle it reads the
// stored matrix (131 KB) and ImatAddt, writes Imat(:,:,ai) (131 KB) and ALSO writes and re-reads its factor through
// global memory (90 KB of fragments, re-read once per block column; four workgroups per CU overflow the L2 share) —
// 36 GB per launch of 65 536 particles = 8.1 ms at 4.7 TB/s.  Here nothing but the matrix itself crosses the memory
// system: all loads of a particle are issued at once, the factorisation is right-looking on register tiles, and the only
// data exchanged between the four waves of a workgroup is the current block column's solved panel (through LDS).
//
// One workgroup = 4 wave64 per particle; wave w owns the row tiles w, w + 4, w + 8 with all their lower tiles (at most
// 15 tiles = 120 registers), held negated, Z = -(A - W), in the MFMA operand layout (lane l: row l & 15, columns
// (l >> 4) + 4 q of the tile).  Block column j:
//   1. the owner of row tile j factorises the diagonal tile (chol_diag_tile_frag) and publishes -inv(Ld) in LDS;
//   2. barrier; every wave solves its tiles of column j, X = V inv(Ld)' (4 MFMAs per tile), keeps X and publishes it;
//   3. barrier; every wave updates its tiles right of j:  Z(rt, ct) += X(ct) X(rt)'  (4 MFMAs per tile, A from LDS).
// The right-hand side is the extra row M of the augmented matrix, as in the other two kernels; sum(log(diag)) and v'v
// are collected from the registers on the way.
#pragma once

constexpr int kCsWaves = 4, kCsThreads = 256, kCsMaxRT = 9;

// (negated) element quad (i, jb + 4 q) of the augmented matrix [A rhs; rhs' *], 32-bit index arithmetic, one scalar base
// per array.  Same operations in the same order as chol_aug_elems.
template <int MODE>
__device__ inline v4d cs_elems(const CholArgs& a, int p, int i, int jb, int M, const double* rhs_s, const double* Hs,
                               const double* RH, double jit) {
  const int ld = (MODE == 0) ? M : a.n;
  const int ic = min(i, M - 1);
  int j[4], jc[4];
  unsigned off[4];
  double v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { j[q] = jb + 4 * q; jc[q] = min(j[q], M - 1); off[q] = (unsigned)(ic + ld * jc[q]); }
  if (MODE == 0) {
    const double* src = a.S + (size_t)p * M * M;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = src[off[q]];
    if (a.R) {
      const int* dv = reinterpret_cast<const int*>(Hs);                      // (index / d) << 3 | index % d
      const int di = dv[ic];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int dj = dv[jc[q]];
        const double rr = a.R[(di & 7) + a.d * (dj & 7)];                    // kron(eye, R)
        v[q] += ((di >> 3) == (dj >> 3)) ? rr : 0.0;
      }
    }
  } else {
    double ad[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = a.Imat[off[q]]; ad[q] = a.ImatAdd[off[q]]; }   // a.Imat: resolved by the caller
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
      double* dst = a.ImatOut + (size_t)p * a.n * a.n;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j[q] < M) __builtin_nontemporal_store(v[q], &dst[off[q]]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += ad[q];                               // :225
  }
  v4d z;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (i == j[q]) v[q] += jit;
    if (i == M) v[q] = rhs_s[jc[q]];
    z[q] = (j[q] < M && i <= M && i >= j[q]) ? -v[q] : 0.0;
  }
  return z;
}

// The whole factorisation as seen by wave WV (compile-time: its tile set is static, so every tile is a named register
// set).  Returns through sl / vv this wave's share of sum(log(diag)) and v'v (per lane, to be reduced by the caller).
template <int MODE, int WV>
__device__ inline void cs_wave(const CholArgs& a, int p, int M, int RT, const double* rhs_s, const double* Hs, const double* RH,
                               double jit, int lane, double* NIs, double* Xs, int* sfail, double& sl, double& vv) {
  constexpr int NS = (WV == 0) ? 3 : 2;                                      // row tiles WV, WV + 4 (, 8)
  const int r = lane & 15, g = lane >> 4;
  v4d T[NS][kCsMaxRT];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int rt = WV + 4 * s;
#pragma unroll
    for (int ct = 0; ct < kCsMaxRT; ++ct)
      if (ct <= rt && rt < RT) T[s][ct] = cs_elems<MODE>(a, p, 16 * rt + r, 16 * ct + g, M, rhs_s, Hs, RH, jit);
  }
  const int rM = M & 15, tM = M >> 4;                                        // the right-hand-side row: row rM of row tile tM
#pragma unroll
  for (int j = 0; j < kCsMaxRT; ++j) {
    if (j < RT) {                                                            // wave-uniform; barriers are reached by all waves
      if ((j & 3) == WV) {                                                   // 1. diagonal tile (owner)
        v4d V = -T[(j >> 2) < NS ? (j >> 2) : NS - 1][j], NI;
        const bool bad = chol_diag_tile_frag(V, NI, M - 16 * j, lane);
        if (bad && lane == 0) *sfail = 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          NIs[q * 64 + lane] = NI[q];
          const int c = 16 * j + 4 * q + g;                                  // column of this register
          if (r == 4 * q + g && c < M) sl += log(V[q]);                      // diagonal entry (column < M: not padding)
          if (j == tM && r == rM && c < M) vv = fma(V[q], V[q], vv);         // right-hand-side row inside the diagonal tile
        }
      }
      __syncthreads();
      v4d xr[NS];                                                            // 2. solves of column j
      double ni[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) ni[q] = NIs[q * 64 + lane];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int rt = WV + 4 * s;
        xr[s] = (v4d){0.0, 0.0, 0.0, 0.0};
        if (rt > j && rt < RT) {
          xr[s] = mfma4(ni, T[s][j], xr[s]);                                 // X' = inv(Ld) V'  (ni = -inv, T = -V')
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            Xs[(rt * 4 + q) * 64 + lane] = xr[s][q];
            if (rt == tM && r == rM && 16 * j + 4 * q + g < M) vv = fma(xr[s][q], xr[s][q], vv);
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int ct = j + 1; ct < kCsMaxRT; ++ct) {                            // 3. trailing update of the columns right of j
        if (ct < RT) {
          bool need = false;
#pragma unroll
          for (int s = 0; s < NS; ++s) need |= (WV + 4 * s >= ct && WV + 4 * s < RT);
          if (need) {
            double xa[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xa[q] = Xs[(ct * 4 + q) * 64 + lane];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              const int rt = WV + 4 * s;
              if (rt >= ct && rt < RT) T[s][ct] = mfma4(xa, xr[s], T[s][ct]);   // Z(rt,ct) += X(ct) X(rt)'
            }
          }
        }
      }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(kCsThreads, 2) void chol_small_kernel(CholArgs a_in) {
  extern __shared__ double csm[];
  CholArgs a = a_in;
  const int p = blockIdx.x, tid = threadIdx.x, M = a.Msz;
  if (MODE == 1) {
    const int src = a.imat_anc ? a.imat_anc[p] : p;
    const bool remote = a.rec != nullptr && src >= a.n_bank_local;
    a.Imat = remote ? a.rec + (size_t)(src - a.n_bank_local) * a.rec_stride + a.rec_off_Imat
                    : a.Imat + (size_t)src * a.imat_stride;
    a.imat_stride = 0;
  }
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int RT = (M + 1 + 15) >> 4;
  double* NIs = csm;                              // [4][64]      -inv(Ld) of the current diagonal tile
  double* Xs = NIs + 256;                         // [9][4][64]   solved tiles of the current block column
  double* red = Xs + kCsMaxRT * 256;              // [16]
  double* rhs_s = red + 16;                       // [M]
  int* sfail = reinterpret_cast<int*>(rhs_s + M);
  const bool pend = (MODE == 1 && a.Hb != nullptr);
  double* Hs = (pend || MODE == 0) ? rhs_s + M + 2 : nullptr;
  double* RH = pend ? Hs + (size_t)a.d * M : nullptr;
  chol_prologue(a, p, tid, kCsThreads, M, rhs_s, Hs, RH, pend);
  double jit = 0.0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (tid == 0) *sfail = 0;
    __syncthreads();
    double sl = 0.0, vv = 0.0;
    switch (wv) {
      case 0: cs_wave<MODE, 0>(a, p, M, RT, rhs_s, Hs, RH, jit, lane, NIs, Xs, sfail, sl, vv); break;
      case 1: cs_wave<MODE, 1>(a, p, M, RT, rhs_s, Hs, RH, jit, lane, NIs, Xs, sfail, sl, vv); break;
      case 2: cs_wave<MODE, 2>(a, p, M, RT, rhs_s, Hs, RH, jit, lane, NIs, Xs, sfail, sl, vv); break;
      default: cs_wave<MODE, 3>(a, p, M, RT, rhs_s, Hs, RH, jit, lane, NIs, Xs, sfail, sl, vv); break;
    }
    __syncthreads();
    const int failed = *sfail;
    sl = wave_sum(sl); vv = wave_sum(vv);
    if (lane == 0) { red[wv] = sl; red[8 + wv] = vv; }
    __syncthreads();
    if (!failed) {
      if (tid == 0) {
        sl = (red[0] + red[1]) + (red[2] + red[3]);
        vv = (red[8] + red[9]) + (red[10] + red[11]);
        double lw;
        if (MODE == 0) lw = -sl - 0.5 * vv - 0.5 * (double)M * 1.8378770664093453;     // log(2*pi)
        else lw = -0.5 * a.qf[p] - a.hld[p] - sl + 0.5 * vv;
        a.pant_log[p] += lw;
      }
      return;
    }
    if (MODE == 1 || attempt == 1) {
      if (tid == 0) { atomicOr(a.status, 2); a.pant_log[p] = nan(""); }
      return;
    }
    jit = a.jitter;                                                         // particleSmoother.m:223
  }
}

static size_t chol_small_lds_bytes(int M, int d) {
  return ((size_t)256 + kCsMaxRT * 256 + 16 + M + 2 + (d ? 2 * (size_t)d * M : (size_t)M)) * sizeof(double);
}

static hipError_t launch_chol_small(const CholArgs& ca, int batch, int d_lds, hipStream_t st) {
  const size_t lds = chol_small_lds_bytes(ca.Msz, d_lds);
  // information form only: inlined fifteen times per wave, the covariance form's kron(I, R) / jitter variant of the loader
  // does not fit the registers (449 spilled), and its matrices are small problems anyway (they keep the 16-column kernel)
  if (ca.mode != 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL((chol_small_kernel<1>), dim3(batch), dim3(kCsThreads), lds, st, ca);
  return hipGetLastError();
}
