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
// One workgroup = 4 wave64 per particle; the 45 tiles of the lower block triangle are dealt round-robin to the waves (at
// most 12 tiles = 96 registers each; three workgroups per CU), held negated, Z = -(A - W), in the MFMA operand layout
// (lane l: row l & 15, columns (l >> 4) + 4 q of the tile).  Block column j:
//   1. the owner of tile (j, j) factorises it (chol_diag_tile_frag) and publishes -inv(Ld) in LDS;
//   2. barrier; every wave solves its tiles of column j, X = V inv(Ld)' (4 MFMAs per tile), and publishes them;
//   3. barrier; every wave updates its tiles right of j:  Z(rt, ct) += X(ct) X(rt)'  (4 MFMAs per tile, both operands
//      from LDS: the solved tile's layout is at once the A and the B operand layout).
// The right-hand side is the extra row M of the augmented matrix, as in the other two kernels; sum(log(diag)) and v'v
// are collected from the registers on the way.  Measured (dense-radio smoother, 65 536 particles): 5.65 ms per launch
// against 8.1 ms (16-column kernel) — the remaining traffic's floor is 5.5 ms.
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
      double* dst = a.ImatOut + (size_t)p * a.imat_out_stride;
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

// Tile (rt, ct), rt >= ct, in column-major order of the lower block triangle; tiles are dealt round-robin to the four
// waves (owner = index % 4, register slot = index / 4: at most 12 tiles = 96 registers per wave), so every block
// column's solves and every trailing update are spread over all waves whatever the column.
__host__ __device__ constexpr int cs_tile_idx(int rt, int ct) { return ct * kCsMaxRT - ct * (ct - 1) / 2 + (rt - ct); }
constexpr int kCsTiles = cs_tile_idx(kCsMaxRT - 1, kCsMaxRT - 1) + 1;        // 45
constexpr int kCsSlots = (cs_tile_idx(kCsMaxRT - 1, kCsMaxRT - 1) + 4) / 4;

// The whole factorisation as seen by wave WV (compile-time: its tile set is static, so every tile is a named register
// set).  Returns through sl / vv this wave's share of sum(log(diag)) and v'v (per lane, to be reduced by the caller).
// NW: waves that share one particle (4: the workgroup kernel above; 1: one wave owns all 45 tiles -- 180 fp64 registers, no
// other wave to wait for: the barriers below then involve this wave only).
template <int MODE, int WV, int NW = 4>
__device__ inline void cs_wave(const CholArgs& a, int p, int M, int RT, const double* rhs_s, const double* Hs, const double* RH,
                               double jit, int lane, double* NIs, double* Xs, int* sfail, double& sl, double& vv) {
  const int r = lane & 15, g = lane >> 4;
  v4d T[(kCsTiles + NW - 1) / NW];
#pragma unroll
  for (int ct = 0; ct < kCsMaxRT; ++ct)
#pragma unroll
    for (int rt = ct; rt < kCsMaxRT; ++rt)
      if (cs_tile_idx(rt, ct) % NW == WV && rt < RT)
        T[cs_tile_idx(rt, ct) / NW] = cs_elems<MODE>(a, p, 16 * rt + r, 16 * ct + g, M, rhs_s, Hs, RH, jit);
  const int rM = M & 15, tM = M >> 4;                                        // the right-hand-side row: row rM of row tile tM
#pragma unroll
  for (int j = 0; j < kCsMaxRT; ++j) {
    if (j < RT) {                                                            // wave-uniform; barriers are reached by all waves
      if (cs_tile_idx(j, j) % NW == WV) {                                     // 1. diagonal tile (owner)
        v4d V = -T[cs_tile_idx(j, j) / NW], NI;
        const bool bad = chol_diag_tile_frag(V, NI, M - 16 * j, lane);
        if (bad && lane == 0) *sfail = 1;
        double dv = 1.0;                                                     // this lane's diagonal entry, if it holds one (log 1 = 0):
#pragma unroll                                                               // ONE log per tile and lane, not four -- 12 % of the kernel
        for (int q = 0; q < 4; ++q) {
          NIs[q * 64 + lane] = NI[q];
          const int c = 16 * j + 4 * q + g;                                  // column of this register
          if (r == 4 * q + g && c < M) dv = V[q];                            // diagonal entry (column < M: not padding)
          if (j == tM && r == rM && c < M) vv = fma(V[q], V[q], vv);         // right-hand-side row inside the diagonal tile
        }
        sl += log(dv);
      }
      __syncthreads();
      double ni[4];                                                          // 2. solves of column j
#pragma unroll
      for (int q = 0; q < 4; ++q) ni[q] = NIs[q * 64 + lane];
#pragma unroll
      for (int rt = j + 1; rt < kCsMaxRT; ++rt) {
        if (cs_tile_idx(rt, j) % NW == WV && rt < RT) {
          const v4d x = mfma4(ni, T[cs_tile_idx(rt, j) / NW], (v4d){0.0, 0.0, 0.0, 0.0});   // X' = inv(Ld) V' (ni = -inv, T = -V')
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            Xs[(rt * 4 + q) * 64 + lane] = x[q];
            if (rt == tM && r == rM && 16 * j + 4 * q + g < M) vv = fma(x[q], x[q], vv);
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int ct = j + 1; ct < kCsMaxRT; ++ct) {                            // 3. trailing update of the columns right of j
#pragma unroll
        for (int rt = ct; rt < kCsMaxRT; ++rt) {
          if (cs_tile_idx(rt, ct) % NW == WV && rt < RT) {
            double xa[4], xb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { xa[q] = Xs[(ct * 4 + q) * 64 + lane]; xb[q] = Xs[(rt * 4 + q) * 64 + lane]; }
            v4d& Z = T[cs_tile_idx(rt, ct) / NW];
#pragma unroll
            for (int q = 0; q < 4; ++q) Z = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[q], xb[q], Z, 0, 0, 0);   // Z(rt,ct) += X(ct) X(rt)'
          }
        }
      }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(kCsThreads, 3) void chol_small_kernel(CholArgs a_in) {
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

// Fewer waves per particle (VERDICT r02 item 5 asked for one): with NW = 1 the whole lower block triangle (45 tiles = 180 fp64
// registers per lane, VGPRs + AGPRs at one wave per SIMD) belongs to one wave -- no workgroup barrier, four particles per CU
// working independently -- but the loader keeps every element of the particle and of ImatAddt in flight on top of it and the
// kernel spills (measured slower, see RBPF_CS_WAVES); NW = 2 fits 249 registers at two waves per SIMD (four particles per CU
// instead of three, half the barrier participants) and is the default.  The same operations on every tile as chol_small_kernel (the tile
// ownership changes, not the arithmetic; only the final sums of log(diag) and v'v are associated differently).
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 1 ? 1 : 2)) void chol_smallw_kernel(CholArgs a_in) {
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
  double* NIs = csm;
  double* Xs = NIs + 256;
  double* red = Xs + kCsMaxRT * 256;
  double* rhs_s = red + 16;
  int* sfail = reinterpret_cast<int*>(rhs_s + M);
  const bool pend = (MODE == 1 && a.Hb != nullptr);
  double* Hs = (pend || MODE == 0) ? rhs_s + M + 2 : nullptr;
  double* RH = pend ? Hs + (size_t)a.d * M : nullptr;
  chol_prologue(a, p, tid, 64 * NW, M, rhs_s, Hs, RH, pend);
  if (tid == 0) *sfail = 0;
  __syncthreads();
  double sl = 0.0, vv = 0.0;
  if (NW == 1 || wv == 0) cs_wave<MODE, 0, NW>(a, p, M, RT, rhs_s, Hs, RH, 0.0, lane, NIs, Xs, sfail, sl, vv);
  else cs_wave<MODE, (NW > 1 ? 1 : 0), NW>(a, p, M, RT, rhs_s, Hs, RH, 0.0, lane, NIs, Xs, sfail, sl, vv);
  __syncthreads();
  const int failed = *sfail;
  sl = wave_sum(sl); vv = wave_sum(vv);
  if (lane == 0) { red[wv] = sl; red[8 + wv] = vv; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < NW; ++w) { sl += red[w]; vv += red[8 + w]; }
    if (!failed) a.pant_log[p] += -0.5 * a.qf[p] - a.hld[p] - sl + 0.5 * vv;
    else { atomicOr(a.status, 2); a.pant_log[p] = nan(""); }
  }
}

// ---- one wave per particle, LEFT-looking on register tiles (r03) -----------------------------------------------
// chol_smallw_kernel<., 1> above loads the whole matrix first and then works right-looking: all 45 tiles plus every load of the
// particle and of ImatAddt in flight at once do not fit 512 registers.  Left-looking, a tile is only needed when its block column
// comes up: block column ct's tiles are formed from the raw loads issued one block column earlier (so their latency hides behind
// that column's products, factorisation and solves), Z(rt, ct) = -(A) + sum_{k < ct} X(ct, k) X(rt, k)' is accumulated from the
// finished tiles in registers (a solved tile's layout is at once the A and the B operand layout), the diagonal tile is factorised
// in the wave (chol_diag_tile_frag) and the tiles below it are solved with four MFMAs each.  Live registers peak at the 46 tile
// equivalents of block column 6 (368 of the 512 a wave has at one wave per SIMD); no LDS traffic, no barrier, no other wave on the
// SIMD to stretch the serial chain of the diagonal tiles.  Four particles per CU, as with two waves per particle.
// The element arithmetic is cs_elems' (split into the issue of the loads and the rest), tile by tile the same operations as
// chol_small_kernel except that the products of a tile are summed column by column (left-looking) instead of applied one block
// column at a time in the same order -- i.e. the same sums in the same order.
template <int MODE>
__device__ inline void cs1_issue(const CholArgs& a, int i, int jb, int M, double (&v)[4], double (&ad)[4]) {
  const int ld = (MODE == 0) ? M : a.n;
  const int ic = min(i, M - 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned off = (unsigned)(ic + ld * min(jb + 4 * q, M - 1));
    v[q] = a.Imat[off];                                                      // a.Imat: resolved by the caller
    ad[q] = a.ImatAdd[off];
  }
}

template <int MODE>
__device__ inline v4d cs1_finish(const CholArgs& a, int p, int i, int jb, int M, const double* rhs_s, const double* Hs, const double* RH,
                                 const double (&vin)[4], const double (&ad)[4]) {
  const int ld = (MODE == 0) ? M : a.n;
  const int ic = min(i, M - 1);
  int j[4], jc[4];
  double v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { j[q] = jb + 4 * q; jc[q] = min(j[q], M - 1); v[q] = vin[q]; }
  if (Hs) {                                                                  // + dyi'/R*dyi of the last update (:334)
    double sacc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int aa = 0; aa < a.d; ++aa) {
      const double h = Hs[aa * M + ic];
#pragma unroll
      for (int q = 0; q < 4; ++q) sacc[q] = fma(h, RH[aa * M + jc[q]], sacc[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += sacc[q];
  }
  if (a.ImatOut && i < M) {                                                  // Imat(:,:,i) of the new generation
    double* dst = a.ImatOut + (size_t)p * a.imat_out_stride;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (j[q] < M) __builtin_nontemporal_store(v[q], &dst[(unsigned)(ic + ld * jc[q])]);
  }
  v4d z;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] += ad[q];                                                           // :225
    if (i == M) v[q] = rhs_s[jc[q]];
    z[q] = (j[q] < M && i <= M && i >= j[q]) ? -v[q] : 0.0;
  }
  return z;
}

// block column CT of the one-wave kernel (a template so that every tile index is a compile-time constant: the nine columns in one
// unrolled loop exceed the unroller's size limit and the tiles would live in scratch)
template <int MODE, int CT>
__device__ __forceinline__ void cs1_column(const CholArgs& a, int p, int M, int RT, const double* rhs_s, const double* Hs, const double* RH,
                                           int lane, v4d (&T)[kCsTiles], double (&rv)[kCsMaxRT][4], double (&ra)[kCsMaxRT][4],
                                           double& sl, double& vv, bool& bad) {
  if (CT >= RT) return;                                                      // wave-uniform
  const int r = lane & 15, g = lane >> 4;
  const int rM = M & 15, tM = M >> 4;                                        // the right-hand-side row: row rM of row tile tM
#pragma unroll
  for (int rt = CT; rt < kCsMaxRT; ++rt)
    if (rt < RT) T[cs_tile_idx(rt, CT)] = cs1_finish<MODE>(a, p, 16 * rt + r, 16 * CT + g, M, rhs_s, Hs, RH, rv[rt], ra[rt]);
#pragma unroll
  for (int rt = CT + 1; rt < kCsMaxRT; ++rt)                                 // the next block column's loads, a column's work ahead
    if (rt < RT) cs1_issue<MODE>(a, 16 * rt + r, 16 * (CT + 1) + g, M, rv[rt], ra[rt]);
#pragma unroll
  for (int k = 0; k < CT; ++k) {                                             // Z(rt, CT) += X(CT, k) X(rt, k)'
#pragma unroll
    for (int rt = CT; rt < kCsMaxRT; ++rt) {
      if (rt < RT) {
        v4d& Z = T[cs_tile_idx(rt, CT)];
        const v4d xa = T[cs_tile_idx(CT, k)];
        const v4d xb = T[cs_tile_idx(rt, k)];
#pragma unroll
        for (int q = 0; q < 4; ++q) Z = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[q], xb[q], Z, 0, 0, 0);
      }
    }
  }
  v4d V = -T[cs_tile_idx(CT, CT)], NI;                                        // diagonal tile
#ifdef RBPF_CS1_NODIAG                            // timing experiment only (wrong results): what does the serial chain cost?
  NI = V;
#else
  bad |= chol_diag_tile_frag(V, NI, M - 16 * CT, lane);
#endif
  double dv = 1.0;                                                           // this lane's diagonal entry, if it holds one (log 1 = 0)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 16 * CT + 4 * q + g;                                       // column of this register
    if (r == 4 * q + g && c < M) dv = V[q];                                  // diagonal entry (column < M: not padding)
    if (CT == tM && r == rM && c < M) vv = fma(V[q], V[q], vv);              // right-hand-side row inside the diagonal tile
  }
#ifdef RBPF_CS1_NOLOG
  sl += dv;
#else
  sl += log(dv);
#endif
#pragma unroll
  for (int rt = CT + 1; rt < kCsMaxRT; ++rt) {                               // solves of block column CT
    if (rt < RT) {
      const v4d x = mfma4(NI, T[cs_tile_idx(rt, CT)], (v4d){0.0, 0.0, 0.0, 0.0});   // X' = inv(Ld) V' (NI = -inv, T = -V')
      T[cs_tile_idx(rt, CT)] = x;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (rt == tM && r == rM && 16 * CT + 4 * q + g < M) vv = fma(x[q], x[q], vv);
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(64, 1) void chol_small1_kernel(CholArgs a_in) {
  extern __shared__ double csm[];
  CholArgs a = a_in;
  const int p = blockIdx.x, lane = threadIdx.x, M = a.Msz;
  if (MODE == 1) {
    const int src = a.imat_anc ? a.imat_anc[p] : p;
    const bool remote = a.rec != nullptr && src >= a.n_bank_local;
    a.Imat = remote ? a.rec + (size_t)(src - a.n_bank_local) * a.rec_stride + a.rec_off_Imat
                    : a.Imat + (size_t)src * a.imat_stride;
    a.imat_stride = 0;
  }
  const int RT = (M + 1 + 15) >> 4;
  double* rhs_s = csm;                            // [M]
  const bool pend = (MODE == 1 && a.Hb != nullptr);
  double* Hs = pend ? rhs_s + M + 2 : nullptr;
  double* RH = pend ? Hs + (size_t)a.d * M : nullptr;
  chol_prologue(a, p, lane, 64, M, rhs_s, Hs, RH, pend);
  __syncthreads();
  v4d T[kCsTiles];                                                           // static indices only: every tile is a named register set
  double rv[kCsMaxRT][4], ra[kCsMaxRT][4];                                   // raw loads of the next block column, by row tile
  double sl = 0.0, vv = 0.0;
  bool bad = false;
#pragma unroll
  for (int rt = 0; rt < kCsMaxRT; ++rt)
    if (rt < RT) cs1_issue<MODE>(a, 16 * rt + (lane & 15), lane >> 4, M, rv[rt], ra[rt]);
#define RBPF_CS1(CT_) cs1_column<MODE, CT_>(a, p, M, RT, rhs_s, Hs, RH, lane, T, rv, ra, sl, vv, bad)
  RBPF_CS1(0); RBPF_CS1(1); RBPF_CS1(2); RBPF_CS1(3); RBPF_CS1(4); RBPF_CS1(5); RBPF_CS1(6); RBPF_CS1(7); RBPF_CS1(8);
#undef RBPF_CS1
  static_assert(kCsMaxRT == 9, "nine block columns are written out");
  sl = wave_sum(sl); vv = wave_sum(vv);
  const bool failed = __any(bad);
  if (lane == 0) {
    if (!failed) a.pant_log[p] += -0.5 * a.qf[p] - a.hld[p] - sl + 0.5 * vv;
    else { atomicOr(a.status, 2); a.pant_log[p] = nan(""); }
  }
}

static size_t chol_small1_lds_bytes(int M, int d) { return ((size_t)M + 2 + 2 * (size_t)d * M) * sizeof(double); }

static size_t chol_small_lds_bytes(int M, int d) {
  return ((size_t)256 + kCsMaxRT * 256 + 16 + M + 2 + (d ? 2 * (size_t)d * M : (size_t)M)) * sizeof(double);
}

#ifndef RBPF_CS_WAVES
#define RBPF_CS_WAVES 2         // waves per particle.  Measured r03 (16 384 matrices, n = 128; dense-radio smoother step at N_P = 65 536):
                                // 4 waves 1.35 ms / 8.85 ms, 2 waves 1.24 ms / 8.52 ms, 1 wave 1.78 ms / 11.75 ms (180 fp64 tile
                                // registers + every load of a particle in flight do not fit 512 registers: 274 spilled)
#endif

static hipError_t launch_chol_small(const CholArgs& ca, int batch, int d_lds, hipStream_t st, int waves = 0) {
  size_t lds = chol_small_lds_bytes(ca.Msz, d_lds);
  if (const char* pad = tuning_env("RBPF_CS_LDS_PAD")) lds += (size_t)atoi(pad) * 1024;   // tuning: fewer workgroups per CU
  // information form only: inlined fifteen times per wave, the covariance form's kron(I, R) / jitter variant of the loader
  // does not fit the registers (449 spilled), and its matrices are small problems anyway (they keep the 16-column kernel)
  if (ca.mode != 1) return hipErrorInvalidValue;
  const int nw = waves ? waves : RBPF_CS_WAVES;
  if (nw == 10) hipLaunchKernelGGL((chol_small1_kernel<1>), dim3(batch), dim3(64), chol_small1_lds_bytes(ca.Msz, d_lds), st, ca);
  else if (nw == 1) hipLaunchKernelGGL((chol_smallw_kernel<1, 1>), dim3(batch), dim3(64), lds, st, ca);
  else if (nw == 2) hipLaunchKernelGGL((chol_smallw_kernel<1, 2>), dim3(batch), dim3(128), lds, st, ca);
  else hipLaunchKernelGGL((chol_small_kernel<1>), dim3(batch), dim3(kCsThreads), lds, st, ca);
  return hipGetLastError();
}
