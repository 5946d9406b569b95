// Ancestor-weight factorisation of the information form, 128-column variant (included by rbpf_smoother.hip after rbpf_chol64.hpp,
// whose element loaders, diagonal-tile product, diagonal-block routine and argument block it shares).
//
//   particleSmootherInformationForm.m:224-236   cIend = chol(Imat_i + ImatAddt), v = cIend \ (ivec_i + ivecAddt)
//
// Why a third kernel.  Counters of the 64-column kernel at n = 515 (profiles/r03_mag_chol64_pmc_summary.txt): 85 GB per launch of
// 8192 matrices against 19 GB of matrix in / out -- the left-looking panel products re-read the finished factor once per 64 columns
// as B operands (3.05 MB per matrix), at 5.25 TB/s: the kernel is bound by traffic it creates itself.  Here the factor is re-read once
// per 128 columns, in the 64-column kernel's own asynchronous schedule:
//
//   * a SUPER-BLOCK is 128 columns = two 64-column blocks J0, J1.  A strip (row tile) below J0's diagonal block accumulates the panel
//     product over the finished columns for BOTH blocks at once (8 accumulator tiles): its own rows stream through once per 128
//     columns (B operands: n^3 / 768 elements instead of n^3 / 384);
//   * J0's diagonal block: its four row tiles are formed FIRST by waves 7..4 (c64_diag_product, as in the 64-column kernel) and
//     handed to wave 0 through LDS and a counter, so that wave 0 factorises it (c64_diag_block) WHILE the seven workers load their
//     elements and run their products -- nobody waits for it before the solve;
//   * after the solve against J0 the solved tiles stay in registers: they are the B operands of the K = 64 update of the strip's J1
//     half with the rows of J1's diagonal block (published in LDS by their owners, waves 4..7, which carry them as ordinary strips);
//     those owners then hand J1's diagonal tiles to wave 0, and while it factorises them the workers load the J1 half of their
//     ELEMENTS, which the product did not need (it ran from zero): the second diagonal block hides behind a memory stream;
//   * row tiles beyond the ten of the first pass run in further passes of fourteen, with both diagonal blocks' operands still in LDS.
//
// r04 history: a lock-step version with the A operands staged through an LDS ring (18.0 ms per 8192 against the
// 64-column kernel's 15.5) and a two-team version that left its operands to the L2 (19.1 ms, 82 GB: the L2 retains nothing) came
// before this one.
//
// Factor storage: as in the 64-column kernel (row-tile major, fragment order; complete, so the carried-factor refresh can read it).
#pragma once

#ifndef RBPF_C128_GRING
#define RBPF_C128_GRING 3                        // column groups of A / B operand loads in flight
#endif
#ifdef RBPF_C128_STAMPS                          // tuning aid: per-phase clocks of waves 0 / 1 / 4 of workgroup 0
#define C128_STAMP(k) do { const long long now_ = clock64(); cst[k] += now_ - clast; clast = now_; } while (0)
#define C128_STAMP_ARGS , long long (&cst)[16], long long& clast
#define C128_STAMP_PASS , cst, clast
#else
#define C128_STAMP(k) do { } while (0)
#define C128_STAMP_ARGS
#define C128_STAMP_PASS
#endif

constexpr int kC128MaxSpins = 1 << 18;           // ~ 10 ms: a legitimate wait is a few ten thousand clocks

__device__ inline bool c128_wait_ge(int* ctr, int target) {
  int spins = 0;
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target && spins < kC128MaxSpins) {
    __builtin_amdgcn_s_sleep(1);
    ++spins;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return spins < kC128MaxSpins;
}

// Elements of a 16 x 64 strip (row tile rt, 64-column block J) through the loader that fits it: interior strips and the last row
// tile call-free, everything else through the general loader.  ACC: subtract from Zs (half a strip of loads in flight) instead of
// assigning.
template <bool ACC>
__device__ inline void c128_strip64(const CholArgs& a, int p, int rt, int J, int M, const double* rhs_s, const double* Hs,
                                    const double* RH, double jit, int lane, v4d (&Zs)[4]) {
  const int RTl = (M + 1 + 15) >> 4;
  const bool whole = 64 * J + 64 <= M;
  if (rt >= 4 * J + RBPF_C64_INT1 && rt < RTl - 1 && whole) {
    c64_strip_fast<1, ACC>(a, p, rt, J, M, Hs, RH, lane, Zs);
  } else if ((M & 15) != 0 && rt == RTl - 1 && rt >= 4 * J + 4 && whole) {
    c64_strip_fast<1, ACC, true>(a, p, rt, J, M, Hs, RH, lane, Zs, rhs_s);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const v4d e = c64_elems_general<1>(c64_kernarg(), a.Imat, p, 16 * rt + (lane & 15), 64 * J + 16 * c + (lane >> 4), M, rhs_s, Hs, RH, jit);
      Zs[c] = ACC ? Zs[c] - e : -e;
    }
  }
}

// The I + 1 lower tiles of row tile I of the diagonal block of 64-column block J.
template <int I>
__device__ inline void c128_tri64(const CholArgs& a, int p, int J, int M, const double* rhs_s, const double* Hs, const double* RH,
                                  double jit, int lane, v4d (&Z)[I + 1]) {
  if (64 * J + 64 <= M) {
    c64_diag_elems_fast<I, 1>(a, p, J, M, Hs, RH, jit, lane, Z);
  } else {
#pragma unroll
    for (int c = 0; c <= I; ++c)
      Z[c] = -c64_elems_general<1>(c64_kernarg(), a.Imat, p, 16 * (4 * J + I) + (lane & 15), 64 * J + 16 * c + (lane >> 4), M, rhs_s, Hs, RH, jit);
  }
}

// X = V inv(Ld)' for the four sub-columns of half h (NLs: -inv(Ld_cc), Lds: Ld(c', c), both as MFMA A fragments in LDS) of the strips
// S0 .. NS - 1; the solved tiles replace the accumulators (they are the B operands of what follows) and go to the factor at column
// group kg0.
template <int NS, int S0>
__device__ inline void c128_solve_half(v4d (&Z)[2][2][4], int h, const double* NLs, const double* Lds, double* __restrict__ Lt, int KGS,
                                       const int (&rt)[2], int kg0, int lane) {
  if (S0 >= NS) return;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double ni[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ni[q] = NLs[(c * 4 + q) * 64 + lane];
    v4d x[2];
#pragma unroll
    for (int s = S0; s < NS; ++s) {
      x[s] = mfma4(ni, Z[s][h][c], (v4d){0.0, 0.0, 0.0, 0.0});
      double* dx = Lt + ((size_t)rt[s] * KGS + kg0 + 4 * c) * 64 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) dx[q * 64] = x[s][q];
    }
#pragma unroll
    for (int cp = c + 1; cp < 4; ++cp) {
      double lf[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) lf[q] = Lds[(c64_pair(cp, c) * 4 + q) * 64 + lane];
#pragma unroll
      for (int s = S0; s < NS; ++s) Z[s][h][cp] = mfma4(lf, x[s], Z[s][h][cp]);
    }
#pragma unroll
    for (int s = S0; s < NS; ++s) Z[s][h][c] = x[s];
  }
}

// One pass of a worker wave over super-block J2 (64-column blocks 2 J2 and 2 J2 + 1): NS strips, row tiles rt[0..NS).
// DI >= 0 (waves 7 - DI = 4..7, first pass): strip 0 is row 4 + E, E = 3 - DI, of the super-block's diagonal rows -- an ordinary
// strip below block 2 J2 whose second half holds the tiles 0..E of block 2 J2 + 1's diagonal block; its solved first half is
// published in LDS (Lds2) and its second half handed to wave 0 after the update.
// FIRST: the pass that runs beside wave 0's factorisations (four workgroup barriers); later passes find every operand in LDS.
template <int NS, int DI, bool FIRST>
__device__ inline void c128_pass(const CholArgs& a, int p, double* __restrict__ Lt, int KGS, int J2, const int (&rt)[2], bool hasB, int nd2,
                                 int M, const double* rhs_s, const double* Hs, const double* RH, int lane, double* hb0, double* hb1,
                                 double* Lds2 C128_STAMP_ARGS) {
  constexpr int E = DI < 0 ? 0 : 3 - DI;
  constexpr bool DG = DI >= 0;
  constexpr int SD = DG ? 1 : 0;                                              // first ordinary strip
  const bool dstrip = DG && hasB;                                             // strip 0 is a diagonal row of block 2 J2 + 1 (wave-uniform)
  const int nkg = 32 * J2, nh1 = max(0, nd2 - 4);
  v4d Z[2][2][4];                                                            // [strip][half][sub-column]
  // ---- elements: first halves; second halves only where nothing hides them (later passes, the diagonal row) --------------
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    c128_strip64<false>(a, p, rt[s], 2 * J2, M, rhs_s, Hs, RH, 0.0, lane, Z[s][0]);
    if (DG && s == 0) {
      v4d T[E + 1];
      if (dstrip) c128_tri64<E>(a, p, 2 * J2 + 1, M, rhs_s, Hs, RH, 0.0, lane, T);
#pragma unroll
      for (int c = 0; c < 4; ++c) Z[0][1][c] = (dstrip && c <= E) ? T[c <= E ? c : 0] : (v4d){0.0, 0.0, 0.0, 0.0};
    } else if (FIRST) {
#pragma unroll
      for (int c = 0; c < 4; ++c) Z[s][1][c] = (v4d){0.0, 0.0, 0.0, 0.0};
    } else {
      c128_strip64<false>(a, p, rt[s], 2 * J2 + 1, M, rhs_s, Hs, RH, 0.0, lane, Z[s][1]);
    }
  }
  C128_STAMP(5);
  // ---- panel product over the finished columns, both halves: A operands = the eight diagonal row tiles, B operands = own rows --
  if (nkg > 0 && NS > 0) {
    constexpr int kR = RBPF_C128_GRING;
    const double* pa = Lt + (size_t)(8 * J2) * KGS * 64;
    const size_t ts = (size_t)KGS * 64;
    const double* pb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) pb[s] = Lt + (size_t)rt[s < NS ? s : 0] * ts;
    double A[kR][8], B[kR][2];
#pragma unroll
    for (int b = 0; b < kR; ++b) {
      C64_PIN();
#pragma unroll
      for (int r = 0; r < 8; ++r) A[b][r] = (pa + r * ts + (size_t)min(b, nkg - 1) * 64)[lane];
#pragma unroll
      for (int s = 0; s < NS; ++s) B[b][s] = (pb[s] + (size_t)min(b, nkg - 1) * 64)[lane];
      C64_PIN();
    }
    for (int kg = 0; kg < nkg; kg += kR) {
#pragma unroll
      for (int b = 0; b < kR; ++b) {
        if (kg + b < nkg) {
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int c = 0; c < 8; ++c) Z[s][c >> 2][c & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[b][c], B[b][s], Z[s][c >> 2][c & 3], 0, 0, 0);
        }
        const size_t kn = (size_t)min(kg + kR + b, nkg - 1) * 64;
        C64_PIN();
#pragma unroll
        for (int r = 0; r < 8; ++r) A[b][r] = (pa + r * ts + kn)[lane];
#pragma unroll
        for (int s = 0; s < NS; ++s) B[b][s] = (pb[s] + kn)[lane];
        C64_PIN();
      }
    }
  }
  C128_STAMP(6);
  if (FIRST) __syncthreads();                                                 // BA: wave 0 has factorised block 2 J2 (it ran beside the product)
  C128_STAMP(7);
  if (nd2 > 4) {
    c128_solve_half<NS, 0>(Z, 0, hb0, hb0 + 1024, Lt, KGS, rt, 32 * J2, lane);
    if (dstrip) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) Lds2[((E * 4 + c) * 4 + q) * 64 + lane] = Z[0][0][c][q];
    }
  }
  C128_STAMP(8);
  if (FIRST) __syncthreads();                                                 // BB: rows 4..7 of the diagonal rows, first-half columns, are in LDS
  // ---- second half += first half's solved tiles * (those rows)'  -- B operands straight from the registers ---------------
  if (nh1 > 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double lf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lf[q] = Lds2[((e * 4 + c) * 4 + q) * 64 + lane];
#pragma unroll
        for (int s = 0; s < NS; ++s) Z[s][1][e] = mfma4(lf, Z[s][0][c], Z[s][1][e]);
      }
  }
  C128_STAMP(9);
  if (FIRST) {
    if (dstrip) {
#pragma unroll
      for (int c = 0; c <= E; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) hb1[(c64_tri(E, c) * 4 + q) * 64 + lane] = Z[0][1][c][q];
    }
    __syncthreads();                                                          // BC: block 2 J2 + 1's diagonal tiles are in LDS
    // the second half's elements of the ordinary strips, while wave 0 factorises (the product ran from zero)
#pragma unroll
    for (int s = SD; s < NS; ++s) c128_strip64<true>(a, p, rt[s], 2 * J2 + 1, M, rhs_s, Hs, RH, 0.0, lane, Z[s][1]);
    C128_STAMP(12);
    __syncthreads();                                                          // BD: wave 0 has factorised block 2 J2 + 1
    C128_STAMP(10);
  }
  if (nd2 == 8) c128_solve_half<NS, SD>(Z, 1, hb1, hb1 + 1024, Lt, KGS, rt, 32 * J2 + 16, lane);
  C128_STAMP(11);
}

constexpr size_t kC128MaxLds = 160 * 1024;
static size_t chol128_lds_doubles(int M, int d) { return (size_t)5120 + 4096 + 32 + M + 4 + 2 * (size_t)d * M; }
static size_t chol128_lds_bytes(int M, int d) { return chol128_lds_doubles(M, d) * sizeof(double); }

__global__ __launch_bounds__(512, 1) void chol_solve128_kernel(CholArgs a_in) {
  extern __shared__ double csm[];
  constexpr int kThreads = 512;
  const int p = blockIdx.x;
  CholArgs a = a_in;
  const int tid = threadIdx.x, M = a.Msz;
  {
    const int src = a.imat_anc ? a.imat_anc[p] : p;
    const bool remote = a.rec != nullptr && src >= a.n_bank_local;
    a.Imat = remote ? a.rec + (size_t)(src - a.n_bank_local) * a.rec_stride + a.rec_off_Imat : a.Imat + (size_t)src * a.imat_stride;
    a.imat_stride = 0;
  }
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int RT = (M + 1 + 15) >> 4, KGS = 4 * RT;
  double* Lt = a.Lbuf + (size_t)p * a.ldL;
  // LDS: hb0 / hb1 [2560] a 64 x 64 diagonal block's tiles on their way to wave 0, then -inv(Ld_cc) [4][4][64] and Ld(c',c) [6][4][64];
  //      Lds2 [4][4][4][64] rows 4..7 of the super-block's diagonal rows, first-half columns, as MFMA A fragments; scalars and vectors
  double* hb0 = csm;
  double* hb1 = csm + 2560;
  double* Lds2 = csm + 5120;
  double* red = Lds2 + 4096;                                          // [32]
  double* rhs_s = red + 32;                                           // [M]
  int* flags = reinterpret_cast<int*>(rhs_s + M);                     // sfail, ready: 4 doubles
  const bool pend = a.Hb != nullptr;
  double* Hs = pend ? rhs_s + M + 4 : nullptr;
  double* RH = pend ? Hs + (size_t)a.d * M : nullptr;
  chol_prologue(a, p, tid, kThreads, M, rhs_s, Hs, RH, pend);
  int* sfail = flags;
  int* ready = flags + 1;
  if (tid < 8) flags[tid] = 0;
  __syncthreads();
#ifdef RBPF_C128_STAMPS
  long long cst[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, clast = clock64();
#endif
  const int NJ2 = (RT + 7) >> 3;
  int handed = 0;
  for (int J2 = 0; J2 < NJ2; ++J2) {
    const int nd2 = min(8, RT - 8 * J2), nh0 = min(4, nd2), nh1 = nd2 - nh0;
    const int first_below = 8 * J2 + 8, n_below = max(0, RT - first_below);
    const int nlate = n_below > 10 ? (n_below - 10 + 13) / 14 : 0;
    handed += nh0;
    if (wv == 0) {
      const bool okw = c128_wait_ge(ready, handed);                   // the four row tiles of block 2 J2's diagonal block (waves 7..4)
      C128_STAMP(0);
      bool bad = c64_diag_block(Lt, KGS, 2 * J2, nh0, M, lane, hb0, hb0, hb0 + 1024) || !okw;
      C128_STAMP(1);
      __syncthreads();                                                // BA
      __syncthreads();                                                // BB
      __syncthreads();                                                // BC
      if (nh1 > 0) bad = c64_diag_block(Lt, KGS, 2 * J2 + 1, nh1, M, lane, hb1, hb1, hb1 + 1024) || bad;
      C128_STAMP(2);
      __syncthreads();                                                // BD
      if (bad && lane == 0) *sfail = 1;
      C128_STAMP(3);
    } else {
      const int DIw = 7 - wv;                                         // waves 4..7: 3..0
      if (wv >= 4 && DIw < nd2) {
        // row tile DIw of block 2 J2's diagonal block: elements + product over the finished columns, handed to wave 0 at once
#define RBPF_C128D(I_) c64_diag_product<I_, 1, true>(a, p, Lt, KGS, 2 * J2, M, rhs_s, Hs, RH, 0.0, lane, hb0)
        switch (DIw) {
          case 0: RBPF_C128D(0); break;
          case 1: RBPF_C128D(1); break;
          case 2: RBPF_C128D(2); break;
          default: RBPF_C128D(3); break;
        }
#undef RBPF_C128D
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      C128_STAMP(13);
      for (int q = 0; q <= nlate; ++q) {
        // first pass: waves 1..3 two strips below (u = w - 1, w + 2); waves 4..7 their diagonal row (8 J2 + 4 + E, E = w - 4) and one
        // strip below (u = w + 2).  Later passes: two strips per worker (u = 10 + 14 (q - 1) + (w - 1) + 7 s)
        int rt[2] = {0, 0}, ns = 0;
        bool hasB = false;
        if (q == 0 && wv >= 4) {
          hasB = 4 + (wv - 4) < nd2;
          const int u = wv + 2;
          if (hasB) rt[ns++] = 8 * J2 + 4 + (wv - 4);
          if (u < n_below) rt[ns++] = first_below + u;
        } else {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int u = (q == 0) ? (wv - 1) + 3 * s : 10 + 14 * (q - 1) + (wv - 1) + 7 * s;
            if (u < n_below) rt[ns++] = first_below + u;
          }
        }
        if (ns < 2) rt[1] = rt[0];
#define RBPF_C128(NS_, DI_, F_) c128_pass<NS_, DI_, F_>(a, p, Lt, KGS, J2, rt, hasB, nd2, M, rhs_s, Hs, RH, lane, hb0, hb1, Lds2 C128_STAMP_PASS)
        if (q > 0) {
          if (ns == 2) RBPF_C128(2, -1, false); else if (ns == 1) RBPF_C128(1, -1, false);
        } else if (wv <= 3 || !hasB) {
          // (a wave 4..7 without a diagonal row -- the last, partial super-block -- runs the ordinary pass)
          if (ns == 2) RBPF_C128(2, -1, true); else if (ns == 1) RBPF_C128(1, -1, true); else RBPF_C128(0, -1, true);
        } else {
          switch (2 * DIw + (ns - 1)) {
            case 0: RBPF_C128(1, 0, true); break;
            case 1: RBPF_C128(2, 0, true); break;
            case 2: RBPF_C128(1, 1, true); break;
            case 3: RBPF_C128(2, 1, true); break;
            case 4: RBPF_C128(1, 2, true); break;
            case 5: RBPF_C128(2, 2, true); break;
            case 6: RBPF_C128(1, 3, true); break;
            default: RBPF_C128(2, 3, true); break;
          }
        }
#undef RBPF_C128
      }
      C128_STAMP(14);
    }
    __syncthreads();                                                  // BE: the super-block is visible to the next panel products
    C128_STAMP(4);
  }
#ifdef RBPF_C128_STAMPS
  if (p == 0 && lane == 0 && (wv == 0 || wv == 1 || wv == 4))
    printf("chol128 M=%d wave %d clocks: w0 wait-tiles %lld D0 %lld D1 %lld rest %lld | end-barrier %lld | diag-tiles %lld elems0 %lld product %lld wait-D0 %lld solve0 %lld update %lld elems1 %lld wait-D1 %lld solve1 %lld tail %lld\n",
           M, wv, cst[0], cst[1], cst[2], cst[3], cst[4], cst[13], cst[5], cst[6], cst[7], cst[8], cst[9], cst[12], cst[10], cst[11], cst[14]);
#endif
  const int failed = *sfail;
  __syncthreads();
  if (!failed) {
    double sl = 0.0, vv = 0.0;
    for (int j = tid; j < M; j += kThreads) {
      const size_t off = (size_t)(j >> 2) * 64 + (size_t)(j & 3) * 16;
      const double dj = Lt[(size_t)(j >> 4) * KGS * 64 + off + (j & 15)];
      const double vj = Lt[(size_t)(M >> 4) * KGS * 64 + off + (M & 15)];
      sl += log(dj);
      vv = fma(vj, vj, vv);
    }
    sl = wave_sum(sl); vv = wave_sum(vv);
    if (lane == 0) { red[wv] = sl; red[16 + wv] = vv; }
    __syncthreads();
    if (tid == 0) {
      sl = 0.0; vv = 0.0;
      for (int w = 0; w < 8; ++w) { sl += red[w]; vv += red[16 + w]; }
      a.pant_log[p] += -0.5 * a.qf[p] - a.hld[p] - sl + 0.5 * vv;     // particleSmootherInformationForm.m:234-236
    }
  } else if (tid == 0) {
    atomicOr(a.status, 2);                                            // quirk Q4: a failed information-form factorisation is an error
    a.pant_log[p] = nan("");
  }
}

static hipError_t launch_chol128(const CholArgs& ca, int batch, int d_lds, hipStream_t st) {
  static std::atomic<uint64_t> attr{0};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&chol_solve128_kernel), (int)kC128MaxLds, attr)) return e;
  CholArgs cb = ca;
  cb.batch = batch;
  hipLaunchKernelGGL(chol_solve128_kernel, dim3(batch), dim3(512), chol128_lds_bytes(ca.Msz, d_lds), st, cb);
  return hipGetLastError();
}

// usable for: information form, more than 27 row tiles (the 8-wave shape), LDS fits
static bool chol128_ok(const CholArgs& ca, int d_lds) {
  const int RT = (ca.Msz + 1 + 15) >> 4;
  return ca.mode == 1 && RT > 27 && ca.l_slots == 0 && chol128_lds_bytes(ca.Msz, d_lds) <= kC128MaxLds;
}
