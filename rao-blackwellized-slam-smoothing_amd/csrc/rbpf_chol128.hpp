// Ancestor-weight factorisation of the information form, 128-column variant (included by rbpf_smoother.hip after rbpf_chol64.hpp,
// whose element loaders, diagonal-block routine and argument block it shares).
//
//   particleSmootherInformationForm.m:224-236   cIend = chol(Imat_i + ImatAddt), v = cIend \ (ivec_i + ivecAddt)
//
// Why a third kernel.  Counters of the 64-column kernel at n = 515 (profiles/r03_mag_chol64_pmc_summary.txt): 85 GB per launch of
// 8192 matrices against 19 GB of matrix in / out -- the left-looking panel products re-read the finished factor once per 64 columns
// as B operands (3.05 MB per matrix) and the rows of the diagonal block as A operands by seven waves each (1.2-1.9 MB reach the
// memory), at 5.4 TB/s: the kernel is bound by traffic it creates itself.  Here
//
//   * a SUPER-BLOCK is 128 columns (8 sub-columns of 16): the panel product of a row tile accumulates 8 tiles per pass over the
//     finished columns, so the factor is re-read once per 128 columns (B operands: n^3 / 768 elements);
//   * the A operands -- the eight row tiles of the super-block's diagonal rows -- are staged ONCE per pass through a ring of LDS
//     buffers (16 KB chunks of 4 column groups x 8 row tiles) by wave 0, which has nothing else to do during the products; the
//     seven worker waves take them from LDS (one producer; a monotonic fill counter and one drain counter per ring slot in LDS, bounded waits); the diagonal
//     rows' own products take BOTH operands from the ring, so those rows are read once per super-block in all;
//   * the 128 x 128 diagonal block is factorised as two 64 x 64 halves by wave 0 with the routine of the 64-column kernel
//     (c64_diag_block); between the halves the strips below solve their first four sub-columns, keep the solved tiles in registers
//     (they are the B operands of the next step as they stand) and update their last four sub-columns with the rows 4..7 of the
//     diagonal block, published in LDS by their owners (a K = 64 product that touches no memory).
//
// Work split (8 waves, one workgroup per CU): wave 0 = ring producer + the two diagonal blocks; waves 7, 6, 5, 4 form the diagonal
// strips (d_i, d_{7-i}), i = 0..3 (9 tiles each) and carry one strip below; waves 1..3 carry two strips below.  Row tiles beyond
// the ten of the first pass run in further passes of fourteen (two per worker), with the diagonal block's operands still in LDS.
//
// Factor storage: as in the 64-column kernel (row-tile major, fragment order; complete, so the carried-factor refresh can read it).
#pragma once

#ifndef RBPF_C128_DBG
#define RBPF_C128_DBG 0
#endif
#ifndef RBPF_C128_NB
#define RBPF_C128_NB 3                           // ring buffers (16 KB each)
#endif
#ifdef RBPF_C128_STAMPS                          // tuning aid: per-phase clocks of waves 0 / 1 / 4 of workgroup 0
#define C128_STAMP(k) do { const long long now_ = clock64(); cst[k] += now_ - clast; clast = now_; } while (0)
#define C128_STAMP_ARGS , long long (&cst)[14], long long& clast
#define C128_STAMP_PASS , cst, clast
#else
#define C128_STAMP(k) do { } while (0)
#define C128_STAMP_ARGS
#define C128_STAMP_PASS
#endif

constexpr int kC128Workers = 7;
constexpr int kC128ChunkDoubles = 2048;          // [4 column groups][8 row tiles][64]
constexpr int kC128MaxSpins = 1 << 18;         // ~ 10 ms: a legitimate wait is a few thousand clocks

__device__ inline bool c128_wait_ge(int* ctr, int target) {
  int spins = 0;
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target && spins < kC128MaxSpins) {
    __builtin_amdgcn_s_sleep(1);
    ++spins;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return spins < kC128MaxSpins;
}

// Wave 0: chunks g0 .. g0 + nch - 1 of the ring = column groups 4 s .. 4 s + 3 of the eight diagonal row tiles of super-block J2.
// Two chunks of loads in flight (registers), a ring slot is rewritten once all seven workers have drained it.
__device__ inline bool c128_produce(const double* __restrict__ Lt, int KGS, int RT, int J2, int nch, int lane, double* ring,
                                    int* filled, int* done, int g0) {
  if (nch <= 0) return true;
  const double* src[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) src[r] = Lt + (size_t)min(8 * J2 + r, RT - 1) * KGS * 64 + 4 * lane;   // 256 doubles per (row tile, chunk)
  const int wofs = ((lane >> 4) * 8) * 64 + 4 * (lane & 15);                  // [column group l / 16][row tile r][4 (l % 16) ..]
  v4d R0[8], R1[8];
  bool ok = true;
#pragma unroll
  for (int r = 0; r < 8; ++r) R0[r] = *reinterpret_cast<const v4d*>(src[r]);
#pragma unroll
  for (int r = 0; r < 8; ++r) R1[r] = *reinterpret_cast<const v4d*>(src[r] + (size_t)min(1, nch - 1) * 256);
  for (int s = 0; s < nch; s += 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int sc = s + b;
      if (sc < nch) {
        const int g = g0 + sc;
        // (one drain counter PER SLOT: a sum over all chunks would let six fast workers vouch for a slow seventh)
        if (g >= RBPF_C128_NB) ok = c128_wait_ge(done + g % RBPF_C128_NB, kC128Workers * (g / RBPF_C128_NB)) && ok;
        double* dst = ring + (size_t)(g % RBPF_C128_NB) * kC128ChunkDoubles + wofs;
        const size_t nxt = (size_t)min(sc + 2, nch - 1) * 256;
        if (b == 0) {
#pragma unroll
          for (int r = 0; r < 8; ++r) *reinterpret_cast<v4d*>(dst + r * 64) = R0[r];
          C64_PIN();
#pragma unroll
          for (int r = 0; r < 8; ++r) R0[r] = *reinterpret_cast<const v4d*>(src[r] + nxt);
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) *reinterpret_cast<v4d*>(dst + r * 64) = R1[r];
          C64_PIN();
#pragma unroll
          for (int r = 0; r < 8; ++r) R1[r] = *reinterpret_cast<const v4d*>(src[r] + nxt);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(filled, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  return ok;
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

// X = V inv(Ld)' for the four sub-columns of one half (NLs: -inv(Ld_cc), Lds: Ld(c', c), both as MFMA A fragments in LDS); the solved
// tiles replace the accumulators (they are the B operands of what follows) and go to the factor at column group kg0.
template <int NS>
__device__ inline void c128_solve_half(v4d (&Z)[2][2][4], int h, const double* NLs, const double* Lds, double* __restrict__ Lt, int KGS,
                                       const int (&rt)[2], int kg0, int lane) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double ni[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ni[q] = NLs[(c * 4 + q) * 64 + lane];
    v4d x[NS > 0 ? NS : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
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
      for (int s = 0; s < NS; ++s) Z[s][h][cp] = mfma4(lf, x[s], Z[s][h][cp]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) Z[s][h][c] = x[s];
  }
}

// One pass of a worker wave over super-block J2: NS strips below the diagonal block (row tiles rt[0..NS)), and -- DI >= 0, first pass
// only -- the diagonal strips d_DI (tiles 0..DI) and d_{7-DI} (tiles 0..3 of the first half, 0..3-DI of the second).
// FIRST: the pass that contains the diagonal strips; it runs the five workgroup barriers wave 0 runs.
template <int NS, int DI, bool FIRST>
__device__ inline void c128_pass(const CholArgs& a, int p, double* __restrict__ Lt, int KGS, int RT, int J2, const int (&rt)[2], int nd2,
                                 int M, const double* rhs_s, const double* Hs, const double* RH, double jit, int lane, double* hb0, double* hb1,
                                 double* Lds2, const double* ring, int* filled, int* done, int g0, int nch, int* sfail, v4d (&Zn)[2][4], bool pre,
                                 const int (&rtn)[2], int nsn, int J2n C128_STAMP_ARGS) {
  constexpr int I = DI < 0 ? 0 : DI, E = DI < 0 ? 0 : 3 - DI;
  constexpr bool DG = DI >= 0;
  const bool hasA = DG && I < nd2, hasB = DG && 4 + E < nd2;                 // wave-uniform
  v4d Z[2][2][4];                                                            // [strip][half][sub-column]
  v4d ZA[I + 1], ZB0[4], ZB1[E + 1];
  // ---- elements of the FIRST half (the second half's are subtracted after the product, while wave 0 factorises) ---------------
  // pre: this pass's strips came prefetched from the previous pass (Zn: loaded while that pass waited for its second diagonal block)
  if (DG) {
    if (hasA) c128_tri64<I>(a, p, 2 * J2, M, rhs_s, Hs, RH, jit, lane, ZA);
    else {
#pragma unroll
      for (int c = 0; c <= I; ++c) ZA[c] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) ZB0[c] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c <= E; ++c) ZB1[c] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (hasB) c128_strip64<true>(a, p, 8 * J2 + 4 + E, 2 * J2, M, rhs_s, Hs, RH, jit, lane, ZB0);
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (pre) {
#pragma unroll
      for (int c = 0; c < 4; ++c) Z[s][0][c] = Zn[s][c];
    } else if (DG) {
#pragma unroll
      for (int c = 0; c < 4; ++c) Z[s][0][c] = (v4d){0.0, 0.0, 0.0, 0.0};
      c128_strip64<true>(a, p, rt[s], 2 * J2, M, rhs_s, Hs, RH, jit, lane, Z[s][0]);
    } else {
      c128_strip64<false>(a, p, rt[s], 2 * J2, M, rhs_s, Hs, RH, jit, lane, Z[s][0]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) Z[s][1][c] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
  C128_STAMP(5);
  // ---- panel product over the finished columns: A operands (and, for the diagonal strips, B operands) from the ring -----
  if (nch > 0) {
    const double* pb[NS > 0 ? NS : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) pb[s] = Lt + (size_t)rt[s] * KGS * 64;       // wave-uniform base, + lane per load
    double B0[NS > 0 ? NS : 1][4], B1[NS > 0 ? NS : 1][4];
    C64_PIN();
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int k = 0; k < 4; ++k) B0[s][k] = (pb[s] + (size_t)k * 64)[lane];
    C64_PIN();
    bool ok = true;
    auto chunk = [&](int sc, double (&Bc)[NS > 0 ? NS : 1][4], double (&Bn)[NS > 0 ? NS : 1][4]) {
      const size_t kn = (size_t)min(sc + 1, nch - 1) * 4 * 64;
      C64_PIN();
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) Bn[s][k] = (pb[s] + kn + (size_t)k * 64)[lane];
      C64_PIN();
      const int g = g0 + sc;
      ok = c128_wait_ge(filled, g + 1) && ok;
      const double* rp = ring + (size_t)(g % RBPF_C128_NB) * kC128ChunkDoubles + lane;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double F[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) F[r] = rp[(k * 8 + r) * 64];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int c = 0; c < 8; ++c) Z[s][c >> 2][c & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[c], Bc[s][k], Z[s][c >> 2][c & 3], 0, 0, 0);
        if (DG) {
#pragma unroll
          for (int c = 0; c <= I; ++c) ZA[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[c], F[I], ZA[c], 0, 0, 0);
#pragma unroll
          for (int c = 0; c < 4; ++c) ZB0[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[c], F[4 + E], ZB0[c], 0, 0, 0);
#pragma unroll
          for (int c = 0; c <= E; ++c) ZB1[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[4 + c], F[4 + E], ZB1[c], 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(done + g % RBPF_C128_NB, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (int sc = 0; sc < nch; sc += 2) {                                     // (nch = 8 J2: even)
      chunk(sc, B0, B1);
      chunk(sc + 1, B1, B0);
    }
    if (!ok && lane == 0) *sfail = 1;
  }
  C128_STAMP(6);
  // ---- first half: hand d_DI to wave 0, wait for its factorisation, solve ---------------------------------------------
  const int nh1 = max(0, nd2 - 4);
  if (FIRST) {
    if (DG && hasA) {
#pragma unroll
      for (int c = 0; c <= I; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) hb0[(c64_tri(I, c) * 4 + q) * 64 + lane] = ZA[c][q];
    }
    __syncthreads();                                                          // B1: the first half's diagonal tiles are in LDS
  }
  // the second half's elements, while wave 0 factorises the first diagonal block (FIRST) -- the product ran from zero
#pragma unroll
  for (int s = 0; s < NS; ++s) c128_strip64<true>(a, p, rt[s], 2 * J2 + 1, M, rhs_s, Hs, RH, jit, lane, Z[s][1]);
  if (DG && hasB) {
    v4d T[E + 1];
    c128_tri64<E>(a, p, 2 * J2 + 1, M, rhs_s, Hs, RH, jit, lane, T);
#pragma unroll
    for (int c = 0; c <= E; ++c) ZB1[c] += T[c];
  }
  C128_STAMP(12);
  if (FIRST) __syncthreads();                                                 // B2: wave 0 has factorised the first half
  C128_STAMP(7);
  const double* NL0 = hb0;
  const double* Ld0 = hb0 + 1024;
  if (nd2 > 4) {                                                              // (strips below the first half exist)
    c128_solve_half<NS>(Z, 0, NL0, Ld0, Lt, KGS, rt, 32 * J2, lane);
    if (DG && hasB) {                                                         // d_{4+E}: a strip below the first half
      v4d ZT[2][2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) ZT[0][0][c] = ZB0[c];
      const int rtb[2] = {8 * J2 + 4 + E, 8 * J2 + 4 + E};
      c128_solve_half<1>(ZT, 0, NL0, Ld0, Lt, KGS, rtb, 32 * J2, lane);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ZB0[c] = ZT[0][0][c];
#pragma unroll
        for (int q = 0; q < 4; ++q) Lds2[((E * 4 + c) * 4 + q) * 64 + lane] = ZB0[c][q];
      }
    }
  }
  C128_STAMP(8);
  if (FIRST) __syncthreads();                                                 // B3: rows 4..7 of the diagonal block, first half, are in LDS
  // ---- second half: update with the first half's solved tiles (no memory), hand d_{4+E} to wave 0, solve --------------
  if (nh1 > 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (NS > 0 || (DG && e <= E)) {
          double lf[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) lf[q] = Lds2[((e * 4 + c) * 4 + q) * 64 + lane];
#pragma unroll
          for (int s = 0; s < NS; ++s) Z[s][1][e] = mfma4(lf, Z[s][0][c], Z[s][1][e]);
          if (DG && e <= E) ZB1[e] = mfma4(lf, ZB0[c], ZB1[e]);
        }
      }
  }
  if (FIRST) {
    if (DG && hasB) {
#pragma unroll
      for (int c = 0; c <= E; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) hb1[(c64_tri(E, c) * 4 + q) * 64 + lane] = ZB1[c][q];
    }
    C128_STAMP(9);
    __syncthreads();                                                          // B4: the second half's diagonal tiles are in LDS
  }
  // the next pass's first-half elements, while wave 0 factorises the second diagonal block (FIRST): the solved first-half tiles
  // are dead by now, their registers take the prefetch
  auto prefetch = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (s < nsn) {
        c128_strip64<false>(a, p, rtn[s], 2 * J2n, M, rhs_s, Hs, RH, jit, lane, Zn[s]);
      } else {                                                                // (every pass redefines all of Zn: nothing of it lives through a pass)
#pragma unroll
        for (int c = 0; c < 4; ++c) Zn[s][c] = (v4d){0.0, 0.0, 0.0, 0.0};
      }
  };
  if (FIRST) {
    prefetch();
    C128_STAMP(13);
    __syncthreads();                                                          // B5: wave 0 has factorised the second half
    C128_STAMP(10);
  }
  if (nd2 == 8) c128_solve_half<NS>(Z, 1, hb1, hb1 + 1024, Lt, KGS, rt, 32 * J2 + 16, lane);
  if (!FIRST) prefetch();
  C128_STAMP(11);
}

constexpr size_t kC128MaxLds = 160 * 1024;
static size_t chol128_lds_doubles(int M, int d) {
  return (size_t)5120 + 4096 + (size_t)RBPF_C128_NB * kC128ChunkDoubles + 32 + M + 4 + 2 * (size_t)d * M;
}
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
  // LDS: hb0 / hb1 [2560] the two halves' diagonal tiles on their way to wave 0, then -inv(Ld_cc) [4][4][64] and Ld(c',c) [6][4][64];
  //      Lds2 [4][4][4][64] rows 4..7 of the diagonal block, first-half columns, as MFMA A fragments; the ring; scalars and vectors
  double* hb0 = csm;
  double* hb1 = csm + 2560;
  double* Lds2 = csm + 5120;
  double* ring = Lds2 + 4096;
  double* red = ring + (size_t)RBPF_C128_NB * kC128ChunkDoubles;     // [32]
  double* rhs_s = red + 32;                                           // [M]
  int* flags = reinterpret_cast<int*>(rhs_s + M);                     // sfail, filled, done[NB <= 6]: 4 doubles
  const bool pend = a.Hb != nullptr;
  double* Hs = pend ? rhs_s + M + 4 : nullptr;
  double* RH = pend ? Hs + (size_t)a.d * M : nullptr;
  chol_prologue(a, p, tid, kThreads, M, rhs_s, Hs, RH, pend);
  int* sfail = flags;
  int* filled = flags + 1;
  int* done = flags + 2;
  if (tid < 8) flags[tid] = 0;
  __syncthreads();
#ifdef RBPF_C128_STAMPS
  long long cst[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, clast = clock64();
#endif
  const double jit = 0.0;
  const int NJ2 = (RT + 7) >> 3;
  int g = 0;                                                          // chunks staged so far (every wave keeps the same count)
  v4d Zn[2][4];                                                       // workers: the next pass's first-half elements, prefetched
  bool pre = false;
  for (int J2 = 0; J2 < NJ2; ++J2) {
    const int nd2 = min(8, RT - 8 * J2), nh0 = min(4, nd2), nh1 = nd2 - nh0;
    const int first_below = 8 * J2 + 8, n_below = max(0, RT - first_below), nch = 8 * J2;
    const int nlate = n_below > 10 ? (n_below - 10 + 13) / 14 : 0;
    if (wv == 0) {
      bool ok = c128_produce(Lt, KGS, RT, J2, nch, lane, ring, filled, done, g);
      g += nch;
      C128_STAMP(0);
      __syncthreads();                                                // B1
      bool bad = c64_diag_block(Lt, KGS, 2 * J2, nh0, M, lane, hb0, hb0, hb0 + 1024);
      C128_STAMP(1);
      __syncthreads();                                                // B2
      __syncthreads();                                                // B3
      __syncthreads();                                                // B4
      if (nh1 > 0) bad = c64_diag_block(Lt, KGS, 2 * J2 + 1, nh1, M, lane, hb1, hb1, hb1 + 1024) || bad;
      C128_STAMP(2);
      __syncthreads();                                                // B5
      for (int lp = 0; lp < nlate; ++lp) {
        ok = c128_produce(Lt, KGS, RT, J2, nch, lane, ring, filled, done, g) && ok;
        g += nch;
      }
      if ((bad || !ok) && lane == 0) *sfail = 1;
      C128_STAMP(3);
    } else {
      // pass q of super-block J2: first pass = waves 1..3 two strips below (u = w - 1, w + 2), waves 4..7 the diagonal strips and one
      // strip below (u = w + 2); later passes two strips per worker (u = 10 + 14 (q - 1) + (w - 1) + 7 s)
      auto assign = [&](int J2a, int q, int (&rta)[2]) -> int {
        const int fb = 8 * J2a + 8, nb = max(0, RT - fb);
        int ns = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int u = (q == 0) ? ((wv <= 3) ? (wv - 1) + 3 * s : (s == 0 ? wv + 2 : nb)) : 10 + 14 * (q - 1) + (wv - 1) + 7 * s;
          rta[s] = fb + min(u, max(nb - 1, 0));
          ns += (u < nb) ? 1 : 0;
        }
        return ns;
      };
      for (int q = 0; q <= nlate; ++q) {
        int rt[2], rtn[2] = {0, 0};
        const int ns = assign(J2, q, rt);
        const int J2n = (q < nlate) ? J2 : J2 + 1, qn = (q < nlate) ? q + 1 : 0;
        const int nsn = (J2n < NJ2) ? assign(J2n, qn, rtn) : 0;
#define RBPF_C128(NS_, DI_, F_) c128_pass<NS_, DI_, F_>(a, p, Lt, KGS, RT, J2, rt, nd2, M, rhs_s, Hs, RH, jit, lane, hb0, hb1, Lds2, ring, filled, done, g, nch, sfail, Zn, pre, rtn, nsn, J2n C128_STAMP_PASS)
        if (q > 0) {
          if (ns == 2) RBPF_C128(2, -1, false); else if (ns == 1) RBPF_C128(1, -1, false); else RBPF_C128(0, -1, false);
        } else if (wv <= 3) {
          if (ns == 2) RBPF_C128(2, -1, true); else if (ns == 1) RBPF_C128(1, -1, true); else RBPF_C128(0, -1, true);
        } else {
          switch (2 * (7 - wv) + ns) {
            case 0: RBPF_C128(0, 0, true); break;
            case 1: RBPF_C128(1, 0, true); break;
            case 2: RBPF_C128(0, 1, true); break;
            case 3: RBPF_C128(1, 1, true); break;
            case 4: RBPF_C128(0, 2, true); break;
            case 5: RBPF_C128(1, 2, true); break;
            case 6: RBPF_C128(0, 3, true); break;
            default: RBPF_C128(1, 3, true); break;
          }
        }
#undef RBPF_C128
        pre = true;
        g += nch;
      }
      C128_STAMP(1);
    }
    __syncthreads();                                                  // B6: the super-block is visible to the next panel products
    C128_STAMP(4);
  }
#ifdef RBPF_C128_STAMPS
  if (p == 0 && lane == 0 && (wv == 0 || wv == 1 || wv == 4))
    printf("chol128 M=%d wave %d clocks: w0 produce %lld D0 %lld D1 %lld late-produce %lld | end-barrier %lld | elems0 %lld product %lld elems1 %lld wait-D0 %lld solve0 %lld miniP %lld prefetch %lld wait-D1 %lld solve1 %lld\n",
           M, wv, cst[0], cst[1], cst[2], cst[3], cst[4], cst[5], cst[6], cst[12], cst[7], cst[8], cst[9], cst[13], cst[10], cst[11]);
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
