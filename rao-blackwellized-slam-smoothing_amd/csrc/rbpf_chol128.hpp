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
//   * the 128 x 128 diagonal block is factorised as two 64 x 64 halves by wave 0 with the routine of the 64-column kernel
//     (c64_diag_block); between the halves a strip solves its first four sub-columns, keeps the solved tiles in registers (they are the
//     B operands of the next step as they stand) and updates its last four sub-columns with rows 4..7 of the diagonal block
//     (a K = 64 product whose B operands never touch memory);
//   * the workgroup is TWO TEAMS that only meet through flags in LDS, never at a workgroup barrier (a first version that ran all
//     eight waves in lockstep -- elements, product, wait for the diagonal block, solve -- took 18.0 ms per 8192 matrices against the
//     64-column kernel's 15.5: every phase idled either the memory or the matrix pipes):
//       - the CRITICAL team, wave 0 + waves 4..7, carries the chain  rows of the next diagonal block -> their panel product ->
//         64 x 64 factorisation -> solve -> update -> 64 x 64 factorisation.  Per super-block s the four waves first finish the
//         eight row tiles of block s as strips BELOW block s - 1 (two each), then form them as DIAGONAL strips of block s
//         (d_i and d_{7-i}: 9 tiles each); both products take their A operands -- and the diagonal strips their B operands too --
//         from a ring of 16 KB LDS chunks that wave 0 stages (one producer, a fill counter and one drain counter per slot);
//       - the BULK team, waves 1..3, owns every row tile from the second block below the diagonal on (row tile r belongs to wave
//         1 + r % 3 for the whole factorisation, so it meets no other wave's writes of its own rows), two strips at a time, A
//         operands straight from the L2: elements, product, then -- when the flags say the diagonal block is there -- solve,
//         update, solve.  Its element streams and products overlap the critical team's serial sections by construction.
//     The operands of the solves (-inv(Ld_cc), Ld(c', c), rows 4..7 of the first half) are read from the factor itself, where wave 0
//     and the critical team leave them (the inverse of a diagonal tile in the otherwise unused tile to its right).
//
// Factor storage: as in the 64-column kernel (row-tile major, fragment order; complete, so the carried-factor refresh can read it).
#pragma once

#ifndef RBPF_C128_NB
#define RBPF_C128_NB 4                           // ring buffers (16 KB each)
#endif
#ifndef RBPF_C128_GRING
#define RBPF_C128_GRING 3                        // bulk team: column groups of A / B operand loads in flight
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

constexpr int kC128Crit = 4;                     // consumers of the ring: waves 4..7
constexpr int kC128ChunkDoubles = 2048;          // [4 column groups][8 row tiles][64]
constexpr int kC128MaxSpins = 1 << 18;           // ~ 10 ms: a legitimate wait is a few ten thousand clocks

// LDS flags (ints)
enum { C128_SFAIL = 0, C128_FILLED, C128_CB, C128_DDONE, C128_X2DONE, C128_C1DONE, C128_DONE0 /* [NB <= 8] */, C128_BPRIO0 = C128_DONE0 + 8 /* [16] */,
       C128_NFLAGS = C128_BPRIO0 + 16 };

__device__ inline bool c128_wait_ge(int* ctr, int target) {
  int spins = 0;
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target && spins < kC128MaxSpins) {
    __builtin_amdgcn_s_sleep(1);
    ++spins;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return spins < kC128MaxSpins;
}

// my global stores are on their way to the L2 (which the whole workgroup reads through one L1) before the flag moves
__device__ inline void c128_publish(int* ctr, int add, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_fetch_add(ctr, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// barrier of the critical team (five waves), k = how many the caller has passed
__device__ inline bool c128_cbar(int* cb, int& k, int lane) {
  c128_publish(cb, 1, lane);
  ++k;
  return c128_wait_ge(cb, 5 * k);
}

// Wave 0: chunks g0 .. g0 + nch - 1 of the ring = column groups 4 s .. 4 s + 3 of the eight row tiles rt0 .. rt0 + 7.
// Two chunks of loads in flight (registers); a ring slot is rewritten once all four consumers have drained it.
__device__ inline bool c128_produce(const double* __restrict__ Lt, int KGS, int RT, int rt0, int nch, int lane, double* ring,
                                    int* flags, int g0) {
  if (nch <= 0) return true;
  const double* src[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) src[r] = Lt + (size_t)min(rt0 + r, RT - 1) * KGS * 64 + 4 * lane;   // 256 doubles per (row tile, chunk)
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
        // (one drain counter PER SLOT: a sum over all chunks would let fast consumers vouch for a slow one)
        if (g >= RBPF_C128_NB) ok = c128_wait_ge(flags + C128_DONE0 + g % RBPF_C128_NB, kC128Crit * (g / RBPF_C128_NB)) && ok;
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
        __hip_atomic_store(flags + C128_FILLED, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

// Where the factorised diagonal block of 64-column block J leaves the operands of the solves, all as MFMA A fragments (+ q * 64):
//   -inv(Ld_cc)   in the tile to the right of diagonal tile c (row tile 4 J + c, column tile 4 J + c + 1: never part of the factor)
//   Ld(cp, c)     the factor's own tile (row tile 4 J + cp, sub-column c of the block)
__device__ inline const double* c128_nl(const double* Lt, int KGS, int J, int c) { return Lt + ((size_t)(4 * J + c) * KGS + 4 * (4 * J + c + 1)) * 64; }
__device__ inline const double* c128_ld(const double* Lt, int KGS, int J, int cp, int c) { return Lt + ((size_t)(4 * J + cp) * KGS + 16 * J + 4 * c) * 64; }

// X = V inv(Ld)' for the four sub-columns of half h of the strips' super-block (64-column block J); the solved tiles replace the
// accumulators (they are the B operands of what follows) and go to the factor.
template <int NS>
__device__ inline void c128_solve_half(v4d (&Z)[2][2][4], int h, double* __restrict__ Lt, int KGS, int J, const int (&rt)[2], int lane) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double ni[4];
    const double* np_ = c128_nl(Lt, KGS, J, c) + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) ni[q] = np_[q * 64];
    v4d x[NS > 0 ? NS : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      x[s] = mfma4(ni, Z[s][h][c], (v4d){0.0, 0.0, 0.0, 0.0});
      double* dx = Lt + ((size_t)rt[s] * KGS + 16 * J + 4 * c) * 64 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) dx[q * 64] = x[s][q];
    }
#pragma unroll
    for (int cp = c + 1; cp < 4; ++cp) {
      double lf[4];
      const double* lp_ = c128_ld(Lt, KGS, J, cp, c) + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) lf[q] = lp_[q * 64];
#pragma unroll
      for (int s = 0; s < NS; ++s) Z[s][h][cp] = mfma4(lf, x[s], Z[s][h][cp]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) Z[s][h][c] = x[s];
  }
}

// second half += first half's solved tiles * (rows 4..7 of the diagonal block, first-half columns)'   (operands from the factor)
template <int NS>
__device__ inline void c128_update_half(v4d (&Z)[2][2][4], const double* __restrict__ Lt, int KGS, int j, int lane) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double lf[4];
      const double* lp_ = Lt + ((size_t)(8 * j + 4 + e) * KGS + 32 * j + 4 * c) * 64 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) lf[q] = lp_[q * 64];
#pragma unroll
      for (int s = 0; s < NS; ++s) Z[s][1][e] = mfma4(lf, Z[s][0][c], Z[s][1][e]);
    }
}

// NS strips (row tiles rt[0..NS)) below the diagonal block of super-block j, from their elements to the factor.
// RING: the critical team's pass -- A operands from the LDS ring (chunks g0 .. g0 + 8 j - 1), its flags are behind it by
// construction except the second diagonal block's; otherwise the bulk team's -- A operands from the L2, every operand behind a flag.
template <int NS, bool RING>
__device__ inline void c128_below(const CholArgs& a, int p, double* __restrict__ Lt, int KGS, int j, const int (&rt)[2], int M,
                                  const double* rhs_s, const double* Hs, const double* RH, int lane, const double* ring, int* flags, int g0,
                                  int x2_target, bool& ok C128_STAMP_ARGS) {
  v4d Z[2][2][4];                                                            // [strip][half][sub-column]
  const int nch = 8 * j;
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) c128_strip64<false>(a, p, rt[s], 2 * j + h, M, rhs_s, Hs, RH, 0.0, lane, Z[s][h]);
  C128_STAMP(5);
  if (nch > 0) {
    const double* pb[NS > 0 ? NS : 1];
#pragma unroll
    for (int s = 0; s < NS; ++s) pb[s] = Lt + (size_t)rt[s] * KGS * 64;       // wave-uniform base, + lane per load
    if (RING) {
      double B0[NS > 0 ? NS : 1][4], B1[NS > 0 ? NS : 1][4];
      C64_PIN();
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) B0[s][k] = (pb[s] + (size_t)k * 64)[lane];
      C64_PIN();
      auto chunk = [&](int sc, double (&Bc)[NS > 0 ? NS : 1][4], double (&Bn)[NS > 0 ? NS : 1][4]) {
        const size_t kn = (size_t)min(sc + 1, nch - 1) * 4 * 64;
        C64_PIN();
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int k = 0; k < 4; ++k) Bn[s][k] = (pb[s] + kn + (size_t)k * 64)[lane];
        C64_PIN();
        const int g = g0 + sc;
        ok = c128_wait_ge(flags + C128_FILLED, g + 1) && ok;
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
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(flags + C128_DONE0 + g % RBPF_C128_NB, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      for (int sc = 0; sc < nch; sc += 2) {                                   // (nch = 8 j: even)
        chunk(sc, B0, B1);
        chunk(sc + 1, B1, B0);
      }
    } else {
      // A operands (the eight row tiles of the diagonal block) and B operands (own rows) straight from the L2, a ring of column groups
      constexpr int kR = RBPF_C128_GRING;
      const double* pa = Lt + (size_t)(8 * j) * KGS * 64;
      const size_t ts = (size_t)KGS * 64;
      const int nkg = 32 * j;
      double A[kR][8], B[kR][NS > 0 ? NS : 1];
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
  }
  C128_STAMP(6);
  if (!RING) ok = c128_wait_ge(flags + C128_DDONE, 2 * j + 1) && ok;          // the first diagonal block of super-block j
  C128_STAMP(7);
  c128_solve_half<NS>(Z, 0, Lt, KGS, 2 * j, rt, lane);
  if (!RING) ok = c128_wait_ge(flags + C128_X2DONE, x2_target) && ok;         // rows 4..7 of the diagonal block, first-half columns
  C128_STAMP(8);
  c128_update_half<NS>(Z, Lt, KGS, j, lane);
  C128_STAMP(9);
  ok = c128_wait_ge(flags + C128_DDONE, 2 * j + 2) && ok;                     // the second diagonal block
  C128_STAMP(10);
  c128_solve_half<NS>(Z, 1, Lt, KGS, 2 * j + 1, rt, lane);
  C128_STAMP(11);
}

// The diagonal strips of super-block s on critical wave 7 - DI: d_DI (tiles 0..DI) and d_{7-DI} (tiles 0..3 of the first half,
// 0..3-DI of the second): elements, product with both operands from the ring, the tiles of the two 64 x 64 blocks handed to wave 0.
template <int DI>
__device__ inline void c128_diag(const CholArgs& a, int p, double* __restrict__ Lt, int KGS, int s, int nd2, int M, const double* rhs_s,
                                 const double* Hs, const double* RH, int lane, double* hb0, double* hb1, const double* ring, int* flags,
                                 int g0, int& kbar, bool& ok C128_STAMP_ARGS) {
  constexpr int I = DI, E = 3 - DI;
  const bool hasA = I < nd2, hasB = 4 + E < nd2;                             // wave-uniform
  const int nch = 8 * s;
  v4d ZA[I + 1], ZB0[4], ZB1[E + 1];
  if (hasA) c128_tri64<I>(a, p, 2 * s, M, rhs_s, Hs, RH, 0.0, lane, ZA);
  else {
#pragma unroll
    for (int c = 0; c <= I; ++c) ZA[c] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
  if (hasB) {
    c128_strip64<false>(a, p, 8 * s + 4 + E, 2 * s, M, rhs_s, Hs, RH, 0.0, lane, ZB0);
    c128_tri64<E>(a, p, 2 * s + 1, M, rhs_s, Hs, RH, 0.0, lane, ZB1);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) ZB0[c] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c <= E; ++c) ZB1[c] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
  C128_STAMP(12);
  for (int sc = 0; sc < nch; ++sc) {
    const int g = g0 + sc;
    ok = c128_wait_ge(flags + C128_FILLED, g + 1) && ok;
    const double* rp = ring + (size_t)(g % RBPF_C128_NB) * kC128ChunkDoubles + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double F[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) F[r] = rp[(k * 8 + r) * 64];
#pragma unroll
      for (int c = 0; c <= I; ++c) ZA[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[c], F[I], ZA[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) ZB0[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[c], F[4 + E], ZB0[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c <= E; ++c) ZB1[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[4 + c], F[4 + E], ZB1[c], 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(flags + C128_DONE0 + g % RBPF_C128_NB, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  C128_STAMP(13);
  if (hasA) {
#pragma unroll
    for (int c = 0; c <= I; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) hb0[(c64_tri(I, c) * 4 + q) * 64 + lane] = ZA[c][q];
  }
  ok = c128_cbar(flags + C128_CB, kbar, lane) && ok;                          // (2) the first half's diagonal tiles are in LDS
  ok = c128_cbar(flags + C128_CB, kbar, lane) && ok;                          // (3) wave 0 has factorised them (operands in the factor)
  C128_STAMP(14);
  if (hasB) {                                                                 // d_{4+E} is a strip below the first half: solve
    v4d ZT[2][2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) ZT[0][0][c] = ZB0[c];
    const int rtb[2] = {8 * s + 4 + E, 8 * s + 4 + E};
    c128_solve_half<1>(ZT, 0, Lt, KGS, 2 * s, rtb, lane);
#pragma unroll
    for (int c = 0; c < 4; ++c) ZB0[c] = ZT[0][0][c];
  }
  ok = c128_cbar(flags + C128_CB, kbar, lane) && ok;                          // (4) rows 4..7, first-half columns, are in the factor
  if (hasB) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e <= E; ++e) {
        double lf[4];
        const double* lp_ = Lt + ((size_t)(8 * s + 4 + e) * KGS + 32 * s + 4 * c) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) lf[q] = lp_[q * 64];
        ZB1[e] = mfma4(lf, ZB0[c], ZB1[e]);
      }
#pragma unroll
    for (int c = 0; c <= E; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) hb1[(c64_tri(E, c) * 4 + q) * 64 + lane] = ZB1[c][q];
  }
  ok = c128_cbar(flags + C128_CB, kbar, lane) && ok;                          // (5) the second half's diagonal tiles are in LDS
  C128_STAMP(15);
}

constexpr size_t kC128MaxLds = 160 * 1024;
static size_t chol128_lds_doubles(int M, int d) {
  return (size_t)5120 + (size_t)RBPF_C128_NB * kC128ChunkDoubles + 32 + M + C128_NFLAGS / 2 + 2 * (size_t)d * M;
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
  // LDS: hb0 / hb1 [2560] the two halves' diagonal tiles on their way to wave 0 (then its scratch for -inv(Ld_cc), Ld(c',c)); the ring;
  //      scalars, vectors, flags
  double* hb0 = csm;
  double* hb1 = csm + 2560;
  double* ring = csm + 5120;
  double* red = ring + (size_t)RBPF_C128_NB * kC128ChunkDoubles;     // [32]
  double* rhs_s = red + 32;                                           // [M]
  int* flags = reinterpret_cast<int*>(rhs_s + M);                     // [C128_NFLAGS]
  const bool pend = a.Hb != nullptr;
  double* Hs = pend ? rhs_s + M + C128_NFLAGS / 2 : nullptr;
  double* RH = pend ? Hs + (size_t)a.d * M : nullptr;
  chol_prologue(a, p, tid, kThreads, M, rhs_s, Hs, RH, pend);
  if (tid < C128_NFLAGS) flags[tid] = 0;
  __syncthreads();
#ifdef RBPF_C128_STAMPS
  long long cst[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, clast = clock64();
#endif
  const int NJ2 = (RT + 7) >> 3;
  bool ok = true;
  if (wv == 0) {
    // ---- wave 0: ring producer of the critical team and the diagonal blocks ---------------------------------------------
    int g = 0, kbar = 0;
    bool bad = false;
    auto diag_block = [&](int J, int nd, double* hb) {
      bad = c64_diag_block(Lt, KGS, J, nd, M, lane, hb, hb, hb + 1024) || bad;
#pragma unroll
      for (int c = 0; c < 4; ++c)                                     // -inv(Ld_cc) into the tile to the right of the diagonal tile
        if (c < nd && 4 * J + c + 1 < RT) {
          double* dst = Lt + ((size_t)(4 * J + c) * KGS + 4 * (4 * J + c + 1)) * 64 + lane;
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[q * 64] = hb[(c * 4 + q) * 64 + lane];
        }
      c128_publish(flags + C128_DDONE, 1, lane);
    };
    for (int s = 0; s < NJ2; ++s) {
      const int nd2 = min(8, RT - 8 * s), nh0 = min(4, nd2), nh1 = nd2 - nh0;
      if (s >= 1) {                                                   // rows of block s as strips below block s - 1
        ok = c128_produce(Lt, KGS, RT, 8 * (s - 1), 8 * (s - 1), lane, ring, flags, g) && ok;
        g += 8 * (s - 1);
      }
      C128_STAMP(0);
      ok = c128_cbar(flags + C128_CB, kbar, lane) && ok;             // (1) rows of block s are final left of it
      ok = c128_produce(Lt, KGS, RT, 8 * s, 8 * s, lane, ring, flags, g) && ok;
      g += 8 * s;
      C128_STAMP(1);
      ok = c128_cbar(flags + C128_CB, kbar, lane) && ok;             // (2)
      diag_block(2 * s, nh0, hb0);
      C128_STAMP(2);
      ok = c128_cbar(flags + C128_CB, kbar, lane) && ok;             // (3)
      ok = c128_cbar(flags + C128_CB, kbar, lane) && ok;             // (4)
      ok = c128_cbar(flags + C128_CB, kbar, lane) && ok;             // (5)
      if (nh1 > 0) diag_block(2 * s + 1, nh1, hb1);
      else c128_publish(flags + C128_DDONE, 1, lane);
      C128_STAMP(3);
    }
    if ((bad || !ok) && lane == 0) flags[C128_SFAIL] = 1;
  } else if (wv >= 4) {
    // ---- critical team ------------------------------------------------------------------------------------------------
    int g = 0, kbar = 0;
    const int DIw = 7 - wv;
    for (int s = 0; s < NJ2; ++s) {
      const int nd2 = min(8, RT - 8 * s);
      if (s >= 1) {
        // my two rows of block s (8 s + DI, 8 s + 7 - DI) as strips below block s - 1; their columns left of block s - 1 are the bulk
        // team's work at the super-blocks before
        int rt[2] = {8 * s + DIw, 8 * s + 7 - DIw};
        const int ns = (rt[0] < RT ? 1 : 0) + (rt[1] < RT ? 1 : 0);
        if (rt[0] >= RT) rt[0] = RT - 1;
        if (rt[1] >= RT) rt[1] = rt[0];
        if (s >= 2) ok = c128_wait_ge(flags + C128_BPRIO0 + (s - 2), min(8, RT - 8 * s)) && ok;
#define RBPF_C128B(NS_) c128_below<NS_, true>(a, p, Lt, KGS, s - 1, rt, M, rhs_s, Hs, RH, lane, ring, flags, g, 0, ok C128_STAMP_PASS)
        if (ns == 2) RBPF_C128B(2); else if (ns == 1) RBPF_C128B(1); else RBPF_C128B(0);
#undef RBPF_C128B
        g += 8 * (s - 1);
      }
      ok = c128_cbar(flags + C128_CB, kbar, lane) && ok;             // (1)
      if (lane == 0 && wv == 4) __hip_atomic_store(flags + C128_C1DONE, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#define RBPF_C128D(DI_) c128_diag<DI_>(a, p, Lt, KGS, s, nd2, M, rhs_s, Hs, RH, lane, hb0, hb1, ring, flags, g, kbar, ok C128_STAMP_PASS)
      switch (DIw) {
        case 0: RBPF_C128D(0); break;
        case 1: RBPF_C128D(1); break;
        case 2: RBPF_C128D(2); break;
        default: RBPF_C128D(3); break;
      }
#undef RBPF_C128D
      g += 8 * s;
      // (after barrier (4) every existing row 4..7 of block s has its first-half columns in the factor)
      if (lane == 0 && wv == 4) __hip_atomic_store(flags + C128_X2DONE, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (!ok && lane == 0) flags[C128_SFAIL] = 1;
  } else {
    // ---- bulk team: row tile r belongs to wave 1 + r % 3 --------------------------------------------------------------
    for (int j = 0; 8 * (j + 2) < RT; ++j) {
      // the diagonal rows of super-block j are final left of it once the critical team has passed barrier (1) of step j
      ok = c128_wait_ge(flags + C128_C1DONE, j) && ok;
      int r0 = 8 * (j + 2);
      r0 += ((wv - 1) - r0 % 3 + 3) % 3;                              // my first row tile
      for (; r0 < RT; r0 += 6) {
        int rt[2] = {r0, r0 + 3};
        const int ns = (rt[1] < RT) ? 2 : 1;
        if (ns == 1) rt[1] = rt[0];
#define RBPF_C128B(NS_) c128_below<NS_, false>(a, p, Lt, KGS, j, rt, M, rhs_s, Hs, RH, lane, ring, flags, 0, j + 1, ok C128_STAMP_PASS)
        if (ns == 2) RBPF_C128B(2); else RBPF_C128B(1);
#undef RBPF_C128B
        const int prio = (rt[0] < 8 * (j + 3) ? 1 : 0) + ((ns == 2 && rt[1] < 8 * (j + 3)) ? 1 : 0);   // rows of the next diagonal block but one
        if (prio) c128_publish(flags + C128_BPRIO0 + j, prio, lane);
      }
    }
    if (!ok && lane == 0) flags[C128_SFAIL] = 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                    // the only workgroup barrier after the prologue
#ifdef RBPF_C128_STAMPS
  if (p == 0 && lane == 0 && (wv == 0 || wv == 1 || wv == 4))
    printf("chol128 M=%d wave %d clocks: w0 produce1 %lld produce2 %lld D0 %lld D1 %lld | below: elems %lld product %lld wait-D0 %lld solve0 %lld update %lld wait-D1 %lld solve1 %lld | diag: elems %lld product %lld wait-D0 %lld rest %lld\n",
           M, wv, cst[0], cst[1], cst[2], cst[3], cst[5], cst[6], cst[7], cst[8], cst[9], cst[10], cst[11], cst[12], cst[13], cst[14], cst[15]);
#endif
  const int failed = flags[C128_SFAIL];
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
  return ca.mode == 1 && RT > 27 && RT <= 16 * 8 && ca.l_slots == 0 && chol128_lds_bytes(ca.Msz, d_lds) <= kC128MaxLds;
}
