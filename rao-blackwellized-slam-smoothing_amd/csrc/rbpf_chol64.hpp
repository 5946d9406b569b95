// Ancestor-weight factorisation, 64-column variant (included by rbpf_smoother.hip inside namespace rbpf, after the
// 16-column kernel whose argument block, element loader and scalar helpers it shares).
//
//   particleSmoother.m:221-229                  cS = chol(S,'lower') (+ jitter retry), v = cS \ e, sum(log(diag(cS)))
//   particleSmootherInformationForm.m:224-236   cIend = chol(Imat_i + ImatAddt), v = cIend \ (ivec_i + ivecAddt)
//
// Why a second kernel.  The 16-column left-looking kernel re-reads the finished factor once per 16 columns:
// n^3/96 elements = 12 MB per particle at n = 515, 100 GB per launch of 8192 particles, which caps it at ~14 ms even at
// the HBM rate, and every block column has a one-wave diagonal tile with fifteen waves waiting.  Here a block column is
// 64 wide (4 sub-columns of 16), so the factor is re-read four times less, and the serial part runs beside the bulk:
//
//   * one workgroup = 8 wave64 (256 registers each) per particle;
//   * wave 0 owns the FACTORISATION of the 64 x 64 diagonal block (row tiles 4J..4J+3): the four 16 x 16 tile
//     factorisations and the solves / updates between them stay in that wave's registers — no barrier inside;
//   * waves 7, 6, 5, 4 first form row tile 0, 1, 2, 3 of that block (elements + panel product) and hand it to wave 0
//     through LDS (release / acquire on an LDS counter; only wave 0 waits, bounded);
//   * waves 1..7 own the row tiles below (up to 4 each, 16 accumulators = one 16 x 64 strip per tile) and run their
//     panel product  Z += L(diag rows, k) * L(own rows, k)'  meanwhile (A operand: fragments of the four diagonal-block
//     rows, B operand: fragments of the own rows, four-deep register ring of 512 B operand loads, pinned with scheduling
//     barriers: left alone, the compiler gathers the ring's loads at the top of the loop and consumes them at once);
//   * ONE barrier; then waves 1..7 solve their strips against the diagonal block: X_c = V_c * inv(Ld_cc)', and
//     V_c' -= X_c * Ld(c',c)' for c' > c, with -inv(Ld_cc) and Ld(c',c) read from LDS as ready-made MFMA operands;
//   * a second barrier publishes the block column.
//
// Accumulators hold Z = -(A - W) so that neither the products nor the updates need a negated operand; the diagonal tile
// negates its four registers once.  The diagonal tile is factorised *in the MFMA operand layout* (lane l: row l & 15,
// columns (l >> 4) + 4 q), four columns (= one register) at a time: ds_bpermute broadcasts inside the 4-column panel,
// ONE MFMA per panel for the trailing columns, the inverse swept alongside (also one MFMA per panel), so the tile never
// goes through LDS and the inverse is directly an MFMA A operand (4.5 K clocks per tile instead of 15 K).
//
// Measured (MI355X, n = 515): 3.50 ms per 2048 matrices standalone (26.6 TFLOP/s = 34 % of the fp64 matrix peak; the
// 16-column kernel: 5.1-5.7 ms), 15.8 ms per launch of 8192 inside the information-form smoother (r01: 29.5 ms, r02: 16.6 ms), where
// it moves ~80 GB (stored matrix + ImatAddt + factor re-reads in, factor + Imat(:,:,ai) out) at 5-5.5 TB/s: bound by the memory
// system (DESIGN.md 4.3 has the measurements: per-CU rates, clock stamps, CU mask, loader variants).
//
// Element loaders.  chol_aug_elems (a call: c64_elems_general) handles every edge -- clamps, triangle mask, jitter, right-hand-side
// row; interior strips, the tiles of interior diagonal blocks and the last row tile have call-free loaders with all loads of a
// strip in flight at once (c64_strip_fast, c64_diag_elems_fast).  The information matrices may come in packed block-lower storage
// (CholArgs::imat_packed, imat_packed_index in rbpf_smoother.hip): a 16 x 64 strip is then 8 KB of consecutive memory.
//
// Factor storage: row-tile major, fragment order — the 64 values L(16 rt + r, 4 kg + kk) sit at
// ((rt * KGS + kg) * 64 + kk * 16 + r), KGS = 4 RT, so a row tile streams through consecutive 512 B fragments.
#pragma once
#include <type_traits>

// keeps the operand ring as written: without it the compiler gathers the ring's loads at the top of the loop body and
// consumes them in the same iteration (no prefetch distance left)
#define C64_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

#ifndef RBPF_C64_LATE
#define RBPF_C64_LATE 0
#endif
#ifndef RBPF_C64_LASTFAST
#define RBPF_C64_LASTFAST 1                      // call-free loader for the last row tile (right-hand-side row)
#endif
#ifndef RBPF_C64_LOOKAHEAD
#define RBPF_C64_LOOKAHEAD 0                     // 1: the diagonal block's tiles are formed one block column ahead (off the serial chain).
                                                 // Measured r03 (n = 515, 8192 matrices, in the smoother): 16.2 ms against 15.8 ms with 0 -- it
                                                 // reads the next diagonal rows before their owners stream them (+ 0.8 MB per particle) and
                                                 // the kernel is bound by the memory system
#endif
#ifndef RBPF_C64_RING4
#define RBPF_C64_RING4 4                         // operand ring depth of the four-row-tile panel product (3: sixteen registers less)
#endif
#ifndef RBPF_C64_DIAGFAST
#define RBPF_C64_DIAGFAST 1                      // call-free loader for the tiles of an interior diagonal block
#endif
#ifndef RBPF_C64_LARING
#define RBPF_C64_LARING 8                        // operand ring depth of the look-ahead diagonal products
#endif
#ifndef RBPF_C64_INT1
#define RBPF_C64_INT1 4                          // information form: first row tile below the diagonal block that takes the fast loader
#endif
#ifdef RBPF_C64_STAMPS                           // tuning aid: per-phase clocks of waves 0 / 1 of workgroup 0
#define C64_STAMP(k) do { const long long now_ = clock64(); cst[k] += now_ - clast; clast = now_; } while (0)
#define C64_STAMP_ARGS , long long (&cst)[8], long long& clast
#define C64_STAMP_PASS , cst, clast
#else
#define C64_STAMP(k) do { } while (0)
#define C64_STAMP_ARGS
#define C64_STAMP_PASS
#endif

// Two shapes: W = 8 waves of 256 registers, up to 4 row tiles per worker wave, one workgroup per CU (large matrices);
// W = 4 waves, up to 2 row tiles per worker wave, two workgroups per CU (12..27 row tiles: the serial diagonal blocks
// of two particles overlap).  Three or four workgroups per CU would need a call-free kernel below 170 registers; inlined,
// the diagonal block and the general element loader spill (389 registers), and a callee does not honour the kernel's bound.

// Pointers that reach a NON-inlined function as arguments, or are read from the kernel-argument segment there, are generic to the
// compiler: it emits flat_load / flat_store, which count on BOTH memory counters -- every LDS wait (ds_bpermute in the tile
// factorisation, operand reads) then also waits for the factor stores on their way to memory.  These casts restore global_* accesses.
typedef __attribute__((address_space(1))) double c64_gdouble;
__device__ inline c64_gdouble* c64_g(double* p) { return (c64_gdouble*)p; }
__device__ inline const c64_gdouble* c64_g(const double* p) { return (const c64_gdouble*)p; }
typedef __attribute__((address_space(3))) double c64_ldouble;                  // ... and ds_* accesses for LDS pointers
__device__ inline c64_ldouble* c64_l(double* p) { return (c64_ldouble*)p; }
__device__ inline const c64_ldouble* c64_l(const double* p) { return (const c64_ldouble*)p; }

__device__ inline double bperm_f64(double v, int src_lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b & 0xffffffffLL));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

#ifndef RBPF_DIAG_BCAST_SWAP
#define RBPF_DIAG_BCAST_SWAP 0                   // 1: lane broadcasts of the 16 x 16 tile factorisation without the LDS crossbar (below).  Measured
                                                 // r04 with wave 0 on the critical path (RBPF_C64_SOLO_MAXJ = 5): 15.1 ms per 8192 against 14.9 with
                                                 // ds_bpermute -- the tile factorisation itself gets shorter (0.13 -> 0.09 M clocks per matrix), the
                                                 // chain as a whole does not
#endif
// every lane (g, r) gets the value of lane (KK, r): v_permlane16_swap leaves rows [0, 0, 2, 2] / [1, 1, 3, 3] of a register swapped with
// itself, v_permlane32_swap then the lower / upper half twice -- four VALU operations for a double, no LDS round trip
__device__ __forceinline__ double bcast_group_f64(double v, int KK) {   // (KK: a constant after unrolling)
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto p0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto p1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const unsigned alo = (KK & 1) ? p0[1] : p0[0], ahi = (KK & 1) ? p1[1] : p1[0];
  const auto q0 = __builtin_amdgcn_permlane32_swap(alo, alo, false, false);
  const auto q1 = __builtin_amdgcn_permlane32_swap(ahi, ahi, false, false);
  return (KK >> 1) ? __hiloint2double((int)q1[1], (int)q0[1]) : __hiloint2double((int)q1[0], (int)q0[0]);
}

__device__ inline v4d mfma4(const double (&a)[4], const v4d& b, v4d acc) {
#pragma unroll
  for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
  return acc;
}
__device__ inline v4d mfma4(const v4d& a, const v4d& b, v4d acc) {
#pragma unroll
  for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
  return acc;
}

// 16 x 16 Cholesky of V (lower triangle used) and the negated inverse of its factor, both in the MFMA operand layout
// "natural": lane l holds row r = l & 15, columns g + 4 q (g = l >> 4) in register q.  Columns >= nvalid are padding
// (pivot := 1).  Returns true when a pivot was not positive.
//
// Blocked by 4 columns = one register: inside a panel the four columns sit in the four lane groups, a column step is
// pivot -> sqrt / 1/sqrt (uniform) -> scale -> two ds_bpermute broadcasts -> one fma; the trailing columns take ONE
// MFMA per panel (D = P P' with the panel register as both operands; by symmetry the result layout is the natural one).
// The inverse X = inv(Ld) is swept alongside in the MFMA *result* layout (lane: column, register/group: row), where
// the rank-4 update of the rows below a panel is again one MFMA (A = panel of Ld, B = the four finished rows of X);
// a final product with -I transposes it into the natural layout, i.e. into a ready-made A operand.
__device__ inline bool chol_diag_tile_frag(v4d& V, v4d& NI, int nvalid, int lane) {
  const int r = lane & 15, g = lane >> 4;
  bool bad = false;
  double Lv[4], XT[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { Lv[q] = V[q]; XT[q] = (r == 4 * q + g) ? 1.0 : 0.0; }
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 4 * kb + kk;
      double piv = readlane_f64(Lv[kb], kk * 16 + k);
      if (k >= nvalid) piv = 1.0;
      bad |= !(piv > 0.0);
      double ljj, rinv;
      sqrt_rsqrt(piv, ljj, rinv);
      const double cv = (r == k) ? ljj : ((r > k) ? Lv[kb] * rinv : 0.0);
      Lv[kb] = (g == kk) ? cv : Lv[kb];                                     // column k of Ld, row k of X are final
      XT[kb] *= (g == kk) ? rinv : 1.0;                                     // (selects, not branches: this is a serial chain)
      if (kk < 3) {
#if RBPF_DIAG_BCAST_SWAP
        // (the serial chain of a block column is sixty-four of these steps: with ds_bpermute each one is two LDS round trips)
        const double lrk = bcast_group_f64(Lv[kb], kk);                     // Ld(r, k)
        const double xk = bcast_group_f64(XT[kb], kk);                      // X(k, c), c = lane & 15
        double mck = 0.0;                                                   // Ld(4 kb + g, k) for the groups right of column k
#pragma unroll
        for (int gg = kk + 1; gg < 4; ++gg) {
          const double lg = readlane_f64(Lv[kb], kk * 16 + 4 * kb + gg);
          mck = (g == gg) ? lg : mck;
        }
#else
        const double lrk = bperm_f64(Lv[kb], kk * 16 + r);                  // Ld(r, k)
        const double lck = bperm_f64(Lv[kb], kk * 16 + 4 * kb + g);         // Ld(4 kb + g, k)
        const double xk = bperm_f64(XT[kb], kk * 16 + r);                   // X(k, c), c = lane & 15
        const double mck = (g > kk) ? lck : 0.0;
#endif
        Lv[kb] = fma(-lrk, mck, Lv[kb]);
        XT[kb] = fma(-mck, xk, XT[kb]);
      }
    }
    if (kb < 3) {
      const v4d zero = (v4d){0.0, 0.0, 0.0, 0.0};
      const v4d D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lv[kb], Lv[kb], zero, 0, 0, 0);
      const v4d D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lv[kb], XT[kb], zero, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q > kb) { Lv[q] -= D1[q]; XT[q] -= D2[q]; }
    }
  }
  v4d ni = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    V[q] = Lv[q];
    ni = __builtin_amdgcn_mfma_f64_16x16x4f64(XT[q], (r == 4 * q + g) ? -1.0 : 0.0, ni, 0, 0, 0);
  }
  NI = ni;
  return bad;
}

__device__ inline const CholArgs* c64_kernarg() {   // the kernel's only parameter, at offset 0 of its argument segment
  return (const CholArgs*)__builtin_amdgcn_kernarg_segment_ptr();
}

// The general element loader (chol_aug_elems: clamps, triangle mask, jitter, right-hand-side row) as a real call.  It
// is only needed for the few edge strips of a block column; inlined sixteen times per pass its hoisted index arithmetic
// was kept in scratch (91 spill stores per block column and thread = 1.1 MB of extra HBM writes per matrix at n = 515).
// The argument block reaches it by pointer to the kernel-argument segment (by value it would travel through the stack of
// every lane); imat_src is the resolved stored matrix of this particle (ancestor's bank entry or received record).
template <int MODE>
__device__ __attribute__((noinline)) v4d c64_elems_general(const CholArgs* ka, const double* imat_src, int p, int i, int jb, int M,
                                                           const double* rhs_s, const double* Hs, const double* RH, double jit) {
  CholArgs a = *ka;
  if (MODE == 1) { a.Imat = imat_src; a.imat_stride = 0; }
  v4d e;
  chol_aug_elems<MODE>(a, p, i, jb, M, rhs_s, Hs, RH, jit, e);
  return e;
}

// (Negated) matrix elements of an INTERIOR strip — row tile rt_s strictly below the diagonal block, not the last row
// tile, block column entirely left of column M — i.e. no index clamp, no triangle mask, no jitter, no right-hand-side
// row (and, covariance form, at least 17 rows below the diagonal: no kron(I, R) entry).  Same operations in the same
// order as chol_aug_elems; what goes is its per-element address and predicate arithmetic (the general loader is
// VALU-bound: ~1.2 K clocks per four elements, 45 % of the information-form kernel at n = 515): one per-lane offset
// r + ld g, everything else wave-uniform, so every load / store is scalar base + that one register.
// LAST: the last row tile (matrix rows up to M - 1, the right-hand-side row M, padding): the row index is clamped per lane for the
// loads, nothing is stored for rows >= M, row M takes rhs_s.
template <int MODE, bool ACC, bool LAST = false>
__device__ inline void c64_strip_fast(const CholArgs& a, int p, int rt_s, int J, int M, const double* Hs, const double* RH,
                                      int lane, v4d (&Zs)[4], const double* rhs_s = nullptr) {
  const int g = lane >> 4, i = 16 * rt_s + (lane & 15);
  const int r = LAST ? min(lane & 15, M - 1 - 16 * rt_s) : (lane & 15);   // (LAST needs M % 16 != 0: the tile holds a matrix row; otherwise
  const int ld = (MODE == 0) ? M : a.n;                                   //  the general loader takes it)
  const unsigned lo = (unsigned)(r + ld * g);                                     // unsigned: scalar base + 32-bit lane offset addressing
  const bool pk = (MODE == 1) && a.imat_packed;   // packed storage (imat_packed_index): the strip is 8 KB of consecutive memory
  const size_t t0 = pk ? imat_packed_row(rt_s) + (size_t)(16 * J) * 64 : (size_t)16 * rt_s + (size_t)ld * (64 * J);   // wave-uniform
  const double* src = ((MODE == 0) ? a.S + (size_t)p * M * M : a.Imat + (size_t)p * a.imat_stride) + t0;
  const double* add = (MODE == 1) ? a.ImatAdd + t0 : nullptr;
  double* dst = (MODE == 1 && a.ImatOut) ? a.ImatOut + (size_t)p * a.imat_out_stride + t0 : nullptr;
  const double* hrow = Hs ? Hs + 16 * rt_s + r : nullptr;                    // Hs[aa * M + i]
  const double* rcol = RH ? RH + 64 * J + g : nullptr;                       // RH[aa * M + j]
  // all 32 loads of the strip in flight before the first use (one memory round trip per strip, not one per 16 columns:
  // the element phase is latency-bound -- 40 % of a worker wave's clocks with the loads issued 16 columns at a time)
  // (ACC: the accumulators are live -- half a strip at a time)
  // The sixteen column offsets are kept as per-lane 32-bit registers, opaque to the compiler: as wave-uniform bases (its choice when
  // it sees through them) three streams x sixteen columns are 48 scalar register pairs -- they overflow into vector lanes and from
  // there into scratch.
  constexpr int kB = ACC ? 2 : 4;
  unsigned off[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) { off[c][q] = pk ? (unsigned)(lane + 64 * (4 * c + q)) : lo + (unsigned)(ld * (16 * c + 4 * q)); asm volatile("" : "+v"(off[c][q])); }
  double ad[4][4], v[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c % kB == 0) {
      C64_PIN();
#pragma unroll
      for (int cc = c; cc < c + kB; ++cc)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[cc][q] = src[off[cc][q]];
          ad[cc][q] = (MODE == 1) ? add[off[cc][q]] : 0.0;
        }
      C64_PIN();
    }
    if (MODE == 1) {
      if (Hs) {                                                              // + dyi'/R*dyi of the last update (:334)
        double sacc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int aa = 0; aa < a.d; ++aa) {
          const double h = hrow[aa * M];
#pragma unroll
          for (int q = 0; q < 4; ++q) sacc[q] = fma(h, rcol[aa * M + 16 * c + 4 * q], sacc[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[c][q] += sacc[q];
      }
      if (dst && (!LAST || i < M)) {                                         // Imat(:,:,i) of the new generation
#pragma unroll
        for (int q = 0; q < 4; ++q) __builtin_nontemporal_store(v[c][q], &dst[off[c][q]]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) v[c][q] += ad[c][q];                       // :225
    }
    if (LAST) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[c][q] = (i < M) ? v[c][q] : ((i == M) ? rhs_s[64 * J + 16 * c + 4 * q + g] : 0.0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) Zs[c][q] = ACC ? Zs[c][q] - v[c][q] : -v[c][q];
  }
}

__device__ inline int c64_pair(int cp, int c) { return cp * (cp - 1) / 2 + c; }   // (c' > c) -> 0..5

__device__ inline int c64_tri(int i, int c) { return i * (i + 1) / 2 + c; }         // (i >= c) -> 0..9

// Elements of row tile I of an INTERIOR diagonal block (64 J + 64 <= M: no clamp, no right-hand-side row): the I + 1 lower
// tiles, all loads in flight together.  Same operations in the same order as chol_aug_elems.
template <int I, int MODE>
__device__ inline void c64_diag_elems_fast(const CholArgs& a, int p, int J, int M, const double* Hs, const double* RH, double jit,
                                           int lane, v4d (&Z)[I + 1]) {
  const int r = lane & 15, g = lane >> 4;
  const int ld = (MODE == 0) ? M : a.n;
  const unsigned lo = (unsigned)(r + ld * g);
  const int i0 = 16 * (4 * J + I);
  const bool pk = (MODE == 1) && a.imat_packed;                               // packed storage (imat_packed_index)
  const size_t t0 = pk ? imat_packed_row(4 * J + I) + (size_t)(16 * J) * 64 : (size_t)i0 + (size_t)ld * (64 * J);   // wave-uniform
  const double* src = ((MODE == 0) ? a.S + (size_t)p * M * M : a.Imat + (size_t)p * a.imat_stride) + t0;
  const double* add = (MODE == 1) ? a.ImatAdd + t0 : nullptr;
  double* dst = (MODE == 1 && a.ImatOut) ? a.ImatOut + (size_t)p * a.imat_out_stride + t0 : nullptr;
  unsigned off[I + 1][4];                                                     // (per-lane offsets, opaque: see c64_strip_fast)
#pragma unroll
  for (int c = 0; c <= I; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) { off[c][q] = pk ? (unsigned)(lane + 64 * (4 * c + q)) : lo + (unsigned)(ld * (16 * c + 4 * q)); asm volatile("" : "+v"(off[c][q])); }
  double v[I + 1][4], ad[I + 1][4];
  C64_PIN();
#pragma unroll
  for (int c = 0; c <= I; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[c][q] = src[off[c][q]];
      ad[c][q] = (MODE == 1) ? add[off[c][q]] : 0.0;
    }
  C64_PIN();
#pragma unroll
  for (int c = 0; c <= I; ++c) {
    if (MODE == 0) {
      if (a.R) {                                                             // kron(eye, R)
        const int* dv = reinterpret_cast<const int*>(Hs);
        const int di = dv[i0 + r];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dj = dv[64 * J + 16 * c + 4 * q + g];
          const double rr = a.R[(di & 7) + a.d * (dj & 7)];
          v[c][q] += ((di >> 3) == (dj >> 3)) ? rr : 0.0;
        }
      }
    } else {
      if (Hs) {                                                              // + dyi'/R*dyi of the last update (:334)
        double sacc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int aa = 0; aa < a.d; ++aa) {
          const double h = Hs[aa * M + i0 + r];
#pragma unroll
          for (int q = 0; q < 4; ++q) sacc[q] = fma(h, RH[aa * M + 64 * J + 16 * c + 4 * q + g], sacc[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[c][q] += sacc[q];
      }
      if (dst) {                                                             // Imat(:,:,i) of the new generation
#pragma unroll
        for (int q = 0; q < 4; ++q)                                            // (packed storage: the lower triangle only)
          if (!pk || 16 * I + r >= 16 * c + 4 * q + g) __builtin_nontemporal_store(v[c][q], &dst[off[c][q]]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) v[c][q] += ad[c][q];                       // :225
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * I + r, col = 16 * c + 4 * q + g;                  // inside the block
      if (row == col) v[c][q] += jit;
      Z[c][q] = (row >= col) ? -v[c][q] : 0.0;
    }
  }
}

// Row tile I of the diagonal block of block column J (one of waves 4..7): elements and panel product of its I + 1 lower tiles,
// handed to wave 0 through LDS (Zd: [10][4][64]).  RBPF_C64_LOOKAHEAD = 1: formed ONE BLOCK COLUMN AHEAD, over the block columns
// < J - 1 only (block column J - 1 is still being solved: wave 0 adds that part before it factorises).
template <int I, int MODE, bool CALLS>
__device__ inline void c64_diag_product(const CholArgs& a, int p, const double* __restrict__ Lt, int KGS, int J, int M,
                                        const double* rhs_s, const double* Hs, const double* RH, double jit, int lane,
                                        double* Zd C64_STAMP_ARGS) {
  v4d Z[I + 1];
  if (RBPF_C64_DIAGFAST && 64 * J + 64 <= M) {
    c64_diag_elems_fast<I, MODE>(a, p, J, M, Hs, RH, jit, lane, Z);
  } else {
#pragma unroll
    for (int c = 0; c <= I; ++c) {
      if (CALLS) {
        Z[c] = -c64_elems_general<MODE>(c64_kernarg(), a.Imat, p, 16 * (4 * J + I) + (lane & 15), 64 * J + 16 * c + (lane >> 4), M, rhs_s, Hs, RH, jit);
      } else {
        v4d e;
        chol_aug_elems<MODE>(a, p, 16 * (4 * J + I) + (lane & 15), 64 * J + 16 * c + (lane >> 4), M, rhs_s, Hs, RH, jit, e);
        Z[c] = -e;
      }
    }
  }
  C64_STAMP(0);
  if (J > RBPF_C64_LOOKAHEAD) {                   // look-ahead: block column J - 1 is not final yet, wave 0 adds its part
    const double* pf[I + 1];
#pragma unroll
    for (int c = 0; c <= I; ++c) pf[c] = Lt + (size_t)(4 * J + c) * KGS * 64;   // wave-uniform bases, + lane per load
    const int nkg = 16 * (J - RBPF_C64_LOOKAHEAD);
    constexpr int kRing = RBPF_C64_LARING;                      // I + 1 products per column group cover little latency: a deep ring
    double F[kRing][I + 1];
#pragma unroll
    for (int b = 0; b < kRing; ++b) {
      C64_PIN();
#pragma unroll
      for (int c = 0; c <= I; ++c) F[b][c] = (pf[c] + (size_t)min(b, nkg - 1) * 64)[lane];
      C64_PIN();
    }
    for (int kg = 0; kg < nkg; kg += kRing) {
#pragma unroll
      for (int b = 0; b < kRing; ++b) {
#pragma unroll
        for (int c = 0; c <= I; ++c) Z[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[b][c], F[b][I], Z[c], 0, 0, 0);
        const size_t kn = (size_t)min(kg + kRing + b, nkg - 1) * 64;
        C64_PIN();
#pragma unroll
        for (int c = 0; c <= I; ++c) F[b][c] = (pf[c] + kn)[lane];
        C64_PIN();
      }
    }
  }
#pragma unroll
  for (int c = 0; c <= I; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) Zd[(c64_tri(I, c) * 4 + q) * 64 + lane] = Z[c][q];
  C64_STAMP(1);
}

// Factorisation of a diagonal block whose (negated) tiles Z(i, c), c <= i, sit in this wave's registers: the four 16 x 16 tile
// factorisations with the solves / updates between them; the factor goes to Lt, -inv(Ld_cc) and Ld(c', c) to LDS as MFMA operands.
__device__ inline bool c64_diag_factor(v4d (&Z)[4][4], double* __restrict__ Lt, int KGS, int J, int nd, int M, int lane, double* NLs,
                                       double* Lds C64_STAMP_ARGS) {
  bool bad = false;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < nd) {
      v4d V = -Z[c][c], NI;

      bad |= chol_diag_tile_frag(V, NI, M - (64 * J + 16 * c), lane);
      C64_STAMP(2);
      c64_gdouble* dst = c64_g(Lt) + ((size_t)(4 * J + c) * KGS + 16 * J + 4 * c) * 64 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) { dst[q * 64] = V[q]; c64_l(NLs)[(c * 4 + q) * 64 + lane] = NI[q]; }
      v4d xs[4];
#pragma unroll
      for (int i = c + 1; i < 4; ++i) {
        xs[i] = (v4d){0.0, 0.0, 0.0, 0.0};
        if (i < nd) {
          xs[i] = mfma4(NI, Z[i][c], xs[i]);                                // X' = inv(Ld) V'  (NI = -inv, Z = -V')
          c64_gdouble* dx = c64_g(Lt) + ((size_t)(4 * J + i) * KGS + 16 * J + 4 * c) * 64 + lane;
#pragma unroll
          for (int q = 0; q < 4; ++q) { dx[q * 64] = xs[i][q]; c64_l(Lds)[(c64_pair(i, c) * 4 + q) * 64 + lane] = xs[i][q]; }
        }
      }
      C64_STAMP(7);
#pragma unroll
      for (int i = c + 1; i < 4; ++i)
#pragma unroll
        for (int cp = c + 1; cp <= i; ++cp) Z[i][cp] = mfma4(xs[cp], xs[i], Z[i][cp]);   // Z(i,c') += Ld(c',c) X(i,c)'
      C64_STAMP(4);
    }
  }
  return bad;
}

// Wave 0: factorisation of the diagonal block of block column J from the tiles waves 4..7 left in LDS.  nd = number of
// its row tiles that exist (4 except at the very end).
// (noinline: as a real call it gets registers of its own — inlined, the values the kernel keeps live across the block
// column loop pushed parts of this serial chain into scratch, 15 K clocks per tile instead of 4 K)
__device__ inline bool c64_diag_block_body(double* __restrict__ Lt, int KGS, int J, int nd, int M, int lane,
                                           const double* Zd, double* NLs, double* Lds C64_STAMP_ARGS) {
  v4d Z[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c <= i; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) Z[i][c][q] = (i < nd) ? c64_l(Zd)[(c64_tri(i, c) * 4 + q) * 64 + lane] : 0.0;
  if (RBPF_C64_LOOKAHEAD && J > 0) {
    // the part of the panel product the look-ahead could not have: block column J - 1, final since the last barrier
    const int RTc = KGS >> 2;
    const c64_gdouble* pf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) pf[c] = c64_g(Lt) + ((size_t)min(4 * J + c, RTc - 1) * KGS + 16 * (J - 1)) * 64;
    double F[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      C64_PIN();
#pragma unroll
      for (int c = 0; c < 4; ++c) F[b][c] = (pf[c] + (size_t)b * 64)[lane];
      C64_PIN();
    }
    for (int kg = 0; kg < 16; kg += 4) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c <= i; ++c) Z[i][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[b][c], F[b][i], Z[i][c], 0, 0, 0);
        const size_t kn = (size_t)min(kg + 4 + b, 15) * 64;
        C64_PIN();
#pragma unroll
        for (int c = 0; c < 4; ++c) F[b][c] = (pf[c] + kn)[lane];
        C64_PIN();
      }
    }
    if (nd < 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c <= i; ++c)
          if (i >= nd) Z[i][c] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
  }
  C64_STAMP(1);
  return c64_diag_factor(Z, Lt, KGS, J, nd, M, lane, NLs, Lds C64_STAMP_PASS);
}

__device__ __attribute__((noinline)) bool c64_diag_block(double* __restrict__ Lt, int KGS, int J, int nd, int M, int lane,
                                                         const double* Zd, double* NLs, double* Lds C64_STAMP_ARGS) {
  return c64_diag_block_body(Lt, KGS, J, nd, M, lane, Zd, NLs, Lds C64_STAMP_PASS);
}

// Wave 0's side of the block column loop, a function of its own (registers of its own).  Per block column: the diagonal block's
// tiles handed over by waves 4..7, factorisation, barrier A (the workers' solves may start), barrier B.  Executes exactly the
// barriers of the workers' loop.  (r04 also measured wave 0 forming the interior diagonal blocks itself and prefetching the next
// block's elements behind the workers' solves -- "variant 649": 15.6 -> 14.8 ms per 8192 stand-alone, no gain inside the smoother;
// removed in r05, see commit 4711b84 and DESIGN_NOTEBOOK.md 9.)
template <int MODE, int W>
__device__ __attribute__((noinline)) void c64_wave0_loop(double* __restrict__ Lt, int KGS, int RT, int M, int lane,
                                                         double* csm, int* sfail, int* ready C64_STAMP_ARGS) {
  const int NJ = (RT + 3) >> 2;
  int handed = 0;
  for (int J = 0; J < NJ; ++J) {
    const int nd = min(4, RT - 4 * J);
    double* NLs = csm + (size_t)(J & 1) * 2560;
    double* Lds = NLs + 1024;
    handed += nd;
    int spins = 0;
    if (!RBPF_C64_LOOKAHEAD) {                    // only this wave waits; bounded, so a lost hand-off cannot hang the GPU
      while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < handed && spins < (1 << 24)) {
        __builtin_amdgcn_s_sleep(4);
        ++spins;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    C64_STAMP(0);
    const bool bad = c64_diag_block(Lt, KGS, J, nd, M, lane, NLs, NLs, Lds C64_STAMP_PASS) || spins >= (1 << 24);
    if (bad && lane == 0) sfail[J & 1] = 1;
    __syncthreads();                              // A
    C64_STAMP(3);
    __syncthreads();                              // B: the block column is visible to the next panel products
    C64_STAMP(5);
#if !defined(RBPF_C64_DIAG_AFIXED) && !defined(RBPF_C64_DIAG_BFIXED)
    if (sfail[J & 1]) break;
#endif
  }
}

// Waves 1..7: NT row tiles below the diagonal block (all four sub-columns exist: nd == 4 whenever such tiles exist).
// LATE (experiment, RBPF_C64_LATE = 1, off): the panel product runs first, from zero, and the matrix elements are
// subtracted afterwards, on waves 4..7 only, so that on every SIMD one wave streams elements while the other multiplies.
// Measured slower (information form, n = 515: 21.8 instead of 18.7 ms per launch): the element phase is bound by the
// CU's memory rate (~10 B/clk), not by a SIMD, and then competes with the factor re-reads of the products.
template <int NT, int MODE, bool LATE, bool CALLS>
__device__ inline void c64_tile_pass(const CholArgs& a, int p, double* __restrict__ Lt, int KGS, int J,
                                     const int (&rt)[4], int M, const double* rhs_s, const double* Hs, const double* RH,
                                     double jit, int lane, const double* NLs, const double* Lds, bool barrier C64_STAMP_ARGS) {
  v4d Z[NT][4];
  const int RTl = (M + 1 + 15) >> 4;
  // interior strip (wave-uniform): the fast loader; LATE defers exactly these (the general loader is a call: with the accumulators
  // live around it they would all be saved and restored)
  auto interior = [&](int s) { return rt[s] >= 4 * J + (MODE == 0 ? 5 : RBPF_C64_INT1) && rt[s] < RTl - 1 && 64 * J + 64 <= M; };   // (MODE 0: no kron(I, R) entry)
#pragma unroll
  for (int s = 0; s < NT; ++s) {
    if (interior(s)) {
      if (LATE && J > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) Z[s][c] = (v4d){0.0, 0.0, 0.0, 0.0};
      } else {
        c64_strip_fast<MODE, false>(a, p, rt[s], J, M, Hs, RH, lane, Z[s]);
      }
    } else if (RBPF_C64_LASTFAST && (M & 15) != 0 && rt[s] == RTl - 1 && rt[s] >= 4 * J + (MODE == 0 ? 5 : 4) && 64 * J + 64 <= M) {
      c64_strip_fast<MODE, false, true>(a, p, rt[s], J, M, Hs, RH, lane, Z[s], rhs_s);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v4d e;
        if (CALLS) e = c64_elems_general<MODE>(c64_kernarg(), a.Imat, p, 16 * rt[s] + (lane & 15), 64 * J + 16 * c + (lane >> 4), M, rhs_s, Hs, RH, jit);
        else chol_aug_elems<MODE>(a, p, 16 * rt[s] + (lane & 15), 64 * J + 16 * c + (lane >> 4), M, rhs_s, Hs, RH, jit, e);
        Z[s][c] = -e;
      }
    }
  }
  C64_STAMP(0);
  if (J > 0) {
    const double* pa = Lt + (size_t)(4 * J) * KGS * 64;                     // wave-uniform bases, + lane per load
    const size_t ts = (size_t)KGS * 64;                                     // stride between row tiles
    const double* pb[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) pb[s] = Lt + (size_t)rt[s] * ts;
    const int nkg = 16 * J;
    constexpr int kR = (NT == 4) ? RBPF_C64_RING4 : 4;                      // ring depth (column groups in flight)
    double A[kR][4], B[kR][NT];
#pragma unroll
    for (int b = 0; b < kR; ++b) {
      C64_PIN();                                                            // same issue order as in the loop: the wait
#pragma unroll                                                              // counts at the loop head then match
      for (int c = 0; c < 4; ++c) A[b][c] = (pa + c * ts + (size_t)b * 64)[lane];
#pragma unroll
      for (int s = 0; s < NT; ++s) B[b][s] = (pb[s] + (size_t)b * 64)[lane];
      C64_PIN();
    }
    for (int kg = 0; kg < nkg; kg += kR) {
#pragma unroll
      for (int b = 0; b < kR; ++b) {
        if (kR == 4 || kg + b < nkg) {                                      // (nkg is a multiple of 4)
#pragma unroll
          for (int s = 0; s < NT; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) Z[s][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[b][c], B[b][s], Z[s][c], 0, 0, 0);
        }
        const size_t kn = (size_t)min(kg + kR + b, nkg - 1) * 64;
#ifdef RBPF_C64_DIAG_AFIXED                       // timing experiment only (wrong results): the common operand always out of the L1
        const size_t kna = (size_t)b * 64;
#else
        const size_t kna = kn;
#endif
#ifdef RBPF_C64_DIAG_BFIXED                       // likewise the private operand
        const size_t knb = (size_t)b * 64;
#else
        const size_t knb = kn;
#endif
        C64_PIN();
#pragma unroll
        for (int c = 0; c < 4; ++c) A[b][c] = (pa + c * ts + kna)[lane];
#pragma unroll
        for (int s = 0; s < NT; ++s) B[b][s] = (pb[s] + knb)[lane];
        C64_PIN();
      }
    }
  }
  if (LATE && J > 0) {
#pragma unroll
    for (int s = 0; s < NT; ++s)
      if (interior(s)) c64_strip_fast<MODE, true>(a, p, rt[s], J, M, Hs, RH, lane, Z[s]);
  }
  C64_STAMP(1);
  if (barrier) __syncthreads();                                             // the diagonal block's LDS operands are ready
  C64_STAMP(3);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double ni[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ni[q] = NLs[(c * 4 + q) * 64 + lane];
    v4d x[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) {
      x[s] = mfma4(ni, Z[s][c], (v4d){0.0, 0.0, 0.0, 0.0});
      double* dx = Lt + ((size_t)rt[s] * KGS + 16 * J + 4 * c) * 64 + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) dx[q * 64] = x[s][q];
    }
#pragma unroll
    for (int cp = c + 1; cp < 4; ++cp) {
      double lf[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) lf[q] = Lds[(c64_pair(cp, c) * 4 + q) * 64 + lane];
#pragma unroll
      for (int s = 0; s < NT; ++s) Z[s][cp] = mfma4(lf, x[s], Z[s][cp]);
    }
  }
  C64_STAMP(4);
}

template <int MODE, int W>
__global__ __launch_bounds__(W * 64, W == 8 ? 1 : 2) void chol_solve64_kernel(CholArgs a_in) {
  constexpr int kThreads = W * 64, kNTMax = (W == 8) ? 4 : 2, kTilesPerPass = kNTMax * (W - 1);
  extern __shared__ double csm[];
  // l_slots > 0 (r03 experiment, off in the product: see chol64_workspace_slots): PERSISTENT workgroups -- the grid has l_slots
  // workgroups, each walks the particles blockIdx.x, blockIdx.x + gridDim.x, ... and keeps its factor in ONE workspace slot.
  const int nslots = a_in.l_slots;
  const int p_end = nslots > 0 ? a_in.batch : (int)blockIdx.x + 1, p_step = nslots > 0 ? (int)gridDim.x : 1;
  for (int p = blockIdx.x; p < p_end; p += p_step) {
  // a reused slot: the previous particle's factor may still sit in this CU's vector L1 (stores write through to the L2 but do not
  // update lines other waves cached) -- drop it before the new factor is read back
  if (nslots > 0 && p != (int)blockIdx.x) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  {
  CholArgs a = a_in;
  const int tid = threadIdx.x, M = a.Msz;
  if (MODE == 1) {
    const int src = a.imat_anc ? a.imat_anc[p] : p;
    const bool remote = a.rec != nullptr && src >= a.n_bank_local;
    a.Imat = remote ? a.rec + (size_t)(src - a.n_bank_local) * a.rec_stride + a.rec_off_Imat
                    : a.Imat + (size_t)src * a.imat_stride;
    a.imat_stride = 0;
  }
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile indices and operand bases in SGPRs
  const int RT = (M + 1 + 15) >> 4, KGS = 4 * RT;
  double* Lt = a.Lbuf + (size_t)(nslots > 0 ? (int)blockIdx.x : p) * a.ldL;
  // two 20 KB buffers, block column J uses buffer J & 1:
  //   Zd  [10][4][64]  the diagonal block's tiles (elements + panel product) on their way from waves 4..7 to wave 0 (with
  //                    RBPF_C64_LOOKAHEAD formed one block column ahead, during the products of block column J - 1)
  //   NLs [4][4][64]   -inv(Ld_cc) as MFMA A fragments          } written by wave 0 over Zd (which it holds in registers by then),
  //   Lds [6][4][64]   Ld(c',c), c' > c, as MFMA A fragments    } read by the workers' solves of block column J
  double* red = csm + 5120;                       // [32]
  double* rhs_s = red + 32;                       // [M]
  int* sfail = reinterpret_cast<int*>(rhs_s + M);   // [2]: block column J reports in slot J & 1 (sticky), so a fast wave 0 cannot
                                                    // overtake the check of the previous block column
  const bool pend = (MODE == 1 && a.Hb != nullptr);
  double* Hs = (pend || MODE == 0) ? rhs_s + M + 2 : nullptr;
  double* RH = pend ? Hs + (size_t)a.d * M : nullptr;
  chol_prologue(a, p, tid, kThreads, M, rhs_s, Hs, RH, pend);
  const int NJ = (RT + 3) >> 2;
#ifdef RBPF_C64_STAMPS
  long long cst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, clast = clock64();
#endif
  double jit = 0.0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (tid == 0) { sfail[0] = 0; sfail[1] = 0; }
    int* ready = sfail + 2;                       // (without look-ahead) tiles of the diagonal block handed over so far
    // row tile i of the diagonal block of block column JL is formed by wave W - 1 - i % (W - 1): waves 7, 6, 5, 4 (W = 8: the longest
    // row on wave 4, which shares its SIMD with wave 0, idle half of the time) or 3, 2, 1, 3 (W = 4)
    auto lookahead = [&](int JL) {
      const int ndl = min(4, RT - 4 * JL);
      double* Zd = csm + (size_t)(JL & 1) * 2560;
      for (int di = W - 1 - wv; di < ndl; di += W - 1) {
#define RBPF_C64D(I_) c64_diag_product<I_, MODE, true>(a, p, Lt, KGS, JL, M, rhs_s, Hs, RH, jit, lane, Zd C64_STAMP_PASS)
        switch (di) {
          case 0: RBPF_C64D(0); break;
          case 1: RBPF_C64D(1); break;
          case 2: RBPF_C64D(2); break;
          default: RBPF_C64D(3); break;
        }
#undef RBPF_C64D
        if (!RBPF_C64_LOOKAHEAD) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_fetch_add(ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    };
    if (tid == 0) *ready = 0;
    __syncthreads();
    if (RBPF_C64_LOOKAHEAD) {
      if (wv != 0) lookahead(0);
      __syncthreads();
    }
    if (wv == 0) {
      c64_wave0_loop<MODE, W>(Lt, KGS, RT, M, lane, csm, sfail, ready C64_STAMP_PASS);
    } else
    for (int J = 0; J < NJ; ++J) {
      const int nd = min(4, RT - 4 * J);
      const int first = 4 * J + nd, count = RT - first;
      double* NLs = csm + (size_t)(J & 1) * 2560;
      double* Lds = NLs + 1024;
      const int nwork = W - 1, widx = wv - 1;      // workers of this block column and this wave's place among them
      const int tiles_per_pass = kNTMax * nwork;
      const int npass = max(1, (count + tiles_per_pass - 1) / tiles_per_pass);
      {
        if (RBPF_C64_LOOKAHEAD) { if (J + 1 < NJ) lookahead(J + 1); }
        else lookahead(J);
        for (int pass = 0; pass < npass; ++pass) {
          int rt[4], nt = 0;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int u = tiles_per_pass * pass + widx + nwork * s;
            rt[s] = first + min(u, count - 1);
            nt += (s < kNTMax && u < count) ? 1 : 0;
          }
#define RBPF_C64(NT_, LATE_) c64_tile_pass<NT_, MODE, LATE_, true>(a, p, Lt, KGS, J, rt, M, rhs_s, Hs, RH, jit, lane, NLs, Lds, pass == 0 C64_STAMP_PASS)
#if RBPF_C64_LATE
          switch (nt == 0 ? 0 : nt + (wv >= 4 ? 4 : 0)) {   // 0: no tile in this pass (only the barrier)
#else
          switch (nt) {
#endif
            case 1: RBPF_C64(1, false); break;
            case 2: RBPF_C64(2, false); break;
            case 3: if (kNTMax >= 3) RBPF_C64(3, false); break;
            case 4: if (kNTMax >= 4) RBPF_C64(4, false); break;
#if RBPF_C64_LATE
            case 5: RBPF_C64(1, true); break;
            case 6: RBPF_C64(2, true); break;
            case 7: RBPF_C64(3, true); break;
            case 8: RBPF_C64(4, true); break;
#endif
            default: if (pass == 0) __syncthreads(); break;
          }
#undef RBPF_C64
        }
      }
      __syncthreads();                            // the block column is visible to the next panel products
      C64_STAMP(5);
#if !defined(RBPF_C64_DIAG_AFIXED) && !defined(RBPF_C64_DIAG_BFIXED)       // (the timing experiments run every block column)
      if (sfail[J & 1]) break;
#endif
    }
#ifdef RBPF_C64_STAMPS
    if (p == 0 && lane == 0 && M >= 200)
      printf("chol64 M=%d wave %d clocks: elems(w0:spin) %lld product %lld diagtile %lld waitA %lld solve/update %lld waitB %lld w0:prefetch %lld w0:stores+solves %lld\n", M, wv, cst[0], cst[1], cst[2], cst[3], cst[4], cst[5], cst[6], cst[7]);
#endif
    __syncthreads();
    const int failed = sfail[0] | sfail[1];
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
        for (int w = 0; w < W; ++w) { sl += red[w]; vv += red[16 + w]; }
        double lw;
        if (MODE == 0) lw = -sl - 0.5 * vv - 0.5 * (double)M * 1.8378770664093453;     // log(2*pi)
        else lw = -0.5 * a.qf[p] - a.hld[p] - sl + 0.5 * vv;
        a.pant_log[p] += lw;
      }
      goto next_particle;
    }
    if (MODE == 1 || attempt == 1) {
      if (tid == 0) { atomicOr(a.status, 2); a.pant_log[p] = nan(""); }
      goto next_particle;
    }
    jit = a.jitter;                                                         // particleSmoother.m:223
  }
  }
next_particle:
  if (nslots > 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();                                 // the slot and the LDS state are free for the workgroup's next particle
  }
}

constexpr size_t kC64MaxLds = 160 * 1024;         // the whole LDS of a CU
static size_t chol64_lds_bytes(int M, int d) {
  return ((size_t)5120 + 32 + M + 2 + (d ? 2 * (size_t)d * M : (size_t)M)) * sizeof(double);
}

template <int MODE, int W>
static hipError_t launch_chol64_mode(const CholArgs& ca, int batch, size_t lds, hipStream_t st) {
  static std::atomic<uint64_t> attr{0};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&chol_solve64_kernel<MODE, W>), (int)kC64MaxLds, attr)) return e;
  CholArgs cb = ca;
  cb.batch = batch;
  const int grid = (cb.l_slots > 0) ? std::min(batch, cb.l_slots) : batch;
  hipLaunchKernelGGL((chol_solve64_kernel<MODE, W>), dim3(grid), dim3(W * 64), lds, st, cb);
  return hipGetLastError();
}

// waves: 8 or 4 (0: by size)
static hipError_t launch_chol64(const CholArgs& ca, int batch, int d_lds, hipStream_t st, int waves = 0) {
  const size_t lds = chol64_lds_bytes(ca.Msz, d_lds);
  const int RT = (ca.Msz + 1 + 15) >> 4;
  if (waves == 0) {
    const char* we = tuning_env("RBPF_CHOL64_WAVES");                           // tuning: force 4 / 8
    waves = (we && (atoi(we) == 4 || atoi(we) == 8)) ? atoi(we) : (RT > 27) ? 8 : 4;
  }
  if (waves == 8) return ca.mode == 1 ? launch_chol64_mode<1, 8>(ca, batch, lds, st) : launch_chol64_mode<0, 8>(ca, batch, lds, st);
  return ca.mode == 1 ? launch_chol64_mode<1, 4>(ca, batch, lds, st) : launch_chol64_mode<0, 4>(ca, batch, lds, st);
}
