// The filter's step kernel on SYMMETRIC covariance storage (rbpf_options.storage = 2), gfx950.
//
// particleFilter.m:198 `P = P - K*SS*K'` keeps every P_i symmetric up to rounding, and the step kernel of rbpf_kernels.hip sits
// on the HBM roofline, so the only lever left on a particle-step is the number of bytes: here the stored matrix is the lower
// block triangle in 64 x 64 tiles (Layout::sym, 0.5625 n^2 elements at nLin = 515).  The same fused chain as step_kernel
// (resample-gather -> measModel -> importance weight -> Kalman update, particleFilter.m:100-204) with two differences:
//
// * Every stored off-diagonal tile T(I,J), I > J, contributes twice to P*H':  rows of block I get T*H_J' (lanes own rows, no
//   reduction, as before) and rows of block J get T'*H_I' -- a sum over the tile's ROWS, i.e. over lanes.  A wave owns the tile
//   rows {w, 7-w} (9 tiles each: balanced), accumulates per lane the products of its (up to two) rows for a pair of columns,
//   and reduces the six sums of a column pair across the wave in registers: v_permlane32_swap / v_permlane16_swap fold four
//   values into one register (two adds each halve the lanes), DPP row rotations finish the 16-lane rows.  Each wave keeps its
//   column contributions in its own LDS strip (fixed summation order -> bit-reproducible), combined after the stream by the
//   lane that owns the row.
// * Read-only steps of the multi-step lazy update do not touch the pending factor sets per element:
//   (P - sum_s KS_s K_s') H' = P H' - sum_s KS_s (K_s' H'), and K_s' H' is d x d -- an O(n d^2) epilogue instead of n^2 d
//   multiply-adds in the stream (same algebra, agreement to rounding as for lazy_depth itself).  A flush applies the sets
//   element-wise as before (it has to write P+), with the column factors of the current 64 columns in a wave-private LDS stage.
//
// * Read-only steps at eight tile rows use the QUAD mapping (sym_block_quad): a lane takes four rows of a tile and one column pair
//   of a quad of four, sums its 4 rows x 2 tile rows per column before anything crosses lanes, and the six column sums of a quad
//   are one 16-lane butterfly -- the cross-lane reductions were half of the kernel's instructions, and once the families of
//   particles that share a stored matrix sit on one XCD (xcd_position, rbpf_internal.hpp) the kernel is bound by its instruction
//   stream and by one memory round trip per round of loads, not by HBM.
// * At a flush step of the filter with two banks only one child per parent runs the flush variant; its siblings run the read-only
//   variant over the same source and point at the writer's new entry (shared flush: launch_share_plan, StepArgs::share_flush).
//
// * Sixteen tile rows (nLin = 1027: BASELINE.json configs[4]; r05): a wave owns the rows {w, 15 - w} (17 tiles each).  Its column strip
//   would be 64 (15 - w) columns -- 118 KB for the seven strips, more than the CU has next to H and P H' -- so a wave keeps the
//   column sums of ONE block column in a 1.5 KB LDS stage and copies them out to its strip in a global workspace (StepArgs::strip_ws,
//   118 KB per workgroup, three coalesced 512-byte stores per block column); the combine after the stream reads them back in the
//   same fixed order.  3 % more bytes than the stored matrix, all of them L2 hits in practice.
// * TS = float (rbpf_options.storage = 3): the tiles hold fp32 in column QUADS (Layout::sym = 2: a lane's 16-byte load is four columns
//   of its row -- with the pair order and 8-byte loads the read-only step moved 5.6 TB/s out of the L2s against the fp64 kernel's 9.2:
//   the vector memory path is bound by load instructions, not bytes), all arithmetic fp64, a flush rounds once on the way out -- as
//   storage = 1 does for the full square.  Filter only.
//
// Supported: dense families with ny = 3; fp64: 512 <= mc < 640 core rows (nLin = 515: BASELINE.json configs[2]) or four tile rows
// (nLin = 259), filter and both smoothers, single-GPU and sharded; sixteen tile rows (nLin = 1027) and fp32 tiles: the filter.
#include <type_traits>
#include "rbpf_internal.hpp"
#include "rbpf_device.hpp"
#include "rbpf_model_dev.hpp"

#ifdef RBPF_STAMPS                               // diagnostic builds: phase times of one workgroup (100 MHz ticks), printed by ctx_step
#define RBPF_SYM_KSTAMP(k) if (blockIdx.x == 4000 && threadIdx.x == 0 && a.stamps) a.stamps[k] = __builtin_amdgcn_s_memrealtime()
#else
#define RBPF_SYM_KSTAMP(k)
#endif
#ifndef RBPF_SYM_LIGHT_WGS
#define RBPF_SYM_LIGHT_WGS 2       // workgroups per CU the read-only filter variant is compiled for (3: xl stays in global memory)
#endif

namespace rbpf {

typedef double dbl2s __attribute__((ext_vector_type(2)));
typedef float flt2s __attribute__((ext_vector_type(2)));
typedef float flt4s __attribute__((ext_vector_type(4)));
template <typename TS> struct SymTile { static constexpr int cg = 2; };       // columns per 16-byte lane load: fp64 tiles hold column pairs,
template <> struct SymTile<float> { static constexpr int cg = 4; };           // fp32 tiles column quads (Layout::sym = 2, sym_t_index)

// a column pair of one stored row: TS = double (16 bytes per lane) or float (fp32 tiles, 8 bytes per lane; arithmetic stays fp64)
template <typename TS> __device__ __forceinline__ dbl2s ld_tile(const TS* p);
template <> __device__ __forceinline__ dbl2s ld_tile<double>(const double* p) { return *reinterpret_cast<const dbl2s*>(p); }
template <> __device__ __forceinline__ dbl2s ld_tile<float>(const float* p) {
  const flt2s v = *reinterpret_cast<const flt2s*>(p);
  dbl2s o; o.x = (double)v.x; o.y = (double)v.y;
  return o;
}
template <typename TS> __device__ __forceinline__ void st_tile(TS* p, dbl2s v);
template <> __device__ __forceinline__ void st_tile<double>(double* p, dbl2s v) { __builtin_nontemporal_store(v, reinterpret_cast<dbl2s*>(p)); }
template <> __device__ __forceinline__ void st_tile<float>(float* p, dbl2s v) {
  flt2s o; o.x = (float)v.x; o.y = (float)v.y;
  __builtin_nontemporal_store(o, reinterpret_cast<flt2s*>(p));
}

constexpr int kSymRows = 2;            // tile rows per wave: rows rp and CH - 1 - rp (CH + 1 tiles whatever rp: balanced)
// CH = 8 tile rows (nLin = 515): wave w owns the row pair rp = w and every column pair.  CH = 4 (nLin = 259): two waves share a row
// pair (rp = w & 1) and split its column pairs by parity (column phase cp = w >> 1): their row sums are added after the stream, their
// column sums go to disjoint columns of one strip.
constexpr int kSymRed = 64;            // doubles per wave of the block-reduction scratch (up to 7 * 3 * 3 = 63 values)
constexpr int kSymStage = 32;          // columns of pending column factors a wave keeps in LDS at a time (flush)

bool sym_supported(int n, int d) {
  const int mc = (n / kChunkRows) * kChunkRows;
  const int ch = mc / kSymChunk;
  if (d == 1) return ch == 2 && n == mc;             // dense-radio (n_y = 1, nLin = m = 128: two tile rows, no border rows)
  return d == 3 && (ch == 8 || ch == 4 || ch == 16);
}

// sixteen tile rows: doubles of the global column-strip workspace per workgroup (the seven strips of row pairs 1..7)
size_t sym_strip_doubles(const Layout& lay, int d) {
  if (!lay.sym || lay.CH64 != 16) return 0;
  size_t o = 0;
  for (int rp = 1; rp < lay.CH64 / 2; ++rp) o += (size_t)d * kSymChunk * (lay.CH64 - 1 - rp);
  return o;
}

Layout make_layout_sym(int n, int d, int fp32) {
  Layout L = make_layout(n, d);
  L.sym = fp32 ? 2 : 1;
  L.CH64 = L.mc / kSymChunk;
  L.szT = (size_t)L.CH64 * (L.CH64 + 1) / 2 * kSymTile;
  return L;
}

// LDS plan (doubles).  Column strips: row pair rp > 0 keeps its column contributions for core columns [0, 64 (CH - 1 - rp)); row
// pair 0's go straight into PHt.  off_row: the row sums of the column phases 1 .. NPH - 1 (CH = 4: two phases; CH = 2: four).
struct SymPlan { int off_H, off_xl, off_PHt, off_col1, off_row, off_tab, off_misc, off_red, off_kst, total; };
// strip of row pair rp (rp >= 1): D x ld_col(rp) doubles at off_col1 + D * 64 * sum_{v=1}^{rp-1} (CH - 1 - v)   (closed forms: no
// indexed arrays, which would live in scratch memory)
__host__ __device__ inline int sym_ld_col(int ch, int rp) { return kSymChunk * (ch - 1 - rp); }
__host__ __device__ inline int sym_off_col(int off_col1, int D, int ch, int rp) { return off_col1 + D * kSymChunk * ((rp - 1) * (ch - 1) - (rp - 1) * rp / 2); }

__host__ __device__ inline int sym_even(int x) { return (x + 1) & ~1; }
// waves of a workgroup = row pairs x column phases: four (CH = 8: four row pairs; CH = 4: two row pairs x two phases), eight at CH = 16
__host__ __device__ constexpr int sym_waves(int ch) { return ch == 16 ? 8 : kWaves; }
// column phases = waves that share a row pair and split its column pairs: CH / 2 row pairs x phases = waves
__host__ __device__ constexpr int sym_phases(int ch) { return ch == 4 ? 2 : (ch == 2 ? 4 : 1); }

__host__ __device__ inline SymPlan sym_plan(int n, int D, int ldx, int ktot, int nd_stage, int ch, int xl_lds = 1) {
  SymPlan p;
  const int waves = sym_waves(ch);
  int o = 0;
  p.off_H = o;   o += sym_even(n * D + 2);           // + pad so that the first core column pair is 16-byte aligned
  p.off_xl = o;  o += xl_lds ? ldx : 0;
  p.off_PHt = o; o += D * ldx;
  p.off_col1 = o;
  if (ch == 16) o += waves * D * kSymChunk;        // sixteen tile rows: one block column of column sums per wave (the strips are global)
  else for (int rp = 1; rp < ch / 2; ++rp) o += D * sym_ld_col(ch, rp);
  p.off_row = o;  o += (sym_phases(ch) - 1) * D * ch * kSymChunk;
  p.off_tab = o;  o += sym_even(2 * (ktot > 0 ? ktot : 1));
  p.off_misc = o; o += 64;
  p.off_red = o;  o += waves * kSymRed;
  p.off_kst = o;  o += waves * kSymStage * nd_stage;
  p.total = o;
  return p;
}

size_t step_sym_lds_bytes(const ModelDev& m, const Layout& lay, int n_sets, int write_base, int extra) {
  const int xl_lds = ((RBPF_SYM_LIGHT_WGS > 2 && !write_base && extra == 0) || lay.CH64 == 16) ? 0 : 1;
  return (size_t)sym_plan(lay.n, m.d + extra, lay.ldx, m.ktot, write_base ? n_sets * m.d : 0, lay.CH64, xl_lds).total * sizeof(double);
}

// ---- wave-level reduction primitives ---------------------------------------------------------------------------------------
// lanes 0..31: a[l] + a[l + 32]; lanes 32..63: b[l - 32] + b[l]   (v_permlane32_swap: upper half of the first operand <-> lower
// half of the second)
__device__ __forceinline__ double fold32(double a, double b) {
  const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
  const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
  const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
  return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
}
// rows of 16 lanes: [a.r0 + a.r1, b.r0 + b.r1, a.r2 + a.r3, b.r2 + b.r3]   (v_permlane16_swap: odd rows of the first operand <->
// even rows of the second)
__device__ __forceinline__ double fold16(double a, double b) {
  const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
  const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
  const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
  return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
}
#ifndef RBPF_SYM_DPP_NOINIT
#define RBPF_SYM_DPP_NOINIT 1
#endif
#ifndef RBPF_SYM_BFI
#define RBPF_SYM_BFI 1
#endif
// a row rotation reads a valid lane everywhere, so the destination needs no initial value (bound_ctrl: no v_mov_b32 dst, 0 before every v_mov_b32_dpp)
template <int CTRL>
__device__ __forceinline__ int dpp_rot32(int x) {
#if RBPF_SYM_DPP_NOINIT
  return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, true);
#else
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, false);
#endif
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  const int lo = dpp_rot32<CTRL>(__double2loint(x));
  const int hi = dpp_rot32<CTRL>(__double2hiint(x));
  return __hiloint2double(hi, lo);
}
// (a & m) | (b & ~m) in one instruction
__device__ __forceinline__ unsigned blend32(unsigned m, unsigned a, unsigned b) {
#if RBPF_SYM_BFI
  unsigned r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
#else
  return (a & m) | (b & ~m);
#endif
}
// one butterfly stage: the lane keeps u (sel = false) or v (sel = true), passes the other one on by the row rotation CTRL and adds what arrives
// (bit blends of the 32-bit halves, m = 0 or ~0: with selects on a lane predicate the register allocation of the whole kernel fell
// apart -- 359 spilled registers)
template <int CTRL>
__device__ __forceinline__ double bfly_pair(double u, double v, unsigned m) {
  const unsigned ulo = (unsigned)__double2loint(u), uhi = (unsigned)__double2hiint(u);
  const unsigned vlo = (unsigned)__double2loint(v), vhi = (unsigned)__double2hiint(v);
  const unsigned klo = blend32(m, vlo, ulo), khi = blend32(m, vhi, uhi);
  const unsigned slo = blend32(m, ulo, vlo), shi = blend32(m, uhi, vhi);
  const int rlo = dpp_rot32<CTRL>((int)slo);
  const int rhi = dpp_rot32<CTRL>((int)shi);
  return __hiloint2double((int)khi, (int)klo) + __hiloint2double(rhi, rlo);
}
// every lane of a 16-lane row gets the row's sum (rotations by 8, 4, 2, 1: a fixed order)
__device__ __forceinline__ double row16_sum(double x) {
  x += dpp_mov<0x128>(x);      // row_ror:8
  x += dpp_mov<0x124>(x);      // row_ror:4
  x += dpp_mov<0x122>(x);      // row_ror:2
  x += dpp_mov<0x121>(x);      // row_ror:1
  return x;
}
// Sums of four per-lane values over the 64 lanes.  Afterwards every lane of row rho (= lane / 16) holds the total of
// x[(rho & 1) * 2 + (rho >> 1)]:  rows 0..3 <-> x0, x2, x1, x3.
__device__ __forceinline__ double wave_sum4(double x0, double x1, double x2, double x3) {
  return row16_sum(fold16(fold32(x0, x1), fold32(x2, x3)));
}

__global__ void probe_wave_reduce_kernel(const double* __restrict__ in, double* __restrict__ out) {
  const int lane = threadIdx.x;
  const double r = wave_sum4(in[lane], in[64 + lane], in[128 + lane], in[192 + lane]);
  if ((lane & 15) == 0) { const int rho = lane >> 4; out[(rho & 1) * 2 + (rho >> 1)] = r; }
}

hipError_t launch_probe_wave_reduce(const double* in, double* out, hipStream_t s) {
  hipLaunchKernelGGL(probe_wave_reduce_kernel, dim3(1), dim3(64), 0, s, in, out);
  return hipGetLastError();
}

// ---- one block column (64 columns = 32 pairs) of a wave's tiles ---------------------------------------------------------------
// NACT active tile rows (the last NACT of the wave's rows); DIAG: the first active row's tile is the diagonal tile (row
// contribution only).  src / dst: the tiles' base + 2 * lane.  Hc: H of the block's first column ([col][D], 16-byte aligned),
// kst: the wave's stage of column factors [pair][ND][2] (flush only), colp: the wave's strip at the block's first column.
// Q0: index of the first active row in ks / hown / accr (default: the last NACT rows); ADD: the strip entries of these columns already
// hold the other row's contribution of this block column (split flush) -- add to them.
// NPH column phases: this wave takes the pairs pbeg + cp, pbeg + cp + NPH, ... of the stage.
template <typename TS, int D, int DE, int NS, bool WR, int NACT, bool DIAG, int Q0 = kSymRows - NACT, bool ADD = false, int KR = kSymRows, int NPH = 1>
__device__ __forceinline__ void sym_block(const TS* const (&src)[kSymRows], TS* const (&dst)[kSymRows],
                                          const double* __restrict__ Hc, const double* __restrict__ kst, int pbeg, int cp,
                                          const double (&ks)[KR][NS * D > 0 ? NS * D : 1], const double (&hown)[kSymRows][DE],
                                          double (&accr)[kSymRows][DE], double* __restrict__ colp, int ldc, int lane) {
  constexpr int ND = NS * D;
#ifndef RBPF_SYM_FLUSH_LOADS
#define RBPF_SYM_FLUSH_LOADS 4
#endif
#ifndef RBPF_SYM_LIGHT_LOADS
#define RBPF_SYM_LIGHT_LOADS 8
#endif
  // column pairs per round = wave-wide 1 KB loads in flight / active rows: 8 loads in the read-only steps, 4 in a flush (with 8 the
  // three- and four-set flushes kept 28-85 registers in scratch inside the block-column loop: 10 % more HBM writes, flush 29.1 ->
  // 26.1 ms at N = 65 536 with 4)
  constexpr int UP0 = (WR ? RBPF_SYM_FLUSH_LOADS : RBPF_SYM_LIGHT_LOADS) / NACT;   // (fp32 tiles: UP / 2 loads of a column quad)
  constexpr int UP = UP0 * NPH > kSymStage / 2 ? kSymStage / 2 / NPH : UP0;          // a round stays inside the stage of kSymStage / 2 pairs
  constexpr int PB = (WR || UP < 4) ? 2 : 4;            // pairs per compute / reduction batch (a flush carries 2 * ND factor values per pair)
  constexpr bool kCol = !(DIAG && NACT == 1);           // any off-diagonal tile in this block column?
  constexpr bool kF4 = std::is_same<TS, float>::value;     // fp32 tiles: one 16-byte load = two column pairs of a row (pairs 2 j, 2 j + 1)
  static_assert(!kF4 || (NPH == 1 && UP % 2 == 0 && PB == 2), "fp32 tiles: whole column quads per round");
  for (int p0 = pbeg + cp; p0 < pbeg + kSymStage / 2; p0 += UP * NPH) {   // the kSymStage columns whose pending factors are staged
    dbl2s v[UP][NACT];
    if constexpr (kF4) {
      flt4s raw[UP / 2][NACT];
#pragma unroll
      for (int j = 0; j < UP / 2; ++j)
#pragma unroll
        for (int q = 0; q < NACT; ++q) raw[j][q] = *reinterpret_cast<const flt4s*>(src[Q0 + q] + (size_t)((p0 >> 1) + j) * (4 * kSymChunk));
#pragma unroll
      for (int j = 0; j < UP / 2; ++j)
#pragma unroll
        for (int q = 0; q < NACT; ++q) {
          v[2 * j][q].x = (double)raw[j][q].x; v[2 * j][q].y = (double)raw[j][q].y;
          v[2 * j + 1][q].x = (double)raw[j][q].z; v[2 * j + 1][q].y = (double)raw[j][q].w;
        }
    } else {
#pragma unroll
    for (int u = 0; u < UP; ++u)
#pragma unroll
      for (int q = 0; q < NACT; ++q) v[u][q] = ld_tile<TS>(src[Q0 + q] + (size_t)(p0 + u * NPH) * (2 * kSymChunk));
    }
#pragma unroll
    for (int hh = 0; hh < UP / PB; ++hh) {
      double pc[PB][2][DE];
      dbl2s outv[kF4 && WR ? PB : 1][NACT];                  // fp32 tiles: the two pairs of a column quad leave in one 16-byte store
#pragma unroll
      for (int uu = 0; uu < PB; ++uu) {
        const int u = hh * PB + uu, p = p0 + u * NPH;
        double h0[DE], h1[DE];
        {
          // the pair's 2 * DE values of [H | ivec] sit contiguously
          double hb[2 * DE + 2];
#pragma unroll
          for (int k2 = 0; k2 < (2 * DE + 1) / 2; ++k2) {
            const dbl2s t = *reinterpret_cast<const dbl2s*>(Hc + (size_t)p * 2 * DE + 2 * k2);
            hb[2 * k2] = t.x; hb[2 * k2 + 1] = t.y;
          }
#pragma unroll
          for (int k = 0; k < DE; ++k) { h0[k] = hb[k]; h1[k] = hb[DE + k]; }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int k = 0; k < DE; ++k) pc[uu][e][k] = 0.0;
        // pending column factors of the pair, at most kChunk values (four sets) live at a time: with more sets (split flush,
        // NACT == 1) the next chunk is only read after the previous one was applied
        constexpr int kChunk = 4 * D, NCH = (ND + kChunk - 1) / kChunk;
        double kc0[kChunk], kc1[kChunk];
        // up to four sets: the column factors go through the registers ONE SET (D values per column) AT A TIME, the next set's LDS
        // reads in flight during the multiply-adds of the current one (all 2 * ND of them live across both rows' downdates cost 48
        // registers at four sets: 256 -> 215).  Measured: the same 16 ms per flush launch, with four or with eight wave loads per
        // round -- the flush is bound by its instruction stream (~500 cycles per column pair and wave: 72 multiply-adds, twelve
        // LDS reads, 1.5 wave_sum4), not by loads in flight
        double pq0[NACT], pq1[NACT];
#pragma unroll
        for (int q = 0; q < NACT; ++q) { pq0[q] = v[u][q].x; pq1[q] = v[u][q].y; }
        if (WR && NCH == 1 && ND > 0) {
          double ca[D], cb[D], na[D], nb_[D];
#pragma unroll
          for (int k = 0; k < D; ++k) {
            const dbl2s t = *reinterpret_cast<const dbl2s*>(kst + ((size_t)(p % (kSymStage / 2)) * ND + k) * 2);
            ca[k] = t.x; cb[k] = t.y;
          }
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) {
#pragma unroll
              for (int k = 0; k < D; ++k) {
                const dbl2s t = *reinterpret_cast<const dbl2s*>(kst + ((size_t)(p % (kSymStage / 2)) * ND + (s + 1) * D + k) * 2);
                na[k] = t.x; nb_[k] = t.y;
              }
            }
#pragma unroll
            for (int q = 0; q < NACT; ++q)
#pragma unroll
              for (int k = 0; k < D; ++k) {
                pq0[q] = fma(-ks[KR == 1 ? 0 : Q0 + q][s * D + k], ca[k], pq0[q]);
                pq1[q] = fma(-ks[KR == 1 ? 0 : Q0 + q][s * D + k], cb[k], pq1[q]);
              }
            if (s + 1 < NS) {
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int k = 0; k < D; ++k) { ca[k] = na[k]; cb[k] = nb_[k]; }
            }
          }
        }
#pragma unroll
        for (int q = 0; q < NACT; ++q) {
          double p0v = pq0[q], p1v = pq1[q];
          if (WR) {
            if (NCH == 1) {
            } else {
#pragma unroll
              for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
                for (int k = 0; k < kChunk; ++k) {
                  if (ch * kChunk + k < ND) {
                    const dbl2s t = *reinterpret_cast<const dbl2s*>(kst + ((size_t)(p % (kSymStage / 2)) * ND + ch * kChunk + k) * 2);
                    kc0[k] = t.x; kc1[k] = t.y;
                  }
                }
#pragma unroll
                for (int k = 0; k < kChunk; ++k)
                  if (ch * kChunk + k < ND) {
                    p0v = fma(-ks[KR == 1 ? 0 : Q0 + q][ch * kChunk + k], kc0[k], p0v); p1v = fma(-ks[KR == 1 ? 0 : Q0 + q][ch * kChunk + k], kc1[k], p1v);
                  }
                if (ch + 1 < NCH) __builtin_amdgcn_sched_barrier(0);
              }
            }
            dbl2s o; o.x = p0v; o.y = p1v;
            if constexpr (kF4) outv[uu][q] = o;
            else st_tile<TS>(dst[Q0 + q] + (size_t)p * (2 * kSymChunk), o);
          }
#pragma unroll
          for (int k = 0; k < DE; ++k) accr[Q0 + q][k] = fma(p1v, h1[k], fma(p0v, h0[k], accr[Q0 + q][k]));
          if (!(DIAG && q == 0)) {
#pragma unroll
            for (int k = 0; k < DE; ++k) { pc[uu][0][k] = fma(p0v, hown[Q0 + q][k], pc[uu][0][k]); pc[uu][1][k] = fma(p1v, hown[Q0 + q][k], pc[uu][1][k]); }
          }
        }
      }
      if constexpr (kF4 && WR) {
#pragma unroll
        for (int q = 0; q < NACT; ++q) {
          flt4s o4; o4.x = (float)outv[0][q].x; o4.y = (float)outv[0][q].y; o4.z = (float)outv[1][q].x; o4.w = (float)outv[1][q].y;
          __builtin_nontemporal_store(o4, reinterpret_cast<flt4s*>(dst[Q0 + q] + (size_t)((p0 + hh * PB) >> 1) * (4 * kSymChunk)));
        }
      }
      if (kCol) {
        // four values per register: the two columns of two pairs of one output k; row rho of the wave ends up with value
        // (rho & 1) * 2 + (rho >> 1), i.e. pair (rho & 1), column rho >> 1 of it
        const int rho = lane >> 4;
#pragma unroll
        for (int g = 0; g < PB / 2; ++g)
#pragma unroll
          for (int k = 0; k < DE; ++k) {
            const double r = wave_sum4(pc[2 * g][0][k], pc[2 * g][1][k], pc[2 * g + 1][0][k], pc[2 * g + 1][1][k]);
            if ((lane & 15) == 0) {
              double* cq = &colp[(size_t)k * ldc + 2 * (p0 + (hh * PB + 2 * g + (rho & 1)) * NPH) + (rho >> 1)];
              *cq = ADD ? *cq + r : r;
            }
          }
      }
    }
  }
}

// ---- read-only steps, CH = 8: the "quad" mapping ------------------------------------------------------------------------------
// sym_block gives a lane ONE row of a tile and both columns of a column pair, so every column sum of the transposed product is a
// 64-lane reduction -- 1.5 wave_sum4 per column pair, half of the read-only kernel's instructions.  Here a lane
// (r16 = lane & 15, g = lane >> 4) takes FOUR rows of a tile, r16 + 16 rq, of the column pair 4 t + g: one wave load still moves
// 1 KB (four pairs x 16 rows x 2 columns), but a lane now sums its 4 rows x 2 tile rows per column before anything crosses lanes,
// and a column sum is a 16-lane reduction (four DPP rotate-adds) shared by the four pairs of the quad: 72 instead of 150
// instructions per four column pairs, and 3 instead of 12 LDS reads of H.  The row sums of a row are then spread over the four
// lane groups (each saw a quarter of the columns); they are folded once after the stream (fold32 / fold16: no rotate-adds).
// src[q]: the tile's base + 128 g + 2 r16;  Hc: H of the block's first column;  colp: the wave's strip at the block's first column.
#ifndef RBPF_SYM_QUAD
#define RBPF_SYM_QUAD 1
#endif
#ifndef RBPF_SYM_QUAD_EMAX
#define RBPF_SYM_QUAD_EMAX 1                     // quad mapping for the filter (0) and the information-form step (1)
#endif
#ifndef RBPF_SYM_QUAD_LOADS
#define RBPF_SYM_QUAD_LOADS 16     // wave loads in flight per round (8: 218 registers; 16: 256 with two spilled, 3.5 % faster)
#endif
#ifndef RBPF_SYM_BUTTERFLY
#define RBPF_SYM_BUTTERFLY 1
#endif
template <typename TS, int D, int DE, int NACT, bool DIAG, int Q0>
__device__ __forceinline__ void sym_block_quad(const TS* const (&src)[kSymRows], const double* __restrict__ Hc,
                                               const double (&hown)[kSymRows][4][DE], double (&accr)[kSymRows][4][DE],
                                               double* __restrict__ colp, int ldc, int lane) {
  const int r16 = lane & 15, g = lane >> 4;
  constexpr bool kCol = !(DIAG && NACT == 1);           // any off-diagonal tile in this block column?
  // quads per round: RBPF_SYM_QUAD_LOADS wave loads in flight whatever the active rows (information form, DE = 4: eight -- its
  // four more row-sum and four more H registers per row do not leave room for sixteen)
  constexpr int TQ = ((DE > 3 ? 8 : RBPF_SYM_QUAD_LOADS) / 4) / NACT;
  // one quad of column pairs (4 t .. 4 t + 3; this lane: pair 4 t + g), vq: its rows r16 + 16 rq of the active tile rows
  auto quad = [&](const dbl2s (&vq)[NACT][4], const int t) {
      double h0[DE], h1[DE];
      {
        double hb[2 * DE + 2];                          // the pair's 2 * DE values of [H | ivec] sit contiguously
#pragma unroll
        for (int k2 = 0; k2 < (2 * DE + 1) / 2; ++k2) {
          const dbl2s tt = *reinterpret_cast<const dbl2s*>(Hc + (size_t)(4 * t + g) * 2 * DE + 2 * k2);
          hb[2 * k2] = tt.x; hb[2 * k2 + 1] = tt.y;
        }
#pragma unroll
        for (int k = 0; k < DE; ++k) { h0[k] = hb[k]; h1[k] = hb[DE + k]; }
      }
      double pc[2][DE];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int k = 0; k < DE; ++k) pc[e][k] = 0.0;
#pragma unroll
      for (int q = 0; q < NACT; ++q)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const double p0v = vq[q][rq].x, p1v = vq[q][rq].y;
#pragma unroll
          for (int k = 0; k < DE; ++k) accr[Q0 + q][rq][k] = fma(p1v, h1[k], fma(p0v, h0[k], accr[Q0 + q][rq][k]));
          if (!(DIAG && q == 0)) {
#pragma unroll
            for (int k = 0; k < DE; ++k) { pc[0][k] = fma(p0v, hown[Q0 + q][rq][k], pc[0][k]); pc[1][k] = fma(p1v, hown[Q0 + q][rq][k], pc[1][k]); }
          }
        }
      if (kCol) {
        if constexpr (DE == 3 && RBPF_SYM_BUTTERFLY) {
          // six 16-lane sums as a butterfly: rotations by 1, 2, 4, 8 (in this order a lane's selection bit of one stage is untouched by
          // the later rotations); a stage pairs two registers -- every lane keeps the one its bit selects, passes the other one on,
          // and adds what arrives -- so the register count halves while it can: 43 instead of 72 instructions.  Afterwards lane l
          // holds in q the sum (0,0) (0,1) (0,2) (1,0) for l & 3 = 0..3 and in w the sum (1,1) (1,2) for l & 1 = 0, 1.
          __builtin_amdgcn_sched_barrier(0);      // (left to itself the scheduler spreads these chains over the next round's loads: 359 spills)
          const unsigned b0 = 0u - (unsigned)(lane & 1), b1 = 0u - (unsigned)((lane >> 1) & 1);
          const double r0 = bfly_pair<0x121>(pc[0][0], pc[0][1], b0), r1 = bfly_pair<0x121>(pc[0][2], pc[1][0], b0);
          double w = bfly_pair<0x121>(pc[1][1], pc[1][2], b0);
          double q = bfly_pair<0x122>(r0, r1, b1);
          w += dpp_mov<0x122>(w);
          q += dpp_mov<0x124>(q); w += dpp_mov<0x124>(w);
          q += dpp_mov<0x128>(q); w += dpp_mov<0x128>(w);
          const int col = 2 * (4 * t + g);
          if (r16 < 4) colp[(size_t)(r16 == 3 ? 0 : r16) * ldc + col + (r16 == 3 ? 1 : 0)] = q;
          if (r16 < 2) colp[(size_t)(1 + r16) * ldc + col + 1] = w;
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < DE; ++k) {
              const double r = row16_sum(pc[e][k]);
              if (r16 == 0) colp[(size_t)k * ldc + 2 * (4 * t + g) + e] = r;
            }
        }
      }
  };
  // (a round = TQ quads loaded, then consumed.  r05 tried to keep loads in flight all the time -- half rounds double-buffered, or every
  //  quad's registers reloaded right after their use: both forms spilled 120 registers, 2.8 instead of 5.7 M particle-steps/s; the second
  //  form without a peeled last round and only where two tile rows are active: 11 spilled registers, 5.47 against 5.76 M.  The loop must
  //  not be unrolled either: the compiler then hoists the next rounds' loads and spills.)
#pragma unroll 1
  for (int t0 = 0; t0 < kSymChunk / 8; t0 += TQ) {      // quads of column pairs
    dbl2s v[TQ][NACT][4];
#pragma unroll
    for (int u = 0; u < TQ; ++u)
#pragma unroll
      for (int q = 0; q < NACT; ++q)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
          v[u][q][rq] = ld_tile<TS>(src[Q0 + q] + (size_t)(t0 + u) * (8 * kSymChunk) + rq * 32);
#pragma unroll
    for (int u = 0; u < TQ; ++u) quad(v[u], t0 + u);
  }
}

// The same mapping on fp32 tiles (column quads): the lane's 16-byte load brings the column quad 4 t + g of row r16 + 16 rq, i.e. the column
// pairs 2 (4 t + g) and 2 (4 t + g) + 1; each goes through the arithmetic of sym_block_quad.  src[q]: the tile's base + 256 g + 4 r16.
template <int D, int DE, int NACT, bool DIAG, int Q0>
__device__ __forceinline__ void sym_block_quad_f4(const float* const (&src)[kSymRows], const double* __restrict__ Hc,
                                                  const double (&hown)[kSymRows][4][DE], double (&accr)[kSymRows][4][DE],
                                                  double* __restrict__ colp, int ldc, int lane) {
  static_assert(DE == 3 && RBPF_SYM_BUTTERFLY, "fp32 tiles: the filter");
  const int r16 = lane & 15, g = lane >> 4;
  constexpr bool kCol = !(DIAG && NACT == 1);
#ifndef RBPF_SYM_QUAD_LOADS_F4
#define RBPF_SYM_QUAD_LOADS_F4 8   // wave loads in flight per round: each is worth two of the fp64 kernel's (sixteen: 145 spilled registers)
#endif
  constexpr int TQ = (RBPF_SYM_QUAD_LOADS_F4 / 4) / NACT;   // groups of four column quads per round
#pragma unroll 1
  for (int t0 = 0; t0 < kSymChunk / 16; t0 += TQ) {
    flt4s v[TQ][NACT][4];
#pragma unroll
    for (int u = 0; u < TQ; ++u)
#pragma unroll
      for (int q = 0; q < NACT; ++q)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
          v[u][q][rq] = *reinterpret_cast<const flt4s*>(src[Q0 + q] + (size_t)(t0 + u) * (16 * kSymChunk) + rq * 64);
#pragma unroll
    for (int u = 0; u < TQ; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pi = 2 * (4 * (t0 + u) + g) + h;            // column pair of the tile
        __builtin_amdgcn_sched_barrier(0);                    // one pair at a time (conversions and H reads of later pairs hoisted: 140 spilled registers)
        double h0[DE], h1[DE];
        {
          double hb[2 * DE + 2];
#pragma unroll
          for (int k2 = 0; k2 < (2 * DE + 1) / 2; ++k2) {
            const dbl2s tt = *reinterpret_cast<const dbl2s*>(Hc + (size_t)pi * 2 * DE + 2 * k2);
            hb[2 * k2] = tt.x; hb[2 * k2 + 1] = tt.y;
          }
#pragma unroll
          for (int k = 0; k < DE; ++k) { h0[k] = hb[k]; h1[k] = hb[DE + k]; }
        }
        double pc[2][DE];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int k = 0; k < DE; ++k) pc[e][k] = 0.0;
#pragma unroll
        for (int q = 0; q < NACT; ++q)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const double p0v = (double)(h ? v[u][q][rq].z : v[u][q][rq].x), p1v = (double)(h ? v[u][q][rq].w : v[u][q][rq].y);
#pragma unroll
            for (int k = 0; k < DE; ++k) accr[Q0 + q][rq][k] = fma(p1v, h1[k], fma(p0v, h0[k], accr[Q0 + q][rq][k]));
            if (!(DIAG && q == 0)) {
#pragma unroll
              for (int k = 0; k < DE; ++k) { pc[0][k] = fma(p0v, hown[Q0 + q][rq][k], pc[0][k]); pc[1][k] = fma(p1v, hown[Q0 + q][rq][k], pc[1][k]); }
            }
          }
        if (kCol) {
          __builtin_amdgcn_sched_barrier(0);
          const unsigned b0 = 0u - (unsigned)(lane & 1), b1 = 0u - (unsigned)((lane >> 1) & 1);
          const double r0 = bfly_pair<0x121>(pc[0][0], pc[0][1], b0), r1 = bfly_pair<0x121>(pc[0][2], pc[1][0], b0);
          double w = bfly_pair<0x121>(pc[1][1], pc[1][2], b0);
          double q = bfly_pair<0x122>(r0, r1, b1);
          w += dpp_mov<0x122>(w);
          q += dpp_mov<0x124>(q); w += dpp_mov<0x124>(w);
          q += dpp_mov<0x128>(q); w += dpp_mov<0x128>(w);
          const int col = 2 * pi;
          if (r16 < 4) colp[(size_t)(r16 == 3 ? 0 : r16) * ldc + col + (r16 == 3 ? 1 : 0)] = q;
          if (r16 < 2) colp[(size_t)(1 + r16) * ldc + col + 1] = w;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  }
}

// E = 1: information form (particleSmootherInformationForm.m:274-335): one more streamed right-hand side, P * ivec.  The
// reference also needs P * ivecPlus with ivecPlus = ivec + H' R^-1 y (:292) -- that is P * ivec + (P H') (R^-1 y), formed from
// the accumulated columns instead of streamed (same algebra; step_kernel<.., E = 2> streams both).
// (r04 also built the read-only filter step as a family GEMM on the matrix cores -- P_base * [H_1' ... H_f'] per family of particles
// sharing a stored matrix -- and measured it 0.4-0.8 ms per step SLOWER than this kernel: removed in r05, see commit 4711b84 and
// DESIGN_NOTEBOOK.md 10.)
template <typename TS, int D, int NS, bool WR, int E, int CH>
__global__ __launch_bounds__(64 * sym_waves(CH), CH == 16 ? 1 : ((!WR && E == 0) ? RBPF_SYM_LIGHT_WGS : 2)) void step_sym_kernel(const StepArgs a) {
  constexpr int NW = sym_waves(CH), NT = 64 * NW;              // CH = 16: eight waves, one workgroup per CU (the same eight waves per CU as 2 x 4)
  constexpr int NPH = sym_phases(CH);                          // column phases (waves per row pair)
  constexpr bool kGStrip = (CH == 16);                         // column strips in the global workspace, one block column staged in LDS
  extern __shared__ double smem[];
  constexpr int DE = D + E, ND = NS * D, NDA = ND > 0 ? ND : 1, NSA = NS > 0 ? NS : 1;
  // more than four pending sets in a flush: the wave's two tile rows go through every block column one after the other, so that
  // the row factors KS(r, .) of ONE row (NS * D registers) are live at a time
  constexpr bool kSplit = WR && NS > 4;
  const ModelDev& M = a.mdl;
  const Layout& Ly = a.lay;
  const int n = Ly.n, nb = Ly.nb, mc = Ly.mc, ldx = Ly.ldx, ldb = Ly.ldb;
  const int pos = xcd_position((int)blockIdx.x, (int)gridDim.x);   // processing position (sorted by the matrix the particle reads)
  const int* pre_i = a.pre_i + (size_t)pos * kPreInts;
  if (a.phase >= 0 && pre_i[5] != a.phase) return;      // single-bank flush: not this launch's share (workgroup-uniform)
  const int i = pre_i[0];
  const int dslot = WR ? pre_i[4] : i;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const SymPlan lp = sym_plan(n, DE, ldx, M.ktot, WR ? ND : 0, CH, ((RBPF_SYM_LIGHT_WGS > 2 && !WR && E == 0) || CH == 16) ? 0 : 1);
  double* Hs = smem + lp.off_H + ((nb * DE) & 1);             // [H | ivec] of column c at Hs[c * DE ..): core pairs 16-byte aligned
  double* xls = smem + lp.off_xl;
  double* PHt = smem + lp.off_PHt;                            // [DE][ldx]
  double* tabS = smem + lp.off_tab;
  double* tabC = tabS + (M.ktot > 0 ? M.ktot : 1);
  double* misc = smem + lp.off_misc;
  double* red = smem + lp.off_red;
#ifdef RBPF_SYM_DIAG_SAMEBASE                    // timing experiment only (wrong results): every particle streams one of 64 matrices
  const int ancb = pre_i[2], baseb = pre_i[3] & 63;
#else
  const int ancb = pre_i[2], baseb = pre_i[3];
#endif
  // sharded filter: a bank index >= n_bank_local refers to a received record [T | B | F | xl] (same layout as the banks)
  const bool remote = a.rec != nullptr && ancb >= a.n_bank_local;
  const double* recp = remote ? a.rec + (size_t)(ancb - a.n_bank_local) * a.rec_stride : nullptr;
  const bool remoteP = a.rec != nullptr && baseb >= a.n_bank_local;
  const double* recP = remoteP ? a.rec + (size_t)(baseb - a.n_bank_local) * a.rec_stride : nullptr;
  const TS* srcT = remoteP ? reinterpret_cast<const TS*>(recP) : reinterpret_cast<const TS*>(a.Pt_old) + (size_t)baseb * a.Pt_old_stride;
  const TS* srcB = remoteP ? reinterpret_cast<const TS*>(recP + a.rec_off_B) : reinterpret_cast<const TS*>(a.Pb_old) + (size_t)baseb * a.Pb_old_stride;
  const double* srcX = remote ? recp + a.rec_off_X : a.xl_old + (size_t)ancb * a.xl_old_stride;
  const double* Fs[NSA];
#pragma unroll
  for (int s = 0; s < NSA; ++s) Fs[s] = nullptr;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (a.fset[s]) Fs[s] = a.fset[s] + (size_t)pre_i[kPreSet0 + s] * 2 * D * ldx;
    else Fs[s] = remote ? recp + a.rec_off_F : a.F_old + (size_t)ancb * 2 * D * ldx;
  }

  RBPF_SYM_KSTAMP(0);
  // ---- A: propagated state (propagate_kernel ran first), prior mean ----
  if (tid < kPreDoubles) misc[tid] = a.pre_d[(size_t)pos * kPreDoubles + tid];
  constexpr bool kXlLds = !(RBPF_SYM_LIGHT_WGS > 2 && !WR && E == 0) && CH != 16;   // three workgroups per CU: no room for the prior mean in LDS
  if (kXlLds) for (int c = tid; c < n; c += NT) xls[c] = srcX[c];
  double Riy[D];                                               // R^-1 y (:292)
  if (E > 0) {
    const double* iv = remote ? recp + a.rec_off_I : a.ivec_old + (size_t)ancb * a.ivec_old_stride;
    for (int c = tid; c < n; c += NT) Hs[c * DE + D] = iv[c];
#pragma unroll
    for (int aa = 0; aa < D; ++aa) {
      double sacc = 0.0;
#pragma unroll
      for (int bb = 0; bb < D; ++bb) sacc = fma(M.Rinv[aa + D * bb], a.y[bb], sacc);
      Riy[aa] = sacc;
    }
  }
  __syncthreads();
  RBPF_SYM_KSTAMP(1);
  // ---- B: per-axis sin / cos tables ----
  for (int q = tid; q < M.ktot; q += NT) basis_table_entry(M, q, misc, tabS, tabC);
  __syncthreads();
  RBPF_SYM_KSTAMP(2);
  // ---- C: measurement Jacobian, one column per thread ----
  for (int c = tid; c < n; c += NT) {
    double h[D];
    if (a.H_ext != nullptr) {
#pragma unroll
      for (int k = 0; k < D; ++k) h[k] = a.H_ext[((size_t)i * D + k) * ldx + c];
    } else {
      H_column<D>(M, c, tabS, tabC, &misc[8], h);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) Hs[c * DE + k] = h[k];
    if (E > 0) {
      double sp = Hs[c * DE + D];                               // ivecPlus = ivec + dyi'/R*yt' (:292), the new information vector (:333)
#pragma unroll
      for (int k = 0; k < D; ++k) sp = fma(h[k], Riy[k], sp);
#pragma unroll
      for (int k = 0; k < D; ++k) a.Hb_new[((size_t)i * D + k) * ldx + c] = h[k];
      a.ivec_new[(size_t)i * ldx + c] = sp;
    }
  }
  __syncthreads();

  RBPF_SYM_KSTAMP(3);
  {
  // ---- D: stream the stored tiles once ----
  const int rp = (NPH == 1) ? wave : (wave % (CH / 2)), cp = (NPH == 1) ? 0 : (wave / (CH / 2));   // row pair, column phase
  const int rows[kSymRows] = {rp, CH - 1 - rp};                // ascending
  constexpr bool kQuad = RBPF_SYM_QUAD && !WR && NPH == 1 && E <= RBPF_SYM_QUAD_EMAX;   // read-only steps at CH = 8: sym_block_quad
  double accr[kSymRows][DE], hown[kSymRows][DE], ks[kSplit ? 1 : kSymRows][NDA];
  double accq[kQuad ? kSymRows : 1][4][DE], hq[kQuad ? kSymRows : 1][4][DE];
  if constexpr (kQuad) {
#pragma unroll
    for (int q = 0; q < kSymRows; ++q)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
#pragma unroll
        for (int k = 0; k < DE; ++k) { accq[q][rq][k] = 0.0; hq[q][rq][k] = Hs[(nb + rows[q] * kSymChunk + 16 * rq + (lane & 15)) * DE + k]; }
  }
#pragma unroll
  for (int q = 0; q < kSymRows; ++q) {
    const int r = nb + rows[q] * kSymChunk + lane;
#pragma unroll
    for (int k = 0; k < DE; ++k) { accr[q][k] = 0.0; hown[q][k] = Hs[r * DE + k]; }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int k = 0; k < D; ++k) ks[kSplit ? 0 : q][s * D + k] = (WR && !kSplit) ? Fs[s][(size_t)k * ldx + r] : 0.0;
    // the border COLUMNS of this row, P(r, b) = B(b, r), downdated like the border phase does when this is a flush.  Read here,
    // before anything is stored: in the second launch of a single-bank flush the border phase overwrites these very values.
    for (int b = 0; b < nb && cp == 0 && !kQuad; ++b) {        // (once per row: the first column phase; quad mapping: after the stream)
      double pv = (double)srcB[(size_t)b * ldb + r];
      if (WR) {
#pragma unroll
        for (int sset = 0; sset < NS; ++sset)
#pragma unroll
          for (int k = 0; k < D; ++k) pv = fma(-Fs[sset][(size_t)k * ldx + b], Fs[sset][(size_t)(D + k) * ldx + r], pv);
      }
#pragma unroll
      for (int k = 0; k < DE; ++k) accr[q][k] = fma(pv, Hs[b * DE + k], accr[q][k]);
    }
  }
  if (WR) __syncthreads();                                     // every wave has read the old border block before any wave stores it
  {
    TS* dT = reinterpret_cast<TS*>(a.Pt_new) + (size_t)dslot * Ly.szT;
    double* kst = smem + lp.off_kst + (size_t)wave * kSymStage * ND;
    double* colw = (rp == 0) ? PHt + nb : (kGStrip ? smem + lp.off_col1 + (size_t)wave * DE * kSymChunk : smem + sym_off_col(lp.off_col1, DE, CH, rp));
    const int ldc = (rp == 0) ? ldx : (kGStrip ? kSymChunk : sym_ld_col(CH, rp));
    // sixteen tile rows: this wave's strip in the global workspace
    double* gstrip = kGStrip ? a.strip_ws + (size_t)blockIdx.x * a.strip_ws_stride + sym_off_col(0, DE, CH, rp) : nullptr;
    const double* Hcore = Hs + (size_t)nb * DE;
    // column factors K(c, .) of kSymStage columns of every pending set -> the wave's LDS stage [pair][k][e] (lane = column;
    // wave-private: program order is the only synchronisation; the fetch latency is paid once per 32 columns and hidden by the
    // other seven waves of the CU -- a register prefetch would cost 2 * ND registers across the whole stream loop)
    double kpre[NDA];
    auto fetch = [&](int col0) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int k = 0; k < D; ++k) kpre[s * D + k] = Fs[s][(size_t)(D + k) * ldx + col0 + (lane & (kSymStage - 1))];
    };
    auto park = [&]() {
      // the stage is written as doubles and read as 16-byte pairs: the compiler barriers keep the two kinds of access in program order
      // (type-based alias analysis sees them as unrelated; seen to go wrong in the one-set flush at two tile rows)
      asm volatile("" ::: "memory");
      if (lane < kSymStage) {
#pragma unroll
        for (int k = 0; k < ND; ++k) kst[((size_t)(lane >> 1) * ND + k) * 2 + (lane & 1)] = kpre[k];
      }
      asm volatile("" ::: "memory");
    };
    const int last = rows[kSymRows - 1];
    for (int J = 0; J <= last; ++J) {
      const TS* src[kSymRows]; TS* dst[kSymRows];
#pragma unroll
      for (int q = 0; q < kSymRows; ++q) {
        const size_t off = ((size_t)rows[q] * (rows[q] + 1) / 2 + J) * kSymTile + SymTile<TS>::cg * lane;
        src[q] = srcT + off; dst[q] = dT + off;
      }
      const double* Hc = Hcore + (size_t)J * kSymChunk * DE;
      double* colp = (kGStrip && rp > 0) ? colw : colw + (size_t)J * kSymChunk;
      if constexpr (kQuad) {
        const TS* srq[kSymRows];
#pragma unroll
        for (int q = 0; q < kSymRows; ++q)
          srq[q] = srcT + ((size_t)rows[q] * (rows[q] + 1) / 2 + J) * kSymTile + SymTile<TS>::cg * (kSymChunk * (lane >> 4) + (lane & 15));
        if constexpr (std::is_same<TS, float>::value) {
          if (J < rows[0]) sym_block_quad_f4<D, DE, 2, false, 0>(srq, Hc, hq, accq, colp, ldc, lane);
          else if (J == rows[0]) sym_block_quad_f4<D, DE, 2, true, 0>(srq, Hc, hq, accq, colp, ldc, lane);
          else if (J < last) sym_block_quad_f4<D, DE, 1, false, 1>(srq, Hc, hq, accq, colp, ldc, lane);
          else sym_block_quad_f4<D, DE, 1, true, 1>(srq, Hc, hq, accq, colp, ldc, lane);
        } else {
        if (J < rows[0]) sym_block_quad<TS, D, DE, 2, false, 0>(srq, Hc, hq, accq, colp, ldc, lane);
        else if (J == rows[0]) sym_block_quad<TS, D, DE, 2, true, 0>(srq, Hc, hq, accq, colp, ldc, lane);
        else if (J < last) sym_block_quad<TS, D, DE, 1, false, 1>(srq, Hc, hq, accq, colp, ldc, lane);
        else sym_block_quad<TS, D, DE, 1, true, 1>(srq, Hc, hq, accq, colp, ldc, lane);
        }
      } else if constexpr (!kSplit) {
        for (int pbeg = 0; pbeg < kSymChunk / 2; pbeg += kSymStage / 2) {
          if (WR && ND > 0) { fetch(nb + J * kSymChunk + 2 * pbeg); park(); }
          if (J < rows[0]) sym_block<TS, D, DE, NS, WR, 2, false, 0, false, kSymRows, NPH>(src, dst, Hc, kst, pbeg, cp, ks, hown, accr, colp, ldc, lane);
          else if (J == rows[0]) sym_block<TS, D, DE, NS, WR, 2, true, 0, false, kSymRows, NPH>(src, dst, Hc, kst, pbeg, cp, ks, hown, accr, colp, ldc, lane);
          else if (J < last) sym_block<TS, D, DE, NS, WR, 1, false, 1, false, kSymRows, NPH>(src, dst, Hc, kst, pbeg, cp, ks, hown, accr, colp, ldc, lane);
          else sym_block<TS, D, DE, NS, WR, 1, true, 1, false, kSymRows, NPH>(src, dst, Hc, kst, pbeg, cp, ks, hown, accr, colp, ldc, lane);
        }
      } else {
        // row 1 (always active), then row 0 where it reaches this block column; row 0's column sums are added to row 1's
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int k = 0; k < D; ++k) ks[0][s * D + k] = Fs[s][(size_t)k * ldx + nb + rows[1] * kSymChunk + lane];
        for (int pbeg = 0; pbeg < kSymChunk / 2; pbeg += kSymStage / 2) {
          fetch(nb + J * kSymChunk + 2 * pbeg); park();
          if (J < last) sym_block<TS, D, DE, NS, WR, 1, false, 1, false, 1, NPH>(src, dst, Hc, kst, pbeg, cp, ks, hown, accr, colp, ldc, lane);
          else sym_block<TS, D, DE, NS, WR, 1, true, 1, false, 1, NPH>(src, dst, Hc, kst, pbeg, cp, ks, hown, accr, colp, ldc, lane);
        }
        if (J <= rows[0]) {
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int k = 0; k < D; ++k) ks[0][s * D + k] = Fs[s][(size_t)k * ldx + nb + rows[0] * kSymChunk + lane];
          for (int pbeg = 0; pbeg < kSymChunk / 2; pbeg += kSymStage / 2) {
            fetch(nb + J * kSymChunk + 2 * pbeg); park();
            if (J < rows[0]) sym_block<TS, D, DE, NS, WR, 1, false, 0, true, 1, NPH>(src, dst, Hc, kst, pbeg, cp, ks, hown, accr, colp, ldc, lane);
            else sym_block<TS, D, DE, NS, WR, 1, true, 0, true, 1, NPH>(src, dst, Hc, kst, pbeg, cp, ks, hown, accr, colp, ldc, lane);
          }
        }
      }
      if (kGStrip && rp > 0 && J < last) {
        // the block column's sums leave the stage: rows k of the strip, 64 consecutive columns each (the stage is wave-private: program
        // order is the only synchronisation, as for kst)
#pragma unroll
        for (int k = 0; k < DE; ++k) gstrip[(size_t)k * sym_ld_col(CH, rp) + J * kSymChunk + lane] = colw[k * kSymChunk + lane];
      }
    }
  }
  // border rows (row-major block B, all n columns): lanes walk columns, wave-reduce per row (as in step_kernel)
  for (int b = wave; b < nb; b += NW) {
    const TS* src = srcB + (size_t)b * ldb;
    TS* dstb = reinterpret_cast<TS*>(a.Pb_new) + (size_t)dslot * Ly.szB + (size_t)b * ldb;
    double ksb[NDA];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int k = 0; k < D; ++k) ksb[s * D + k] = WR ? Fs[s][(size_t)k * ldx + b] : 0.0;
    double accb[DE];
#pragma unroll
    for (int k = 0; k < DE; ++k) accb[k] = 0.0;
    for (int c = 2 * lane; c < ldb; c += 128) {
      const dbl2s vv = ld_tile<TS>(src + c);
      double p[2] = {vv.x, vv.y};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int cc = c + e;
        if (cc < n) {
          if (WR) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
              for (int k = 0; k < D; ++k) p[e] = fma(-ksb[s * D + k], Fs[s][(size_t)(D + k) * ldx + cc], p[e]);
          }
#pragma unroll
          for (int k = 0; k < DE; ++k) accb[k] = fma(p[e], Hs[cc * DE + k], accb[k]);
        }
      }
      if (WR) { dbl2s o; o.x = p[0]; o.y = p[1]; st_tile<TS>(dstb + c, o); }
    }
#pragma unroll
    for (int k = 0; k < DE; ++k) {
      const double s = wave_sum(accb[k]);
      if (lane == 0) PHt[(size_t)k * ldx + b] = s;
    }
  }
  if (NPH > 1 && cp > 0) {                                     // the other column phases: their row sums go through LDS
    double* rowp = smem + lp.off_row + (size_t)(cp - 1) * DE * CH * kSymChunk;
#pragma unroll
    for (int q = 0; q < kSymRows; ++q)
#pragma unroll
      for (int k = 0; k < DE; ++k) rowp[(size_t)k * (CH * kSymChunk) + rows[q] * kSymChunk + lane] = accr[q][k];
  }
  __syncthreads();
  // combine, by the lane (of the first column phase) that owns the row: row part incl. the border columns (registers) + the other
  // phase's row part + the row pairs' column parts in order
  if (cp == 0) {
#pragma unroll
    for (int q = 0; q < kSymRows; ++q) {
      // quad mapping: the four lane groups hold the row sums of rows r16 + 16 rq over their quarter of the columns; two folds add the
      // groups and leave row 16 * {0, 2, 1, 3}[lane >> 4] + r16 in this lane (wave_sum4's order), plus the border columns' part,
      // which the plain mapping accumulated for row `lane`
      const int rho = lane >> 4;
      const int rc = rows[q] * kSymChunk + (kQuad ? 16 * ((rho & 1) * 2 + (rho >> 1)) + (lane & 15) : lane);   // core coordinate
      double s[DE];
#pragma unroll
      for (int k = 0; k < DE; ++k) s[k] = accr[q][k];
      if constexpr (kQuad) {
        // the border columns of this lane's row, P(r, b) = B(b, r) (not live across the stream: twelve registers for its loads)
#pragma unroll
        for (int k = 0; k < DE; ++k) s[k] = 0.0;
        for (int b = 0; b < nb; ++b) {
          const double pv = (double)srcB[(size_t)b * ldb + nb + rc];
#pragma unroll
          for (int k = 0; k < DE; ++k) s[k] = fma(pv, Hs[b * DE + k], s[k]);
        }
#pragma unroll
        for (int k = 0; k < DE; ++k)
          s[k] = fold16(fold32(accq[q][0][k], accq[q][1][k]), fold32(accq[q][2][k], accq[q][3][k])) + s[k];
      }
      if (NPH > 1) {
#pragma unroll
        for (int ph = 1; ph < NPH; ++ph) {                     // phases in order: a fixed summation order
          const double* rowp = smem + lp.off_row + (size_t)(ph - 1) * DE * CH * kSymChunk;
#pragma unroll
          for (int k = 0; k < DE; ++k) s[k] += rowp[(size_t)k * (CH * kSymChunk) + rc];
        }
      }
#pragma unroll
      for (int w = 0; w < CH / 2; ++w) {
        if (rows[q] < CH - 1 - w) {                           // row pair w holds off-diagonal tiles in this block column
          const double* cw = (w == 0) ? PHt + nb : (kGStrip ? a.strip_ws + (size_t)blockIdx.x * a.strip_ws_stride + sym_off_col(0, DE, CH, w)
                                                            : smem + sym_off_col(lp.off_col1, DE, CH, w));
          const int ldw = (w == 0) ? ldx : sym_ld_col(CH, w);
#pragma unroll
          for (int k = 0; k < DE; ++k) s[k] += cw[(size_t)k * ldw + rc];
        }
      }
#pragma unroll
      for (int k = 0; k < DE; ++k) PHt[(size_t)k * ldx + nb + rc] = s[k];
    }
  }
  __syncthreads();
  }
  if (!WR && NS > 0) {
    // read-only step: PHt holds P_base * H'; subtract sum_s KS_s * (K_s' * H')
    constexpr int NG = NS * D * DE > 0 ? NS * D * DE : 1;
    double g[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) g[q] = 0.0;
    for (int c = tid; c < n; c += NT) {
      double h[DE];
#pragma unroll
      for (int k = 0; k < DE; ++k) h[k] = Hs[c * DE + k];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const double kv = Fs[s][(size_t)(D + k) * ldx + c];
#pragma unroll
          for (int j = 0; j < DE; ++j) g[(s * D + k) * DE + j] = fma(kv, h[j], g[(s * D + k) * DE + j]);
        }
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const double s = wave_sum(g[q]);
      if (lane == 0) red[wave * kSymRed + q] = s;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      double s = red[q];
      for (int w = 1; w < NW; ++w) s += red[w * kSymRed + q];
      g[q] = s;
    }
    for (int r = tid; r < n; r += NT) {
      double ph[DE];
#pragma unroll
      for (int j = 0; j < DE; ++j) ph[j] = PHt[(size_t)j * ldx + r];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const double ksv = Fs[s][(size_t)k * ldx + r];
#pragma unroll
          for (int j = 0; j < DE; ++j) ph[j] = fma(-ksv, g[(s * D + k) * DE + j], ph[j]);
        }
#pragma unroll
      for (int j = 0; j < DE; ++j) PHt[(size_t)j * ldx + r] = ph[j];
    }
    __syncthreads();
  }

  RBPF_SYM_KSTAMP(4);
  // ---- E: S = H (P H') + R, e = y - H xl   (particleFilter.m:139-150) ----
  constexpr int NRED = D * D + D + 2 * E;
  {
    double part[NRED];
#pragma unroll
    for (int q = 0; q < NRED; ++q) part[q] = 0.0;
    for (int r = tid; r < n; r += NT) {
      double h[D], ph[D];
#pragma unroll
      for (int k = 0; k < D; ++k) { h[k] = Hs[r * DE + k]; ph[k] = PHt[(size_t)k * ldx + r]; }
      const double x = kXlLds ? xls[r] : srcX[r];
#pragma unroll
      for (int bb = 0; bb < D; ++bb)
#pragma unroll
        for (int aa = 0; aa < D; ++aa) part[aa + D * bb] = fma(h[aa], ph[bb], part[aa + D * bb]);
#pragma unroll
      for (int aa = 0; aa < D; ++aa) part[D * D + aa] = fma(h[aa], x, part[D * D + aa]);
      if (E > 0) {
        // ivec' P ivec and ivecPlus' P ivecPlus (:301-303), P = the (downdated) prior covariance
        const double iv = Hs[r * DE + D], piv = PHt[(size_t)D * ldx + r];
        double ivp = iv, pivp = piv;
#pragma unroll
        for (int k = 0; k < D; ++k) { ivp = fma(h[k], Riy[k], ivp); pivp = fma(ph[k], Riy[k], pivp); }
        part[D * D + D] = fma(iv, piv, part[D * D + D]);
        part[D * D + D + 1] = fma(ivp, pivp, part[D * D + D + 1]);
      }
    }
#pragma unroll
    for (int q = 0; q < NRED; ++q) {
      const double s = wave_sum(part[q]);
      if (lane == 0) red[wave * kSymRed + q] = s;
    }
  }
  __syncthreads();
  if (tid == 0) {
    double SS[D * D], e[D], cS[D * D], v[D];
    for (int q = 0; q < D * D; ++q) {
      double s = red[q];
      for (int w = 1; w < NW; ++w) s += red[w * kSymRed + q];
      SS[q] = s + M.R[q];                                                   // particleFilter.m:141
    }
    for (int q = 0; q < D; ++q) {
      double s = red[D * D + q];
      for (int w = 1; w < NW; ++w) s += red[w * kSymRed + D * D + q];
      e[q] = a.y[q] - s;                                                    // :140
    }
    bool ok = chol_lower_small<D>(SS, cS);                                  // :145
    if (!ok) {
      double SJ[D * D];
      for (int q = 0; q < D * D; ++q) SJ[q] = SS[q];
      for (int q = 0; q < D; ++q) SJ[q + D * q] += M.jitter;                // :147
      ok = chol_lower_small<D>(SJ, cS);
    }
    double lw = 0.0;
    if (ok) {
      fwd_subst<D>(cS, e, v);                                               // :149
      double vv = 0.0, sl = 0.0;
      for (int q = 0; q < D; ++q) { sl += log(cS[q + D * q]); vv += v[q] * v[q]; }
      lw = -sl - 0.5 * vv + M.logconst;                                     // :150
    } else {
      atomicOr(a.status, 1);
      lw = nan("");
      for (int q = 0; q < D * D; ++q) cS[q] = 0.0;
      for (int q = 0; q < D; ++q) cS[q + D * q] = 1.0;
    }
    if (E == 0) a.logw[i] = lw;
    for (int q = 0; q < D * D; ++q) { misc[20 + q] = cS[q]; misc[30 + q] = SS[q]; }
    for (int q = 0; q < D; ++q) misc[40 + q] = e[q];
    if (E > 0) {
      double qa = red[D * D + D], qb = red[D * D + D + 1];
      for (int w = 1; w < NW; ++w) { qa += red[w * kSymRed + D * D + D]; qb += red[w * kSymRed + D * D + D + 1]; }
      double sl2 = 0.0;
      for (int q = 0; q < D; ++q) sl2 += log(cS[q + D * q]);
      misc[44] = qa; misc[45] = qb; misc[46] = ok ? sl2 : nan("");
    }
  }
  __syncthreads();

  RBPF_SYM_KSTAMP(5);
  // ---- F: Kalman gain rows, mean update, new pending factors (particleFilter.m:194-198) ----
  {
    double cS[D * D], SS[D * D], e[D];
#pragma unroll
    for (int q = 0; q < D * D; ++q) { cS[q] = misc[20 + q]; SS[q] = misc[30 + q]; }
#pragma unroll
    for (int q = 0; q < D; ++q) e[q] = misc[40 + q];
    double* KSn = a.F_new + ((size_t)i * 2 + 0) * D * ldx;
    double* Kn = a.F_new + ((size_t)i * 2 + 1) * D * ldx;
    if (tid == 0) {
      if (a.base_new) a.base_new[i] = WR ? dslot : (a.share_flush ? pre_i[4] : baseb);
      if (!WR) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
          if (a.fset_idx_new[s]) a.fset_idx_new[s][i] = pre_i[kPreSet0 + s];
      }
      if (a.fself_idx_new) a.fself_idx_new[i] = i;
    }
    double* xln = a.xl_new + (size_t)i * ldx;
    double uK[D];
#pragma unroll
    for (int k = 0; k < D; ++k) uK[k] = 0.0;
    for (int r = tid; r < n; r += NT) {
      double ph[D], u[D], kk[D];
#pragma unroll
      for (int k = 0; k < D; ++k) ph[k] = PHt[(size_t)k * ldx + r];
      fwd_subst<D>(cS, ph, u);
      bwd_subst_T<D>(cS, u, kk);
      double xn_ = kXlLds ? xls[r] : srcX[r];
#pragma unroll
      for (int k = 0; k < D; ++k) xn_ = fma(kk[k], e[k], xn_);              // :197
      xln[r] = xn_;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) s = fma(kk[k], SS[k + D * j], s);
        KSn[(size_t)j * ldx + r] = s;
        Kn[(size_t)j * ldx + r] = kk[j];
      }
      if (E > 0) {
        double ivp = Hs[r * DE + D];
#pragma unroll
        for (int k = 0; k < D; ++k) ivp = fma(Hs[r * DE + k], Riy[k], ivp);
#pragma unroll
        for (int k = 0; k < D; ++k) uK[k] = fma(ivp, kk[k], uK[k]);            // ivecPlus' * K
      }
    }
    if (E > 0) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double s = wave_sum(uK[k]);
        if (lane == 0) red[wave * kSymRed + k] = s;
      }
      __syncthreads();
      if (tid == 0) {
        double u[D];
        for (int k = 0; k < D; ++k) { double s = red[k]; for (int w = 1; w < NW; ++w) s += red[w * kSymRed + k]; u[k] = s; }
        double corr = 0.0;                                                  // ivecPlus' * (K*SS*K') * ivecPlus
        for (int bb = 0; bb < D; ++bb) {
          double t = 0.0;
          for (int aa = 0; aa < D; ++aa) t = fma(u[aa], SS[aa + D * bb], t);
          corr = fma(t, u[bb], corr);
        }
        const double qa = misc[44], qbp = misc[45] - corr, sl = misc[46];
        const double hld = remote ? recp[a.rec_off_hld] : a.hld_old[(size_t)ancb * a.hld_old_stride];
        const double hldp = -sl + M.halfLogDetR + hld;                       // :298
        double yRy = 0.0;
        for (int bb = 0; bb < D; ++bb) {
          double t = 0.0;
          for (int aa = 0; aa < D; ++aa) t = fma(a.y[aa], M.Rinv[aa + D * bb], t);
          yRy = fma(t, a.y[bb], yRy);
        }
        // :301-304   (1/2*log((2*pi)^ny*det(R)) = -logconst + halfLogDetR)
        a.logw[i] = -0.5 * qa - hld + hldp + 0.5 * qbp - 0.5 * yRy - (-M.logconst + M.halfLogDetR);
        a.hld_new[i] = hldp;
        a.qf_new[i] = qbp;
      }
    }
  }
  RBPF_SYM_KSTAMP(6);
}

template <typename TS, int D, int NS, bool WR, int E, int CH>
static hipError_t launch_sym_kc(const StepArgs& a, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&step_sym_kernel<TS, D, NS, WR, E, CH>), 160 * 1024, attr_done)) return e;
  const size_t lds = step_sym_lds_bytes(a.mdl, a.lay, NS, WR ? 1 : 0, E);
  if (CH == 16 && (!a.strip_ws || lds > 160 * 1024)) return hipErrorInvalidValue;
  hipLaunchKernelGGL((step_sym_kernel<TS, D, NS, WR, E, CH>), dim3(a.N), dim3(64 * sym_waves(CH)), lds, s, a);
  return hipGetLastError();
}

template <typename TS, int D, int NS, bool WR, int E>
static hipError_t launch_sym_k(const StepArgs& a, hipStream_t s) {
  if constexpr (std::is_same<TS, double>::value) {
    if (a.lay.CH64 == 8) return launch_sym_kc<TS, D, NS, WR, E, 8>(a, s);
    if constexpr (NS <= 4) { if (a.lay.CH64 == 4) return launch_sym_kc<TS, D, NS, WR, E, 4>(a, s); }      // four tile rows: lazy_depth <= 4
  }
  // sixteen tile rows (fp64 and fp32 tiles) and fp32 tiles at eight: the filter, lazy_depth <= 4
  if constexpr (NS <= 4 && E == 0) {
    if (a.lay.CH64 == 16) return launch_sym_kc<TS, D, NS, WR, E, 16>(a, s);
    if constexpr (std::is_same<TS, float>::value) { if (a.lay.CH64 == 8) return launch_sym_kc<TS, D, NS, WR, E, 8>(a, s); }
  }
  return hipErrorInvalidValue;
}

template <typename TS>
static hipError_t launch_step_sym_filter(const StepArgs& a, hipStream_t s) {
  if (a.write_base) {
    switch (a.n_sets) {
      case 0: return launch_sym_k<TS, 3, 0, true, 0>(a, s);
      case 1: return launch_sym_k<TS, 3, 1, true, 0>(a, s);
      case 2: return launch_sym_k<TS, 3, 2, true, 0>(a, s);
      case 3: return launch_sym_k<TS, 3, 3, true, 0>(a, s);
      case 4: return launch_sym_k<TS, 3, 4, true, 0>(a, s);
      case 5: return launch_sym_k<TS, 3, 5, true, 0>(a, s);
      case 6: return launch_sym_k<TS, 3, 6, true, 0>(a, s);
      case 7: return launch_sym_k<TS, 3, 7, true, 0>(a, s);
      case 8: return launch_sym_k<TS, 3, 8, true, 0>(a, s);
      default: return hipErrorInvalidValue;
    }
  }
  switch (a.n_sets) {
    case 1: return launch_sym_k<TS, 3, 1, false, 0>(a, s);
    case 2: return launch_sym_k<TS, 3, 2, false, 0>(a, s);
    case 3: return launch_sym_k<TS, 3, 3, false, 0>(a, s);
    case 4: return launch_sym_k<TS, 3, 4, false, 0>(a, s);
    case 5: return launch_sym_k<TS, 3, 5, false, 0>(a, s);
    case 6: return launch_sym_k<TS, 3, 6, false, 0>(a, s);
    case 7: return launch_sym_k<TS, 3, 7, false, 0>(a, s);
    default: return hipErrorInvalidValue;
  }
}

// dense-radio (n_y = 1, nLin = 128): two tile rows, four waves share the one row pair; filter and information form, lazy_depth <= 3
template <int NS, bool WR, int E>
static hipError_t launch_sym_radio(const StepArgs& a, hipStream_t s) { return launch_sym_kc<double, 1, NS, WR, E, 2>(a, s); }

static hipError_t launch_step_sym_radio(const StepArgs& a, hipStream_t s) {
  if (a.lay.CH64 != 2 || a.fp32 || a.lay.nb != 0) return hipErrorInvalidValue;
  if (a.info) {
    if (a.write_base) {
      switch (a.n_sets) {
        case 0: return launch_sym_radio<0, true, 1>(a, s);
        case 1: return launch_sym_radio<1, true, 1>(a, s);
        case 2: return launch_sym_radio<2, true, 1>(a, s);
        case 3: return launch_sym_radio<3, true, 1>(a, s);
        default: return hipErrorInvalidValue;
      }
    }
    switch (a.n_sets) {
      case 1: return launch_sym_radio<1, false, 1>(a, s);
      case 2: return launch_sym_radio<2, false, 1>(a, s);
      default: return hipErrorInvalidValue;
    }
  }
  if (a.write_base) {
    switch (a.n_sets) {
      case 0: return launch_sym_radio<0, true, 0>(a, s);
      case 1: return launch_sym_radio<1, true, 0>(a, s);
      case 2: return launch_sym_radio<2, true, 0>(a, s);
      case 3: return launch_sym_radio<3, true, 0>(a, s);
      case 4: return launch_sym_radio<4, true, 0>(a, s);
      default: return hipErrorInvalidValue;
    }
  }
  switch (a.n_sets) {
    case 1: return launch_sym_radio<1, false, 0>(a, s);
    case 2: return launch_sym_radio<2, false, 0>(a, s);
    case 3: return launch_sym_radio<3, false, 0>(a, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_step_sym(const StepArgs& a, hipStream_t s) {
  if (a.lay.sym && a.mdl.d == 1) return launch_step_sym_radio(a, s);
  if (!a.lay.sym || a.mdl.d != 3 || (a.lay.CH64 != 8 && a.lay.CH64 != 4 && a.lay.CH64 != 16)) return hipErrorInvalidValue;
  if (a.fp32) return a.info ? hipErrorInvalidValue : launch_step_sym_filter<float>(a, s);
  if (a.info) {                                            // information form: lazy_depth <= 3
    if (a.write_base) {
      switch (a.n_sets) {
        case 0: return launch_sym_k<double, 3, 0, true, 1>(a, s);
        case 1: return launch_sym_k<double, 3, 1, true, 1>(a, s);
        case 2: return launch_sym_k<double, 3, 2, true, 1>(a, s);
        case 3: return launch_sym_k<double, 3, 3, true, 1>(a, s);
        default: return hipErrorInvalidValue;
      }
    }
    switch (a.n_sets) {
      case 1: return launch_sym_k<double, 3, 1, false, 1>(a, s);
      case 2: return launch_sym_k<double, 3, 2, false, 1>(a, s);
      case 3: return launch_sym_k<double, 3, 3, false, 1>(a, s);     // the readers of a shared flush at lazy_depth 3
      default: return hipErrorInvalidValue;
    }
  }
  return launch_step_sym_filter<double>(a, s);
}

}  // namespace rbpf
