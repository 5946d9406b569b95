// HIP kernels of the Rao-Blackwellized particle filter hot path, written for gfx950 (MI355X).
//
// One particle-step of the reference (src/particleFilter.m:100-204) is
//   resample-gather (:112-113)  ->  dynModel (:108)  ->  measModel (:124)
//   -> importance weight (:139-150)  ->  Kalman map update (:184-198)
// and touches the particle's n x n covariance three times.  Here the whole chain is ONE kernel that
// reads the ancestor's covariance exactly once and writes the child's exactly once:
//
//   The Kalman downdate  P+ = P - (K*SS)*K'  (particleFilter.m:198) is rank-d, so it is carried as a
//   *pending* pair of n x d factors (KS = K*SS, K) next to the stored matrix.  The next step's kernel
//   applies it element-wise while the matrix streams through the CU, accumulates P+*H' for the new
//   H in the same pass (lanes own rows, so no cross-lane reduction), stores P+ into the child's slot,
//   and only then (epilogue, O(n d^2)) forms S, chol(S), logw, K, K*SS and the mean update.
//
// HBM traffic per particle-step = n^2 read + n^2 write (+ O(n d)): the algorithmic minimum of
// SURVEY 8(d).  The kernel is HBM-bound (12 flop per 16 B).
#include "rbpf_internal.hpp"
#include "rbpf_device.hpp"
#include "rbpf_model_dev.hpp"

#include <cstdlib>

#ifndef RBPF_UC
#define RBPF_UC 8          // columns kept in flight per wave in the covariance stream
#endif
#ifndef RBPF_UC2
#define RBPF_UC2 4         // ... when a wave owns two row chunks per column
#endif
#ifndef RBPF_UC3
#define RBPF_UC3 2         // ... three row chunks
#endif
#ifndef RBPF_NT_LOAD
#define RBPF_NT_LOAD 0     // 1: nontemporal loads (defeats the Infinity-Cache reuse of sibling reads: keep 0)
#endif
#ifndef RBPF_NT_STORE
#define RBPF_NT_STORE 1    // 1: nontemporal stores of the streamed covariance
#endif

namespace rbpf {

typedef double dbl2 __attribute__((ext_vector_type(2)));

// Two consecutive rows of a stored covariance.  TS = double (the reference's precision) or float (fp32 STORAGE of the
// covariance banks, rbpf_options.storage = 1: half the HBM traffic; all arithmetic stays fp64).
typedef float flt2 __attribute__((ext_vector_type(2)));

template <typename TS> __device__ __forceinline__ dbl2 ld_pair(const TS* p);
template <> __device__ __forceinline__ dbl2 ld_pair<double>(const double* p) { return *reinterpret_cast<const dbl2*>(p); }
template <> __device__ __forceinline__ dbl2 ld_pair<float>(const float* p) {
  const flt2 v = *reinterpret_cast<const flt2*>(p);
  dbl2 o; o.x = (double)v.x; o.y = (double)v.y;
  return o;
}
template <typename TS> __device__ __forceinline__ void st_pair(TS* p, dbl2 v);
template <> __device__ __forceinline__ void st_pair<double>(double* p, dbl2 v) { *reinterpret_cast<dbl2*>(p) = v; }
template <> __device__ __forceinline__ void st_pair<float>(float* p, dbl2 v) {
  flt2 o; o.x = (float)v.x; o.y = (float)v.y;
  *reinterpret_cast<flt2*>(p) = o;
}

template <typename TS> struct Pair;                       // two stored rows as they sit in memory
template <> struct Pair<double> { typedef dbl2 type; };
template <> struct Pair<float> { typedef flt2 type; };
template <typename TS> __device__ __forceinline__ typename Pair<TS>::type ld_stream_raw(const TS* p) {
#if RBPF_NT_LOAD
  return __builtin_nontemporal_load(reinterpret_cast<const typename Pair<TS>::type*>(p));
#else
  return *reinterpret_cast<const typename Pair<TS>::type*>(p);
#endif
}

template <typename TS> __device__ __forceinline__ dbl2 ld_stream(const TS* p);
template <> __device__ __forceinline__ dbl2 ld_stream<double>(const double* p) {
#if RBPF_NT_LOAD
  return __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(p));
#else
  return *reinterpret_cast<const dbl2*>(p);
#endif
}
template <> __device__ __forceinline__ dbl2 ld_stream<float>(const float* p) {
#if RBPF_NT_LOAD
  const flt2 v = __builtin_nontemporal_load(reinterpret_cast<const flt2*>(p));
#else
  const flt2 v = *reinterpret_cast<const flt2*>(p);
#endif
  dbl2 o; o.x = (double)v.x; o.y = (double)v.y;
  return o;
}

template <typename TS> __device__ __forceinline__ void st_stream(TS* p, dbl2 v);
template <> __device__ __forceinline__ void st_stream<double>(double* p, dbl2 v) {
#if RBPF_NT_STORE
  __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(p));
#else
  *reinterpret_cast<dbl2*>(p) = v;
#endif
}
template <> __device__ __forceinline__ void st_stream<float>(float* p, dbl2 v) {
  flt2 o; o.x = (float)v.x; o.y = (float)v.y;
#if RBPF_NT_STORE
  __builtin_nontemporal_store(o, reinterpret_cast<flt2*>(p));
#else
  *reinterpret_cast<flt2*>(p) = o;
#endif
}

// ---------------------------------------------------------------------------------------------
// layout helpers (host + device)
// ---------------------------------------------------------------------------------------------
struct LdsPlan {
  int off_HK, off_xl, off_PHt, off_parts, off_tab, off_misc, off_red, off_kst, total;  // in doubles
};

__host__ __device__ inline int even_up(int x) { return (x + 1) & ~1; }

constexpr int kKBlock = 64;        // columns per stage of the blocked pending-factor buffer (KB variants of the step kernel)

// kb != 0: the pending column factors K of the NS sets are NOT kept for all n columns (n * NS * D doubles: 74 KB at n = 1027,
// NS = 2, and 37 KB on top of the information form's 5 right-hand sides at n = 515 -- one workgroup per CU); they pass through a
// double-buffered stage of kKBlock columns instead, refilled while the previous block streams.  With one column phase
// (CS == 1) the per-phase partial sums are not needed either: the accumulators go straight to PHt.
__host__ __device__ inline LdsPlan lds_plan(int n, int d, int e, int ns, int ldx, int CS, int mc, int ktot, int kb = 0) {
  LdsPlan p;
  int o = 0;
  p.off_HK = o;    o += even_up(n * (d + e + (kb ? 0 : ns * d)));
  p.off_xl = o;    o += ldx;
  p.off_PHt = o;   o += (d + e) * ldx;
  p.off_parts = o; o += (kb && CS == 1) ? 0 : CS * (d + e) * mc;
  p.off_tab = o;   o += even_up(2 * (ktot > 0 ? ktot : 1));
  p.off_misc = o;  o += 64;
  p.off_red = o;   o += kWaves * 16;
  p.off_kst = o;   o += kb ? 2 * kKBlock * ns * d : 0;
  p.total = o;
  return p;
}

// the 2 x 2 wave decomposition (one row chunk per wave): fewer registers per lane, used by the step-kernel
// variants that carry three or four pending factor sets
Layout make_layout_low_regs(int n, int d) {
  Layout L = make_layout(n, d);
  if (L.CH == 2) { L.RS = 2; L.CS = 2; L.CPL = L.CH / L.RS; }
  return L;
}

// The blocked variant is taken when the plain plan would leave one workgroup per CU -- or two that fill the CU's 160 KB to the
// last KB: measured at nLin = 515 (r02), plans of 80.2 KB (filter, three pending sets; information form, one set) are faster
// through the blocked stage (28.5 vs 29.1 ms per filter step, 8.65 vs 8.93 ms per smoother step), plans of 68 KB are not
// (29.8 vs 29.2 ms) -- and the wave decomposition has no remainder chunk (every wave runs the same block loop: it contains
// workgroup barriers).
#ifndef RBPF_KB_THRESHOLD_KB
#define RBPF_KB_THRESHOLD_KB 72
#endif
bool step_use_blocked(const ModelDev& m, const Layout& lay, int extra, int n_sets) {
  // (CPL <= 2: the blocked variants are only instantiated for one or two row chunks per wave -- launch_step_t; with three chunks,
  //  384 <= nLin < 512, the plain plan stays whatever its size.  r04: this function used to say "blocked" there, the launch sized
  //  the LDS for the blocked plan and then ran the plain kernel over it: wrong results for nLin 441..511 and for the information
  //  form from nLin = 384 on, sizes no test had touched; tests/test_gpu_filter.py test_every_row_chunk_count)
  if (n_sets < 1 || lay.mc == 0 || lay.CPL < 1 || lay.CPL > 2 || lay.CH != lay.CPL * lay.RS || lay.RS * lay.CS != kWaves) return false;
  return (size_t)lds_plan(lay.n, m.d, extra, n_sets, lay.ldx, lay.CS, lay.mc, m.ktot, 0).total * sizeof(double) > (size_t)RBPF_KB_THRESHOLD_KB * 1024;
}

size_t step_lds_bytes(const ModelDev& m, const Layout& lay, int extra, int n_sets) {
  const int kb = step_use_blocked(m, lay, extra, n_sets) ? 1 : 0;
  return (size_t)lds_plan(lay.n, m.d, extra, n_sets, lay.ldx, lay.CS, lay.mc, m.ktot, kb).total * sizeof(double);
}

Layout make_layout(int n, int d) {
  Layout L;
  L.n = n;
  L.mc = (n / kChunkRows) * kChunkRows;
  L.nb = n - L.mc;
  L.ldb = (n + 1) & ~1;
  L.ldx = (n + 1) & ~1;
  L.CH = L.mc / kChunkRows;
  // wave decomposition of the core block: RS waves along row chunks, CS column phases.
  if (L.CH >= kWaves) { L.RS = kWaves; L.CS = 1; }
  else if (L.CH == 3) { L.RS = 1; L.CS = kWaves; }
  else if (L.CH == 2) { L.RS = 1; L.CS = kWaves; }   // whole 2 KB columns per wave: measured 8% faster than 2x2
  else { L.RS = 1; L.CS = kWaves; }
  if (const char* e = tuning_env("RBPF_RS")) {          // tuning override: RS x CS must be <= 4
    const int rs = atoi(e);
    const char* e2 = tuning_env("RBPF_CS");
    const int cs = e2 ? atoi(e2) : kWaves / (rs > 0 ? rs : 1);
    if (rs >= 1 && cs >= 1 && rs * cs <= kWaves && L.CH > 0) { L.RS = rs; L.CS = cs; }
  }
  L.CPL = L.CH / L.RS;      // full rounds; the remaining CH % RS chunks go to the first waves
  L.sym = 0; L.CH64 = L.mc / kSymChunk;
  L.szT = (size_t)n * L.mc;
  L.szB = (size_t)L.nb * L.ldb;
  (void)d;
  return L;
}

// ---------------------------------------------------------------------------------------------
// the streamed core block: lanes own rows (2 per 128-row chunk), waves walk columns.
// Per column the wave needs DE = D+E right-hand-side values (H rows, then the E extra vectors of the
// information form) and the D pending column factors; LDS record per column: [H(D) | X(E) | K(D)].
// ---------------------------------------------------------------------------------------------
template <int NSA>
struct SetPtrs { const double* p[NSA]; };

template <typename TS, int D, int E, int CPL, int UC, int NS, bool WR>
__device__ __forceinline__ void stream_core(const TS* __restrict__ src, TS* __restrict__ dst,
                                            const double* __restrict__ HK, const SetPtrs<(NS > 0 ? NS : 1)> KSrows,
                                            int ldx, int n, int nb, int mc, int chunk0, int chunk_stride, int CS,
                                            int wc, int lane, double* __restrict__ out_acc /* [D+E][mc] */) {
  // Every chunk handled here is valid (the caller decides wave-uniformly), so the loop body carries
  // no predicates: all UC*CPL loads of a round are issued back to back and stay in flight together.
  // NS pending rank-D factor sets are applied on the fly; WR = false (a "light" step of the multi-step
  // lazy update) only reads the base matrix.
  constexpr int DE = D + E, ND = NS * D, REC = DE + ND, NDA = ND > 0 ? ND : 1;
  double acc[CPL][2][DE];
  double ks[CPL][2][NDA];
  int r0[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    r0[q] = (chunk0 + q * chunk_stride) * kChunkRows + 2 * lane;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int k = 0; k < DE; ++k) acc[q][e][k] = 0.0;
#pragma unroll
      for (int sset = 0; sset < NS; ++sset)
#pragma unroll
        for (int k = 0; k < D; ++k) ks[q][e][sset * D + k] = KSrows.p[sset][(size_t)k * ldx + nb + r0[q] + e];
    }
  }
  const size_t colstep = (size_t)CS * mc;               // elements between two columns of this wave
  const TS* sp = src + (size_t)wc * mc;
  TS* dp = dst + (size_t)wc * mc;
  const double* hk = HK + (size_t)wc * REC;
  const int hkstep = CS * REC;

  int c = wc;
  for (; c + (UC - 1) * CS < n; c += UC * CS) {
    typename Pair<TS>::type v[UC][CPL];                 // kept in the stored type until used: float keeps half the registers
#pragma unroll
    for (int u = 0; u < UC; ++u)
#pragma unroll
      for (int q = 0; q < CPL; ++q) v[u][q] = ld_stream_raw<TS>(sp + u * colstep + r0[q]);
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      double h[DE], kc[NDA];
#pragma unroll
      for (int k = 0; k < DE; ++k) h[k] = hk[u * hkstep + k];
#pragma unroll
      for (int k = 0; k < ND; ++k) kc[k] = hk[u * hkstep + DE + k];
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        double p0 = (double)v[u][q].x, p1 = (double)v[u][q].y;
#pragma unroll
        for (int k = 0; k < ND; ++k) { p0 = fma(-ks[q][0][k], kc[k], p0); p1 = fma(-ks[q][1][k], kc[k], p1); }
#pragma unroll
        for (int k = 0; k < DE; ++k) { acc[q][0][k] = fma(p0, h[k], acc[q][0][k]); acc[q][1][k] = fma(p1, h[k], acc[q][1][k]); }
        if (WR) { dbl2 o; o.x = p0; o.y = p1; st_stream(dp + u * colstep + r0[q], o); }
      }
    }
    sp += UC * colstep; dp += UC * colstep; hk += UC * hkstep;
  }
  for (; c < n; c += CS) {
    double h[DE], kc[NDA];
#pragma unroll
    for (int k = 0; k < DE; ++k) h[k] = hk[k];
#pragma unroll
    for (int k = 0; k < ND; ++k) kc[k] = hk[DE + k];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const dbl2 vv = ld_stream(sp + r0[q]);
      double p0 = vv.x, p1 = vv.y;
#pragma unroll
      for (int k = 0; k < ND; ++k) { p0 = fma(-ks[q][0][k], kc[k], p0); p1 = fma(-ks[q][1][k], kc[k], p1); }
#pragma unroll
      for (int k = 0; k < DE; ++k) { acc[q][0][k] = fma(p0, h[k], acc[q][0][k]); acc[q][1][k] = fma(p1, h[k], acc[q][1][k]); }
      if (WR) { dbl2 o; o.x = p0; o.y = p1; st_stream(dp + r0[q], o); }
    }
    sp += colstep; dp += colstep; hk += hkstep;
  }
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
#pragma unroll
    for (int k = 0; k < DE; ++k) {
      out_acc[(size_t)k * mc + r0[q]] = acc[q][0][k];
      out_acc[(size_t)k * mc + r0[q] + 1] = acc[q][1][k];
    }
  }
}

// The same stream with the pending column factors of the NS sets passing through a double-buffered LDS stage of kKBlock
// columns (KB variants).  All four waves of the workgroup run the block loop together (it has one barrier per block): the
// factors of block b + 1 are fetched into registers before block b streams and parked in the other stage afterwards.
// HK holds [H(D) | X(E)] per column only.  acc_ld: row stride of out_acc (mc for the per-phase partials, ldx when the
// accumulators go straight to PHt).
template <typename TS, int D, int E, int CPL, int UC, int NS, bool WR>
__device__ __forceinline__ void stream_core_blocked(const TS* __restrict__ src, TS* __restrict__ dst,
                                                    const double* __restrict__ HK, double* __restrict__ Kst,
                                                    const SetPtrs<(NS > 0 ? NS : 1)> Fsets, int ldx, int n, int nb, int mc,
                                                    int chunk0, int chunk_stride, int CS, int wc, int lane, int tid,
                                                    double* __restrict__ out_acc, int acc_ld) {
  constexpr int DE = D + E, ND = NS * D, NDA = ND > 0 ? ND : 1;
  constexpr int kPer = (NDA + 3) / 4;                                   // staged values per thread and block
  static_assert(kKBlock == 64, "one lane per column of a block");
  double acc[CPL][2][DE];
  double ks[CPL][2][NDA];
  int r0[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    r0[q] = (chunk0 + q * chunk_stride) * kChunkRows + 2 * lane;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int k = 0; k < DE; ++k) acc[q][e][k] = 0.0;
#pragma unroll
      for (int sset = 0; sset < NS; ++sset)
#pragma unroll
        for (int k = 0; k < D; ++k) ks[q][e][sset * D + k] = Fsets.p[sset][(size_t)k * ldx + nb + r0[q] + e];
    }
  }
  const size_t colstep = (size_t)CS * mc;
  const int nblk = (n + kKBlock - 1) / kKBlock;
  // stage entry (column cc of the block, factor k) at cc * ND + k.  Lane l fetches column l of the block; factor k is
  // fetched by wave k % 4 (static loop, wave-uniform predicate: no dynamic indexing of the set pointers)
  const int wave_id = tid >> 6;
  auto fetch = [&](int b, double (&v)[kPer]) {
    const int c = min(b * kKBlock + lane, n - 1);
#pragma unroll
    for (int k = 0; k < ND; ++k)
      if ((k & 3) == wave_id) v[k >> 2] = Fsets.p[NS > 0 ? k / D : 0][(size_t)(D + k % D) * ldx + c];
  };
  auto park = [&](int b, const double (&v)[kPer]) {
    double* st = Kst + (size_t)(b & 1) * kKBlock * NDA;
#pragma unroll
    for (int k = 0; k < ND; ++k)
      if ((k & 3) == wave_id) st[(size_t)lane * NDA + k] = v[k >> 2];
  };
  double pre[kPer];
  fetch(0, pre);
  park(0, pre);
  __syncthreads();
  for (int b = 0; b < nblk; ++b) {
    if (b + 1 < nblk) fetch(b + 1, pre);
    const double* st = Kst + (size_t)(b & 1) * kKBlock * NDA;
    const int cb = b * kKBlock, ce = min(n, cb + kKBlock);
    // first column of this wave inside the block: the smallest c >= cb with c % CS == wc
    int c = cb + ((wc - cb % CS) + CS) % CS;
    const TS* sp = src + (size_t)c * mc;
    TS* dp = dst + (size_t)c * mc;
    const double* hk = HK + (size_t)c * DE;
    const int hkstep = CS * DE;
    for (; c + (UC - 1) * CS < ce; c += UC * CS) {
      typename Pair<TS>::type v[UC][CPL];
#pragma unroll
      for (int u = 0; u < UC; ++u)
#pragma unroll
        for (int q = 0; q < CPL; ++q) v[u][q] = ld_stream_raw<TS>(sp + u * colstep + r0[q]);
#pragma unroll
      for (int u = 0; u < UC; ++u) {
        double h[DE], kc[NDA];
#pragma unroll
        for (int k = 0; k < DE; ++k) h[k] = hk[u * hkstep + k];
#pragma unroll
        for (int k = 0; k < ND; ++k) kc[k] = st[(size_t)(c + u * CS - cb) * NDA + k];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          double p0 = (double)v[u][q].x, p1 = (double)v[u][q].y;
#pragma unroll
          for (int k = 0; k < ND; ++k) { p0 = fma(-ks[q][0][k], kc[k], p0); p1 = fma(-ks[q][1][k], kc[k], p1); }
#pragma unroll
          for (int k = 0; k < DE; ++k) { acc[q][0][k] = fma(p0, h[k], acc[q][0][k]); acc[q][1][k] = fma(p1, h[k], acc[q][1][k]); }
          if (WR) { dbl2 o; o.x = p0; o.y = p1; st_stream(dp + u * colstep + r0[q], o); }
        }
      }
      sp += UC * colstep; dp += UC * colstep; hk += UC * hkstep;
    }
    for (; c < ce; c += CS) {
      double h[DE], kc[NDA];
#pragma unroll
      for (int k = 0; k < DE; ++k) h[k] = hk[k];
#pragma unroll
      for (int k = 0; k < ND; ++k) kc[k] = st[(size_t)(c - cb) * NDA + k];
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const dbl2 vv = ld_stream(sp + r0[q]);
        double p0 = vv.x, p1 = vv.y;
#pragma unroll
        for (int k = 0; k < ND; ++k) { p0 = fma(-ks[q][0][k], kc[k], p0); p1 = fma(-ks[q][1][k], kc[k], p1); }
#pragma unroll
        for (int k = 0; k < DE; ++k) { acc[q][0][k] = fma(p0, h[k], acc[q][0][k]); acc[q][1][k] = fma(p1, h[k], acc[q][1][k]); }
        if (WR) { dbl2 o; o.x = p0; o.y = p1; st_stream(dp + r0[q], o); }
      }
      sp += colstep; dp += colstep; hk += hkstep;
    }
    if (b + 1 < nblk) park(b + 1, pre);
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
#pragma unroll
    for (int k = 0; k < DE; ++k) {
      out_acc[(size_t)k * acc_ld + r0[q]] = acc[q][0][k];
      out_acc[(size_t)k * acc_ld + r0[q] + 1] = acc[q][1][k];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dynModel for every slot, one thread per particle (examples/slam-dense-mag/run_dense3D_magfield.m:301-308,
// examples/slam-dense-radio/run_dense2D_withHeading.m:75-76; particleFilter.m:108).  Kept out of the step
// kernel: there it was a single-lane serial section of 10-20 us per workgroup.
// ---------------------------------------------------------------------------------------------
__global__ void propagate_kernel(const StepArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;      // processing position of the step kernel
  if (b >= a.N) return;
  const ModelDev& M = a.mdl;
  const int nN = M.nN;
  const int i = a.order ? a.order[b] : b;
  const int gslot = a.slot_ids ? a.slot_ids[i] : a.slot_offset + i;
  const int anc = a.ai ? a.ai[a.slot_ids ? gslot : i] : i;
  const int ancb = a.ai_bank ? a.ai_bank[i] : anc;
  double x[8], xp[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) x[c] = (c < nN) ? a.xn_old[(size_t)c * a.xn_old_stride + anc] : 0.0;
  if (a.xn_ext != nullptr) {
#pragma unroll
    for (int c = 0; c < 8; ++c) xp[c] = (c < nN) ? a.xn_ext[(size_t)c * a.N + i] : 0.0;     // dynModel was evaluated on the host
  } else if (a.xref != nullptr && gslot == a.xref_gslot) {
#pragma unroll
    for (int c = 0; c < 8; ++c) xp[c] = (c < nN) ? a.xref[c] : 0.0;                 // particleSmoother.m:242
  } else if (a.propagate) {
    double z[8];
    if (a.rng_mode == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) z[k] = (k < M.nw) ? a.Z[(size_t)gslot * M.nw + k] : 0.0;
    } else {
      philox_normals(a.seed, gslot, a.t, a.k_iter, M.nw, z);
    }
    if (M.kind == 1) dyn_model_mag(x, a.odo, a.cholQ, z, xp);
    else dyn_model_radio(x, a.odo, a.cholQ, z, xp);
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) xp[c] = x[c];
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (c < nN) a.xn_new[(size_t)c * a.xn_new_stride + i] = xp[c];
  // next step's resampling uniforms (tools/sample.m:31) for this slot: off the critical path here, the
  // single-workgroup resample kernel then just loads them
  if (a.u_next) a.u_next[gslot] = philox_resample_uniform(a.seed, gslot, a.t + 1, a.k_iter);
  // per-workgroup descriptor: everything the step kernel would otherwise fetch through a chain of dependent
  // scattered loads while HBM is saturated
  int* pi = a.pre_i + (size_t)b * kPreInts;
  // lineage tables are indexed by the ancestor's slot in the local banks (== anc on one GPU; the bank index
  // in the sharded filter).  A migrated child (ancb in the record region) starts a fresh lineage: its record
  // holds the ancestor's matrix with every pending set already applied, so the old sets map to the zero entry.
  const bool imported = a.rec != nullptr && ancb >= a.n_bank_local;
  const int tix = a.slot_ids ? ancb : anc;
  pi[0] = i; pi[1] = anc; pi[2] = ancb;
  pi[3] = imported ? ancb : (a.base_old ? a.base_old[tix] : ancb);
#pragma unroll
  for (int sset = 0; sset < kMaxSets; ++sset) {
    int v = ancb;
    if (sset < a.n_sets && a.fset[sset]) v = imported ? a.zero_set_idx : (a.fset_idx_old[sset] ? a.fset_idx_old[sset][tix] : ancb);
    pi[kPreSet0 + sset] = v;
  }
  pi[4] = a.dst_slot ? a.dst_slot[i] : i;
  pi[5] = a.phase_of ? a.phase_of[i] : 0;
  if (a.distinct_mark) {                                    // accounting of the timed launches only
    const bool fresh = atomicExch(&a.distinct_mark[pi[3]], a.distinct_tag) != a.distinct_tag;
    const unsigned long long m = __ballot(fresh);
    if (fresh && (__ffsll((long long)m) - 1) == (int)(threadIdx.x & 63)) atomicAdd(a.distinct_counter, (unsigned long long)__popcll(m));
  }
  double* pd = a.pre_d + (size_t)b * kPreDoubles;
#pragma unroll
  for (int c = 0; c < 8; ++c) pd[c] = xp[c];
  if (M.kind == 1) quat2rmat_dev(&xp[3], &pd[8]);
  else { for (int c = 0; c < 9; ++c) pd[8 + c] = 0.0; }
}

hipError_t launch_propagate(const StepArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(propagate_kernel, dim3((a.N + 63) / 64), dim3(64), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// THE step kernel: one workgroup (4 wave64) per particle slot.
//   E = 0 : particleFilter.m:100-204 / particleSmoother.m:124-340 (covariance-form weights)
//   E = 1 : particleSmootherInformationForm.m:274-335 -- additionally streams P*ivec.  The reference also needs P*ivecPlus with
//           ivecPlus = ivec + H' R^-1 y (:292): that is P*ivec + (P H')(R^-1 y), formed from the accumulated columns instead of
//           streamed (r05; as rbpf_step_sym.hip does).  Same algebra, one right-hand side less in the stream -- and the two
//           quadratic forms of the importance weight (:301-303) then SHARE the rounding of P*ivec, which cancels in their difference:
//           against the extended-precision arbiter the weights of dense-radio went from 4.5e-9 (both products streamed; the fp64
//           C restatement: 2.9e-9) to the level of the block-lower kernel
// ---------------------------------------------------------------------------------------------
#ifndef RBPF_MINWAVES
#define RBPF_MINWAVES 1     // min waves per SIMD requested from the register allocator (tuning)
#endif
// UFX: extra unroll of the covariance stream for float storage when a wave owns two row chunks of WHOLE columns
// (RS = 4, CS = 1: n >= 1024); measured at n = 1027: 1: 0.62, 2: 0.39, 3: 0.68, 4: 0.76, 6: 0.67, 8: 0.70 M/s.  With
// CS = 4 (n = 259) the same factor costs a third of the throughput, hence a launch-time choice.
// KB: the pending column factors go through the blocked LDS stage (stream_core_blocked) instead of a record per column.
template <typename TS, int D, int E, int CPL, int NS, bool WR, int UFX = 1, bool KB = false>
__global__ __launch_bounds__(kThreads, RBPF_MINWAVES) void step_kernel(const StepArgs a) {
  extern __shared__ double smem[];
  constexpr int DE = D + E, ND = NS * D, REC = KB ? DE : DE + ND, NSA = NS > 0 ? NS : 1;
  const ModelDev& M = a.mdl;
  const Layout& Ly = a.lay;
  const int n = Ly.n, nb = Ly.nb, mc = Ly.mc, ldx = Ly.ldx, ldb = Ly.ldb;
  const int N = a.N;
  // children are processed in ancestor order so that siblings' reads of the same covariance hit the
  // Infinity Cache; the order only permutes the schedule, slot i still reads / writes slot i's data.
  // propagate_kernel resolved the indirections of this workgroup into one descriptor.
  const int pos = xcd_position((int)blockIdx.x, (int)gridDim.x);   // rbpf_internal.hpp: a family of siblings on ONE XCD
  const int* pre_i = a.pre_i + (size_t)pos * kPreInts;
  if (a.phase >= 0 && pre_i[5] != a.phase) return;     // single-bank flush: not this launch's share (workgroup-uniform)
  const int i = pre_i[0];
  const int dslot = WR ? pre_i[4] : i;                         // bank entry the rewritten matrix goes to
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const LdsPlan lp = lds_plan(n, D, E, NS, ldx, Ly.CS, mc, M.ktot, KB ? 1 : 0);
  double* HK = smem + lp.off_HK;        // per column c: H[0..D) | X[0..E) | Kcol of every pending set [NS][D] (not with KB)
  double* xls = smem + lp.off_xl;
  double* PHt = smem + lp.off_PHt;      // [DE][ldx]
  double* parts = smem + lp.off_parts;  // [CS][DE][mc]
  double* tabS = smem + lp.off_tab;
  double* tabC = tabS + (M.ktot > 0 ? M.ktot : 1);
  double* misc = smem + lp.off_misc;    // 0..7 xn_new, 8..16 Rnb, 20.. epilogue broadcast
  double* red = smem + lp.off_red;

#ifdef RBPF_STAMPS
#define RBPF_KSTAMP(k) if (blockIdx.x == 4000 && threadIdx.x == 0 && a.stamps) a.stamps[k] = __builtin_amdgcn_s_memrealtime()
#else
#define RBPF_KSTAMP(k)
#endif
  RBPF_KSTAMP(0);
  const int anc = pre_i[1];      // ancestor id for the non-linear / information-form state banks
  const int ancb = pre_i[2];     // ancestor id in the map bank (local | remote records)
  // sources of the ancestor's map state: the local bank, or a received record
  const bool remote = a.rec != nullptr && ancb >= a.n_bank_local;          // mean / legacy factors in a record
  const double* recp = remote ? a.rec + (size_t)(ancb - a.n_bank_local) * a.rec_stride : nullptr;
  // multi-step lazy update: the stored ("base") matrix of the ancestor's lineage may live in another slot
  const int baseb = pre_i[3];
  const bool remoteP = a.rec != nullptr && baseb >= a.n_bank_local;        // stored matrix in a (persisting) record
  const double* recP = remoteP ? a.rec + (size_t)(baseb - a.n_bank_local) * a.rec_stride : nullptr;
  // (a record keeps the covariance blocks in the stored type; rec_off_B is in doubles)
  const TS* srcT = remoteP ? reinterpret_cast<const TS*>(recP) : reinterpret_cast<const TS*>(a.Pt_old) + (size_t)baseb * a.Pt_old_stride;
  const TS* srcB = remoteP ? reinterpret_cast<const TS*>(recP + a.rec_off_B) : reinterpret_cast<const TS*>(a.Pb_old) + (size_t)baseb * a.Pb_old_stride;
  const double* srcX = remote ? recp + a.rec_off_X : a.xl_old + (size_t)ancb * a.xl_old_stride;
  // pending factor sets (KS rows then K columns, [2][D][ldx] each), oldest first
  SetPtrs<NSA> srcFs;
#pragma unroll
  for (int sset = 0; sset < NSA; ++sset) srcFs.p[sset] = nullptr;
#pragma unroll
  for (int sset = 0; sset < NS; ++sset) {
    if (a.fset[sset]) srcFs.p[sset] = a.fset[sset] + (size_t)pre_i[kPreSet0 + sset] * 2 * D * ldx;
    else srcFs.p[sset] = remote ? recp + a.rec_off_F : a.F_old + (size_t)ancb * 2 * D * ldx;
  }
  const int nN = M.nN;

  // ---- A: fetch the propagated non-linear state (propagate_kernel ran first), stage xl / pending K / ivec ----
  if (tid < kPreDoubles) misc[tid] = a.pre_d[(size_t)pos * kPreDoubles + tid];   // xn_new[0..8), Rnb[8..17)
  {
    // all loads of a pass are issued before the first LDS store (clamped indices, predicated stores): the
    // sources are scattered small arrays, so this phase is pure latency
    const double* xl_src = srcX;
    const double* iv = (E > 0) ? (remote ? recp + a.rec_off_I : a.ivec_old + (size_t)ancb * a.ivec_old_stride) : nullptr;
    constexpr int PB = 2;                                   // columns per thread per pass
    for (int c0 = tid; c0 < n; c0 += PB * kThreads) {
      double xv[PB], ivv[PB], kv[PB][ND > 0 ? ND : 1];
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const int c = min(c0 + u * kThreads, n - 1);
        xv[u] = xl_src[c];
        ivv[u] = (E > 0) ? iv[c] : 0.0;
        if (!KB) {
#pragma unroll
          for (int sset = 0; sset < NS; ++sset)
#pragma unroll
            for (int k = 0; k < D; ++k) kv[u][sset * D + k] = srcFs.p[sset][(size_t)(D + k) * ldx + c];
        }
      }
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const int c = c0 + u * kThreads;
        if (c < n) {
          xls[c] = xv[u];
          if (!KB) {
#pragma unroll
            for (int k = 0; k < ND; ++k) HK[c * REC + DE + k] = kv[u][k];
          }
          if (E > 0) HK[c * REC + D] = ivv[u];
        }
      }
    }
  }
  __syncthreads();

  RBPF_KSTAMP(1);
  // ---- B: per-axis sin/cos tables of the reduced-rank basis at the new position ----
  for (int q = tid; q < M.ktot; q += kThreads) basis_table_entry(M, q, misc, tabS, tabC);
  __syncthreads();

  RBPF_KSTAMP(2);
  // ---- C: measurement Jacobian H_i, column per thread (+ ivecPlus = ivec + dyi'/R*yt', :292) ----
  double Riy[D];                                               // R^-1 y
  if (E > 0) {
#pragma unroll
    for (int aa = 0; aa < D; ++aa) {
      double s = 0.0;
#pragma unroll
      for (int bb = 0; bb < D; ++bb) s = fma(M.Rinv[aa + D * bb], a.y[bb], s);
      Riy[aa] = s;
    }
  }
  {
    for (int c = tid; c < n; c += kThreads) {
      double h[D];
      if (a.H_ext != nullptr) {
#pragma unroll
        for (int k = 0; k < D; ++k) h[k] = a.H_ext[((size_t)i * D + k) * ldx + c];         // measModel was evaluated on the host
      } else {
        H_column<D>(M, c, tabS, tabC, &misc[8], h);
      }
#pragma unroll
      for (int k = 0; k < D; ++k) HK[c * REC + k] = h[k];
      if (E > 0) {
        double s = HK[c * REC + D];                                          // ivecPlus (:292), the new information vector (:333)
#pragma unroll
        for (int k = 0; k < D; ++k) s = fma(h[k], Riy[k], s);
#pragma unroll
        for (int k = 0; k < D; ++k) a.Hb_new[((size_t)i * D + k) * ldx + c] = h[k];
        a.ivec_new[(size_t)i * ldx + c] = s;
      }
    }
  }
  __syncthreads();

  RBPF_KSTAMP(3);
  // ---- D: stream the covariance once: apply pending downdate, store, accumulate P+ [H' X] ----
  {
    const int wr = wave % Ly.RS, wc = wave / Ly.RS;
#ifndef RBPF_UF32
#define RBPF_UF32 2
#endif
    // a float load brings half the bytes: twice the columns in flight keep the same bytes outstanding
    constexpr int kUF = ((sizeof(TS) == 4 && CPL <= 1) ? RBPF_UF32 : 1) * UFX;
    if (KB) {
      // every wave of the workgroup runs the block loop (launch-time guarantee: RS * CS == 4 waves, no remainder chunk)
      const TS* src = srcT;
      TS* dst = reinterpret_cast<TS*>(a.Pt_new) + (size_t)dslot * Ly.szT;
      const bool direct = Ly.CS == 1;                       // one column phase: accumulators go straight to PHt
      double* out_acc = direct ? PHt + nb : parts + (size_t)wc * DE * mc;
      stream_core_blocked<TS, D, E, (CPL > 0 ? CPL : 1), kUF * (CPL <= 1 ? RBPF_UC : (CPL == 2 ? RBPF_UC2 : RBPF_UC3)), NS, WR>(
          src, dst, HK, smem + lp.off_kst, srcFs, ldx, n, nb, mc, wr, Ly.RS, Ly.CS, wc, lane, tid, out_acc, direct ? ldx : mc);
    } else if (mc > 0 && wc < Ly.CS) {
      const TS* src = srcT;
      TS* dst = reinterpret_cast<TS*>(a.Pt_new) + (size_t)dslot * Ly.szT;
      double* out_acc = parts + (size_t)wc * DE * mc;
      // CPL full rounds of RS chunks (every wave), then one remainder chunk for the first CH % RS waves:
      // both decisions are wave-uniform, so the streaming loops are branch-free
      if (CPL > 0)
        stream_core<TS, D, E, (CPL > 0 ? CPL : 1), kUF * (CPL <= 1 ? RBPF_UC : (CPL == 2 ? RBPF_UC2 : RBPF_UC3)), NS, WR>(src, dst, HK, srcFs, ldx, n, nb, mc, wr, Ly.RS, Ly.CS, wc, lane, out_acc);
      const int rem = Ly.CH - CPL * Ly.RS;
      if (wr < rem)
        stream_core<TS, D, E, 1, kUF * RBPF_UC, NS, WR>(src, dst, HK, srcFs, ldx, n, nb, mc, CPL * Ly.RS + wr, 1, Ly.CS, wc, lane, out_acc);
    }
    // border rows (row-major block B): lanes walk columns, wave-reduce per row
    for (int b = wave; b < nb; b += kWaves) {
      const TS* src = srcB + (size_t)b * ldb;
      TS* dst = reinterpret_cast<TS*>(a.Pb_new) + (size_t)dslot * Ly.szB + (size_t)b * ldb;
      double ksb[ND > 0 ? ND : 1];
#pragma unroll
      for (int sset = 0; sset < NS; ++sset)
#pragma unroll
        for (int k = 0; k < D; ++k) ksb[sset * D + k] = srcFs.p[sset][(size_t)k * ldx + b];
      double accb[DE];
#pragma unroll
      for (int k = 0; k < DE; ++k) accb[k] = 0.0;
      for (int c = 2 * lane; c < ldb; c += 128) {
        const dbl2 vv = ld_pair<TS>(src + c);
        double p[2] = {vv.x, vv.y};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int cc = c + e;
          if (cc < n) {
#pragma unroll
            for (int k = 0; k < ND; ++k) {
              // pending column factor K(cc, k): from the per-column record, or (KB) straight from the factor set
              const double kcv = KB ? srcFs.p[NS > 0 ? k / D : 0][(size_t)(D + k % D) * ldx + cc] : HK[cc * REC + DE + k];
              p[e] = fma(-ksb[k], kcv, p[e]);
            }
#pragma unroll
            for (int k = 0; k < DE; ++k) accb[k] = fma(p[e], HK[cc * REC + k], accb[k]);
          }
        }
        if (WR) { dbl2 o; o.x = p[0]; o.y = p[1]; st_pair<TS>(dst + c, o); }
      }
#pragma unroll
      for (int k = 0; k < DE; ++k) {
        const double s = wave_sum(accb[k]);
        if (lane == 0) PHt[(size_t)k * ldx + b] = s;
      }
    }
  }
  __syncthreads();
  if (mc > 0 && !(KB && Ly.CS == 1)) {
    // fixed-order combine of the column phases (deterministic)
    for (int r = tid; r < mc; r += kThreads) {
#pragma unroll
      for (int k = 0; k < DE; ++k) {
        double s = parts[(size_t)k * mc + r];
        for (int w = 1; w < Ly.CS; ++w) s += parts[((size_t)w * DE + k) * mc + r];
        PHt[(size_t)k * ldx + nb + r] = s;
      }
    }
    __syncthreads();
  }

  RBPF_KSTAMP(4);
  // ---- E: innovation covariance S = H (P H') + R, innovation e = y - H xl (+ quadratic forms) ----
  constexpr int NRED = D * D + D + 2 * E;
  {
    double part[NRED];
#pragma unroll
    for (int q = 0; q < NRED; ++q) part[q] = 0.0;
    for (int r = tid; r < n; r += kThreads) {
      double h[D], ph[D];
#pragma unroll
      for (int k = 0; k < D; ++k) { h[k] = HK[r * REC + k]; ph[k] = PHt[(size_t)k * ldx + r]; }
      const double x = xls[r];
#pragma unroll
      for (int bb = 0; bb < D; ++bb)
#pragma unroll
        for (int aa = 0; aa < D; ++aa) part[aa + D * bb] = fma(h[aa], ph[bb], part[aa + D * bb]);
#pragma unroll
      for (int aa = 0; aa < D; ++aa) part[D * D + aa] = fma(h[aa], x, part[D * D + aa]);
      if (E > 0) {
        // ivec'*P*ivec and ivecPlus'*P*ivecPlus (:301-303) with P = the (downdated) prior covariance; P*ivecPlus = P*ivec + (P H')(R^-1 y)
        const double iv = HK[r * REC + D], piv = PHt[(size_t)D * ldx + r];
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
      if (lane == 0) red[wave * 16 + q] = s;
    }
  }
  __syncthreads();
  if (tid == 0) {
    double SS[D * D], e[D], cS[D * D], v[D];
    for (int q = 0; q < D * D; ++q) {
      double s = red[q];
      for (int w = 1; w < kWaves; ++w) s += red[w * 16 + q];
      SS[q] = s + M.R[q];                                                   // particleFilter.m:141
    }
    for (int q = 0; q < D; ++q) {
      double s = red[D * D + q];
      for (int w = 1; w < kWaves; ++w) s += red[w * 16 + D * D + q];
      e[q] = a.y[q] - s;                                                    // :140
    }
    bool ok = chol_lower_small<D>(SS, cS);                                  // :145
    if (!ok) {
      double SJ[D * D];
      for (int q = 0; q < D * D; ++q) SJ[q] = SS[q];
      for (int q = 0; q < D; ++q) SJ[q + D * q] += M.jitter;                // :147
      ok = chol_lower_small<D>(SJ, cS);
    }
    double lw = 0.0, sl = 0.0;
    if (ok) {
      fwd_subst<D>(cS, e, v);                                               // :149
      double vv = 0.0;
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
      for (int w = 1; w < kWaves; ++w) { qa += red[w * 16 + D * D + D]; qb += red[w * 16 + D * D + D + 1]; }
      misc[44] = qa; misc[45] = qb; misc[46] = ok ? sl : nan("");
    }
  }
  __syncthreads();

  RBPF_KSTAMP(5);
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
      // lineage bookkeeping of the multi-step lazy update: where this particle's stored matrix and its
      // surviving pending sets live (a flush makes them all obsolete)
      if (a.base_new) a.base_new[i] = WR ? dslot : (a.share_flush ? pre_i[4] : baseb);
      if (!WR) {
#pragma unroll
        for (int sset = 0; sset < NS; ++sset)
          if (a.fset_idx_new[sset]) a.fset_idx_new[sset][i] = pre_i[kPreSet0 + sset];
      }
      if (a.fself_idx_new) a.fself_idx_new[i] = i;
    }
    double* xln = a.xl_new + (size_t)i * ldx;
    double uK[D];
#pragma unroll
    for (int k = 0; k < D; ++k) uK[k] = 0.0;
    for (int r = tid; r < n; r += kThreads) {
      double ph[D], u[D], kk[D];
#pragma unroll
      for (int k = 0; k < D; ++k) ph[k] = PHt[(size_t)k * ldx + r];
      fwd_subst<D>(cS, ph, u);          // (P H') / cS'
      bwd_subst_T<D>(cS, u, kk);        // ... / cS   -> row r of K
      double xn_ = xls[r];
#pragma unroll
      for (int k = 0; k < D; ++k) xn_ = fma(kk[k], e[k], xn_);              // :197
      xln[r] = xn_;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) s = fma(kk[k], SS[k + D * j], s);       // row r of K*SS
        KSn[(size_t)j * ldx + r] = s;
        Kn[(size_t)j * ldx + r] = kk[j];
      }
      if (E > 0) {
        double ip = HK[r * REC + D];
#pragma unroll
        for (int k = 0; k < D; ++k) ip = fma(HK[r * REC + k], Riy[k], ip);
#pragma unroll
        for (int k = 0; k < D; ++k) uK[k] = fma(ip, kk[k], uK[k]);          // ivecPlus' * K
      }
    }
    if (E > 0) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double s = wave_sum(uK[k]);
        if (lane == 0) red[wave * 16 + k] = s;
      }
      __syncthreads();
      if (tid == 0) {
        double u[D];
        for (int k = 0; k < D; ++k) { double s = red[k]; for (int w = 1; w < kWaves; ++w) s += red[w * 16 + k]; u[k] = s; }
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
  __syncthreads();
  RBPF_KSTAMP(6);
}

template <typename TS, int D, int E, int CPL, int NS, bool WR, int UFX, bool KB>
static hipError_t launch_step_k(const StepArgs& a, size_t lds, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&step_kernel<TS, D, E, CPL, NS, WR, UFX, KB>), 160 * 1024, attr_done)) return e;
  hipLaunchKernelGGL((step_kernel<TS, D, E, CPL, NS, WR, UFX, KB>), dim3(a.N), dim3(kThreads), lds, s, a);
  return hipGetLastError();
}

template <typename TS, int D, int E, int CPL, int NS, bool WR>
static hipError_t launch_step_t(const StepArgs& a, hipStream_t s) {
  const size_t lds = step_lds_bytes(a.mdl, a.lay, E, NS);
  // launch-time unroll choices that depend on the wave decomposition (see step_kernel): float storage with whole
  // columns per wave and two row chunks (n >= 1024) x4; double storage in the 2 x 2 decomposition (the flush kernel of
  // the lazy update at n = 259) x2 (8.2 vs 7.7 M particle-steps/s at N = 8192)
  constexpr int kWide = (sizeof(TS) == 4 && CPL == 2) ? 4 : ((sizeof(TS) == 8 && CPL == 1 && E == 0) ? 2 : 1);
  bool wide = false;
  if constexpr (kWide > 1) wide = (sizeof(TS) == 4) ? (a.lay.CS == 1) : (a.lay.CS == 2 && a.lay.RS == 2);
  // pending factors through the blocked LDS stage when the per-column records would leave one workgroup per CU
  if constexpr (NS >= 1 && CPL >= 1 && CPL <= 2) {
    if (step_use_blocked(a.mdl, a.lay, E, NS)) {
      if constexpr (kWide > 1) { if (wide) return launch_step_k<TS, D, E, CPL, NS, WR, kWide, true>(a, lds, s); }
      return launch_step_k<TS, D, E, CPL, NS, WR, 1, true>(a, lds, s);
    }
  }
  if constexpr (kWide > 1) { if (wide) return launch_step_k<TS, D, E, CPL, NS, WR, kWide, false>(a, lds, s); }
  return launch_step_k<TS, D, E, CPL, NS, WR, 1, false>(a, lds, s);
}

template <typename TS, int D, int E, int NS, bool WR>
static hipError_t launch_step_cpl(const StepArgs& a, hipStream_t s) {
  switch (a.lay.CPL) {
    case 0: return launch_step_t<TS, D, E, 0, NS, WR>(a, s);
    case 1: return launch_step_t<TS, D, E, 1, NS, WR>(a, s);
    case 2: return launch_step_t<TS, D, E, 2, NS, WR>(a, s);
    case 3: return launch_step_t<TS, D, E, 3, NS, WR>(a, s);
    default: return hipErrorInvalidValue;
  }
}

// multi-step lazy variants: up to kMaxSets pending sets (3 for the information form, E = 1), light (read-only) or flush
template <typename TS, int D, int E>
static hipError_t launch_step_lazy(const StepArgs& a, hipStream_t s) {
  if (a.lay.CPL < 1 || a.lay.CPL > 2) return hipErrorInvalidValue;
#define RBPF_LZ(NS_, WR_) (a.lay.CPL == 1 ? launch_step_t<TS, D, E, 1, NS_, WR_>(a, s) : launch_step_t<TS, D, E, 2, NS_, WR_>(a, s))
  if (a.write_base) {
    switch (a.n_sets) {
      case 2: return RBPF_LZ(2, true);
      case 3: return RBPF_LZ(3, true);
      case 4: if constexpr (E == 0) return RBPF_LZ(4, true); else break;
      default: break;
    }
  } else {
    switch (a.n_sets) { case 1: return RBPF_LZ(1, false); case 2: return RBPF_LZ(2, false); case 3: if constexpr (E == 0) return RBPF_LZ(3, false); else break; default: break; }
  }
#undef RBPF_LZ
  return hipErrorInvalidValue;
}

hipError_t launch_step(const StepArgs& a, hipStream_t s) {
  const int D = a.mdl.d;
  if (a.n_sets < 0 || a.n_sets > kMaxSets) return hipErrorInvalidValue;
  if (a.lay.sym) return launch_step_sym(a, s);             // symmetric storage: rbpf_step_sym.hip
  const bool legacy = a.write_base && a.n_sets <= 1;       // one pending set, rewritten every step
  if (a.fp32) {
    // fp32 storage of the covariance banks: filter only (E = 0), dense-mag outputs (D = 3)
    if (a.info || D != 3) return hipErrorInvalidValue;
    if (!legacy) return launch_step_lazy<float, 3, 0>(a, s);
    return a.n_sets ? launch_step_cpl<float, 3, 0, 1, true>(a, s) : launch_step_cpl<float, 3, 0, 0, true>(a, s);
  }
  if (!legacy) {
    if (a.info) {                                          // information form: lazy_depth <= 3 (flush with 2 or 3 sets)
      if (D == 3) return launch_step_lazy<double, 3, 1>(a, s);
      if (D == 1) return launch_step_lazy<double, 1, 1>(a, s);
      return hipErrorInvalidValue;
    }
    if (D == 3) return launch_step_lazy<double, 3, 0>(a, s);
    if (D == 1) return launch_step_lazy<double, 1, 0>(a, s);
    return hipErrorInvalidValue;
  }
  if (a.info) {
    if (D == 3) return a.n_sets ? launch_step_cpl<double, 3, 1, 1, true>(a, s) : launch_step_cpl<double, 3, 1, 0, true>(a, s);
    if (D == 1) return a.n_sets ? launch_step_cpl<double, 1, 1, 1, true>(a, s) : launch_step_cpl<double, 1, 1, 0, true>(a, s);
    return hipErrorInvalidValue;
  }
  if (D == 3) return a.n_sets ? launch_step_cpl<double, 3, 0, 1, true>(a, s) : launch_step_cpl<double, 3, 0, 0, true>(a, s);
  if (D == 1) return a.n_sets ? launch_step_cpl<double, 1, 0, 1, true>(a, s) : launch_step_cpl<double, 1, 0, 0, true>(a, s);
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// weight normalisation (particleFilter.m:154-161) + strict left-to-right cumsum (tools/sample.m:30)
// one workgroup of 1024 threads
// ---------------------------------------------------------------------------------------------
constexpr int kNormThreads = 1024;
constexpr int kScanChunk = 4096;

__device__ inline double block_sum_1024(double v, double* sred) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sred[wave] = v;
  __syncthreads();
  double s = sred[0];
  for (int w = 1; w < kNormThreads / 64; ++w) s += sred[w];
  return s;
}

__device__ inline double block_max_1024(double v, double* sred) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if (lane == 0) sred[wave] = v;
  __syncthreads();
  double s = sred[0];
  for (int w = 1; w < kNormThreads / 64; ++w) s = fmax(s, sred[w]);
  return s;
}

// Strict left-to-right running sum (bit-identical to a sequential cumsum) by one lane.  The chain of
// dependent fp64 adds is the floor (~N x add latency); loads / stores are batched 16 wide through two
// distinct LDS arrays so they never sit on the dependency chain.
__device__ inline void strict_cumsum_block(int N, const double* __restrict__ w, double* __restrict__ wc,
                                           double* __restrict__ sin_, double* __restrict__ sout, double* scarry) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (tid == 0) *scarry = 0.0;
  __syncthreads();
  for (int base = 0; base < N; base += kScanChunk) {
    const int cnt = min(kScanChunk, N - base);
    for (int j = tid; j < cnt; j += nthr) sin_[j] = w[base + j];
    __syncthreads();
    if (tid == 0) {
      double run = *scarry;
      int j = 0;
      for (; j + 16 <= cnt; j += 16) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = sin_[j + k];
#pragma unroll
        for (int k = 0; k < 16; ++k) { run += v[k]; v[k] = run; }
#pragma unroll
        for (int k = 0; k < 16; ++k) sout[j + k] = v[k];
      }
      for (; j < cnt; ++j) { run += sin_[j]; sout[j] = run; }
      *scarry = run;
    }
    __syncthreads();
    for (int j = tid; j < cnt; j += nthr) wc[base + j] = sout[j];
    __syncthreads();
  }
}

// All passes below work on register batches of NB elements per thread with clamped (always in-bounds)
// indices and selects, so the NB loads of a batch are in flight together: a lone workgroup is latency-
// bound, and hipcc does not overlap the loads of a plain `for (i = tid; i < N; i += blockDim)` loop.
constexpr int kNB = 8;

__device__ inline void normalise_block(const NormArgs& a, double* sred, int* sidx, double* sbuf, double* sbuf2, double* scarry_p) {
  const int tid = threadIdx.x, N = a.N;
  const int lane = tid & 63, wave = tid >> 6;
  const int step = kNormThreads * kNB;

  // c = max(logw)   (fmax skips NaN like MATLAB's max)
  double mx = -INFINITY;
  for (int base = 0; base < N; base += step) {
    double v[kNB];
#pragma unroll
    for (int k = 0; k < kNB; ++k) { const int i = base + k * kNormThreads + tid; v[k] = a.logw[min(i, N - 1)]; }
#pragma unroll
    for (int k = 0; k < kNB; ++k) { const int i = base + k * kNormThreads + tid; mx = fmax(mx, i < N ? v[k] : -INFINITY); }
  }
  const double c = block_max_1024(mx, sred);
  // lse = c + log(sum(exp(logw - c)))
  double se = 0.0;
  for (int base = 0; base < N; base += step) {
    double v[kNB];
#pragma unroll
    for (int k = 0; k < kNB; ++k) { const int i = base + k * kNormThreads + tid; v[k] = a.logw[min(i, N - 1)]; }
#pragma unroll
    for (int k = 0; k < kNB; ++k) { const int i = base + k * kNormThreads + tid; const double e = exp(v[k] - c); se += (i < N) ? e : 0.0; }
  }
  const double tot = block_sum_1024(se, sred);
  const double lse = c + log(tot);
  // w = exp(logw - lse); [~,iw_max] = max(w) (first maximum); traj_mean(:,t) = sum(xn.*w,2) -- one pass,
  // one combined block reduction (fixed order -> deterministic)
  double bw = -1.0;
  int bi = 0x7fffffff;
  double ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int base = 0; base < N; base += step) {
    double v[kNB], wv[kNB];
#pragma unroll
    for (int k = 0; k < kNB; ++k) { const int i = base + k * kNormThreads + tid; v[k] = a.logw[min(i, N - 1)]; }
#pragma unroll
    for (int k = 0; k < kNB; ++k) {
      const int i = base + k * kNormThreads + tid;
      wv[k] = exp(v[k] - lse);
      if (i < N) a.w[i] = wv[k];
      const bool better = (i < N) && (wv[k] > bw);
      bw = better ? wv[k] : bw;
      bi = better ? i : bi;
    }
    for (int cix = 0; cix < a.nN; ++cix) {
      double xv[kNB];
#pragma unroll
      for (int k = 0; k < kNB; ++k) { const int i = base + k * kNormThreads + tid; xv[k] = a.xn[(size_t)cix * N + min(i, N - 1)]; }
#pragma unroll
      for (int k = 0; k < kNB; ++k) { const int i = base + k * kNormThreads + tid; ts[cix] = (i < N) ? fma(xv[k], wv[k], ts[cix]) : ts[cix]; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ow = __shfl_xor(bw, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ow > bw || (ow == bw && oi < bi)) { bw = ow; bi = oi; }
  }
  for (int cix = 0; cix < a.nN; ++cix) ts[cix] = wave_sum(ts[cix]);
  __syncthreads();
  if (lane == 0) {
    sred[wave] = bw; sidx[wave] = bi;
    for (int cix = 0; cix < a.nN; ++cix) sbuf[wave * 8 + cix] = ts[cix];
  }
  __syncthreads();
  double gw = sred[0];
  int gi = sidx[0];
  for (int w = 1; w < kNormThreads / 64; ++w)
    if (sred[w] > gw || (sred[w] == gw && sidx[w] < gi)) { gw = sred[w]; gi = sidx[w]; }
  if (gi < 0 || gi >= N) gi = 0;     // all weights NaN (failed chol): MATLAB's max returns index 1
  if (tid == 0) {
    *a.iw_max = gi;
    if (a.lse_out) *a.lse_out = lse;
  }
  if (tid < a.nN) {
    double tm = sbuf[tid];
    for (int w = 1; w < kNormThreads / 64; ++w) tm += sbuf[w * 8 + tid];
    if (a.traj_mean) a.traj_mean[tid] = tm;
    if (a.traj_max) a.traj_max[tid] = a.xn[(size_t)tid * N + gi];
  }
  __syncthreads();
  if (!a.parallel_scan) {
    // wc = cumsum(w): one lane, strict left-to-right, staged through LDS in chunks
    strict_cumsum_block(N, a.w, a.wc, sbuf, sbuf2, scarry_p);
    return;
  }
  // Parallel prefix sum (fixed association order -> deterministic).  It differs from the strict
  // left-to-right cumsum of tools/sample.m:30 only by rounding; the search bounds that difference
  // and recomputes the strict sum whenever a draw could be affected.
  {
    const int S = (N + kNormThreads - 1) / kNormThreads;
    const int j0 = min(tid * S, N), j1 = min(j0 + S, N);
    double tot = 0.0;
    for (int jb = j0; jb < j1; jb += kNB) {
      double v[kNB];
#pragma unroll
      for (int k = 0; k < kNB; ++k) v[k] = a.w[min(jb + k, N - 1)];
#pragma unroll
      for (int k = 0; k < kNB; ++k) tot += (jb + k < j1) ? v[k] : 0.0;
    }
    double inc = tot;                                   // inclusive scan of the per-thread totals
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    double exc = __shfl_up(inc, 1, 64);
    if (lane == 0) exc = 0.0;
    __syncthreads();
    if (lane == 63) sred[wave] = inc;
    __syncthreads();
    double woff = 0.0;
    for (int w = 0; w < wave; ++w) woff += sred[w];
    double run = woff + exc;
    for (int jb = j0; jb < j1; jb += kNB) {
      double v[kNB];
#pragma unroll
      for (int k = 0; k < kNB; ++k) v[k] = a.w[min(jb + k, N - 1)];
#pragma unroll
      for (int k = 0; k < kNB; ++k)
        if (jb + k < j1) { run += v[k]; a.wc[jb + k] = run; }
    }
  }
}

// Lower bound of SB uniforms at once: the SB binary searches advance in lock step so that their
// (dependent-chain) loads overlap instead of serialising.  Result = #{j : wc_j < u}.
template <int SB>
__device__ __forceinline__ void lower_bound_batch(const double* __restrict__ wc, int N, int rounds, const double* u,
                                                  const bool* valid, int* out) {
  int lo[SB], hi[SB];
#pragma unroll
  for (int k = 0; k < SB; ++k) { lo[k] = 0; hi[k] = valid[k] ? N : 0; }
  for (int r = 0; r < rounds; ++r) {
    double v[SB];
    int mid[SB];
#pragma unroll
    for (int k = 0; k < SB; ++k) { mid[k] = (lo[k] + hi[k]) >> 1; v[k] = wc[min(mid[k], N - 1)]; }
#pragma unroll
    for (int k = 0; k < SB; ++k) {                    // selects only: no divergent control flow
      const bool active = lo[k] < hi[k];
      const bool less = v[k] < u[k];
      lo[k] = (active && less) ? mid[k] + 1 : lo[k];
      hi[k] = (active && !less) ? mid[k] : hi[k];
    }
  }
#pragma unroll
  for (int k = 0; k < SB; ++k) out[k] = lo[k];
}

// block-wide search of n_draw uniforms with the exact fallback inline.  One CU cannot sustain N scattered
// 8-byte global loads per bisection round, so the bisection runs on an LDS table: the running sums
// themselves when N <= 2*kScanChunk, otherwise the last element of every group of G consecutive sums
// followed by a short linear count inside the one group (a single cache line for G <= 8).
__device__ inline void search_block(const SearchArgs& a, double* sbuf, double* sbuf2, double* scarry_p, int* sflag) {
  constexpr int SB = 8;
  constexpr int CAP = 2 * kScanChunk;                  // sbuf and sbuf2 are one contiguous array
  const int tid = threadIdx.x, N = a.N;
  double* T = sbuf;
  if (tid == 0) *sflag = 0;
  __syncthreads();
  const double S = (double)((N + kNormThreads - 1) / kNormThreads);
  int G = 1;
  while ((N + G - 1) / G > CAP) G <<= 1;
  const int ng = (N + G - 1) / G;
  int rounds = 1;
  while ((1 << rounds) <= ng) ++rounds;               // ceil(log2(ng+1)) rounds settle every search
  for (int pass = 0; pass < 2; ++pass) {
    const double* wc = (pass == 0) ? a.wc : a.wc_exact;
    for (int base = 0; base < ng; base += kNormThreads * SB) {
      double v[SB];
#pragma unroll
      for (int k = 0; k < SB; ++k) { const int j = base + k * kNormThreads + tid; v[k] = wc[min(min(j, ng - 1) * G + G - 1, N - 1)]; }
#pragma unroll
      for (int k = 0; k < SB; ++k) { const int j = base + k * kNormThreads + tid; if (j < ng) T[j] = v[k]; }
    }
    __syncthreads();
    for (int base = 0; base < a.n_draw; base += kNormThreads * SB) {
      double u[SB];
      bool valid[SB];
      int res[SB];
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        const int i0 = base + k * kNormThreads + tid;
        valid[k] = i0 < a.n_draw;
        const int i = a.slot0 + i0;
        u[k] = !valid[k] ? 0.0
               : (a.rng_mode == 0) ? a.U[a.u_is_scalar ? 0 : i] : philox_resample_uniform(a.seed, i, a.t, a.k_iter);
      }
      lower_bound_batch<SB>(T, ng, rounds, u, valid, res);          // group index (LDS)
      double p_hi[SB], p_lo[SB];
      if (G > 1) {                                                   // count inside the group (global)
        for (int e = 0; e < G; ++e) {
          double w[SB];
#pragma unroll
          for (int k = 0; k < SB; ++k) w[k] = wc[min(min(res[k], ng - 1) * G + e, N - 1)];
#pragma unroll
          for (int k = 0; k < SB; ++k) {
            const int j = res[k] * G + e;          // res[k] still the group index in this loop
            p_lo[k] = w[k];                        // placeholder use keeps the load live
            if (e == 0) p_hi[k] = 0.0;
            p_hi[k] += (res[k] < ng && j < N && w[k] < u[k]) ? 1.0 : 0.0;
          }
        }
#pragma unroll
        for (int k = 0; k < SB; ++k) res[k] = (res[k] >= ng) ? N : min(res[k] * G + (int)p_hi[k], N);
#pragma unroll
        for (int k = 0; k < SB; ++k) { p_hi[k] = wc[min(res[k], N - 1)]; p_lo[k] = wc[max(res[k] - 1, 0)]; }
      } else {
#pragma unroll
        for (int k = 0; k < SB; ++k) { p_hi[k] = T[min(res[k], N - 1)]; p_lo[k] = T[max(res[k] - 1, 0)]; }
      }
      bool amb = false;
      int nover = 0;
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        const int lo = res[k];
        const double tol = 1.7e-16 * ((double)lo + S + 32.0);
        const bool a_hi = (lo < N) && (fabs(p_hi[k] - u[k]) <= tol * fabs(p_hi[k]));
        const bool a_lo = (lo > 0) && (fabs(p_lo[k] - u[k]) <= tol * fabs(p_lo[k]));
        amb |= valid[k] && (a_hi || a_lo);
        nover += (valid[k] && lo >= N) ? 1 : 0;
        res[k] = min(lo, N - 1);
      }
#pragma unroll
      for (int k = 0; k < SB; ++k)
        if (valid[k]) a.ai[a.slot0 + base + k * kNormThreads + tid] = res[k];
      if (pass == 0 && a.approx && amb) atomicOr(sflag, 1);
      if (pass == 0 && a.overflow && nover) atomicAdd(a.overflow, nover);
    }
    __syncthreads();
    if (pass == 1 || !*sflag) break;
    // rare: a draw within the rounding bound of a bin edge -> strict cumsum, then redo every draw exactly
    strict_cumsum_block(N, a.w, a.wc_exact, sbuf, sbuf2, scarry_p);
    __syncthreads();
    if (tid == 0 && a.ambiguous) atomicAdd(a.ambiguous + 1, 1);
  }
}

// counting sort of the slots by key (ancestor): order[] lists the slots so that equal keys are adjacent.
// The order among equal keys is arbitrary (atomics) -- it only changes the schedule, never a result.
__device__ inline void order_block(int n_slots, int range, const int* __restrict__ key, int* __restrict__ order,
                                   int* __restrict__ counts_global, int* sred_i, int* lds_counts, int lds_capacity,
                                   const int* __restrict__ remap = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* counts = (range <= lds_capacity) ? lds_counts : counts_global;     // LDS atomics when the histogram fits
  __syncthreads();
  for (int j = tid; j < range; j += kNormThreads) counts[j] = 0;
  __syncthreads();
  for (int i = tid; i < n_slots; i += kNormThreads) atomicAdd(&counts[remap ? remap[key[i]] : key[i]], 1);
  __syncthreads();
  const int S = (range + kNormThreads - 1) / kNormThreads;
  const int j0 = min(tid * S, range), j1 = min(j0 + S, range);
  int tot = 0;
  for (int j = j0; j < j1; ++j) tot += __hip_atomic_load(&counts[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) sred_i[wave] = inc;
  __syncthreads();
  int run = inc - tot;
  for (int w = 0; w < wave; ++w) run += sred_i[w];
  for (int j = j0; j < j1; ++j) {
    const int cnt = __hip_atomic_load(&counts[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&counts[j], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    run += cnt;
  }
  __syncthreads();
  for (int i = tid; i < n_slots; i += kNormThreads) order[atomicAdd(&counts[remap ? remap[key[i]] : key[i]], 1)] = i;
}

__global__ __launch_bounds__(kNormThreads) void normalise_scan_kernel(const NormArgs a) {
  __shared__ double sred[16];
  __shared__ int sidx[16];
  __shared__ double sbig[2 * kScanChunk];
  __shared__ double scarry;
  normalise_block(a, sred, sidx, sbig, sbig + kScanChunk, &scarry);
}

// normalise step t, then (filter fast path) draw the ancestors of step t+1 and their processing order
__global__ __launch_bounds__(kNormThreads) void normalise_resample_kernel(const NormArgs a, const SearchArgs sa,
                                                                          int* order, int* counts, const int* remap) {
  __shared__ double sred[16];
  __shared__ int sidx[16];
  __shared__ double sbig[2 * kScanChunk];
  __shared__ double scarry;
  __shared__ int sflag;
  double* sbuf = sbig;
  double* sbuf2 = sbig + kScanChunk;
#ifdef RBPF_STAMPS   // diagnostic build only: phase times (100 MHz ticks) into the tail of the counting-sort scratch
#define RBPF_STAMP(k) if (threadIdx.x == 0) { reinterpret_cast<unsigned long long*>(counts + 2 * sa.N)[k] = __builtin_amdgcn_s_memrealtime(); reinterpret_cast<unsigned long long*>(counts + 2 * sa.N)[4 + k] = __builtin_amdgcn_s_memtime(); }
#else
#define RBPF_STAMP(k)
#endif
  RBPF_STAMP(0);
  normalise_block(a, sred, sidx, sbuf, sbuf2, &scarry);
  __syncthreads();
  RBPF_STAMP(1);
  search_block(sa, sbuf, sbuf2, &scarry, &sflag);
  __syncthreads();
  RBPF_STAMP(2);
  if (order) order_block(sa.n_draw, sa.N, sa.ai, order, counts, sidx, reinterpret_cast<int*>(sbuf), 2 * kScanChunk, remap);
  __syncthreads();
  RBPF_STAMP(3);
}

__global__ __launch_bounds__(kNormThreads) void order_kernel(int n_slots, int range, const int* key, int* order, int* counts,
                                                             const int* remap) {
  __shared__ int sidx[16];
  __shared__ int scnt[2 * kScanChunk];
  order_block(n_slots, range, key, order, counts, sidx, scnt, 2 * kScanChunk, remap);
}

hipError_t launch_normalise_scan(const NormArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(normalise_scan_kernel, dim3(1), dim3(kNormThreads), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_normalise_resample(const NormArgs& a, const SearchArgs& sa, int* order, int* counts, hipStream_t s,
                                     const int* remap) {
  hipLaunchKernelGGL(normalise_resample_kernel, dim3(1), dim3(kNormThreads), 0, s, a, sa, order, counts, remap);
  return hipGetLastError();
}

hipError_t launch_order(int n_slots, int range, const int* key, int* order, int* counts, hipStream_t s, const int* remap) {
  if (range > kSingleWgResampleMaxN) return launch_order_large(n_slots, range, key, remap, order, counts, s);   // no LDS histogram
  hipLaunchKernelGGL(order_kernel, dim3(1), dim3(kNormThreads), 0, s, n_slots, range, key, order, counts, remap);
  return hipGetLastError();
}

// standalone strict left-to-right cumsum (tools/sample.m:30) used by rbpf_sample and the smoother
__global__ __launch_bounds__(kNormThreads) void cumsum_kernel(int N, const double* __restrict__ w,
                                                              double* __restrict__ wc) {
  __shared__ double sbuf[kScanChunk];
  __shared__ double sbuf2[kScanChunk];
  __shared__ double scarry;
  strict_cumsum_block(N, w, wc, sbuf, sbuf2, &scarry);
}

hipError_t launch_cumsum(int N, const double* w, double* wc, hipStream_t s) {
  hipLaunchKernelGGL(cumsum_kernel, dim3(1), dim3(kNormThreads), 0, s, N, w, wc);
  return hipGetLastError();
}

// ind = sum(wc < u) (0-based), clamped to N-1 (the reference would return N+1 -> MATLAB error).
// approx mode: wc is the parallel prefix p~.  Both p~_j and the strict cumsum wc_j approximate the
// exact prefix S_j: |wc_j - S_j| <= (j+1) u S_j, |p~_j - S_j| <= (S + 24) u S_j (u = 2^-53), so every
// comparison (wc_j < u) is decided by p~_j unless |p~_j - u| <= eps_j = 1.5 (j + S + 32) u p~_j.  Since
// p~ is monotone up to rounding, only the two entries bracketing the search result can be that close.
__global__ void search_kernel(const SearchArgs a) {
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 >= a.n_draw) return;
  const int i = a.slot0 + i0;
  const double u = (a.rng_mode == 0) ? a.U[a.u_is_scalar ? 0 : i] : philox_resample_uniform(a.seed, i, a.t, a.k_iter);
  int lo = 0, hi = a.N;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a.wc[mid] < u) lo = mid + 1; else hi = mid;
  }
  if (a.approx) {
    const double S = a.scan_depth > 0 ? (double)a.scan_depth : (double)((a.N + kNormThreads - 1) / kNormThreads) + 32.0;
    bool amb = false;
    if (lo < a.N) { const double p = a.wc[lo]; amb |= fabs(p - u) <= 1.7e-16 * ((double)lo + S) * fabs(p); }
    if (lo > 0) { const double p = a.wc[lo - 1]; amb |= fabs(p - u) <= 1.7e-16 * ((double)lo + S) * fabs(p); }
    if (amb) atomicAdd(a.ambiguous, 1);
  }
  if (lo >= a.N) { lo = a.N - 1; if (a.overflow) atomicAdd(a.overflow, 1); }
  a.ai[i] = lo;
}

// Runs after an approx search: nothing to do unless a draw was flagged; then the strict cumsum is
// formed (one lane, sequential) and every slot is re-drawn against it.
__global__ __launch_bounds__(kNormThreads) void resample_fixup_kernel(const SearchArgs a) {
  __shared__ double sbuf[kScanChunk];
  __shared__ double sbuf2[kScanChunk];
  __shared__ double scarry;
  if (*a.ambiguous == 0) return;
  strict_cumsum_block(a.N, a.w, a.wc_exact, sbuf, sbuf2, &scarry);
  __syncthreads();
  for (int i0 = threadIdx.x; i0 < a.n_draw; i0 += blockDim.x) {
    const int i = a.slot0 + i0;
    const double u = (a.rng_mode == 0) ? a.U[a.u_is_scalar ? 0 : i] : philox_resample_uniform(a.seed, i, a.t, a.k_iter);
    int lo = 0, hi = a.N;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (a.wc_exact[mid] < u) lo = mid + 1; else hi = mid;
    }
    if (lo >= a.N) lo = a.N - 1;
    a.ai[i] = lo;
  }
  __syncthreads();
  if (threadIdx.x == 0) { atomicAdd(a.ambiguous + 1, 1); *a.ambiguous = 0; }   // [+1]: fallback counter (diagnostics)
}

hipError_t launch_resample_fixup(const SearchArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(resample_fixup_kernel, dim3(1), dim3(kNormThreads), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_search(const SearchArgs& a, hipStream_t s) {
  const int nb = (a.n_draw + 255) / 256;
  if (nb > 0) hipLaunchKernelGGL(search_kernel, dim3(nb), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// pack / unpack / extraction kernels
// ---------------------------------------------------------------------------------------------
template <typename TS>
__global__ void pack_P_kernel(Layout L, const double* __restrict__ P, size_t src_stride, TS* __restrict__ Pt,
                              TS* __restrict__ Pb) {
  const int p = blockIdx.x;
  const double* src = P + (size_t)p * src_stride;
  TS* t = Pt + (size_t)p * L.szT;
  TS* b = Pb + (size_t)p * L.szB;
  const size_t nn = (size_t)L.n * L.n;
  for (size_t q = threadIdx.x; q < nn; q += blockDim.x) {
    const int r = (int)(q % L.n), c = (int)(q / L.n);
    const double v = src[q];
    if (r < L.nb) b[(size_t)r * L.ldb + c] = (TS)v;
    else if (!L.sym) t[(size_t)c * L.mc + (r - L.nb)] = (TS)v;
    else if (c >= L.nb && (r - L.nb) / kSymChunk >= (c - L.nb) / kSymChunk) t[sym_t_index(r - L.nb, c - L.nb, sym_cg(L))] = (TS)v;   // lower block triangle
  }
  // zero the pad column of the border block
  if (L.ldb > L.n)
    for (int r = threadIdx.x; r < L.nb; r += blockDim.x) b[(size_t)r * L.ldb + L.n] = (TS)0;
}

hipError_t launch_pack_P(const Layout& lay, const double* P_colmajor, size_t src_stride, double* Pt, double* Pb,
                         int count, hipStream_t s, int fp32) {
  if (fp32) hipLaunchKernelGGL(pack_P_kernel<float>, dim3(count), dim3(256), 0, s, lay, P_colmajor, src_stride,
                               reinterpret_cast<float*>(Pt), reinterpret_cast<float*>(Pb));
  else hipLaunchKernelGGL(pack_P_kernel<double>, dim3(count), dim3(256), 0, s, lay, P_colmajor, src_stride, Pt, Pb);
  return hipGetLastError();
}

// flushes the pending downdates: P(r,c) = stored(r,c) - sum over sets, k of KS(r,k) K(c,k)
struct UnpackSets {
  int n_sets;
  const double* fset[kMaxSets];
  const int* fidx[kMaxSets];      // entry of each particle in that bank (null: the particle's own index)
  const int* base;                // slot of the stored matrix (null: the particle's own index)
};

// grid (count, ceil(n / 16)): a workgroup expands 16 columns of one particle
constexpr int kUnpackCols = 16;
template <typename TS>
__global__ __launch_bounds__(256) void unpack_P_kernel(Layout L, int d, const TS* __restrict__ Pt,
                                                       const TS* __restrict__ Pb, UnpackSets us,
                                                       const int* __restrict__ index, double* __restrict__ P) {
  const int p = blockIdx.x;
  const int src = index ? index[p] : p;
  const int bsl = us.base ? us.base[src] : src;
  const TS* t = Pt + (size_t)bsl * L.szT;
  const TS* b = Pb + (size_t)bsl * L.szB;
  double* dst = P + (size_t)p * L.n * L.n;
  const int c0 = blockIdx.y * kUnpackCols, c1 = min(L.n, c0 + kUnpackCols);
  const double* F[kMaxSets];
  for (int sset = 0; sset < kMaxSets; ++sset)
    F[sset] = sset < us.n_sets ? us.fset[sset] + (size_t)(us.fidx[sset] ? us.fidx[sset][src] : src) * 2 * d * L.ldx : nullptr;
  for (int r = threadIdx.x; r < L.n; r += 256) {
    for (int c = c0; c < c1; ++c) {
      double v;
      if (r < L.nb) v = (double)b[(size_t)r * L.ldb + c];
      else if (!L.sym) v = (double)t[(size_t)c * L.mc + (r - L.nb)];
      else v = (c < L.nb) ? (double)b[(size_t)c * L.ldb + r] : (double)t[sym_t_index(r - L.nb, c - L.nb, sym_cg(L))];   // mirror image where not stored
      for (int sset = 0; sset < us.n_sets; ++sset)
        for (int k = 0; k < d; ++k) v = fma(-F[sset][(size_t)k * L.ldx + r], F[sset][(size_t)(d + k) * L.ldx + c], v);
      dst[(size_t)c * L.n + r] = v;
    }
  }
}

static hipError_t launch_unpack_impl(const Layout& lay, int d, const double* Pt, const double* Pb, const UnpackSets& us,
                                     const int* index, int count, double* P_colmajor, hipStream_t s, int fp32) {
  const dim3 grid(count, (lay.n + kUnpackCols - 1) / kUnpackCols);
  if (fp32) hipLaunchKernelGGL(unpack_P_kernel<float>, grid, dim3(256), 0, s, lay, d, reinterpret_cast<const float*>(Pt),
                               reinterpret_cast<const float*>(Pb), us, index, P_colmajor);
  else hipLaunchKernelGGL(unpack_P_kernel<double>, grid, dim3(256), 0, s, lay, d, Pt, Pb, us, index, P_colmajor);
  return hipGetLastError();
}

hipError_t launch_unpack_P(const Layout& lay, int d, const double* Pt, const double* Pb, const double* F,
                           const int* index, int count, double* P_colmajor, hipStream_t s, int fp32) {
  UnpackSets us;
  us.n_sets = F ? 1 : 0; us.base = nullptr;
  for (int q = 0; q < kMaxSets; ++q) { us.fset[q] = nullptr; us.fidx[q] = nullptr; }
  us.fset[0] = F;
  return launch_unpack_impl(lay, d, Pt, Pb, us, index, count, P_colmajor, s, fp32);
}

hipError_t launch_unpack_P_sets(const Layout& lay, int d, const double* Pt, const double* Pb, int n_sets,
                                const double* const* fset, const int* const* fidx, const int* base, const int* index,
                                int count, double* P_colmajor, hipStream_t s, int fp32) {
  UnpackSets us;
  us.n_sets = n_sets; us.base = base;
  for (int q = 0; q < kMaxSets; ++q) { us.fset[q] = q < n_sets ? fset[q] : nullptr; us.fidx[q] = q < n_sets ? fidx[q] : nullptr; }
  return launch_unpack_impl(lay, d, Pt, Pb, us, index, count, P_colmajor, s, fp32);
}

// xl_mean = sum(xl.*w,2)  (particleFilter.m:226)
__global__ void weighted_mean_xl_kernel(int N, int n, int ldx, const double* __restrict__ xl,
                                        const double* __restrict__ w, double* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double s = 0.0;
  for (int i = 0; i < N; ++i) s = fma(xl[(size_t)i * ldx + r], w[i], s);
  out[r] = s;
}

hipError_t launch_weighted_mean_xl(int N, int n, int ldx, const double* xl, const double* w, double* out,
                                   hipStream_t s) {
  hipLaunchKernelGGL(weighted_mean_xl_kernel, dim3((n + 63) / 64), dim3(64), 0, s, N, n, ldx, xl, w, out);
  return hipGetLastError();
}

// Ancestral paths: the lazily evaluated equivalent of the eager history permutation
// xn_traj(:,:,1:t-1) = xn_traj(:,ai,1:t-1)  (particleFilter.m:118).  X: [T][nN][N], A: [T][N].
__global__ void backtrace_kernel(int N, int nN, int T, const double* __restrict__ X, const int* __restrict__ A,
                                 const int* __restrict__ start_index, int n_paths, double* __restrict__ out, int path0) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_paths) return;
  int j = start_index ? start_index[p] : path0 + p;          // (path0: first path of a chunk of all paths)
  for (int t = T - 1; t >= 0; --t) {
    for (int c = 0; c < nN; ++c)
      out[c + (size_t)nN * (p + (size_t)n_paths * t)] = X[((size_t)t * nN + c) * N + j];
    if (t > 0) j = A[(size_t)t * N + j];
  }
}

hipError_t launch_backtrace(int N, int nN, int T, const double* X, const int* A, const int* start_index,
                            int n_paths, double* out, hipStream_t s, int path0) {
  hipLaunchKernelGGL(backtrace_kernel, dim3((n_paths + 63) / 64), dim3(64), 0, s, N, nN, T, X, A, start_index,
                     n_paths, out, path0);
  return hipGetLastError();
}

__global__ void philox_fill_kernel(unsigned long long seed, int k_iter, int N, int T, int nw, double* U, double* Z,
                                   double* Ufin) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q == 0 && Ufin) *Ufin = philox_resample_uniform(seed, 0, T, k_iter);
  if (q >= (size_t)N * (T - 1)) return;
  const int i = (int)(q % N), t = (int)(q / N) + 1;
  U[q] = philox_resample_uniform(seed, i, t, k_iter);
  double z[8];
  philox_normals(seed, i, t, k_iter, nw, z);
  for (int k = 0; k < nw; ++k) Z[q * nw + k] = z[k];
}

hipError_t launch_philox_fill(unsigned long long seed, int k_iter, int N, int T, int nw, double* U, double* Z,
                              double* Ufin, hipStream_t s) {
  const size_t tot = (size_t)N * (T > 1 ? T - 1 : 0);
  const int nb = (int)((tot + 255) / 256);
  hipLaunchKernelGGL(philox_fill_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, s, seed, k_iter, N, T, nw, U, Z, Ufin);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// standalone model kernels (parity tests of SURVEY 8a rows a3-a8, a17, a19)
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ void meas_model_kernel(ModelDev M, int npred, const double* __restrict__ xn, double* __restrict__ dy,
                                  int layout) {
  extern __shared__ double sm[];
  double* tabS = sm;
  double* tabC = sm + (M.ktot > 0 ? M.ktot : 1);
  double* misc = tabC + (M.ktot > 0 ? M.ktot : 1);
  const int p = blockIdx.x;
  if (threadIdx.x == 0) {
    for (int c = 0; c < M.nN; ++c) misc[c] = xn[(size_t)p * M.nN + c];
    if (M.kind == 1) quat2rmat_dev(&misc[3], &misc[8]);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < M.ktot; q += blockDim.x) basis_table_entry(M, q, misc, tabS, tabC);
  __syncthreads();
  for (int c = threadIdx.x; c < M.n; c += blockDim.x) {
    double h[D];
    H_column<D>(M, c, tabS, tabC, &misc[8], h);
    for (int k = 0; k < D; ++k) dy[layout ? ((size_t)p * D + k) * M.n + c : ((size_t)p * M.n + c) * D + k] = h[k];
  }
}

hipError_t launch_meas_model(const ModelDev& m, int npred, const double* xn, double* dy, hipStream_t s, int layout) {
  const size_t lds = (size_t)(2 * (m.ktot > 0 ? m.ktot : 1) + 32) * sizeof(double);
  if (m.d == 3) hipLaunchKernelGGL((meas_model_kernel<3>), dim3(npred), dim3(256), lds, s, m, npred, xn, dy, layout);
  else if (m.d == 1) hipLaunchKernelGGL((meas_model_kernel<1>), dim3(npred), dim3(256), lds, s, m, npred, xn, dy, layout);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

__global__ void dyn_model_kernel(ModelDev M, int np, const double* __restrict__ xn, const double* __restrict__ odo,
                                 const double* __restrict__ cholQ, const double* __restrict__ z,
                                 double* __restrict__ xn_next) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  double x[8], xp[8], zz[8];
  for (int c = 0; c < M.nN; ++c) x[c] = xn[(size_t)i * M.nN + c];
  for (int k = 0; k < M.nw; ++k) zz[k] = z[(size_t)i * M.nw + k];
  if (M.kind == 1) dyn_model_mag(x, odo, cholQ, zz, xp);
  else dyn_model_radio(x, odo, cholQ, zz, xp);
  for (int c = 0; c < M.nN; ++c) xn_next[(size_t)i * M.nN + c] = xp[c];
}

hipError_t launch_dyn_model(const ModelDev& m, int np, const double* xn, const double* odo, const double* cholQ,
                            const double* z, double* xn_next, hipStream_t s) {
  hipLaunchKernelGGL(dyn_model_kernel, dim3((np + 63) / 64), dim3(64), 0, s, m, np, xn, odo, cholQ, z, xn_next);
  return hipGetLastError();
}

// eDyn = r' / chol(dt*Q,'lower'): solve x*Lq = r'  <=>  Lq' x' = r (back substitution on Lq')
__device__ inline void dyn_res_norm_dev(const ModelDev& M, const double* xk, const double* xi, const double* odo,
                                        const double* Lq, double* ed) {
  double r[8];
  const int nw = M.nw;
  if (M.use_dyn_res_norm && M.kind == 1) {
    // run_dense3D_magfield.m:202-203
    for (int c = 0; c < 3; ++c) r[c] = xk[c] - xi[c] - odo[c];
    const double oqi[4] = {odo[3], -odo[4], -odo[5], -odo[6]};
    const double xqi[4] = {xi[3], -xi[4], -xi[5], -xi[6]};
    double t1[4], t2[4];
    qleft_mul(oqi, xqi, t1);
    qleft_mul(t1, &xk[3], t2);
    logq_dev(t2, &r[3]);
  } else if (M.use_dyn_res_norm && M.kind == 2) {
    r[0] = xk[2] - xi[2] - odo[2];                          // run_dense2D_withHeading.m:77
  } else {
    for (int c = 0; c < nw; ++c) r[c] = xk[c] - xi[c] - odo[c];   // particleSmoother.m:176
  }
  for (int q = nw - 1; q >= 0; --q) {
    double s = r[q];
    for (int k = q + 1; k < nw; ++k) s -= Lq[k + nw * q] * ed[k];
    ed[q] = s / Lq[q + nw * q];
  }
}

__global__ void dyn_res_norm_kernel(ModelDev M, int np, const double* __restrict__ xnk_t,
                                    const double* __restrict__ xn, const double* __restrict__ odo,
                                    const double* __restrict__ Lq, double* __restrict__ e_dyn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  double xi[8], xk[8], ed[8];
  for (int c = 0; c < M.nN; ++c) { xi[c] = xn[(size_t)i * M.nN + c]; xk[c] = xnk_t[c]; }
  dyn_res_norm_dev(M, xk, xi, odo, Lq, ed);
  for (int k = 0; k < M.nw; ++k) e_dyn[(size_t)i * M.nw + k] = ed[k];
}

hipError_t launch_dyn_res_norm(const ModelDev& m, int np, const double* xnk_t, const double* xn, const double* odo,
                               const double* cholQfull, double* e_dyn, hipStream_t s) {
  hipLaunchKernelGGL(dyn_res_norm_kernel, dim3((np + 63) / 64), dim3(64), 0, s, m, np, xnk_t, xn, odo, cholQfull,
                     e_dyn);
  return hipGetLastError();
}

// tools/JacobianPhi3D.m:29-64
__global__ void jacobian_phi3d_kernel(ModelDev M, int np, const double* __restrict__ x, const double* __restrict__ lo,
                                      const double* __restrict__ up, double* __restrict__ J) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (size_t)M.m * np) return;
  const int j = (int)(q % M.m), i = (int)(q / M.m);
  double f[3], s[3], c[3];
  for (int d = 0; d < 3; ++d) {
    const double jd = (double)M.NN[d * M.m + j];
    const double ba = up[d] - lo[d];
    f[d] = (RBPF_PI * jd) / ba;
    const double core = RBPF_PI * jd * (x[(size_t)i * 3 + d] - lo[d]) / ba;
    const double mult = 1.0 / sqrt(0.5 * ba);
    double sn, cs;
    sincos(core, &sn, &cs);
    s[d] = sn * mult;
    c[d] = cs * mult;
  }
  double* Jp = J + ((size_t)i * M.m + j) * 9;   // [3 x 3] column-major per (j,i)
  Jp[0 + 3 * 0] = -f[0] * f[0] * s[0] * s[1] * s[2];
  Jp[0 + 3 * 1] = f[0] * f[1] * c[0] * c[1] * s[2];
  Jp[0 + 3 * 2] = f[0] * f[2] * c[0] * s[1] * c[2];
  Jp[1 + 3 * 0] = f[1] * f[0] * c[0] * c[1] * s[2];
  Jp[1 + 3 * 1] = -f[1] * f[1] * s[0] * s[1] * s[2];
  Jp[1 + 3 * 2] = f[1] * f[2] * s[0] * c[1] * c[2];
  Jp[2 + 3 * 0] = f[2] * f[0] * c[0] * s[1] * c[2];
  Jp[2 + 3 * 1] = f[2] * f[1] * s[0] * c[1] * c[2];
  Jp[2 + 3 * 2] = -f[2] * f[2] * s[0] * s[1] * s[2];
}

hipError_t launch_jacobian_phi3d(const ModelDev& m, int np, const double* x, const double* lo, const double* up,
                                 double* J, hipStream_t s) {
  const size_t tot = (size_t)m.m * np;
  hipLaunchKernelGGL(jacobian_phi3d_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, m, np, x, lo, up, J);
  return hipGetLastError();
}

// tools/ quaternion helpers, one thread per column (rbpf_quat_helpers)
__global__ void quat_helpers_kernel(int op, int n, const double* __restrict__ in, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int nin = (op <= 1 || op == 8) ? 3 : 4;
  double v[4] = {0.0, 0.0, 0.0, 0.0}, r[16];
  for (int c = 0; c < nin; ++c) v[c] = in[(size_t)i * nin + c];
  int nout = 0;
  switch (op) {
    case 0: expq_dev(v, r); nout = 4; break;
    case 1: expq_batched_dev(v, r); nout = 4; break;
    case 2: logq_dev(v, r); nout = 3; break;
    case 3: logq_batched_dev(v, r); nout = 3; break;
    case 4: qleft_mat_dev(v, r); nout = 16; break;
    case 5: qright_mat_dev(v, r); nout = 16; break;
    case 6: qinv_dev(v, r); nout = 4; break;
    case 7: {
      double Rm[9];
      quat2rmat_dev(v, Rm);                                   // Rm[row*3+col] -> column-major
      for (int rr = 0; rr < 3; ++rr) for (int cc = 0; cc < 3; ++cc) r[rr + 3 * cc] = Rm[rr * 3 + cc];
      nout = 9; break;
    }
    case 8: mcross_dev(v, r); nout = 9; break;
    default: break;
  }
  for (int c = 0; c < nout; ++c) out[(size_t)i * nout + c] = r[c];
}

hipError_t launch_quat_helpers(int op, int n, const double* in, double* out, hipStream_t s) {
  hipLaunchKernelGGL(quat_helpers_kernel, dim3((n + 127) / 128), dim3(128), 0, s, op, n, in, out);
  return hipGetLastError();
}

// SoA [nN][N] -> MATLAB [nN x N]
__global__ void transpose_soa_kernel(int N, int nN, const double* __restrict__ soa, double* __restrict__ aos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int c = 0; c < nN; ++c) aos[(size_t)i * nN + c] = soa[(size_t)c * N + i];
}

hipError_t launch_transpose_soa(int N, int nN, const double* soa, double* aos, hipStream_t s) {
  hipLaunchKernelGGL(transpose_soa_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, nN, soa, aos);
  return hipGetLastError();
}

// copy the bank entries of `count` particles into particle-major records [Pt | Pb | F | xl].  With fp32 storage the two
// covariance blocks keep their stored type inside the record (they take szT / 2 and szB / 2 doubles; both are even).
template <typename TS>
__global__ void pack_records_kernel(Layout L, int d, const int* __restrict__ idx, const TS* __restrict__ Pt,
                                    const TS* __restrict__ Pb, const double* __restrict__ F,
                                    const double* __restrict__ xl, double* __restrict__ rec, size_t rec_stride) {
  constexpr size_t PER = sizeof(double) / sizeof(TS);       // stored elements per double
  const int p = blockIdx.x;
  const int src = idx[p];
  const size_t szF = (size_t)2 * d * L.ldx;
  const size_t offB = L.szT / PER, offF = (L.szT + L.szB) / PER;
  const size_t recsz = rec_stride ? rec_stride : offF + szF + L.ldx;
  double* r = rec + (size_t)p * recsz;
  const double* a = reinterpret_cast<const double*>(Pt + (size_t)src * L.szT);       // raw 8-byte words
  const double* bsrc = reinterpret_cast<const double*>(Pb + (size_t)src * L.szB);
  for (size_t q = threadIdx.x; q < offB; q += blockDim.x) r[q] = a[q];
  for (size_t q = threadIdx.x; q < L.szB / PER; q += blockDim.x) r[offB + q] = bsrc[q];
  for (size_t q = threadIdx.x; q < szF; q += blockDim.x) r[offF + q] = F[(size_t)src * szF + q];
  for (size_t q = threadIdx.x; q < (size_t)L.ldx; q += blockDim.x) r[offF + szF + q] = xl[(size_t)src * L.ldx + q];
}

// Same records, but with the lineage's pending factor sets applied while packing (sharded filter with the
// multi-step lazy update): the receiver gets a plain matrix and starts a fresh lineage.
struct PackSets {
  int n_sets, n_bank_local;
  const double* fset[kMaxSets];
  const int* fidx[kMaxSets];
  const int* base;
  const double* rec; size_t rec_stride;       // bases of previously imported lineages live in the record buffer
};

template <typename TS>
__global__ void pack_records_flushed_kernel(Layout L, int d, const int* __restrict__ idx, const TS* __restrict__ Pt,
                                            const TS* __restrict__ Pb, PackSets ps, const double* __restrict__ xl,
                                            double* __restrict__ out) {
  constexpr size_t PER = sizeof(double) / sizeof(TS);
  const int p = blockIdx.x;
  const int src = idx[p];
  const size_t szF = (size_t)2 * d * L.ldx;
  const size_t offB = L.szT / PER, offF = (L.szT + L.szB) / PER;
  const size_t recsz = ps.rec_stride ? ps.rec_stride : offF + szF + L.ldx;
  double* r = out + (size_t)p * recsz;
  TS* rt = reinterpret_cast<TS*>(r);
  TS* rb = reinterpret_cast<TS*>(r + offB);
  const int bsl = ps.base ? ps.base[src] : src;
  const bool inrec = ps.rec != nullptr && bsl >= ps.n_bank_local;
  const double* recp = inrec ? ps.rec + (size_t)(bsl - ps.n_bank_local) * ps.rec_stride : nullptr;
  const TS* t = inrec ? reinterpret_cast<const TS*>(recp) : Pt + (size_t)bsl * L.szT;
  const TS* b = inrec ? reinterpret_cast<const TS*>(recp + offB) : Pb + (size_t)bsl * L.szB;
  const double* F[kMaxSets];
  for (int s = 0; s < ps.n_sets; ++s) F[s] = ps.fset[s] + (size_t)(ps.fidx[s] ? ps.fidx[s][src] : src) * szF;
  for (size_t q = threadIdx.x; q < L.szT; q += blockDim.x) {
    int c, rr;
    if (!L.sym) { c = (int)(q / L.mc); rr = L.nb + (int)(q % L.mc); }
    else {
      // inverse of sym_t_index: tile (I, J) of the lower block triangle, column pair, lane, element
      const int tile = (int)(q / kSymTile), within = (int)(q % kSymTile);
      int I = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
      while ((I + 1) * (I + 2) / 2 <= tile) ++I;
      while (I * (I + 1) / 2 > tile) --I;
      const int J = tile - I * (I + 1) / 2;
      const int cg = sym_cg(L);
      rr = L.nb + I * kSymChunk + (within % (cg * kSymChunk)) / cg;
      c = L.nb + J * kSymChunk + cg * (within / (cg * kSymChunk)) + (within % cg);
    }
    double v = (double)t[q];
    for (int s = 0; s < ps.n_sets; ++s)
      for (int k = 0; k < d; ++k) v = fma(-F[s][(size_t)k * L.ldx + rr], F[s][(size_t)(d + k) * L.ldx + c], v);
    rt[q] = (TS)v;
  }
  for (size_t q = threadIdx.x; q < L.szB; q += blockDim.x) {
    const int rr = (int)(q / L.ldb), c = (int)(q % L.ldb);
    double v = (double)b[q];
    if (c < L.n)
      for (int s = 0; s < ps.n_sets; ++s)
        for (int k = 0; k < d; ++k) v = fma(-F[s][(size_t)k * L.ldx + rr], F[s][(size_t)(d + k) * L.ldx + c], v);
    rb[q] = (TS)v;
  }
  for (size_t q = threadIdx.x; q < szF; q += blockDim.x) r[offF + q] = 0.0;
  for (size_t q = threadIdx.x; q < (size_t)L.ldx; q += blockDim.x) r[offF + szF + q] = xl[(size_t)src * L.ldx + q];
}

hipError_t launch_pack_records_flushed(const Layout& lay, int d, const int* idx, int count, const double* Pt,
                                       const double* Pb, int n_sets, const double* const* fset, const int* const* fidx,
                                       const int* base, int n_bank_local, const double* rec, size_t rec_stride,
                                       const double* xl, double* out, hipStream_t s, int fp32) {
  if (count <= 0) return hipSuccess;
  PackSets ps;
  ps.n_sets = n_sets; ps.n_bank_local = n_bank_local; ps.base = base; ps.rec = rec; ps.rec_stride = rec_stride;
  for (int q = 0; q < kMaxSets; ++q) { ps.fset[q] = q < n_sets ? fset[q] : nullptr; ps.fidx[q] = q < n_sets ? fidx[q] : nullptr; }
  if (fp32) hipLaunchKernelGGL(pack_records_flushed_kernel<float>, dim3(count), dim3(256), 0, s, lay, d, idx,
                               reinterpret_cast<const float*>(Pt), reinterpret_cast<const float*>(Pb), ps, xl, out);
  else hipLaunchKernelGGL(pack_records_flushed_kernel<double>, dim3(count), dim3(256), 0, s, lay, d, idx, Pt, Pb, ps, xl, out);
  return hipGetLastError();
}

hipError_t launch_pack_records(const Layout& lay, int d, const int* idx, int count, const double* Pt, const double* Pb,
                               const double* F, const double* xl, double* rec, hipStream_t s, size_t rec_stride, int fp32) {
  if (count <= 0) return hipSuccess;
  if (fp32) hipLaunchKernelGGL(pack_records_kernel<float>, dim3(count), dim3(256), 0, s, lay, d, idx, reinterpret_cast<const float*>(Pt),
                               reinterpret_cast<const float*>(Pb), F, xl, rec, rec_stride);
  else hipLaunchKernelGGL(pack_records_kernel<double>, dim3(count), dim3(256), 0, s, lay, d, idx, Pt, Pb, F, xl, rec, rec_stride);
  return hipGetLastError();
}

// all_gather layout [world][rows][Nloc] in PHYSICAL slot order (rows 0..nN-1: xn, row nN: logw; rows = nN + 2 in the sharded
// smoother, whose extra row carries the measurement part of the ancestor log-weights) ->
// logical order: logw[i], SoA xn[c][i], extra[i], with i = logical slot id and phys_of_logical[i] = rank*Nloc + idx
__global__ void permute_fwd_kernel(int N, int nN, int rows, int Nloc, const int* __restrict__ phys_of_logical,
                                   const double* __restrict__ fwd_gather, double* __restrict__ logw,
                                   double* __restrict__ xn_soa, double* __restrict__ extra) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int ph = phys_of_logical ? phys_of_logical[i] : i;
  const int r = ph / Nloc, j = ph % Nloc;
  const double* blk = fwd_gather + (size_t)r * rows * Nloc;
  for (int c = 0; c < nN; ++c) xn_soa[(size_t)c * N + i] = blk[(size_t)c * Nloc + j];
  if (logw) logw[i] = blk[(size_t)nN * Nloc + j];
  if (extra) extra[i] = blk[(size_t)(nN + 1) * Nloc + j];
}

hipError_t launch_permute_fwd(int N, int nN, int world, int Nloc, const int* phys_of_logical, const double* fwd_gather,
                              double* logw, double* xn_soa, hipStream_t s, int rows, double* extra) {
  (void)world;
  hipLaunchKernelGGL(permute_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, nN, rows > 0 ? rows : nN + 1, Nloc, phys_of_logical,
                     fwd_gather, logw, xn_soa, extra);
  return hipGetLastError();
}

// xl bank [N][ldx] -> MATLAB [n x N]
__global__ void gather_xl_kernel(int N, int n, int ldx, const double* __restrict__ xl, double* __restrict__ out) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (size_t)N * n) return;
  const int r = (int)(q % n), i = (int)(q / n);
  out[q] = xl[(size_t)i * ldx + r];
}

hipError_t launch_gather_xl(int N, int n, int ldx, const double* xl, double* out_colmajor, hipStream_t s) {
  const size_t tot = (size_t)N * n;
  hipLaunchKernelGGL(gather_xl_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, N, n, ldx, xl,
                     out_colmajor);
  return hipGetLastError();
}

}  // namespace rbpf
