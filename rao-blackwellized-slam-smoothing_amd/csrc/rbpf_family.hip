// P_base * [H_1' ... H_f'] for a FAMILY of particles that share one stored covariance, on the fp64 matrix cores (r04).
//
// particleFilter.m:139-141,185-198 needs P_i * H_i' per particle.  Between two flushes of the lazy update the particles of a family
// read the SAME stored matrix (0.27 .. 0.43 N distinct matrices per step at N = 65 536 and lazy_depth 4, fewer the longer the last
// flush lies back), and the one-workgroup-per-particle read-only step kernel is bound by its instruction stream, not by memory
// (rank 3 against 64 lanes: every transposed tile product ends in cross-lane sums).  With the members' Jacobians side by side,
// H = [H_1' ... H_f'] is n x 3 f -- a real GEMM: 5 members fill 15 of the 16 columns of a v_mfma_f64_16x16x4 tile, both products
// of a stored tile T(I, J), T * H_J (rows of block I) and T' * H_I (rows of block J), come out of the matrix pipe with no cross-lane
// add at all, and the matrix is read ONCE per family by construction instead of once per member through the L2.
//
// A read-only step of the filter on block-lower storage is then three launches (launch_step_sym_family):
//   family_prepare_kernel   measModel of every processing position -> H [pos][3][ldx]; family_index_kernel: the runs of equal stored
//                           matrix in the (sorted) processing order -> family table
//   family_pht_kernel       this file's GEMM: core rows of P_base * H_i' -> [pos][3][mc]
//   step_sym_kernel<.., PX> everything else of the step per particle (border rows / columns, pending sets, S, weight, gain, mean):
//                           rbpf_step_sym.hip with the tile stream replaced by a read of the product
//
// family_pht_kernel: one workgroup (8 waves) per family at a time.  The members' H in LDS ([column][16], 64 KB at 512 core columns);
// the stored tiles stream through LDS whole (32 KB of consecutive memory, two buffers, four more tiles in flight in registers),
// column pairs padded to 132 doubles so that the transposed operand reads spread over the banks; four waves form T * H_J, four form
// T' * H_I (see the kernel), 16 MFMAs per wave and tile; the two sums meet in LDS at the end.  Core rows only.
#include "rbpf_internal.hpp"
#include "rbpf_device.hpp"
#include "rbpf_model_dev.hpp"

namespace rbpf {

typedef double fv4d __attribute__((ext_vector_type(4)));
typedef double fv2d __attribute__((ext_vector_type(2)));

constexpr int kFamThreads = 512;
#ifndef RBPF_FAM_PS
#define RBPF_FAM_PS 132
#endif
constexpr int kFamPS = RBPF_FAM_PS;                       // doubles between two column pairs of a half tile in LDS (128 + pad)
constexpr int kFamTile = 32 * kFamPS;             // one tile buffer
constexpr int kFamMembers = 5;                    // members per pass: 15 of the 16 MFMA columns
#ifndef RBPF_FAM_DEPTH
#define RBPF_FAM_DEPTH 4                         // tiles in flight per workgroup (registers): 32 KB each
#endif
constexpr int kFamDepth = RBPF_FAM_DEPTH;

// T: [n_mat][CH (CH + 1) / 2 tiles][4096]  (Layout::sym block T; an entry >= n_bank_local is record (entry - n_bank_local) of
// `rec`, the sharded filter's received particles), H: [N][3][ldh] (core column c at off + c), PHt: [N][3][mc]; family f = positions
// fam_start[f] .. fam_start[f + 1] - 1, all reading matrix fam_base[f]; *n_fam families.
//
// Roles per stored tile (I, J) (128 MFMAs): wave w < 4 forms T * H_J for row tile w (K = the 64 tile columns), wave 4 + c forms
// T' * H_I for column tile c (K = the 64 tile rows): 16 MFMAs per wave and tile whatever the wave, no split of K, CH accumulators.
//
// Persistent workgroups (one per CU) walk the families blockIdx.x, + gridDim.x, ...; a family of more than 5 members takes several
// passes over its matrix.  The ring of D tiles in flight runs on ACROSS passes and families (the tiles of the next pass are
// requested during the last D steps of this one), so that the memory pipe stays full through the end of a pass, the exchange of
// the sums and the layout of the next H.
template <int CH, int D>
__global__ __launch_bounds__(kFamThreads, 1) void family_pht_kernel(const FamilyArgs fa) {
  const double* __restrict__ T = fa.T;
  const double* __restrict__ H = fa.H;
  const int* __restrict__ fam_start = fa.fam_start;
  const int* __restrict__ fam_base = fa.fam_base;
  double* __restrict__ PHt = fa.PHt;
  const size_t t_stride = fa.t_stride, ldh = fa.ldh;
  const int F = *fa.n_fam;
  auto matrix = [&](int entry) -> const double* {
    return (fa.rec != nullptr && entry >= fa.n_bank_local) ? fa.rec + (size_t)(entry - fa.n_bank_local) * fa.rec_stride : T + (size_t)entry * t_stride;
  };
  constexpr int mc = 64 * CH, NT = CH * (CH + 1) / 2;
  static_assert(NT % D == 0 && NT % 2 == 0, "the ring slot and the buffer of a tile must not depend on the pass");
  extern __shared__ double fsm[];
  double* Hm = fsm;                                // [mc][16]
  double* tb = fsm + (size_t)mc * 16;              // [2][kFamTile]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  int fam = blockIdx.x;
  if (fam >= F) return;
  int pm = fam_start[fam], p1 = fam_start[fam + 1];
  const double* Tm = matrix(fam_base[fam]);
  // staging: thread t moves doubles 2 t, 2 t + 1 of each quarter of a tile (1024 consecutive doubles = 8 column pairs; one request of a
  // wave = 1 KB of consecutive memory): pair t / 64, offset (2 t) % 128
  const int st_src = 2 * tid, st_dst = (tid >> 6) * kFamPS + ((2 * tid) & 127);
  const bool rows_role = wv < 4;
  const int wq = wv & 3;
  // operand addresses inside a tile buffer (doubles): rows role steps 2 kFamPS per K group, columns role 8
  const int a_off = rows_role ? (g >> 1) * kFamPS + (16 * wq + r16) * 2 + (g & 1) : (8 * wq + (r16 >> 1)) * kFamPS + g * 2 + (r16 & 1);
  const int a_step = rows_role ? 2 * kFamPS : 8;
  // H of a pass travels in registers, requested before that pass's tiles: thread t holds member column t / 32 of the core columns
  // t % 32 + 32 i
  const int hn = tid >> 5, hj = hn / 3, hk = hn - 3 * hj, hc = tid & 31;
  double hv[mc / 32];
  {
    const double* hp = H + ((size_t)(pm + min(hj, min(kFamMembers, p1 - pm) - 1)) * 3 + min(hk, 2)) * ldh + fa.h_off + hc;
#pragma unroll
    for (int i = 0; i < mc / 32; ++i) hv[i] = hp[32 * i];
  }
  fv2d R[D][4];
#pragma unroll
  for (int u = 0; u < D; ++u) {
    const double* src = Tm + (size_t)u * 4096 + st_src;
#pragma unroll
    for (int x = 0; x < 4; ++x) R[u][x] = *reinterpret_cast<const fv2d*>(src + 1024 * x);
  }
  for (;;) {
    // the pass after this one: the same matrix again, the next family's, or none (then the ring refills from this matrix, unused)
    int nfam = fam, npm = pm + kFamMembers, np1 = p1;
    const double* Tn = Tm;
    bool more = true;
    if (npm >= p1) {
      nfam = fam + (int)gridDim.x;
      if (nfam < F) {
        npm = fam_start[nfam];
        np1 = fam_start[nfam + 1];
        Tn = matrix(fam_base[nfam]);
      } else {
        more = false;
        npm = pm;                                  // (nothing follows: the requests of the last steps re-read this pass's operands)
      }
    }
    const int fm = min(kFamMembers, p1 - pm);
    __syncthreads();                               // (the previous pass has left the LDS)
#pragma unroll
    for (int i = 0; i < mc / 32; ++i) Hm[(hc + 32 * i) * 16 + hn] = (hn < 3 * fm) ? hv[i] : 0.0;
    fv4d acc[CH];                                  // rows role: block I; columns role: block J
#pragma unroll
    for (int b = 0; b < CH; ++b) acc[b] = (fv4d){0.0, 0.0, 0.0, 0.0};
    // the tiles in storage order, fully unrolled: block row I, block column J and the ring slot of a step are compile-time values
#pragma unroll
    for (int I = 0; I < CH; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J) {
        const int s = I * (I + 1) / 2 + J, u = s % D;
        double* buf = tb + (size_t)(s & 1) * kFamTile;
#pragma unroll
        for (int x = 0; x < 4; ++x) *reinterpret_cast<fv2d*>(buf + 8 * x * kFamPS + st_dst) = R[u][x];
        if (s == NT - D) {                         // H of the next pass, ahead of its first tile (the last pass re-reads its own)
          const int nfm = min(kFamMembers, np1 - npm);
          const double* hp = H + ((size_t)(npm + min(hj, nfm - 1)) * 3 + min(hk, 2)) * ldh + fa.h_off + hc;
#pragma unroll
          for (int i = 0; i < mc / 32; ++i) hv[i] = hp[32 * i];
        }
        {
          const double* src = (s + D < NT ? Tm + (size_t)(s + D) * 4096 : Tn + (size_t)(s + D - NT) * 4096) + st_src;
#pragma unroll
          for (int x = 0; x < 4; ++x) R[u][x] = *reinterpret_cast<const fv2d*>(src + 1024 * x);
        }
        __syncthreads();
        asm volatile("" ::: "memory");             // (the operand reads below belong to THIS step: no reuse of an earlier step's)
        const double* ap = buf + a_off;
        if (rows_role) {
          const double* bp = Hm + (size_t)(64 * J + g) * 16 + r16;
#pragma unroll
          for (int kg = 0; kg < 16; ++kg)
            acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[(size_t)kg * a_step], bp[(size_t)(4 * kg) * 16], acc[I], 0, 0, 0);
        } else if (I != J) {
          const double* bp = Hm + (size_t)(64 * I + g) * 16 + r16;
#pragma unroll
          for (int kg = 0; kg < 16; ++kg)
            acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[(size_t)kg * a_step], bp[(size_t)(4 * kg) * 16], acc[J], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
      }
    // the columns role hands its sums over in LDS, the rows role adds its own and stores: lane = (member column n = lane & 15,
    // rows (lane >> 4) + 4 reg); slot (block b, row tile)
    __syncthreads();
    double* xch = fsm;                             // [CH][4][64][4]: 64 KB at CH = 8, over H and the tile buffers
    if (!rows_role) {
#pragma unroll
      for (int b = 0; b < CH; ++b) *reinterpret_cast<fv4d*>(xch + ((size_t)(b * 4 + wq) * 64 + lane) * 4) = acc[b];
    }
    __syncthreads();
    if (rows_role) {
      const int j = r16 / 3, k = r16 - 3 * j;
#pragma unroll
      for (int b = 0; b < CH; ++b) {
        const fv4d o = *reinterpret_cast<const fv4d*>(xch + ((size_t)(b * 4 + wq) * 64 + lane) * 4);
        if (r16 < 3 * fm) {
          double* dst = PHt + ((size_t)(pm + j) * 3 + k) * mc + 64 * b + 16 * wq + g;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) dst[4 * rr] = acc[b][rr] + o[rr];
        }
      }
    }
    if (!more) break;
    fam = nfam;
    pm = npm;
    p1 = np1;
    Tm = Tn;
  }
}

hipError_t launch_family_pht(int CH, const FamilyArgs& fa, int max_families, hipStream_t s) {
  if (CH != 8 && CH != 4) return hipErrorInvalidValue;
  if (max_families <= 0) return hipSuccess;
  const size_t lds = ((size_t)64 * CH * 16 + (size_t)2 * kFamTile) * sizeof(double);
  int dev = 0, cus = 0;
  if (hipError_t e = hipGetDevice(&dev)) return e;
  if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) return e;
  const int grid = std::min(max_families, std::max(cus, 1));  // one persistent workgroup per CU (the LDS admits no second)
  static std::atomic<uint64_t> a8{0}, a4{0};
  if (CH == 8) {
    if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&family_pht_kernel<8, kFamDepth>), 160 * 1024, a8)) return e;
    hipLaunchKernelGGL((family_pht_kernel<8, kFamDepth>), dim3(grid), dim3(kFamThreads), lds, s, fa);
  } else {
    if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&family_pht_kernel<4, 2>), 160 * 1024, a4)) return e;
    hipLaunchKernelGGL((family_pht_kernel<4, 2>), dim3(grid), dim3(kFamThreads), lds, s, fa);
  }
  return hipGetLastError();
}

// ---- the family table of a step: runs of equal stored matrix in the processing order ----------------------------------------
// pre_i: the step's descriptors (propagate_kernel), [3] = entry of the stored matrix.  The order is sorted by that entry when the
// step has one (launch_order / the fused resample kernel); without it every run of equal neighbours is still a valid family.
constexpr int kFamIndexThreads = 1024;
// pass 1: heads per block of 1024 positions; pass 2: every block adds the counts of the blocks before it, ranks its own heads (wave
// ballots + a scan of the sixteen wave totals) and writes the table
__global__ __launch_bounds__(kFamIndexThreads) void family_count_kernel(int N, const int* __restrict__ pre_i, int* __restrict__ blk_count) {
  __shared__ int wsum[kFamIndexThreads / 64];
  const int p = blockIdx.x * kFamIndexThreads + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool head = p < N && (p == 0 || pre_i[(size_t)p * kPreInts + 3] != pre_i[(size_t)(p - 1) * kPreInts + 3]);
  const unsigned long long m = __ballot(head);
  if (lane == 0) wsum[wv] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int w = 0; w < kFamIndexThreads / 64; ++w) c += wsum[w];
    blk_count[blockIdx.x] = c;
  }
}

__global__ __launch_bounds__(kFamIndexThreads) void family_index_kernel(int N, const int* __restrict__ pre_i, const int* __restrict__ blk_count,
                                                                         int* __restrict__ fam_start, int* __restrict__ fam_base,
                                                                         int* __restrict__ n_fam) {
  __shared__ int wsum[kFamIndexThreads / 64];
  __shared__ int red[kFamIndexThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // families in the blocks before this one
  int before = 0;
  for (int b = tid; b < (int)blockIdx.x; b += kFamIndexThreads) before += blk_count[b];
  before = (int)wave_sum((double)before);                          // (< 2^24 families: exact in a double)
  if (lane == 0) red[wv] = before;
  const int p = blockIdx.x * kFamIndexThreads + tid;
  const int base = p < N ? pre_i[(size_t)p * kPreInts + 3] : -1;
  const bool head = p < N && (p == 0 || base != pre_i[(size_t)(p - 1) * kPreInts + 3]);
  const unsigned long long m = __ballot(head);
  if (lane == 0) wsum[wv] = __popcll(m);
  __syncthreads();
  int off = 0, mine = 0;
  for (int w = 0; w < kFamIndexThreads / 64; ++w) { off += red[w]; if (w < wv) mine += wsum[w]; }
  const int rank = off + mine + __popcll(m & ((1ull << lane) - 1ull));
  if (head) { fam_start[rank] = p; fam_base[rank] = base; }
  if (blockIdx.x == gridDim.x - 1 && tid == kFamIndexThreads - 1) {
    int total = off;
    for (int w = 0; w < kFamIndexThreads / 64; ++w) total += wsum[w];
    fam_start[total] = N;
    *n_fam = total;
  }
}

// ---- measModel of every processing position (the step kernel's phases A-C on their own) ----------------------------------------
template <int D>
__global__ __launch_bounds__(kThreads) void family_prepare_kernel(const StepArgs a) {
  extern __shared__ double psm[];
  const ModelDev& M = a.mdl;
  double* misc = psm;                                  // xn_new[8], Rnb[9] of this position
  double* tabS = psm + 32;
  double* tabC = tabS + (M.ktot > 0 ? M.ktot : 1);
  const int pos = blockIdx.x, tid = threadIdx.x, n = a.lay.n, ldx = a.lay.ldx;
  if (tid < kPreDoubles) misc[tid] = a.pre_d[(size_t)pos * kPreDoubles + tid];
  __syncthreads();
  for (int q = tid; q < M.ktot; q += kThreads) basis_table_entry(M, q, misc, tabS, tabC);
  __syncthreads();
  for (int c = tid; c < n; c += kThreads) {
    double h[D];
    H_column<D>(M, c, tabS, tabC, &misc[8], h);
#pragma unroll
    for (int k = 0; k < D; ++k) a.fam_H[((size_t)pos * D + k) * ldx + c] = h[k];
  }
}

// read-only step of the filter through the family product (see the head of this file); a: the step's arguments with the family
// workspace set (fam_H, fam_PHt, fam_idx).  The caller (launch_step_sym) has checked the configuration.
hipError_t launch_family_products(const StepArgs& a, hipStream_t s) {
  const int N = a.N, CH = a.lay.CH64;
  int* fam_start = a.fam_idx;
  int* fam_base = a.fam_idx + (size_t)N + 1;
  int* n_fam = a.fam_idx + 2 * (size_t)N + 1;
  int* blk_count = a.fam_idx + 2 * (size_t)N + 2;                  // [ceil(N / 1024)]
  const int nblk = (N + kFamIndexThreads - 1) / kFamIndexThreads;
  hipLaunchKernelGGL(family_count_kernel, dim3(nblk), dim3(kFamIndexThreads), 0, s, N, a.pre_i, blk_count);
  if (hipError_t e = hipGetLastError()) return e;
  hipLaunchKernelGGL(family_index_kernel, dim3(nblk), dim3(kFamIndexThreads), 0, s, N, a.pre_i, blk_count, fam_start, fam_base, n_fam);
  if (hipError_t e = hipGetLastError()) return e;
  const size_t lds = (size_t)(32 + 2 * (a.mdl.ktot > 0 ? a.mdl.ktot : 1)) * sizeof(double);
  hipLaunchKernelGGL(family_prepare_kernel<3>, dim3(N), dim3(kThreads), lds, s, a);
  if (hipError_t e = hipGetLastError()) return e;
  FamilyArgs fa;
  fa.T = a.Pt_old; fa.t_stride = a.Pt_old_stride;
  fa.rec = a.rec; fa.rec_stride = a.rec_stride; fa.n_bank_local = a.n_bank_local;
  fa.H = a.fam_H; fa.ldh = (size_t)a.lay.ldx; fa.h_off = a.lay.nb;
  fa.fam_start = fam_start; fa.fam_base = fam_base; fa.n_fam = n_fam;
  fa.PHt = a.fam_PHt;
  return launch_family_pht(CH, fa, N, s);
}

}  // namespace rbpf

