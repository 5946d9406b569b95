// Internal declarations shared by the kernel TU and the C-ABI TU.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <atomic>

namespace rbpf {

// hipFuncAttributeMaxDynamicSharedMemorySize applies to the CURRENT device only: the opt-in is made once per (kernel, device) --
// `done` is the launcher's own bit set, one bit per device ordinal.  (The in-library multi-device driver launches the same kernels
// from one thread per GPU: a process-wide flag would opt in the first device only.)
inline hipError_t lds_opt_in(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (dev < 64 && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  if (dev < 64) done.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}

// Tuning overrides read from the environment exist only in diagnostic builds (-DRBPF_TUNING, tools/tune_variants.py);
// the product library never consults the environment.
#ifdef RBPF_TUNING
inline const char* tuning_env(const char* key) { return getenv(key); }
#else
inline const char* tuning_env(const char*) { return nullptr; }
#endif

// Workgroup -> processing position of the step kernels.  The hardware deals consecutive workgroups round-robin to the 8 XCDs, each
// with its own L2; the positions are sorted by the stored covariance ("base") a particle reads, and between two flushes of the lazy
// update 40-70 % of the particles share theirs with a sibling or cousin (tools/base_sharing_stats.py).  With position = workgroup
// the family is spread over eight L2s and every member fetches the matrix from memory; dealing each XCD a CONTIGUOUS range of
// positions puts a family on one XCD, back to back in its dispatch order, where the followers stream the leader's lines out of
// the L2.  Bijective for any grid size.
#ifndef RBPF_XCD_CHUNKS
#define RBPF_XCD_CHUNKS 8
#endif
__device__ __forceinline__ int xcd_position(int wg, int grid) {
  if (RBPF_XCD_CHUNKS <= 1) return wg;
  const int x = wg % RBPF_XCD_CHUNKS, q = grid / RBPF_XCD_CHUNKS, r = grid % RBPF_XCD_CHUNKS;
  return x * q + (x < r ? x : r) + wg / RBPF_XCD_CHUNKS;
}

constexpr int kThreads = 256;      // stream-kernel workgroup: 4 wave64
constexpr int kWaves = kThreads / 64;
constexpr int kChunkRows = 128;    // rows covered by one wave-wide 16-B-per-lane load
constexpr int kPreInts = 16;        // per-workgroup descriptor written by propagate_kernel: slot, anc, ancb, base, destination slot
                                    // of the stored matrix, flush phase, set idx[kMaxSets], 2 x pad   (r02 layout: set idx[4],
                                    // destination slot of the stored matrix, flush phase, 2 x pad
constexpr int kPreDoubles = 17;    // ... and xn_new[8], Rnb[9]
constexpr int kPreSet0 = 6;        // first set index of the descriptor; [4] destination slot, [5] flush phase
constexpr int kMaxSets = 8;        // pending rank-d factor sets a step kernel can carry (full storage: 4; symmetric storage: 8)
constexpr int kMaxSetsFull = 4;    // ... of the full-square step kernel (rbpf_kernels.hip)

// Model constants resident in kernel arguments (scalar registers).
struct ModelDev {
  int kind, m, dim;
  int nN, n, d, nw, nodo;
  const int* NN;       // device, [dim][m] (axis-major)
  double L[3];
  int kmax[3];         // largest index per axis
  int ktot;            // kmax[0]+kmax[1]+kmax[2]
  double R[9];         // d x d column-major
  double jitter;
  double logconst;     // -0.5 * d * log(2*pi)
  int use_dyn_res_norm;
  double Rinv[9];      // inv(R), d x d column-major
  double halfLogDetR;  // 0.5*log(det(R))
  double cam[3];       // sparse-visual family: f, fp, fw
  const double* Rdev;  // sparse-visual family: full R [d x d] in device memory (d may exceed 3)
};

// HBM layout of one particle's covariance ("bank" entry).  Natural state order is kept for the
// index space; rows are split so that the streamed part is perfectly aligned:
//   border rows  r in [0, nb)  -> block B, row-major  B[r*ldb + c],          c in [0,n)
//   core rows    r in [nb, n)  -> block T, column-major T[c*mc + (r - nb)],  c in [0,n)
// with mc = 128*floor(n/128), nb = n - mc, ldb = n rounded up to even.
//
// Symmetric storage (rbpf_options.storage = 2, `sym` != 0): the covariance is symmetric (particleFilter.m:198 keeps it so up to
// rounding), so block T holds only the lower block triangle of the core x core part, in 64 x 64 tiles (I >= J) of 32 column pairs:
//   element (r, c), core coordinates, tile (I, J) = (r / 64, c / 64):  T[(I (I + 1) / 2 + J) * 4096 + ((c % 64) / 2) * 128 + (r % 64) * 2 + c % 2]
// i.e. one wave-wide 16-byte-per-lane load brings rows 64 I .. 64 I + 63 of the two columns of a pair.  Diagonal tiles are stored
// in full (both triangles).  The border rows stay in block B as above (they hold the border columns too, by symmetry), so the
// core rows x border columns part of T disappears.  0.5625 n^2 elements at nLin = 515 instead of n^2.
constexpr int kSymChunk = 64;                         // rows / columns of a tile
constexpr int kSymTile = kSymChunk * kSymChunk;       // elements of a tile
struct Layout {
  int n, nb, mc, ldb, ldx;   // ldx: padded length of per-particle vectors (xl, K, KS)
  int CH;                    // mc / 128
  int RS, CS, CPL;           // wave decomposition of the core stream (rows x column phases)
  int sym, CH64;             // symmetric storage (1: fp64 tiles, column PAIRS; 2: fp32 tiles, column QUADS -- 16 bytes per lane either way): mc / 64 tile rows
  size_t szT, szB;           // elements per particle
};

// offset of core element (r, c) (core coordinates, r, c in [0, mc)) inside block T of the symmetric layout; (r, c) above the
// block diagonal is read from its mirror image
// (cg: columns a lane's 16-byte load covers -- 2 in fp64 tiles, 4 in fp32 tiles: Layout::sym = 1 / 2)
__host__ __device__ inline size_t sym_t_index(int r, int c, int cg = 2) {
  int I = r / kSymChunk, J = c / kSymChunk;
  if (J > I) { const int t = r; r = c; c = t; I = r / kSymChunk; J = c / kSymChunk; }
  return ((size_t)I * (I + 1) / 2 + J) * kSymTile + (size_t)((c % kSymChunk) / cg) * (cg * kSymChunk) + (size_t)(r % kSymChunk) * cg + (c % cg);
}


__host__ __device__ inline int sym_cg(const Layout& L) { return L.sym == 2 ? 4 : 2; }

struct StepArgs {
  ModelDev mdl;
  Layout lay;
  int N, t, propagate;
  const int* ai;                 // ancestors of this step (null: identity); index into xn_old
  const int* ai_bank;            // ancestor index in the bank address space (null: same as ai)
  const int* order;              // processing order: workgroup b handles slot order[b] (null: b)
  // multi-step lazy update.  Pending set s of particle j is entry fset_idx_old[s][j] of bank fset[s]; when
  // fset[s] is null the legacy single set (F_old, addressed like the map bank) is used.
  int n_sets;                    // pending sets applied on the fly (0 only at t = 0)
  int write_base;                // 1: store the downdated matrix into the child's slot (flush); 0: light step
  const double* fset[kMaxSets];
  const int* fset_idx_old[kMaxSets];
  int* fset_idx_new[kMaxSets];   // propagated entries of the surviving sets (light steps)
  int* fself_idx_new;            // index table of the set this step produces: [i] = i
  const int* base_old; int* base_new;   // slot of the stored matrix of each particle's lineage
  int share_flush;                      // shared flush: a read-only workgroup's stored matrix becomes its family writer's new entry (descriptor [4])
  // timed launches: how many DISTINCT stored matrices does this step read?  (mark[slot] <- tag; a changed mark counts once)
  int* distinct_mark; unsigned long long* distinct_counter; int distinct_tag;
  // single-bank ("in place") flush: the rewritten matrix of slot i goes to bank entry dst_slot[i] (null: i) and the
  // launch only processes the slots whose phase_of[i] equals `phase` (phase < 0: all).  Phase 0 = children that move
  // to a dead entry, phase 1 = the first child of every stored matrix, which overwrites it after its siblings read it.
  const int* dst_slot; const int* phase_of; int phase;
  int fp32;                      // 1: the covariance banks (Pt / Pb) hold float; strides stay in elements
  double* strip_ws; size_t strip_ws_stride;   // symmetric storage at sixteen tile rows: column-strip workspace, [workgroup][sym_strip_doubles]
  // generic model family (arbitrary host callbacks): the propagated states and the measurement Jacobians of this step
  // were evaluated on the host and uploaded
  const double* xn_ext;          // SoA [nN][N] new non-linear states (null: dynModel runs on the device)
  const double* H_ext;           // [N][d][ldx] measurement Jacobian H_i of every slot (null: measModel on the device)
  const int* slot_ids;           // logical (global) id of each local slot (null: slot_offset + i); when set,
                                 // `ai` is indexed by that logical id
  // remote ancestors (sharded filter): bank index >= n_bank_local refers to record (index - n_bank_local) of
  // `rec`, a particle-major buffer [Pt | Pb | F | xl] per record
  int n_bank_local; const double* rec; size_t rec_stride, rec_off_B, rec_off_F, rec_off_X;
  size_t rec_off_I, rec_off_hld;            // information form: ivec [ldx] and halfLogDetP of a record (sharded smoother)
  int zero_set_idx;              // entry of every factor-set bank that holds zeros (fresh lineages)
  int slot_offset;               // global id of local slot 0 (RNG counters / replay rows)
  size_t xn_old_stride, xn_new_stride;      // component stride of the SoA state arrays
  const double* xn_old; double* xn_new;     // SoA [nN][stride]
  const double* xl_old; double* xl_new;     // [N][ldx]
  size_t xl_old_stride;                     // ldx, or 0 to broadcast x0_lin
  const double* F_old; double* F_new;       // pending rank-d factors [N][2][d][ldx]: KS then K
  const double* Pt_old; double* Pt_new;
  const double* Pb_old; double* Pb_new;
  size_t Pt_old_stride, Pb_old_stride;      // szT / szB, or 0 to broadcast P0
  double* logw;                             // [N]
  int rng_mode; int k_iter;
  const double* Z;                          // replay normals of this step [N][nw]
  unsigned long long seed;
  const double* odo;                        // [nodo]
  const double* cholQ;                      // [nw*nw] lower factor(s) used by dynModel
  const double* y;                          // [d]
  const double* xref;                       // CPF-AS: state of the reference slot at this step (or null)
  int xref_gslot;                           // logical id of the reference slot (N_P - 1 of the global filter)
  int* status;
  unsigned long long* stamps;               // diagnostic builds only (RBPF_STAMPS)
  int* pre_i; double* pre_d;                // [N][kPreInts], [N][kPreDoubles] descriptors (propagate_kernel -> step_kernel)
  double* u_next;                           // [N] out: Philox uniforms of the next step's resampling (or null)
  // information form (particleSmootherInformationForm.m): extra per-particle state
  int info;
  const double* ivec_old; size_t ivec_old_stride; double* ivec_new;   // [N][ldx]
  const double* hld_old; size_t hld_old_stride; double* hld_new;      // halfLogDetP [N]
  double* qf_new;                                                     // ivec'*P*ivec after the update [N]
  double* Hb_new;                                                     // [N][d][ldx] H_i of this step
};

struct NormArgs {
  int N, nN, t;
  const double* logw;
  double* w;            // normalised weights out
  double* wc;           // running sum out
  const double* xn;     // SoA [nN][N] of this step
  double* traj_max;     // [nN] column t (or null)
  double* traj_mean;    // [nN] column t (or null)
  int* iw_max;          // device scalar
  double* lse_out;      // device scalar (or null)
  int parallel_scan = 0; // 1: wc holds a parallel prefix sum (searches must run in `approx` mode + fixup)
};

struct SearchArgs {
  int N, n_draw, t;
  int slot0 = 0;        // first slot drawn: slot = slot0 + i (uniform and output index)
  int u_is_scalar = 0;  // replay: every draw uses U[0] (the `ak = sample(w)` draw)
  const double* wc;
  int rng_mode; int k_iter;
  const double* U;      // replay uniforms of this step [N]
  unsigned long long seed;
  int* ai;              // out [N]
  int* overflow;        // device counter of clamped draws (u > wc(end))
  int scan_depth = 0;   // extra summation depth of the parallel prefix beyond the element index (0: N/1024 + 32)
  int approx = 0;       // 1: wc is a parallel prefix; flag draws that fall within the rounding bound of an edge
  int* ambiguous = nullptr;   // device counter of such draws (resolved exactly by launch_resample_fixup)
  const double* w = nullptr;  // weights (fixup only)
  double* wc_exact = nullptr; // scratch for the strict cumsum (fixup only; may alias wc)
};

size_t step_lds_bytes(const ModelDev& m, const Layout& lay, int extra = 0, int n_sets = 1);
bool step_use_blocked(const ModelDev& m, const Layout& lay, int extra, int n_sets);
Layout make_layout(int n, int d);
Layout make_layout_low_regs(int n, int d);
Layout make_layout_sym(int n, int d, int fp32 = 0);   // symmetric storage (see Layout); sym_supported: the sizes the step kernel takes
bool sym_supported(int n, int d);
size_t sym_strip_doubles(const Layout& lay, int d);   // per workgroup, 0 unless sixteen tile rows
size_t step_sym_lds_bytes(const ModelDev& m, const Layout& lay, int n_sets, int write_base, int extra = 0);   // extra = 1: information form
hipError_t launch_step_sym(const StepArgs& a, hipStream_t s);
// wave-level reduction primitives of the symmetric step kernel on their own (tests): in [4][64] -> out [4] lane sums
hipError_t launch_probe_wave_reduce(const double* in, double* out, hipStream_t s);

// dynModel for all slots (must run before launch_step of the same StepArgs)
hipError_t launch_propagate(const StepArgs& a, hipStream_t s);
hipError_t launch_step(const StepArgs& a, hipStream_t s);
hipError_t launch_normalise_scan(const NormArgs& a, hipStream_t s);
hipError_t launch_search(const SearchArgs& a, hipStream_t s);
hipError_t launch_cumsum(int N, const double* w, double* wc, hipStream_t s);
// normalise step t and, fused, draw the ancestors of step t+1 (+ their ancestor-sorted processing order)
hipError_t launch_normalise_resample(const NormArgs& a, const SearchArgs& sa, int* order, int* counts, hipStream_t s,
                                     const int* remap = nullptr);
hipError_t launch_order_large(int n_slots, int range, const int* key, const int* remap, int* order, int* counts, hipStream_t s);
// counting sort of the slots by remap[key[i]] (remap null: key[i]); range: number of distinct key values
hipError_t launch_order(int n_slots, int range, const int* key, int* order, int* counts, hipStream_t s, const int* remap = nullptr);
// plan of a single-bank flush: order [N] lists the slots sorted by the entry base[ai[.]] of their ancestor's stored
// matrix.  dst [N] / phase [N] out; scratch: 3 N ints.
hipError_t launch_inplace_plan(int N, const int* order, const int* ai, const int* base, int* dst, int* phase,
                               int* scratch, hipStream_t s);
hipError_t launch_share_inplace_plan(int N, const int* order, const int* ai, const int* base, int* dst, int* phase, int* scratch,
                                     unsigned long long* writers, hipStream_t s);   // scratch: 6 N ints
hipError_t launch_share_plan(int N, int nkeys, const int* ai, int* lead, int* dst, int* phase, unsigned long long* writers, hipStream_t s);
// multi-workgroup equivalent for large N (rbpf_resample.hip); sa may be null (normalise only)
size_t resample_scratch_doubles(int N);
hipError_t launch_resample_pipeline(const NormArgs& nm, const SearchArgs* sa, int* order, int* counts, const int* remap,
                                    double* scratch, hipStream_t s);
constexpr int kMaxParticles = 1024 * 1024;     // largest global particle count (rs_offsets_kernel: <= 1024 blocks of 1024)
constexpr int kSingleWgResampleMaxN = 8192;   // above this the multi-workgroup pipeline is used
// exact re-draw of every slot with the strict left-to-right cumsum if any draw was flagged ambiguous
hipError_t launch_resample_fixup(const SearchArgs& a, hipStream_t s);

// pack / unpack between MATLAB column-major n x n images and the bank layout
// fp32 != 0: the banks hold float (rbpf_options.storage = 1); pointers stay typed double* on the host side
hipError_t launch_pack_P(const Layout& lay, const double* P_colmajor, size_t src_stride, double* Pt,
                         double* Pb, int count, hipStream_t s, int fp32 = 0);
hipError_t launch_unpack_P(const Layout& lay, int d, const double* Pt, const double* Pb, const double* F,
                           const int* index, int count, double* P_colmajor, hipStream_t s, int fp32 = 0);
hipError_t launch_unpack_P_sets(const Layout& lay, int d, const double* Pt, const double* Pb, int n_sets,
                                const double* const* fset, const int* const* fidx, const int* base, const int* index,
                                int count, double* P_colmajor, hipStream_t s, int fp32 = 0);
hipError_t launch_weighted_mean_xl(int N, int n, int ldx, const double* xl, const double* w, double* out,
                                   hipStream_t s);
hipError_t launch_backtrace(int N, int nN, int T, const double* X, const int* A, const int* start_index,
                            int n_paths, double* out, hipStream_t s, int path0 = 0);
hipError_t launch_philox_fill(unsigned long long seed, int k_iter, int N, int T, int nw, double* U, double* Z,
                              double* Ufin, hipStream_t s);
// layout 0: dy[p][c][k] (MATLAB [ny x nLin x Npred]); layout 1: dy[p][k][c] (rows of H contiguous)
hipError_t launch_meas_model(const ModelDev& m, int npred, const double* xn, double* dy, hipStream_t s, int layout = 0);
hipError_t launch_dyn_model(const ModelDev& m, int np, const double* xn, const double* odo, const double* cholQ,
                            const double* z, double* xn_next, hipStream_t s);
hipError_t launch_dyn_res_norm(const ModelDev& m, int np, const double* xnk_t, const double* xn,
                               const double* odo, const double* cholQfull, double* e_dyn, hipStream_t s);
hipError_t launch_jacobian_phi3d(const ModelDev& m, int np, const double* x, const double* lo, const double* up,
                                 double* J, hipStream_t s);
hipError_t launch_quat_helpers(int op, int n, const double* in, double* out, hipStream_t s);
hipError_t launch_transpose_soa(int N, int nN, const double* soa, double* aos, hipStream_t s);
hipError_t launch_pack_records(const Layout& lay, int d, const int* idx, int count, const double* Pt, const double* Pb,
                               const double* F, const double* xl, double* rec, hipStream_t s, size_t rec_stride = 0,
                               int fp32 = 0);
hipError_t launch_pack_records_flushed(const Layout& lay, int d, const int* idx, int count, const double* Pt,
                                       const double* Pb, int n_sets, const double* const* fset, const int* const* fidx,
                                       const int* base, int n_bank_local, const double* rec, size_t rec_stride,
                                       const double* xl, double* out, hipStream_t s, int fp32 = 0);
hipError_t launch_permute_fwd(int N, int nN, int world, int Nloc, const int* phys_of_logical, const double* fwd_gather,
                              double* logw, double* xn_soa, hipStream_t s, int rows = 0, double* extra = nullptr);
hipError_t launch_gather_xl(int N, int n, int ldx, const double* xl, double* out_colmajor, hipStream_t s);

}  // namespace rbpf
