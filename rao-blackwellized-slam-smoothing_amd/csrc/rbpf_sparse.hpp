// sparseFeatures branch (examples/slam-sparse-visual): kernel arguments, see rbpf_sparse.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace rbpf {

struct SparseStepArgs {
  int N, t, n, d, nN, nw, ldx, ldb, propagate;
  size_t szB;
  double f, fp;                       // camera (measurement.m:46)
  const int* ai;                      // ancestors of this step (null: identity)
  const double* xn_old; size_t xn_old_stride; double* xn_new; size_t xn_new_stride;   // SoA [nN][stride]
  const double* xl_old; size_t xl_old_stride; double* xl_new;                         // [N][ldx]
  const double* Pb_old; size_t Pb_old_stride; double* Pb_new;                         // [N][n][ldb] row-major
  const double* y;                    // [d] outputs of this step, NaN = not observed
  const double* R;                    // [d x d] column-major
  const double* odo;                  // [nN]
  const double* Ssqrt;                // [nw x nw] element-wise sqrt(dt*Q)  (pfslam.m:81)
  int rng_mode, k_iter; unsigned long long seed; const double* Z;                     // replay normals [N][nw]
  const double* xref;                 // CPF-AS: state of slot N-1 (or null)
  double jitter;
  double* logw;
  int* status;
};

struct SparseAncArgs {
  int n, d, nN, ldx, ldb, M, off;
  size_t szB;
  double f, fp;
  const int* pair_t; const int* pair_j;     // observed (time, landmark) pairs of the whole run, time-major
  const double* xnk;                  // [T][nN] reference trajectory
  const double* y;                    // [T][d]
  const double* xl; const double* Pb; // particle banks (current generation)
  const double* R;                    // [d x d]
  double* rhs;                        // [N][M]
  double* S;                          // [N][M*M] column-major
};

size_t sparse_step_lds_bytes(int n, int d);
hipError_t launch_sparse_step(const SparseStepArgs& a, hipStream_t s);
hipError_t launch_sparse_anc(const SparseAncArgs& a, int N, hipStream_t s);

}  // namespace rbpf
