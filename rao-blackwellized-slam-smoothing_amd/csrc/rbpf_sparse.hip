// sparseFeatures = true branch of the reference (src/particleFilter.m:127-137,165-181; src/particleSmoother.m:194-217,
// 267-277,306-321) for the model family of examples/slam-sparse-visual: 2-D pose, point landmarks seen by a 1-D pinhole
// camera (examples/slam-sparse-visual/measurement.m:32-84), unobserved outputs marked NaN in y.
//
// Per particle the measurement model is linearised at the particle's own map (an EKF step): the innovation is
// y - yhat(xn_i, xl_i), the Jacobian has two non-zeros per row, and only the observed outputs enter S, the weight
// and the gain.  The problem is tiny (nLin = 40, <= 20 outputs), so one wave per particle does the whole step out of
// LDS; the covariances live in the same banks as the dense path (border-only layout, no pending factors).
#include "../../include/rbpf.h"
#include "rbpf_internal.hpp"
#include "rbpf_device.hpp"
#include "rbpf_sparse.hpp"

namespace rbpf {

// measurement.m:35-52,61,74,77 for one landmark (mx, my) seen from pose (px, py, th) -> yhat, d yhat / d (mx, my)
__device__ inline void pinhole(double f, double fp, double px, double py, double th, double mx, double my, double& yhat,
                               double& dm1, double& dm2) {
  double s, c;
  sincos(th, &s, &c);
  // K * [R' , -R'*p] with R' = [c s; -s c]
  const double t1 = -(c * px + s * py), t2 = -(-s * px + c * py);
  const double a00 = f * c + fp * (-s), a01 = f * s + fp * c, a02 = f * t1 + fp * t2;
  const double a10 = -s, a11 = c, a12 = t2;
  const double u1 = a00 * mx + a01 * my + a02;
  const double u2 = a10 * mx + a11 * my + a12;
  yhat = u1 / u2;                                                           // :52
  const double q = my * c - py * c - mx * s + px * s;
  const double div = q * q;                                                 // :61
  dm1 = (f * (my - py)) / div;                                              // :74
  dm2 = -(f * (mx - px)) / div;                                             // :77
}

// One time step for every particle: resample-gather, dynModel (pfslam.m:81), EKF weight and update.
__global__ __launch_bounds__(64) void sparse_step_kernel(const SparseStepArgs a) {
  extern __shared__ double sm[];
  const int i = blockIdx.x, tid = threadIdx.x;
  const int n = a.n, d = a.d, ldb = a.ldb;
  double* Ps = sm;                       // [n][n] row-major
  double* PH = Ps + (size_t)n * n;       // [n][d]  P * dy(ind,:)'
  double* Kg = PH + (size_t)n * d;       // [n][d]
  double* KS = Kg + (size_t)n * d;       // [n][d]  K * SS
  double* SS = KS + (size_t)n * d;       // [d][d] column-major
  double* cS = SS + (size_t)d * d;       // [d][d]
  double* xls = cS + (size_t)d * d;      // [n]
  double* yh = xls + n;                  // [d] yhat
  double* d1 = yh + d;                   // [d]
  double* d2 = d1 + d;                   // [d]
  double* ev = d2 + d;                   // [d] innovation of the observed outputs
  double* vv = ev + d;                   // [d]
  double* xs = vv + d;                   // [8] new non-linear state
  int* obs = reinterpret_cast<int*>(xs + 8);   // [d] observed outputs
  int* meta = obs + d;                   // [0] number observed, [1] chol ok

  const int anc = a.ai ? a.ai[i] : i;
  if (tid == 0) {
    double x[8], xp[8];
    for (int c = 0; c < a.nN; ++c) x[c] = a.xn_old[(size_t)c * a.xn_old_stride + anc];
    if (a.xref != nullptr && i == a.N - 1) {
      for (int c = 0; c < a.nN; ++c) xp[c] = a.xref[c];                      // particleSmoother.m:242
    } else if (a.propagate) {
      double z[8];
      if (a.rng_mode == 0) { for (int k = 0; k < a.nw; ++k) z[k] = a.Z[(size_t)i * a.nw + k]; }
      else philox_normals(a.seed, i, a.t, a.k_iter, a.nw, z);
      for (int c = 0; c < a.nN; ++c) {                                       // xn + dx' + sqrt(dt*Q)*randn  (pfslam.m:81)
        double s = 0.0;
        for (int k = 0; k < a.nw; ++k) s = fma(a.Ssqrt[c + a.nw * k], z[k], s);
        xp[c] = x[c] + a.odo[c] + s;
      }
    } else {
      for (int c = 0; c < a.nN; ++c) xp[c] = x[c];
    }
    for (int c = 0; c < a.nN; ++c) { a.xn_new[(size_t)c * a.xn_new_stride + i] = xp[c]; xs[c] = xp[c]; }
    int no = 0;
    for (int j = 0; j < d; ++j)
      if (!isnan(a.y[j])) obs[no++] = j;                                     // ind = ~isnan(yt)  (:134)
    meta[0] = no;
  }
  const double* xl_src = a.xl_old + (size_t)anc * a.xl_old_stride;
  const double* P_src = a.Pb_old + (size_t)anc * a.Pb_old_stride;
  for (int q = tid; q < n; q += 64) xls[q] = xl_src[q];
  for (int q = tid; q < n * n; q += 64) { const int r = q / n, c = q % n; Ps[q] = P_src[(size_t)r * ldb + c]; }
  __syncthreads();
  const int no = meta[0];
  for (int j = tid; j < d; j += 64)                                          // [yhat,dy] = measModel(xn_i, xl_i)  (:129)
    pinhole(a.f, a.fp, xs[0], xs[1], xs[2], xls[2 * j], xls[2 * j + 1], yh[j], d1[j], d2[j]);
  __syncthreads();
  for (int q = tid; q < no; q += 64) ev[q] = a.y[obs[q]] - yh[obs[q]];       // :131,135
  for (int q = tid; q < n * no; q += 64) {                                   // P * dy(ind,:)'
    const int r = q / no, b = q % no, j = obs[b];
    PH[r * d + b] = Ps[r * n + 2 * j] * d1[j] + Ps[r * n + 2 * j + 1] * d2[j];
  }
  __syncthreads();
  for (int q = tid; q < no * no; q += 64) {                                  // SS = dy*P*dy' + R, observed part (:132,136)
    const int aa = q % no, bb = q / no, ja = obs[aa], jb = obs[bb];
    SS[aa + d * bb] = (d1[ja] * PH[(2 * ja) * d + bb] + d2[ja] * PH[(2 * ja + 1) * d + bb]) + a.R[ja + a.d * jb];
  }
  __syncthreads();
  if (tid == 0) {
    bool ok = true;
    double lw = 0.0;
    if (no > 0) {
      for (int attempt = 0; attempt < 2; ++attempt) {                        // :145-148
        ok = true;
        const double jit = attempt ? a.jitter : 0.0;
        for (int j = 0; j < no && ok; ++j) {
          double s = SS[j + d * j] + jit;
          for (int k = 0; k < j; ++k) s -= cS[j + d * k] * cS[j + d * k];
          if (!(s > 0.0)) { ok = false; break; }
          const double ljj = sqrt(s);
          cS[j + d * j] = ljj;
          for (int r = j + 1; r < no; ++r) {
            double v = SS[r + d * j];
            for (int k = 0; k < j; ++k) v -= cS[r + d * k] * cS[j + d * k];
            cS[r + d * j] = v / ljj;
          }
        }
        if (ok) break;
      }
      if (ok) {
        double sl = 0.0, q2 = 0.0;
        for (int r = 0; r < no; ++r) {                                       // v = cS\e  (:149)
          double v = ev[r];
          for (int k = 0; k < r; ++k) v -= cS[r + d * k] * vv[k];
          vv[r] = v / cS[r + d * r];
          sl += log(cS[r + d * r]);
          q2 += vv[r] * vv[r];
        }
        lw = -sl - 0.5 * q2 - 0.5 * (double)no * 1.8378770664093453;        // :150
      } else {
        atomicOr(a.status, 1);
        lw = nan("");
      }
    }
    a.logw[i] = lw;
    meta[1] = ok ? 1 : 0;
  }
  __syncthreads();
  const bool ok = meta[1] != 0;
  // K = P*((dy(ind,:)'/cS')/cS)  (:181): row r of K solves  k * (cS*cS') = PH(r,:)
  for (int r = tid; r < n; r += 64) {
    double w[32], k[32];
    if (ok) {
      for (int b = 0; b < no; ++b) {                                         // x*cS' = PH(r,:)
        double v = PH[r * d + b];
        for (int q = 0; q < b; ++q) v -= cS[b + d * q] * w[q];
        w[b] = v / cS[b + d * b];
      }
      for (int b = no - 1; b >= 0; --b) {                                    // k*cS = x
        double v = w[b];
        for (int q = b + 1; q < no; ++q) v -= cS[q + d * b] * k[q];
        k[b] = v / cS[b + d * b];
      }
    } else {
      for (int b = 0; b < no; ++b) k[b] = 0.0;
    }
    double xn_ = xls[r];
    for (int b = 0; b < no; ++b) { Kg[r * d + b] = k[b]; xn_ = fma(k[b], ev[b], xn_); }   // xl + K*e  (:197)
    a.xl_new[(size_t)i * a.ldx + r] = xn_;
    for (int b = 0; b < no; ++b) {                                           // K*SS
      double s = 0.0;
      for (int q = 0; q < no; ++q) s = fma(k[q], SS[q + d * b], s);
      KS[r * d + b] = s;
    }
  }
  __syncthreads();
  double* P_dst = a.Pb_new + (size_t)i * a.szB;
  for (int q = tid; q < n * n; q += 64) {                                    // P - K*SS*K'  (:198)
    const int r = q / n, c = q % n;
    double s = 0.0;
    for (int b = 0; b < no; ++b) s = fma(KS[r * d + b], Kg[c * d + b], s);
    P_dst[(size_t)r * ldb + c] = Ps[q] - s;
  }
}

size_t sparse_step_lds_bytes(int n, int d) {
  return ((size_t)n * n + 3 * (size_t)n * d + 2 * (size_t)d * d + n + 6 * (size_t)d + 8) * sizeof(double) + ((size_t)d + 4) * sizeof(int);
}

hipError_t launch_sparse_step(const SparseStepArgs& a, hipStream_t s) {
  if (a.d > 32 || a.n > 96) return hipErrorInvalidValue;
  const size_t lds = sparse_step_lds_bytes(a.n, a.d);
  static std::atomic<uint64_t> attr{0};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&sparse_step_kernel), 150 * 1024, attr)) return e;
  hipLaunchKernelGGL(sparse_step_kernel, dim3(a.N), dim3(64), lds, s, a);
  return hipGetLastError();
}

// Ancestor weights of the reference trajectory, sparse branch (particleSmoother.m:194-217): for particle p the future
// observations (time ti >= t, landmark j observed at ti) are linearised at the particle's map:
//   e = [y(ti,j) - yhat(x'_ti, xl_p)],  S = dy*P_p*dy' + blkdiag(R(ind,ind))
// -> rhs [N][M], S [N][M*M] column-major for the batched Cholesky.
__global__ __launch_bounds__(256) void sparse_anc_kernel(const SparseAncArgs a) {
  extern __shared__ double sm[];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int n = a.n, M = a.M;
  double* Ps = sm;                       // [n][n]
  double* d1 = Ps + (size_t)n * n;       // [M]
  double* d2 = d1 + M;                   // [M]
  int* pj = reinterpret_cast<int*>(d2 + M);   // [M] landmark
  int* pt = pj + M;                      // [M] time
  const double* xl = a.xl + (size_t)p * a.ldx;
  const double* P_src = a.Pb + (size_t)p * a.szB;
  for (int q = tid; q < n * n; q += 256) { const int r = q / n, c = q % n; Ps[q] = P_src[(size_t)r * a.ldb + c]; }
  for (int q = tid; q < M; q += 256) {
    const int ti = a.pair_t[a.off + q], j = a.pair_j[a.off + q];
    const double* x = a.xnk + (size_t)ti * a.nN;
    double yh, m1, m2;
    pinhole(a.f, a.fp, x[0], x[1], x[2], xl[2 * j], xl[2 * j + 1], yh, m1, m2);    // measModel(xnk(:,ti), xl(:,i))  (:204)
    d1[q] = m1; d2[q] = m2; pj[q] = j; pt[q] = ti;
    a.rhs[(size_t)p * M + q] = a.y[(size_t)ti * a.d + j] - yh;                      // :207-208
  }
  __syncthreads();
  double* S = a.S + (size_t)p * M * M;
  for (size_t q = tid; q < (size_t)M * M; q += 256) {
    const int aa = (int)(q % M), bb = (int)(q / M);
    const int ja = pj[aa], jb = pj[bb];
    const double g0 = d1[aa] * Ps[(2 * ja) * n + 2 * jb] + d2[aa] * Ps[(2 * ja + 1) * n + 2 * jb];          // (dy*P)(a, 2jb)
    const double g1 = d1[aa] * Ps[(2 * ja) * n + 2 * jb + 1] + d2[aa] * Ps[(2 * ja + 1) * n + 2 * jb + 1];  // (dy*P)(a, 2jb+1)
    double v = g0 * d1[bb] + g1 * d2[bb];
    if (pt[aa] == pt[bb]) v += a.R[ja + a.d * jb];                                  // blkdiag(RS, R(ind,ind))  (:211)
    S[q] = v;                                                                       // :215
  }
}

hipError_t launch_sparse_anc(const SparseAncArgs& a, int N, hipStream_t s) {
  if (a.M <= 0) return hipSuccess;
  const size_t lds = ((size_t)a.n * a.n + 2 * (size_t)a.M) * sizeof(double) + 2 * (size_t)a.M * sizeof(int);
  static std::atomic<uint64_t> attr{0};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&sparse_anc_kernel), 150 * 1024, attr)) return e;
  hipLaunchKernelGGL(sparse_anc_kernel, dim3(N), dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace rbpf
