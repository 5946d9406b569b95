// Device-side helper math for the RBPF kernels (gfx950).  Each helper cites the reference
// function whose arithmetic it follows (paths relative to /root/reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rbpf {

#ifndef RBPF_PI
#define RBPF_PI 3.14159265358979323846
#endif

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al., SC'11).  counter = (slot, step, lane, iter)
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// two uniforms in (0,1) with 53 random bits each
__host__ __device__ inline void philox_uniform2(unsigned long long seed, uint32_t slot, uint32_t step,
                                                uint32_t lane, uint32_t iter, double& u0, double& u1) {
  uint32_t c[4] = {slot, step, lane, iter};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint64_t a = (((uint64_t)c[0] << 32) | c[1]) >> 11;
  const uint64_t b = (((uint64_t)c[2] << 32) | c[3]) >> 11;
  u0 = ((double)a + 0.5) * (1.0 / 9007199254740992.0);
  u1 = ((double)b + 0.5) * (1.0 / 9007199254740992.0);
}

// lane 0: the uniform that replaces `rand` in tools/sample.m:31
__device__ inline double philox_resample_uniform(unsigned long long seed, int slot, int step, int iter) {
  double u0, u1;
  philox_uniform2(seed, (uint32_t)slot, (uint32_t)step, 0u, (uint32_t)iter, u0, u1);
  return u0;
}

// lanes 1..: Box-Muller pairs that replace `randn` inside the dynModel closures
__device__ inline void philox_normals(unsigned long long seed, int slot, int step, int iter, int nw, double* z) {
  for (int j = 0; j < nw; j += 2) {
    double u0, u1;
    philox_uniform2(seed, (uint32_t)slot, (uint32_t)step, (uint32_t)(1 + j / 2), (uint32_t)iter, u0, u1);
    const double r = sqrt(-2.0 * log(u0));
    double s, c;
    sincos(2.0 * RBPF_PI * u1, &s, &c);
    z[j] = r * c;
    if (j + 1 < nw) z[j + 1] = r * s;
  }
}

// ---------------------------------------------------------------------------------------------
// quaternion algebra  (tools/qLeft.m, expq.m, logq.m, qInv.m, quat2rmat.m)
// ---------------------------------------------------------------------------------------------
// r = qLeft(q) * p  -- tools/qLeft.m:30-35 as a 4x4 mat-vec, summed left to right
__device__ inline void qleft_mul(const double q[4], const double p[4], double r[4]) {
  r[0] = q[0] * p[0] + (-q[1]) * p[1] + (-q[2]) * p[2] + (-q[3]) * p[3];
  r[1] = q[1] * p[0] + q[0] * p[1] + (-q[3]) * p[2] + q[2] * p[3];
  r[2] = q[2] * p[0] + q[3] * p[1] + q[0] * p[2] + (-q[1]) * p[3];
  r[3] = q[3] * p[0] + (-q[2]) * p[1] + q[1] * p[2] + q[0] * p[3];
}

// tools/expq.m:22-31 (vector branch): [cos|phi| ; phi/|phi| sin|phi|], flipped when eq(1) < 0
__device__ inline void expq_dev(const double phi[3], double eq[4]) {
  const double mag = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
  const double den = mag + (mag == 0.0 ? 1.0 : 0.0);
  double s, c;
  sincos(mag, &s, &c);
  eq[0] = c;
  eq[1] = phi[0] / den * s;
  eq[2] = phi[1] / den * s;
  eq[3] = phi[2] / den * s;
  if (eq[0] < 0.0) { eq[0] = -eq[0]; eq[1] = -eq[1]; eq[2] = -eq[2]; eq[3] = -eq[3]; }
}

// tools/logq.m:25-31 (vector branch); q0 > 1 by rounding is clamped (MATLAB acos would go complex)
__device__ inline void logq_dev(const double qin[4], double lq[3]) {
  double q[4] = {qin[0], qin[1], qin[2], qin[3]};
  if (q[0] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  const double na = acos(fmin(q[0], 1.0));
  const double den = sin(na) + (na == 0.0 ? 1.0 : 0.0);
  lq[0] = na * q[1] / den;
  lq[1] = na * q[2] / den;
  lq[2] = na * q[3] / den;
}

// tools/expq.m:33-37 (batched branch, one row): same value, but the sign flip also fires for q0 == 0 (quirk Q7)
__device__ inline void expq_batched_dev(const double phi[3], double eq[4]) {
  const double mag = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
  const double den = mag + (mag == 0.0 ? 1.0 : 0.0);
  double s, c;
  sincos(mag, &s, &c);
  eq[0] = c;
  eq[1] = phi[0] / den * s;
  eq[2] = phi[1] / den * s;
  eq[3] = phi[2] / den * s;
  if (eq[0] <= 0.0) { eq[0] = -eq[0]; eq[1] = -eq[1]; eq[2] = -eq[2]; eq[3] = -eq[3]; }
}

// tools/logq.m:32-35 (batched branch, one row): flip when q0 <= 0; na .* q(2:4) ./ (sin(na) + (na == 0))
__device__ inline void logq_batched_dev(const double qin[4], double lq[3]) {
  double q[4] = {qin[0], qin[1], qin[2], qin[3]};
  if (q[0] <= 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  const double na = acos(fmin(q[0], 1.0));
  const double den = sin(na) + (na == 0.0 ? 1.0 : 0.0);
  lq[0] = na * q[1] / den;
  lq[1] = na * q[2] / den;
  lq[2] = na * q[3] / den;
}

// tools/mcross.m:33-42: [0 -v3 v2; v3 0 -v1; -v2 v1 0], column-major 3 x 3
__device__ inline void mcross_dev(const double v[3], double M[9]) {
  M[0] = 0.0;   M[3] = -v[2]; M[6] = v[1];
  M[1] = v[2];  M[4] = 0.0;   M[7] = -v[0];
  M[2] = -v[1]; M[5] = v[0];  M[8] = 0.0;
}

// tools/qLeft.m:30-40: [q0 -qv'; qv q0*I + [qv x]], column-major 4 x 4
__device__ inline void qleft_mat_dev(const double q[4], double M[16]) {
  double X[9];
  mcross_dev(&q[1], X);
  M[0] = q[0];
  for (int r = 0; r < 3; ++r) { M[1 + r] = q[1 + r]; M[4 * (1 + r)] = -q[1 + r]; }
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) M[(1 + r) + 4 * (1 + c)] = q[0] * (r == c ? 1.0 : 0.0) + X[r + 3 * c];
}

// tools/qRight.m:29-39: [q0 -qv'; qv q0*I - [qv x]], column-major 4 x 4
__device__ inline void qright_mat_dev(const double q[4], double M[16]) {
  double X[9];
  mcross_dev(&q[1], X);
  M[0] = q[0];
  for (int r = 0; r < 3; ++r) { M[1 + r] = q[1 + r]; M[4 * (1 + r)] = -q[1 + r]; }
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) M[(1 + r) + 4 * (1 + c)] = q[0] * (r == c ? 1.0 : 0.0) - X[r + 3 * c];
}

// tools/qInv.m:27-31
__device__ inline void qinv_dev(const double q[4], double r[4]) { r[0] = q[0]; r[1] = -q[1]; r[2] = -q[2]; r[3] = -q[3]; }

// tools/quat2rmat.m:27-33 ; Rm[row*3+col]
__device__ inline void quat2rmat_dev(const double q[4], double Rm[9]) {
  const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  Rm[0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3; Rm[1] = 2 * q1 * q2 - 2 * q0 * q3; Rm[2] = 2 * q1 * q3 + 2 * q0 * q2;
  Rm[3] = 2 * q1 * q2 + 2 * q0 * q3; Rm[4] = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3; Rm[5] = 2 * q2 * q3 - 2 * q0 * q1;
  Rm[6] = 2 * q1 * q3 - 2 * q0 * q2; Rm[7] = 2 * q2 * q3 + 2 * q0 * q1; Rm[8] = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
}

// ---------------------------------------------------------------------------------------------
// dynModel closures
// ---------------------------------------------------------------------------------------------
// examples/slam-dense-mag/run_dense3D_magfield.m:301-308.  cholQ: 6x6 column-major holding
// chol(dt*Q(1:3,1:3),'lower') and chol(dt*Q(4:6,4:6),'lower') on its diagonal blocks.
__device__ inline void dyn_model_mag(const double x[7], const double* odo, const double* cholQ, const double z[6],
                                     double xp[7]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double s = 0.0;
    for (int c = 0; c <= r; ++c) s += cholQ[r + 6 * c] * z[c];
    xp[r] = x[r] + odo[r] + s;                                                  // :304
  }
  double phi[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double s = 0.0;
    for (int c = 0; c <= r; ++c) s += cholQ[(3 + r) + 6 * (3 + c)] * z[3 + c];
    phi[r] = s;
  }
  double eq[4], dq[4];
  expq_dev(phi, eq);
  const double oq[4] = {odo[3], odo[4], odo[5], odo[6]};
  qleft_mul(oq, eq, dq);                                                        // :305
  qleft_mul(&x[3], dq, &xp[3]);                                                 // :306
}

// examples/slam-dense-radio/run_dense2D_withHeading.m:75-76
__device__ inline void dyn_model_radio(const double x[3], const double* odo, const double* cholQ, const double z[1],
                                       double xp[3]) {
  double s, c;
  sincos(x[2], &s, &c);
  // [c -s; s c]' * dx(1:2)'
  xp[0] = x[0] + (c * odo[0] + s * odo[1]);
  xp[1] = x[1] + ((-s) * odo[0] + c * odo[1]);
  xp[2] = x[2] + odo[2] + cholQ[0] * z[0];
}

// ---------------------------------------------------------------------------------------------
// small dense Cholesky (lower), forward / backward substitution on D x D, column-major
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ inline bool chol_lower_small(const double* S, double* Lc) {
  // Lc column-major D x D, lower.  Returns false if a pivot is <= 0 or NaN (MATLAB chol flag>0).
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double s = S[j + D * j];
    for (int k = 0; k < j; ++k) s -= Lc[j + D * k] * Lc[j + D * k];
    if (!(s > 0.0)) return false;
    const double ljj = sqrt(s);
    Lc[j + D * j] = ljj;
    for (int i = j + 1; i < D; ++i) {
      double v = S[i + D * j];
      for (int k = 0; k < j; ++k) v -= Lc[i + D * k] * Lc[j + D * k];
      Lc[i + D * j] = v / ljj;
    }
    for (int i = 0; i < j; ++i) Lc[i + D * j] = 0.0;
  }
  return true;
}

template <int D>
__device__ inline void fwd_subst(const double* Lc, const double* b, double* x) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= Lc[i + D * k] * x[k];
    x[i] = s / Lc[i + D * i];
  }
}

template <int D>
__device__ inline void bwd_subst_T(const double* Lc, const double* b, double* x) {
  // solve Lc' x = b
#pragma unroll
  for (int i = D - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < D; ++k) s -= Lc[k + D * i] * x[k];
    x[i] = s / Lc[i + D * i];
  }
}

// wave64 sum of a double (deterministic butterfly)
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace rbpf
