// Measurement-model pieces shared by the step kernels (rbpf_kernels.hip, rbpf_step_sym.hip) and the standalone helper kernels.
#pragma once
#include "rbpf_internal.hpp"
#include "rbpf_device.hpp"

namespace rbpf {

// ---------------------------------------------------------------------------------------------
// basis / measurement model pieces shared by the step kernel and the standalone test kernels
// ---------------------------------------------------------------------------------------------
// sin / cos of pi*k*(x_a+L_a)/(2 L_a) for k = 1..kmax[a]  (tools/domain_cartesian_dx.m:91,154)
__device__ inline void basis_table_entry(const ModelDev& M, int q, const double* pos, double* tabS, double* tabC) {
  int a = 0, k = q;
  if (k >= M.kmax[0]) { k -= M.kmax[0]; a = 1; if (k >= M.kmax[1]) { k -= M.kmax[1]; a = 2; } }
  const double La = M.L[a];
  const double arg = RBPF_PI * (double)(k + 1) * (pos[a] + La) / (2.0 * La);
  double s, c;
  sincos(arg, &s, &c);
  tabS[q] = s;
  tabC[q] = c;
}

// Column c of H_i (ny x nLin).
//   dense-mag   : Rnb' * [e_c] for c<3, Rnb' * [dphi_x; dphi_y; dphi_z](j=c-3) otherwise
//                 (examples/slam-dense-mag/run_dense3D_magfield.m:267-277,
//                  tools/domain_cartesian_dx.m:146-170 evaluation order kept)
//   dense-radio : phi_c(x,y) (run_dense2D_withHeading.m:168, domain_cartesian_dx.m:88-93)
template <int D>
__device__ inline void H_column(const ModelDev& M, int c, const double* tabS, const double* tabC, const double* Rm,
                                double* h) {
  if (M.kind == 1) {
    double g[3];
    if (c < 3) {
      g[0] = (c == 0); g[1] = (c == 1); g[2] = (c == 2);
    } else {
      const int j = c - 3;
      int base[3] = {0, M.kmax[0], M.kmax[0] + M.kmax[1]};
      int nn[3];
      for (int a = 0; a < 3; ++a) nn[a] = M.NN[a * M.m + j];
      for (int di = 0; di < 3; ++di) {
        double v = 1.0;
        for (int a = 0; a < 3; ++a) {
          const double La = M.L[a];
          const int q = base[a] + nn[a] - 1;
          if (a == di) v = v * RBPF_PI * (double)nn[a] / (2.0 * La * sqrt(La)) * tabC[q];
          else v = v * 1.0 / sqrt(La) * tabS[q];
        }
        g[di] = v;
      }
    }
    // (Rnb' * g)_k = sum_a Rnb(a,k) g_a
    for (int k = 0; k < D; ++k) h[k] = Rm[0 * 3 + k] * g[0] + Rm[1 * 3 + k] * g[1] + Rm[2 * 3 + k] * g[2];
  } else {
    double v = 1.0;
    int base[2] = {0, M.kmax[0]};
    for (int a = 0; a < 2; ++a) {
      const int nn = M.NN[a * M.m + c];
      v = v * 1.0 / sqrt(M.L[a]) * tabS[base[a] + nn - 1];
    }
    h[0] = v;
    for (int k = 1; k < D; ++k) h[k] = 0.0;
  }
}

}  // namespace rbpf
