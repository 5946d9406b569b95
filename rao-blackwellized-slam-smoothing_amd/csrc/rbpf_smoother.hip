// Conditional particle filter with ancestor sampling (src/particleSmoother.m,
// src/particleSmootherInformationForm.m) on the device.
#include "../../include/rbpf.h"
#include "rbpf_internal.hpp"
#include "rbpf_ctx.hpp"

namespace rbpf {
struct SmootherState {};
void smoother_free(rbpf_ctx* c) { delete c->sm; c->sm = nullptr; }
}  // namespace rbpf

extern "C" int rbpf_particle_smoother(const rbpf_model*, const rbpf_problem*, const rbpf_rng*, const rbpf_options*,
                                      int32_t, int32_t, rbpf_smoother_out*) {
  rbpf::set_error("smoother: not built yet");
  return RBPF_ERR_UNSUPPORTED;
}
