// Microbenchmark: how many bytes per clock can ONE CU pull through its vector L1 miss path, from HBM and from the L2, with the
// load shapes the factorisation kernels use (8 B per lane = one 512 B MFMA operand fragment per wave instruction; 16 B per lane)?
// One 512-thread workgroup per CU (as chol_solve64_kernel), every wave sums a private stream; run on all CUs and on half of them
// (hipExtStreamCreateWithCUMask): a per-CU ceiling shows as the same B/clk/CU in both.
// Build: hipcc --offload-arch=gfx950 -O3 tools/cu_rate_probe.hip -o gpurun_out/cu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// every wave: `iters` rounds of DEPTH loads of W doubles per lane, consecutive 64*W*8 B fragments; the stream of a wave wraps inside
// `span` doubles (span small: L2 / L1 hits; span = whole share: HBM)
template <int W, int DEPTH>
__global__ __launch_bounds__(512, 1) void pull(const double* __restrict__ in, double* __restrict__ out, size_t span, int iters) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t frag = 64 * W;                                   // doubles per wave instruction
  const double* base = in + ((size_t)blockIdx.x * 8 + wv) * span;
  const size_t nfrag = span / frag;
  double acc = 0.0;
  size_t f = 0;
  for (int it = 0; it < iters; ++it) {
    double v[DEPTH][W];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const double* p = base + f * frag + (size_t)lane * W;
      if (W == 2) { const dbl2 t = *reinterpret_cast<const dbl2*>(p); v[d][0] = t.x; v[d][W - 1] = t.y; }
      else v[d][0] = *p;
      f = (f + 1 == nfrag) ? 0 : f + 1;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int w = 0; w < W; ++w) acc += v[d][w];
  }
  if (acc == 123.456) out[0] = acc;
}

// the element loader's pattern: per workgroup a column-major n x n matrix (ld = n), every wave reads 16 x 64 strips of its lower
// triangle -- per wave instruction 4 segments of 128 B (16 rows of 4 columns, column stride 8 n bytes), 16 instructions in flight;
// PACKED: the same strips as 8 KB contiguous (fragment order), i.e. what a tile-packed storage of the matrix would give
template <bool PACKED>
__global__ __launch_bounds__(512, 1) void strips(const double* __restrict__ in, double* __restrict__ out, int n, int reps_) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int RT = n >> 4;
  double acc = 0.0;
  for (int rep = 0; rep < reps_; ++rep) {
    const double* base = in + ((size_t)blockIdx.x * reps_ + rep) * (size_t)n * n;
    for (int J = 0; 64 * J + 64 <= n; ++J)
      for (int rt = 4 * J + 4 + wv; rt < RT; rt += 8) {
        double v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c)
          v[c] = PACKED ? base[((size_t)rt * (4 * RT) + 16 * J + c) * 64 + lane] : base[(size_t)16 * rt + r + (size_t)n * (64 * J + 4 * c + g)];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc += v[c];
      }
  }
  if (acc == 123.456) out[0] = acc;
}

template <bool PACKED>
static void run_strips(const char* name, hipStream_t st, int ncu, const double* in, double* out) {
  const int n = 512, reps_ = 16;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((strips<PACKED>), dim3(ncu), dim3(512), 0, st, in, out, n, reps_);
  CHECK(hipEventRecord(a, st));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((strips<PACKED>), dim3(ncu), dim3(512), 0, st, in, out, n, reps_);
  CHECK(hipEventRecord(b, st));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= 3;
  double strips_per_matrix = 0;
  for (int J = 0; 64 * J + 64 <= n; ++J) strips_per_matrix += (n >> 4) - (4 * J + 4);
  const double bytes = (double)ncu * reps_ * strips_per_matrix * 8192.0;
  printf("%-44s CUs %3d  %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU (2.4 GHz)\n", name, ncu, ms, bytes / ms * 1e-9, bytes / ncu / (ms * 1e-3 * 2.4e9));
}

template <int W, int DEPTH>
static void run(const char* name, hipStream_t st, int ncu, const double* in, double* out, size_t span, size_t bytes_per_wave) {
  const int iters = (int)(bytes_per_wave / (64 * W * 8 * DEPTH));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((pull<W, DEPTH>), dim3(ncu), dim3(512), 0, st, in, out, span, iters);
  CHECK(hipEventRecord(a, st));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((pull<W, DEPTH>), dim3(ncu), dim3(512), 0, st, in, out, span, iters);
  CHECK(hipEventRecord(b, st));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= 3;
  const double bytes = (double)ncu * 8 * iters * 64.0 * W * 8 * DEPTH;
  printf("%-44s CUs %3d  %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU (2.4 GHz)\n", name, ncu, ms, bytes / ms * 1e-9, bytes / ncu / (ms * 1e-3 * 2.4e9));
}

int main() {
  const size_t per_wave_hbm = (size_t)4 << 20;                  // doubles: 32 MB per wave, 256 CUs x 8 waves = 64 GB? no: see below
  // HBM case: span = 1 M doubles (8 MB) per wave -> 256 x 8 x 8 MB = 16 GB footprint, streamed once per launch
  const size_t span_hbm = (size_t)1 << 20, span_l2 = (size_t)1 << 10, span_mall = (size_t)1 << 14;   // 8 KB (L1/L2), 128 KB per wave (256 MB total: MALL-ish)
  (void)per_wave_hbm;
  double *in, *out;
  CHECK(hipMalloc(&in, (size_t)256 * 8 * span_hbm * sizeof(double)));
  CHECK(hipMalloc(&out, 4096));
  CHECK(hipMemset(in, 0, (size_t)256 * 8 * span_hbm * sizeof(double)));
  for (int half = 0; half < 2; ++half) {
    hipStream_t st;
    uint32_t mask[8];
    for (int w = 0; w < 8; ++w) mask[w] = half ? 0x0000ffffu : 0xffffffffu;
    CHECK(hipExtStreamCreateWithCUMask(&st, 8, mask));
    const int ncu = half ? 128 : 256;
    const size_t bw = (size_t)8 << 20;                          // bytes per wave per launch
    run_strips<false>("lower 16x64 strips of column-major matrices", st, ncu, in, out);
    run_strips<true>("the same strips, tile-packed (8 KB contiguous)", st, ncu, in, out);
    run<1, 8>("HBM stream, 8 B/lane, 8 loads in flight", st, ncu, in, out, span_hbm, bw);
    run<1, 16>("HBM stream, 8 B/lane, 16 loads in flight", st, ncu, in, out, span_hbm, bw);
    run<1, 32>("HBM stream, 8 B/lane, 32 loads in flight", st, ncu, in, out, span_hbm, bw);
    run<2, 8>("HBM stream, 16 B/lane, 8 loads in flight", st, ncu, in, out, span_hbm, bw);
    run<2, 16>("HBM stream, 16 B/lane, 16 loads in flight", st, ncu, in, out, span_hbm, bw);
    run<1, 16>("128 KB per wave (MALL/L2), 8 B/lane, 16", st, ncu, in, out, span_mall, bw);
    run<2, 16>("128 KB per wave (MALL/L2), 16 B/lane, 16", st, ncu, in, out, span_mall, bw);
    run<1, 16>("8 KB per wave (L2: 64 KB per CU > L1), 8 B/lane, 16", st, ncu, in, out, span_l2, bw);
    run<2, 16>("8 KB per wave (L2), 16 B/lane, 16", st, ncu, in, out, span_l2, bw);
    CHECK(hipStreamDestroy(st));
  }
  return 0;
}
