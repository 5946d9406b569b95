// Microbenchmark (r05): if two particles that share a stored covariance ran in ONE workgroup, in lock-step, would the second one's
// loads be served by the CU's vector L1 (or merged with the first one's outstanding misses) instead of crossing L2 -> L1 again?
// Workgroups of 8 waves; wave w streams a private 1.2 MB "matrix" in rounds of 16 wave loads of 1 KB (dbl2 per lane), as the
// read-only step kernel does.  shared = 0: the 8 waves read 8 different matrices; shared = 1: waves w and w + 4 read the SAME matrix
// (4 distinct per workgroup), a workgroup barrier every `sync` rounds keeping them together.  Same number of loads either way;
// matrices are spread over a 16 GB buffer (HBM / Infinity Cache resident like the banks).  Prints GB/s of bytes REQUESTED.
// Build: hipcc --offload-arch=gfx950 -O3 tools/pair_probe.hip -o gpurun_out/pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr size_t kMat = 152576;                 // doubles per matrix (1.19 MB: the block-lower covariance at nLin = 515)

template <int SHARED>
__global__ __launch_bounds__(512, 1) void stream(const double* __restrict__ in, double* __restrict__ out, int n_mat, int sync) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // matrix of this wave: consecutive workgroups read neighbouring matrices (as the sorted processing order does)
  const int slot = SHARED ? (wv & 3) : wv;
  const size_t mat = ((size_t)blockIdx.x * (SHARED ? 4 : 8) + slot) % (size_t)n_mat;
  const double* base = in + mat * kMat;
  double acc = 0.0;
  const int rounds = (int)(kMat / (16 * 128));   // 16 wave loads of 128 doubles per round
  for (int r = 0; r < rounds; ++r) {
    dbl2 v[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) v[d] = *reinterpret_cast<const dbl2*>(base + ((size_t)r * 16 + d) * 128 + 2 * lane);
#pragma unroll
    for (int d = 0; d < 16; ++d) acc += v[d].x * 1.0000001 + v[d].y;
    if (sync > 0 && (r % sync) == sync - 1) __syncthreads();
  }
  if (acc == 123.456) out[0] = acc;
}

int main() {
  const int n_mat = 12000;                       // 14.3 GB of matrices
  double *in, *out;
  CHECK(hipMalloc(&in, (size_t)n_mat * kMat * 8));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(in, 0, (size_t)n_mat * kMat * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int particles = 65536;                   // wave-streams per launch
  for (int sync : {0, 1, 4}) {
    for (int shared = 0; shared < 2; ++shared) {
      const int grid = particles / 8;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        if (shared) hipLaunchKernelGGL(stream<1>, dim3(grid), dim3(512), 0, 0, in, out, n_mat, sync);
        else hipLaunchKernelGGL(stream<0>, dim3(grid), dim3(512), 0, 0, in, out, n_mat, sync);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 1)
          printf("sync every %d rounds, %s: %.3f ms for %d streams of 1.19 MB = %.0f GB/s requested (%d distinct matrices per workgroup)\n", sync,
                 shared ? "waves w / w + 4 share a matrix" : "8 private matrices          ", ms, particles, particles * kMat * 8.0 / ms / 1e6, shared ? 4 : 8);
      }
    }
  }
  return 0;
}
