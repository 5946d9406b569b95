// Microbenchmark: what HBM rate does the step kernel's access pattern allow?  One workgroup per
// "particle" (530 KB read from a permuted source slot + 530 KB written), variants of the copy loop.
// Build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o gpurun_out/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int UC, bool NT, int FMA, bool NTS = NT>
__global__ __launch_bounds__(256) void copy_wg(const double* __restrict__ in, double* __restrict__ out, const int* __restrict__ perm, size_t per) {
  const int b = blockIdx.x;
  const dbl2* src = reinterpret_cast<const dbl2*>(in + (size_t)perm[b] * per);
  dbl2* dst = reinterpret_cast<dbl2*>(out + (size_t)b * per);
  const size_t n2 = per / 2;
  double acc = 0.0;
  for (size_t q = threadIdx.x; q < n2; q += 256 * UC) {
    dbl2 v[UC];
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const size_t qq = q + (size_t)u * 256;
      if (qq < n2) v[u] = NT ? __builtin_nontemporal_load(src + qq) : src[qq];
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const size_t qq = q + (size_t)u * 256;
      if (qq < n2) {
        dbl2 o = v[u];
#pragma unroll
        for (int f = 0; f < FMA; ++f) { o.x = fma(o.x, 1.0000001, 1e-9); o.y = fma(o.y, 1.0000001, 1e-9); acc += o.x; }
        if (NTS) __builtin_nontemporal_store(o, dst + qq); else dst[qq] = o;
      }
    }
  }
  if (FMA && acc == 123.456) out[0] = acc;
}

template <int UC, bool NT>
__global__ __launch_bounds__(256) void copy_flat(const double* __restrict__ in, double* __restrict__ out, size_t n2) {
  const dbl2* src = reinterpret_cast<const dbl2*>(in);
  dbl2* dst = reinterpret_cast<dbl2*>(out);
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n2; q += stride * UC) {
    dbl2 v[UC];
#pragma unroll
    for (int u = 0; u < UC; ++u) { const size_t qq = q + u * stride; if (qq < n2) v[u] = NT ? __builtin_nontemporal_load(src + qq) : src[qq]; }
#pragma unroll
    for (int u = 0; u < UC; ++u) { const size_t qq = q + u * stride; if (qq < n2) { if (NT) __builtin_nontemporal_store(v[u], dst + qq); else dst[qq] = v[u]; } }
  }
}

template <typename F>
static double time_ms(F f, int reps) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f(); f();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) f();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 8192;
  const int n = argc > 2 ? atoi(argv[2]) : 259;
  const size_t per = (size_t)n * 256;   // doubles per particle (core block of n=259: 259*256)
  double *in, *out; int* perm;
  CHECK(hipMalloc(&in, N * per * 8)); CHECK(hipMalloc(&out, N * per * 8)); CHECK(hipMalloc(&perm, N * 4));
  CHECK(hipMemset(in, 0x11, N * per * 8));
  std::vector<int> id(N), rnd(N), res(N);
  std::iota(id.begin(), id.end(), 0);
  std::mt19937 g(1);
  rnd = id; std::shuffle(rnd.begin(), rnd.end(), g);
  std::uniform_int_distribution<int> U(0, N - 1);
  for (int i = 0; i < N; ++i) res[i] = U(g);              // multinomial-like ancestors (duplicates)
  std::vector<int> srt = res; std::sort(srt.begin(), srt.end());
  std::vector<int> xcd(N);   // sorted order, but consecutive sorted positions share an XCD (blocks b, b+8)
  for (int s = 0; s < N; ++s) { const int chunk = N / 8; const int b = (s % chunk) * 8 + s / chunk; xcd[b] = srt[s]; }
  const double bytes = 2.0 * N * per * 8;
  struct { const char* name; std::vector<int>* p; } perms[] = {{"identity", &id}, {"shuffled", &rnd}, {"resampled", &res}, {"resampled-sorted", &srt}, {"sorted-xcd-grouped", &xcd}};
  for (auto& pm : perms) {
    CHECK(hipMemcpy(perm, pm.p->data(), N * 4, hipMemcpyHostToDevice));
    auto rep = [&](const char* nm, double ms) { printf("%-18s %-22s %8.3f ms  %7.0f GB/s\n", pm.name, nm, ms, bytes / ms / 1e6); };
    rep("wg uc4", time_ms([&] { hipLaunchKernelGGL((copy_wg<4, false, 0>), dim3(N), dim3(256), 0, 0, in, out, perm, per); }, 10));
    rep("wg uc8", time_ms([&] { hipLaunchKernelGGL((copy_wg<8, false, 0>), dim3(N), dim3(256), 0, 0, in, out, perm, per); }, 10));
    rep("wg uc8 nt", time_ms([&] { hipLaunchKernelGGL((copy_wg<8, true, 0>), dim3(N), dim3(256), 0, 0, in, out, perm, per); }, 10));
    rep("wg uc16 nt", time_ms([&] { hipLaunchKernelGGL((copy_wg<16, true, 0>), dim3(N), dim3(256), 0, 0, in, out, perm, per); }, 10));
    rep("wg uc8 ld+ntst", time_ms([&] { hipLaunchKernelGGL((copy_wg<8, false, 0, true>), dim3(N), dim3(256), 0, 0, in, out, perm, per); }, 10));
    rep("wg uc8 ld+ntst fma6", time_ms([&] { hipLaunchKernelGGL((copy_wg<8, false, 6, true>), dim3(N), dim3(256), 0, 0, in, out, perm, per); }, 10));
    rep("wg uc8 nt fma6", time_ms([&] { hipLaunchKernelGGL((copy_wg<8, true, 6>), dim3(N), dim3(256), 0, 0, in, out, perm, per); }, 10));
  }
  const size_t n2 = (size_t)N * per / 2;
  for (int grid : {2048, 4096, 8192, 16384}) {
    double ms = time_ms([&] { hipLaunchKernelGGL((copy_flat<8, false>), dim3(grid), dim3(256), 0, 0, in, out, n2); }, 10);
    printf("flat grid=%-6d uc8            %8.3f ms  %7.0f GB/s\n", grid, ms, bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((copy_flat<8, true>), dim3(grid), dim3(256), 0, 0, in, out, n2); }, 10);
    printf("flat grid=%-6d uc8 nt         %8.3f ms  %7.0f GB/s\n", grid, ms, bytes / ms / 1e6);
  }
  double ms = time_ms([&] { CHECK(hipMemcpyAsync(out, in, N * per * 8, hipMemcpyDeviceToDevice, 0)); }, 5);
  printf("hipMemcpy D2D                         %8.3f ms  %7.0f GB/s\n", ms, bytes / ms / 1e6);
  return 0;
}
