// store_probe.hip - entry-contiguous CLV stores: each lane writes its own 128 bytes (8 x 16 B at a
// 128-byte stride between lanes) against the same bytes written as dense 1 KB rows per instruction
// (what an LDS transpose would produce) and against the tiled layout's 8-byte rows.
// hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef double dbl2 __attribute__((ext_vector_type(2)));

__global__ void k_lane_entry(double *out, unsigned n)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dbl2 *p = reinterpret_cast<dbl2 *>(out + (size_t)i * 16);
  dbl2 v;
  v.x = i;
  v.y = 1.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) p[q] = v;
}

__global__ void k_dense_rows(double *out, unsigned n)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned wave0 = (i & ~63u), lane = i & 63u;
  dbl2 *p = reinterpret_cast<dbl2 *>(out + (size_t)wave0 * 16);
  dbl2 v;
  v.x = i;
  v.y = 1.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) p[q * 64 + lane] = v; // 64 lanes x 16 B = 1 KB contiguous per instruction
}

__global__ void k_tiled(double *out, unsigned n)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double *p = out + (size_t)(i >> 6) * 1024 + (i & 63u);
#pragma unroll
  for (int q = 0; q < 16; ++q) p[q * 64] = (double)i;
}

int main()
{
  const unsigned n = 1u << 21; // 2M entries = 256 MB
  double *out;
  hipMalloc(&out, (size_t)n * 128);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const char *names[3] = {"lane writes its entry (8 x 16 B, stride 128)", "dense 1 KB rows per instruction", "tiled 8-byte rows (512 B per instruction)"};
  for (int k = 0; k < 3; ++k)
  {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep)
    {
      hipEventRecord(a);
      if (k == 0) hipLaunchKernelGGL(k_lane_entry, dim3(n / 256), dim3(256), 0, 0, out, n);
      if (k == 1) hipLaunchKernelGGL(k_dense_rows, dim3(n / 256), dim3(256), 0, 0, out, n);
      if (k == 2) hipLaunchKernelGGL(k_tiled, dim3(n / 256), dim3(256), 0, 0, out, n);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      best = std::min(best, ms);
    }
    printf("%-50s %8.1f us  %7.1f GB/s\n", names[k], best * 1e3, (double)n * 128 / (best * 1e-3) / 1e9);
  }
  return 0;
}
