// gather_probe.hip - how fast can a wave read the 16 doubles of SCATTERED CLV entries, tiled layout
// ([tile][16 values][64 lanes]) against entry-contiguous layout ([entry][16 values])? Decides the
// layout of class-compressed DNA CLVs (site repeats: children are addressed through site_id maps).
// hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

__global__ void k_tiled(const double *clv, const unsigned *idx, double *out, unsigned n)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned e = idx[i];
  const double *p = clv + (size_t)(e >> 6) * 1024 + (e & 63u);
  double s = 0;
#pragma unroll
  for (int v = 0; v < 16; ++v) s += __builtin_nontemporal_load(p + v * 64);
  out[i] = s;
}

__global__ void k_aos(const double *clv, const unsigned *idx, double *out, unsigned n)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned e = idx[i];
  const double2 *p = reinterpret_cast<const double2 *>(clv + (size_t)e * 16);
  double s = 0;
#pragma unroll
  for (int v = 0; v < 8; ++v)
  {
    const double2 x = p[v];
    s += x.x + x.y;
  }
  out[i] = s;
}

int main()
{
  const unsigned n = 1u << 20; // 1M entries = 128 MB
  std::vector<unsigned> h(n);
  for (unsigned i = 0; i < n; ++i) h[i] = i;
  double *clv, *out;
  unsigned *idx;
  hipMalloc(&clv, (size_t)n * 128);
  hipMalloc(&out, (size_t)n * 8);
  hipMalloc(&idx, (size_t)n * 4);
  hipMemset(clv, 0, (size_t)n * 128);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const char *names[3] = {"identity", "mostly increasing (local shuffle of 256)", "random permutation"};
  for (int mode = 0; mode < 3; ++mode)
  {
    std::mt19937 rng(7);
    for (unsigned i = 0; i < n; ++i) h[i] = i;
    if (mode == 1)
      for (unsigned s = 0; s + 256 <= n; s += 256) std::shuffle(h.begin() + s, h.begin() + s + 256, rng);
    if (mode == 2) std::shuffle(h.begin(), h.end(), rng);
    hipMemcpy(idx, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    for (int k = 0; k < 2; ++k)
    {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep)
      {
        hipEventRecord(a);
        if (k == 0)
          hipLaunchKernelGGL(k_tiled, dim3(n / 256), dim3(256), 0, 0, clv, idx, out, n);
        else
          hipLaunchKernelGGL(k_aos, dim3(n / 256), dim3(256), 0, 0, clv, idx, out, n);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
      }
      printf("%-45s %-6s %8.1f us  %7.1f GB/s useful\n", names[mode], k ? "aos" : "tiled", best * 1e3, (double)n * 128 / (best * 1e-3) / 1e9);
    }
  }
  return 0;
}
