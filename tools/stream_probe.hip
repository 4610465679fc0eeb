// stream_probe.hip - what the part sustains for pure streams of the CLV tile shape (every lane 8 B,
// a wave covers 512 contiguous bytes per instruction, 16 instructions per tile = 8 KB): store-only
// (plain / non-temporal), load-only, and copy, over buffer sizes from Infinity-Cache-sized to 8 GB.
// The store-only line is the ceiling for the tip-fed group launches (k_partials_dna_cc reads 8 B and
// writes 924 B per site).
// hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/stp && /tmp/stp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __global__ __launch_bounds__(256) void k_store(double *out, size_t tiles)
{
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= tiles) return;
  double *p = out + wave * 1024 + (threadIdx.x & 63);
  const double v = (double)wave;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (NT) __builtin_nontemporal_store(v + k, p + k * 64);
    else p[k * 64] = v + k;
}

template <bool NT> __global__ __launch_bounds__(256) void k_load(const double *in, double *sink, size_t tiles)
{
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= tiles) return;
  const double *p = in + wave * 1024 + (threadIdx.x & 63);
  double a = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) a += NT ? __builtin_nontemporal_load(p + k * 64) : p[k * 64];
  if (a == 12345.678) sink[0] = a;
}

__global__ __launch_bounds__(256) void k_copy(const double *in, double *out, size_t tiles)
{
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= tiles) return;
  const double *p = in + wave * 1024 + (threadIdx.x & 63);
  double *q = out + wave * 1024 + (threadIdx.x & 63);
  double v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = __builtin_nontemporal_load(p + k * 64);
#pragma unroll
  for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(v[k] * 1.5, q + k * 64);
}

int main()
{
  const size_t sizes[] = {128ull << 20, 745ull << 20, 2200ull << 20, 8000ull << 20};
  double *a, *b;
  CK(hipMalloc(&a, sizes[3]));
  CK(hipMalloc(&b, sizes[3]));
  CK(hipMemset(a, 0, sizes[3]));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (size_t bytes : sizes)
  {
    const size_t tiles = bytes / 8192;
    const unsigned grid = (unsigned)((tiles + 3) / 4);
    auto run = [&](const char *name, auto launch, double factor) {
      for (int i = 0; i < 2; ++i) launch();
      CK(hipEventRecord(e0));
      const int reps = 10;
      for (int i = 0; i < reps; ++i) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%6zu MB %-10s %8.1f us  %6.2f TB/s\n", bytes >> 20, name, ms / reps * 1e3, factor * bytes / (ms / reps * 1e-3) / 1e12);
    };
    run("store", [&] { k_store<false><<<grid, 256>>>(b, tiles); }, 1);
    run("store.nt", [&] { k_store<true><<<grid, 256>>>(b, tiles); }, 1);
    run("load", [&] { k_load<false><<<grid, 256>>>(a, b, tiles); }, 1);
    run("load.nt", [&] { k_load<true><<<grid, 256>>>(a, b, tiles); }, 1);
    run("copy.nt", [&] { k_copy<<<grid, 256>>>(a, b, tiles); }, 2);
  }
  return 0;
}
