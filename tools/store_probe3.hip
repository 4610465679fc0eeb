// store_probe3.hip - round 6: ONE ceiling on ONE box (round-5 verdict, weak #5). tools/store_probe2.hip timed ten launches
// between two events (launch gaps inside the figure) and gave 6.39 TB/s for the best bare store stream and 5.5 TB/s for the
// "7 streams x 8 groups" pattern, yet k_partials_dna_cc stores 739.5 MB in 108.8 us = 6.80 TB/s by rocprof's kernel
// durations. This probe is made to be run UNDER rocprofv3 --kernel-trace --stats, in the same gpurun call as the C2 and C3
// bench lines, so that every figure is a kernel duration from the same box; each variant is its own kernel name, launched
// 20 times. What it varies is what the kernel does and probe2 did not:
//   * exactly C2's shape: 8 groups x 7 arrays x 1563 tiles of 8 KB (100 000 sites x 128 B), 256-thread workgroups,
//     one tile per wave, the workgroup order of kernels_common.h: xcd_linear
//   * streaming (nt) stores on all seven arrays, on six of seven, on none
//   * the stores of an array leaving as the array's values exist - `PACE` dependent FMAs per stored row ahead of each
//     array's sixteen stores (the kernel forms a CLV with ~150 FMAs per lane, then stores it) - instead of 112 stores
//     back to back
//   * + the 4-byte scaler word per site and array
// hipcc --offload-arch=gfx950 -O3 tools/store_probe3.hip -o /tmp/stp3 && rocprofv3 --kernel-trace --stats ... -- /tmp/stp3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcd_linear(unsigned total, bool on)
{
  const unsigned per = gridDim.x >> 3;
  const unsigned l = on ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
  return l < total ? l : ~0u;
}

template <bool NT> __device__ __forceinline__ void st(double *p, double v)
{
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// NTMASK: bit s set = array s leaves with streaming stores. PACE: dependent FMAs ahead of each row's store. SCALERS: + one
// 4-byte word per site and array.
template <unsigned NTMASK, int PACE, bool XCD, bool SCALERS>
__device__ __forceinline__ void streams7(double *out, unsigned *sc, unsigned tiles_per_array, unsigned groups, double seed)
{
  const unsigned nx = (tiles_per_array + 3u) / 4u;
  const unsigned l = xcd_linear(nx * groups, XCD);
  if (l == ~0u) return;
  const unsigned g = l / nx, bx = l - g * nx;
  const unsigned tile = bx * 4u + (threadIdx.x >> 6);
  if (tile >= tiles_per_array) return;
  const unsigned lane = threadIdx.x & 63u;
  double acc = seed + tile;
#pragma unroll
  for (unsigned s = 0; s < 7u; ++s)
  {
    double *p = out + ((size_t)(g * 7u + s) * tiles_per_array + tile) * 1024u + lane;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
    {
#pragma unroll
      for (int f = 0; f < PACE; ++f) acc = __builtin_fma(acc, 1.0000001, 0.5);
      v[k] = acc + k;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
    {
      if ((NTMASK >> s) & 1u) st<true>(p + k * 64, v[k]);
      else st<false>(p + k * 64, v[k]);
    }
    if (SCALERS) sc[(size_t)(g * 7u + s) * tiles_per_array * 64u + (size_t)tile * 64u + lane] = (unsigned)s;
  }
}

#define VARIANT(name, NTMASK, PACE, XCD, SCALERS)                                                                             \
  __global__ __launch_bounds__(256) void name(double *out, unsigned *sc, unsigned tpa, unsigned groups, double seed)           \
  {                                                                                                                           \
    streams7<NTMASK, PACE, XCD, SCALERS>(out, sc, tpa, groups, seed);                                                          \
  }
VARIANT(p3_7x8_plain_natural, 0x00u, 0, false, false)
VARIANT(p3_7x8_plain_xcd, 0x00u, 0, true, false)
VARIANT(p3_7x8_nt_natural, 0x7Fu, 0, false, false)
VARIANT(p3_7x8_nt_xcd, 0x7Fu, 0, true, false)
VARIANT(p3_7x8_nt6_xcd, 0x3Fu, 0, true, false)
VARIANT(p3_7x8_nt_xcd_paced4, 0x7Fu, 4, true, false)
VARIANT(p3_7x8_nt_xcd_paced9, 0x7Fu, 9, true, false)
VARIANT(p3_7x8_nt_xcd_paced9_scalers, 0x7Fu, 9, true, true)
VARIANT(p3_7x8_nt6_xcd_paced9_scalers, 0x3Fu, 9, true, true)
VARIANT(p3_7x8_plain_xcd_paced9_scalers, 0x00u, 9, true, true)
VARIANT(p3_7x8_nt_natural_paced9_scalers, 0x7Fu, 9, false, true)

// one array, every XCD its own contiguous eighth (probe2's best bare stream), plain and streaming
template <bool NT> __device__ __forceinline__ void bare(double *out, unsigned tiles)
{
  const unsigned l = xcd_linear((tiles + 3u) / 4u, true);
  if (l == ~0u) return;
  const unsigned tile = l * 4u + (threadIdx.x >> 6);
  if (tile >= tiles) return;
  double *p = out + (size_t)tile * 1024u + (threadIdx.x & 63u);
#pragma unroll
  for (int k = 0; k < 16; ++k) st<NT>(p + k * 64, (double)tile + k);
}
__global__ __launch_bounds__(256) void p3_bare_plain_xcd(double *out, unsigned tiles) { bare<false>(out, tiles); }
__global__ __launch_bounds__(256) void p3_bare_nt_xcd(double *out, unsigned tiles) { bare<true>(out, tiles); }

int main()
{
  const unsigned sites = 100000, groups = 8, tpa = (sites + 63) / 64; // C2: 1563 tiles per array
  const size_t arrays = (size_t)groups * 7, bytes = arrays * tpa * 8192;
  double *b;
  unsigned *sc;
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&sc, arrays * tpa * 64 * sizeof(unsigned)));
  CK(hipMemset(b, 0, bytes));
  const unsigned nx = (tpa + 3) / 4, grid = (nx * groups + 7) / 8 * 8;
  const unsigned tiles = (unsigned)(arrays * tpa), gridb = ((tiles + 3) / 4 + 7) / 8 * 8;
  printf("store_probe3: %zu arrays x %u tiles x 8 KB = %.1f MB (+ %.1f MB of scaler words where a variant writes them); grid %u x 256\n", arrays, tpa,
         bytes / 1e6, arrays * tpa * 256 / 1e6, grid);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
#define RUN7(name)                                                                                          \
  {                                                                                                         \
    for (int i = 0; i < 3; ++i) name<<<grid, 256>>>(b, sc, tpa, groups, 1.0);                                \
    CK(hipEventRecord(e0));                                                                                 \
    for (int i = 0; i < 20; ++i) name<<<grid, 256>>>(b, sc, tpa, groups, 1.0);                               \
    CK(hipEventRecord(e1));                                                                                 \
    CK(hipEventSynchronize(e1));                                                                            \
    float ms;                                                                                               \
    CK(hipEventElapsedTime(&ms, e0, e1));                                                                   \
    printf("%-40s %8.1f us per launch between events (gaps included)\n", #name, ms / 20 * 1e3);             \
  }
  RUN7(p3_7x8_plain_natural) RUN7(p3_7x8_plain_xcd) RUN7(p3_7x8_nt_natural) RUN7(p3_7x8_nt_xcd) RUN7(p3_7x8_nt6_xcd)
  RUN7(p3_7x8_nt_xcd_paced4) RUN7(p3_7x8_nt_xcd_paced9) RUN7(p3_7x8_nt_xcd_paced9_scalers) RUN7(p3_7x8_nt6_xcd_paced9_scalers)
  RUN7(p3_7x8_plain_xcd_paced9_scalers) RUN7(p3_7x8_nt_natural_paced9_scalers)
  for (int i = 0; i < 23; ++i) p3_bare_plain_xcd<<<gridb, 256>>>(b, tiles);
  for (int i = 0; i < 23; ++i) p3_bare_nt_xcd<<<gridb, 256>>>(b, tiles);
  CK(hipDeviceSynchronize());
  return 0;
}
