// store_probe2.hip - is the 5.6 TB/s "plain-store ceiling" (profiles/r1_stream_probe.txt) a property of the part or
// of the store idiom? Pure store streams over buffers well beyond the 256 MiB Infinity Cache, varying
//   * bytes per lane and instruction (8 / 16), i.e. 512 B or 1 KB contiguous per wave instruction
//   * cache policy bits: plain, nt, sc0 sc1 (write-through), sc1, nt sc0 sc1
//   * how much a wave writes and in what order (one 8 KB tile per wave / grid-stride persistent waves)
//   * workgroup -> address mapping: consecutive workgroups on consecutive tiles (they land on different XCDs), or every
//     XCD on its own contiguous eighth of the buffer
// hipcc --offload-arch=gfx950 -O3 tools/store_probe2.hip -o /tmp/stp2 && /tmp/stp2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

enum Pol { PLAIN, NT, SC01, SC1, NTSC01 };

template <int POL> __device__ __forceinline__ void st8(double *p, double v)
{
  if (POL == PLAIN) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (POL == NT) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (POL == SC01) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (POL == SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (POL == NTSC01) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int POL> __device__ __forceinline__ void st16(double *p, d2 v)
{
  if (POL == PLAIN) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (POL == NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (POL == SC01) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (POL == SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (POL == NTSC01) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// tile = 8 KB. XCDMAP: workgroup b runs on XCD b % 8 (round-robin dispatch); give XCD x the x-th eighth of the tiles.
template <int POL, int WIDTH, bool XCDMAP> __global__ __launch_bounds__(256) void k_store(double *out, size_t tiles)
{
  size_t wg = blockIdx.x;
  if (XCDMAP)
  {
    const size_t per = (gridDim.x + 7) / 8;
    wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  const size_t wave = wg * 4 + (threadIdx.x >> 6);
  if (wave >= tiles) return;
  const unsigned lane = threadIdx.x & 63;
  const double v = (double)wave;
  if (WIDTH == 8)
  {
    double *p = out + wave * 1024 + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) st8<POL>(p + k * 64, v + k);
  }
  else
  {
    double *p = out + wave * 1024 + 2 * lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) st16<POL>(p + k * 128, d2{v + k, v - k});
  }
}

// persistent: `waves` waves in all, each walks tiles wave, wave + waves, ...
template <int POL, int WIDTH> __global__ __launch_bounds__(256) void k_store_persistent(double *out, size_t tiles)
{
  const size_t nw = (size_t)gridDim.x * 4;
  const unsigned lane = threadIdx.x & 63;
  for (size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); wave < tiles; wave += nw)
  {
    const double v = (double)wave;
    if (WIDTH == 8)
    {
      double *p = out + wave * 1024 + lane;
#pragma unroll
      for (int k = 0; k < 16; ++k) st8<POL>(p + k * 64, v + k);
    }
    else
    {
      double *p = out + wave * 1024 + 2 * lane;
#pragma unroll
      for (int k = 0; k < 8; ++k) st16<POL>(p + k * 128, d2{v + k, v - k});
    }
  }
}

// the same question for reads and for a 2 reads + 1 write stream (the shape of an inner x inner level)
template <bool XCDMAP, bool NT> __global__ __launch_bounds__(256) void k_load(const double *in, double *sink, size_t tiles)
{
  size_t wg = blockIdx.x;
  if (XCDMAP)
  {
    const size_t per = (gridDim.x + 7) / 8;
    wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  const size_t wave = wg * 4 + (threadIdx.x >> 6);
  if (wave >= tiles) return;
  const double *p = in + wave * 1024 + (threadIdx.x & 63);
  double a = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) a += NT ? __builtin_nontemporal_load(p + k * 64) : p[k * 64];
  if (a == 12345.678) sink[0] = a;
}

template <bool XCDMAP> __global__ __launch_bounds__(256) void k_ii(const double *l, const double *r, double *out, size_t tiles)
{
  size_t wg = blockIdx.x;
  if (XCDMAP)
  {
    const size_t per = (gridDim.x + 7) / 8;
    wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  const size_t wave = wg * 4 + (threadIdx.x >> 6);
  if (wave >= tiles) return;
  const size_t o = wave * 1024 + (threadIdx.x & 63);
  double v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = __builtin_nontemporal_load(l + o + k * 64);
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] *= __builtin_nontemporal_load(r + o + k * 64);
#pragma unroll
  for (int k = 0; k < 16; ++k) out[o + k * 64] = v[k];
}

// every XCD owns contiguous runs of `run` workgroups (run = 1: the natural order, run = grid / 8: one eighth each)
__global__ __launch_bounds__(256) void k_store_runs(double *out, size_t tiles, unsigned run)
{
  const unsigned xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const size_t wg = ((size_t)(idx / run) * 8u + xcd) * run + idx % run;
  const size_t wave = wg * 4 + (threadIdx.x >> 6);
  if (wave >= tiles) return;
  double *p = out + wave * 1024 + (threadIdx.x & 63);
  const double v = (double)wave;
#pragma unroll
  for (int k = 0; k < 16; ++k) st8<PLAIN>(p + k * 64, v + k);
}

// the shape of k_partials_dna_cc: a wave writes tile t of NS different arrays (a group's seven CLVs), `groups` groups
template <int NS, bool XCDMAP, int POL> __global__ __launch_bounds__(256) void k_store_streams(double *out, size_t tiles_per_array, unsigned groups)
{
  const size_t nx = (tiles_per_array + 3) / 4, total = nx * groups;
  size_t l = blockIdx.x;
  if (XCDMAP)
  {
    const size_t per = gridDim.x >> 3;
    l = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  if (l >= total) return;
  const size_t g = l / nx, bx = l - g * nx;
  const size_t tile = bx * 4 + (threadIdx.x >> 6);
  if (tile >= tiles_per_array) return;
  const double v = (double)tile;
#pragma unroll
  for (int s = 0; s < NS; ++s)
  {
    double *p = out + ((g * NS + s) * tiles_per_array + tile) * 1024 + (threadIdx.x & 63);
#pragma unroll
    for (int k = 0; k < 16; ++k) st8<POL>(p + k * 64, v + k + s);
  }
}

int main()
{
  const size_t sizes[] = {745ull << 20, 1550ull << 20, 4000ull << 20};
  double *b, *b2;
  CK(hipMalloc(&b2, 4096));
  CK(hipMalloc(&b, sizes[2]));
  CK(hipMemset(b, 0, sizes[2]));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (size_t bytes : sizes)
  {
    const size_t tiles = bytes / 8192;
    const unsigned grid = (unsigned)((tiles + 3) / 4);
    auto run_ = [&](const char *name, auto launch) {
      for (int i = 0; i < 2; ++i) launch();
      CK(hipEventRecord(e0));
      const int reps = 10;
      for (int i = 0; i < reps; ++i) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%6zu MB %-28s %8.1f us  %6.2f TB/s\n", bytes >> 20, name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
      fflush(stdout);
    };
    auto run = run_;
#define RUN(POL, W, X) run(#POL " " #W "B" #X, [&] { k_store<POL, W, X><<<grid, 256>>>(b, tiles); })
    RUN(PLAIN, 8, false); RUN(PLAIN, 16, false); RUN(NT, 8, false); RUN(NT, 16, false);
    RUN(SC01, 8, false); RUN(SC01, 16, false); RUN(SC1, 16, false); RUN(NTSC01, 16, false);
    RUN(PLAIN, 8, true); RUN(PLAIN, 16, true); RUN(NT, 16, true); RUN(SC01, 16, true);
#define RUNP(POL, W, G) run(#POL " " #W "B persistent " #G, [&] { k_store_persistent<POL, W><<<G, 256>>>(b, tiles); })
    run("load plain", [&] { k_load<false, false><<<grid, 256>>>(b, b2, tiles); });
    run("load plain xcd", [&] { k_load<true, false><<<grid, 256>>>(b, b2, tiles); });
    run("load nt", [&] { k_load<false, true><<<grid, 256>>>(b, b2, tiles); });
    run("load nt xcd", [&] { k_load<true, true><<<grid, 256>>>(b, b2, tiles); });
    {
      const size_t t3 = tiles / 3; // three arrays of a third each: bytes moved = the same total
      const unsigned g3 = (unsigned)((t3 + 3) / 4);
      run("2 loads + 1 store", [&] { k_ii<false><<<g3, 256>>>(b, b + t3 * 1024, b + 2 * t3 * 1024, t3); });
      run("2 loads + 1 store xcd", [&] { k_ii<true><<<g3, 256>>>(b, b + t3 * 1024, b + 2 * t3 * 1024, t3); });
    }
    for (unsigned run : {1u, 4u, 16u, 64u, 256u, 1024u, 4096u})
    {
      char name[64];
      snprintf(name, sizeof name, "store, runs of %u wgs per XCD", run);
      const unsigned g8 = (grid + 8 * run - 1) / (8 * run) * (8 * run);
      run_(name, [&] { k_store_runs<<<g8, 256>>>(b, tiles, run); });
    }
    {
      const unsigned groups = 8;
      const size_t tpa = tiles / (7 * groups);
      const unsigned nx = (unsigned)((tpa + 3) / 4), g8 = (nx * groups + 7) / 8 * 8;
      run("7 streams x 8 groups", [&] { k_store_streams<7, false, PLAIN><<<g8, 256>>>(b, tpa, groups); });
      run("7 streams x 8 groups xcd", [&] { k_store_streams<7, true, PLAIN><<<g8, 256>>>(b, tpa, groups); });
      run("7 streams x 8 groups xcd nt", [&] { k_store_streams<7, true, NT><<<g8, 256>>>(b, tpa, groups); });
      const size_t tpa1 = tiles / groups;
      const unsigned nx1 = (unsigned)((tpa1 + 3) / 4), g81 = (nx1 * groups + 7) / 8 * 8;
      run("1 stream x 8 groups xcd", [&] { k_store_streams<1, true, PLAIN><<<g81, 256>>>(b, tpa1, groups); });
      const size_t tpa3 = tiles / (3 * groups);
      const unsigned nx3 = (unsigned)((tpa3 + 3) / 4), g83 = (nx3 * groups + 7) / 8 * 8;
      run("3 streams x 8 groups", [&] { k_store_streams<3, false, PLAIN><<<g83, 256>>>(b, tpa3, groups); });
      run("3 streams x 8 groups xcd", [&] { k_store_streams<3, true, PLAIN><<<g83, 256>>>(b, tpa3, groups); });
    }
    RUNP(PLAIN, 8, 512); RUNP(PLAIN, 16, 512); RUNP(PLAIN, 16, 1024); RUNP(PLAIN, 16, 2048); RUNP(NT, 16, 1024); RUNP(SC01, 16, 1024);
  }
  return 0;
}
