// chain_probe.hip - what limits a chain kernel (k_partials_dna_chain): every wave walks S steps, each
// step stores 4 x 512 B (one rate category of one 64-site tile) into a different CLV buffer; variants
// add the one-step-ahead byte load of a tip sibling (consumed in the next step, so the wave has to wait
// for everything older - its previous stores included - before it may go on).
// hipcc --offload-arch=gfx950 -O3 tools/chain_probe.hip -o /tmp/cp && /tmp/cp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bufs { double *clv[64]; const unsigned char *tip[64]; };

template <int LOADS, bool NT, int DEPTH>
__global__ __launch_bounds__(256) void k_steps(const Bufs b, unsigned steps, unsigned entries)
{
  const unsigned lane = threadIdx.x & 63u, rate = threadIdx.x >> 6;
  const unsigned n = blockIdx.x * 64u + lane;
  const size_t off = (size_t)blockIdx.x * 1024 + rate * 256 + lane;
  double v0 = n, v1 = n + 1, v2 = n + 2, v3 = n + 3;
  unsigned code[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) code[d] = LOADS ? b.tip[d][n] : 0u;
  for (unsigned s = 0; s < steps; s += DEPTH)
  {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
    {
      const unsigned c = code[d];
      if (LOADS) code[d] = b.tip[(s + d + DEPTH) & 63u][n]; // DEPTH steps ahead
      v0 = v0 * 1.0000001 + (double)(c & 1u);
      v1 = v1 * 1.0000001 + (double)(c & 2u);
      v2 = v2 * 1.0000001 + (double)(c & 4u);
      v3 = v3 * 1.0000001 + (double)(c & 8u);
      double *p = b.clv[(s + d) & 63u] + off;
      if (NT)
      {
        __builtin_nontemporal_store(v0, p);
        __builtin_nontemporal_store(v1, p + 64);
        __builtin_nontemporal_store(v2, p + 128);
        __builtin_nontemporal_store(v3, p + 192);
      }
      else
      {
        p[0] = v0;
        p[64] = v1;
        p[128] = v2;
        p[192] = v3;
      }
    }
  }
}

int main()
{
  const unsigned entries = 100000, tiles = (entries + 63) / 64, steps = 64;
  Bufs b;
  for (int i = 0; i < 64; ++i)
  {
    CK(hipMalloc(&b.clv[i], (size_t)tiles * 1024 * 8));
    unsigned char *t;
    CK(hipMalloc(&t, tiles * 64));
    CK(hipMemset(t, i & 15, tiles * 64));
    b.tip[i] = t;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)tiles * 8192 * steps;
    printf("%-28s %8.1f us  %6.2f TB/s  (%.2f us per step)\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, ms / reps * 1e3 / steps);
  };
  run("stores only, nt", [&] { k_steps<0, true, 1><<<tiles, 256>>>(b, steps, entries); });
  run("stores only, plain", [&] { k_steps<0, false, 1><<<tiles, 256>>>(b, steps, entries); });
  run("+ byte load 1 ahead, nt", [&] { k_steps<1, true, 1><<<tiles, 256>>>(b, steps, entries); });
  run("+ byte load 2 ahead, nt", [&] { k_steps<1, true, 2><<<tiles, 256>>>(b, steps, entries); });
  run("+ byte load 4 ahead, nt", [&] { k_steps<1, true, 4><<<tiles, 256>>>(b, steps, entries); });
  run("+ byte load 8 ahead, nt", [&] { k_steps<1, true, 8><<<tiles, 256>>>(b, steps, entries); });
  return 0;
}
