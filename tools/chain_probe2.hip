// chain_probe2.hip - chain_probe.hip with the chain kernel's per-step extras switched on one at a
// time: disabled (zero-size) buffer loads / stores, the per-step barrier, per-step descriptor fetches
// through the scalar cache, more arithmetic.
// hipcc --offload-arch=gfx950 -O3 tools/chain_probe2.hip -o /tmp/cp2 && /tmp/cp2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Step { double *clv; const unsigned char *tip; const double *mat; unsigned clv_bytes, tip_bytes; };
typedef const Step __attribute__((address_space(4))) *cstep_p;
typedef const double __attribute__((address_space(4))) *cdouble_p;
typedef unsigned u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

template <int DIS_LD, int DIS_ST, bool BARRIER, bool MATS, int FLOPS, bool EXECOFF = false>
__global__ __launch_bounds__(256) void k_steps(const Step *steps_, unsigned steps, unsigned entries)
{
  __shared__ unsigned long long ballots[2][4];
  cstep_p st = (cstep_p)(uintptr_t)steps_;
  const unsigned lane = threadIdx.x & 63u, rate = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned n = blockIdx.x * 64u + lane;
  const unsigned voff = (blockIdx.x * 1024u + rate * 256u + lane) * 8u;
  double v[4] = {(double)n, n + 1.0, n + 2.0, n + 3.0};
  unsigned code = __builtin_amdgcn_raw_buffer_load_b8(rsrc(st[0].tip, st[0].tip_bytes), n, 0, 0);
  for (unsigned s = 0; s < steps; ++s)
  {
    const unsigned c = code;
    double extra = 0.0;
    {
      const unsigned char *tp = st[s + 1].tip;
      const unsigned tb = st[s + 1].tip_bytes;
      code = __builtin_amdgcn_raw_buffer_load_b8(rsrc(tp, tb), n, 0, 0);
      if (EXECOFF)
      {
        // the same loads with a full-size descriptor, but under a lane mask that is 0 at run time
        unsigned pred = tb == 0xffffffffu ? 1u : 0u; // never true
        asm volatile("v_mov_b32 %0, %1" : "=v"(pred) : "s"(pred));
        double e[DIS_LD > 0 ? DIS_LD : 1];
#pragma unroll
        for (int d = 0; d < DIS_LD; ++d) e[d] = 0.0;
        if (pred)
        {
#pragma unroll
          for (int d = 0; d < DIS_LD; ++d)
            e[d] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc(st[s + 1].clv, st[s + 1].clv_bytes), voff + d * 512u, 0, 2));
        }
#pragma unroll
        for (int d = 0; d < DIS_LD; ++d) extra += e[d];
      }
      else
      {
#pragma unroll
        for (int d = 0; d < DIS_LD; ++d)
          extra += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc(tp, 0u), voff + d * 512u, 0, 2));
      }
    }
    double m[4] = {1.0000001, 1.0000002, 1.0000003, 1.0000004};
    if (MATS)
    {
      cdouble_p mp = (cdouble_p)(uintptr_t)st[s].mat + rate * 16u;
#pragma unroll
      for (int i = 0; i < 4; ++i) m[i] = mp[i];
    }
#pragma unroll
    for (int f = 0; f < FLOPS; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fma(v[i], m[(i + f) & 3], (double)((c >> i) & 1u));
    if (BARRIER)
    {
      const unsigned long long mine = __ballot(v[0] < 1e-77);
      if (lane == 0) ballots[s & 1u][rate] = mine;
      __syncthreads();
      const unsigned long long all = ballots[s & 1u][0] & ballots[s & 1u][1] & ballots[s & 1u][2] & ballots[s & 1u][3];
      if ((all >> lane) & 1ull) v[0] *= 1e77;
    }
    double *cp = st[s].clv;
    const unsigned cb = st[s].clv_bytes;
    const __amdgpu_buffer_rsrc_t rc = rsrc(cp, cb);
#pragma unroll
    for (int d = 0; d < DIS_ST; ++d) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, extra), rsrc(cp, 0u), voff + d * 512u, 0, 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v[i] + extra), rc, voff + i * 512u, 0, 2);
  }
}

int main()
{
  const unsigned entries = 100000, tiles = (entries + 63) / 64, steps = 62;
  std::vector<Step> hs(steps + 1);
  double *mat;
  CK(hipMalloc(&mat, 64 * 8 * 64));
  std::vector<double> hm(64 * 64, 1.0000001);
  CK(hipMemcpy(mat, hm.data(), hm.size() * 8, hipMemcpyHostToDevice));
  for (unsigned i = 0; i <= steps; ++i)
  {
    CK(hipMalloc(&hs[i].clv, (size_t)tiles * 1024 * 8));
    unsigned char *t;
    CK(hipMalloc(&t, tiles * 64));
    CK(hipMemset(t, i & 15, tiles * 64));
    hs[i].tip = t;
    hs[i].mat = mat + (i % 60) * 64;
    hs[i].clv_bytes = tiles * 8192;
    hs[i].tip_bytes = i < steps ? tiles * 64 : 0;
  }
  Step *ds;
  CK(hipMalloc(&ds, hs.size() * sizeof(Step)));
  CK(hipMemcpy(ds, hs.data(), hs.size() * sizeof(Step), hipMemcpyHostToDevice));
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
    printf("%-44s %8.1f us  %6.2f TB/s  (%.2f us per step)\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, ms / reps * 1e3 / steps);
  };
  run("base (descriptors via s_load, 1 flop round)", [&] { k_steps<0, 0, false, false, 1><<<tiles, 256>>>(ds, steps, entries); });
  run("+ 11 disabled loads", [&] { k_steps<11, 0, false, false, 1><<<tiles, 256>>>(ds, steps, entries); });
  run("+ 5 disabled stores", [&] { k_steps<0, 5, false, false, 1><<<tiles, 256>>>(ds, steps, entries); });
  run("+ barrier", [&] { k_steps<0, 0, true, false, 1><<<tiles, 256>>>(ds, steps, entries); });
  run("+ matrices via s_load", [&] { k_steps<0, 0, false, true, 1><<<tiles, 256>>>(ds, steps, entries); });
  run("+ 8 flop rounds", [&] { k_steps<0, 0, false, false, 8><<<tiles, 256>>>(ds, steps, entries); });
  run("all of them", [&] { k_steps<11, 5, true, true, 8><<<tiles, 256>>>(ds, steps, entries); });
  run("all but disabled ops", [&] { k_steps<0, 0, true, true, 8><<<tiles, 256>>>(ds, steps, entries); });
  run("+ 8 loads under an empty lane mask", [&] { k_steps<8, 0, false, false, 1, true><<<tiles, 256>>>(ds, steps, entries); });
  run("all, loads under an empty lane mask", [&] { k_steps<8, 5, true, true, 8, true><<<tiles, 256>>>(ds, steps, entries); });
  run("all but barrier", [&] { k_steps<11, 5, false, true, 8><<<tiles, 256>>>(ds, steps, entries); });
  return 0;
}
