// mfma61_probe.hip - how busy ONE or TWO waves per SIMD keep the fp64 matrix pipe in the contraction loop of the
// 33..64-state CLV update (kernels_mfma.h): A operand = a 4 x 4 block of P from LDS (one ds_read_b64 per block),
// B operand = x of NSG groups of 16 sites in registers, NGI x NSG accumulators. Reports TFLOP/s of the MFMAs issued
// and the clock the chip holds (s_memtime / s_memrealtime) on random data.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma61_probe.hip -o /tmp/mfma61_probe && /tmp/mfma61_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr unsigned kFrag = 17;

typedef double probe_d2 __attribute__((ext_vector_type(2)));

// HBM = 1: every pass also streams NGJ x 1 KB per wave in (the next x, non-temporal) and NGI / 2 x 1 KB out - the memory
// traffic of the real kernel (kernels_mfma_wide.h) beside its MFMAs, to see what clock the chip holds with both
template <int NSG, int NGJ, int NGI, int MINW, int HBM = 0>
__global__ __launch_bounds__(256, MINW) void k_loop(const double *__restrict__ pm, const double *__restrict__ xin, double *__restrict__ out,
                                                    int iters, unsigned long long *__restrict__ stamps, const double *__restrict__ stream_in = nullptr,
                                                    double *__restrict__ stream_out = nullptr, size_t stream_doubles = 0)
{
  extern __shared__ double lds[];
  for (unsigned t = threadIdx.x; t < 16u * 16u * kFrag; t += 256u) lds[t] = pm[t];
  __syncthreads();
  const unsigned lane = threadIdx.x & 63u, row = lane >> 4;
  const unsigned fragoff = row * 4u + (lane & 3u);
  double x[NGJ][NSG], D[NGI][NSG];
#pragma unroll
  for (int jg = 0; jg < NGJ; ++jg)
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) x[jg][sg] = xin[((blockIdx.x * 256u + threadIdx.x) * NGJ + jg) * NSG + sg];
#pragma unroll
  for (int ig = 0; ig < NGI; ++ig)
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) D[ig][sg] = 0.0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  // the A fragments of contraction group jg + 1 are requested while group jg multiplies: one ds_read_b64 then NSG
  // MFMAs, sixteen times (sched_group_barrier keeps that interleave; without it the compiler hoists every read of
  // the unrolled pass and spills)
  double a[2][NGI];
#pragma unroll
  for (int ig = 0; ig < NGI; ++ig) a[0][ig] = lds[(ig * 16 + 0) * kFrag + fragoff];
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int jg = 0; jg < NGJ; ++jg)
    {
      const int nj = (jg + 1) % NGJ;
#pragma unroll
      for (int ig = 0; ig < NGI; ++ig)
      {
        a[(jg + 1) & 1][ig] = lds[(ig * 16 + nj) * kFrag + fragoff];
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg) D[ig][sg] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[jg & 1][ig], x[jg][sg], D[ig][sg], 0, 0, 0);
      }
#pragma unroll
      for (int ig = 0; ig < NGI; ++ig)
      {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
        __builtin_amdgcn_sched_group_barrier(0x008, NSG, 0); // NSG MFMAs
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (HBM && NSG == 2)
    {
      // one pass of the real kernel: NGJ row groups of 64 lanes x 16 bytes in, NGI / 2 out, each wave its own stream
      const size_t wave_id = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6);
      const size_t pass = (wave_id * (size_t)iters + (size_t)it) % (stream_doubles / (128u * 16u));
      const double *src = stream_in + pass * (128u * 16u) + (threadIdx.x & 63u) * 2u;
      double *dst = stream_out + pass * (128u * 16u) + (threadIdx.x & 63u) * 2u;
#pragma unroll
      for (int jg = 0; jg < NGJ; ++jg)
      {
        const probe_d2 v = __builtin_nontemporal_load((const probe_d2 *)(src + jg * 128));
        x[jg][0] = v.x * 1e-3 + x[jg][0] * 0.5;
        x[jg][NSG - 1] = v.y * 1e-3 + x[jg][NSG - 1] * 0.5;
      }
#pragma unroll
      for (int ig = 0; ig < NGI; ig += 2) *(probe_d2 *)(dst + (ig / 2) * 128) = probe_d2{D[ig][0], D[ig + 1][NSG - 1]};
    }
    // keep the values bounded and the loop honest: x changes sign pattern from pass to pass
#pragma unroll
    for (int jg = 0; jg < NGJ; ++jg)
#pragma unroll
      for (int sg = 0; sg < NSG; ++sg) x[jg][sg] = -x[jg][sg];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0.0;
#pragma unroll
  for (int ig = 0; ig < NGI; ++ig)
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) s += D[ig][sg];
  out[blockIdx.x * 256u + threadIdx.x] = s;
  if (threadIdx.x == 0)
  {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int NSG, int NGJ, int NGI, int MINW, int HBM = 0>
static void run(const char *what, int wg_per_cu, const double *pm, const double *xin, double *out, unsigned long long *stamps,
                const double *sin = nullptr, double *sout = nullptr, size_t sdoubles = 0)
{
  const int blocks = 256 * wg_per_cu;
  if (blocks > 1024) return; // the buffers in main() are sized for 1024 blocks
  const int iters = 4000 / NSG;
  const size_t lds = 16 * 16 * kFrag * sizeof(double);
  hipFuncSetAttribute((const void *)k_loop<NSG, NGJ, NGI, MINW, HBM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep)
  {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_loop<NSG, NGJ, NGI, MINW, HBM>), dim3(blocks), dim3(256), lds, 0, pm, xin, out, iters, stamps, sin, sout, sdoubles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int b = 0; b < blocks; ++b) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
  std::sort(clk.begin(), clk.end());
  const double mfmas = (double)blocks * 4 * iters * NGJ * NGI * NSG;
  const double cyc_per_mfma = (double)h[0] / ((double)iters * NGJ * NGI * NSG) / (wg_per_cu > 1 ? 1.0 : 1.0);
  const double gb = HBM ? (double)blocks * 4 * iters * (NGJ + NGI / 2) * 1024.0 / 1e9 : 0.0;
  if (HBM) printf("  (+ %.0f GB/s of streaming beside the MFMAs) ", gb / (best * 1e-3));
  printf("%-44s %6.2f TFLOP/s  %.3f ms  clock %.0f MHz  wave cycles per own MFMA %.2f (x%d waves per SIMD)\n", what,
         mfmas * 512 / (best * 1e-3) / 1e12, best, clk[clk.size() / 2], cyc_per_mfma, wg_per_cu);
  fflush(stdout);
}

int main()
{
  double *pm, *xin, *out;
  unsigned long long *stamps;
  const int kMaxBlocks = 1024; // 4 workgroups per CU at most (run<>'s wg_per_cu)
  const size_t nx = (size_t)kMaxBlocks * 256 * 16 * 4;
  hipMalloc(&pm, 16 * 16 * kFrag * 8);
  hipMalloc(&xin, nx * 8);
  hipMalloc(&out, (size_t)kMaxBlocks * 256 * 8);
  hipMalloc(&stamps, 2 * (size_t)kMaxBlocks * 8);
  std::vector<double> hp(16 * 16 * kFrag), hx(nx);
  srand(7);
  for (auto &v : hp) v = (rand() / (double)RAND_MAX) / 64.0;
  for (auto &v : hx) v = rand() / (double)RAND_MAX * 2.0 - 1.0;
  hipMemcpy(pm, hp.data(), hp.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(xin, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
  run<2, 16, 16, 2>("32 sites, 16x16 groups, 2 waves / SIMD", 2, pm, xin, out, stamps);
  run<2, 16, 16, 2>("32 sites, 16x16 groups, 1 wave / SIMD", 1, pm, xin, out, stamps);
  run<4, 16, 16, 1>("64 sites, 16x16 groups, 1 wave / SIMD", 1, pm, xin, out, stamps);
  run<4, 15, 15, 1>("64 sites, 15x15 groups, 1 wave / SIMD", 1, pm, xin, out, stamps);
  run<3, 16, 16, 1>("48 sites, 16x16 groups, 1 wave / SIMD", 1, pm, xin, out, stamps);
  run<2, 15, 15, 2>("32 sites, 15x15 groups, 2 waves / SIMD", 2, pm, xin, out, stamps);
  run<1, 16, 16, 4>("16 sites, 16x16 groups, 4 waves / SIMD", 4, pm, xin, out, stamps);
  {
    // the real kernel's mix: 15 x 16 groups, two waves per SIMD, plus its HBM streams (1 GiB each way: past the Infinity Cache)
    const size_t sd = (size_t)1 << 27;
    double *sin, *sout;
    hipMalloc(&sin, sd * 8);
    hipMalloc(&sout, sd * 8);
    hipMemset(sin, 0x3c, sd * 8);
    run<2, 15, 16, 2, 1>("32 sites, 15x16 groups, 2 waves / SIMD + HBM", 2, pm, xin, out, stamps, sin, sout, sd);
    run<2, 15, 16, 2, 0>("32 sites, 15x16 groups, 2 waves / SIMD", 2, pm, xin, out, stamps);
    run<2, 15, 16, 2, 1>("32 sites, 15x16 groups, 2 waves / SIMD + HBM", 2, pm, xin, out, stamps, sin, sout, sd);
  }
  return 0;
}
