// mfma_f64_probe.hip - (1) issue rate of v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 vs v_fma_f64
// on gfx950, (2) the lane <-> element maps of the 16x16x4 f64 form, checked with exact integers.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma16(double *out, int iters, long long *cyc)
{
  d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double a = threadIdx.x * 0.001 + 1.0, b = threadIdx.x * 0.002 + 0.5;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i)
  {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_mfma4(double *out, int iters, long long *cyc)
{
  double acc[4] = {0, 0, 0, 0};
  double a = threadIdx.x * 0.001 + 1.0, b = threadIdx.x * 0.002 + 0.5;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i)
  {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[u], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_fma(double *out, int iters, long long *cyc, const double *coef)
{
  double acc[16];
  for (int u = 0; u < 16; ++u) acc[u] = u;
  double x = threadIdx.x * 0.001 + 1.0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i)
  {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = fma(acc[u], 0.999999, x);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int u = 0; u < 16; ++u) s += acc[u];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[2] = t1 - t0;
}

// layout check: D = A(16x4) * B(4x16) with A[i][k] = 100 i + k, B[k][j] = 10 k + j + 1 (asymmetric)
__global__ void k_layout(double *D /*16x16 row-major*/)
{
  const int l = threadIdx.x;
  const int i = l & 15, k = l >> 4;       // A: row l&15, k = l>>4
  const double a = 100.0 * i + k;
  const int j = l & 15;                   // B: k = l>>4, col l&15
  const double b = 10.0 * k + j + 1;
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r]; // row = (lane>>4) + 4 reg, col = lane&15
}

int main()
{
  double *out;
  long long *cyc;
  hipMalloc(&out, 1 << 24);
  hipMalloc(&cyc, 64);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep)
  {
    hipLaunchKernelGGL(k_mfma16, dim3(1024), dim3(256), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL(k_mfma4, dim3(1024), dim3(256), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL(k_fma, dim3(1024), dim3(256), 0, 0, out, iters, cyc, (const double *)out);
  }
  hipDeviceSynchronize();
  long long h[3];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  // per wave: s_memtime ticks at 100 MHz? report raw and per-instruction
  printf("memtime ticks: mfma16x16x4 %lld for %d instr/wave, mfma4x4x4 %lld for %d, fma %lld for %d\n", h[0], iters * 4, h[1], iters * 4, h[2], iters * 16);
  // wall-clock throughput with events
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms;
  const int blocks = 256 * 8; // 8 blocks of 4 waves per CU
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_mfma16, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("mfma_f64_16x16x4: %.2f TFLOP/s (%.3f ms)\n", (double)blocks * 4 * iters * 4 * 2048 / (ms * 1e-3) / 1e12, ms);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_mfma4, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("mfma_f64_4x4x4_4b: %.2f TFLOP/s (%.3f ms)\n", (double)blocks * 4 * iters * 4 * 512 / (ms * 1e-3) / 1e12, ms);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters, cyc, (const double *)out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("v_fma_f64: %.2f TFLOP/s (%.3f ms)\n", (double)blocks * 4 * iters * 16 * 128 / (ms * 1e-3) / 1e12, ms);

  double *D;
  hipMalloc(&D, 256 * 8);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, D);
  std::vector<double> hD(256);
  hipMemcpy(hD.data(), D, 256 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j)
    {
      double ref = 0;
      for (int k = 0; k < 4; ++k) ref += (100.0 * i + k) * (10.0 * k + j + 1);
      if (hD[i * 16 + j] != ref) ++bad;
    }
  printf("layout check (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15): %d mismatches of 256\n", bad);
  return 0;
}
