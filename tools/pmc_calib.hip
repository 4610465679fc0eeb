// pmc_calib.hip - known-byte-count kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// for the access shapes the likelihood kernels use (8 B per lane, dense 512 B per wave instruction,
// plain and non-temporal). MI355X_MICROARCH.md section HBM: FETCH_SIZE reads 1/2 for 16 B/lane
// streams; other widths must be calibrated before an absolute is trusted.
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o gpurun_out/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <bool NT>
__global__ __launch_bounds__(256) void k_read8(const double *__restrict__ src, double *__restrict__ sink, size_t n)
{
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    acc += NT ? __builtin_nontemporal_load(src + i) : src[i];
  if (acc == 12345.678) sink[0] = acc; // never true: keeps the loads alive
}

__global__ __launch_bounds__(256) void k_write8(double *__restrict__ dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = 1.0;
}

__global__ __launch_bounds__(256) void k_read16(const double2 *__restrict__ src, double *__restrict__ sink, size_t n)
{
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
  {
    double2 v = src[i];
    acc += v.x + v.y;
  }
  if (acc == 12345.678) sink[0] = acc;
}

int main()
{
  const size_t bytes = (size_t)1 << 30; // 1 GiB, well past the 256 MiB Infinity Cache
  const size_t n = bytes / 8;
  double *a, *b;
  hipMalloc(&a, bytes);
  hipMalloc(&b, bytes);
  hipMemset(a, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep)
  {
    hipLaunchKernelGGL(k_read8<false>, dim3(4096), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_read8<true>, dim3(4096), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const double2 *)a, b, n / 2);
    hipLaunchKernelGGL(k_write8, dim3(4096), dim3(256), 0, 0, b, n);
  }
  hipDeviceSynchronize();
  printf("each kernel moves %zu bytes\n", bytes);
  return 0;
}
