// mfma_4x4_layout.hip - empirical lane maps of v_mfma_f64_4x4x4_4b_f64 on gfx950:
// for every (A lane la, B lane lb) set A=1 only in la, B=1 only in lb and record which D lanes
// become non-zero. Prints the inferred (block, row, k) / (block, k, col) / (block,row,col) maps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *hit /*[64][64] -> D lane or -1*/)
{
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb)
    {
      double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      if (d != 0.0) hit[la * 64 + lb] = l;
    }
}
int main()
{
  int *d;
  hipMalloc(&d, 4096 * 4);
  hipMemset(d, 0xFF, 4096 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<int> h(4096);
  hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
  // for each A lane list the B lanes it pairs with and the D lane
  for (int la = 0; la < 64; ++la)
  {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] >= 0) printf(" (B%2d->D%2d)", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
