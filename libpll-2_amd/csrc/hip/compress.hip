// compress.hip - site-pattern compression on the device (second half;
// src/compress.c:171-410).
//
// The reference transposes the alignment into columns, sorts them with a randomised multikey
// quicksort (comparison of encoded characters as signed char) and merges equal neighbours. The
// result does not depend on the pivots: unique columns in lexicographic order, their multiplicities,
// and for every original site the index of its pattern. The same result on the device:
//   1. a stable LSD radix sort of the site indices, most significant key = first sequence. Several
//      sequences are packed into one 64-bit key (bits per character from the largest code), one
//      rocPRIM radix_sort_pairs per packed group, from the last group to the first;
//   2. head[i] = the column at sorted position i differs from its predecessor (full comparison);
//   3. inclusive scan of the heads numbers the patterns; run lengths are the weights.
// Integer/byte work: the outputs are identical to the reference's (tests/test_gpu_compress.py).
#include "../../../include/pll_amd_device.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

namespace
{
thread_local char g_cerr[256];

int cfail(const char *what, hipError_t e)
{
  snprintf(g_cerr, sizeof g_cerr, "%s: %s", what, hipGetErrorString(e));
  return PLLGPU_ERUNTIME;
}

#define CTRY(call)                                 \
  do                                               \
  {                                                \
    hipError_t e_ = (call);                        \
    if (e_ != hipSuccess) { rc = cfail(#call, e_); goto done; } \
  } while (0)

__global__ void k_iota(unsigned *p, unsigned n)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

// key of the site at sorted position i for sequences [t0, t0 + nt): first sequence most significant;
// characters as signed char (the reference compares `char`), hence the sign flip when needed
__global__ void k_pack_keys(const unsigned char *__restrict__ enc, const unsigned *__restrict__ perm, unsigned long long *__restrict__ keys,
                            unsigned length, unsigned t0, unsigned nt, unsigned bits, unsigned flip)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= length) return;
  const unsigned site = perm[i];
  unsigned long long k = 0;
  for (unsigned b = 0; b < nt; ++b)
    k = (k << bits) | (unsigned long long)(enc[(size_t)(t0 + b) * length + site] ^ flip);
  keys[i] = k;
}

__global__ void k_heads(const unsigned char *__restrict__ enc, const unsigned *__restrict__ perm, unsigned *__restrict__ head,
                        unsigned length, unsigned count)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= length) return;
  unsigned h = 1;
  if (i > 0)
  {
    const unsigned a = perm[i], b = perm[i - 1];
    h = 0;
    for (unsigned t = 0; t < count; ++t)
      if (enc[(size_t)t * length + a] != enc[(size_t)t * length + b])
      {
        h = 1;
        break;
      }
  }
  head[i] = h;
}

// idx = inclusive scan of head (1-based pattern number at every sorted position)
__global__ void k_finish(const unsigned *__restrict__ perm, const unsigned *__restrict__ head, const unsigned *__restrict__ idx,
                         unsigned *__restrict__ start, unsigned *__restrict__ site_map, unsigned length)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= length) return;
  const unsigned p = idx[i] - 1u;
  if (head[i]) start[p] = i;
  if (site_map) site_map[perm[i]] = p;
  if (i == length - 1) start[idx[i]] = length; // sentinel behind the last pattern
}

__global__ void k_emit(const unsigned char *__restrict__ enc, const unsigned *__restrict__ perm, const unsigned *__restrict__ start,
                       unsigned char *__restrict__ comp, unsigned *__restrict__ weights, unsigned length, unsigned count, unsigned patterns)
{
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= patterns) return;
  const unsigned s = start[p];
  weights[p] = start[p + 1] - s;
  const unsigned site = perm[s];
  for (unsigned t = 0; t < count; ++t) comp[(size_t)t * patterns + p] = enc[(size_t)t * length + site];
}
} // namespace

extern "C" const char *pllgpu_compress_last_error(void) { return g_cerr; }

extern "C" int pllgpu_compress_patterns(const unsigned char *encoded, unsigned count, unsigned length, unsigned char *compressed,
                                        unsigned *weights, unsigned *site_pattern_map, unsigned *patterns_out, int device)
{
  int rc = 0, ndev = 0;
  g_cerr[0] = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
  {
    snprintf(g_cerr, sizeof g_cerr, "no HIP device visible");
    return PLLGPU_ENODEVICE;
  }
  int prev_device = -1;
  if (hipGetDevice(&prev_device) != hipSuccess) prev_device = -1;
  if (device < 0)
  {
    // PLL_AMD_DEVICE=<n>, else (unset or "auto") the calling thread's current device - like pllgpu_create
    const char *env = getenv("PLL_AMD_DEVICE");
    device = (env && strcmp(env, "auto") != 0) ? atoi(env) : (prev_device >= 0 ? prev_device : 0);
  }
  if (device >= ndev || !count || !length)
  {
    snprintf(g_cerr, sizeof g_cerr, "invalid argument (device %d of %d, %u sequences, %u sites)", device, ndev, count, length);
    return PLLGPU_EINVAL;
  }
  hipStream_t st = nullptr;
  unsigned char *d_enc = nullptr, *d_comp = nullptr;
  unsigned *d_perm = nullptr, *d_perm2 = nullptr, *d_head = nullptr, *d_idx = nullptr, *d_start = nullptr, *d_map = nullptr, *d_w = nullptr;
  unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
  void *d_tmp = nullptr;
  size_t tmp_bytes = 0, scan_bytes = 0;
  unsigned patterns = 0;
  const unsigned nb = (length + 255) / 256;

  // bits per character and whether any code has the sign bit set
  unsigned maxc = 0;
  for (size_t i = 0; i < (size_t)count * length; ++i)
    if (encoded[i] > maxc) maxc = encoded[i];
  const unsigned flip = maxc >= 128 ? 0x80u : 0u;
  unsigned bits = 1;
  while (bits < 8 && (flip ? 255u : maxc) >> bits) ++bits;
  const unsigned per_key = 64 / bits;

  CTRY(hipSetDevice(device));
  CTRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CTRY(hipMalloc(&d_enc, (size_t)count * length));
  CTRY(hipMalloc(&d_comp, (size_t)count * length));
  CTRY(hipMalloc(&d_perm, sizeof(unsigned) * length));
  CTRY(hipMalloc(&d_perm2, sizeof(unsigned) * length));
  CTRY(hipMalloc(&d_head, sizeof(unsigned) * length));
  CTRY(hipMalloc(&d_idx, sizeof(unsigned) * length));
  CTRY(hipMalloc(&d_start, sizeof(unsigned) * ((size_t)length + 1)));
  CTRY(hipMalloc(&d_map, sizeof(unsigned) * length));
  CTRY(hipMalloc(&d_w, sizeof(unsigned) * length));
  CTRY(hipMalloc(&d_keys, sizeof(unsigned long long) * length));
  CTRY(hipMalloc(&d_keys2, sizeof(unsigned long long) * length));
  CTRY(hipMemcpyAsync(d_enc, encoded, (size_t)count * length, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_iota, dim3(nb), dim3(256), 0, st, d_perm, length);

  CTRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, d_keys2, d_perm, d_perm2, (size_t)length, 0u, 64u, st));
  CTRY(rocprim::inclusive_scan(nullptr, scan_bytes, d_head, d_idx, (size_t)length, rocprim::plus<unsigned>(), st));
  if (scan_bytes > tmp_bytes) tmp_bytes = scan_bytes;
  CTRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8));

  // LSD: last group of sequences first
  {
    const unsigned groups = (count + per_key - 1) / per_key;
    for (unsigned g = groups; g-- > 0;)
    {
      const unsigned t0 = g * per_key, nt = (t0 + per_key <= count) ? per_key : count - t0;
      hipLaunchKernelGGL(k_pack_keys, dim3(nb), dim3(256), 0, st, d_enc, d_perm, d_keys, length, t0, nt, bits, flip);
      size_t tb = tmp_bytes;
      CTRY(rocprim::radix_sort_pairs(d_tmp, tb, d_keys, d_keys2, d_perm, d_perm2, (size_t)length, 0u, nt * bits, st));
      unsigned *sw = d_perm;
      d_perm = d_perm2;
      d_perm2 = sw;
    }
  }
  hipLaunchKernelGGL(k_heads, dim3(nb), dim3(256), 0, st, d_enc, d_perm, d_head, length, count);
  {
    size_t tb = tmp_bytes;
    CTRY(rocprim::inclusive_scan(d_tmp, tb, d_head, d_idx, (size_t)length, rocprim::plus<unsigned>(), st));
  }
  hipLaunchKernelGGL(k_finish, dim3(nb), dim3(256), 0, st, d_perm, d_head, d_idx, d_start, site_pattern_map ? d_map : nullptr, length);
  CTRY(hipMemcpyAsync(&patterns, d_idx + (length - 1), sizeof(unsigned), hipMemcpyDeviceToHost, st));
  CTRY(hipStreamSynchronize(st));
  hipLaunchKernelGGL(k_emit, dim3((patterns + 255) / 256), dim3(256), 0, st, d_enc, d_perm, d_start, d_comp, d_w, length, count, patterns);
  CTRY(hipGetLastError());
  CTRY(hipMemcpyAsync(weights, d_w, sizeof(unsigned) * patterns, hipMemcpyDeviceToHost, st));
  CTRY(hipMemcpyAsync(compressed, d_comp, (size_t)count * patterns, hipMemcpyDeviceToHost, st));
  if (site_pattern_map) CTRY(hipMemcpyAsync(site_pattern_map, d_map, sizeof(unsigned) * length, hipMemcpyDeviceToHost, st));
  CTRY(hipStreamSynchronize(st));
  *patterns_out = patterns;
done:
  (void)hipFree(d_enc);
  (void)hipFree(d_comp);
  (void)hipFree(d_perm);
  (void)hipFree(d_perm2);
  (void)hipFree(d_head);
  (void)hipFree(d_idx);
  (void)hipFree(d_start);
  (void)hipFree(d_map);
  (void)hipFree(d_w);
  (void)hipFree(d_keys);
  (void)hipFree(d_keys2);
  (void)hipFree(d_tmp);
  if (st) (void)hipStreamDestroy(st);
  if (prev_device >= 0 && prev_device != device) (void)hipSetDevice(prev_device); // leave the thread's device as found
  return rc;
}
