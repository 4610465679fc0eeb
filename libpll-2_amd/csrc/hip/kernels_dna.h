// kernels_dna.h - 4-state x 4-rate kernels (the headline DNA/Gamma4 configuration).
//
// Thread mapping: one lane per (entry, rate). The four lanes of a DPP quad hold the four rate
// categories of one site; lane t of the grid touches bytes [32 t, 32 t + 32) of a CLV, so every
// wave-wide load/store is 64 x 16 B fully coalesced (two global_load_dwordx4 per CLV per lane).
// The two 4x4 transition matrices of the lane's rate category live in registers for the whole
// kernel (64 VGPRs); per-site reductions (scaling test, rate mixing) are DPP quad exchanges.
//
// Algorithmic traffic per site-CLV-update: ii 3*128 B (+12 B scalers), ti 2*128+1, tt 128+2.
// Arithmetic: src/core_partials.c:709-764 (ii), :290-351 (ti), :1032-1070 + :68-79 (tt, the
// lookup table is replaced by 8 masked adds in registers), :819-879 (repeats: gathers).
#pragma once
#include "kernels_common.h"

struct Mat4
{
  double m[4][4]; // m[i][j]: parent state i, child state j
};

// PT layout for SPT == 4: pt[(k*4 + j)*4 + i]
__device__ __forceinline__ void load_mat4(const double *__restrict__ pt, unsigned k, Mat4 &M)
{
  const double4 *p = reinterpret_cast<const double4 *>(pt) + k * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j)
  {
    double4 v = p[j];
    M.m[0][j] = v.x;
    M.m[1][j] = v.y;
    M.m[2][j] = v.z;
    M.m[3][j] = v.w;
  }
}

__device__ __forceinline__ double4 matvec4(const Mat4 &M, const double4 x)
{
  double4 r;
  r.x = fma(M.m[0][3], x.w, fma(M.m[0][2], x.z, fma(M.m[0][1], x.y, M.m[0][0] * x.x)));
  r.y = fma(M.m[1][3], x.w, fma(M.m[1][2], x.z, fma(M.m[1][1], x.y, M.m[1][0] * x.x)));
  r.z = fma(M.m[2][3], x.w, fma(M.m[2][2], x.z, fma(M.m[2][1], x.y, M.m[2][0] * x.x)));
  r.w = fma(M.m[3][3], x.w, fma(M.m[3][2], x.z, fma(M.m[3][1], x.y, M.m[3][0] * x.x)));
  return r;
}

// row sums over the states present in a 4-bit tip code (src/core_partials.c:304-312)
__device__ __forceinline__ double4 masksum4(const Mat4 &M, unsigned code)
{
  double4 x;
  x.x = (code & 1u) ? 1.0 : 0.0;
  x.y = (code & 2u) ? 1.0 : 0.0;
  x.z = (code & 4u) ? 1.0 : 0.0;
  x.w = (code & 8u) ? 1.0 : 0.0;
  return matvec4(M, x);
}

template <bool LTIP, bool RTIP, bool GATHER>
__device__ __forceinline__ void dna_site(const DevOp &op, const Mat4 &L, const Mat4 &R, unsigned n,
                                         unsigned k, int scale_mode)
{
  const double4 *__restrict__ left = reinterpret_cast<const double4 *>(op.left);
  const double4 *__restrict__ right = reinterpret_cast<const double4 *>(op.right);
  double4 *__restrict__ parent = reinterpret_cast<double4 *>(op.parent);

  unsigned le = n, re = n;
  if (GATHER)
  {
    const unsigned site = op.id_site ? op.id_site[n] : n;
    le = op.lsid ? op.lsid[site] : site;
    re = op.rsid ? op.rsid[site] : site;
  }
  double4 a, b;
  if (LTIP)
    a = masksum4(L, op.ltip[le]);
  else
    a = matvec4(L, left[(size_t)le * 4 + k]);
  if (RTIP)
    b = masksum4(R, op.rtip[re]);
  else
    b = matvec4(R, right[(size_t)re * 4 + k]);
  double4 v;
  v.x = a.x * b.x;
  v.y = a.y * b.y;
  v.z = a.z * b.z;
  v.w = a.w * b.w;

  if (scale_mode)
  {
    int small = (v.x < PLLGPU_SCALE_THRESHOLD) & (v.y < PLLGPU_SCALE_THRESHOLD) &
                (v.z < PLLGPU_SCALE_THRESHOLD) & (v.w < PLLGPU_SCALE_THRESHOLD);
    if (scale_mode == 1)
    {
      // all 16 entries of the site: AND across the quad (src/core_partials.c:754-763)
      small &= dpp_i32<0xB1>(small);
      small &= dpp_i32<0x4E>(small);
      if (k == 0)
        op.pscaler[n] = (op.lscaler ? op.lscaler[le] : 0u) + (op.rscaler ? op.rscaler[re] : 0u) +
                        (unsigned)small;
    }
    else
    {
      op.pscaler[(size_t)n * 4 + k] = (op.lscaler ? op.lscaler[(size_t)le * 4 + k] : 0u) +
                                      (op.rscaler ? op.rscaler[(size_t)re * 4 + k] : 0u) +
                                      (unsigned)small;
    }
    if (small)
    {
      v.x *= PLLGPU_SCALE_FACTOR;
      v.y *= PLLGPU_SCALE_FACTOR;
      v.z *= PLLGPU_SCALE_FACTOR;
      v.w *= PLLGPU_SCALE_FACTOR;
    }
  }
  parent[(size_t)n * 4 + k] = v;
}

// grid: x = entry chunks of `epb` (multiple of 64), y = op within the pack
template <bool LTIP, bool RTIP, bool GATHER>
__global__ __launch_bounds__(256) void k_partials_dna(const OpPack pack, int scale_mode, unsigned epb)
{
  const DevOp &op = pack.ops[blockIdx.y];
  const unsigned begin = blockIdx.x * epb;
  if (begin >= op.entries) return;
  const unsigned end = min(op.entries, begin + epb);
  const unsigned k = threadIdx.x & 3u;
  const unsigned q = threadIdx.x >> 2;
  const int mode = op.pscaler ? scale_mode : 0;

  Mat4 L, R;
  load_mat4(op.lmat, k, L);
  load_mat4(op.rmat, k, R);

  unsigned n = begin + q;
  // two sites per trip: twice the loads in flight per lane
  for (; n + 64 < end; n += 128)
  {
    dna_site<LTIP, RTIP, GATHER>(op, L, R, n, k, mode);
    dna_site<LTIP, RTIP, GATHER>(op, L, R, n + 64, k, mode);
  }
  if (n < end) dna_site<LTIP, RTIP, GATHER>(op, L, R, n, k, mode);
}

// ------------------------------------------------------------------------------------------------
// edge / root log-likelihood, 4 states x 4 rates. Same quad mapping; the quad's rate terms are
// mixed with two DPP adds, lane k==0 finishes the site (log, scaler undo, pattern weight), then a
// wave shuffle tree and an LDS step give one partial sum per block (fixed order: deterministic).
// Arithmetic: src/core_likelihood.c:1388-1490 (ii), :470-578 (ti 4x4), :1077-1183 (repeats),
// :163-207 (root).
template <bool CTIP, bool GATHER>
__global__ __launch_bounds__(256) void k_edge_dna(const DevEdge e, unsigned spb /* sites per block */)
{
  __shared__ double wsum[4];
  const unsigned k = threadIdx.x & 3u;
  const unsigned q = threadIdx.x >> 2;
  const unsigned begin = blockIdx.x * spb;
  const unsigned end = min(e.sites, begin + spb);

  Mat4 P;
  if (!e.is_root) load_mat4(e.mat, k, P);
  const unsigned fi = e.fidx[k];
  const double4 pi = reinterpret_cast<const double4 *>(e.freqs)[fi];
  const double w = e.rate_weights[k];
  const double pinv = e.prop_invar ? e.prop_invar[fi] : 0.0;
  const double4 *__restrict__ parent = reinterpret_cast<const double4 *>(e.parent);
  const double4 *__restrict__ child = reinterpret_cast<const double4 *>(e.child);

  double acc = 0.0;
  for (unsigned n = begin + q; n < end; n += 64)
  {
    unsigned pe = n, ce = n;
    if (GATHER)
    {
      pe = e.psid ? e.psid[n] : n;
      ce = e.csid ? e.csid[n] : n;
    }
    const double4 x = parent[(size_t)pe * 4 + k];
    double4 tb;
    if (e.is_root)
      tb = make_double4(1.0, 1.0, 1.0, 1.0);
    else if (CTIP)
      tb = masksum4(P, e.ctip[ce]);
    else
      tb = matvec4(P, child[(size_t)ce * 4 + k]);
    double t = fma(x.w * pi.w, tb.w, fma(x.z * pi.z, tb.z, fma(x.y * pi.y, tb.y, (x.x * pi.x) * tb.x)));

    unsigned scal;
    if (e.per_rate)
    {
      unsigned rs = (e.pscaler ? e.pscaler[(size_t)pe * 4 + k] : 0u) +
                    (e.cscaler ? e.cscaler[(size_t)ce * 4 + k] : 0u);
      unsigned mn = min(rs, (unsigned)dpp_i32<0xB1>((int)rs));
      mn = min(mn, (unsigned)dpp_i32<0x4E>((int)mn));
      const unsigned ex = min(rs - mn, PLLGPU_RATE_MAXDIFF);
      if (ex) t *= minlh(ex);
      scal = mn;
    }
    else
      scal = (e.pscaler ? e.pscaler[pe] : 0u) + (e.cscaler ? e.cscaler[ce] : 0u);

    double ta, ti = 0.0;
    if (pinv > 0.0)
    {
      ta = w * t * (1.0 - pinv);
      const int inv = e.invariant ? e.invariant[n] : -1;
      if (inv >= 0)
      {
        const double f = inv == 0 ? pi.x : inv == 1 ? pi.y : inv == 2 ? pi.z : pi.w;
        ti = w * f * pinv;
      }
    }
    else
      ta = t * w;
    // mix the four categories in category order: ((t0 + t1) + (t2 + t3))
    ta += dpp_f64_xor1(ta);
    ta += dpp_f64_xor2(ta);
    ti += dpp_f64_xor1(ti);
    ti += dpp_f64_xor2(ti);
    if (k == 0)
    {
      double site = finish_site(ta, ti, scal, e.is_root) * (double)e.pattern_weights[n];
      if (e.persite) e.persite[n] = site;
      acc += site;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) e.block_sums[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// final fixed-order sum of the per-block partials (one block)
__global__ __launch_bounds__(256) void k_sum_blocks(const double *__restrict__ part, unsigned count,
                                                    double *__restrict__ out)
{
  __shared__ double wsum[4];
  double acc = 0.0;
  for (unsigned i = threadIdx.x; i < count; i += 256) acc += part[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *out = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
