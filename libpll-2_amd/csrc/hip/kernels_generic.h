// kernels_generic.h - any number of states (<= 64) and rate categories.
//
// Thread mapping: a 256-thread workgroup owns a tile of 64 site entries; LANE = ENTRY in every
// wave. The four waves split the (rate category, parent-state chunk) work items of the tile, so
// the transition-matrix coefficients a wave needs are wave-uniform: they are fetched with scalar
// loads from the transposed matrix PT (constant address space) and enter v_fma_f64 as SGPR
// operands - no LDS or VGPR traffic for the matrix at all. The child CLV tile is staged through
// LDS (coalesced row copies in, conflict-free per-lane row reads out: odd row stride), one child
// at a time so the tile costs 64 x (RG*SP+1) x 8 B of LDS (41.5 KB for 20 states x 4 rates,
// three workgroups per CU). Results are transposed back through the same LDS tile and stored
// with coalesced row copies.
//
// Tips given as codes (PATTERN_TIP) use the same FMA stream with the 0/1 indicator of the tip
// mask as the "CLV" value (bit j of tipmap[code], src/core_partials.c:478-486), so ti / tt need
// no lookup table. Site repeats only change which rows are staged (GATHER).
//
// Arithmetic: src/core_partials.c:709-764 (ii), :465-507 (ti), :1166-1209 + :188-199 (tt),
// :819-879 (repeats); scaling :729-763. Per-rate scaling is honoured for every child kind (the
// reference's AVX2 kernels do, its generic ti does not - SURVEY 8a "quirks").
#pragma once
#include "kernels_common.h"

// copy rows of a CLV into the LDS tile: wave w copies rows w, w+4, ...; row e of the tile is
// entry idx(e) of the source, columns [col0, col0+rowlen)
__device__ __forceinline__ void stage_rows(double *tile, unsigned LSTR, const double *__restrict__ src,
                                           unsigned idx_lane, unsigned span, unsigned col0,
                                           unsigned rowlen, unsigned wave, unsigned lane)
{
  for (unsigned e = wave; e < 64; e += 4)
  {
    const unsigned ent = __shfl(idx_lane, e, 64);
    const double *row = src + (size_t)ent * span + col0;
    for (unsigned x = lane; x < rowlen; x += 64) tile[e * LSTR + x] = row[x];
  }
}

// acc[i] += PT[k][j][c*ICH + i] * x_j for all contraction indices j
template <int ICH, bool TIP>
__device__ __forceinline__ void contract(double (&acc)[ICH], const double *pt, unsigned k, unsigned c,
                                         const GenGeo &g, const double *tile_row, unsigned col,
                                         unsigned long long mask)
{
  cdouble_p p = as_const(pt) + ((size_t)k * g.S) * g.SPT + c * ICH;
#pragma unroll
  for (int i = 0; i < ICH; ++i) acc[i] = 0.0;
  for (unsigned j = 0; j < g.S; ++j, p += g.SPT)
  {
    double x;
    if (TIP)
      x = ((mask >> j) & 1ull) ? 1.0 : 0.0;
    else
      x = tile_row[col + j];
#pragma unroll
    for (int i = 0; i < ICH; ++i) acc[i] = fma(p[i], x, acc[i]);
  }
}

template <int ICH, bool LTIP, bool RTIP, bool GATHER>
__global__ __launch_bounds__(256) void k_partials_generic(const OpPack pack, const GenGeo g,
                                                          const unsigned long long *__restrict__ tipmap,
                                                          unsigned tiles_per_block)
{
  extern __shared__ double lds[];
  double *tile = lds;                                                    // [64][LSTR]
  unsigned char *flags = reinterpret_cast<unsigned char *>(lds + 64 * g.LSTR); // [4][64]

  const DevOp &op = pack.ops[blockIdx.y];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned span = g.R * g.SP;
  const int mode = op.pscaler ? g.scale_mode : 0;

  for (unsigned t = 0; t < tiles_per_block; ++t)
  {
    const unsigned tile0 = (blockIdx.x * tiles_per_block + t) * 64u;
    if (tile0 >= op.entries) break; // uniform
    const unsigned n = tile0 + lane;
    const bool valid = n < op.entries;
    const unsigned nn = valid ? n : op.entries - 1; // clamp: tail lanes redo the last entry
    unsigned le = nn, re = nn;
    if (GATHER)
    {
      const unsigned site = op.id_site ? op.id_site[nn] : nn;
      le = op.lsid ? op.lsid[site] : site;
      re = op.rsid ? op.rsid[site] : site;
    }
    unsigned long long lmask = 0, rmask = 0;
    if (LTIP) lmask = tipmap ? tipmap[op.ltip[le]] : (unsigned long long)op.ltip[le];
    if (RTIP) rmask = tipmap ? tipmap[op.rtip[re]] : (unsigned long long)op.rtip[re];

    bool site_small = true;
    for (unsigned grp = 0; grp < g.ngroups; ++grp)
    {
      const unsigned k0 = grp * g.RG;
      const unsigned nk = min(g.RG, g.R - k0);
      const unsigned rowlen = nk * g.SP;
      const unsigned nitems = nk * g.nchunks;          // <= 4 by construction of RG
      const bool has_item = wave < nitems;
      const unsigned kk = has_item ? wave / g.nchunks : 0; // rate within the group
      const unsigned c = has_item ? wave % g.nchunks : 0;  // parent-state chunk
      const unsigned k = k0 + kk;

      double A[ICH], B[ICH];
      if (!LTIP)
      {
        stage_rows(tile, g.LSTR, op.left, le, span, k0 * g.SP, rowlen, wave, lane);
        __syncthreads();
      }
      if (has_item) contract<ICH, LTIP>(A, op.lmat, k, c, g, tile + lane * g.LSTR, kk * g.SP, lmask);
      if (!RTIP)
      {
        if (!LTIP) __syncthreads(); // every wave is done reading the left tile
        stage_rows(tile, g.LSTR, op.right, re, span, k0 * g.SP, rowlen, wave, lane);
        __syncthreads();
      }
      if (has_item) contract<ICH, RTIP>(B, op.rmat, k, c, g, tile + lane * g.LSTR, kk * g.SP, rmask);

      bool small = true;
      if (has_item)
      {
#pragma unroll
        for (int i = 0; i < ICH; ++i)
        {
          A[i] *= B[i];
          if (c * ICH + i < g.S) small = small && (A[i] < PLLGPU_SCALE_THRESHOLD);
        }
        if (mode) flags[wave * 64 + lane] = small ? 1 : 0;
      }
      __syncthreads(); // flags visible; tile no longer read by anyone

      if (mode == 2)
      {
        // all states of this (site, rate): AND over the chunks of the rate (src/core_partials.c:736-746)
        bool rs = true;
        for (unsigned cc = 0; cc < g.nchunks; ++cc) rs = rs && flags[(kk * g.nchunks + cc) * 64 + lane];
        if (has_item)
        {
          if (rs)
          {
#pragma unroll
            for (int i = 0; i < ICH; ++i) A[i] *= PLLGPU_SCALE_FACTOR;
          }
          if (c == 0 && valid)
            op.pscaler[(size_t)n * g.R + k] = (op.lscaler ? op.lscaler[(size_t)le * g.R + k] : 0u) +
                                              (op.rscaler ? op.rscaler[(size_t)re * g.R + k] : 0u) +
                                              (rs ? 1u : 0u);
        }
      }
      else if (mode == 1)
      {
        for (unsigned it = 0; it < nitems; ++it) site_small = site_small && flags[it * 64 + lane];
        if (g.ngroups == 1 && site_small && has_item)
        {
#pragma unroll
          for (int i = 0; i < ICH; ++i) A[i] *= PLLGPU_SCALE_FACTOR;
        }
      }

      if (has_item)
      {
        double *row = tile + lane * g.LSTR + kk * g.SP + c * ICH;
#pragma unroll
        for (int i = 0; i < ICH; ++i)
          if (c * ICH + i < g.SP) row[i] = (c * ICH + i < g.S) ? A[i] : 0.0; // padding lanes := 0
      }
      __syncthreads();
      for (unsigned e = wave; e < 64 && tile0 + e < op.entries; e += 4)
      {
        double *dst = op.parent + (size_t)(tile0 + e) * span + k0 * g.SP;
        for (unsigned x = lane; x < rowlen; x += 64) dst[x] = tile[e * g.LSTR + x];
      }
      __syncthreads(); // tile free for the next group / tile
    }

    if (mode == 1)
    {
      if (g.ngroups > 1 && site_small)
      {
        // rare: the site's rate groups were already stored unscaled - rescale the stored row
        // (this workgroup's own stores, ordered by the barrier above)
        if (wave == 0 && valid)
        {
          double *row = op.parent + (size_t)n * span;
          for (unsigned x = 0; x < span; ++x) row[x] = row[x] * PLLGPU_SCALE_FACTOR;
        }
      }
      if (wave == 0 && valid)
        op.pscaler[n] = (op.lscaler ? op.lscaler[le] : 0u) + (op.rscaler ? op.rscaler[re] : 0u) +
                        (site_small ? 1u : 0u);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// edge / root log-likelihood, any states/rates. Same tile/lane mapping: work item (rate, chunk)
// per wave gives the partial sum_i p_i pi_i (P c)_i of its chunk; wave 0 then mixes rates, undoes
// scaling, takes the log and accumulates the block sum in site order.
// Arithmetic: src/core_likelihood.c:1388-1490 (ii), :812-915 (ti), :1077-1183 (repeats), :163-207 (root).
template <int ICH, bool CTIP, bool GATHER>
__global__ __launch_bounds__(256) void k_edge_generic(const DevEdge e, const GenGeo g,
                                                      const unsigned long long *__restrict__ tipmap,
                                                      unsigned tiles_per_block)
{
  extern __shared__ double lds[];
  double *tile = lds;                       // [64][LSTR]
  double *part = lds + 64 * g.LSTR;         // [4][64] chunk partials
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned span = g.R * g.SP;
  double acc = 0.0;

  for (unsigned t = 0; t < tiles_per_block; ++t)
  {
    const unsigned tile0 = (blockIdx.x * tiles_per_block + t) * 64u;
    if (tile0 >= e.sites) break;
    const unsigned n = tile0 + lane;
    const bool valid = n < e.sites;
    const unsigned nn = valid ? n : e.sites - 1;
    unsigned pe = nn, ce = nn;
    if (GATHER)
    {
      pe = e.psid ? e.psid[nn] : nn;
      ce = e.csid ? e.csid[nn] : nn;
    }
    unsigned long long cmask = 0;
    if (CTIP) cmask = tipmap ? tipmap[e.ctip[ce]] : (unsigned long long)e.ctip[ce];

    // wave 0 owns the per-site epilogue state
    unsigned scal = 0;
    if (wave == 0)
    {
      if (e.per_rate)
      {
        scal = 0xFFFFFFFFu;
        for (unsigned k = 0; k < g.R; ++k)
        {
          unsigned rs = (e.pscaler ? e.pscaler[(size_t)pe * g.R + k] : 0u) +
                        (e.cscaler ? e.cscaler[(size_t)ce * g.R + k] : 0u);
          scal = min(scal, rs);
        }
      }
      else
        scal = (e.pscaler ? e.pscaler[pe] : 0u) + (e.cscaler ? e.cscaler[ce] : 0u);
    }
    double terma = 0.0, terminv = 0.0;

    for (unsigned grp = 0; grp < g.ngroups; ++grp)
    {
      const unsigned k0 = grp * g.RG;
      const unsigned nk = min(g.RG, g.R - k0);
      const unsigned rowlen = nk * g.SP;
      const unsigned nitems = nk * g.nchunks;
      const bool has_item = wave < nitems;
      const unsigned kk = has_item ? wave / g.nchunks : 0;
      const unsigned c = has_item ? wave % g.nchunks : 0;
      const unsigned k = k0 + kk;

      double B[ICH];
      if (e.is_root)
      {
#pragma unroll
        for (int i = 0; i < ICH; ++i) B[i] = 1.0;
      }
      else
      {
        if (!CTIP)
        {
          stage_rows(tile, g.LSTR, e.child, ce, span, k0 * g.SP, rowlen, wave, lane);
          __syncthreads();
        }
        if (has_item) contract<ICH, CTIP>(B, e.mat, k, c, g, tile + lane * g.LSTR, kk * g.SP, cmask);
        if (!CTIP) __syncthreads();
      }
      stage_rows(tile, g.LSTR, e.parent, pe, span, k0 * g.SP, rowlen, wave, lane);
      __syncthreads();
      if (has_item)
      {
        cdouble_p pi = as_const(e.freqs) + (size_t)e.fidx[k] * g.SP + c * ICH;
        const double *row = tile + lane * g.LSTR + kk * g.SP + c * ICH;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < ICH; ++i)
          if (c * ICH + i < g.S) s = fma(row[i] * pi[i], B[i], s);
        part[wave * 64 + lane] = s;
      }
      __syncthreads();
      if (wave == 0)
      {
        for (unsigned r = 0; r < nk; ++r)
        {
          const unsigned kr = k0 + r;
          double tr = 0.0;
          for (unsigned cc = 0; cc < g.nchunks; ++cc) tr += part[(r * g.nchunks + cc) * 64 + lane];
          if (e.per_rate)
          {
            unsigned rs = (e.pscaler ? e.pscaler[(size_t)pe * g.R + kr] : 0u) +
                          (e.cscaler ? e.cscaler[(size_t)ce * g.R + kr] : 0u);
            const unsigned ex = min(rs - scal, PLLGPU_RATE_MAXDIFF);
            if (ex) tr *= minlh(ex);
          }
          const unsigned fi = e.fidx[kr];
          const double pinv = e.prop_invar ? e.prop_invar[fi] : 0.0;
          const double w = e.rate_weights[kr];
          if (pinv > 0.0)
          {
            terma += w * tr * (1.0 - pinv);
            const int inv = e.invariant ? e.invariant[nn] : -1;
            if (inv >= 0) terminv += w * e.freqs[(size_t)fi * g.SP + inv] * pinv;
          }
          else
            terma += tr * w;
        }
      }
      __syncthreads(); // part / tile reusable
    }
    if (wave == 0 && valid)
    {
      double site = finish_site(terma, terminv, scal, e.is_root) * (double)e.pattern_weights[n];
      if (e.persite) e.persite[n] = site;
      acc += site;
    }
  }
  if (wave == 0)
  {
    acc = wave_sum(acc);
    if (lane == 0) e.block_sums[blockIdx.x] = acc;
  }
}
