// kernels_generic.h - any number of states (<= 64) and rate categories.
//
// DEVICE LAYOUT for these kernels is "tiled sites-contiguous" (AoSoA), not the host's
// [entry][rate][state]:   clv[tile][rate k][state j][lane],  tile = entry >> 6, lane = entry & 63,
// i.e. the 64 sites of a tile are contiguous for every (rate, state). With LANE = SITE a wavefront
// then reads x_j of its 64 sites as ONE contiguous 512-byte line-pair and writes every result the
// same way: no LDS transposes, no barriers in the streaming part, occupancy bounded by VGPRs only.
// The host mirror keeps the reference layout; k_aos_to_tiled / k_tiled_to_aos convert at upload /
// sync time (never on the hot path). State padding is dropped on the device (stride S, not SP).
//
// Thread mapping: workgroup = one tile of 64 entries; wave w owns the rate categories
// w, w+nw, ... of the tile (nw = min(R,4) waves). For its (tile, rate) a wave computes all parent
// states in chunks of ICH accumulators: acc[i] += PT[k][j][c*ICH+i] * x_j, where the ICH
// coefficients are WAVE-UNIFORM -> fetched by scalar loads from the transposed matrix (constant
// address space) and fed to v_fma_f64 as SGPR operands: the matrix costs no LDS, no VGPRs.
// x_j comes straight from HBM/L2 (coalesced), or - for tips given as codes (PATTERN_TIP) - is
// the 0/1 indicator of bit j of tipmap[code] (src/core_partials.c:478-486), so tip-inner and
// tip-tip updates need no lookup table. Site repeats turn x_j loads into per-lane gathers.
//
// Scaling (src/core_partials.c:729-763): results are stored unscaled as they are produced; each
// lane keeps "all my entries < 2^-256" per rate. Per-rate mode: the wave rescales its own stored
// column of that rate when the flag is set (rare). Per-site mode: flags of all rates meet in a
// R x 64 byte LDS array behind one barrier, then every wave rescales the rates it produced.
//
// Arithmetic: src/core_partials.c:709-764 (ii), :465-507 (ti), :1166-1209 + :188-199 (tt),
// :819-879 (repeats). Per-rate scaling is honoured for every child kind (the reference's AVX2
// kernels do, its generic non-4-state ti does not - SURVEY 8a "quirks").
#pragma once
#include "kernels_common.h"

// acc[i] = sum_j PT[k][j][c*ICH + i] * x_j ; x_j from a tiled CLV (stride 64 between states) or
// from a tip mask
template <int ICH, bool TIP>
__device__ __forceinline__ void contract(double (&acc)[ICH], const double *pt, unsigned k, unsigned c,
                                         const GenGeo &g, const double *__restrict__ x /* &clv[..][k][0][lane] */,
                                         unsigned long long mask)
{
  cdouble_p p = as_const(pt) + ((size_t)k * g.S) * g.SPT + c * ICH;
#pragma unroll
  for (int i = 0; i < ICH; ++i) acc[i] = 0.0;
#pragma unroll 4
  for (unsigned j = 0; j < g.S; ++j)
  {
    double xj;
    if (TIP)
      xj = ((mask >> j) & 1ull) ? 1.0 : 0.0;
    else
      xj = __builtin_nontemporal_load(x + (size_t)j * 64); // read-once stream, see kernels_dna.h
    cdouble_p pj = p + (size_t)j * g.SPT;
#pragma unroll
    for (int i = 0; i < ICH; ++i) acc[i] = fma(pj[i], xj, acc[i]);
  }
}

// A tip whose mask is a single state or a full gap needs no contraction: (P x)_i is column `state`
// of P, or the row sum of P (ascending j, the order of the reference's set-bit walk,
// src/core_partials.c:480-489: bit-identical to the FMA route). The wave stages the transposed
// matrix of its rate category in LDS ((S + 1) x SPT doubles incl. the row sums) and every lane reads
// its column from there: 20 LDS reads instead of 400 FMAs per child and (site, rate) for 20 states.
// (Gathering the columns straight from L2 was slower than the FMA route: 64 separate 8-byte
// requests per load instruction.) Used when every lane of the wave has such a mask and the staged
// matrices fit (S * SPT <= 1024, i.e. up to 32 states; larger ones run in kernels_mfma.h).
template <int ICH>
__device__ __forceinline__ void tip_columns(double (&acc)[ICH], const double *staged, unsigned c, const GenGeo &g,
                                            unsigned long long mask, unsigned long long full)
{
  const unsigned row = mask == full ? g.S : (unsigned)__ffsll((long long)mask) - 1u; // row S = the row sums
  const double *p = staged + row * (g.SPT | 1u) + c * ICH; // odd row stride: lanes with different states hit different banks
#pragma unroll
  for (int i = 0; i < ICH; ++i) acc[i] = p[i];
}

// PT[k] (S x SPT, contiguous) and its column sums over j into this wave's LDS slot
__device__ __forceinline__ void tip_stage(double *staged, const double *pt, unsigned k, const GenGeo &g, unsigned lane)
{
  __builtin_amdgcn_wave_barrier(); // earlier readers of this wave's slot are done (LDS ops of a wave stay in order)
  const double *src = pt + (size_t)k * g.S * g.SPT;
  const unsigned n = g.S * g.SPT, ls = g.SPT | 1u;
  for (unsigned idx = lane; idx < n; idx += 64u) staged[(idx / g.SPT) * ls + idx % g.SPT] = src[idx];
  for (unsigned i = lane; i < g.SPT; i += 64u)
  {
    double s = 0.0;
    for (unsigned j = 0; j < g.S; ++j) s += src[(size_t)j * g.SPT + i];
    staged[g.S * ls + i] = s;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// tiled address of element 0 of (entry e, rate 0, state 0)
__device__ __forceinline__ size_t tiled_base(unsigned e, unsigned tile_sz)
{
  return (size_t)(e >> 6) * tile_sz + (e & 63u);
}

template <int ICH, bool LTIP, bool RTIP, bool GATHER>
__global__ __launch_bounds__(256) void k_partials_tiled(const OpPack pack, const GenGeo g,
                                                        const unsigned long long *__restrict__ tipmap, unsigned tip_lds,
                                                        unsigned tiles_per_block)
{
  __shared__ unsigned char flags[kMaxRates][64];
  extern __shared__ double tipmat[]; // tip_lds: [child][wave][(S + 1) x SPT] staged tip matrices + row sums

  const DevOp &op = pack.ops[blockIdx.y];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned nw = blockDim.x >> 6;
  const int mode = op.pscaler ? g.scale_mode : 0;
  const unsigned ntiles = (op.entries + 63u) / 64u;
  if (blockIdx.x * tiles_per_block >= ntiles) return; // whole workgroup

  const unsigned long long full = g.S >= 64 ? ~0ull : ((1ull << g.S) - 1ull);
  const unsigned slot = (g.S + 1u) * (g.SPT | 1u);
  double *lstage = tipmat + (size_t)wave * slot, *rstage = tipmat + (size_t)(nw + wave) * slot;
  // one rate category per wave (R <= 4): the tip matrices are staged once for all tiles of the workgroup
  const bool stage_once = tip_lds && g.R <= nw && wave < g.R;
  if (stage_once)
  {
    if (LTIP) tip_stage(lstage, op.lmat, wave, g, lane);
    if (RTIP) tip_stage(rstage, op.rmat, wave, g, lane);
  }

  for (unsigned t = 0; t < tiles_per_block; ++t)
  {
    const unsigned tile = blockIdx.x * tiles_per_block + t;
    if (tile >= ntiles) break; // whole workgroup
    const unsigned n = tile * 64u + lane;
    const bool valid = n < op.entries;
    const unsigned nn = valid ? n : op.entries - 1; // tail lanes redo the last entry, store nothing
    unsigned le = nn, re = nn;
    if (GATHER)
    {
      gather_entries(op, nn, le, re);
    }
    unsigned long long lmask = 0, rmask = 0;
    if (LTIP) lmask = tipmap ? tipmap[op.ltip[le]] : (unsigned long long)op.ltip[le];
    if (RTIP) rmask = tipmap ? tipmap[op.rtip[re]] : (unsigned long long)op.rtip[re];
    const double *__restrict__ lx = LTIP ? nullptr : op.left + tiled_base(le, g.tile_sz);
    const double *__restrict__ rx = RTIP ? nullptr : op.right + tiled_base(re, g.tile_sz);
    double *__restrict__ out = op.parent + (size_t)tile * g.tile_sz + lane;

    auto rescale_rate = [&](unsigned k) {
      // this lane's stored column of rate k: same lane wrote it; order the accesses explicitly
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      double *col = out + (size_t)k * g.S * 64;
      for (unsigned s = 0; s < g.S; ++s)
      {
        const double v = __builtin_nontemporal_load(col + (size_t)s * 64);
        col[(size_t)s * 64] = v * PLLGPU_SCALE_FACTOR;
      }
    };

    // tips: does every lane of this wave hold a single state or a full gap?
    const bool lsimple = LTIP && tip_lds && __all(__popcll(lmask) == 1 || lmask == full);
    const bool rsimple = RTIP && tip_lds && __all(__popcll(rmask) == 1 || rmask == full);

    for (unsigned k = wave; k < g.R; k += nw)
    {
      bool small = true;
      if (!stage_once)
      {
        if (lsimple) tip_stage(lstage, op.lmat, k, g, lane);
        if (rsimple) tip_stage(rstage, op.rmat, k, g, lane);
      }
      for (unsigned c = 0; c < g.nchunks; ++c)
      {
        double A[ICH], B[ICH];
        if (lsimple)
          tip_columns<ICH>(A, lstage, c, g, lmask, full);
        else
          contract<ICH, LTIP>(A, op.lmat, k, c, g, LTIP ? nullptr : lx + (size_t)k * g.S * 64, lmask);
        if (rsimple)
          tip_columns<ICH>(B, rstage, c, g, rmask, full);
        else
          contract<ICH, RTIP>(B, op.rmat, k, c, g, RTIP ? nullptr : rx + (size_t)k * g.S * 64, rmask);
        double *dst = out + ((size_t)k * g.S + c * ICH) * 64;
#pragma unroll
        for (int i = 0; i < ICH; ++i)
          if (c * ICH + i < g.S)
          {
            const double v = A[i] * B[i];
            small = small && (v < PLLGPU_SCALE_THRESHOLD);
            if (valid)
            {
              if (LTIP && RTIP)
                __builtin_nontemporal_store(v, dst + (size_t)i * 64); // a tip x tip launch is pure store traffic, far beyond the caches: 260 -> 251 us for C3's 32 ops
              else
                dst[(size_t)i * 64] = v;
            }
          }
      }
      if (mode == 2)
      {
        if (valid)
        {
          if (small) rescale_rate(k);
          op.pscaler[(size_t)n * g.R + k] = (op.lscaler ? op.lscaler[(size_t)le * g.R + k] : 0u) +
                                            (op.rscaler ? op.rscaler[(size_t)re * g.R + k] : 0u) +
                                            (small ? 1u : 0u);
        }
      }
      else if (mode == 1)
        flags[k][lane] = small ? 1 : 0;
    }

    if (mode == 1)
    {
      __syncthreads();
      bool site_small = true;
      for (unsigned k = 0; k < g.R; ++k) site_small = site_small && flags[k][lane];
      if (valid)
      {
        if (site_small)
          for (unsigned k = wave; k < g.R; k += nw) rescale_rate(k);
        if (wave == 0)
          op.pscaler[n] = (op.lscaler ? op.lscaler[le] : 0u) + (op.rscaler ? op.rscaler[re] : 0u) +
                          (site_small ? 1u : 0u);
      }
      __syncthreads(); // flags[] is reused by the next tile
    }
  }
}

// ------------------------------------------------------------------------------------------------
// edge / root log-likelihood, any states/rates, tiled layout. Workgroup = one tile at a time; wave
// w of min(R,4) owns the rate categories w, w+nw, ... : it forms (P c)_i in chunks exactly like the
// update kernel, dots it with p_i * pi_i, applies the per-rate scaler excess, the rate weight and
// the invariant-site share, and leaves its partial (terma, terminv) in LDS. Wave 0 adds the waves'
// partials in wave order, undoes the scaling, takes the log and keeps the running sum in site
// order. Splitting the rates over the waves keeps the dependent chain of one evaluation short
// (an lnL call on a small partition is latency-, not bandwidth-bound).
// Arithmetic: src/core_likelihood.c:1388-1490 (ii), :812-915 (ti), :1077-1183 (repeats), :163-207 (root).
template <int ICH, bool CTIP, bool GATHER>
__global__ __launch_bounds__(256) void k_edge_tiled(const DevEdge e, const GenGeo g,
                                                    const unsigned long long *__restrict__ tipmap,
                                                    unsigned tiles_per_block)
{
  __shared__ double part[2][4][64];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned nw = blockDim.x >> 6;
  const unsigned ntiles = (e.sites + 63u) / 64u;
  double acc = 0.0;

  for (unsigned t = 0; t < tiles_per_block; ++t)
  {
    const unsigned tile = blockIdx.x * tiles_per_block + t;
    if (tile >= ntiles) break; // whole workgroup
    const unsigned n = tile * 64u + lane;
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
    const double *__restrict__ px = e.parent + tiled_base(pe, g.tile_sz);
    const double *__restrict__ cx = (CTIP || e.is_root) ? nullptr : e.child + tiled_base(ce, g.tile_sz);

    unsigned scal;
    if (e.per_rate)
    {
      scal = 0xFFFFFFFFu;
      for (unsigned k = 0; k < g.R; ++k)
      {
        const unsigned rs = (e.pscaler ? e.pscaler[(size_t)pe * g.R + k] : 0u) +
                            (e.cscaler ? e.cscaler[(size_t)ce * g.R + k] : 0u);
        scal = min(scal, rs);
      }
    }
    else
      scal = (e.pscaler ? e.pscaler[pe] : 0u) + (e.cscaler ? e.cscaler[ce] : 0u);

    double terma = 0.0, terminv = 0.0;
    for (unsigned k = wave; k < g.R; k += nw)
    {
      const unsigned fi = e.fidx[k];
      double tr = 0.0;
      for (unsigned c = 0; c < g.nchunks; ++c)
      {
        double B[ICH];
        if (e.is_root)
        {
#pragma unroll
          for (int i = 0; i < ICH; ++i) B[i] = 1.0;
        }
        else
          contract<ICH, CTIP>(B, e.mat, k, c, g, CTIP ? nullptr : cx + (size_t)k * g.S * 64, cmask);
        cdouble_p pi = as_const(e.freqs) + (size_t)fi * g.SP + c * ICH;
        const double *pk = px + ((size_t)k * g.S + c * ICH) * 64;
#pragma unroll
        for (int i = 0; i < ICH; ++i)
          if (c * ICH + i < g.S) tr = fma(__builtin_nontemporal_load(pk + (size_t)i * 64) * pi[i], B[i], tr);
      }
      if (e.per_rate)
      {
        const unsigned rs = (e.pscaler ? e.pscaler[(size_t)pe * g.R + k] : 0u) +
                            (e.cscaler ? e.cscaler[(size_t)ce * g.R + k] : 0u);
        const unsigned ex = min(rs - scal, PLLGPU_RATE_MAXDIFF);
        if (ex) tr *= minlh(ex);
      }
      const double pinv = e.prop_invar ? e.prop_invar[fi] : 0.0;
      const double w = e.rate_weights[k];
      if (pinv > 0.0)
      {
        terma += w * tr * (1.0 - pinv);
        const int inv = e.invariant ? e.invariant[nn] : -1;
        if (inv >= 0) terminv += w * e.freqs[(size_t)fi * g.SP + inv] * pinv;
      }
      else
        terma += tr * w;
    }
    part[0][wave][lane] = terma;
    part[1][wave][lane] = terminv;
    __syncthreads();
    if (wave == 0 && valid)
    {
      double ta = part[0][0][lane], ti = part[1][0][lane];
      for (unsigned w = 1; w < nw; ++w)
      {
        ta += part[0][w][lane];
        ti += part[1][w][lane];
      }
      const double site = finish_site(ta, ti, scal, e.is_root) * (double)e.pattern_weights[n];
      if (e.persite) e.persite[n] = site;
      acc += site;
    }
    __syncthreads(); // part[] is reused by the next tile
  }
  // only wave 0 holds a sum
  publish_block_sum(e, wave == 0 ? wave_sum(acc) : 0.0, 1u);
}

// ------------------------------------------------------------------------------------------------
// layout converters between the host mirror [entry][rate][SP] and the tiled device layout. One
// thread per tiled element (coalesced on the tiled side). Not on the hot path.
__global__ __launch_bounds__(256) void k_aos_to_tiled(const double *__restrict__ aos, double *__restrict__ tiled,
                                                      unsigned entries, unsigned S, unsigned SP, unsigned R)
{
  const size_t total = (size_t)((entries + 63u) / 64u) * R * S * 64u;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
  {
    const unsigned lane = idx & 63u;
    size_t r = idx >> 6;
    const unsigned j = r % S;
    r /= S;
    const unsigned k = r % R;
    const size_t tile = r / R;
    const size_t ent = tile * 64 + lane;
    tiled[idx] = ent < entries ? aos[(ent * R + k) * SP + j] : 0.0;
  }
}

__global__ __launch_bounds__(256) void k_tiled_to_aos(const double *__restrict__ tiled, double *__restrict__ aos,
                                                      unsigned entries, unsigned S, unsigned SP, unsigned R)
{
  const size_t total = (size_t)((entries + 63u) / 64u) * R * SP * 64u;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
  {
    const unsigned lane = idx & 63u;
    size_t r = idx >> 6;
    const unsigned j = r % SP;
    r /= SP;
    const unsigned k = r % R;
    const size_t tile = r / R;
    const size_t ent = tile * 64 + lane;
    if (ent < entries) aos[(ent * R + k) * SP + j] = j < S ? tiled[((tile * R + k) * S + j) * 64 + lane] : 0.0;
  }
}
