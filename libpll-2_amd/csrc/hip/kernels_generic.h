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

// the same from an ENTRY-CONTIGUOUS child (class-compressed node, see k_partials_tiled): x = the lane's S
// consecutive values of rate k. Two states per 16-byte load where the host stride keeps them aligned; a
// compressed child is a table its parent's entries keep coming back to: cacheable loads.
template <int ICH>
__device__ __forceinline__ void contract_aos(double (&acc)[ICH], const double *pt, unsigned k, unsigned c, const GenGeo &g,
                                             const double *__restrict__ x)
{
  cdouble_p p = as_const(pt) + ((size_t)k * g.S) * g.SPT + c * ICH;
#pragma unroll
  for (int i = 0; i < ICH; ++i) acc[i] = 0.0;
  if (!(g.SP & 1u)) // wave-uniform
  {
    typedef double c_dbl2 __attribute__((ext_vector_type(2)));
    const c_dbl2 *x2 = reinterpret_cast<const c_dbl2 *>(x);
#pragma unroll 1
    for (unsigned jc = 0; jc < (g.S + 1u) / 2u; ++jc) // (five requests in flight per wait: 156 registers, the launch 216 -> 289 us)
    {
      const unsigned j = 2u * jc;
      c_dbl2 v;
      if (j + 1u < g.S)
        v = x2[jc];
      else
      {
        v.x = x[j];
        v.y = 0.0;
      }
      cdouble_p pj = p + (size_t)j * g.SPT;
#pragma unroll
      for (int i = 0; i < ICH; ++i) acc[i] = fma(pj[i], v.x, acc[i]);
      if (j + 1u < g.S)
      {
        cdouble_p pj1 = pj + g.SPT;
#pragma unroll
        for (int i = 0; i < ICH; ++i) acc[i] = fma(pj1[i], v.y, acc[i]);
      }
    }
    return;
  }
#pragma unroll 4
  for (unsigned j = 0; j < g.S; ++j)
  {
    const double xj = x[j];
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

// Registers: ARCHITECTED PLUS ACCUMULATION count against the 512 of a SIMD lane - the two halves of gfx950's unified
// file (rocprofv3's VGPR_Count shows only the first half: this kernel "has 68"). The 20-state inner x inner
// instantiation sits at 127 = four waves per SIMD; two more and it runs three (C3: 3.86 -> 3.52 G updates/s), and a
// forced bound (__launch_bounds__(256, 4)) makes the compiler give up its prefetching instead (3.1 G). Hence the
// entry-contiguous addressing below lives in branches of its own that only the GATHER instantiations contain.
// LAY >= 0: the layout bits (kAosLeft | kAosRight | kAosParent) of every op of the launch, known at compile time - the
// gathering 20-state instantiation that decides them at run time carries both addressings through its loops and
// needs 147 registers (three waves per SIMD); -1: read them from the op.
template <int ICH, bool LTIP, bool RTIP, bool GATHER, int LAY = -1>
__global__ __launch_bounds__(256) void k_partials_tiled(const OpPack pack, const GenGeo g,
                                                        const unsigned long long *__restrict__ tipmap, unsigned tip_lds,
                                                        unsigned tiles_per_block, unsigned par_lds)
{
  __shared__ unsigned char flags[kMaxRates][64];
  extern __shared__ double tipmat[]; // tip_lds: [child][wave][(S + 1) x SPT] staged tip matrices + row sums;
                                     // par_lds: then [wave][64][SP + 2], the entries of an entry-contiguous parent on their way out

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
  // An entry-contiguous parent (class-compressed node): lane = entry, so a store instruction of the plain path puts 8
  // bytes into each of 64 different entries, SP x R x 8 bytes apart (C3 with site repeats: the 16 ops over compressed
  // level-2 nodes wrote 160 MB in 182 us). Instead the wave parks its 64 entries of one rate in LDS ([entry][SP + 2]:
  // 16-byte aligned rows, eight consecutive lanes on eight different bank groups) and writes them out 16 bytes per
  // lane, SP / 2 lanes per entry: every store instruction covers whole (entry, rate) blocks.
  const unsigned pst = g.SP + 2u, hp = ICH == 20 ? 10u : g.SP >> 1; // (the host sets par_lds for ICH == 20 only when SP == 20)
  double *pstage = tipmat + (tip_lds ? (size_t)2 * nw * slot : 0) + (size_t)wave * 64u * pst;
  const bool par_via_lds = GATHER && par_lds != 0;
  const unsigned fl_sub = lane % (hp ? hp : 1u), fl_ent = lane / (hp ? hp : 1u), fl_per = 64u / (hp ? hp : 1u);
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
    // per-site mode: the children's counts now, so that their entries need not stay in registers to the end
    unsigned below = 0;
    if (mode == 1 && wave == 0) below = (op.lscaler ? op.lscaler[le] : 0u) + (op.rscaler ? op.rscaler[re] : 0u);
    unsigned long long lmask = 0, rmask = 0;
    if (LTIP) lmask = tipmap ? tipmap[op.ltip[le]] : (unsigned long long)op.ltip[le];
    if (RTIP) rmask = tipmap ? tipmap[op.rtip[re]] : (unsigned long long)op.rtip[re];
    // Class-compressed nodes (site repeats) keep their CLV ENTRY-CONTIGUOUS on the device - the host's own
    // [entry][rate][states_padded] - for every shape, as the 4 x 4 kernels do (kernels_dna.h): the children of a
    // gathering op are addressed through class maps, and a lane's 20 states of one rate are then 160 contiguous
    // bytes instead of 20 pieces in 20 rows of a tile (C3 with site repeats: the launch that reads the compressed
    // level-2 nodes took 579 us - every 8-byte value its own cache line - against 60 us for the same ops without
    // repeats). xs / ks: distance between states / between rate categories of the lane's entry.
    const unsigned lay = LAY >= 0 ? (unsigned)LAY : op.layout;
    const bool laos = GATHER && !LTIP && (lay & kAosLeft), raos = GATHER && !RTIP && (lay & kAosRight), paos = GATHER && (lay & kAosParent);
    const unsigned espan = g.R * g.SP;
    const double *__restrict__ lx = LTIP ? nullptr : laos ? op.left + (size_t)le * espan : op.left + tiled_base(le, g.tile_sz);
    const double *__restrict__ rx = RTIP ? nullptr : raos ? op.right + (size_t)re * espan : op.right + tiled_base(re, g.tile_sz);
    double *__restrict__ out = op.parent + (size_t)tile * g.tile_sz + lane; // tiled parent (an entry-contiguous one is addressed where it is stored)

    // Entry-contiguous CHILDREN through the same LDS slot (one child at a time; ops of one chunk): SP / 2 consecutive
    // lanes fetch the SP x 8 bytes of one (entry, rate) - every cache line requested once per wave. A lane fetching
    // its own entry 16 bytes at a time puts 64 different lines into every request, ten requests long, and the
    // vector cache (12 waves x 2 children x 64 entries) does not keep them from one request to the next: the
    // launch over C3's compressed level-2 nodes asked L2 for 1.1 GB to read 0.3 GB.
    const bool coop = GATHER && par_lds == 1u && g.nchunks == 1u; // par_lds == 2: the parent's way out only (A/B switch)
    auto coop_fetch = [&](const double *__restrict__ base, unsigned my_entry, unsigned k) {
      typedef double dbl2 __attribute__((ext_vector_type(2)));
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier(); // every lane is done with what the slot held
      const double *src0 = base + (size_t)k * g.SP + 2u * fl_sub;
      if (ICH == 20)
      {
        dbl2 w[11]; // all requests first
#pragma unroll
        for (int r = 0; r < 11; ++r)
        {
          const unsigned ent = 6u * r + fl_ent;
          const unsigned src = __shfl(my_entry, ent & 63u, 64);
          if (fl_ent < 6u && ent < 64u) w[r] = *reinterpret_cast<const dbl2 *>(src0 + (size_t)src * espan);
        }
#pragma unroll
        for (int r = 0; r < 11; ++r)
        {
          const unsigned ent = 6u * r + fl_ent;
          if (fl_ent < 6u && ent < 64u) *reinterpret_cast<dbl2 *>(pstage + ent * pst + 2u * fl_sub) = w[r];
        }
      }
      else
        for (unsigned e0 = 0; e0 < 64u; e0 += fl_per)
        {
          const unsigned ent = e0 + fl_ent;
          const unsigned src = __shfl(my_entry, ent & 63u, 64);
          if (fl_ent < fl_per && ent < 64u)
            *reinterpret_cast<dbl2 *>(pstage + ent * pst + 2u * fl_sub) = *reinterpret_cast<const dbl2 *>(src0 + (size_t)src * espan);
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };

    auto rescale_rate = [&](unsigned k) {
      // this lane's stored column of rate k: same lane wrote it; order the accesses explicitly
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      if (GATHER && paos)
      {
        double *col = op.parent + ((size_t)n * g.R + k) * g.SP;
        for (unsigned s = 0; s < g.S; ++s) col[s] *= PLLGPU_SCALE_FACTOR;
        return;
      }
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
        {
          if (GATHER && laos && coop) // wave-uniform
          {
            coop_fetch(op.left, le, k);
            contract_aos<ICH>(A, op.lmat, k, c, g, pstage + lane * pst);
          }
          else if (GATHER && laos)
            contract_aos<ICH>(A, op.lmat, k, c, g, lx + (size_t)k * g.SP);
          else
            contract<ICH, LTIP>(A, op.lmat, k, c, g, LTIP ? nullptr : lx + (size_t)k * g.S * 64, lmask);
        }
        if (rsimple)
          tip_columns<ICH>(B, rstage, c, g, rmask, full);
        else
        {
          if (GATHER && raos && coop)
          {
            coop_fetch(op.right, re, k);
            contract_aos<ICH>(B, op.rmat, k, c, g, pstage + lane * pst);
          }
          else if (GATHER && raos)
            contract_aos<ICH>(B, op.rmat, k, c, g, rx + (size_t)k * g.SP);
          else
            contract<ICH, RTIP>(B, op.rmat, k, c, g, RTIP ? nullptr : rx + (size_t)k * g.S * 64, rmask);
        }
        if (GATHER && paos && par_via_lds) // wave-uniform: entry-contiguous parent, through LDS
        {
          double *dst = pstage + lane * pst + c * ICH;
#pragma unroll
          for (int i = 0; i < ICH; ++i)
            if (c * ICH + i < g.S)
            {
              const double v = A[i] * B[i];
              small = small && (v < PLLGPU_SCALE_THRESHOLD);
              dst[i] = v;
            }
        }
        else if (GATHER && paos) // wave-uniform: entry-contiguous parent
        {
          double *dst = op.parent + ((size_t)n * g.R + k) * g.SP + c * ICH; // formed here: one pointer less across the contractions
#pragma unroll
          for (int i = 0; i < ICH; ++i)
            if (c * ICH + i < g.S)
            {
              const double v = A[i] * B[i];
              small = small && (v < PLLGPU_SCALE_THRESHOLD);
              if (valid) dst[i] = v;
            }
        }
        else
        {
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
      }
      if (GATHER && paos && par_via_lds)
      {
        typedef double dbl2 __attribute__((ext_vector_type(2)));
        for (unsigned s = g.S; s < g.SP; ++s) pstage[lane * pst + s] = 0.0; // the padding lanes of the host layout stay zero
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (fl_ent < fl_per)
          for (unsigned e0 = 0; e0 < 64u; e0 += fl_per)
          {
            const unsigned ent = e0 + fl_ent;
            if (ent < 64u && tile * 64u + ent < op.entries)
            {
              const dbl2 w = *reinterpret_cast<const dbl2 *>(pstage + ent * pst + 2u * fl_sub);
              *reinterpret_cast<dbl2 *>(op.parent + ((size_t)(tile * 64u + ent) * g.R + k) * g.SP + 2u * fl_sub) = w;
            }
          }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier(); // the slot is rewritten by the wave's next rate / tile
      }
      else if (GATHER && paos && valid) // the padding lanes of the host layout stay zero
        for (unsigned s = g.S; s < g.SP; ++s) op.parent[((size_t)n * g.R + k) * g.SP + s] = 0.0;
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
      lds_barrier(); // the flags only: the tile's stores stay in flight
      bool site_small = true;
      for (unsigned k = 0; k < g.R; ++k) site_small = site_small && flags[k][lane];
      if (valid)
      {
        if (site_small)
          for (unsigned k = wave; k < g.R; k += nw) rescale_rate(k);
        if (wave == 0) op.pscaler[n] = below + (site_small ? 1u : 0u);
      }
      lds_barrier(); // flags[] is reused by the next tile
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
    const bool paos = GATHER && (e.layout & kAosParent), caos = GATHER && !CTIP && (e.layout & kAosLeft); // entry-contiguous ends
    const unsigned espan = g.R * g.SP;
    const double *__restrict__ px = paos ? e.parent + (size_t)pe * espan : e.parent + tiled_base(pe, g.tile_sz);
    const double *__restrict__ cx = (CTIP || e.is_root) ? nullptr : caos ? e.child + (size_t)ce * espan : e.child + tiled_base(ce, g.tile_sz);
    const unsigned pxs = GATHER ? (paos ? 1u : 64u) : 64u; // without GATHER a compile-time constant
    const size_t pks = (GATHER && paos) ? g.SP : (size_t)g.S * 64;

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
        {
          if (GATHER && caos) // wave-uniform
            contract_aos<ICH>(B, e.mat, k, c, g, cx + (size_t)k * g.SP);
          else
            contract<ICH, CTIP>(B, e.mat, k, c, g, CTIP ? nullptr : cx + (size_t)k * g.S * 64, cmask);
        }
        cdouble_p pi = as_const(e.freqs) + (size_t)fi * g.SP + c * ICH;
        const double *pk = px + (size_t)k * pks + (size_t)(c * ICH) * pxs;
#pragma unroll
        for (int i = 0; i < ICH; ++i)
          if (c * ICH + i < g.S) tr = fma(__builtin_nontemporal_load(pk + (size_t)i * pxs) * pi[i], B[i], tr);
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
// The flat seam's tip-tip pair (src/pll.h:1049-1071): pll_core_create_lookup leaves, for every pair (j, k) of tip
// codes, the parent entry of a cherry whose tips show j and k - the product of the two sums over the set bits of the
// codes' state masks (ascending states, src/core_partials.c:1013-1071, :1149-1209) - in the caller's table, and
// pll_core_update_partial_tt copies one entry per site (:180-199). Table layout = the reference's: entry (j, k) at
// index (j << log2(maxstates)) + k - 16 j + k for 4 states - of rate_cats x states_padded doubles (padding lanes 0).
// matrices: the caller's layout [rate][row i][states_padded]. One thread per table double.
__global__ __launch_bounds__(256) void k_create_lookup(double *__restrict__ table, const double *__restrict__ lmat, const double *__restrict__ rmat,
                                                       const unsigned long long *__restrict__ tipmap /* null: the code is the mask */,
                                                       unsigned S, unsigned SP, unsigned R, unsigned ncodes, unsigned shift)
{
  const unsigned span = R * SP;
  const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (idx >= (size_t)ncodes * ncodes * span) return;
  const unsigned i = (unsigned)(idx % SP), n = (unsigned)((idx / SP) % R);
  const unsigned pair = (unsigned)(idx / span), j = pair / ncodes, k = pair % ncodes;
  double v = 0.0;
  if (i < S)
  {
    unsigned long long mj = tipmap ? tipmap[j] : (unsigned long long)j, mk = tipmap ? tipmap[k] : (unsigned long long)k;
    const double *lrow = lmat + ((size_t)n * S + i) * SP, *rrow = rmat + ((size_t)n * S + i) * SP;
    double tj = 0.0, tk = 0.0;
    for (unsigned m = 0; m < S; ++m)
    {
      if (mj & 1ull) tj += lrow[m];
      if (mk & 1ull) tk += rrow[m];
      mj >>= 1;
      mk >>= 1;
    }
    v = tj * tk;
  }
  table[((size_t)((j << shift) + k)) * span + (size_t)n * SP + i] = v;
}

// parent[site] = table[(left code << shift) + right code]: span doubles per site, one thread per double
__global__ __launch_bounds__(256) void k_tt_from_lookup(double *__restrict__ parent, const double *__restrict__ table,
                                                        const unsigned char *__restrict__ lc, const unsigned char *__restrict__ rc,
                                                        unsigned sites, unsigned span, unsigned shift)
{
  const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (idx >= (size_t)sites * span) return;
  const unsigned n = (unsigned)(idx / span), q = (unsigned)(idx % span);
  parent[idx] = table[(size_t)(((unsigned)lc[n] << shift) + rc[n]) * span + q];
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

// The same two with the entry-major side in HOST memory (small transfers: pllgpu.hip, stage_take): one thread per element of
// THAT side, so that what crosses the bus are whole consecutive lines, each once - host memory is not cached on the device, and
// indexed by the tiled side every 8-byte read would fetch its own line (sixteen times the bytes at 4 states x 4 rates).
__global__ __launch_bounds__(256) void k_host_aos_to_tiled(const double *__restrict__ aos, double *__restrict__ tiled,
                                                           unsigned entries, unsigned S, unsigned SP, unsigned R)
{
  const size_t padded = (size_t)((entries + 63u) / 64u) * 64u, total = padded * R * SP;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
  {
    const unsigned j = idx % SP;
    size_t r = idx / SP;
    const unsigned k = r % R;
    const size_t ent = r / R;
    if (j < S) tiled[(((ent >> 6) * R + k) * S + j) * 64 + (ent & 63u)] = ent < entries ? aos[idx] : 0.0;
  }
}

__global__ __launch_bounds__(256) void k_host_tiled_to_aos(const double *__restrict__ tiled, double *__restrict__ aos,
                                                           unsigned entries, unsigned S, unsigned SP, unsigned R)
{
  const size_t total = (size_t)entries * R * SP;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256)
  {
    const unsigned j = idx % SP;
    size_t r = idx / SP;
    const unsigned k = r % R;
    const size_t ent = r / R;
    aos[idx] = j < S ? tiled[(((ent >> 6) * R + k) * S + j) * 64 + (ent & 63u)] : 0.0;
  }
}
