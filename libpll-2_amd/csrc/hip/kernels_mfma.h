// kernels_mfma.h - CLV update for LARGE state spaces (33..64 states, e.g. 61-state codon models)
// on the fp64 matrix pipe: v_mfma_f64_4x4x4_4b_f64, transition matrices staged in LDS.
//
// Why MFMA here and nowhere else (profiles/r1_fp64_issue_rates.md): fp64 MFMA has no higher peak
// than v_fma_f64 on gfx950, but an FMA needs one fresh matrix coefficient per 128 flop through the
// scalar-load path, which caps the 61-state FMA kernel at ~0.25 of the fp64 rate. The 4x4x4 MFMA
// takes its 16 coefficients from a VGPR that one 512-byte LDS read fills, and uses them for 512
// flop x (site groups per wave). 64 = 16 x 4: no padding waste in this shape.
//
// Lane maps (found with one-hot probes, tools/mfma_4x4_layout.hip): A[blk][i][k] lane 16k+4blk+i,
// B[blk][k][j] lane 16k+4blk+j, D[blk][i][j] lane 16i+4blk+j. With blk = group of 4 sites, an MFMA
// covers 16 sites and lane l always owns site (l & 15) of its 16-site group and "row offset"
// (l >> 4): contraction index j = 4 jg + (l>>4) on the B side, parent state i = 4 ig + (l>>4) on
// the D side. The A operand (the 4x4 block P[4ig..+3][4jg..+3]) is the same in all four blocks:
// lane l reads element (k = l>>4, i = l&3) of a 128-byte fragment in LDS.
//
// Work split: a workgroup (4 waves) walks the rate categories; for each it stages both children's
// matrices as fragments in LDS (2 x 32 KB) and then every wave processes its items of 32 sites
// (2 MFMA site groups): x fragments of one child live in registers (32 loads of 4 x 128-byte
// segments from the tiled CLV, no re-reads), D_left for all 16 state groups stays in registers
// while D_right is formed 4 state groups at a time, multiplied, range-checked and stored.
// Scaling decisions are kept as per-(item, rate, site) flags in LDS and applied after the rate loop
// by rescaling the (rare) affected stored entries - same policy as kernels_generic.h.
//
// Arithmetic: src/core_partials.c:709-764 (ii), :465-507 (ti), :1166-1209 (tt), :819-879 (repeats).
#pragma once
#include "kernels_common.h"

constexpr int kMfmaItemsMax = 8; // items (32 sites each) per wave and launch

// x fragment of (state group jg, site group sg) for this lane: CLV value or tip-mask bit
template <bool TIP>
__device__ __forceinline__ double mfma_x(const double *__restrict__ base /* entry base of the lane's site in group sg */,
                                         unsigned long long mask, unsigned k, unsigned S, unsigned j)
{
  if (TIP) return (j < S && ((mask >> j) & 1ull)) ? 1.0 : 0.0;
  const unsigned jj = j < S ? j : S - 1; // rows beyond S meet zero matrix columns; stay in bounds
  return __builtin_nontemporal_load(base + ((size_t)k * S + jj) * 64);
}

template <bool LTIP, bool RTIP, bool GATHER>
__global__ __launch_bounds__(256, 2) void k_partials_mfma(const OpPack pack, const GenGeo g,
                                                          const unsigned long long *__restrict__ tipmap,
                                                          unsigned items_per_wave)
{
  extern __shared__ double lds[];
  double *PL = lds;                 // [16 ig][16 jg][4 k][4 i]
  double *PR = lds + 4096;
  unsigned char *flags = reinterpret_cast<unsigned char *>(lds + 8192); // [4 waves][items][R<=?][32 sites]

  const DevOp &op = pack.ops[blockIdx.y];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned row = lane >> 4;   // k on the input side, i on the output side
  const unsigned col = lane & 15u;  // site within a 16-site group
  const unsigned S = g.S, R = g.R;
  const unsigned nitems = (op.entries + 31u) / 32u;
  const unsigned item0 = (blockIdx.x * 4u + wave) * items_per_wave;
  if (blockIdx.x * 4u * items_per_wave >= nitems) return; // whole workgroup
  const int mode = op.pscaler ? g.scale_mode : 0;
  const unsigned fragoff = (row * 4u + (lane & 3u));

  for (unsigned k = 0; k < R; ++k)
  {
    __syncthreads(); // previous rate's fragments no longer read
    // stage fragments: frag[ig][jg][kk][ii] = P[4ig+ii][4jg+kk] = PT[k][4jg+kk][4ig+ii]
    for (unsigned idx = threadIdx.x; idx < 4096; idx += 256)
    {
      const unsigned ii = idx & 3u, kk = (idx >> 2) & 3u, jg = (idx >> 4) & 15u, ig = idx >> 8;
      const unsigned j = 4 * jg + kk, i = 4 * ig + ii;
      double l = 0.0, r = 0.0;
      if (j < S && i < g.SPT)
      {
        l = op.lmat[((size_t)k * S + j) * g.SPT + i];
        r = op.rmat[((size_t)k * S + j) * g.SPT + i];
      }
      PL[idx] = l;
      PR[idx] = r;
    }
    __syncthreads();

    for (unsigned it = 0; it < items_per_wave; ++it)
    {
      const unsigned item = item0 + it;
      if (item >= nitems) break; // wave-uniform
      // the lane's two sites (site group 0 and 1 of the item)
      unsigned e[2], le[2], re[2];
      bool valid[2];
      const double *lb[2], *rb[2];
      unsigned long long lm[2] = {0, 0}, rm[2] = {0, 0};
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
      {
        e[sg] = item * 32u + sg * 16u + col;
        valid[sg] = e[sg] < op.entries;
        const unsigned nn = valid[sg] ? e[sg] : op.entries - 1;
        le[sg] = re[sg] = nn;
        if (GATHER)
        {
          const unsigned site = op.id_site ? op.id_site[nn] : nn;
          le[sg] = op.lsid ? op.lsid[site] : site;
          re[sg] = op.rsid ? op.rsid[site] : site;
        }
        if (LTIP) lm[sg] = tipmap ? tipmap[op.ltip[le[sg]]] : (unsigned long long)op.ltip[le[sg]];
        if (RTIP) rm[sg] = tipmap ? tipmap[op.rtip[re[sg]]] : (unsigned long long)op.rtip[re[sg]];
        lb[sg] = LTIP ? nullptr : op.left + (size_t)(le[sg] >> 6) * g.tile_sz + (le[sg] & 63u);
        rb[sg] = RTIP ? nullptr : op.right + (size_t)(re[sg] >> 6) * g.tile_sz + (re[sg] & 63u);
      }

      double x[16][2];
      double DL[16][2];
      // ---- left child: all 16 state groups
#pragma unroll
      for (int jg = 0; jg < 16; ++jg)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) x[jg][sg] = mfma_x<LTIP>(lb[sg], lm[sg], k, S, 4 * jg + row);
#pragma unroll
      for (int ig = 0; ig < 16; ++ig) DL[ig][0] = DL[ig][1] = 0.0;
#pragma unroll
      for (int jg = 0; jg < 16; ++jg)
#pragma unroll
        for (int ig = 0; ig < 16; ++ig)
        {
          const double a = PL[(ig * 16 + jg) * 16 + fragoff];
          DL[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][0], DL[ig][0], 0, 0, 0);
          DL[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][1], DL[ig][1], 0, 0, 0);
          if ((ig & 7) == 7) __builtin_amdgcn_sched_barrier(0); // bound the fragment look-ahead (registers)
        }
      // ---- right child, 4 state groups at a time; product, range test, store
#pragma unroll
      for (int jg = 0; jg < 16; ++jg)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) x[jg][sg] = mfma_x<RTIP>(rb[sg], rm[sg], k, S, 4 * jg + row);
      bool small[2] = {true, true};
#pragma unroll
      for (int c = 0; c < 4; ++c)
      {
        double DR[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) DR[q][0] = DR[q][1] = 0.0;
#pragma unroll
        for (int jg = 0; jg < 16; ++jg)
#pragma unroll
          for (int q = 0; q < 4; ++q)
          {
            const double a = PR[((c * 4 + q) * 16 + jg) * 16 + fragoff];
            DR[q][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][0], DR[q][0], 0, 0, 0);
            DR[q][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][1], DR[q][1], 0, 0, 0);
            if (q == 3 && (jg & 1)) __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
          const unsigned i = 4 * (c * 4 + q) + row;
          if (i < S)
          {
#pragma unroll
            for (int sg = 0; sg < 2; ++sg)
            {
              const double v = DL[c * 4 + q][sg] * DR[q][sg];
              small[sg] = small[sg] && (v < PLLGPU_SCALE_THRESHOLD);
              if (valid[sg]) op.parent[(size_t)(e[sg] >> 6) * g.tile_sz + (e[sg] & 63u) + ((size_t)k * S + i) * 64] = v;
            }
          }
        }
      }
      if (mode)
      {
        // a site's states are spread over the four row groups of the wave: AND them together
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
        {
          int s = small[sg] ? 1 : 0;
          s &= __shfl_xor(s, 16, 64);
          s &= __shfl_xor(s, 32, 64);
          if (row == 0) flags[((wave * kMfmaItemsMax + it) * R + k) * 32 + sg * 16 + col] = (unsigned char)s;
        }
      }
    }
  }

  if (!mode) return;
  // ---- scaling epilogue: row group 0 owns the sites; flags were written by this same wave
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  if (row != 0) return;
  for (unsigned it = 0; it < items_per_wave; ++it)
  {
    const unsigned item = item0 + it;
    if (item >= nitems) break;
    for (int sg = 0; sg < 2; ++sg)
    {
      const unsigned n = item * 32u + sg * 16u + col;
      if (n >= op.entries) continue;
      unsigned le = n, re = n;
      if (GATHER)
      {
        const unsigned site = op.id_site ? op.id_site[n] : n;
        le = op.lsid ? op.lsid[site] : site;
        re = op.rsid ? op.rsid[site] : site;
      }
      double *base = op.parent + (size_t)(n >> 6) * g.tile_sz + (n & 63u);
      const unsigned char *f = flags + ((wave * kMfmaItemsMax + it) * R) * 32 + sg * 16 + col;
      if (mode == 1)
      {
        bool all = true;
        for (unsigned k = 0; k < R; ++k) all = all && f[k * 32];
        if (all)
          for (unsigned q = 0; q < R * S; ++q) base[(size_t)q * 64] = __builtin_nontemporal_load(base + (size_t)q * 64) * PLLGPU_SCALE_FACTOR;
        op.pscaler[n] = (op.lscaler ? op.lscaler[le] : 0u) + (op.rscaler ? op.rscaler[re] : 0u) + (all ? 1u : 0u);
      }
      else
      {
        for (unsigned k = 0; k < R; ++k)
        {
          const bool sm = f[k * 32] != 0;
          if (sm)
            for (unsigned q = 0; q < S; ++q)
            {
              double *p = base + ((size_t)k * S + q) * 64;
              *p = __builtin_nontemporal_load(p) * PLLGPU_SCALE_FACTOR;
            }
          op.pscaler[(size_t)n * R + k] = (op.lscaler ? op.lscaler[(size_t)le * R + k] : 0u) +
                                          (op.rscaler ? op.rscaler[(size_t)re * R + k] : 0u) + (sm ? 1u : 0u);
        }
      }
    }
  }
}
