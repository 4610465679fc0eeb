// kernels_mfma_wide.h - inner x inner CLV update for 33..64 states (61-state codon models) on the fp64 matrix pipe,
// second generation. Same arithmetic, operand maps and LDS fragments as k_partials_mfma (kernels_mfma.h:
// src/core_partials.c:709-764); what changed is what bounded that kernel (profiles/r3_c5_pmc.txt: matrix pipe 73 % busy
// while its waves were resident, clock 1.9 of 2.4 GHz, 9 % of the MFMAs spent on the padding 61 -> 64):
//
// * A lane owns ADJACENT sites: an item is a 64-entry tile (four MFMA site groups: even / odd sites of each half
//   tile) or one half tile (two groups). Children arrive as 16-byte loads, parents leave as 16-byte stores - half the
//   vector-memory instructions - and one 512-byte LDS read of a 4 x 4 block of P feeds FOUR MFMAs (two in a half-tile
//   item): half the LDS reads per flop. With MAINSG = 4 the kernel is written for ONE wave per SIMD (x and D_left of
//   64 sites are 256 registers): the A fragments of the next contraction group are requested a whole group ahead
//   (64 MFMAs), so the single wave never waits for LDS.
// * No padded contraction for 61 states (NGJ = 15, TAIL = 1): the matrix pipe walks the 15 full groups of four
//   contraction states (240 instead of 256 MFMAs per child and site group); the 61st column enters through the vector
//   ALU, which runs beside the matrix pipe, as the INITIAL value of every accumulator chain: D[i] = P[i][60] x[60],
//   then the MFMAs add j = 0 .. 59 in ascending order (the reference adds j = 60 last, src/core_partials.c:739-757:
//   the same sum, associated differently - well inside the 1e-10 the path is held to). Other state counts run
//   padded (NGJ = 16, TAIL = 0) and are bit-identical to k_partials_mfma.
// * Work is dealt in half tiles (32 entries), so 2 ops x 20 000 sites x 4 rates still fill 1024 SIMDs evenly.
//
// Scaling, LDS staging and the flag buffer are k_partials_mfma's; tips and site repeats stay with that kernel.
#pragma once
#include "kernels_mfma.h"

typedef double wide_d2 __attribute__((ext_vector_type(2)));

struct WideItem
{
  unsigned tile;   // 64-entry tile
  unsigned off[2]; // entry offset inside the tile of the lane pair's first site, per half (a half-tile item: both the same half)
};

__device__ __forceinline__ WideItem wide_item(unsigned half_tile, bool whole, unsigned col)
{
  WideItem w;
  w.tile = half_tile >> 1;
  w.off[0] = (whole ? 0u : (half_tile & 1u) * 32u) + 2u * col;
  w.off[1] = whole ? 32u + 2u * col : w.off[0];
  return w;
}

// contraction row group jg of a child: x[0..1] = sites (2 col, 2 col + 1) of the first half, x[2..3] (XW = 4) of the second
template <int NGJ, int TAIL, int XW>
__device__ __forceinline__ void wide_request(double (&x)[XW], const double *__restrict__ child, const WideItem &it, unsigned S, int jg, unsigned row)
{
  unsigned j = (TAIL && jg == NGJ) ? 4u * NGJ : 4u * jg + row; // the 61st state: the same row for all four row groups
  if (!TAIL) j = j < S ? j : S - 1u;                            // rows beyond S meet zero matrix columns; stay in bounds
  const double *p = child + (size_t)j * 64u;
  const wide_d2 a = __builtin_nontemporal_load((const wide_d2 *)(p + it.off[0]));
  x[0] = a.x;
  x[1] = a.y;
  if (XW == 4)
  {
    const wide_d2 b = __builtin_nontemporal_load((const wide_d2 *)(p + it.off[1]));
    x[XW - 2] = b.x;
    x[XW - 1] = b.y;
  }
}

// one item: NSG = 4 (a whole tile) or 2 (a half tile). x holds the LEFT child's rows on entry and the next item's on exit.
template <int NGJ, int TAIL, int NSG, int XW>
__device__ __forceinline__ void wide_body(const DevOp &op, const GenGeo &g, const double *__restrict__ PL, const double *__restrict__ PR,
                                          double (&x)[NGJ + TAIL][XW], const WideItem &cur, const WideItem &nxt, bool has_next, unsigned k,
                                          unsigned row, unsigned fragoff, int mode, unsigned char *__restrict__ flagbuf, unsigned flag_stride)
{
  constexpr int NG = 16; // parent state groups, and the fragment array's row length
  constexpr int CH = 4;  // parent state groups per D_right pass
  constexpr int PF = 8;  // A fragments in flight (ring): block t is requested PF - 1 blocks before its own MFMAs
  const unsigned S = g.S;
  const size_t tile_rate = (size_t)cur.tile * g.tile_sz + (size_t)k * S * 64u;
  const double *rchild = op.right + tile_rate;
  const double *nleft = op.left + (size_t)nxt.tile * g.tile_sz + (size_t)k * S * 64u;
  double *parent = op.parent + tile_rate;

  // ---- left child: D_left for every parent state group
  double DL[NG][NSG];
#pragma unroll
  for (int ig = 0; ig < NG; ++ig)
  {
    // TAIL: the chain starts from the 61st column's term, P[4 ig + row][60] x[60] (x[NGJ] holds x[60] in every row group)
    const double c = TAIL ? PL[(ig * NG + NGJ) * kFrag + row] : 0.0;
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) DL[ig][sg] = TAIL ? c * x[NGJ][sg] : 0.0;
  }
  if (TAIL) wide_request<NGJ, TAIL, XW>(x[NGJ], rchild, cur, S, NGJ, row);
  {
    constexpr int NT = NGJ * NG;
    double a[PF];
#pragma unroll
    for (int t = 0; t < PF - 1; ++t) a[t] = PL[((t % NG) * NG + t / NG) * kFrag + fragoff];
#pragma unroll
    for (int jg = 0; jg < NGJ; ++jg)
    {
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
      {
        const int t = jg * NG + ig, tn = t + PF - 1;
        if (tn < NT) a[tn % PF] = PL[((tn % NG) * NG + tn / NG) * kFrag + fragoff];
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg) DL[ig][sg] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t % PF], x[jg][sg], DL[ig][sg], 0, 0, 0);
      }
      wide_request<NGJ, TAIL, XW>(x[jg], rchild, cur, S, jg, row); // the left child's rows jg are dead: the right child's
      __builtin_amdgcn_sched_barrier(0);                             // keep the request here and the fragment look-ahead bounded
    }
  }

  // ---- right child, CH parent state groups at a time: product, range test, 16-byte stores
  bool small[NSG];
#pragma unroll
  for (int sg = 0; sg < NSG; ++sg) small[sg] = true;
  const unsigned e0 = cur.tile * 64u + cur.off[0], e1 = cur.tile * 64u + cur.off[1];
  const bool v00 = e0 < op.entries, v01 = e0 + 1u < op.entries, v10 = e1 < op.entries, v11 = e1 + 1u < op.entries;
#pragma unroll
  for (int c = 0; c < NG / CH; ++c)
  {
    constexpr int LASTC = NG / CH - 1, NT = NGJ * CH;
    double DR[CH][NSG];
#pragma unroll
    for (int q = 0; q < CH; ++q)
    {
      const double cc = TAIL ? PR[((c * CH + q) * NG + NGJ) * kFrag + row] : 0.0;
#pragma unroll
      for (int sg = 0; sg < NSG; ++sg) DR[q][sg] = TAIL ? cc * x[NGJ][sg] : 0.0;
    }
    double a[PF];
#pragma unroll
    for (int t = 0; t < PF - 1; ++t) a[t] = PR[((c * CH + t % CH) * NG + t / CH) * kFrag + fragoff];
#pragma unroll
    for (int jg = 0; jg < NGJ; ++jg)
    {
#pragma unroll
      for (int q = 0; q < CH; ++q)
      {
        const int t = jg * CH + q, tn = t + PF - 1;
        if (tn < NT) a[tn % PF] = PR[((c * CH + tn % CH) * NG + tn / CH) * kFrag + fragoff];
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg) DR[q][sg] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t % PF], x[jg][sg], DR[q][sg], 0, 0, 0);
      }
      // last chunk: this item is done with rows jg - request the next item's left child
      if (c == LASTC && has_next) wide_request<NGJ, TAIL, XW>(x[jg], nleft, nxt, S, jg, row);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TAIL && c == LASTC && has_next) wide_request<NGJ, TAIL, XW>(x[NGJ], nleft, nxt, S, NGJ, row);
#pragma unroll
    for (int q = 0; q < CH; ++q)
    {
      const int ig = c * CH + q;
      const unsigned i = 4u * ig + row;
      if (i < S)
      {
        double v[NSG];
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg)
        {
          v[sg] = DL[ig][sg] * DR[q][sg];
          small[sg] = small[sg] && (v[sg] < PLLGPU_SCALE_THRESHOLD);
        }
        double *pp = parent + (size_t)i * 64u;
        if (v01) *(wide_d2 *)(pp + cur.off[0]) = wide_d2{v[0], v[1]};
        else if (v00) pp[cur.off[0]] = v[0];
        if (NSG == 4)
        {
          if (v11) *(wide_d2 *)(pp + cur.off[1]) = wide_d2{v[NSG - 2], v[NSG - 1]};
          else if (v10) pp[cur.off[1]] = v[NSG - 2];
        }
      }
    }
  }
  if (mode)
  {
    // a site's states are spread over the four row groups of the wave: AND them together
    unsigned bits = 0;
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) bits |= (small[sg] ? 1u : 0u) << sg;
    bits &= (unsigned)__shfl_xor((int)bits, 16, 64);
    bits &= (unsigned)__shfl_xor((int)bits, 32, 64);
    if (row == 0)
    {
      unsigned char *f = flagbuf + ((size_t)blockIdx.y * g.R + k) * flag_stride;
      if (v00) f[e0] = (unsigned char)(bits & 1u);
      if (v01) f[e0 + 1u] = (unsigned char)((bits >> 1) & 1u);
      if (NSG == 4)
      {
        if (v10) f[e1] = (unsigned char)((bits >> 2) & 1u);
        if (v11) f[e1 + 1u] = (unsigned char)((bits >> 3) & 1u);
      }
    }
  }
}

// grid = (blocks of 4 waves, ops, rate categories); each wave walks `halves_per_wave` consecutive half tiles of its op
template <int NGJ, int TAIL, int MAINSG>
__global__ __launch_bounds__(256, MAINSG == 4 ? 1 : 2) void k_partials_mfma_wide(const OpPack pack, const GenGeo g, unsigned halves_per_wave,
                                                                                 unsigned char *__restrict__ flagbuf, unsigned flag_stride)
{
  extern __shared__ double lds[];
  typedef MfmaGeo<16> MG;
  double *PL = lds;
  double *PR = lds + MG::frag_array;
  const DevOp &op = pack.ops[blockIdx.y];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned row = lane >> 4, col = lane & 15u;
  const unsigned S = g.S, k = blockIdx.z;
  const unsigned nhalves = (op.entries + 31u) / 32u;
  if (blockIdx.x * 4u * halves_per_wave >= nhalves) return; // whole workgroup
  const int mode = op.pscaler ? g.scale_mode : 0;
  const unsigned fragoff = row * 4u + (lane & 3u);
  {
    double *const dst[2] = {PL, PR};
    const double *const src[2] = {op.lmat + (size_t)k * S * g.SPT, op.rmat + (size_t)k * S * g.SPT};
    mfma_stage<16, 2>(dst, src, S, g.SPT);
  }
  __syncthreads();
  unsigned h = (blockIdx.x * 4u + wave) * halves_per_wave;
  if (h >= nhalves) return; // no barriers below
  const unsigned h1 = min(h + halves_per_wave, nhalves);

  // the wave's items: a leading odd half tile, whole tiles, a trailing half tile (MAINSG = 2: half tiles only)
  auto whole_at = [&](unsigned hh) { return MAINSG == 4 && !(hh & 1u) && hh + 2u <= h1; };
  bool whole = whole_at(h);
  WideItem cur = wide_item(h, whole, col);
  double x[NGJ + TAIL][MAINSG];
  {
    const double *left = op.left + (size_t)cur.tile * g.tile_sz + (size_t)k * S * 64u;
#pragma unroll
    for (int jg = 0; jg < NGJ + TAIL; ++jg) wide_request<NGJ, TAIL, MAINSG>(x[jg], left, cur, S, jg, row);
  }
  while (h < h1)
  {
    const unsigned hn = h + (whole ? 2u : 1u);
    const bool has_next = hn < h1;
    const bool nwhole = has_next && whole_at(hn);
    const WideItem nxt = has_next ? wide_item(hn, nwhole, col) : cur;
    if (MAINSG == 4 && whole)
      wide_body<NGJ, TAIL, MAINSG, MAINSG>(op, g, PL, PR, x, cur, nxt, has_next, k, row, fragoff, mode, flagbuf, flag_stride);
    else
      wide_body<NGJ, TAIL, 2, MAINSG>(op, g, PL, PR, x, cur, nxt, has_next, k, row, fragoff, mode, flagbuf, flag_stride);
    cur = nxt;
    whole = nwhole;
    h = hn;
  }
}
