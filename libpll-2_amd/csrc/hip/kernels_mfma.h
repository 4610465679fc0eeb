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
// Work split: grid = (item blocks, ops, rate categories). A workgroup (4 waves) stages the two
// children's matrices of ITS rate category once as fragments in LDS (2 x 32 KB, two workgroups per
// CU) and every wave then walks its items of 32 sites (2 MFMA site groups). Per item the x
// fragments of one child live in 64 VGPRs, D_left for all 16 state groups in another 64; D_right is
// formed 4 state groups at a time, multiplied, range-checked and stored. The loop is software
// pipelined by hand: the right child's x[jg] is requested as soon as the left pass is done with
// x[jg], and the next item's left x[jg] while the last D_right chunk runs, so that HBM latency
// hides behind ~120 MFMAs with only two waves per SIMD.
//
// Tip children (1-byte codes -> state masks): when every site of an item is a single state or a
// full gap, P x is a column of P (or its row sum) - 16 LDS reads instead of 256 MFMAs per site
// group; other ambiguity patterns take the MFMA route with a 0/1 x.
//
// Scaling: each rate category is a different workgroup, so the "all entries below 2^-256" bits go to
// a byte buffer in HBM ([op][rate][entry]) and k_mfma_scale_epilogue applies them afterwards
// (rescales the - rare - affected entries, fills the parent scaler). Same policy as
// kernels_generic.h; the epilogue is launched only when the level has ops with a parent scaler.
//
// Arithmetic: src/core_partials.c:709-764 (ii), :465-507 (ti), :1166-1209 (tt), :819-879 (repeats).
#pragma once
#include "kernels_common.h"

// LDS stride between the 256 fragments of a matrix: 17 doubles instead of 16 spreads the per-lane
// column reads of the tip route (address ~ state >> 2) over the banks; the MFMA operand reads (16
// consecutive doubles of one fragment, broadcast to the four blocks) do not care
constexpr unsigned kFrag = 17;
// Staging NM transition matrices of one rate category (PT[j][i], row stride SPT) as MFMA fragments in LDS:
// frag[ig][jg][kk][ii] = P[4 ig + ii][4 jg + kk] = PT[4 jg + kk][4 ig + ii], zero beyond S. Every thread of the
// 256 issues ALL its requests (coalesced: thread t takes elements t, t + 256, ...) before the first LDS write - a
// loop of load / wait / write made this 16 L2 round trips long, a fifth of a 61-state workgroup's time.
template <int NG, int NM, unsigned NT = 256u>
__device__ __forceinline__ void mfma_stage(double *const (&dst)[NM], const double *const (&src)[NM], unsigned S, unsigned SPT)
{
  constexpr unsigned W = 4 * NG, N = W * W, PER = (N + NT - 1u) / NT; // NT = threads of the workgroup
  double v[NM][PER];
#pragma unroll
  for (unsigned q = 0; q < PER; ++q)
  {
    const unsigned lin = threadIdx.x + NT * q, j = lin / W, i = lin % W;
    const bool in = lin < N && j < S && i < SPT;
    const size_t off = in ? (size_t)j * SPT + i : 0;
#pragma unroll
    for (int m = 0; m < NM; ++m)
    {
      const double x = src[m][off];
      v[m][q] = in ? x : 0.0;
    }
  }
#pragma unroll
  for (unsigned q = 0; q < PER; ++q)
  {
    const unsigned lin = threadIdx.x + NT * q, j = lin / W, i = lin % W;
    if (lin < N)
    {
      const unsigned pos = ((i >> 2) * NG + (j >> 2)) * kFrag + (j & 3u) * 4u + (i & 3u);
#pragma unroll
      for (int m = 0; m < NM; ++m) dst[m][pos] = v[m][q];
    }
  }
}

// NG = number of 4-state groups the kernels are compiled for: 16 (33..64 states), 8 (21..32), 5 (17..20: the
// 20-state protein models - the coefficient delivery is what the matrix pipe is used for there too, the
// scalar-load-fed FMA contraction of kernels_generic.h reaches a third of the fp64 rate)
template <int NG> struct MfmaGeo
{
  static constexpr unsigned frag_array = NG * NG * kFrag;  // doubles per staged matrix
  static constexpr unsigned rowsum_off = 2 * frag_array;    // doubles: after the two fragment arrays come 2 x 4 NG row sums
  static constexpr unsigned lds_doubles = rowsum_off + 2 * 4 * NG;
  static constexpr int chunk = (NG % 4 == 0 && NG > 8) ? 4 : NG; // parent state groups per D_right pass
};
constexpr unsigned kFragArray = MfmaGeo<16>::frag_array;
constexpr int kMfmaRowsumOff = MfmaGeo<16>::rowsum_off;

struct MfmaItem
{
  unsigned e[2];           // parent entry of the lane's site in site group 0 / 1
  bool valid[2];
  const double *lb[2];     // child entry base (+ rate offset) in the tiled CLV, or null for tips
  const double *rb[2];
  unsigned long long lm[2], rm[2]; // tip state masks
};

template <bool LTIP, bool RTIP, bool GATHER>
__device__ __forceinline__ MfmaItem mfma_item(const DevOp &op, const GenGeo &g, const unsigned long long *__restrict__ tipmap,
                                              unsigned item, unsigned col, unsigned k)
{
  MfmaItem m;
#pragma unroll
  for (int sg = 0; sg < 2; ++sg)
  {
    m.e[sg] = item * 32u + sg * 16u + col;
    m.valid[sg] = m.e[sg] < op.entries;
    const unsigned nn = m.valid[sg] ? m.e[sg] : op.entries - 1;
    unsigned le = nn, re = nn;
    if (GATHER)
    {
      gather_entries(op, nn, le, re);
    }
    m.lm[sg] = m.rm[sg] = 0;
    if (LTIP) m.lm[sg] = tipmap ? tipmap[op.ltip[le]] : (unsigned long long)op.ltip[le];
    if (RTIP) m.rm[sg] = tipmap ? tipmap[op.rtip[re]] : (unsigned long long)op.rtip[re];
    m.lb[sg] = LTIP ? nullptr : op.left + (size_t)(le >> 6) * g.tile_sz + (le & 63u) + (size_t)k * g.S * 64;
    m.rb[sg] = RTIP ? nullptr : op.right + (size_t)(re >> 6) * g.tile_sz + (re & 63u) + (size_t)k * g.S * 64;
  }
  return m;
}

// x fragment value for contraction index j: CLV entry or tip-mask bit
template <bool TIP>
__device__ __forceinline__ double mfma_x(const double *__restrict__ base, unsigned long long mask, unsigned S, unsigned j)
{
  if (TIP) return (j < S && ((mask >> j) & 1ull)) ? 1.0 : 0.0;
  const unsigned jj = j < S ? j : S - 1; // rows beyond S meet zero matrix columns; stay in bounds
  return __builtin_nontemporal_load(base + (size_t)jj * 64);
}

// is every site of the item a single state or a full gap? (wave-uniform answer)
__device__ __forceinline__ bool mfma_simple_tips(const unsigned long long m[2], unsigned long long full)
{
  const bool ok = (__popcll(m[0]) == 1 || m[0] == full) && (__popcll(m[1]) == 1 || m[1] == full);
  return __all(ok);
}

// (P x)[4 ig + row] for a simple tip: column `code` of P, or the row sum for a gap - one LDS read.
// frag[ig][jg][kk][ii] = P[4ig+ii][4jg+kk]; the lane wants P[4ig + row][code]: not this lane's own
// fragment element, so a per-lane address.
template <int NG = 16>
__device__ __forceinline__ double mfma_tip_column(const double *__restrict__ frag, const double *__restrict__ rowsum,
                                                  unsigned long long m, unsigned long long full, unsigned row, int ig)
{
  const unsigned code = (unsigned)__ffsll((long long)m) - 1u;
  const double *p = m == full ? rowsum + 4 * ig + row : frag + (ig * NG + (code >> 2)) * kFrag + (code & 3u) * 4u + row;
  return *p;
}

// The same with the address formed ONCE per (tip, site): where the lane's column starts in the fragment array; state
// group ig is then NG * kFrag doubles further on - an immediate offset. The row sums a full gap stands for sit in the
// fragments' padding slots ((ig, jg = row) slot 16: mfma_rowsums_into_padding), so both cases share that stride.
// (With the column / row-sum choice and the address arithmetic inside the loops over ig the tip x tip launch of C5
// issued 1270 vector-ALU instructions per item of 32 sites and was bound by instruction issue, not by its stores.)
template <int NG>
__device__ __forceinline__ const double *mfma_tip_column_base(const double *__restrict__ frag, unsigned long long m, unsigned long long full,
                                                              unsigned row)
{
  const unsigned code = (unsigned)__ffsll((long long)m) - 1u;
  return m == full ? frag + row * kFrag + 16u : frag + (code >> 2) * kFrag + (code & 3u) * 4u + row;
}

template <int NG>
__device__ __forceinline__ void mfma_rowsums_into_padding(double *__restrict__ frag, const double *__restrict__ rowsum)
{
  // frag[(ig * NG + r) * kFrag + 16] = rowsum[4 ig + r]
  for (unsigned t = threadIdx.x; t < 4u * NG; t += 256u) frag[((t >> 2) * NG + (t & 3u)) * kFrag + 16u] = rowsum[t];
}

template <int NG, bool LTIP, bool RTIP, bool GATHER>
__global__ __launch_bounds__(256, NG > 8 ? 2 : 4) void k_partials_mfma(const OpPack pack, const GenGeo g,
                                                          const unsigned long long *__restrict__ tipmap,
                                                          unsigned items_per_wave, unsigned char *__restrict__ flagbuf,
                                                          unsigned flag_stride /* bytes per (op, rate) */)
{
  extern __shared__ double lds[];
  typedef MfmaGeo<NG> MG;
  constexpr int CH = MG::chunk;
  double *PL = lds;                 // [NG ig][NG jg][4 k][4 i]
  double *PR = lds + MG::frag_array;
  double *RS = lds + MG::rowsum_off; // row sums of P_left [4 NG], P_right [4 NG]

  const unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z; // (item blocks, ops, rates)
  const DevOp &op = pack.ops[by];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned row = lane >> 4;   // k on the input side, i on the output side
  const unsigned col = lane & 15u;  // site within a 16-site group
  const unsigned S = g.S;
  const unsigned k = bz;
  const unsigned nitems = (op.entries + 31u) / 32u;
  if (bx * 4u * items_per_wave >= nitems) return; // whole workgroup
  const int mode = op.pscaler ? g.scale_mode : 0;
  const unsigned fragoff = (row * 4u + (lane & 3u));
  const unsigned long long full = S >= 64 ? ~0ull : ((1ull << S) - 1ull);

  // stage fragments: frag[ig][jg][kk][ii] = P[4ig+ii][4jg+kk] = PT[k][4jg+kk][4ig+ii]
  {
    double *const dst[2] = {PL, PR};
    const double *const src[2] = {op.lmat + (size_t)k * S * g.SPT, op.rmat + (size_t)k * S * g.SPT};
    mfma_stage<NG, 2>(dst, src, S, g.SPT);
  }
  __syncthreads();
  if (LTIP || RTIP)
  {
    if (threadIdx.x < 8 * NG)
    {
      // row sums in ascending j like the reference's set-bit walk (core_partials.c:480-489)
      const double *F = threadIdx.x < 4 * NG ? PL : PR;
      const unsigned i = threadIdx.x % (4 * NG);
      double s = 0.0;
      for (unsigned j = 0; j < S; ++j) s += F[((i >> 2) * NG + (j >> 2)) * kFrag + (j & 3u) * 4 + (i & 3u)];
      RS[threadIdx.x] = s;
    }
    __syncthreads();
    mfma_rowsums_into_padding<NG>(PL, RS);
    mfma_rowsums_into_padding<NG>(PR, RS + 4 * NG);
    __syncthreads();
  }

  const unsigned item0 = (bx * 4u + wave) * items_per_wave;
  if (item0 >= nitems) return; // no barriers below
  const unsigned nmine = min(items_per_wave, nitems - item0);

  // x: CLV fragments of the inner child in flight - the left one, then the right one (ii), or
  // the right one only (ti); tip children take their 0/1 x from the mask on the fly
  MfmaItem cur = mfma_item<LTIP, RTIP, GATHER>(op, g, tipmap, item0, col, k);
  double x[NG][2];
  if (!LTIP || !RTIP)
  {
#pragma unroll
    for (int jg = 0; jg < NG; ++jg)
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) x[jg][sg] = mfma_x<false>(LTIP ? cur.rb[sg] : cur.lb[sg], 0, S, 4 * jg + row);
  }

  for (unsigned it = 0; it < nmine; ++it)
  {
    const bool has_next = it + 1 < nmine;
    MfmaItem nxt = cur;
    if (has_next) nxt = mfma_item<LTIP, RTIP, GATHER>(op, g, tipmap, item0 + it + 1, col, k);

    double DL[NG][2];
    // ---- left child: all 16 parent state groups
    const bool lsimple = LTIP && mfma_simple_tips(cur.lm, full);
    if (lsimple)
    {
      const double *c0 = mfma_tip_column_base<NG>(PL, cur.lm[0], full, row), *c1 = mfma_tip_column_base<NG>(PL, cur.lm[1], full, row);
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
      {
        DL[ig][0] = c0[ig * NG * kFrag];
        DL[ig][1] = c1[ig * NG * kFrag];
      }
    }
    else
    {
#pragma unroll
      for (int ig = 0; ig < NG; ++ig) DL[ig][0] = DL[ig][1] = 0.0;
#pragma unroll
      for (int jg = 0; jg < NG; ++jg)
      {
        const double x0 = LTIP ? mfma_x<true>(nullptr, cur.lm[0], S, 4 * jg + row) : x[jg][0];
        const double x1 = LTIP ? mfma_x<true>(nullptr, cur.lm[1], S, 4 * jg + row) : x[jg][1];
#pragma unroll
        for (int ig = 0; ig < NG; ++ig)
        {
          const double a = PL[(ig * NG + jg) * kFrag + fragoff];
          DL[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x0, DL[ig][0], 0, 0, 0);
          DL[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x1, DL[ig][1], 0, 0, 0);
        }
        // ii: x[jg] of the left child is dead, request the right child's
        if (!LTIP)
        {
          x[jg][0] = mfma_x<false>(cur.rb[0], 0, S, 4 * jg + row);
          x[jg][1] = mfma_x<false>(cur.rb[1], 0, S, 4 * jg + row);
        }
        __builtin_amdgcn_sched_barrier(0); // keep the request here and the fragment look-ahead bounded
      }
    }

    // ---- right child, 4 parent state groups at a time; product, range test, store
    const bool rsimple = RTIP && mfma_simple_tips(cur.rm, full);
    bool small[2] = {true, true};
    double *pb[2];
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
      pb[sg] = op.parent + (size_t)(cur.e[sg] >> 6) * g.tile_sz + (cur.e[sg] & 63u) + (size_t)k * S * 64;
    if (rsimple)
    {
      const double *rc[2] = {mfma_tip_column_base<NG>(PR, cur.rm[0], full, row), mfma_tip_column_base<NG>(PR, cur.rm[1], full, row)};
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
      {
        const unsigned i = 4 * ig + row;
        if (i < S)
        {
#pragma unroll
          for (int sg = 0; sg < 2; ++sg)
          {
            const double v = DL[ig][sg] * rc[sg][ig * NG * kFrag];
            small[sg] = small[sg] && (v < PLLGPU_SCALE_THRESHOLD);
            if (cur.valid[sg]) pb[sg][(size_t)i * 64] = v;
          }
        }
      }
    }
    else
    {
#pragma unroll
      for (int c = 0; c < NG / CH; ++c)
      {
        double DR[CH][2];
#pragma unroll
        for (int q = 0; q < CH; ++q) DR[q][0] = DR[q][1] = 0.0;
#pragma unroll
        for (int jg = 0; jg < NG; ++jg)
        {
          const double x0 = RTIP ? mfma_x<true>(nullptr, cur.rm[0], S, 4 * jg + row) : x[jg][0];
          const double x1 = RTIP ? mfma_x<true>(nullptr, cur.rm[1], S, 4 * jg + row) : x[jg][1];
#pragma unroll
          for (int q = 0; q < CH; ++q)
          {
            const double a = PR[((c * CH + q) * NG + jg) * kFrag + fragoff];
            DR[q][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x0, DR[q][0], 0, 0, 0);
            DR[q][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x1, DR[q][1], 0, 0, 0);
          }
          // last chunk: x[jg] of this item is dead, request the next item's first inner child
          if (c == NG / CH - 1 && !RTIP && has_next)
          {
            x[jg][0] = mfma_x<false>(LTIP ? nxt.rb[0] : nxt.lb[0], 0, S, 4 * jg + row);
            x[jg][1] = mfma_x<false>(LTIP ? nxt.rb[1] : nxt.lb[1], 0, S, 4 * jg + row);
          }
          if (c == NG / CH - 1 || (jg & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < CH; ++q)
        {
          const unsigned i = 4 * (c * CH + q) + row;
          if (i < S)
          {
#pragma unroll
            for (int sg = 0; sg < 2; ++sg)
            {
              const double v = DL[c * CH + q][sg] * DR[q][sg];
              small[sg] = small[sg] && (v < PLLGPU_SCALE_THRESHOLD);
              if (cur.valid[sg]) pb[sg][(size_t)i * 64] = v;
            }
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
        if (row == 0 && cur.valid[sg])
          flagbuf[((size_t)by * g.R + k) * flag_stride + cur.e[sg]] = (unsigned char)s;
      }
    }
    cur = nxt;
  }
}

// Plain tip x tip levels of the large state spaces as what they are: a STORE STREAM (round 6). For a tip that is a single
// state (or a full gap) P x is a column of P (its row sum), so a parent entry is a product of two column elements - no
// contraction at all. k_partials_mfma<.., true, true> already takes the columns from LDS, but in the matrix pipe's lane
// map: a store instruction covers 4 state rows x 16 sites, four 128-byte pieces 512 bytes apart, and a tile's row is
// completed by four instructions of different items (C5: 119.6 us for 625 MB, 0.66 of the HBM peak). Here lane = site:
// the wave that owns (tile, rate) writes its S rows of 64 sites one 512-byte instruction after the other - 31 KB of
// consecutive addresses for 61 states - with two LDS reads and a multiplication per value. The matrices sit in LDS as
// the host stores them (PT[j][i]: row j IS column j of P) at an ODD row stride, so that lanes with different codes
// start in different banks, plus one row of row sums (ascending j, like the reference's set-bit walk,
// src/core_partials.c:480-489 - the value k_partials_mfma uses). Codes with several but not all bits set walk their bits
// in ascending order per lane (rare: not the matrix pipe's summation order - within 1e-15 of it, not the same bits).
// Scaling decisions go to flagbuf for k_mfma_scale_epilogue exactly like k_partials_mfma's.
// grid: 1-D, XCD-aware (kernels_common.h: xcd_linear), logical order rate category fastest - the four workgroups of a
// tile block write one contiguous run; LDS 2 x (S + 1) x LD doubles.
__device__ __forceinline__ unsigned tt_stream_ld(unsigned S) { return (S + 1u) | 1u; }

constexpr unsigned kTtStreamThreads = 1024; // sixteen waves share one copy of the matrices: two workgroups per CU = eight waves per SIMD

struct TtCodes
{
  unsigned long long ml, mr;
};
__device__ __forceinline__ TtCodes tt_stream_codes(const DevOp &op, const unsigned long long *__restrict__ tipmap, unsigned tile, unsigned lane)
{
  const unsigned n = tile * 64u + lane;
  const unsigned nn = n < op.entries ? n : op.entries - 1u;
  TtCodes c;
  c.ml = tipmap ? tipmap[op.ltip[nn]] : (unsigned long long)op.ltip[nn];
  c.mr = tipmap ? tipmap[op.rtip[nn]] : (unsigned long long)op.rtip[nn];
  return c;
}

__global__ __launch_bounds__(kTtStreamThreads) void k_partials_tt_stream(const OpPack pack, const GenGeo g, const unsigned long long *__restrict__ tipmap,
                                                                         unsigned tiles_per_wave, unsigned char *__restrict__ flagbuf, unsigned flag_stride,
                                                                         unsigned nx, unsigned nops, unsigned xcd_order)
{
  extern __shared__ double lds[];
  constexpr unsigned NW = kTtStreamThreads / 64u;
  const unsigned S = g.S, LD = tt_stream_ld(S);
  double *const ML = lds, *const MR = lds + (size_t)(S + 1u) * LD;
  const unsigned l = xcd_linear(nx * nops * g.R, xcd_order);
  if (l == ~0u) return;
  const unsigned k = l % g.R, bx = (l / g.R) % nx, by = l / (g.R * nx);
  const DevOp &op = pack.ops[by];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ntiles = (op.entries + 63u) / 64u;
  const unsigned tile0 = (bx * NW + wave) * tiles_per_wave;
  // the first tile's codes are on their way while the matrices are staged
  TtCodes cur = tt_stream_codes(op, tipmap, tile0 < ntiles ? tile0 : 0u, lane);
  // ---- the two matrices of this rate category: every request before the first LDS write
  {
    constexpr unsigned PER = (64u * 64u + kTtStreamThreads - 1u) / kTtStreamThreads;
    const double *__restrict__ sl = op.lmat + (size_t)k * S * g.SPT, *__restrict__ sr = op.rmat + (size_t)k * S * g.SPT;
    double vl[PER], vr[PER];
#pragma unroll
    for (unsigned q = 0; q < PER; ++q)
    {
      const unsigned lin = threadIdx.x + kTtStreamThreads * q, j = lin / S, i = lin - j * S;
      const bool in = lin < S * S;
      const size_t off = in ? (size_t)j * g.SPT + i : 0;
      vl[q] = sl[off];
      vr[q] = sr[off];
    }
#pragma unroll
    for (unsigned q = 0; q < PER; ++q)
    {
      const unsigned lin = threadIdx.x + kTtStreamThreads * q, j = lin / S, i = lin - j * S;
      if (lin < S * S)
      {
        ML[j * LD + i] = vl[q];
        MR[j * LD + i] = vr[q];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 128u && lane < S)
  {
    // row sums in ascending j (wave 0: left, wave 1: right) into row S
    double *M = wave == 0u ? ML : MR;
    double sum = 0.0;
    for (unsigned j = 0; j < S; ++j) sum += M[j * LD + lane];
    M[S * LD + lane] = sum;
  }
  __syncthreads();

  const int mode = op.pscaler ? g.scale_mode : 0;
  const unsigned long long full = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
  for (unsigned t = 0; t < tiles_per_wave; ++t)
  {
    const unsigned tile = tile0 + t;
    if (tile >= ntiles) break; // wave-uniform; no barriers below
    const unsigned n = tile * 64u + lane;
    const bool valid = n < op.entries;
    const unsigned long long ml = cur.ml, mr = cur.mr;
    if (t + 1u < tiles_per_wave && tile + 1u < ntiles) cur = tt_stream_codes(op, tipmap, tile + 1u, lane); // the next tile's, ahead of this one's stores
    const bool lsimple = __popcll(ml) == 1 || ml == full, rsimple = __popcll(mr) == 1 || mr == full;
    const double *cl = ML + (ml == full ? S : (unsigned)__ffsll((long long)ml) - 1u) * LD;
    const double *cr = MR + (mr == full ? S : (unsigned)__ffsll((long long)mr) - 1u) * LD;
    double *__restrict__ out = op.parent + (size_t)tile * g.tile_sz + (size_t)k * S * 64 + lane;
    bool small = true;
    if (__all(lsimple && rsimple))
    {
      unsigned i = 0;
      for (; i + 4u <= S; i += 4u)
      {
        double v[4];
#pragma unroll
        for (unsigned u = 0; u < 4u; ++u) v[u] = cl[i + u] * cr[i + u];
#pragma unroll
        for (unsigned u = 0; u < 4u; ++u)
        {
          small = small && (v[u] < PLLGPU_SCALE_THRESHOLD);
          if (valid) out[(size_t)(i + u) * 64] = v[u];
        }
      }
      for (; i < S; ++i)
      {
        const double v = cl[i] * cr[i];
        small = small && (v < PLLGPU_SCALE_THRESHOLD);
        if (valid) out[(size_t)i * 64] = v;
      }
    }
    else
    {
      for (unsigned i = 0; i < S; ++i)
      {
        double a, b;
        if (lsimple) a = cl[i];
        else
        {
          a = 0.0;
          for (unsigned long long m = ml & full; m; m &= m - 1ull) a += ML[((unsigned)__ffsll((long long)m) - 1u) * LD + i];
        }
        if (rsimple) b = cr[i];
        else
        {
          b = 0.0;
          for (unsigned long long m = mr & full; m; m &= m - 1ull) b += MR[((unsigned)__ffsll((long long)m) - 1u) * LD + i];
        }
        const double v = a * b;
        small = small && (v < PLLGPU_SCALE_THRESHOLD);
        if (valid) out[(size_t)i * 64] = v;
      }
    }
    if (mode && valid) flagbuf[((size_t)by * g.R + k) * flag_stride + n] = small ? 1u : 0u;
  }
}

// applies the scaling decisions k_partials_mfma left in flagbuf: one thread per parent entry
template <bool GATHER>
__global__ __launch_bounds__(256) void k_mfma_scale_epilogue(const OpPack pack, const GenGeo g,
                                                             const unsigned char *__restrict__ flagbuf, unsigned flag_stride)
{
  const DevOp &op = pack.ops[blockIdx.y];
  if (!op.pscaler || !g.scale_mode) return;
  const unsigned n = blockIdx.x * 256u + threadIdx.x;
  if (n >= op.entries) return;
  const unsigned S = g.S, R = g.R;
  unsigned le = n, re = n;
  if (GATHER)
  {
    gather_entries(op, n, le, re);
  }
  double *base = op.parent + (size_t)(n >> 6) * g.tile_sz + (n & 63u);
  const unsigned char *f = flagbuf + (size_t)blockIdx.y * R * flag_stride + n;
  if (g.scale_mode == 1)
  {
    // (the children's counts are requested before the flags are looked at: one memory round trip, not two - the
    // kernel is nothing but latency, 4 of them per 61-state traversal)
    const unsigned below = (op.lscaler ? op.lscaler[le] : 0u) + (op.rscaler ? op.rscaler[re] : 0u);
    bool all = true;
    for (unsigned k = 0; k < R; ++k) all = all && f[(size_t)k * flag_stride];
    if (all)
      for (unsigned q = 0; q < R * S; ++q) base[(size_t)q * 64] *= PLLGPU_SCALE_FACTOR;
    op.pscaler[n] = below + (all ? 1u : 0u);
  }
  else
  {
    for (unsigned k = 0; k < R; ++k)
    {
      const bool sm = f[(size_t)k * flag_stride] != 0;
      if (sm)
        for (unsigned q = 0; q < S; ++q) base[((size_t)k * S + q) * 64] *= PLLGPU_SCALE_FACTOR;
      op.pscaler[(size_t)n * R + k] = (op.lscaler ? op.lscaler[(size_t)le * R + k] : 0u) +
                                      (op.rscaler ? op.rscaler[(size_t)re * R + k] : 0u) + (sm ? 1u : 0u);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// (tip x tip, tip x tip -> inner x inner) groups on the matrix pipe (C3: 20 states, NG = 5): an op P whose two
// children are cherries produced by the same call is evaluated together with them; nothing is read back from
// HBM. A cherry entry for the lane's states j = 4 jg + row is a product of two tip columns (what
// k_partials_mfma<.., true, true> stores) - and that register layout IS the B operand of the next MFMA stage (D
// and B share their lane map), so P's contraction starts from registers: D_left = P_left x_a, D_right = P_right
// x_b, 2 x NG x NG MFMAs per 16 sites. The launch is store traffic only (3 CLVs per group and site).
//
// What the kernel is built around: instruction count. The group's arithmetic per 32 sites and rate is 100 MFMAs
// (1600 cycles of the matrix pipe) and a quarter of the store time; a first version spent 900 vector-ALU
// instructions per item on addresses, masks and selects and ran at half the store rate. Here
// * the six matrices sit in LDS exactly as the host stores them (PT[j][i], row stride 4 NG) plus one row of row
//   sums: row j IS tip column j, so a tip's contribution is one LDS read at (column of the code) * stride + state
//   with the state group as the instruction's immediate offset, and the same array serves the A operand
//   (element (i, k) of block (ig, jg) at (4 jg + k) * stride + 4 ig + i);
// * a code's column index (state, "row sums" for the full gap, or "ambiguous") and the cherries' scaling
//   decisions (k_cherry_bits) are byte / halfword tables in LDS;
// * a lane owns two ADJACENT sites (2 col, 2 col + 1 of the item's 32): tip codes come in as one halfword per
//   tip, every CLV row goes out as one 16-byte store whose address is a wave-uniform base + a lane constant;
// * the next item's tip codes are requested before the current item is worked on.
// Items with an ambiguous code anywhere take the MFMA route with 0/1 operands for that tip, like the level kernel.
// The arithmetic per op is that of k_partials_mfma, in the same order: bit-identical to the level-by-level
// launches on this pipe. Scaling: the cherries' decisions come from k_cherry_bits (every rate's answer per pair of
// tip codes) and are applied in registers before the parent is formed; the parent's own "all below 2^-256" bits
// go to flagbuf per (group, rate, entry) and k_mfma_scale_epilogue applies them, as for any op of this pipe.
// grid = (item blocks, groups, rate categories).
//
// Which cherries are rescaled? A cherry entry depends on the two tip codes only, so "all S values of rate k
// below 2^-256" is a property of (cherry, rate, code pair): bits[cherry][code_l * ncodes + code_r] has bit k set
// when it holds. k_partials_mfma_cc runs one rate category per workgroup, but the per-site decision needs all of
// them and has to be known BEFORE the parent's contraction (the parent is formed from the rescaled cherry);
// with this small table every workgroup knows every rate's answer. Values as the kernels form them: ascending
// sums over the set bits of each mask, then the product (src/core_partials.c:1166-1209). A table is a function of
// the cherry's two tip matrices and the code map: the host keeps kCherrySlots of them on the device and recomputes
// one only when either matrix was written since (launch_mfma_cc).
constexpr unsigned kCherrySlots = 512; // cherry tables kept on the device (one per pair of tip matrices)

struct CherryTips // by value: tip matrices and table slots of the cherries whose tables are (re)computed
{
  const double *lmat[2 * kMaxGroups];
  const double *rmat[2 * kMaxGroups];
  unsigned short slot[2 * kMaxGroups];
};

// grid = (cherries, rate categories); out[slot][rate][code_l * ncodes + code_r] = 1 when all S values are below 2^-256
// rowsums (or null): [slot][rate][left, right][64] - the row sums of the cherry's two tip matrices (ascending j), what a
// full-gap tip contributes; the 33..64-state group kernel reads its tip columns from HBM / L2 and these beside them
__global__ __launch_bounds__(256) void k_cherry_bits(const CherryTips mats, const GenGeo g, const unsigned long long *__restrict__ tipmap,
                                                      unsigned ncodes, unsigned char *__restrict__ out, double *__restrict__ rowsums,
                                                      unsigned staged /* the two matrices fit into LDS beside the columns */)
{
  extern __shared__ double sh[]; // CL[ncodes][S], CR[ncodes][S]; staged: then ML[S][S], MR[S][S] (PT: [j][i])
  const unsigned c = blockIdx.x, k = blockIdx.y, S = g.S, npairs = ncodes * ncodes;
  double *CL = sh, *CR = CL + ncodes * S;
  const double *lm = mats.lmat[c] + (size_t)k * S * g.SPT, *rm = mats.rmat[c] + (size_t)k * S * g.SPT;
  const double *ML = lm, *MR = rm; // element (j, i) at j * ms + i
  unsigned ms = g.SPT;
  if (staged)
  {
    double *sl = CR + ncodes * S, *sr = sl + S * S;
    for (unsigned idx = threadIdx.x; idx < S * S; idx += 256u)
    {
      const unsigned j = idx / S, i = idx % S;
      sl[idx] = lm[(size_t)j * g.SPT + i];
      sr[idx] = rm[(size_t)j * g.SPT + i];
    }
    ML = sl;
    MR = sr;
    ms = S;
  }
  __syncthreads();
  if (rowsums && threadIdx.x < 128u)
  {
    const unsigned which = threadIdx.x >> 6, i = threadIdx.x & 63u;
    const double *Mx = which ? MR : ML;
    double sum = 0.0;
    if (i < S)
      for (unsigned j = 0; j < S; ++j) sum += Mx[(size_t)j * ms + i];
    rowsums[(((size_t)mats.slot[c] * g.R + k) * 2u + which) * 64u + i] = sum;
  }
  for (unsigned idx = threadIdx.x; idx < ncodes * S; idx += 256u)
  {
    const unsigned code = idx / S, i = idx % S;
    const unsigned long long mask = tipmap ? tipmap[code] : (unsigned long long)code;
    double a = 0.0, b = 0.0;
    // ascending over the set bits; a clear bit adds +0.0, which changes nothing
    for (unsigned m = 0; m < S; ++m)
    {
      const bool on = (mask >> m) & 1ull;
      a += on ? ML[(size_t)m * ms + i] : 0.0;
      b += on ? MR[(size_t)m * ms + i] : 0.0;
    }
    CL[idx] = a;
    CR[idx] = b;
  }
  __syncthreads();
  unsigned char *o = out + ((size_t)mats.slot[c] * g.R + k) * npairs;
  for (unsigned pr = threadIdx.x; pr < npairs; pr += 256u)
  {
    const double *cl = CL + (pr / ncodes) * S, *cr = CR + (pr % ncodes) * S;
    bool small = true;
    for (unsigned i = 0; i < S; ++i) small = small && (cl[i] * cr[i] < PLLGPU_SCALE_THRESHOLD);
    o[pr] = small ? 1 : 0;
  }
}

struct CherrySlots // by value: where the tables of a launch's cherries are (cherry 2 g: group g's left child, 2 g + 1: right)
{
  unsigned short s[2 * kMaxGroups];
};

constexpr unsigned kCcAmbiguous = 255u; // column index of a code that is neither one state nor the full gap

template <int NG> struct CcGeo
{
  static constexpr unsigned LD = 4 * NG + 1;          // row stride (doubles) of a staged matrix: ODD. A lane reads the tip column of ITS site's
                                                      // code - row (column index) x LD + state - so what must not share banks are the ROWS: with 4 NG = 20
                                                      // the 21 rows fall on 8 bank groups and lanes with different codes wait for each other (round-5 counters:
                                                      // 40.9 M conflict cycles against 9 M cycles of active LDS instructions in C3's launch); 21 is coprime to
                                                      // the 32 eight-byte banks. The A-operand reads (row = 4 jg + k, 16 distinct addresses) pay a two-way
                                                      // conflict on three of them instead.
  static constexpr unsigned rows = 4 * NG + 1;        // PT rows j < S, zero rows up to 4 NG, then the row sums
  static constexpr unsigned mat = rows * LD;          // doubles per matrix
  static constexpr unsigned gap_col = 4 * NG;
  static size_t lds_bytes(unsigned ncodes) { return (size_t)6 * mat * sizeof(double) + (size_t)2 * ncodes * ncodes * sizeof(unsigned short) + 256; }
};

template <int NG>
__global__ __launch_bounds__(256, 4) void k_partials_mfma_cc(const FusePack pack, const GenGeo g, const unsigned long long *__restrict__ tipmap,
                                                             unsigned entries, unsigned items_per_wave, unsigned char *__restrict__ flagbuf,
                                                             unsigned flag_stride, const unsigned char *__restrict__ bits, const CherrySlots slots,
                                                             unsigned ncodes, unsigned stream_parent, unsigned nx, unsigned ny, unsigned xcd_order)
{
  // A store-bound launch (kernels_common.h: xcd_block). Logical order with the XCD-aware mapping: rate category fastest,
  // then the item block, then the group - the four workgroups that write the four rate pieces of the same tiles follow
  // each other on one XCD, which so writes whole tiles in runs; natural order (A/B): item block, group, rate.
  unsigned bx, by, k;
  {
    const unsigned l = xcd_linear(nx * ny * g.R, xcd_order);
    if (l == ~0u) return;
    if (xcd_order) k = l % g.R, bx = (l / g.R) % nx, by = l / (g.R * nx);
    else bx = l % nx, by = (l / nx) % ny, k = l / (nx * ny);
  }
  typedef CcGeo<NG> CG;
  constexpr unsigned LD = CG::LD;
  typedef double __attribute__((ext_vector_type(2))) double2v;
  extern __shared__ double lds[];
  double *M = lds;                                                                  // [6][rows][LD]: a.l a.r b.l b.r p.l p.r
  unsigned short *BITS = reinterpret_cast<unsigned short *>(lds + 6u * CG::mat);    // [2][ncodes * ncodes]
  unsigned char *CIDX = reinterpret_cast<unsigned char *>(BITS + 2u * ncodes * ncodes); // [ncodes]

  const FGroup &grp = pack.g[by];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned row = lane >> 4, col = lane & 15u;
  const unsigned S = g.S;
  const unsigned nitems = (entries + 31u) / 32u;
  if (bx * 4u * items_per_wave >= nitems) return; // whole workgroup
  const unsigned long long full = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
  {
    const double *src[6] = {grp.a.lmat, grp.a.rmat, grp.b.lmat, grp.b.rmat, grp.p.lmat, grp.p.rmat};
    for (unsigned idx = threadIdx.x; idx < (CG::rows - 1u) * LD; idx += 256u)
    {
      const unsigned j = idx / LD, i = idx % LD;
      const bool in = j < S && i < S;
      const size_t off = ((size_t)k * S + j) * g.SPT + i;
#pragma unroll
      for (int m = 0; m < 6; ++m) M[m * CG::mat + idx] = in ? src[m][off] : 0.0;
    }
    const unsigned npairs = ncodes * ncodes;
    for (unsigned idx = threadIdx.x; idx < 2u * npairs; idx += 256u)
    {
      const unsigned ch = idx >= npairs ? 1u : 0u, pr = idx - ch * npairs;
      const unsigned char *t = bits + (size_t)slots.s[2u * by + ch] * g.R * npairs + pr;
      unsigned v = 0;
      for (unsigned kk = 0; kk < g.R; ++kk) v |= (unsigned)t[(size_t)kk * npairs] << kk;
      BITS[idx] = (unsigned short)v;
    }
    if (threadIdx.x < ncodes)
    {
      const unsigned long long mk = tipmap[threadIdx.x];
      CIDX[threadIdx.x] = (unsigned char)(mk == full ? CG::gap_col : __popcll(mk) == 1 ? (unsigned)__ffsll((long long)mk) - 1u : kCcAmbiguous);
    }
  }
  __syncthreads();
  if (threadIdx.x < 4u * LD)
  {
    // row sums in ascending j like the reference's set-bit walk (core_partials.c:480-489)
    const unsigned m = threadIdx.x / LD, i = threadIdx.x % LD;
    double s = 0.0;
    for (unsigned j = 0; j < S; ++j) s += M[m * CG::mat + j * LD + i];
    M[m * CG::mat + CG::gap_col * LD + i] = s;
  }
  __syncthreads();
  const int ma = grp.a.pscaler ? g.scale_mode : 0, mb = grp.b.pscaler ? g.scale_mode : 0, mp = grp.p.pscaler ? g.scale_mode : 0;
  const unsigned item0 = (bx * 4u + wave) * items_per_wave;
  if (item0 >= nitems) return; // no barriers below
  const unsigned nmine = min(items_per_wave, nitems - item0);
  const unsigned last_pair = (entries - 1u) & ~1u;
  const unsigned allr = (1u << g.R) - 1u;
  const unsigned char *tips[4] = {grp.a.ltip, grp.a.rtip, grp.b.ltip, grp.b.rtip};
  const double *afrag = M + row * LD + (lane & 3u); // + m * mat + 4 jg * LD + 4 ig: element (i = lane & 3, k = row) of block (ig, jg)
  const unsigned lane_off = row * 64u + 2u * col;    // the lane's place in a CLV tile row group (+ 256 per state group)

  auto fetch_codes = [&](unsigned item, unsigned (&cw)[4]) {
    const unsigned e0 = min(item * 32u + 2u * col, last_pair);
#pragma unroll
    for (int t = 0; t < 4; ++t) cw[t] = *reinterpret_cast<const unsigned short *>(tips[t] + e0);
  };
  unsigned cw[4];
  fetch_codes(item0, cw);

  for (unsigned it = 0; it < nmine; ++it)
  {
    const unsigned item = item0 + it; // wave-uniform
    unsigned nw[4] = {cw[0], cw[1], cw[2], cw[3]};
    if (it + 1 < nmine) fetch_codes(item + 1u, nw);

    const unsigned e0 = item * 32u + 2u * col;
    const bool valid[2] = {e0 < entries, e0 + 1u < entries};
    unsigned code[4][2], cx[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
    {
      code[t][0] = cw[t] & 0xffu;
      code[t][1] = (min(e0, last_pair) + 1u < entries) ? (cw[t] >> 8) : code[t][0];
      cx[t][0] = CIDX[code[t][0]];
      cx[t][1] = CIDX[code[t][1]];
    }
    bool scale_a[2], scale_b[2]; // is the cherry entry rescaled (for this rate category)?
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
    {
      const unsigned ba = BITS[code[0][sg] * ncodes + code[1][sg]];
      const unsigned bb = BITS[ncodes * ncodes + code[2][sg] * ncodes + code[3][sg]];
      scale_a[sg] = ma == 1 ? ba == allr : ma == 2 ? ((ba >> k) & 1u) != 0 : false;
      scale_b[sg] = mb == 1 ? bb == allr : mb == 2 ? ((bb >> k) & 1u) != 0 : false;
    }
    // (P_t x_t) for the lane's states 4 ig + row of both sites: a column of P_t (or its row sums), else MFMAs on 0/1 x
    auto tip_side = [&](int t, double (&d)[NG][2]) {
      const bool simple = __all(cx[t][0] != kCcAmbiguous && cx[t][1] != kCcAmbiguous);
      if (simple)
      {
        const double *c0 = M + t * CG::mat + cx[t][0] * LD + row, *c1 = M + t * CG::mat + cx[t][1] * LD + row;
#pragma unroll
        for (int ig = 0; ig < NG; ++ig)
        {
          d[ig][0] = c0[4 * ig];
          d[ig][1] = c1[4 * ig];
        }
      }
      else
      {
        const unsigned long long m0 = tipmap[code[t][0]], m1 = tipmap[code[t][1]];
#pragma unroll
        for (int ig = 0; ig < NG; ++ig) d[ig][0] = d[ig][1] = 0.0;
#pragma unroll
        for (int jg = 0; jg < NG; ++jg)
        {
          const double x0 = mfma_x<true>(nullptr, m0, S, 4 * jg + row), x1 = mfma_x<true>(nullptr, m1, S, 4 * jg + row);
#pragma unroll
          for (int ig = 0; ig < NG; ++ig)
          {
            const double a = afrag[t * CG::mat + 4 * jg * LD + 4 * ig];
            d[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x0, d[ig][0], 0, 0, 0);
            d[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x1, d[ig][1], 0, 0, 0);
          }
        }
      }
    };
    auto cherry = [&](int t0, double (&x)[NG][2]) {
      double r[NG][2];
      tip_side(t0, x);
      tip_side(t0 + 1, r);
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
      {
        x[ig][0] *= r[ig][0];
        x[ig][1] *= r[ig][1];
      }
    };
    // one CLV row group of an op: states 4 ig + row of the lane's two sites, 16 bytes
    auto put = [&](const FOp &op, const double (&v)[NG][2], bool stream) {
      double *ub = op.parent + (size_t)(item >> 1) * g.tile_sz + (size_t)k * S * 64 + (item & 1u) * 32u; // wave-uniform
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
      {
        if (4u * ig + row < S)
        {
          double *q = ub + (lane_off + 256u * ig);
          if (valid[1])
          {
            double2v w;
            w.x = v[ig][0];
            w.y = v[ig][1];
            if (stream)
              __builtin_nontemporal_store(w, reinterpret_cast<double2v *>(q));
            else
              *reinterpret_cast<double2v *>(q) = w;
          }
          else if (valid[0])
            q[0] = v[ig][0];
        }
      }
    };
    // a cherry's scaler entry: its own decision (children are tips)
    auto put_scaler = [&](const FOp &op, int mode, const bool (&scaled)[2]) {
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
        if (row == 0 && valid[sg])
        {
          if (mode == 2) op.pscaler[(size_t)(e0 + sg) * g.R + k] = scaled[sg] ? 1u : 0u;
          if (mode == 1 && k == 0) op.pscaler[e0 + sg] = scaled[sg] ? 1u : 0u;
        }
    };
    double xa[NG][2], xb[NG][2];
    cherry(0, xa);
    cherry(2, xb);
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
    {
      if (scale_a[sg])
      {
#pragma unroll
        for (int ig = 0; ig < NG; ++ig) xa[ig][sg] *= PLLGPU_SCALE_FACTOR;
      }
      if (scale_b[sg])
      {
#pragma unroll
        for (int ig = 0; ig < NG; ++ig) xb[ig][sg] *= PLLGPU_SCALE_FACTOR;
      }
    }
    put(grp.a, xa, true);
    put(grp.b, xb, true);
    // the parent: both contractions from registers
    double DL[NG][2], DR[NG][2];
#pragma unroll
    for (int ig = 0; ig < NG; ++ig) DL[ig][0] = DL[ig][1] = DR[ig][0] = DR[ig][1] = 0.0;
#pragma unroll
    for (int jg = 0; jg < NG; ++jg)
    {
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
      {
        const double a = afrag[4 * CG::mat + 4 * jg * LD + 4 * ig];
        DL[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, xa[jg][0], DL[ig][0], 0, 0, 0);
        DL[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, xa[jg][1], DL[ig][1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int jg = 0; jg < NG; ++jg)
    {
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
      {
        const double a = afrag[5 * CG::mat + 4 * jg * LD + 4 * ig];
        DR[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, xb[jg][0], DR[ig][0], 0, 0, 0);
        DR[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, xb[jg][1], DR[ig][1], 0, 0, 0);
      }
    }
    bool sp[2] = {true, true};
#pragma unroll
    for (int ig = 0; ig < NG; ++ig)
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
      {
        DL[ig][sg] *= DR[ig][sg];
        if (4 * ig + row < S) sp[sg] = sp[sg] && (DL[ig][sg] < PLLGPU_SCALE_THRESHOLD);
      }
    put(grp.p, DL, stream_parent != 0u);
    if (ma) put_scaler(grp.a, ma, scale_a);
    if (mb) put_scaler(grp.b, mb, scale_b);
    if (mp)
    {
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
      {
        int sm = sp[sg] ? 1 : 0; // a site's states are spread over the four row groups of the wave
        sm &= __shfl_xor(sm, 16, 64);
        sm &= __shfl_xor(sm, 32, 64);
        if (row == 0 && valid[sg]) flagbuf[((size_t)by * g.R + k) * flag_stride + e0 + sg] = (unsigned char)sm;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) cw[t] = nw[t];
  }
}

// ------------------------------------------------------------------------------------------------
// Edge log-likelihood for 33..64 states on the matrix pipe (src/core_likelihood.c:1388-1490 ii,
// :812-915 ti, :1077-1183 repeats). Same fragments and lane maps as k_partials_mfma. One RATE CATEGORY per
// workgroup (grid = item blocks x R, block b works on rate b % R of item block b / R): it stages the edge's
// transition matrix of that rate once, every wave forms D = P x (child side) for its items of 32 sites with MFMAs
// (or reads columns / row sums of P for simple tips), dots it with parent_i * pi_i over the 16 parent states a lane
// owns and leaves the rate-weighted partial per (rate, row group, site) in HBM. The workgroup that finishes an item
// block LAST (a ticket per item block) adds the partials up in a fixed order - rates ascending within a row
// group, then (r0 + r1) + (r2 + r3) - and finishes the sites (scaling undone, invariant share, log, pattern weight).
// (One workgroup walking all R rates of its items, re-staging the matrix for each, left C5's 20k sites with 157
// workgroups on 256 CUs and took 65-71 us; the evaluation is 78 MB of reads and 8 us of MFMAs.)
template <bool CTIP, bool GATHER>
__global__ __launch_bounds__(256, 2) void k_edge_mfma(const DevEdge e, const GenGeo g,
                                                      const unsigned long long *__restrict__ tipmap, unsigned ipw,
                                                      double *__restrict__ partials /* [R][4][pstride] */, unsigned pstride,
                                                      unsigned *__restrict__ tickets /* [item blocks], zero between launches */)
{
  extern __shared__ double lds[];
  __shared__ unsigned finisher;
  double *PM = lds;                 // [16 ig][16 jg][4 k][4 i]
  double *RS = lds + kFragArray;    // row sums [64]
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned row = lane >> 4, col = lane & 15u;
  const unsigned S = g.S, R = g.R;
  const unsigned k = blockIdx.x % R, ib = blockIdx.x / R;
  const unsigned nitems = (e.sites + 31u) / 32u;
  const unsigned fragoff = row * 4u + (lane & 3u);
  const unsigned long long full = S >= 64 ? ~0ull : ((1ull << S) - 1ull);

  {
    double *const dst[1] = {PM};
    const double *const src[1] = {e.mat + (size_t)k * S * g.SPT};
    mfma_stage<16, 1>(dst, src, S, g.SPT);
  }
  __syncthreads();
  if (CTIP)
  {
    if (threadIdx.x < 64)
    {
      const unsigned i = threadIdx.x;
      double s = 0.0;
      for (unsigned j = 0; j < S; ++j) s += PM[((i >> 2) * 16 + (j >> 2)) * kFrag + (j & 3u) * 4 + (i & 3u)];
      RS[i] = s;
    }
    __syncthreads();
    mfma_rowsums_into_padding<16>(PM, RS);
    __syncthreads();
  }
  {
    const unsigned fi = e.fidx[k];
    const double pinv = e.prop_invar ? e.prop_invar[fi] : 0.0;
    const double wk = pinv > 0.0 ? e.rate_weights[k] * (1.0 - pinv) : e.rate_weights[k];
    // pi_i of the 16 parent states this lane owns
    double pif[16];
#pragma unroll
    for (int ig = 0; ig < 16; ++ig) pif[ig] = (4 * ig + row < S) ? e.freqs[(size_t)fi * g.SP + 4 * ig + row] : 0.0;
    const unsigned item0 = (ib * 4u + wave) * ipw;
    double *mine = partials + ((size_t)k * 4u + row) * pstride;

#pragma unroll 1
    for (unsigned it = 0; it < ipw; ++it)
    {
      const unsigned item = item0 + it;
      if (item >= nitems) break; // wave-uniform
      unsigned pe[2], ce[2];
      unsigned long long cm[2] = {0, 0};
      double ex[2];
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
      {
        const unsigned n = item * 32u + sg * 16u + col;
        const unsigned nn = n < e.sites ? n : e.sites - 1;
        pe[sg] = ce[sg] = nn;
        if (GATHER)
        {
          pe[sg] = e.psid ? e.psid[nn] : nn;
          ce[sg] = e.csid ? e.csid[nn] : nn;
        }
        if (CTIP) cm[sg] = tipmap ? tipmap[e.ctip[ce[sg]]] : (unsigned long long)e.ctip[ce[sg]];
        ex[sg] = 1.0;
        if (e.per_rate)
        {
          unsigned mn = 0xFFFFFFFFu, own = 0;
          for (unsigned q = 0; q < R; ++q)
          {
            const unsigned rs = (e.pscaler ? e.pscaler[(size_t)pe[sg] * R + q] : 0u) + (e.cscaler ? e.cscaler[(size_t)ce[sg] * R + q] : 0u);
            mn = min(mn, rs);
            if (q == k) own = rs;
          }
          const unsigned d = min(own - mn, PLLGPU_RATE_MAXDIFF);
          if (d) ex[sg] = minlh(d);
        }
      }
      double D[16][2];
      const bool simple = CTIP && mfma_simple_tips(cm, full);
      if (simple)
      {
        const double *c0 = mfma_tip_column_base<16>(PM, cm[0], full, row), *c1 = mfma_tip_column_base<16>(PM, cm[1], full, row);
#pragma unroll
        for (int ig = 0; ig < 16; ++ig)
        {
          D[ig][0] = c0[ig * 16 * kFrag];
          D[ig][1] = c1[ig * 16 * kFrag];
        }
      }
      else
      {
        double x[16][2];
#pragma unroll
        for (int jg = 0; jg < 16; ++jg)
#pragma unroll
          for (int sg = 0; sg < 2; ++sg)
            x[jg][sg] = CTIP ? mfma_x<true>(nullptr, cm[sg], S, 4 * jg + row)
                             : mfma_x<false>(e.child + tiled_base(ce[sg], g.tile_sz) + (size_t)k * S * 64, 0, S, 4 * jg + row);
#pragma unroll
        for (int ig = 0; ig < 16; ++ig) D[ig][0] = D[ig][1] = 0.0;
#pragma unroll
        for (int jg = 0; jg < 16; ++jg)
        {
#pragma unroll
          for (int ig = 0; ig < 16; ++ig)
          {
            const double a = PM[(ig * 16 + jg) * kFrag + fragoff];
            D[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][0], D[ig][0], 0, 0, 0);
            D[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][1], D[ig][1], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
      {
        const double *pb = e.parent + tiled_base(pe[sg], g.tile_sz) + (size_t)k * S * 64;
        double tr = 0.0;
#pragma unroll
        for (int ig = 0; ig < 16; ++ig)
        {
          const unsigned i = 4 * ig + row;
          const double pv = i < S ? __builtin_nontemporal_load(pb + (size_t)i * 64) : 0.0;
          tr = fma(pv * pif[ig], D[ig][sg], tr);
        }
        const unsigned n = item * 32u + sg * 16u + col;
        if (n < e.sites) partial_store(&mine[n], wk * (tr * ex[sg])); // at the coherent level, like the block sums (kernels_common.h)
      }
    }
  }
  // ---- the last workgroup of this item block finishes its sites. The hand-off follows publish_block_sum's rules:
  // no fences (an agent-scope release writes back the XCD's L2), every wave waits for its own partial stores
  handoff_before_ticket(e.fenced);
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const unsigned t = __hip_atomic_fetch_add(&tickets[ib], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    finisher = (t == R - 1u) ? 1u : 0u;
    if (finisher)
    {
      __hip_atomic_store(&tickets[ib], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      handoff_after_last_ticket(e.fenced);
    }
  }
  __syncthreads();
  double acc = 0.0;
  if (finisher)
  {
    const unsigned n0 = ib * 4u * ipw * 32u, n1 = min(n0 + 4u * ipw * 32u, e.sites);
    for (unsigned n = n0 + threadIdx.x; n < n1; n += 256u)
    {
      // rates ascending within a row group, then the row groups as the shuffles used to pair them
      double tr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
      {
        double t = 0.0;
        for (unsigned q = 0; q < R; ++q) t += partial_load(partials + ((size_t)q * 4u + r) * pstride + n);
        tr[r] = t;
      }
      const double t = (tr[0] + tr[1]) + (tr[2] + tr[3]);
      unsigned pe = n, ce = n;
      if (GATHER)
      {
        pe = e.psid ? e.psid[n] : n;
        ce = e.csid ? e.csid[n] : n;
      }
      unsigned scal;
      if (e.per_rate)
      {
        scal = 0xFFFFFFFFu;
        for (unsigned q = 0; q < R; ++q)
          scal = min(scal, (e.pscaler ? e.pscaler[(size_t)pe * R + q] : 0u) + (e.cscaler ? e.cscaler[(size_t)ce * R + q] : 0u));
      }
      else
        scal = (e.pscaler ? e.pscaler[pe] : 0u) + (e.cscaler ? e.cscaler[ce] : 0u);
      double terminv = 0.0;
      const int inv = e.invariant ? e.invariant[n] : -1;
      if (inv >= 0 && e.prop_invar)
        for (unsigned q = 0; q < R; ++q)
        {
          const unsigned fi = e.fidx[q];
          const double pinv = e.prop_invar[fi];
          if (pinv > 0.0) terminv += e.rate_weights[q] * e.freqs[(size_t)fi * g.SP + inv] * pinv;
        }
      const double site = finish_site(t, terminv, scal, 0) * (double)e.pattern_weights[n];
      if (e.persite) e.persite[n] = site;
      acc += site;
    }
  }
  publish_block_sum_slot(e, wave_sum(acc), 4u, finisher != 0u, ib, (nitems + 4u * ipw - 1u) / (4u * ipw));
}
