// kernels_mfma_wide.h - inner x inner CLV update for 33..64 states (61-state codon models) on the fp64 matrix pipe,
// second generation. Same arithmetic, operand maps and LDS fragments as k_partials_mfma (kernels_mfma.h;
// src/core_partials.c:709-764). What round 3 measured about that kernel and this one (profiles/r3_c5_*.txt,
// profiles/README.md "61 states"):
//
// * A lane owns two ADJACENT sites of a 32-entry half tile (MFMA site group 0 = the even sites, 1 = the odd ones): a
//   child's row group arrives as ONE 16-byte load, a parent's leaves as ONE 16-byte store - half the vector-memory
//   instructions of the first kernel, whole 128-byte lines per row group either way; a child's sixteen requests share
//   one 32-bit lane offset over scalar bases instead of sixteen 64-bit address registers.
// * No padded contraction for 61 states (NGJ = 15, TAIL = 1): the matrix pipe walks the 15 full groups of four
//   contraction states (240 instead of 256 MFMAs per child and site group); the 61st column enters through the vector
//   ALU as the INITIAL value of every accumulator chain: D[i] = P[i][60] x[60], then the MFMAs add j = 0 .. 59 in
//   ascending order (the reference adds j = 60 last, src/core_partials.c:739-757: the same sum, associated
//   differently - well inside the 1e-10 the path is held to). Other state counts run padded (NGJ = 16, TAIL = 0) and
//   are bit-identical to k_partials_mfma. (The 61st PARENT state through the vector ALU as well - 225 MFMAs - was
//   built and measured: no faster, see below, and gone.)
// * The A fragments (4 x 4 blocks of P, one 512-byte LDS read each) go through a ring of eight registers, requested
//   seven blocks ahead of their MFMAs.
// * NO memory instruction sits under a branch or an exec mask, and the child loads are issued and waited for BY HAND
//   (inline asm, counted s_waitcnt). With compiler-managed loads every item began with "s_waitcnt vmcnt(0)": across
//   the loop's back edge - and behind any memory instruction it cannot count, i.e. one under a branch - the compiler
//   waits for EVERYTHING in flight, the parent stores issued moments earlier included. Here a load is waited for with
//   the exact number of younger operations (stores included) allowed to stay in flight; stores are never waited for.
//   What that needs: every item issues the SAME memory instructions in the same order - validity is handled by buffer
//   descriptors that drop lanes beyond their size (state rows beyond S, flags when there is no scaling) and by the
//   fact that CLVs are allocated in whole tiles (entries of the last tile beyond op.entries are padding nobody reads);
//   behind a wave's last item the "next" item is that item again - and NO register may be spilled (a scratch access is
//   a memory operation the count does not know): the host checks the built kernel and falls back to k_partials_mfma
//   if the compiler ever does (pllgpu.hip: wide_kernel_is_sound).
// * Work is dealt in half tiles (32 entries): ONE workgroup of eight waves per CU (the matrices are staged once per CU),
//   one round; the two waves of a SIMD split the SIMD's share, so an odd share costs nobody a whole item (2 ops x
//   20 000 sites x 4 rates = 5000 half tiles over 1024 SIMDs: 5 item times, not the 2 x 3 of equal shares per wave).
//
// What bounds it: ENERGY. tools/mfma61_probe.hip: this loop's MFMAs + LDS reads alone run at 76 TF and 2.39 GHz; with
// the kernel's HBM streams beside them (5.3 TB/s) the chip holds 1.97 GHz and 55 TF. C5's launches: 45-47.5 TF (0.58-0.60
// of the 78.6 TF the matrix pipe has at 2.4 GHz, 0.82-0.86 of what the probe reaches under the same memory load). Taking work
// away (tools/r3_wide_experiments.sh; results wrong on purpose): children served from cache - no change; a quarter
// of the MFMAs gone - 12 % less time; 6 % of the MFMAs moved to the vector ALU (the 61st parent state) - no change;
// no store drain at the top of an item - 3.5 % fewer wave cycles, the same time. Cycles saved come back as clock lost.
//
// Scaling, LDS staging and the flag buffer are k_partials_mfma's; tips and site repeats stay with that kernel.
// (A one-wave-per-SIMD form with whole-tile items - x and D_left of 64 sites in 256 registers, one LDS read per four
// MFMAs - measured slower, 137 against 125 us on C5's launches.)
#pragma once
#include "kernels_mfma.h"

#ifndef WIDE_EXPERIMENT
#define WIDE_EXPERIMENT 0 /* measurement builds only (tools/r3_wide_experiments.sh): 1 = children read from the first 64 tiles only (8 MB: cache hits), 2 = parent stores dropped, 3 = both, 4 = half of the left child's MFMAs skipped. Results are wrong in all of them. */
#endif
typedef double wide_d2 __attribute__((ext_vector_type(2)));
typedef unsigned wide_u4 __attribute__((ext_vector_type(4)));

// ---- hand-counted vector memory -------------------------------------------------------------------------------------
// 16 bytes, streaming, into a register pair the compiler does not track: NOTHING may read `dst` before wide_wait<N>(dst)
__device__ __forceinline__ void wide_load(wide_d2 &dst, const double *p)
{
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
// the same with the address as a wave-uniform base (scalar registers) + a 32-bit lane offset: the sixteen requests of a
// child share ONE address register instead of sixteen 64-bit ones
__device__ __forceinline__ void wide_load(wide_d2 &dst, const double *uniform_base, unsigned lane_byte_off)
{
  asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(lane_byte_off), "s"(uniform_base) : "memory");
}
// until at most N younger vector-memory operations of this wave are in flight; `v` passes through, so every use of it
// stays behind the wait
template <int N>
__device__ __forceinline__ void wide_wait(wide_d2 &v)
{
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}
// 16 bytes through a buffer descriptor: a lane whose offset lies beyond the descriptor's size is dropped by the
// hardware - a store that some lanes must not make costs a select, not a branch around a memory instruction
__device__ __forceinline__ void wide_store(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, double a, double b)
{
  const wide_d2 v = {a, b};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wide_u4, v), rsrc, byte_off, 0, 0);
}


template <int NGJ, int TAIL> struct WideGeo
{
  static constexpr int ngi = 16;               // parent state groups (61 states: the last one holds one live row)
  static constexpr int ch = 4;                 // ... per D_right pass
  static constexpr int tail_stores = ch + 1;   // stores behind an item's last load request: the last pass's rows + the flags
};

// contraction row group jg of a child, sites (2 col, 2 col + 1) of the half tile: one request. `child` = the (tile, rate)
// block, wave-uniform; off = entry offset of the lane's sites in the tile
template <int NGJ, int TAIL>
__device__ __forceinline__ void wide_request(wide_d2 &x, const double *__restrict__ child, unsigned off, unsigned S, int jg, unsigned row)
{
  if (TAIL)
  {
    // every row exists (4 NGJ <= S): uniform base of the row group + the lane's (row, sites) offset; the 61st state is
    // the same row for all four row groups
    if (jg == NGJ) wide_load(x, child + (size_t)(4u * NGJ) * 64u, off * 8u);
    else wide_load(x, child + (size_t)(4u * jg) * 64u, (row * 64u + off) * 8u);
  }
  else
  {
    unsigned j = 4u * jg + row;
    j = j < S ? j : S - 1u; // rows beyond S meet zero matrix columns; stay in bounds
    wide_load(x, child + (size_t)j * 64u + off);
  }
}

// One item = one half tile. On entry x holds (in flight) the LEFT child's rows, requested in the order 61st state,
// 0, 1, ..., followed by WideGeo::tail_stores stores; on exit the same for the next item.
template <int NGJ, int TAIL>
__device__ __forceinline__ void wide_body(const DevOp &op, const GenGeo &g, const double *__restrict__ PL, const double *__restrict__ PR,
                                          wide_d2 (&x)[NGJ + TAIL], unsigned tile, unsigned off, unsigned ntile, unsigned noff, unsigned k,
                                          unsigned row, unsigned fragoff, int mode, unsigned char *__restrict__ flagbuf, unsigned flag_stride)
{
  constexpr int NG = 16;        // the fragment array's row length
  constexpr int NGI = WideGeo<NGJ, TAIL>::ngi;
  constexpr int CH = WideGeo<NGJ, TAIL>::ch; // parent state groups per D_right pass
  constexpr int PF = 8;         // A fragments in flight (ring): block t is requested PF - 1 blocks before its own MFMAs
  // stores an item issues behind its last load request: the parent rows of the last pass, the flags
  constexpr int TAIL_STORES = WideGeo<NGJ, TAIL>::tail_stores;
  // younger operations in flight when row jg of the left child is needed: its NGJ - 1 - jg later rows, the stores behind
  // them, and the right child's requests already made (61st state, rows 0 .. jg - 1) - always the same number
  constexpr int LEFT_WAIT = NGJ - 1 + TAIL + TAIL_STORES;
  const unsigned S = g.S;
  // (tile, ntile, k are the same in every lane of the wave: say so - the bases below live in scalar registers)
  const size_t tile_rate = (size_t)__builtin_amdgcn_readfirstlane(tile) * g.tile_sz + (size_t)k * S * 64u;
  const double *rchild = op.right + ((WIDE_EXPERIMENT & 1) && WIDE_EXPERIMENT < 4 ? (size_t)(__builtin_amdgcn_readfirstlane(tile) & 63u) * g.tile_sz + (size_t)k * S * 64u : tile_rate);
  const double *nleft = op.left + ((WIDE_EXPERIMENT & 1) && WIDE_EXPERIMENT < 4 ? (size_t)(__builtin_amdgcn_readfirstlane(ntile) & 63u) * g.tile_sz : (size_t)__builtin_amdgcn_readfirstlane(ntile) * g.tile_sz) + (size_t)k * S * 64u;
  // this (tile, rate) block of the parent: S rows of 64 entries
  const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(op.parent + tile_rate, 0, (WIDE_EXPERIMENT & 2) && WIDE_EXPERIMENT < 4 ? 0u : S * 512u, 0x00020000);

  // ---- left child: D_left for every parent state group
  double DL[NGI][2];
  if (TAIL) wide_wait<LEFT_WAIT>(x[NGJ]);
#pragma unroll
  for (int ig = 0; ig < NGI; ++ig)
  {
    // TAIL: the chain starts from the 61st column's term, P[4 ig + row][60] x[60] (x[NGJ] holds x[60] in every row group)
    const double c = TAIL ? PL[(ig * NG + NGJ) * kFrag + row] : 0.0;
    DL[ig][0] = TAIL ? c * x[NGJ].x : 0.0;
    DL[ig][1] = TAIL ? c * x[NGJ].y : 0.0;
  }
  if (TAIL) wide_request<NGJ, TAIL>(x[NGJ], rchild, off, S, NGJ, row);
  {
    constexpr int NT = NGJ * NGI;
    double a[PF];
#pragma unroll
    for (int t = 0; t < PF - 1; ++t) a[t] = PL[((t % NGI) * NG + t / NGI) * kFrag + fragoff];
#pragma unroll
    for (int jg = 0; jg < NGJ; ++jg)
    {
      wide_wait<LEFT_WAIT>(x[jg]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ig = 0; ig < NGI; ++ig)
      {
        const int t = jg * NGI + ig, tn = t + PF - 1;
        if (tn < NT) a[tn % PF] = PL[((tn % NGI) * NG + tn / NGI) * kFrag + fragoff];
        if (WIDE_EXPERIMENT == 4 && (ig & 1)) continue; // (half of the left child's MFMAs)
        DL[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t % PF], x[jg].x, DL[ig][0], 0, 0, 0);
        DL[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t % PF], x[jg].y, DL[ig][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      wide_request<NGJ, TAIL>(x[jg], rchild, off, S, jg, row); // the left child's rows jg are dead: the right child's
    }
  }
  // ---- right child, CH parent state groups at a time: product, range test, 16-byte stores
  bool small[2] = {true, true};
  const unsigned e0 = tile * 64u + off;
#pragma unroll
  for (int c = 0; c < NGI / CH; ++c)
  {
    constexpr int LASTC = NGI / CH - 1, NT = NGJ * CH;
    double DR[CH][2];
    if (TAIL && c == 0) wide_wait<NGJ>(x[NGJ]); // behind it: the NGJ other requests for the right child
#pragma unroll
    for (int q = 0; q < CH; ++q)
    {
      const double cc = TAIL ? PR[((c * CH + q) * NG + NGJ) * kFrag + row] : 0.0;
      DR[q][0] = TAIL ? cc * x[NGJ].x : 0.0;
      DR[q][1] = TAIL ? cc * x[NGJ].y : 0.0;
    }
    // the right child's 61st state has seeded its last chains: the next item's left child may have the registers
    if (TAIL && c == LASTC) wide_request<NGJ, TAIL>(x[NGJ], nleft, noff, S, NGJ, row);
    double a[PF];
#pragma unroll
    for (int t = 0; t < PF - 1; ++t) a[t] = PR[((c * CH + t % CH) * NG + t / CH) * kFrag + fragoff];
#pragma unroll
    for (int jg = 0; jg < NGJ; ++jg)
    {
      if (c == 0)
      {
        // the right child's requests were made in the order (61st state,) 0, 1, ...: NGJ - 1 - jg are younger than row jg
        switch (NGJ - 1 - jg)
        {
#define WIDE_W(n) case n: wide_wait<n>(x[jg]); break;
          WIDE_W(0) WIDE_W(1) WIDE_W(2) WIDE_W(3) WIDE_W(4) WIDE_W(5) WIDE_W(6) WIDE_W(7)
          WIDE_W(8) WIDE_W(9) WIDE_W(10) WIDE_W(11) WIDE_W(12) WIDE_W(13) WIDE_W(14) WIDE_W(15)
#undef WIDE_W
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < CH; ++q)
      {
        const int t = jg * CH + q, tn = t + PF - 1;
        if (tn < NT) a[tn % PF] = PR[((c * CH + tn % CH) * NG + tn / CH) * kFrag + fragoff];
        DR[q][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t % PF], x[jg].x, DR[q][0], 0, 0, 0);
        DR[q][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t % PF], x[jg].y, DR[q][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // last chunk: this item is done with rows jg - the next item's left child
      if (c == LASTC) wide_request<NGJ, TAIL>(x[jg], nleft, noff, S, jg, row);
    }
#pragma unroll
    for (int q = 0; q < CH; ++q)
    {
      const int ig = c * CH + q;
      const unsigned i = 4u * ig + row;
      const bool live = (TAIL && ig < NGJ) || i < S; // (TAIL: compile-time true for the NGJ full state groups, 4 NGJ <= S)
      const double v0 = DL[ig][0] * DR[q][0], v1 = DL[ig][1] * DR[q][1];
      small[0] = small[0] && (live ? v0 < PLLGPU_SCALE_THRESHOLD : true);
      small[1] = small[1] && (live ? v1 < PLLGPU_SCALE_THRESHOLD : true);
      wide_store(prsrc, (live ? i * 512u : 0x80000000u) + off * 8u, v0, v1);
    }
  }
  {
    // a site's states are spread over the four row groups of the wave: AND them together. The flags leave through a
    // descriptor as well (no scaling: size 0, every lane dropped; flag_stride covers whole tiles)
    unsigned bits = (small[0] ? 1u : 0u) | (small[1] ? 2u : 0u);
    bits &= (unsigned)__shfl_xor((int)bits, 16, 64);
    bits &= (unsigned)__shfl_xor((int)bits, 32, 64);
    const __amdgpu_buffer_rsrc_t frsrc =
        __builtin_amdgcn_make_buffer_rsrc(flagbuf + ((size_t)blockIdx.y * g.R + k) * flag_stride, 0, mode ? flag_stride : 0u, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)((bits & 1u) | ((bits & 2u) << 7)), frsrc, (row == 0 ? 0u : 0x80000000u) + e0, 0, 0);
  }
}

// grid = (workgroups of WAVES waves, ops, rate categories); a workgroup owns `halves_per_wg` consecutive half tiles of
// its op and deals them to its SIMDs evenly, give or take one (waves w and w + 4 of an 8-wave workgroup sit on one
// SIMD and share its share: the two of them finish together whichever way the odd half tile falls)
template <int NGJ, int TAIL, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void k_partials_mfma_wide(const OpPack pack, const GenGeo g, unsigned halves_per_wg,
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
  const unsigned base = blockIdx.x * halves_per_wg;
  if (base >= nhalves) return; // whole workgroup
  unsigned h, count;
  {
    const unsigned n = min(halves_per_wg, nhalves - base), simd = wave & 3u, q4 = n >> 2, r4 = n & 3u;
    count = q4 + (simd < r4 ? 1u : 0u);
    h = base + simd * q4 + min(simd, r4);
    if (WAVES == 8)
    {
      const unsigned first = (count + 1u) >> 1;
      if (wave >= 4u) h += first, count -= first;
      else count = first;
    }
  }
  const int mode = op.pscaler ? g.scale_mode : 0;
  const unsigned fragoff = row * 4u + (lane & 3u);
  const unsigned h1 = h + count;
  const unsigned hfirst = h < nhalves ? h : nhalves - 1u; // (a wave without work requests like the others and leaves behind the barrier)
  wide_d2 x[NGJ + TAIL];
  {
    // the first item's left child, requested exactly as an item requests its successor's: (61st state,) rows 0, 1, ...,
    // then as many stores as an item leaves behind its last request (dropped: a descriptor of size 0) - the counted
    // waits of wide_body hold from the first item on. Before the matrices are staged: the two latencies overlap.
    const double *left = op.left + (size_t)__builtin_amdgcn_readfirstlane(hfirst >> 1) * g.tile_sz + (size_t)k * S * 64u;
    const unsigned off = (hfirst & 1u) * 32u + 2u * col;
    if (TAIL) wide_request<NGJ, TAIL>(x[NGJ], left, off, S, NGJ, row);
#pragma unroll
    for (int jg = 0; jg < NGJ; ++jg) wide_request<NGJ, TAIL>(x[jg], left, off, S, jg, row);
    const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(op.parent, 0, 0u, 0x00020000);
#pragma unroll
    for (int q = 0; q < WideGeo<NGJ, TAIL>::tail_stores; ++q) wide_store(none, 16u * q, 0.0, (double)q); // (distinct, or the compiler folds them into one)
  }
  {
    double *const dst[2] = {PL, PR};
    const double *const src[2] = {op.lmat + (size_t)k * S * g.SPT, op.rmat + (size_t)k * S * g.SPT};
    mfma_stage<16, 2, 64u * WAVES>(dst, src, S, g.SPT);
  }
  __syncthreads();
  if (count == 0u) return; // no barriers below
  for (; h < h1; ++h)
  {
    const unsigned hn = h + 1u < h1 ? h + 1u : h; // behind the last item: the same item again (loads nobody uses)
    wide_body<NGJ, TAIL>(op, g, PL, PR, x, h >> 1, (h & 1u) * 32u + 2u * col, hn >> 1, (hn & 1u) * 32u + 2u * col, k, row, fragoff, mode,
                         flagbuf, flag_stride);
  }
}
