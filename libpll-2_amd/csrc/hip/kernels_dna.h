// kernels_dna.h - 4 states x 4 rate categories (the headline DNA + Gamma4 configuration), tiled
// sites-contiguous layout clv[tile][rate][state][64 lanes] (kernels_generic.h explains the layout).
//
// Thread mapping: LANE = SITE, one WAVE per 64-site tile, all four rate categories in the lane:
//   * every load/store is a dense 512-byte wave access (64 x 8 B): 32 loads + 16 stores per tile;
//   * the 2 x 4 x 16 transition coefficients of the op are wave-uniform -> scalar loads (1 KB per
//     op, resident in the scalar cache), SGPR operands of v_fma_f64; no LDS, no barriers;
//   * a site's 16 results stay in registers until the per-site scaling decision
//     (src/core_partials.c:729-763) is known: no cross-lane traffic at all.
// An earlier variant with the reference's [site][rate][state] layout and one lane per (site, rate)
// (DPP quad reductions, matrices in VGPRs) measured 64 % of HBM peak on C2; this one is faster
// (profiles/README.md) because each wave instruction touches 4 full cache lines instead of 16 half
// ones and the matrices cost no vector registers.
//
// Algorithmic traffic per site-CLV-update: ii 3*128 B (+4 B per scaler vector touched),
// ti 2*128+1, tt 128+2. Arithmetic: src/core_partials.c:709-764 (ii), :290-351 (ti),
// :1032-1070 + :68-79 (tt; the lookup table is replaced by masked sums), :819-879 (repeats).
#pragma once
#include "kernels_common.h"

constexpr unsigned kDnaTile = 16 * 64; // doubles per tile

// x[j] of one (entry, rate): four dense loads, or the bits of a tip code
template <bool TIP>
__device__ __forceinline__ void dna_fetch(double (&x)[4], const double *__restrict__ base, unsigned k, unsigned code)
{
  if (TIP)
  {
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = (code >> j) & 1u ? 1.0 : 0.0;
  }
  else
  {
    // child CLVs are read exactly once per traversal: streaming (nt) loads keep them from
    // displacing the freshly written parent lines the next level will want from L2/MALL.
    // Measured on C2: 5.4 -> 6.3 TB/s (profiles/README.md); nt on the STORES as well loses it again.
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = __builtin_nontemporal_load(base + (k * 4 + j) * 64);
  }
}

// Class-compressed nodes (site repeats) keep their CLV ENTRY-CONTIGUOUS on the device: [entry][16
// values] - which is also the host layout for 4 states. Children are then addressed through site_id
// maps, and a scattered entry is one 128-byte piece instead of 16 pieces in 16 different rows of a
// tile: tools/gather_probe.hip measures 5.3 TB/s against 0.42 TB/s for a random permutation of 1M
// entries (and no loss for in-order access). The flags below travel in DevOp::layout / DevEdge.
typedef double dbl2 __attribute__((ext_vector_type(2)));
constexpr unsigned kAosLeft = 1u, kAosRight = 2u, kAosParent = 4u, kStreamLeft = 16u, kStreamRight = 32u;

// The 64 entries a wave needs from an entry-contiguous CLV, fetched COOPERATIVELY: eight lanes share
// an entry (16 bytes each), so one load instruction covers eight whole 128-byte entries and every
// line is requested once. (With each lane walking its own entry, the eight instructions of a child
// touch the same 64 lines again and again and lean on the 16 KB vector cache, which two children of
// four or five waves overrun.) issue: the loads; finish: through the wave's LDS buffer ([entry][kAosRow])
// to x[rate][state] of the lane's own entry.
struct DnaCoop
{
  dbl2 piece[8];
};

// stream: the child holds about as many entries as the parent, so every entry is wanted once - streaming
// loads, like the dense kernels. A well-compressed child is a small table that every tile of the parent
// gathers from again and again: it must stay in L2, plain loads (kStreamLeft / kStreamRight, set by the host).
__device__ __forceinline__ void dna_coop_issue(DnaCoop &c, const double *__restrict__ clv, unsigned entry, unsigned lane, bool stream)
{
  const unsigned sub = lane & 7u, grp = lane >> 3;
  if (stream) // wave-uniform
  {
#pragma unroll
    for (unsigned q = 0; q < 8; ++q)
    {
      const unsigned e = __shfl(entry, q * 8u + grp, 64);
      c.piece[q] = __builtin_nontemporal_load(reinterpret_cast<const dbl2 *>(clv + (size_t)e * 16) + sub);
    }
  }
  else
  {
#pragma unroll
    for (unsigned q = 0; q < 8; ++q)
    {
      const unsigned e = __shfl(entry, q * 8u + grp, 64);
      c.piece[q] = *(reinterpret_cast<const dbl2 *>(clv + (size_t)e * 16) + sub);
    }
  }
}

__device__ __forceinline__ void dna_coop_finish(const DnaCoop &c, double *mine /* this wave's LDS buffer */, unsigned lane, unsigned row, double (&x)[4][4])
{
  const unsigned sub = lane & 7u, grp = lane >> 3;
  __builtin_amdgcn_wave_barrier(); // earlier readers of the buffer are done (LDS ops of a wave stay in order)
#pragma unroll
  for (unsigned q = 0; q < 8; ++q) *reinterpret_cast<dbl2 *>(mine + (q * 8u + grp) * row + sub * 2u) = c.piece[q];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const dbl2 *r = reinterpret_cast<const dbl2 *>(mine + lane * row);
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    const dbl2 a = r[2 * k], b = r[2 * k + 1];
    x[k][0] = a.x;
    x[k][1] = a.y;
    x[k][2] = b.x;
    x[k][3] = b.y;
  }
}

// r[i] = sum_j PT[k][j][i] * x[j], coefficients through the scalar path
__device__ __forceinline__ void dna_matvec(double (&r)[4], cdouble_p pt_k, const double (&x)[4])
{
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = fma(pt_k[12 + i], x[3], fma(pt_k[8 + i], x[2], fma(pt_k[4 + i], x[1], pt_k[i] * x[0])));
}

// row stride (doubles) of a wave's LDS transpose buffer: 18 keeps 16-byte accesses of 16 consecutive
// lanes on distinct banks both when a lane writes its own entry and when it reads the dense order
constexpr unsigned kAosRow = 18;

#ifndef DNA_GATHER_WAVES
#define DNA_GATHER_WAVES 3
#endif

// Tiles a wave of the update kernels walks one after the other. Rounds 1-3 sized it so that a launch had about 4096
// workgroups (up to 8 tiles per wave at 1M sites); round 4 measured that against one tile per wave wherever more than
// 4096 workgroups exist (tools/round4_calls/r4_tpw_exp.sh, same box): the seven-op groups at 400k sites 498 -> 468 us
// (6.0 -> 6.4 TB/s), the plain inner x inner level 207 -> 195 us, the 1M-site site-repeats step 0.749 -> 0.711 ms, the
// one-level groups unchanged. A second tile behind the first means a wave whose stores are in flight waits before it may
// ask for the next tile's children; more, shorter waves hide that for each other. The kernels keep the loop.
constexpr unsigned kDnaTilesPerWave = 1;
template <bool LTIP, bool RTIP, bool GATHER>
__global__ __launch_bounds__(256, GATHER ? DNA_GATHER_WAVES : 1) void k_partials_dna(const OpPack pack, int scale_mode, unsigned tiles_per_wave, unsigned nx, unsigned ny,
                                                                                    unsigned xcd_order)
{
  unsigned bx, by;
  if (!xcd_block(nx, ny, xcd_order, bx, by)) return;
  // entry-contiguous parents leave through LDS: a lane holds ITS entry's 128 bytes, but 64 lanes
  // writing 16 bytes each at a 128-byte stride reach only half the store bandwidth of dense 1 KB
  // rows (tools/store_probe.hip: 3.3 vs 6.6 TB/s)
  __shared__ double transpose[GATHER ? 4 * 64 * kAosRow : 1];
  const DevOp &op = pack.ops[by];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ntiles = (op.entries + 63u) / 64u;
  const int mode = op.pscaler ? scale_mode : 0;
  cdouble_p lm = as_const(op.lmat), rm = as_const(op.rmat);

  for (unsigned t = 0; t < tiles_per_wave; ++t)
  {
    const unsigned tile = (bx * 4u + wave) * tiles_per_wave + t;
    if (tile >= ntiles) break; // wave-uniform
    const unsigned n = tile * 64u + lane;
    const bool valid = n < op.entries;
    const unsigned nn = valid ? n : op.entries - 1;
    unsigned le = nn, re = nn;
    if (GATHER)
    {
      gather_entries(op, nn, le, re);
    }
    const unsigned lcode = LTIP ? op.ltip[le] : 0u;
    const unsigned rcode = RTIP ? op.rtip[re] : 0u;
    // class-compressed nodes are entry-contiguous (only reachable with GATHER)
    const bool laos = GATHER && (op.layout & kAosLeft), raos = GATHER && (op.layout & kAosRight), paos = GATHER && (op.layout & kAosParent);
    const double *__restrict__ lx = LTIP ? nullptr : laos ? op.left + (size_t)le * 16 : op.left + (size_t)(le >> 6) * kDnaTile + (le & 63u);
    const double *__restrict__ rx = RTIP ? nullptr : raos ? op.right + (size_t)re * 16 : op.right + (size_t)(re >> 6) * kDnaTile + (re & 63u);
    double *__restrict__ out = op.parent + (size_t)tile * kDnaTile + lane;

    double v[4][4];
    bool small[4];
    if (GATHER)
    {
      // One child after the other: fetch (entry-contiguous: cooperatively, through LDS), contract, and only then
      // touch the second child. With both children's 16 values and both fetches' pieces alive at once the kernel
      // needed 246 registers = two waves per SIMD (the unified file: architected + accumulation registers), and a
      // gather launch lives on the number of waves that wait for memory side by side. (Round 4 requested both children's
      // pieces together, as k_partials_dna_gg now does for its producers: this kernel, with its tiled / entry-contiguous
      // / tip branches alive side by side, spilled - 300 bytes of scratch at the 168 registers of three waves. Not kept.)
      double *mine = transpose + (size_t)wave * 64 * kAosRow;
      double a[4][4];
      {
        double cl[4][4];
        if (!LTIP && laos)
        {
          DnaCoop pl;
          dna_coop_issue(pl, op.left, le, lane, (op.layout & kStreamLeft) != 0);
          dna_coop_finish(pl, mine, lane, kAosRow, cl);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          double xl[4];
          if (!LTIP && laos)
          {
#pragma unroll
            for (int j = 0; j < 4; ++j) xl[j] = cl[k][j];
          }
          else
            dna_fetch<LTIP>(xl, lx, k, lcode);
          dna_matvec(a[k], lm + k * 16, xl);
        }
      }
      __builtin_amdgcn_sched_barrier(0); // the right child's fetch starts after the left child's values are dead
      {
        double cr[4][4];
        if (!RTIP && raos)
        {
          DnaCoop pr;
          dna_coop_issue(pr, op.right, re, lane, (op.layout & kStreamRight) != 0);
          dna_coop_finish(pr, mine, lane, kAosRow, cr);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          double xr[4], b[4];
          if (!RTIP && raos)
          {
#pragma unroll
            for (int j = 0; j < 4; ++j) xr[j] = cr[k][j];
          }
          else
            dna_fetch<RTIP>(xr, rx, k, rcode);
          dna_matvec(b, rm + k * 16, xr);
          small[k] = true;
#pragma unroll
          for (int i = 0; i < 4; ++i)
          {
            v[k][i] = a[k][i] * b[i];
            small[k] = small[k] && (v[k][i] < PLLGPU_SCALE_THRESHOLD);
          }
        }
      }
    }
    else
    {
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        double xl[4], xr[4], a[4], b[4];
        dna_fetch<LTIP>(xl, lx, k, lcode);
        dna_fetch<RTIP>(xr, rx, k, rcode);
        dna_matvec(a, lm + k * 16, xl);
        dna_matvec(b, rm + k * 16, xr);
        small[k] = true;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
          v[k][i] = a[i] * b[i];
          small[k] = small[k] && (v[k][i] < PLLGPU_SCALE_THRESHOLD);
        }
      }
    }
    if (mode == 1)
    {
      const bool s = small[0] && small[1] && small[2] && small[3];
      if (s)
      {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[k][i] *= PLLGPU_SCALE_FACTOR;
      }
      if (valid)
        op.pscaler[n] = (op.lscaler ? op.lscaler[le] : 0u) + (op.rscaler ? op.rscaler[re] : 0u) + (s ? 1u : 0u);
    }
    else if (mode == 2)
    {
      uint4 sc = make_uint4(0, 0, 0, 0);
      if (op.lscaler)
      {
        const uint4 l = reinterpret_cast<const uint4 *>(op.lscaler)[le];
        sc.x += l.x; sc.y += l.y; sc.z += l.z; sc.w += l.w;
      }
      if (op.rscaler)
      {
        const uint4 r = reinterpret_cast<const uint4 *>(op.rscaler)[re];
        sc.x += r.x; sc.y += r.y; sc.z += r.z; sc.w += r.w;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (small[k])
        {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[k][i] *= PLLGPU_SCALE_FACTOR;
        }
      sc.x += small[0]; sc.y += small[1]; sc.z += small[2]; sc.w += small[3];
      if (valid) reinterpret_cast<uint4 *>(op.pscaler)[n] = sc;
    }
    if (valid)
    {
      if (!paos)
      {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 4; ++i) out[(k * 4 + i) * 64] = v[k][i];
      }
    }
    if (GATHER && paos) // wave-uniform
    {
      double *mine = transpose + (size_t)wave * 64 * kAosRow;
      __builtin_amdgcn_wave_barrier(); // the previous tile's reads of this buffer are done
      dbl2 *w = reinterpret_cast<dbl2 *>(mine + lane * kAosRow);
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        dbl2 lo, hi;
        lo.x = v[k][0];
        lo.y = v[k][1];
        hi.x = v[k][2];
        hi.y = v[k][3];
        w[2 * k] = lo;
        w[2 * k + 1] = hi;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      dbl2 *o = reinterpret_cast<dbl2 *>(op.parent + (size_t)tile * 64 * 16);
      const unsigned first = tile * 64u;
#pragma unroll
      for (int q = 0; q < 8; ++q)
      {
        const unsigned t = q * 64u + lane, ent = t >> 3, pair = t & 7u;
        const dbl2 x = *reinterpret_cast<const dbl2 *>(mine + ent * kAosRow + pair * 2);
        if (first + ent < op.entries) o[t] = x;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// edge / root log-likelihood, 4 x 4, tiled. One wave per tile; the lane mixes its site's four rate
// categories in registers, takes the log and accumulates in site order; wave shuffle tree + LDS
// give one partial per workgroup; the last workgroup to arrive adds the partials in index order
// (publish_block_sum: deterministic, single launch).
// Arithmetic: src/core_likelihood.c:1388-1490 (ii), :470-578 (ti 4x4), :1077-1183 (repeats),
// :163-207 (root).
template <bool CTIP, bool GATHER>
__global__ __launch_bounds__(256) void k_edge_dna(const DevEdge e, unsigned tiles_per_wave)
{
  __shared__ double transpose[GATHER ? 4 * 64 * kAosRow : 1]; // entry-contiguous ends: cooperative fetch (dna_coop_issue)
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ntiles = (e.sites + 63u) / 64u;
  cdouble_p pm = as_const(e.mat);
  double acc = 0.0;

  for (unsigned t = 0; t < tiles_per_wave; ++t)
  {
    const unsigned tile = (blockIdx.x * 4u + wave) * tiles_per_wave + t;
    if (tile >= ntiles) break;
    const unsigned n = tile * 64u + lane;
    const bool valid = n < e.sites;
    const unsigned nn = valid ? n : e.sites - 1;
    unsigned pe = nn, ce = nn;
    if (GATHER)
    {
      pe = e.psid ? e.psid[nn] : nn;
      ce = e.csid ? e.csid[nn] : nn;
    }
    const unsigned ccode = CTIP ? e.ctip[ce] : 0u;
    const bool paos = GATHER && (e.layout & kAosParent), caos = GATHER && (e.layout & kAosLeft);
    const double *__restrict__ px = paos ? e.parent + (size_t)pe * 16 : e.parent + (size_t)(pe >> 6) * kDnaTile + (pe & 63u);
    const double *__restrict__ cx = (CTIP || e.is_root) ? nullptr
                                    : caos ? e.child + (size_t)ce * 16 : e.child + (size_t)(ce >> 6) * kDnaTile + (ce & 63u);

    unsigned rs[4] = {0, 0, 0, 0}, scal;
    if (e.per_rate)
    {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        rs[k] = (e.pscaler ? e.pscaler[(size_t)pe * 4 + k] : 0u) + (e.cscaler ? e.cscaler[(size_t)ce * 4 + k] : 0u);
      scal = min(min(rs[0], rs[1]), min(rs[2], rs[3]));
    }
    else
      scal = (e.pscaler ? e.pscaler[pe] : 0u) + (e.cscaler ? e.cscaler[ce] : 0u);
    const int inv = e.invariant ? e.invariant[nn] : -1;

    double cp[4][4], cc[4][4];
    if (GATHER)
    {
      DnaCoop qp, qc;
      double *mine = transpose + (size_t)wave * 64 * kAosRow;
      const bool cfetch = !CTIP && !e.is_root && caos;
      if (paos) dna_coop_issue(qp, e.parent, pe, lane, (e.layout & kStreamLeft) != 0);
      if (cfetch) dna_coop_issue(qc, e.child, ce, lane, (e.layout & kStreamRight) != 0);
      if (paos) dna_coop_finish(qp, mine, lane, kAosRow, cp);
      if (cfetch) dna_coop_finish(qc, mine, lane, kAosRow, cc);
    }
    double terma = 0.0, terminv = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      double xp[4], xc[4], tb[4];
      if (GATHER && paos)
      {
#pragma unroll
        for (int j = 0; j < 4; ++j) xp[j] = cp[k][j];
      }
      else
        dna_fetch<false>(xp, px, k, 0u);
      if (e.is_root)
      {
#pragma unroll
        for (int i = 0; i < 4; ++i) tb[i] = 1.0;
      }
      else
      {
        if (GATHER && !CTIP && caos)
        {
#pragma unroll
          for (int j = 0; j < 4; ++j) xc[j] = cc[k][j];
        }
        else
          dna_fetch<CTIP>(xc, cx, k, ccode);
        dna_matvec(tb, pm + k * 16, xc);
      }
      const unsigned fi = e.fidx[k];
      cdouble_p pi = as_const(e.freqs) + (size_t)fi * 4;
      double tr = fma(xp[3] * pi[3], tb[3], fma(xp[2] * pi[2], tb[2], fma(xp[1] * pi[1], tb[1], (xp[0] * pi[0]) * tb[0])));
      if (e.per_rate)
      {
        const unsigned ex = min(rs[k] - scal, PLLGPU_RATE_MAXDIFF);
        if (ex) tr *= minlh(ex);
      }
      const double pinv = e.prop_invar ? e.prop_invar[fi] : 0.0;
      const double w = e.rate_weights[k];
      if (pinv > 0.0)
      {
        terma += w * tr * (1.0 - pinv);
        if (inv >= 0) terminv += w * e.freqs[(size_t)fi * 4 + inv] * pinv;
      }
      else
        terma += tr * w;
    }
    if (valid)
    {
      const double site = finish_site(terma, terminv, scal, e.is_root) * (double)e.pattern_weights[n];
      if (e.persite) e.persite[n] = site;
      acc += site;
    }
  }
  publish_block_sum(e, wave_sum(acc), 4u);
}

// ------------------------------------------------------------------------------------------------
// Fused groups: a CLV update whose child was produced by another op of the same
// pll_update_partials call does not need to read that child back from HBM. k_partials_dna_fused
// evaluates, per site and entirely in registers, up to three ops - the parent P and the ops A / B
// that produce its left / right child - and stores ALL their CLVs and scalers (callers may read any
// of them later), so the arithmetic and every stored value are those of the unfused launches; only
// the re-reads disappear. Bytes per site of a (tt, tt -> ii) group: 4 + 3 * 128 = 388 instead of
// 2 * 130 + 384 = 644; of an (ii, ii -> ii) group: 4 * 128 + 3 * 128 = 896 instead of 3 * 384 = 1152.
// Grouping is decided per call by plan_fusion() (pllgpu.hip) under the same dependency rules as the
// level scheduler. No site repeats in fused groups (lane = site = entry for all three ops).
enum DnaChildKind
{
  CK_INNER = 0, // child CLV read from HBM
  CK_TIP = 1,   // child tip codes read from HBM
  CK_FTT = 2,   // child computed here from two tips
  CK_FTI = 3,   // child computed here from a tip (left) and an inner CLV (right)
  CK_FII = 4    // child computed here from two inner CLVs
};

struct FOp // one CLV update of a fused group, pointers resolved (80 bytes)
{
  double *parent;
  const double *left, *right;
  const unsigned char *ltip, *rtip;
  unsigned *pscaler;
  const unsigned *lscaler, *rscaler;
  const double *lmat, *rmat;
};

struct FGroup
{
  FOp p;    // the parent op; for a fused child its left/right/lscaler/rscaler fields are not read
  FOp a, b; // producer of the left / right child when that child is fused
};

constexpr int kMaxGroups = 16; // 16 * 240 B = 3840 B of the 4 KiB kernarg segment

struct FusePack
{
  FGroup g[kMaxGroups];
};

// scaling decision + scaler words of one op (src/core_partials.c:729-763): v is rescaled in place,
// sc receives the op's scaler entry (per site in .x, or the four per-rate counts)
__device__ __forceinline__ void dna_scale(double (&v)[4][4], const bool (&small)[4], int mode, uint4 lsc, uint4 rsc, uint4 &sc)
{
  sc = make_uint4(0, 0, 0, 0);
  if (mode == 1)
  {
    const bool s = small[0] && small[1] && small[2] && small[3];
    if (s)
    {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[k][i] *= PLLGPU_SCALE_FACTOR;
    }
    sc.x = lsc.x + rsc.x + (s ? 1u : 0u);
  }
  else if (mode == 2)
  {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (small[k])
      {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[k][i] *= PLLGPU_SCALE_FACTOR;
      }
    sc.x = lsc.x + rsc.x + small[0];
    sc.y = lsc.y + rsc.y + small[1];
    sc.z = lsc.z + rsc.z + small[2];
    sc.w = lsc.w + rsc.w + small[3];
  }
}

__device__ __forceinline__ uint4 dna_load_scaler(const unsigned *s, unsigned n, int scale_mode)
{
  if (!s) return make_uint4(0, 0, 0, 0);
  if (scale_mode == 2) return reinterpret_cast<const uint4 *>(s)[n];
  return make_uint4(s[n], 0, 0, 0);
}

// STREAM: the CLV of a fused child is not read again by this traversal (its consumer took it from
// registers), so it goes out with non-temporal stores and leaves L2/MALL to the group parents the
// next level reads. (For unfused launches non-temporal stores LOSE bandwidth, profiles/README.md -
// there the next level does want the lines.)
template <bool STREAM>
__device__ __forceinline__ void dna_store(const FOp &op, size_t off, unsigned n, bool valid, int mode, const double (&v)[4][4], uint4 sc)
{
  if (!valid) return;
  if (mode == 1) op.pscaler[n] = sc.x;
  if (mode == 2) reinterpret_cast<uint4 *>(op.pscaler)[n] = sc;
  double *__restrict__ out = op.parent + off;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
      if (STREAM)
        __builtin_nontemporal_store(v[k][i], out + (k * 4 + i) * 64);
      else
        out[(k * 4 + i) * 64] = v[k][i];
    }
}

// A child of the group parent in three steps, so that the kernel can put the NEXT loads in front of
// the PREVIOUS stores in program order (the compiler must assume that a store to one CLV buffer may
// alias a later load from another and would otherwise serialise store -> load):
//   load    - issue the HBM reads the child needs (its own children for a fused kind)
//   compute - the child's 16 values + scaler words, the producer's scaling decision applied
//   store   - fused kinds only: the producer op's CLV and scaler go out
struct DnaRaw
{
  double xl[4][4], xr[4][4];
  unsigned lcode, rcode;
};

// parts: 1 = what the producer's left child needs (everything for the memory kinds), 2 = its right
// child's CLV; 3 = both. Splitting lets the kernel keep fewer loads (registers) in flight at a time.
template <int KIND>
__device__ __forceinline__ void dna_child_load(const FOp &pop, bool left_side, const FOp &cop, size_t off, unsigned n, DnaRaw &raw,
                                               int parts = 3)
{
  if (KIND >= CK_FTT)
  {
    constexpr bool LT = (KIND == CK_FTT || KIND == CK_FTI), RT = (KIND == CK_FTT);
    if (parts & 1)
    {
      raw.lcode = LT ? cop.ltip[n] : 0u;
      raw.rcode = RT ? cop.rtip[n] : 0u;
      if (!LT)
      {
#pragma unroll
        for (int k = 0; k < 4; ++k) dna_fetch<false>(raw.xl[k], cop.left + off, k, 0u);
      }
    }
    if ((parts & 2) && !RT)
    {
#pragma unroll
      for (int k = 0; k < 4; ++k) dna_fetch<false>(raw.xr[k], cop.right + off, k, 0u);
    }
    return;
  }
  if (!(parts & 1)) return;
  raw.lcode = raw.rcode = 0u;
  if (KIND == CK_INNER)
  {
    const double *__restrict__ x = (left_side ? pop.left : pop.right) + off;
#pragma unroll
    for (int k = 0; k < 4; ++k) dna_fetch<false>(raw.xl[k], x, k, 0u);
  }
  else if (KIND == CK_TIP)
    raw.lcode = (left_side ? pop.ltip : pop.rtip)[n];
  else
  {
    constexpr bool LT = (KIND == CK_FTT || KIND == CK_FTI), RT = (KIND == CK_FTT);
    if (LT) raw.lcode = cop.ltip[n];
    if (RT) raw.rcode = cop.rtip[n];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      if (!LT) dna_fetch<false>(raw.xl[k], cop.left + off, k, 0u);
      if (!RT) dna_fetch<false>(raw.xr[k], cop.right + off, k, 0u);
    }
  }
}

template <int KIND>
__device__ __forceinline__ void dna_child_compute(const FOp &pop, bool left_side, const FOp &cop, unsigned n, int scale_mode,
                                                  const DnaRaw &raw, double (&v)[4][4], uint4 &sc)
{
  if (KIND == CK_INNER)
  {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[k][i] = raw.xl[k][i];
    sc = dna_load_scaler(left_side ? pop.lscaler : pop.rscaler, n, scale_mode);
  }
  else if (KIND == CK_TIP)
  {
#pragma unroll
    for (int k = 0; k < 4; ++k) dna_fetch<true>(v[k], nullptr, k, raw.lcode);
    sc = make_uint4(0, 0, 0, 0);
  }
  else
  {
    constexpr bool LT = (KIND == CK_FTT || KIND == CK_FTI), RT = (KIND == CK_FTT);
    const int mode = cop.pscaler ? scale_mode : 0;
    cdouble_p lm = as_const(cop.lmat), rm = as_const(cop.rmat);
    bool small[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      double xl[4], xr[4], a[4], b[4];
      if (LT) dna_fetch<true>(xl, nullptr, k, raw.lcode);
      if (RT) dna_fetch<true>(xr, nullptr, k, raw.rcode);
      dna_matvec(a, lm + k * 16, LT ? xl : raw.xl[k]);
      dna_matvec(b, rm + k * 16, RT ? xr : raw.xr[k]);
      small[k] = true;
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
        v[k][i] = a[i] * b[i];
        small[k] = small[k] && (v[k][i] < PLLGPU_SCALE_THRESHOLD);
      }
    }
    const uint4 lsc = dna_load_scaler(LT ? nullptr : cop.lscaler, n, scale_mode);
    const uint4 rsc = dna_load_scaler(RT ? nullptr : cop.rscaler, n, scale_mode);
    dna_scale(v, small, mode, lsc, rsc, sc);
  }
}

template <int KIND>
__device__ __forceinline__ void dna_child_store(const FOp &cop, size_t off, unsigned n, bool valid, int scale_mode,
                                                const double (&v)[4][4], uint4 sc)
{
  if (KIND >= CK_FTT) dna_store<true>(cop, off, n, valid, cop.pscaler ? scale_mode : 0, v, sc);
}

template <int LK, int RK>
__global__ __launch_bounds__(256) void k_partials_dna_fused(const FusePack pack, unsigned entries, int scale_mode, unsigned tiles_per_wave,
                                                              unsigned stream_parent, unsigned nx, unsigned ny, unsigned xcd_order)
{
  unsigned bx, by;
  if (!xcd_block(nx, ny, xcd_order, bx, by)) return;
  const FGroup &g = pack.g[by];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ntiles = (entries + 63u) / 64u;
  const int mode = g.p.pscaler ? scale_mode : 0;
  cdouble_p lm = as_const(g.p.lmat), rm = as_const(g.p.rmat);

  for (unsigned t = 0; t < tiles_per_wave; ++t)
  {
    const unsigned tile = (bx * 4u + wave) * tiles_per_wave + t;
    if (tile >= ntiles) break; // wave-uniform
    const unsigned n0 = tile * 64u + lane;
    const bool valid = n0 < entries;
    const unsigned n = valid ? n0 : entries - 1;
    const size_t off = (size_t)(n >> 6) * kDnaTile + (n & 63u);

    double va[4][4], vb[4][4], v[4][4];
    uint4 sca, scb, sc;
    {
      DnaRaw ra, rb;
      dna_child_load<LK>(g.p, true, g.a, off, n, ra);
      dna_child_compute<LK>(g.p, true, g.a, n, scale_mode, ra, va, sca);
      dna_child_load<RK>(g.p, false, g.b, off, n, rb, 1); // in flight while the left producer's CLV is stored
      dna_child_store<LK>(g.a, off, n, valid, scale_mode, va, sca);
      dna_child_load<RK>(g.p, false, g.b, off, n, rb, 2);
      dna_child_compute<RK>(g.p, false, g.b, n, scale_mode, rb, vb, scb);
      dna_child_store<RK>(g.b, off, n, valid, scale_mode, vb, scb);
    }
    bool small[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      double a[4], b[4];
      dna_matvec(a, lm + k * 16, va[k]);
      dna_matvec(b, rm + k * 16, vb[k]);
      small[k] = true;
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
        v[k][i] = a[i] * b[i];
        small[k] = small[k] && (v[k][i] < PLLGPU_SCALE_THRESHOLD);
      }
    }
    dna_scale(v, small, mode, sca, scb, sc);
    // the group parent is read by the next level: keep it cacheable unless the level's output cannot
    // stay in L2/MALL anyway (stream_parent: decided by the launcher from the bytes written)
    if (stream_parent)
      dna_store<true>(g.p, off, n, valid, mode, v, sc);
    else
      dna_store<false>(g.p, off, n, valid, mode, v, sc);
  }
}

// ------------------------------------------------------------------------------------------------
// Site repeats, where compression ends: a group = an op P (uncompressed, like its children) whose two children A, B
// are GATHERING inner x inner ops - uncompressed parents over class-compressed, entry-contiguous children (C4: the 8
// ops of level 4 over 16 compressed level-3 nodes, and the 4 ops above them). The wave fetches the four compressed
// children's entries of its 64 sites cooperatively (k_partials_dna's gather path, one child after the other), forms
// A and B in registers, stores them (streaming: nothing reads them back in this traversal) and forms P from the
// registers: per site 4 x 128 B gathered + 3 CLVs written instead of 4 x 128 B gathered + 2 written, 2 read, 1 written.
constexpr int CK_FGG = 6;

struct GGroup
{
  DevOp a, b; // the gathering producers (maps, layouts and scalers as for a plain launch)
  FOp p;      // the op over them; its memory-side fields are not read
};

constexpr int kMaxGGroups = 12; // 12 * (2 * 112 + 80) B = 3648 B of kernarg

struct GGPack
{
  GGroup g[kMaxGGroups];
};

// one gathering inner x inner op for the lane's site: its 16 values, scaled, and its scaler words. The caller has
// looked the child entries (le, re) up already and - `pl`, `pr` - requested BOTH children's entries: round 2 fetched one
// child after the other (four dependent round trips per group and tile behind four dependent map look-ups) to stay at
// three waves per SIMD; with the look-ups of both producers first and a producer's two fetches in flight together a
// tile waits three times instead of eight and the kernel still fits three waves (round 4: the shard's launch 57 -> see
// profiles/README.md).
__device__ __forceinline__ void dna_gather_op_finish(const DevOp &op, unsigned le, unsigned re, const DnaCoop &pl, const DnaCoop &pr, unsigned lane,
                                                     double *mine, int scale_mode, double (&v)[4][4], uint4 &sc)
{
  cdouble_p lm = as_const(op.lmat), rm = as_const(op.rmat);
  const int mode = op.pscaler ? scale_mode : 0;
  const uint4 lsc = dna_load_scaler(op.lscaler, le, scale_mode), rsc = dna_load_scaler(op.rscaler, re, scale_mode);
  double a[4][4];
  {
    double cl[4][4];
    dna_coop_finish(pl, mine, lane, kAosRow, cl);
#pragma unroll
    for (int k = 0; k < 4; ++k) dna_matvec(a[k], lm + k * 16, cl[k]);
  }
  bool small[4];
  {
    double cr[4][4];
    dna_coop_finish(pr, mine, lane, kAosRow, cr);
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      double b[4];
      dna_matvec(b, rm + k * 16, cr[k]);
      small[k] = true;
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
        v[k][i] = a[k][i] * b[i];
        small[k] = small[k] && (v[k][i] < PLLGPU_SCALE_THRESHOLD);
      }
    }
  }
  dna_scale(v, small, mode, lsc, rsc, sc);
}

#ifndef DNA_GG_WAVES
#define DNA_GG_WAVES 3
#endif
__global__ __launch_bounds__(256, DNA_GG_WAVES) void k_partials_dna_gg(const GGPack pack, unsigned entries, int scale_mode, unsigned tiles_per_wave,
                                                                       unsigned stream_parent, unsigned nx, unsigned ny, unsigned xcd_order)
{
  __shared__ double transpose[4 * 64 * kAosRow];
  unsigned bx, by;
  if (!xcd_block(nx, ny, xcd_order, bx, by)) return;
  const GGroup &g = pack.g[by];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ntiles = (entries + 63u) / 64u;
  const int mode = g.p.pscaler ? scale_mode : 0;
  cdouble_p lm = as_const(g.p.lmat), rm = as_const(g.p.rmat);
  double *mine = transpose + (size_t)wave * 64 * kAosRow;

  for (unsigned t = 0; t < tiles_per_wave; ++t)
  {
    const unsigned tile = (bx * 4u + wave) * tiles_per_wave + t;
    if (tile >= ntiles) break; // wave-uniform
    const unsigned n0 = tile * 64u + lane;
    const bool valid = n0 < entries;
    const unsigned n = valid ? n0 : entries - 1;
    const size_t off = (size_t)tile * kDnaTile + lane;
    auto put = [&](double *parent, unsigned *pscaler, int m, const double (&x)[4][4], uint4 sc, bool stream) {
      if (!valid) return;
      if (m == 1) pscaler[n] = sc.x;
      if (m == 2) reinterpret_cast<uint4 *>(pscaler)[n] = sc;
      double *__restrict__ out = parent + off;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
          if (stream)
            __builtin_nontemporal_store(x[k][i], out + (k * 4 + i) * 64);
          else
            out[(k * 4 + i) * 64] = x[k][i];
        }
    };
    double va[4][4], v[4][4];
    uint4 sca, scb, sc;
    // the four look-ups first (independent loads), then producer A's two children together
    unsigned ale = n, are = n, ble = n, bre = n;
    gather_entries(g.a, n, ale, are);
    gather_entries(g.b, n, ble, bre);
    {
      DnaCoop al, ar;
      dna_coop_issue(al, g.a.left, ale, lane, (g.a.layout & kStreamLeft) != 0);
      dna_coop_issue(ar, g.a.right, are, lane, (g.a.layout & kStreamRight) != 0);
      dna_gather_op_finish(g.a, ale, are, al, ar, lane, mine, scale_mode, va, sca);
    }
    bool small[4];
    {
      double vb[4][4];
      DnaCoop bl, br;
      // producer B's children are requested before A's CLV leaves: the stores and the loads overlap
      dna_coop_issue(bl, g.b.left, ble, lane, (g.b.layout & kStreamLeft) != 0);
      dna_coop_issue(br, g.b.right, bre, lane, (g.b.layout & kStreamRight) != 0);
      put(g.a.parent, g.a.pscaler, g.a.pscaler ? scale_mode : 0, va, sca, true);
      dna_gather_op_finish(g.b, ble, bre, bl, br, lane, mine, scale_mode, vb, scb);
      put(g.b.parent, g.b.pscaler, g.b.pscaler ? scale_mode : 0, vb, scb, true);
#pragma unroll
      for (int k = 0; k < 4; ++k)
      {
        double a[4], b[4];
        dna_matvec(a, lm + k * 16, va[k]);
        dna_matvec(b, rm + k * 16, vb[k]);
        small[k] = true;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
          v[k][i] = a[i] * b[i];
          small[k] = small[k] && (v[k][i] < PLLGPU_SCALE_THRESHOLD);
        }
      }
    }
    dna_scale(v, small, mode, sca, scb, sc);
    put(g.p.parent, g.p.pscaler, mode, v, sc, stream_parent != 0);
  }
}

// (Round 4 also built this group with a PAIR of waves per tile, as k_partials_dna_cc16 has it - one wave gathers and forms
// producer A, the other B, B's values cross through the wave's transpose buffer: 107 registers, four waves per SIMD, half
// the waits per wave. Same box, alternating: slowest shards 0.1088-0.1125 ms against 0.1077-0.1089 with one wave per
// tile, the 1M-site step 0.741 against 0.735 ms. Occupancy is not what a shard's launch waits for; removed.)
// ------------------------------------------------------------------------------------------------
// Two levels of producers: a child of the group parent P may be an inner x inner op A whose own
// children are both CHERRIES (tip x tip ops) of the level below - kind CK_FCC. The four tip codes are
// all such a child needs from HBM; the two cherries and A are formed in registers and stored like
// every other op. A group with two CK_FCC children is a complete 8-tip subtree: 7 updates for 8 bytes
// read and 7 CLVs written (per site: 8 + 7 x 132 B instead of 2 x 400 + 396 for the same ops as two
// (tt, tt -> ii) groups and a plain op). The other child of P may be a CLV or tip codes in HBM.
constexpr int CK_FCC = 5;

struct TOp // a tip x tip op of a CK_FCC child (48 bytes)
{
  double *parent;
  const unsigned char *ltip, *rtip;
  unsigned *pscaler;
  const double *lmat, *rmat;
};

struct CCGroup
{
  FOp p;          // group parent; for a CK_FCC child its memory-side fields are not read
  FOp a, b;       // the inner x inner producers of the left / right child (CK_FCC sides)
  TOp aa, ab;     // cherries under a: producers of a's left / right child
  TOp ba, bb;     // cherries under b
};

constexpr int kMaxCCGroups = 9; // 9 * 432 B = 3888 B of kernarg

struct CCPack
{
  CCGroup g[kMaxCCGroups];
};

// a cherry from its two tip codes: the arithmetic of dna_child_compute<CK_FTT>, stored streaming
__device__ __forceinline__ void dna_cherry(const TOp &t, size_t off, unsigned n, bool valid, int scale_mode, double (&v)[4][4], uint4 &sc)
{
  const unsigned lcode = t.ltip[n], rcode = t.rtip[n];
  const int mode = t.pscaler ? scale_mode : 0;
  cdouble_p lm = as_const(t.lmat), rm = as_const(t.rmat);
  bool small[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    double xl[4], xr[4], a[4], b[4];
    dna_fetch<true>(xl, nullptr, k, lcode);
    dna_fetch<true>(xr, nullptr, k, rcode);
    dna_matvec(a, lm + k * 16, xl);
    dna_matvec(b, rm + k * 16, xr);
    small[k] = true;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
      v[k][i] = a[i] * b[i];
      small[k] = small[k] && (v[k][i] < PLLGPU_SCALE_THRESHOLD);
    }
  }
  dna_scale(v, small, mode, make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), sc);
  FOp st = {};
  st.parent = t.parent;
  st.pscaler = t.pscaler;
  dna_store<true>(st, off, n, valid, mode, v, sc);
}

// an inner x inner op from two register-resident children (values + scaler words)
__device__ __forceinline__ void dna_combine(const FOp &op, int scale_mode, const double (&va)[4][4], uint4 sca, const double (&vb)[4][4],
                                            uint4 scb, double (&v)[4][4], uint4 &sc, int &mode)
{
  mode = op.pscaler ? scale_mode : 0;
  cdouble_p lm = as_const(op.lmat), rm = as_const(op.rmat);
  bool small[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
  {
    double a[4], b[4];
    dna_matvec(a, lm + k * 16, va[k]);
    dna_matvec(b, rm + k * 16, vb[k]);
    small[k] = true;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
      v[k][i] = a[i] * b[i];
      small[k] = small[k] && (v[k][i] < PLLGPU_SCALE_THRESHOLD);
    }
  }
  dna_scale(v, small, mode, sca, scb, sc);
}

// one child of the group parent: CK_INNER / CK_TIP from HBM, or CK_FCC formed here
template <int KIND>
__device__ __forceinline__ void dna_cc_child(const FOp &pop, bool left_side, const FOp &cop, const TOp &x, const TOp &y, size_t off,
                                             unsigned n, bool valid, int scale_mode, double (&v)[4][4], uint4 &sc)
{
  if (KIND == CK_FCC)
  {
    double vx[4][4], vy[4][4];
    uint4 scx, scy;
    int mode;
    dna_cherry(x, off, n, valid, scale_mode, vx, scx);
    dna_cherry(y, off, n, valid, scale_mode, vy, scy);
    dna_combine(cop, scale_mode, vx, scx, vy, scy, v, sc, mode);
    dna_store<true>(cop, off, n, valid, mode, v, sc);
  }
  else
  {
    DnaRaw raw;
    dna_child_load<KIND>(pop, left_side, cop, off, n, raw);
    dna_child_compute<KIND>(pop, left_side, cop, n, scale_mode, raw, v, sc);
  }
}

template <int LK, int RK>
__global__ __launch_bounds__(256) void k_partials_dna_cc(const CCPack pack, unsigned entries, int scale_mode, unsigned tiles_per_wave,
                                                          unsigned stream_parent, unsigned nx, unsigned ny, unsigned xcd_order)
{
  unsigned bx, by;
  if (!xcd_block(nx, ny, xcd_order, bx, by)) return; // a store-bound launch: every XCD on its own run of tiles (kernels_common.h)
  const CCGroup &g = pack.g[by];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ntiles = (entries + 63u) / 64u;
  for (unsigned t = 0; t < tiles_per_wave; ++t)
  {
    const unsigned tile = (bx * 4u + wave) * tiles_per_wave + t;
    if (tile >= ntiles) break; // wave-uniform
    const unsigned n0 = tile * 64u + lane;
    const bool valid = n0 < entries;
    const unsigned n = valid ? n0 : entries - 1;
    const size_t off = (size_t)(n >> 6) * kDnaTile + (n & 63u);
    double va[4][4], vb[4][4], v[4][4];
    uint4 sca, scb, sc;
    int mode;
    dna_cc_child<LK>(g.p, true, g.a, g.aa, g.ab, off, n, valid, scale_mode, va, sca);
    dna_cc_child<RK>(g.p, false, g.b, g.ba, g.bb, off, n, valid, scale_mode, vb, scb);
    dna_combine(g.p, scale_mode, va, sca, vb, scb, v, sc, mode);
    if (stream_parent)
      dna_store<true>(g.p, off, n, valid, mode, v, sc);
    else
      dna_store<false>(g.p, off, n, valid, mode, v, sc);
  }
}

// ------------------------------------------------------------------------------------------------
// Three levels of producers (round 4): a parent whose two children are the parents of COMPLETE 8-tip subtrees -
// (CK_FCC, CK_FCC) groups - is evaluated with both of them: a complete 16-tip subtree, fifteen ops per site from
// sixteen code bytes, every CLV and scaler stored as always. The left subtree's top CLV waits in registers while the
// right subtree is formed (four CLVs live instead of three). What it saves over two seven-op groups and a later step:
// the two 8-tip tops are not read back (2 x 132 B per site) and their parent's store leaves in the streaming launch
// instead of the latency-bound top of the tree. The arithmetic is dna_cherry / dna_combine's: bit-identical to every
// other route (tests/test_gpu_parity.py::test_fifteen_op_groups_are_bit_identical).
constexpr int CK_F8 = 7; // (host planner) a child that is the parent of a complete (CK_FCC, CK_FCC) group

struct CC16Group // 944 bytes
{
  FOp p;
  CCGroup a, b; // complete 8-tip subtrees: their p is the left / right child of this group's p
};

constexpr int kMaxCC16Groups = 4; // 4 * 944 B = 3776 B of kernarg

struct CC16Pack
{
  CC16Group g[kMaxCC16Groups];
};

// a complete 8-tip subtree for the lane's site: seven ops, all stored (streaming), its top left in registers
__device__ __forceinline__ void dna_cc8(const CCGroup &g, size_t off, unsigned n, bool valid, int scale_mode, double (&v)[4][4], uint4 &sc)
{
  double va[4][4], vb[4][4];
  uint4 sca, scb;
  int mode;
  dna_cc_child<CK_FCC>(g.p, true, g.a, g.aa, g.ab, off, n, valid, scale_mode, va, sca);
  dna_cc_child<CK_FCC>(g.p, false, g.b, g.ba, g.bb, off, n, valid, scale_mode, vb, scb);
  dna_combine(g.p, scale_mode, va, sca, vb, scb, v, sc, mode);
  dna_store<true>(g.p, off, n, valid, mode, v, sc);
}

// Work split: a PAIR of waves per 64-site tile - one forms the left 8-tip subtree, the other the right one, the right
// one's top CLV crosses through LDS and the left wave forms and stores the group parent. (The first version gave a wave
// the whole group: 187 registers = two waves per SIMD and half as many, twice as long waves as the seven-op kernel -
// at 100k sites 151 us for the 60 ops that two seven-op launches' worth of bytes would take 121 us for: the last round
// of such waves runs on a nearly empty chip. At 400k sites the two forms were equal per byte.)
__global__ __launch_bounds__(256) void k_partials_dna_cc16(const CC16Pack pack, unsigned entries, int scale_mode, unsigned stream_parent, unsigned nx,
                                                            unsigned ny, unsigned xcd_order)
{
  __shared__ double xv[2][16][64]; // [tile of the workgroup][value][lane]: the right subtree's top CLV
  __shared__ uint4 xs[2][64];      // ... and its scaler words
  unsigned bx, by;
  if (!xcd_block(nx, ny, xcd_order, bx, by)) return; // store traffic: every XCD on its own run of tiles (kernels_common.h)
  const CC16Group &g = pack.g[by];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned pair = wave >> 1, right = wave & 1u;
  const unsigned ntiles = (entries + 63u) / 64u;
  const unsigned tile0 = bx * 2u + pair;
  const bool live = tile0 < ntiles; // wave-uniform; a pair without a tile still meets the barrier (and stores nothing)
  const unsigned tile = live ? tile0 : ntiles - 1u;
  const unsigned n0 = tile * 64u + lane;
  const bool valid = live && n0 < entries;
  const unsigned n = n0 < entries ? n0 : entries - 1;
  const size_t off = (size_t)(n >> 6) * kDnaTile + (n & 63u);
  double v8[4][4];
  uint4 sc8;
  dna_cc8(right ? g.b : g.a, off, n, valid, scale_mode, v8, sc8);
  if (right)
  {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[pair][k * 4 + i][lane] = v8[k][i];
    xs[pair][lane] = sc8;
  }
  __syncthreads();
  if (right) return;
  double vb[4][4], v[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) vb[k][i] = xv[pair][k * 4 + i][lane];
  const uint4 scb = xs[pair][lane];
  uint4 sc;
  int mode;
  dna_combine(g.p, scale_mode, v8, sc8, vb, scb, v, sc, mode);
  if (stream_parent)
    dna_store<true>(g.p, off, n, valid, mode, v, sc);
  else
    dna_store<false>(g.p, off, n, valid, mode, v, sc);
}

// ------------------------------------------------------------------------------------------------
// Site repeats, the bottom of the tree. Near the tips a class-compressed node holds a handful of
// entries (a DNA cherry: at most 16, a 4-tip clade: 256, an 8-tip clade a few thousand), so the level
// scheduler spends its time on launches that each wait for the one before: C4's shard took 40 us for
// four launches with almost no work in them. Here every op whose subtree consists of tips only and is
// at most three ops deep is evaluated FROM THE TIP CODES: the lane of a parent entry follows the
// entry-indexed child maps (kernels_repeats.h) down to the tip classes, forms the ops on its way back
// up in registers (the arithmetic of the other 4x4 kernels, bit for bit) and stores the CLV and scaler
// of the op it stands for. Nothing such an op reads is produced by this traversal, so ALL of them -
// cherries, the ops above them, the ops above those - run in ONE launch (grid.y = op), each computed
// for its own entries; an op that is the child of another is computed twice (once for each of its own
// entries, once more per entry of its parent - a few thousand entries of arithmetic against a launch).
//
// The subtree of an item is FLATTENED by the host into heap positions: 0 the op itself, 2p+1 / 2p+2 the
// children of position p; positions 0..6 may be ops ("nodes"), 1..14 may be tips. The kernel walks it
// breadth-first - all map look-ups of a level are issued together, then all tip codes - so a lane waits
// for memory four times, not once per op (the first version recursed op by op: 45 us for the same work).
struct SubNode // an op of the subtree: matrices of its two children, entry maps to them
{
  const double *lmat, *rmat;
  const unsigned *lent, *rent; // this op's entry -> child entry; null = the same entry
};

struct SubItem // 384 bytes, read through the scalar path from a device array
{
  double *parent;
  unsigned *pscaler;
  SubNode node[7];
  const unsigned char *tip[14]; // tip[q] = codes of the tip at position q + 1 (one per tip entry)
  unsigned entries;
  unsigned node_mask;   // bit p: position p is an op (bit 0 always)
  unsigned tip_mask;    // bit p: position p is a tip
  unsigned scaler_mask; // bit p: the op at position p has a scaler (its scaling decision is taken)
  unsigned flags;       // kSubAos
  unsigned pad;
  unsigned long long *packed; // [entries]: the 4-bit code of position p in bits 4 p .. 4 p + 3, written by k_sub_pack
};
constexpr unsigned kSubAos = 4u; // the parent is class-compressed: entry-contiguous [entry][16]
typedef const SubItem __attribute__((address_space(4))) *csubitem_p;

// Work split as in the chain kernels further down: a workgroup of four waves owns one 64-entry tile, wave k
// the rate category k (four values per lane and position). The first versions gave a wave all four rates:
// 1900 vector instructions and 112 matrix fetches through the scalar path per tile, one wave at a time per
// SIMD - 46 us for C4's shard, nearly all of it waiting for the scalar loads of the next matrix. A quarter of
// that per wave and four times the waves hide it. What the rates of an entry share is the per-site scaling
// decision (SM 1): one ballot exchange through LDS per LEVEL of the subtree (three barriers).
// (Round 3 tried the matrices through LDS instead of the scalar path - every op's pair requested at once with vector
// loads by lanes 0-31 of the rate's wave, coefficients read back as broadcasts: a shard's launch 27 -> 35 us. The
// scalar path is not what a workgroup's 12 us are made of; removed. Nor is the launch about workgroup count or scalar-cache
// locality: two tiles per workgroup 27 -> 30 us, the tiles dealt to the eight XCDs in contiguous eighths 27.0 -> 27.7 us.
// Counters: 49 scalar loads per wave, 36 % of them missing the scalar cache, waves waiting 53 % of their cycles.)
template <int SM>
__device__ __forceinline__ void dna_sub_level(csubitem_p it, unsigned node_mask, unsigned scaler_mask, unsigned rate, int first, int count,
                                              const double (*below)[4], const unsigned *sc_below, const unsigned (&code)[15], double (*out)[4],
                                              unsigned *sc_out, unsigned long long (*ballots)[4])
{
  // positions first .. first + count - 1; their children sit in below[2 i], below[2 i + 1] (values already scaled)
  bool small[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
  {
    if (i >= count) break;
    const int pos = first + i;
    small[i] = false;
    if ((node_mask >> pos) & 1u) // wave-uniform
    {
      double a[4], b[4];
      dna_matvec(a, as_const(it->node[pos].lmat) + rate * 16u, below[2 * i]);
      dna_matvec(b, as_const(it->node[pos].rmat) + rate * 16u, below[2 * i + 1]);
      bool s = true;
#pragma unroll
      for (int j = 0; j < 4; ++j)
      {
        out[i][j] = a[j] * b[j];
        s = s && (out[i][j] < PLLGPU_SCALE_THRESHOLD);
      }
      small[i] = s && ((scaler_mask >> pos) & 1u);
      if (SM == 1 && ((scaler_mask >> pos) & 1u))
      {
        const unsigned long long mine = __ballot(s);
        if ((threadIdx.x & 63u) == 0u) ballots[pos][rate] = mine;
      }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < 4; ++j) out[i][j] = (code[pos] >> j) & 1u ? 1.0 : 0.0;
    }
  }
  if (SM == 1) lds_barrier(); // workgroup-uniform: every wave of the tile walks the same positions (the ballots only: no wait for memory)
#pragma unroll
  for (int i = 0; i < 4; ++i)
  {
    if (i >= count) break;
    const int pos = first + i;
    sc_out[i] = 0u;
    if (!((node_mask >> pos) & 1u) || !((scaler_mask >> pos) & 1u)) continue;
    bool s = small[i];
    if (SM == 1)
    {
      const unsigned long long all = ballots[pos][0] & ballots[pos][1] & ballots[pos][2] & ballots[pos][3];
      s = (all >> (threadIdx.x & 63u)) & 1ull;
    }
    if (s)
    {
#pragma unroll
      for (int j = 0; j < 4; ++j) out[i][j] *= PLLGPU_SCALE_FACTOR;
    }
    sc_out[i] = sc_below[2 * i] + sc_below[2 * i + 1] + (s ? 1u : 0u);
  }
}

// grid.x = the tiles of the launch's items laid end to end; tiles.first[i] = tiles before item i (by value in
// the kernarg segment: a workgroup finds its item by bisection over words it already has in the scalar cache)
constexpr int kSubItemsPerLaunch = 127;
struct SubTiles
{
  unsigned first[kSubItemsPerLaunch + 1];
};

// what a tile needs before any arithmetic: its item, the entry of every position (a chain of dependent map look-ups,
// level by level) and the codes of the tips
struct SubFetch
{
  csubitem_p it;
  unsigned tile, n;
  bool valid;
  unsigned long long codes; // the 4-bit code of position p in bits 4 p .. 4 p + 3
};

__device__ __forceinline__ void dna_sub_fetch(const SubItem *items, const SubTiles &tiles, unsigned nitems, unsigned gtile, unsigned lane, SubFetch &f)
{
  unsigned lo = 0, hi = nitems; // first[lo] <= gtile < first[hi]
  while (hi - lo > 1u)
  {
    const unsigned mid = (lo + hi) >> 1;
    if (tiles.first[mid] <= gtile) lo = mid; else hi = mid;
  }
  csubitem_p it = (csubitem_p)(uintptr_t)items + lo;
  f.it = it;
  f.tile = gtile - tiles.first[lo];
  const unsigned entries = it->entries;
  const unsigned n0 = f.tile * 64u + lane;
  f.valid = n0 < entries;
  f.n = f.valid ? n0 : entries - 1;
  const unsigned node_mask = it->node_mask, tip_mask = it->tip_mask;
  // breadth-first: the entry of every position, then the codes of the tips
  unsigned e[15];
  e[0] = f.n;
#pragma unroll
  for (int p = 0; p < 7; ++p)
  {
    e[2 * p + 1] = e[2 * p + 2] = 0u;
    if ((node_mask >> p) & 1u)
    {
      const unsigned *lent = it->node[p].lent, *rent = it->node[p].rent;
      e[2 * p + 1] = lent ? lent[e[p]] : e[p];
      e[2 * p + 2] = rent ? rent[e[p]] : e[p];
    }
  }
  unsigned long long codes = 0ull;
#pragma unroll
  for (int p = 1; p < 15; ++p)
    if ((tip_mask >> p) & 1u) codes |= (unsigned long long)(it->tip[p - 1][e[p]] & 15u) << (4 * p);
  f.codes = codes;
}

// The look-ups of an entry - maps level by level, then the tip codes: three or four DEPENDENT loads - do not change
// from one traversal to the next as long as class maps and tip data stand: k_sub_pack does them once and leaves the
// entry's tip codes as one 64-bit word (launched by the host when maps, tips or descriptors changed); the evaluation
// then starts from one coalesced load. (C4's shard: a workgroup lived 12 us, 60 % of it waiting in that chain; the
// launch 35 us for 30 MB of output. Two tiles per workgroup with the second tile's chain in flight was slower: 44 us.)
// `changed` / `since` (class maps rewritten, by the class kernels alone, since the words were formed): the words stand unless one
// of those calls - their sequence numbers start at `since` - left a class -> child entry map other than it found it
// (kernels_repeats.h: RepPack::changed). The reference's pll_update_partials recomputes the class maps of the tree it
// re-evaluates on every call; they come out as they were.
__global__ __launch_bounds__(256) void k_sub_pack(const SubItem *items, const SubTiles tiles, unsigned nitems, unsigned total_tiles, const unsigned *changed,
                                                  unsigned since)
{
  if (changed && (int)(*changed - since) < 0) return;
  const unsigned gtile = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (gtile >= total_tiles) return;
  SubFetch f;
  dna_sub_fetch(items, tiles, nitems, gtile, threadIdx.x & 63u, f);
  if (f.valid) f.it->packed[f.n] = f.codes;
}

template <int SM>
__global__ __launch_bounds__(256) void k_partials_dna_sub(const SubItem *items, const SubTiles tiles, unsigned nitems)
{
  __shared__ unsigned long long ballots[7][4];
  unsigned lo = 0, hi = nitems; // first[lo] <= blockIdx.x < first[hi]
  while (hi - lo > 1u)
  {
    const unsigned mid = (lo + hi) >> 1;
    if (tiles.first[mid] <= blockIdx.x) lo = mid; else hi = mid;
  }
  csubitem_p it = (csubitem_p)(uintptr_t)items + lo;
  const unsigned entries = it->entries;
  const unsigned lane = threadIdx.x & 63u;
  const unsigned rate = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned tile = blockIdx.x - tiles.first[lo];
  const unsigned node_mask = it->node_mask, scaler_mask = it->scaler_mask, flags = it->flags;
  const unsigned n0 = tile * 64u + lane;
  const bool valid = n0 < entries;
  const unsigned n = valid ? n0 : entries - 1;
  const unsigned long long codes = it->packed[n];
  unsigned code[15];
#pragma unroll
  for (int p = 0; p < 15; ++p) code[p] = (unsigned)(codes >> (4 * p)) & 15u;
  // bottom-up, level by level
  double v3[8][4], v2[4][4], v1[2][4], v0[1][4];
  unsigned s3[8], s2[4], s1[2], s0[1];
#pragma unroll
  for (int i = 0; i < 8; ++i)
  {
    s3[i] = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) v3[i][j] = (code[7 + i] >> j) & 1u ? 1.0 : 0.0;
  }
  dna_sub_level<SM>(it, node_mask, scaler_mask, rate, 3, 4, v3, s3, code, v2, s2, ballots);
  dna_sub_level<SM>(it, node_mask, scaler_mask, rate, 1, 2, v2, s2, code, v1, s1, ballots);
  dna_sub_level<SM>(it, node_mask, scaler_mask, rate, 0, 1, v1, s1, code, v0, s0, ballots);
  if (!valid) return;
  unsigned *__restrict__ pscaler = it->pscaler;
  if (pscaler)
  {
    if (SM == 2) pscaler[(size_t)n * 4u + rate] = s0[0];
    else if (rate == 0u) pscaler[n] = s0[0];
  }
  double *__restrict__ parent = it->parent;
  if (flags & kSubAos)
  {
    dbl2 *w = reinterpret_cast<dbl2 *>(parent + (size_t)n * 16 + rate * 4u);
    dbl2 lo2, hi2;
    lo2.x = v0[0][0];
    lo2.y = v0[0][1];
    hi2.x = v0[0][2];
    hi2.y = v0[0][3];
    w[0] = lo2;
    w[1] = hi2;
  }
  else
  {
    double *__restrict__ out = parent + (size_t)tile * kDnaTile + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(rate * 4u + i) * 64] = v0[0][i];
  }
}

// ------------------------------------------------------------------------------------------------
// Chains: a path of ops towards the root, each taking the previous op's CLV as one child, evaluated by
// one wave per 64 sites with that CLV held in registers from step to step. The other child of a step
// (the "sibling") is a LEAF - tip codes or a CLV that is in HBM before the launch starts - or an op
// over two leaves formed on the fly (and stored, like every op). On irregular trees this is what
// removes the one-launch-per-level cost: the level scheduler needs as many launches as the tree is
// deep (a 64-taxon ladder: 62), a chain plan as many as chains are nested (ladder: 1; random
// 64-taxon trees: 3-4), and no CLV on a chain is read back. plan_chains() (pllgpu.hip) decides.
// The arithmetic per op is that of the other 4x4 kernels; the two factors of a parent entry commute,
// so which child rides in registers does not change a bit of the result.
struct ChainLeaf
{
  const void *data;       // const double* (tiled CLV) or const unsigned char* (tip codes)
  const unsigned *scaler; // CLV leaves only; may be null
};

// what a step's sibling is, i.e. what has to be fetched for it (low bits of ChainStepLoad::flags)
enum ChainSiblingKind
{
  CS_T = 0,   // tip codes s0
  CS_C = 1,   // CLV s0
  CS_OTT = 2, // op over two tips s0, s1
  CS_OTC = 3, // op over the tip s0 and the CLV s1
  CS_OCC = 4, // op over two CLVs
  CS_END = 5  // the terminal pseudo-step of a chain: nothing
};
constexpr unsigned kChKindMask = 7u;
constexpr unsigned kChStream = 8u; // nothing reads the step's CLV back in this traversal: streaming stores

// Buffer sizes as the kernel's buffer descriptors want them (see chain_rsrc below), filled by the
// host: `clv` = bytes of the tiled CLV (0 for tip codes); `aux` = bytes behind the leaf's 4-byte side
// load - the codes themselves for a tip (size rounded up to 4), the scaler vector for a CLV (0: none).
struct ChainLeafBytes
{
  unsigned clv, aux, pad;
};

struct ChainStepLoad // what has to be known one step ahead (64 bytes)
{
  ChainLeaf s0, s1;
  ChainLeafBytes b0, b1;
  unsigned flags, pad;
};

struct ChainStepOp // 80 bytes
{
  double *parent;
  unsigned *pscaler;
  const double *mat_acc, *mat_sib; // matrices of the child in registers / of the sibling
  double *bparent;                 // sibling op (kChSibOp): its CLV, scaler, matrices of s0 / s1
  unsigned *bpscaler;
  const double *bmat0, *bmat1;
  unsigned p_bytes, psc_bytes, b_bytes, bsc_bytes;
};

struct ChainHead // 48 bytes
{
  ChainLeaf acc0; // the first step's child that rides in registers
  ChainLeafBytes bacc;
  unsigned first, nsteps; // nsteps real steps, followed by one all-disabled step (the last prefetch)
  unsigned acc_tip, pad0, pad1;
};

typedef const ChainStepLoad __attribute__((address_space(4))) *cstepload_p;
typedef const ChainStepOp __attribute__((address_space(4))) *cstepop_p;
typedef const ChainHead __attribute__((address_space(4))) *chead_p;

// descriptors come through the scalar path (constant address space), field by field
__device__ __forceinline__ ChainStepLoad chain_get(cstepload_p p)
{
  ChainStepLoad r;
  r.s0.data = p->s0.data;
  r.s0.scaler = p->s0.scaler;
  r.s1.data = p->s1.data;
  r.s1.scaler = p->s1.scaler;
  r.b0.clv = p->b0.clv;
  r.b0.aux = p->b0.aux;
  r.b1.clv = p->b1.clv;
  r.b1.aux = p->b1.aux;
  r.b0.pad = r.b1.pad = 0u;
  r.flags = p->flags;
  r.pad = 0u;
  return r;
}

__device__ __forceinline__ ChainStepOp chain_get(cstepop_p p)
{
  ChainStepOp r;
  r.parent = p->parent;
  r.pscaler = p->pscaler;
  r.mat_acc = p->mat_acc;
  r.mat_sib = p->mat_sib;
  r.bparent = p->bparent;
  r.bpscaler = p->bpscaler;
  r.bmat0 = p->bmat0;
  r.bmat1 = p->bmat1;
  r.p_bytes = p->p_bytes;
  r.psc_bytes = p->psc_bytes;
  r.b_bytes = p->b_bytes;
  r.bsc_bytes = p->bsc_bytes;
  return r;
}

// The kinds of a step are only known at run time, but the kernel must not BRANCH around memory
// instructions: at a join the compiler's wait-count bookkeeping assumes the shortest path - the fewest
// instructions issued since the load it waits for - and a step would sit out the latency of the
// loads just issued for the NEXT one (tried: a switch over the kinds, lane masks, separate fetch
// registers per kind - every variant ended with waits on the fresh fetch). So every step issues the
// same memory instructions, through buffer descriptors whose size is 0 for the ones it does not need:
// an out-of-range buffer load returns 0 and an out-of-range store is dropped, neither touches
// memory. Measured (tools/chain_probe2.hip): such stores cost nothing; such LOADS still cost their
// slot on the return path (0.12 us per instruction and step over the whole chip), which is why the
// kernel is compiled in variants that leave out the fetch groups no step of a launch needs
// (template flags C0 / S1 / C1). Run-time branches only ever enclose arithmetic.
typedef unsigned chain_u2 __attribute__((ext_vector_type(2)));
typedef unsigned chain_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_rsrc(const void *p, unsigned bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// Work split: a workgroup of four waves owns 64 sites, wave k the rate category k (four values per
// lane). A chain is sequential per site, so its parallelism is sites x rates and nothing else; with
// one wave per (tile, rate) a 100 000-site chain keeps 6252 waves busy instead of 1563, each small
// enough (registers) for six to share a SIMD - the latency of one wave's loads and stores is covered
// by the others. The only thing the rates of a site share is the per-site scaling decision ("all 16
// values small", SM = 1): the waves exchange their 64-lane ballots through LDS, one barrier per
// decision.
struct ChainGeo
{
  unsigned clv_bytes, entries; // bytes of a tiled CLV buffer, entries per CLV
  unsigned voff;               // this lane's byte offset into a tiled CLV: value (rate, 0) of its site
  unsigned n;                  // its site; stores of lanes past the end go to offsets that are dropped
  unsigned voff_store, n_store;
  unsigned rate;               // wave-uniform
};

struct ChainRaw // what one step fetches
{
  double x0[4], x1[4];
  unsigned side0, side1; // tip: 4 codes (this lane's = byte n & 3); CLV: the site's scaler count (SM 1) or this rate's (SM 2)
};

// SM: the partition's scaling mode (1 = one count per site, 2 = one per site and rate)
// `side` = a leaf's 4-byte side load: four tip codes (this lane's among them) or the CLV's scaler
// entry; CLV = false leaves the four value loads out (no step of the launch has a CLV in this place)
template <int SM, bool CLV>
__device__ __forceinline__ void chain_leaf_issue(const ChainLeaf &leaf, const ChainLeafBytes &bytes, bool tip, const ChainGeo &g, double (&x)[4],
                                                 unsigned &side)
{
  const __amdgpu_buffer_rsrc_t ra = chain_rsrc(tip ? leaf.data : (const void *)leaf.scaler, bytes.aux);
  side = __builtin_amdgcn_raw_buffer_load_b32(ra, tip ? (g.n & ~3u) : (SM == 2 ? g.n * 16u + g.rate * 4u : g.n * 4u), 0, 0);
  if (CLV)
  {
    const __amdgpu_buffer_rsrc_t rc = chain_rsrc(leaf.data, bytes.clv);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      x[i] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rc, g.voff + (unsigned)i * 512u, 0, 2)); // nt
  }
  else
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = 0.0;
  }
}

__device__ __forceinline__ bool chain_s0tip(unsigned kind) { return kind == CS_T || kind == CS_OTT || kind == CS_OTC; }

template <int SM, bool C0, bool S1, bool C1>
__device__ __forceinline__ void chain_issue(const ChainStepLoad &ld, const ChainGeo &g, ChainRaw &raw)
{
  const unsigned kind = ld.flags & kChKindMask;
  chain_leaf_issue<SM, C0>(ld.s0, ld.b0, chain_s0tip(kind), g, raw.x0, raw.side0);
  if (S1)
    chain_leaf_issue<SM, C1>(ld.s1, ld.b1, kind == CS_OTT, g, raw.x1, raw.side1);
  else
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) raw.x1[i] = 0.0;
    raw.side1 = 0u;
  }
}

// what the side load brought: a tip's code -> its 0/1 vector (scaler count 0), a CLV's scaler count
__device__ __forceinline__ unsigned chain_leaf_finish(bool tip, unsigned side, unsigned n, double (&x)[4])
{
  if (!tip) return side;
  const unsigned code = (side >> ((n & 3u) * 8u)) & 0xffu;
#pragma unroll
  for (int j = 0; j < 4; ++j) x[j] = (code >> j) & 1u ? 1.0 : 0.0;
  return 0u;
}

// CLV values of this rate + the scaler entry of one op; sizes of 0 turn the instructions into no-ops
template <int SM, bool STREAM>
__device__ __forceinline__ void chain_store(double *parent, unsigned *pscaler, unsigned clv_bytes, unsigned sc_bytes, const ChainGeo &g,
                                            const double (&v)[4], unsigned sc)
{
  const __amdgpu_buffer_rsrc_t rc = chain_rsrc(parent, clv_bytes);
  // SM 1: the four waves hold the same count; the wave of rate 0 writes it
  const __amdgpu_buffer_rsrc_t rs = chain_rsrc(pscaler, (SM == 2 || g.rate == 0u) ? sc_bytes : 0u);
  __builtin_amdgcn_raw_buffer_store_b32(sc, rs, SM == 2 ? g.n_store * 16u + g.rate * 4u : g.n_store * 4u, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(chain_u2, v[i]), rc, g.voff_store + (unsigned)i * 512u, 0, STREAM ? 2 : 0);
}

// one CLV update for this wave's rate: v = (ma . xa) * (mb . xb), scaled like src/core_partials.c:729-763
template <int SM>
__device__ __forceinline__ void chain_combine(const double *ma, const double *mb, const unsigned *pscaler, const ChainGeo &g,
                                              const double (&xa)[4], unsigned sca, const double (&xb)[4], unsigned scb, double (&v)[4],
                                              unsigned &sc, unsigned long long (*ballots)[4], unsigned slot)
{
  double a[4], b[4];
  dna_matvec(a, as_const(ma) + g.rate * 16u, xa);
  dna_matvec(b, as_const(mb) + g.rate * 16u, xb);
  bool small = true;
#pragma unroll
  for (int i = 0; i < 4; ++i)
  {
    v[i] = a[i] * b[i];
    small = small && (v[i] < PLLGPU_SCALE_THRESHOLD);
  }
  sc = 0u;
  if (!pscaler) return; // no scaler on this op: nothing is rescaled (workgroup-uniform)
  if (SM == 1)
  {
    const unsigned long long mine = __ballot(small);
    if ((threadIdx.x & 63u) == 0u) ballots[slot][g.rate] = mine;
    __syncthreads();
    const unsigned long long all = ballots[slot][0] & ballots[slot][1] & ballots[slot][2] & ballots[slot][3];
    small = (all >> (threadIdx.x & 63u)) & 1ull;
  }
  if (small)
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= PLLGPU_SCALE_FACTOR;
  }
  sc = sca + scb + (small ? 1u : 0u);
}

// one step: sibling from `cur` (fetched one step earlier), parent from (acc, sibling); acc <- parent
template <int SM>
__device__ __forceinline__ void chain_step(const ChainStepOp &op, unsigned flags, const ChainGeo &g, ChainRaw &cur, double (&acc)[4], unsigned &sca,
                                           unsigned long long (*ballots)[4], unsigned step)
{
  const unsigned kind = flags & kChKindMask;
  const bool s0tip = chain_s0tip(kind), s1tip = kind == CS_OTT, sibop = kind >= CS_OTT && kind != CS_END;
  const unsigned sc0 = chain_leaf_finish(s0tip, cur.side0, g.n, cur.x0);
  unsigned scb = sc0;
  if (sibop) // arithmetic (and the barrier of the scaling decision) only
  {
    const unsigned sc1 = chain_leaf_finish(s1tip, cur.side1, g.n, cur.x1);
    double vb[4];
    chain_combine<SM>(op.bmat0, op.bmat1, op.bpscaler, g, cur.x0, sc0, cur.x1, sc1, vb, scb, ballots, (step & 1u) * 2u);
#pragma unroll
    for (int i = 0; i < 4; ++i) cur.x0[i] = vb[i];
  }
  chain_store<SM, true>(op.bparent, op.bpscaler, op.b_bytes, op.bsc_bytes, g, cur.x0, scb);
  double v[4];
  unsigned sc;
  chain_combine<SM>(op.mat_acc, op.mat_sib, op.pscaler, g, acc, sca, cur.x0, scb, v, sc, ballots, (step & 1u) * 2u + 1u);
  if (flags & kChStream) // both sides issue the same instructions
    chain_store<SM, true>(op.parent, op.pscaler, op.p_bytes, op.psc_bytes, g, v, sc);
  else
    chain_store<SM, false>(op.parent, op.pscaler, op.p_bytes, op.psc_bytes, g, v, sc);
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = v[i];
  sca = sc;
}

constexpr int kChainPackHeads = 4, kChainPackSteps = 24; // a ChainPack: 4 * 48 + 24 * (64 + 80) = 3648 B of kernarg (+ a DevEdge in the tail kernel)

// where the descriptors come from: device memory (any size) ...
struct ChainSrcMem
{
  chead_p heads;
  cstepload_p loads;
  cstepop_p ops;
  __device__ __forceinline__ ChainHead head(unsigned c) const
  {
    ChainHead h;
    h.acc0.data = heads[c].acc0.data;
    h.acc0.scaler = heads[c].acc0.scaler;
    h.bacc.clv = heads[c].bacc.clv;
    h.bacc.aux = heads[c].bacc.aux;
    h.bacc.pad = 0u;
    h.first = heads[c].first;
    h.nsteps = heads[c].nsteps;
    h.acc_tip = heads[c].acc_tip;
    h.pad0 = h.pad1 = 0u;
    return h;
  }
  __device__ __forceinline__ ChainStepLoad load(unsigned s) const { return chain_get(loads + s); }
  __device__ __forceinline__ ChainStepOp op(unsigned s) const { return chain_get(ops + s); }
};

// ... or the kernarg segment (small plans: no descriptor upload)
struct ChainPack
{
  ChainHead heads[kChainPackHeads];
  ChainStepLoad loads[kChainPackSteps];
  ChainStepOp ops[kChainPackSteps];
};

struct ChainSrcPack
{
  const ChainPack &pack;
  __device__ __forceinline__ ChainHead head(unsigned c) const { return pack.heads[c]; }
  __device__ __forceinline__ ChainStepLoad load(unsigned s) const { return pack.loads[s]; }
  __device__ __forceinline__ ChainStepOp op(unsigned s) const { return pack.ops[s]; }
};

// A chain for this wave's (tile, rate) in two parts, so that a kernel that walks two chains can have
// the second one's first loads in flight while it walks the first:
//   chain_begin - issue the loads of the child that rides in registers and the first step's fetch
//   chain_steps - walk the steps; the top CLV values and scaler count are left in acc / sca
struct ChainStart
{
  ChainRaw ra;
  unsigned fa, side;
};

template <int SM, bool C0, bool S1, bool C1, class SRC>
__device__ __forceinline__ void chain_begin(const SRC &src, const ChainHead &h, const ChainGeo &g, double (&acc)[4], ChainStart &st)
{
  chain_leaf_issue<SM, true>(h.acc0, h.bacc, h.acc_tip != 0u, g, acc, st.side);
  const ChainStepLoad l0 = src.load(h.first);
  st.fa = l0.flags;
  chain_issue<SM, C0, S1, C1>(l0, g, st.ra);
}

template <int SM, bool C0, bool S1, bool C1, class SRC>
__device__ __forceinline__ void chain_steps(const SRC &src, const ChainHead &h, const ChainGeo &g, double (&acc)[4], unsigned &sca, ChainStart &st,
                                            unsigned long long (*ballots)[4])
{
  ChainRaw &ra = st.ra;
  ChainRaw rb;
  unsigned fa = st.fa, fb = 0u;
  sca = chain_leaf_finish(h.acc_tip != 0u, st.side, g.n, acc);
  // two steps per trip: the fetch buffers swap roles instead of being copied; the next step's loads are
  // in flight while the current one is computed and stored. The descriptor list of a chain ends with
  // a CS_END step whose sizes are all 0: the last trip's fetch.
  for (unsigned s = 0; s < h.nsteps; s += 2)
  {
    {
      const ChainStepLoad l1 = src.load(h.first + s + 1);
      fb = l1.flags;
      chain_issue<SM, C0, S1, C1>(l1, g, rb);
    }
    {
      const ChainStepOp o = src.op(h.first + s);
      chain_step<SM>(o, fa, g, ra, acc, sca, ballots, s);
    }
    if (s + 1 >= h.nsteps) break;
    {
      const ChainStepLoad l2 = src.load(h.first + s + 2);
      fa = l2.flags;
      chain_issue<SM, C0, S1, C1>(l2, g, ra);
    }
    {
      const ChainStepOp o = src.op(h.first + s + 1);
      chain_step<SM>(o, fb, g, rb, acc, sca, ballots, s + 1);
    }
  }
}

template <int SM, bool C0, bool S1, bool C1, class SRC>
__device__ __forceinline__ void chain_run(const SRC &src, const ChainHead &h, const ChainGeo &g, double (&acc)[4], unsigned &sca,
                                          unsigned long long (*ballots)[4])
{
  ChainStart st;
  chain_begin<SM, C0, S1, C1>(src, h, g, acc, st);
  chain_steps<SM, C0, S1, C1>(src, h, g, acc, sca, st, ballots);
}

__device__ __forceinline__ ChainGeo chain_geo(unsigned entries)
{
  const unsigned lane = threadIdx.x & 63u;
  const unsigned ntiles = (entries + 63u) / 64u;
  const unsigned n0 = blockIdx.x * 64u + lane;
  const bool valid = n0 < entries;
  ChainGeo g;
  g.rate = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  g.entries = entries;
  g.clv_bytes = ntiles * (kDnaTile * 8u);
  g.n = valid ? n0 : entries - 1;
  g.voff = ((g.n >> 6) * kDnaTile + g.rate * 256u + (g.n & 63u)) * 8u;
  g.n_store = valid ? g.n : 0x0fffffffu; // past every scaler buffer: dropped
  g.voff_store = valid ? g.voff : 0xf0000000u;
  return g;
}

template <int SM, bool C0, bool S1, bool C1, class SRC>
__device__ __forceinline__ void chain_body(const SRC src, unsigned entries)
{
  __shared__ unsigned long long ballots[4][4]; // [step parity x decision][rate]
  // (round 4 tried the XCD-aware workgroup order of the store-bound group launches here, kernels_common.h: random
  // 64-taxon trees 0.206 -> 0.217 ms per step, the ladder within the spread - the siblings' reads want the natural order)
  const ChainHead h = src.head(blockIdx.y);
  const ChainGeo g = chain_geo(entries);
  double acc[4];
  unsigned sca;
  chain_run<SM, C0, S1, C1>(src, h, g, acc, sca, ballots);
}

template <int SM, bool C0, bool S1, bool C1>
__global__ __launch_bounds__(256) void k_partials_dna_chain_pack(const ChainPack pack, unsigned entries)
{
  chain_body<SM, C0, S1, C1>(ChainSrcPack{pack}, entries);
}

template <int SM, bool C0, bool S1, bool C1>
__global__ __launch_bounds__(256) void k_partials_dna_chain(const ChainHead *heads, const ChainStepLoad *loads, const ChainStepOp *ops, unsigned entries)
{
  ChainSrcMem src;
  src.heads = (chead_p)(uintptr_t)heads;
  src.loads = (cstepload_p)(uintptr_t)loads;
  src.ops = (cstepop_p)(uintptr_t)ops;
  chain_body<SM, C0, S1, C1>(src, entries);
}

// ------------------------------------------------------------------------------------------------
// Chain tail: the chains that end in the two ends of the evaluated edge (the last stage of a
// traversal, held back for one call like the ops of k_edge_dna_tail below) run inside the
// log-likelihood kernel: a workgroup takes its tile through the parent end's chain, then the child
// end's, and forms the site log-likelihoods from the registers. An end that is not produced by a held
// chain is a chain of no steps (its `acc0` leaf is the end). The rate terms of a site live in four
// waves: they meet in LDS and the wave of rate 0 finishes the site exactly like k_edge_dna does; the
// tile sums are added up by the last workgroup in k_edge_dna's order (four tiles to a partial, then
// the partials), so the value is bit-identical to evaluating the edge from HBM.
template <int SM, bool C0, bool S1, bool C1, class SRC>
__device__ __forceinline__ void chain_edge_body(const DevEdge &e, const SRC src, const ChainHead &hp, const ChainHead &hc, unsigned entries)
{
  __shared__ unsigned long long ballots[4][4];
  __shared__ double xtr[4][64];
  __shared__ unsigned xrs[4][64];
  __shared__ unsigned last;
  __shared__ double ws[4];
  const ChainGeo g = chain_geo(entries);
  const unsigned lane = threadIdx.x & 63u;
  double vp[4], vc[4];
  unsigned scp, scc;
  {
    // both ends' first loads go out before either chain is walked: the child end's would otherwise queue up
    // behind the parent end's last stores
    ChainStart sp, sc;
    chain_begin<SM, C0, S1, C1>(src, hp, g, vp, sp);
    chain_begin<SM, C0, S1, C1>(src, hc, g, vc, sc);
    chain_steps<SM, C0, S1, C1>(src, hp, g, vp, scp, sp, ballots);
    chain_steps<SM, C0, S1, C1>(src, hc, g, vc, scc, sc, ballots);
  }
  {
    double tb[4];
    dna_matvec(tb, as_const(e.mat) + g.rate * 16u, vc);
    const unsigned fi = e.fidx[g.rate];
    cdouble_p pi = as_const(e.freqs) + (size_t)fi * 4;
    xtr[g.rate][lane] = fma(vp[3] * pi[3], tb[3], fma(vp[2] * pi[2], tb[2], fma(vp[1] * pi[1], tb[1], (vp[0] * pi[0]) * tb[0])));
    xrs[g.rate][lane] = scp + scc;
  }
  __syncthreads();
  if (g.rate == 0u)
  {
    const unsigned n0 = blockIdx.x * 64u + lane;
    const bool valid = n0 < e.sites;
    const unsigned n = valid ? n0 : e.sites - 1;
    unsigned rs[4] = {0, 0, 0, 0}, scal;
    if (e.per_rate)
    {
#pragma unroll
      for (int k = 0; k < 4; ++k) rs[k] = xrs[k][lane];
      scal = min(min(rs[0], rs[1]), min(rs[2], rs[3]));
    }
    else
      scal = xrs[0][lane];
    const int inv = e.invariant ? e.invariant[n] : -1;
    double terma = 0.0, terminv = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      double tr = xtr[k][lane];
      const unsigned fi = e.fidx[k];
      if (e.per_rate)
      {
        const unsigned ex = min(rs[k] - scal, PLLGPU_RATE_MAXDIFF);
        if (ex) tr *= minlh(ex);
      }
      const double pinv = e.prop_invar ? e.prop_invar[fi] : 0.0;
      const double w = e.rate_weights[k];
      if (pinv > 0.0)
      {
        terma += w * tr * (1.0 - pinv);
        if (inv >= 0) terminv += w * e.freqs[(size_t)fi * 4 + inv] * pinv;
      }
      else
        terma += tr * w;
    }
    double site = 0.0;
    if (valid)
    {
      site = finish_site(terma, terminv, scal, 0) * (double)e.pattern_weights[n];
      if (e.persite) e.persite[n] = site;
    }
    const double tile_sum = wave_sum(site);
    if (lane == 0)
    {
      partial_store(&e.block_sums[blockIdx.x], tile_sum); // kernels_common.h: no fences in the hand-off by default
      handoff_before_ticket(e.fenced);
      const unsigned ticket = __hip_atomic_fetch_add(e.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (ticket == gridDim.x - 1) ? 1u : 0u;
      if (last) handoff_after_last_ticket(e.fenced);
    }
  }
  __syncthreads();
  if (!last) return;
  // k_edge_dna's summation tree: workgroups of four tiles (wave sums added in wave order), then the
  // partials strided over 256 threads (publish_block_sum)
  const unsigned wave = threadIdx.x >> 6;
  const unsigned site_tiles = (e.sites + 63u) / 64u, nblocks = (site_tiles + 3u) / 4u;
  double a = 0.0;
  for (unsigned base = 0; base < nblocks; base += 4u * blockDim.x) // all requests of a batch first (sum_partials_strided)
  {
    double v[4][4];
#pragma unroll
    for (unsigned q = 0; q < 4u; ++q)
    {
      const unsigned i = base + q * blockDim.x + threadIdx.x;
#pragma unroll
      for (unsigned w = 0; w < 4u; ++w) v[q][w] = (i < nblocks && 4u * i + w < site_tiles) ? partial_load(&e.block_sums[4u * i + w]) : 0.0;
    }
#pragma unroll
    for (unsigned q = 0; q < 4u; ++q)
      if (base + q * blockDim.x + threadIdx.x < nblocks)
      {
        double sblk = v[q][0];
#pragma unroll
        for (unsigned w = 1; w < 4u; ++w) sblk += v[q][w];
        a += sblk;
      }
  }
  a = wave_sum(a);
  if (lane == 0) ws[wave] = a;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double t = ws[0];
    for (unsigned w = 1; w < 4u; ++w) t += ws[w];
    __hip_atomic_store(e.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(e.result, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    handoff_before_sequence(e.fenced); // the value is in host memory before the sequence word follows
    __hip_atomic_store(e.result + 1, e.sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <int SM, bool C0, bool S1, bool C1>
__global__ __launch_bounds__(256) void k_edge_dna_chain_pack(const DevEdge e, const ChainPack pack, unsigned entries)
{
  chain_edge_body<SM, C0, S1, C1>(e, ChainSrcPack{pack}, pack.heads[0], pack.heads[1], entries);
}

template <int SM, bool C0, bool S1, bool C1>
__global__ __launch_bounds__(256) void k_edge_dna_chain(const DevEdge e, const ChainHead hp, const ChainHead hc, const ChainStepLoad *loads,
                                                        const ChainStepOp *ops, unsigned entries)
{
  ChainSrcMem src;
  src.heads = nullptr;
  src.loads = (cstepload_p)(uintptr_t)loads;
  src.ops = (cstepop_p)(uintptr_t)ops;
  chain_edge_body<SM, C0, S1, C1>(e, src, hp, hc, entries);
}

// ------------------------------------------------------------------------------------------------
// Tail fusion: a traversal's last ops produce the two ends of the edge whose log-likelihood the
// caller asks for next (the universal call sequence: pll_update_partials, then
// pll_compute_edge_loglikelihood on the virtual root). The device layer holds those (at most two)
// ops back for one call; if the next call is the matching edge evaluation, k_edge_dna_tail forms
// the ends per site in registers (storing their CLVs and scalers like any update), and the site
// log-likelihood straight from the registers: one launch instead of two and neither end is read
// back. Any other call launches the held ops as ordinary updates first.
// KP / KC: DnaChildKind of the parent end (never a tip) and of the child end.
template <int KP, int KC>
__global__ __launch_bounds__(256) void k_edge_dna_tail(const DevEdge e, const FGroup g, int scale_mode, unsigned tiles_per_wave)
{
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ntiles = (e.sites + 63u) / 64u;
  cdouble_p pm = as_const(e.mat);
  double acc = 0.0;

  for (unsigned t = 0; t < tiles_per_wave; ++t)
  {
    const unsigned tile = (blockIdx.x * 4u + wave) * tiles_per_wave + t;
    if (tile >= ntiles) break; // wave-uniform
    const unsigned n0 = tile * 64u + lane;
    const bool valid = n0 < e.sites;
    const unsigned n = valid ? n0 : e.sites - 1;
    const size_t off = (size_t)(n >> 6) * kDnaTile + (n & 63u);

    double vp[4][4], vc[4][4];
    uint4 scp, scc;
    {
      // few waves per SIMD here (one tile each): registers are plentiful, latency is not - both
      // ends' loads go out before anything is computed
      DnaRaw rp, rc;
      dna_child_load<KP>(g.p, true, g.a, off, n, rp);
      dna_child_load<KC>(g.p, false, g.b, off, n, rc);
      dna_child_compute<KP>(g.p, true, g.a, n, scale_mode, rp, vp, scp);
      dna_child_store<KP>(g.a, off, n, valid, scale_mode, vp, scp);
      dna_child_compute<KC>(g.p, false, g.b, n, scale_mode, rc, vc, scc);
      dna_child_store<KC>(g.b, off, n, valid, scale_mode, vc, scc);
    }
    unsigned rs[4] = {0, 0, 0, 0}, scal;
    if (e.per_rate)
    {
      rs[0] = scp.x + scc.x;
      rs[1] = scp.y + scc.y;
      rs[2] = scp.z + scc.z;
      rs[3] = scp.w + scc.w;
      scal = min(min(rs[0], rs[1]), min(rs[2], rs[3]));
    }
    else
      scal = scp.x + scc.x;
    const int inv = e.invariant ? e.invariant[n] : -1;
    double terma = 0.0, terminv = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      double tb[4];
      dna_matvec(tb, pm + k * 16, vc[k]);
      const unsigned fi = e.fidx[k];
      cdouble_p pi = as_const(e.freqs) + (size_t)fi * 4;
      double tr = fma(vp[k][3] * pi[3], tb[3], fma(vp[k][2] * pi[2], tb[2], fma(vp[k][1] * pi[1], tb[1], (vp[k][0] * pi[0]) * tb[0])));
      if (e.per_rate)
      {
        const unsigned ex = min(rs[k] - scal, PLLGPU_RATE_MAXDIFF);
        if (ex) tr *= minlh(ex);
      }
      const double pinv = e.prop_invar ? e.prop_invar[fi] : 0.0;
      const double w = e.rate_weights[k];
      if (pinv > 0.0)
      {
        terma += w * tr * (1.0 - pinv);
        if (inv >= 0) terminv += w * e.freqs[(size_t)fi * 4 + inv] * pinv;
      }
      else
        terma += tr * w;
    }
    if (valid)
    {
      const double site = finish_site(terma, terminv, scal, 0) * (double)e.pattern_weights[n];
      if (e.persite) e.persite[n] = site;
      acc += site;
    }
  }
  publish_block_sum(e, wave_sum(acc), 4u);
}
