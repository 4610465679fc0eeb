// kernels_lean.h - CLV updates for 17..20 states (the protein models) on the fp64 matrix pipe, every layout:
// inner x inner, tip x inner, tip x tip; plain, gathering (site repeats) with tiled or entry-contiguous children and
// parents. One launch per level like the kernels it replaces (kernels_generic.h: k_partials_tiled<20, ...>).
//
// Why: the FMA kernels feed every multiply-add a coefficient through the scalar path - a wave has SGPRs for two
// matrix rows in flight, and what it waits for is those loads: inner x inner launches stay at 0.70-0.76 of HBM with
// four waves per SIMD, small launches (2-4 ops) and the gathering ones (three waves, rows missing the scalar cache)
// at half of that. Here a 4x4x4 MFMA takes its 16 coefficients from LDS in one read and uses them for 32 sites.
//
// Layout of the work (as k_partials_mfma_cc, kernels_mfma.h): workgroup = R waves (R <= 4), wave = rate category;
// all waves walk the SAME items of 32 entries, so the per-site scaling decision ("every rate below 2^-256") meets in
// LDS behind one barrier per item - no flag buffer, no epilogue launch. Lane l = (row = l >> 4, col = l & 15) owns
// states 4 g + row of entries 2 col, 2 col + 1 of the item. The matrices of all rates sit in LDS as the host stores
// them (PT[j][i], row stride 20, + zero rows up to 20 + one row of row sums): row j is tip column j and element
// (i, k) of A-block (ig, jg) is at (4 jg + k) * 20 + 4 ig + i.
// Children: tip codes (column of P from LDS, or MFMAs on 0/1 operands for ambiguous codes), tiled CLVs (16-byte
// loads where parent and child entries coincide), entry-contiguous CLVs of class-compressed nodes (site repeats).
// Those go through LDS both ways: ten consecutive lanes move the 160 bytes of one (entry, rate) block, 16 bytes
// each - five requests per child and item that the address unit takes as contiguous pieces - and the MFMA operands
// are read from / written to the staged rows. (Each lane fetching its own states, 8 bytes at 16 entries x 4 places
// per request, kept the address unit busy 56 % of the launch and the wave waiting for it: C3 with repeats, the
// launch over the compressed level-2 nodes 159 us for 230 MB of HBM traffic.)
// The next item's operands are requested before the current item's MFMAs.
// Arithmetic and its order are those of k_partials_mfma<5, ...>: bit-identical to it.
// src/core_partials.c:709-764 (ii), :465-507 (ti), :1166-1209 (tt), :819-879 (repeats), scaling :729-763.
#pragma once
#include "kernels_common.h"
#include "kernels_mfma.h"

template <int NG> struct LeanGeo
{
  static constexpr unsigned LD = 4 * NG;
  static constexpr unsigned rows = 4 * NG + 1;
  static constexpr unsigned mat = rows * LD;
  static constexpr unsigned gap_col = 4 * NG;
  static constexpr unsigned XST = 22; // doubles per staged entry-contiguous entry (rate block of SP <= 20 + 2: 16-byte aligned rows)
  // [R][2 children][mat] doubles, [2][R][32] flag bytes, [256] column indices of the tip codes; gathering launches:
  // then [R][2 children][32 entries][XST] doubles, the entry-contiguous blocks on their way in and out
  static size_t lds_bytes(unsigned R, bool gather)
  {
    return (size_t)R * 2 * mat * sizeof(double) + 2u * R * 32u + 256u + (gather ? (size_t)R * 2 * 32 * XST * sizeof(double) : 0);
  }
};

template <int NG, bool LTIP, bool RTIP, bool GATHER>
__global__ __launch_bounds__(256) void k_partials_lean(const OpPack pack, const GenGeo g, const unsigned long long *__restrict__ tipmap,
                                                       unsigned items_per_block, unsigned ncodes)
{
  typedef LeanGeo<NG> LG;
  constexpr unsigned LD = LG::LD;
  typedef double __attribute__((ext_vector_type(2))) double2v;
  extern __shared__ double lds[];
  const unsigned R = g.R, S = g.S;
  double *M = lds;                                                                    // [R][2][rows][LD]
  unsigned char *FL = reinterpret_cast<unsigned char *>(lds + (size_t)R * 2u * LG::mat); // [2][R][32]
  unsigned char *CIDX = FL + 2u * R * 32u;                                            // [256]
  constexpr unsigned XST = LG::XST;

  const DevOp &op = pack.ops[blockIdx.y];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave = rate category
  const unsigned row = lane >> 4, col = lane & 15u;
  const unsigned nitems = (op.entries + 31u) / 32u;
  const unsigned item_first = blockIdx.x * items_per_block;
  if (item_first >= nitems) return; // whole workgroup
  const unsigned item_end = min(item_first + items_per_block, nitems);
  const unsigned long long full = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
  // ---- stage: every wave its own rate's two matrices
  {
    double *ML = M + (size_t)(2u * k) * LG::mat, *MR = ML + LG::mat;
    const double *sl = op.lmat + (size_t)k * S * g.SPT, *sr = op.rmat + (size_t)k * S * g.SPT;
    {
      // all requests first: a load / wait / write loop made this seven L2 round trips per workgroup
      constexpr unsigned N = (LG::rows - 1u) * LD, PER = (N + 63u) / 64u;
      double vl[PER], vr[PER];
#pragma unroll
      for (unsigned q = 0; q < PER; ++q)
      {
        const unsigned idx = lane + 64u * q, j = idx / LD, i = idx % LD;
        const bool in = idx < N && j < S && i < S;
        const size_t off = in ? (size_t)j * g.SPT + i : 0;
        const double a = sl[off], b = sr[off];
        vl[q] = in ? a : 0.0;
        vr[q] = in ? b : 0.0;
      }
#pragma unroll
      for (unsigned q = 0; q < PER; ++q)
      {
        const unsigned idx = lane + 64u * q;
        if (idx < N)
        {
          ML[idx] = vl[q];
          MR[idx] = vr[q];
        }
      }
    }
    if (LTIP || RTIP)
      for (unsigned c0 = threadIdx.x; c0 < 256u; c0 += blockDim.x)
      {
        unsigned ci = kCcAmbiguous;
        if (c0 < ncodes)
        {
          const unsigned long long mk = tipmap[c0];
          ci = mk == full ? LG::gap_col : __popcll(mk) == 1 ? (unsigned)__ffsll((long long)mk) - 1u : kCcAmbiguous;
        }
        CIDX[c0] = (unsigned char)ci;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (LTIP || RTIP)
    {
      // row sums in ascending j like the reference's set-bit walk (core_partials.c:480-489)
      for (unsigned i = lane; i < 2u * LD; i += 64u)
      {
        double *Mx = i < LD ? ML : MR;
        const unsigned ii = i % LD;
        double s = 0.0;
        for (unsigned j = 0; j < S; ++j) s += Mx[j * LD + ii];
        Mx[LG::gap_col * LD + ii] = s;
      }
    }
  }
  __syncthreads();

  const int mode = op.pscaler ? g.scale_mode : 0;
  const unsigned lay = GATHER ? op.layout : 0u;
  const bool paos = GATHER && (lay & kAosParent); // (inner children of a gathering launch are entry-contiguous: the host sees to it)
  const unsigned espan = R * g.SP;
  const double *ML = M + (size_t)(2u * k) * LG::mat, *MR = ML + LG::mat;
  const unsigned afrag = row * LD + (lane & 3u); // + 4 jg * LD + 4 ig: element (i = lane & 3, k = row) of block (ig, jg)
  const unsigned lane_off = row * 64u + 2u * col;
  const unsigned last = op.entries - 1u;

  // gathering launches: this wave's staging rows, and the lane's five 16-byte pieces of them (piece c = 64 r + lane:
  // entry c / 10 of the item, bytes 16 (c % 10) of its rate block)
  double *XW = reinterpret_cast<double *>(CIDX + 256u) + (size_t)k * 2u * 32u * XST;
  unsigned pc_ent[5], pc_off[5];
#pragma unroll
  for (int r = 0; r < 5; ++r)
  {
    const unsigned c = 64u * r + lane;
    pc_ent[r] = c / 10u;
    pc_off[r] = 2u * (c % 10u);
  }

  struct Item
  {
    unsigned le[2], re[2];
  };
  struct Operands
  {
    double xl[NG][2], xr[NG][2]; // inner children (plain launches): the lane's states of its two entries
    double2v pl[5], pr[5];       // inner children (gathering launches): the lane's pieces of the staged blocks
    unsigned cl[2], cr[2];       // tip children: codes
  };
  auto entries_of = [&](unsigned item, Item &t) {
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
    {
      const unsigned nn = min(item * 32u + 2u * col + sg, last);
      t.le[sg] = t.re[sg] = nn;
      if (GATHER) gather_entries(op, nn, t.le[sg], t.re[sg]);
    }
  };
  // request one child's operands
  auto request = [&](unsigned item, const unsigned (&ce)[2], const double *__restrict__ clv, const unsigned char *__restrict__ tip, bool tipc,
                     double (&x)[NG][2], double2v (&pc)[5], unsigned (&code)[2]) {
    if (tipc)
    {
      code[0] = tip[ce[0]];
      code[1] = tip[ce[1]];
      return;
    }
    if (!GATHER)
    {
      // parent and child entries coincide: 16 bytes per lane, wave-uniform base
      const double *ub = clv + (size_t)(item >> 1) * g.tile_sz + (size_t)k * S * 64 + (item & 1u) * 32u;
#pragma unroll
      for (int jg = 0; jg < NG; ++jg)
      {
        const unsigned j = min(4u * jg + row, S - 1u); // rows beyond S meet zero matrix rows; stay in bounds
        const double2v w = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(ub + (size_t)j * 64u + 2u * col));
        x[jg][0] = w.x;
        x[jg][1] = w.y;
      }
      return;
    }
    // entry-contiguous child: the lane's pieces; entry e of the item belongs to lane col = e >> 1 (any row), slot e & 1
    const double *kb = clv + (size_t)k * g.SP;
#pragma unroll
    for (int r = 0; r < 5; ++r)
    {
      const unsigned s0 = __shfl(ce[0], pc_ent[r] >> 1, 64), s1 = __shfl(ce[1], pc_ent[r] >> 1, 64);
      const unsigned src = (pc_ent[r] & 1u) ? s1 : s0;
      pc[r] = *reinterpret_cast<const double2v *>(kb + (size_t)src * espan + pc_off[r]);
    }
  };
  // gathering launches: the pieces into the wave's staging rows, the lane's MFMA operands out of them
  auto land = [&](unsigned child, const double2v (&pc)[5], double (&x)[NG][2]) {
    double *Xc = XW + (size_t)child * 32u * XST;
#pragma unroll
    for (int r = 0; r < 5; ++r) *reinterpret_cast<double2v *>(Xc + pc_ent[r] * XST + pc_off[r]) = pc[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int jg = 0; jg < NG; ++jg)
    {
      const unsigned j = min(4u * jg + row, S - 1u);
      x[jg][0] = Xc[(2u * col) * XST + j];
      x[jg][1] = Xc[(2u * col + 1u) * XST + j];
    }
  };
  // D = P x for the lane's states of both entries
  auto side = [&](const double *Mx, bool tipc, const double (&x)[NG][2], const unsigned (&code)[2], double (&d)[NG][2]) {
    if (tipc)
    {
      const unsigned c0 = CIDX[code[0]], c1 = CIDX[code[1]];
      if (__all(c0 != kCcAmbiguous && c1 != kCcAmbiguous))
      {
        const double *p0 = Mx + c0 * LD + row, *p1 = Mx + c1 * LD + row;
#pragma unroll
        for (int ig = 0; ig < NG; ++ig)
        {
          d[ig][0] = p0[4 * ig];
          d[ig][1] = p1[4 * ig];
        }
        return;
      }
      const unsigned long long m0 = tipmap[code[0]], m1 = tipmap[code[1]];
#pragma unroll
      for (int ig = 0; ig < NG; ++ig) d[ig][0] = d[ig][1] = 0.0;
#pragma unroll
      for (int jg = 0; jg < NG; ++jg)
      {
        const double x0 = mfma_x<true>(nullptr, m0, S, 4 * jg + row), x1 = mfma_x<true>(nullptr, m1, S, 4 * jg + row);
#pragma unroll
        for (int ig = 0; ig < NG; ++ig)
        {
          const double a = Mx[afrag + 4 * jg * LD + 4 * ig];
          d[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x0, d[ig][0], 0, 0, 0);
          d[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x1, d[ig][1], 0, 0, 0);
        }
      }
      return;
    }
#pragma unroll
    for (int ig = 0; ig < NG; ++ig) d[ig][0] = d[ig][1] = 0.0;
#pragma unroll
    for (int jg = 0; jg < NG; ++jg)
    {
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
      {
        const double a = Mx[afrag + 4 * jg * LD + 4 * ig];
        d[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][0], d[ig][0], 0, 0, 0);
        d[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][1], d[ig][1], 0, 0, 0);
      }
    }
  };

  Item cur, nxt;
  Operands oc;
  entries_of(item_first, cur);
  request(item_first, cur.le, op.left, op.ltip, LTIP, oc.xl, oc.pl, oc.cl);
  request(item_first, cur.re, op.right, op.rtip, RTIP, oc.xr, oc.pr, oc.cr);
  nxt = cur;
  if (item_first + 1u < item_end) entries_of(item_first + 1u, nxt);
  unsigned buf = 0;

  for (unsigned item = item_first; item < item_end; ++item)
  {
    const bool has_next = item + 1u < item_end;
    // the next item's operands first, the class-map entries of the one after it behind them
    Operands on;
    Item nn2 = nxt;
    if (has_next)
    {
      request(item + 1u, nxt.le, op.left, op.ltip, LTIP, on.xl, on.pl, on.cl);
      request(item + 1u, nxt.re, op.right, op.rtip, RTIP, on.xr, on.pr, on.cr);
      if (GATHER && item + 2u < item_end) entries_of(item + 2u, nn2);
    }
    const unsigned e0 = item * 32u + 2u * col;
    const bool valid[2] = {e0 < op.entries, e0 + 1u < op.entries};
    // the children's scaler entries now: they need not stay in registers across the contractions in any other form
    unsigned below[2] = {0u, 0u};
    if (mode == 1)
    {
      if (k == 0 && row == 0)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) below[sg] = (op.lscaler ? op.lscaler[cur.le[sg]] : 0u) + (op.rscaler ? op.rscaler[cur.re[sg]] : 0u);
    }
    else if (mode == 2)
    {
      if (row == 0)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
          below[sg] = (op.lscaler ? op.lscaler[(size_t)cur.le[sg] * R + k] : 0u) + (op.rscaler ? op.rscaler[(size_t)cur.re[sg] * R + k] : 0u);
    }

    double DL[NG][2], DR[NG][2];
    double gxl[NG][2], gxr[NG][2]; // gathering launches: operands live from the staging rows to the MFMAs only
    if (GATHER)
    {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier(); // the previous item's rows have been read / written out by every lane
      if (!LTIP) land(0u, oc.pl, gxl);
      if (!RTIP) land(1u, oc.pr, gxr);
    }
    side(ML, LTIP, *(GATHER ? &gxl : &oc.xl), oc.cl, DL);
    side(MR, RTIP, *(GATHER ? &gxr : &oc.xr), oc.cr, DR);
    bool small[2] = {true, true};
#pragma unroll
    for (int ig = 0; ig < NG; ++ig)
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
      {
        DL[ig][sg] *= DR[ig][sg];
        if (4u * ig + row < S) small[sg] = small[sg] && (DL[ig][sg] < PLLGPU_SCALE_THRESHOLD);
      }
    if (mode)
    {
      bool scale[2];
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
      {
        int sm = small[sg] ? 1 : 0; // an entry's states are spread over the four row groups of the wave
        sm &= __shfl_xor(sm, 16, 64);
        sm &= __shfl_xor(sm, 32, 64);
        scale[sg] = sm != 0;
      }
      if (mode == 1)
      {
        // every rate's answer, through LDS; two flag sets alternate, so one barrier per item
        unsigned char *fl = FL + (size_t)buf * R * 32u;
        if (row == 0)
        {
          fl[k * 32u + 2u * col] = scale[0] ? 1 : 0;
          fl[k * 32u + 2u * col + 1u] = scale[1] ? 1 : 0;
        }
        lds_barrier(); // (not __syncthreads(): that would wait for the loads and stores in flight as well)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
        {
          bool all = true;
          for (unsigned kk = 0; kk < R; ++kk) all = all && fl[kk * 32u + 2u * col + sg];
          scale[sg] = all;
        }
        buf ^= 1u;
        if (k == 0 && row == 0)
#pragma unroll
          for (int sg = 0; sg < 2; ++sg)
            if (valid[sg]) op.pscaler[e0 + sg] = below[sg] + (scale[sg] ? 1u : 0u);
      }
      else if (row == 0)
      {
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
          if (valid[sg]) op.pscaler[(size_t)(e0 + sg) * R + k] = below[sg] + (scale[sg] ? 1u : 0u);
      }
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
        if (scale[sg])
        {
#pragma unroll
          for (int ig = 0; ig < NG; ++ig) DL[ig][sg] *= PLLGPU_SCALE_FACTOR;
        }
    }
    // ---- store
    if (paos)
    {
      // the lane's states into the wave's rows (the operands were read from them before the MFMAs), whole blocks out
      double *Xp = XW;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
      {
        const unsigned i = 4u * ig + row; // < 4 NG = 20 <= XST; the padding of the host layout (i >= S) stays zero
        Xp[(2u * col) * XST + i] = i < S ? DL[ig][0] : 0.0;
        Xp[(2u * col + 1u) * XST + i] = i < S ? DL[ig][1] : 0.0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      double *kb = op.parent + (size_t)k * g.SP;
#pragma unroll
      for (int r = 0; r < 5; ++r)
      {
        const unsigned n = item * 32u + pc_ent[r];
        if (n < op.entries && pc_off[r] < g.SP)
          *reinterpret_cast<double2v *>(kb + (size_t)n * espan + pc_off[r]) = *reinterpret_cast<const double2v *>(Xp + pc_ent[r] * XST + pc_off[r]);
      }
    }
    else
    {
      double *ub = op.parent + (size_t)(item >> 1) * g.tile_sz + (size_t)k * S * 64 + (item & 1u) * 32u; // wave-uniform
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
        if (4u * ig + row < S)
        {
          double *q = ub + (lane_off + 256u * ig);
          if (valid[1])
          {
            double2v w;
            w.x = DL[ig][0];
            w.y = DL[ig][1];
            if (LTIP && RTIP)
              __builtin_nontemporal_store(w, reinterpret_cast<double2v *>(q)); // a tip x tip launch is pure store traffic
            else
              *reinterpret_cast<double2v *>(q) = w;
          }
          else if (valid[0])
            q[0] = DL[ig][0];
        }
    }
    cur = nxt;
    nxt = nn2;
    if (has_next) oc = on;
  }
}

// ------------------------------------------------------------------------------------------------
// (inner x inner, inner x inner -> inner x inner) groups of the same shapes: an op P whose two children A, B are
// produced by the same call is evaluated with them - four CLVs read, three written per site instead of six and three.
// What kept such groups from the 17..32-state kernels so far: P is formed from the RESCALED A and B, and whether an
// entry of A is rescaled depends on every rate category - here all rates of an item sit in one workgroup and meet in
// LDS (one barrier per op). D and B operands share their lane map, so A and B go from the accumulators straight into
// P's MFMAs. LDS: six matrices x R rates as stored (20 x 20), 77 KB for four rates: two workgroups per CU.
template <int NG>
__global__ __launch_bounds__(256, 2) void k_partials_lean3(const FusePack pack, const GenGeo g, unsigned entries, unsigned items_per_block)
{
  constexpr unsigned LD = 4 * NG, MAT = LD * LD;
  typedef double __attribute__((ext_vector_type(2))) double2v;
  extern __shared__ double lds[];
  const unsigned R = g.R, S = g.S;
  double *M = lds;                                                                  // [R][6][LD][LD]: a.l a.r b.l b.r p.l p.r
  unsigned char *FL = reinterpret_cast<unsigned char *>(lds + (size_t)R * 6u * MAT); // [3 ops][2][R][32]

  const FGroup &grp = pack.g[blockIdx.y];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave = rate category
  const unsigned row = lane >> 4, col = lane & 15u;
  const unsigned nitems = (entries + 31u) / 32u;
  const unsigned item_first = blockIdx.x * items_per_block;
  if (item_first >= nitems) return; // whole workgroup
  const unsigned item_end = min(item_first + items_per_block, nitems);
  {
    const double *src[6] = {grp.a.lmat, grp.a.rmat, grp.b.lmat, grp.b.rmat, grp.p.lmat, grp.p.rmat};
    constexpr unsigned PER = (MAT + 63u) / 64u;
#pragma unroll
    for (int h = 0; h < 2; ++h) // three matrices at a time: all their requests first
    {
      double v[3][PER];
#pragma unroll
      for (unsigned q = 0; q < PER; ++q)
      {
        const unsigned idx = lane + 64u * q, j = idx / LD, i = idx % LD;
        const bool in = idx < MAT && j < S && i < S;
        const size_t off = in ? ((size_t)k * S + j) * g.SPT + i : 0;
#pragma unroll
        for (int m = 0; m < 3; ++m)
        {
          const double x = src[3 * h + m][off];
          v[m][q] = in ? x : 0.0;
        }
      }
#pragma unroll
      for (unsigned q = 0; q < PER; ++q)
      {
        const unsigned idx = lane + 64u * q;
        if (idx < MAT)
#pragma unroll
          for (int m = 0; m < 3; ++m) M[((size_t)k * 6u + 3 * h + m) * MAT + idx] = v[m][q];
      }
    }
  }
  __syncthreads();
  const double *Mk = M + (size_t)k * 6u * MAT;
  const unsigned afrag = row * LD + (lane & 3u);
  const unsigned lane_off = row * 64u + 2u * col;
  const FOp *fo[3] = {&grp.a, &grp.b, &grp.p};
  int mode[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) mode[o] = fo[o]->pscaler ? g.scale_mode : 0;

  struct Kids
  {
    double x[4][NG][2]; // a.left a.right b.left b.right: the lane's states of its two entries
    unsigned below[2][2]; // [a, b][entry]: the scaler entries of the producers' children, summed
  };
  // Requested one item ahead, scaler entries first: waits for vector memory are in issue order, so a scaler word
  // fetched after the next item's 20 CLV requests would make its user wait for all of them - no prefetch at all.
  auto request = [&](unsigned item, Kids &kd) {
    {
      const unsigned f0 = min(item * 32u + 2u * col, entries - 1u), f1 = min(item * 32u + 2u * col + 1u, entries - 1u);
      const unsigned fe[2] = {f0, f1};
#pragma unroll
      for (int o = 0; o < 2; ++o)
      {
        const FOp &op = o ? grp.b : grp.a;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
        {
          unsigned b = 0u;
          if (mode[o] == 1) b = (op.lscaler ? op.lscaler[fe[sg]] : 0u) + (op.rscaler ? op.rscaler[fe[sg]] : 0u);
          if (mode[o] == 2) b = (op.lscaler ? op.lscaler[(size_t)fe[sg] * R + k] : 0u) + (op.rscaler ? op.rscaler[(size_t)fe[sg] * R + k] : 0u);
          kd.below[o][sg] = b;
        }
      }
    }
    const double *cl[4] = {grp.a.left, grp.a.right, grp.b.left, grp.b.right};
    const size_t uo = (size_t)(item >> 1) * g.tile_sz + (size_t)k * S * 64 + (item & 1u) * 32u + 2u * col;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int jg = 0; jg < NG; ++jg)
      {
        const unsigned j = min(4u * jg + row, S - 1u);
        const double2v w = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(cl[c] + uo + (size_t)j * 64u));
        kd.x[c][jg][0] = w.x;
        kd.x[c][jg][1] = w.y;
      }
  };
  auto contract = [&](const double *Mx, const double (&x)[NG][2], double (&d)[NG][2]) {
#pragma unroll
    for (int ig = 0; ig < NG; ++ig) d[ig][0] = d[ig][1] = 0.0;
#pragma unroll
    for (int jg = 0; jg < NG; ++jg)
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
      {
        const double a = Mx[afrag + 4 * jg * LD + 4 * ig];
        d[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][0], d[ig][0], 0, 0, 0);
        d[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][1], d[ig][1], 0, 0, 0);
      }
  };

  Kids cur;
  request(item_first, cur);
  unsigned buf = 0;
  for (unsigned item = item_first; item < item_end; ++item)
  {
    const bool has_next = item + 1u < item_end;
    Kids nxt;
    if (has_next) request(item + 1u, nxt);
    const unsigned e0 = item * 32u + 2u * col;
    const bool valid[2] = {e0 < entries, e0 + 1u < entries};
    unsigned count[2][2] = {{0u, 0u}, {0u, 0u}}; // [a, b][entry]: the producers' scaler entries, what P's build on

    // one op: v = D_left o D_right, the scaling decision (all rates, through LDS), the scaler entry, the store
    auto finish = [&](int o, double (&v)[NG][2], const double (&dr)[NG][2], const unsigned (&below)[2], unsigned (&out)[2], bool stream) {
      bool small[2] = {true, true};
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
        {
          v[ig][sg] *= dr[ig][sg];
          if (4u * ig + row < S) small[sg] = small[sg] && (v[ig][sg] < PLLGPU_SCALE_THRESHOLD);
        }
      const FOp &op = *fo[o];
      out[0] = out[1] = 0u;
      if (mode[o])
      {
        bool scale[2];
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
        {
          int sm = small[sg] ? 1 : 0; // an entry's states are spread over the four row groups of the wave
          sm &= __shfl_xor(sm, 16, 64);
          sm &= __shfl_xor(sm, 32, 64);
          scale[sg] = sm != 0;
        }
        if (mode[o] == 1)
        {
          unsigned char *fl = FL + ((size_t)o * 2u + buf) * R * 32u;
          if (row == 0)
          {
            fl[k * 32u + 2u * col] = scale[0] ? 1 : 0;
            fl[k * 32u + 2u * col + 1u] = scale[1] ? 1 : 0;
          }
          lds_barrier(); // (not __syncthreads(): that would wait for the loads and stores in flight as well)
#pragma unroll
          for (int sg = 0; sg < 2; ++sg)
          {
            bool all = true;
            for (unsigned kk = 0; kk < R; ++kk) all = all && fl[kk * 32u + 2u * col + sg];
            scale[sg] = all;
            out[sg] = below[sg] + (all ? 1u : 0u);
          }
          if (k == 0 && row == 0)
#pragma unroll
            for (int sg = 0; sg < 2; ++sg)
              if (valid[sg]) op.pscaler[e0 + sg] = out[sg];
        }
        else
        {
#pragma unroll
          for (int sg = 0; sg < 2; ++sg) out[sg] = below[sg] + (scale[sg] ? 1u : 0u);
          if (row == 0)
#pragma unroll
            for (int sg = 0; sg < 2; ++sg)
              if (valid[sg]) op.pscaler[(size_t)(e0 + sg) * R + k] = out[sg];
        }
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
          if (scale[sg])
          {
#pragma unroll
            for (int ig = 0; ig < NG; ++ig) v[ig][sg] *= PLLGPU_SCALE_FACTOR;
          }
      }
      double *ub = op.parent + (size_t)(item >> 1) * g.tile_sz + (size_t)k * S * 64 + (item & 1u) * 32u; // wave-uniform
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
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
    };
    const unsigned ba[2] = {cur.below[0][0], cur.below[0][1]}, bb[2] = {cur.below[1][0], cur.below[1][1]};

    double va[NG][2], vb[NG][2], dr[NG][2];
    contract(Mk + 0 * MAT, cur.x[0], va);
    contract(Mk + 1 * MAT, cur.x[1], dr);
    finish(0, va, dr, ba, count[0], true);
    contract(Mk + 2 * MAT, cur.x[2], vb);
    contract(Mk + 3 * MAT, cur.x[3], dr);
    finish(1, vb, dr, bb, count[1], true);
    // P from the registers; its children's scaler entries are the producers' (a producer without a scaler buffer: 0)
    unsigned bp[2] = {(mode[0] ? count[0][0] : 0u) + (mode[1] ? count[1][0] : 0u), (mode[0] ? count[0][1] : 0u) + (mode[1] ? count[1][1] : 0u)};
    double vp[NG][2];
    contract(Mk + 4 * MAT, va, vp);
    contract(Mk + 5 * MAT, vb, dr);
    unsigned cp[2];
    finish(2, vp, dr, bp, cp, false);
    buf ^= 1u;
    if (has_next) cur = nxt;
  }
}

// ------------------------------------------------------------------------------------------------
// The same groups with ONE rate category per workgroup (as the cherry groups, k_partials_mfma_cc): 19 KB of LDS, no
// barrier, every wave on items of its own - three workgroups per CU by registers, their phases independent. The
// per-site scaling decision (every rate below 2^-256?) is not available to such a workgroup, so in that mode the
// kernel works SPECULATIVELY: A, B and P are formed and stored unscaled, each (op, rate, entry) leaves its "all below"
// bit in a byte buffer, and k_iii_epilogue finishes the entries: scaler words; an op whose every rate says "below" is
// rescaled in place; and when a PRODUCER was rescaled - P was then formed from the wrong A or B - that entry of P is
// recomputed from the rescaled producers with plain multiply-adds (rare beyond counting on real data; exact up to the
// summation order of that entry). Per-rate scalers need no speculation: the decision is the workgroup's own.
// grid = (item blocks, groups, rate categories); flag buffer: [group][a, b, p][rate][entry].
template <int NG>
__global__ __launch_bounds__(256, 3) void k_partials_mfma_iii(const FusePack pack, const GenGeo g, unsigned entries, unsigned items_per_wave,
                                                              unsigned char *__restrict__ flagbuf, unsigned flag_stride)
{
  constexpr unsigned LD = 4 * NG, MAT = LD * LD;
  typedef double __attribute__((ext_vector_type(2))) double2v;
  extern __shared__ double lds[];
  double *M = lds; // [6][LD][LD] as stored (PT[j][i]), zero beyond S: a.l a.r b.l b.r p.l p.r
  const unsigned R = g.R, S = g.S, k = blockIdx.z;
  const FGroup &grp = pack.g[blockIdx.y];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned row = lane >> 4, col = lane & 15u;
  const unsigned nitems = (entries + 31u) / 32u;
  if (blockIdx.x * 4u * items_per_wave >= nitems) return; // whole workgroup
  {
    const double *src[6] = {grp.a.lmat, grp.a.rmat, grp.b.lmat, grp.b.rmat, grp.p.lmat, grp.p.rmat};
    constexpr unsigned PER = (MAT + 255u) / 256u;
    double v[6][PER];
#pragma unroll
    for (unsigned q = 0; q < PER; ++q)
    {
      const unsigned idx = threadIdx.x + 256u * q, j = idx / LD, i = idx % LD;
      const bool in = idx < MAT && j < S && i < S;
      const size_t off = in ? ((size_t)k * S + j) * g.SPT + i : 0;
#pragma unroll
      for (int m = 0; m < 6; ++m)
      {
        const double x = src[m][off];
        v[m][q] = in ? x : 0.0;
      }
    }
#pragma unroll
    for (unsigned q = 0; q < PER; ++q)
    {
      const unsigned idx = threadIdx.x + 256u * q;
      if (idx < MAT)
#pragma unroll
        for (int m = 0; m < 6; ++m) M[(size_t)m * MAT + idx] = v[m][q];
    }
  }
  __syncthreads();
  const unsigned item0 = (blockIdx.x * 4u + wave) * items_per_wave;
  if (item0 >= nitems) return; // no barriers below
  const unsigned item_end = min(item0 + items_per_wave, nitems);
  const unsigned afrag = row * LD + (lane & 3u);
  const unsigned lane_off = row * 64u + 2u * col;
  const FOp *fo[3] = {&grp.a, &grp.b, &grp.p};
  int mode[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) mode[o] = fo[o]->pscaler ? g.scale_mode : 0;
  unsigned char *flags = flagbuf + (size_t)blockIdx.y * 3u * R * flag_stride + (size_t)k * flag_stride; // + o * R * flag_stride + entry

  struct Kids
  {
    double x[4][NG][2];
  };
  // wave-uniform base + a 32-bit lane offset per state group: the address form that needs no 64-bit registers per request
  unsigned loff[NG];
#pragma unroll
  for (int jg = 0; jg < NG; ++jg) loff[jg] = min(4u * jg + row, S - 1u) * 64u + 2u * col;
  auto request = [&](unsigned item, Kids &kd) {
    const double *cl[4] = {grp.a.left, grp.a.right, grp.b.left, grp.b.right};
    const size_t uo = (size_t)(item >> 1) * g.tile_sz + (size_t)k * S * 64 + (item & 1u) * 32u; // wave-uniform
#pragma unroll
    for (int c = 0; c < 4; ++c)
    {
      const double *ub = cl[c] + uo;
#pragma unroll
      for (int jg = 0; jg < NG; ++jg)
      {
        const double2v w = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(ub + loff[jg]));
        kd.x[c][jg][0] = w.x;
        kd.x[c][jg][1] = w.y;
      }
    }
  };
  auto contract = [&](const double *Mx, const double (&x)[NG][2], double (&d)[NG][2]) {
#pragma unroll
    for (int ig = 0; ig < NG; ++ig) d[ig][0] = d[ig][1] = 0.0;
#pragma unroll
    for (int jg = 0; jg < NG; ++jg)
    {
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
      {
        const double a = Mx[afrag + 4 * jg * LD + 4 * ig];
        d[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][0], d[ig][0], 0, 0, 0);
        d[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x[jg][1], d[ig][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0); // keep the fragment look-ahead bounded (the six contractions' 150 LDS reads otherwise all move up front)
    }
  };

  Kids cur;
  request(item0, cur);
  for (unsigned item = item0; item < item_end; ++item)
  {
    const unsigned e0 = item * 32u + 2u * col;
    const bool valid[2] = {e0 < entries, e0 + 1u < entries};
    const unsigned ec[2] = {min(e0, entries - 1u), min(e0 + 1u, entries - 1u)};
    // v = D_left o D_right of op o; its decision: per rate - applied here; per site - left to the epilogue; store
    auto finish = [&](int o, double (&v)[NG][2], const double (&dr)[NG][2], unsigned (&count)[2], const unsigned (&below)[2], bool stream) {
      bool small[2] = {true, true};
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
        {
          v[ig][sg] *= dr[ig][sg];
          if (4u * ig + row < S) small[sg] = small[sg] && (v[ig][sg] < PLLGPU_SCALE_THRESHOLD);
        }
      const FOp &op = *fo[o];
      count[0] = count[1] = 0u;
      if (mode[o])
      {
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
        {
          int sm = small[sg] ? 1 : 0; // an entry's states are spread over the four row groups of the wave
          sm &= __shfl_xor(sm, 16, 64);
          sm &= __shfl_xor(sm, 32, 64);
          if (mode[o] == 1)
          {
            if (row == 0 && valid[sg]) flags[(size_t)o * R * flag_stride + e0 + sg] = (unsigned char)sm;
          }
          else
          {
            count[sg] = below[sg] + (sm ? 1u : 0u);
            if (row == 0 && valid[sg]) op.pscaler[(size_t)(e0 + sg) * R + k] = count[sg];
            if (sm)
            {
#pragma unroll
              for (int ig = 0; ig < NG; ++ig) v[ig][sg] *= PLLGPU_SCALE_FACTOR;
            }
          }
        }
      }
      double *ub = op.parent + (size_t)(item >> 1) * g.tile_sz + (size_t)k * S * 64 + (item & 1u) * 32u; // wave-uniform
#pragma unroll
      for (int ig = 0; ig < NG; ++ig)
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
    };
    auto below_rate = [&](const FOp &op, int m, unsigned (&b)[2]) {
      b[0] = b[1] = 0u;
      if (m == 2)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg)
          b[sg] = (op.lscaler ? op.lscaler[(size_t)ec[sg] * R + k] : 0u) + (op.rscaler ? op.rscaler[(size_t)ec[sg] * R + k] : 0u);
    };
    unsigned ba[2], bb[2], ca[2], cb[2], cp[2];
    below_rate(grp.a, mode[0], ba);
    below_rate(grp.b, mode[1], bb);
    double va[NG][2], vb[NG][2], dr[NG][2];
    contract(M + 0 * MAT, cur.x[0], va);
    contract(M + 1 * MAT, cur.x[1], dr);
    finish(0, va, dr, ca, ba, true);
    contract(M + 2 * MAT, cur.x[2], vb);
    contract(M + 3 * MAT, cur.x[3], dr);
    // the operands are dead: the next item's requests go out in front of P's MFMAs and the stores
    if (item + 1u < item_end) request(item + 1u, cur);
    finish(1, vb, dr, cb, bb, true);
    const unsigned bp[2] = {ca[0] + cb[0], ca[1] + cb[1]};
    double vp[NG][2];
    contract(M + 4 * MAT, va, vp);
    contract(M + 5 * MAT, vb, dr);
    finish(2, vp, dr, cp, bp, false);
  }
}

// per-site mode: finishes what k_partials_mfma_iii left open - one thread per entry of a group
__global__ __launch_bounds__(256) void k_iii_epilogue(const FusePack pack, const GenGeo g, unsigned entries, const unsigned char *__restrict__ flagbuf,
                                                      unsigned flag_stride)
{
  if (g.scale_mode != 1) return;
  const FGroup &grp = pack.g[blockIdx.y];
  const unsigned n = blockIdx.x * 256u + threadIdx.x;
  if (n >= entries) return;
  const unsigned S = g.S, R = g.R;
  const unsigned char *fl = flagbuf + (size_t)blockIdx.y * 3u * R * flag_stride + n;
  const FOp *fo[3] = {&grp.a, &grp.b, &grp.p};
  bool f[3];
#pragma unroll
  for (int o = 0; o < 3; ++o)
  {
    f[o] = fo[o]->pscaler != nullptr;
    if (f[o])
      for (unsigned kk = 0; kk < R; ++kk) f[o] = f[o] && fl[((size_t)o * R + kk) * flag_stride];
  }
  auto base = [&](const FOp &op) { return op.parent + (size_t)(n >> 6) * g.tile_sz + (n & 63u); };
  auto rescale = [&](const FOp &op) {
    double *b = base(op);
    for (unsigned q = 0; q < R * S; ++q) b[(size_t)q * 64] *= PLLGPU_SCALE_FACTOR;
  };
  unsigned cnt[2] = {0u, 0u};
#pragma unroll
  for (int o = 0; o < 2; ++o)
  {
    const FOp &op = *fo[o];
    if (!op.pscaler) continue;
    if (f[o]) rescale(op);
    cnt[o] = (op.lscaler ? op.lscaler[n] : 0u) + (op.rscaler ? op.rscaler[n] : 0u) + (f[o] ? 1u : 0u);
    op.pscaler[n] = cnt[o];
  }
  if (f[0] || f[1])
  {
    // P was formed from a producer that has been rescaled since: this entry again, from what is in memory now
    const double *A = base(grp.a), *B = base(grp.b);
    double *P = base(grp.p);
    bool all = true;
    for (unsigned kk = 0; kk < R; ++kk)
      for (unsigned i = 0; i < S; ++i)
      {
        double l = 0.0, r = 0.0;
        for (unsigned j = 0; j < S; ++j)
        {
          l = fma(grp.p.lmat[((size_t)kk * S + j) * g.SPT + i], A[((size_t)kk * S + j) * 64], l);
          r = fma(grp.p.rmat[((size_t)kk * S + j) * g.SPT + i], B[((size_t)kk * S + j) * 64], r);
        }
        const double v = l * r;
        all = all && (v < PLLGPU_SCALE_THRESHOLD);
        P[((size_t)kk * S + i) * 64] = v;
      }
    f[2] = grp.p.pscaler != nullptr && all;
  }
  if (grp.p.pscaler)
  {
    if (f[2]) rescale(grp.p);
    grp.p.pscaler[n] = cnt[0] + cnt[1] + (f[2] ? 1u : 0u);
  }
}

// ------------------------------------------------------------------------------------------------
// Edge / root log-likelihood for 17..20 states on the matrix pipe (src/core_likelihood.c:1388-1490 ii, :812-915 ti,
// :163-207 root): workgroup = R waves, wave = rate category, items of 32 sites as above. Per item a wave forms
// D = P x (child side) with 25 MFMAs per 16 sites (or reads the tip's column), dots it with parent_i * pi_i over the
// lane's states, the four row groups of a site meet through two shuffles; the rates' terms meet in LDS and wave 0
// finishes the site (scaling undone, invariant share, log, pattern weight). The scalar-fed FMA kernel spent its life
// waiting for 20 matrix rows one after the other: C3's 50k sites 33 us for 64 MB, one round of 782 workgroups.
template <int NG, bool CTIP>
__global__ __launch_bounds__(256) void k_edge_lean(const DevEdge e, const GenGeo g, const unsigned long long *__restrict__ tipmap,
                                                   unsigned items_per_block, unsigned ncodes)
{
  typedef LeanGeo<NG> LG;
  constexpr unsigned LD = LG::LD;
  typedef double __attribute__((ext_vector_type(2))) double2v;
  extern __shared__ double lds[];
  const unsigned R = g.R, S = g.S;
  double *M = lds;                             // [R][rows][LD]
  double *PART = lds + (size_t)R * LG::mat;    // [2 buffers][2: terma, terminv][R][32]
  unsigned char *CIDX = reinterpret_cast<unsigned char *>(PART + 2u * 2u * R * 32u); // [256]
  const unsigned lane = threadIdx.x & 63u;
  const unsigned k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned row = lane >> 4, col = lane & 15u;
  const unsigned nitems = (e.sites + 31u) / 32u;
  const unsigned item_first = blockIdx.x * items_per_block;
  const unsigned item_end = min(item_first + items_per_block, nitems);
  const unsigned long long full = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
  double *Mk = M + (size_t)k * LG::mat;
  if (!e.is_root)
  {
    constexpr unsigned N = (LG::rows - 1u) * LD, PER = (N + 63u) / 64u;
    const double *src = e.mat + (size_t)k * S * g.SPT;
    double v[PER];
#pragma unroll
    for (unsigned q = 0; q < PER; ++q)
    {
      const unsigned idx = lane + 64u * q, j = idx / LD, i = idx % LD;
      const bool in = idx < N && j < S && i < S;
      const double x = src[in ? (size_t)j * g.SPT + i : 0];
      v[q] = in ? x : 0.0;
    }
#pragma unroll
    for (unsigned q = 0; q < PER; ++q)
      if (lane + 64u * q < N) Mk[lane + 64u * q] = v[q];
    if (CTIP)
    {
      for (unsigned c0 = threadIdx.x; c0 < 256u; c0 += blockDim.x)
      {
        unsigned ci = kCcAmbiguous;
        if (c0 < ncodes)
        {
          const unsigned long long mk = tipmap[c0];
          ci = mk == full ? LG::gap_col : __popcll(mk) == 1 ? (unsigned)__ffsll((long long)mk) - 1u : kCcAmbiguous;
        }
        CIDX[c0] = (unsigned char)ci;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < LD)
      {
        double sum = 0.0; // ascending j like the reference's set-bit walk
        for (unsigned j = 0; j < S; ++j) sum += Mk[j * LD + lane];
        Mk[LG::gap_col * LD + lane] = sum;
      }
    }
  }
  __syncthreads();
  const unsigned fi = e.fidx[k];
  const double pinv = e.prop_invar ? e.prop_invar[fi] : 0.0;
  const double wk = e.rate_weights[k];
  double pif[NG]; // pi_i of the lane's states
#pragma unroll
  for (int ig = 0; ig < NG; ++ig) pif[ig] = (4u * ig + row < S) ? e.freqs[(size_t)fi * g.SP + 4u * ig + row] : 0.0;
  unsigned loff[NG];
#pragma unroll
  for (int jg = 0; jg < NG; ++jg) loff[jg] = min(4u * jg + row, S - 1u) * 64u + 2u * col;
  const unsigned afrag = row * LD + (lane & 3u);
  double acc = 0.0;
  unsigned buf = 0;

  struct Ops
  {
    double x[NG][2], p[NG][2];
    unsigned code[2];
  };
  auto request = [&](unsigned item, Ops &o) {
    const size_t uo = (size_t)(item >> 1) * g.tile_sz + (size_t)k * S * 64 + (item & 1u) * 32u; // wave-uniform
    const double *pb = e.parent + uo;
#pragma unroll
    for (int jg = 0; jg < NG; ++jg)
    {
      const double2v w = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(pb + loff[jg]));
      o.p[jg][0] = w.x;
      o.p[jg][1] = w.y;
    }
    if (CTIP)
    {
      const unsigned e0 = min(item * 32u + 2u * col, e.sites - 1u), e1 = min(item * 32u + 2u * col + 1u, e.sites - 1u);
      o.code[0] = e.ctip[e0];
      o.code[1] = e.ctip[e1];
    }
    else if (!e.is_root)
    {
      const double *cb = e.child + uo;
#pragma unroll
      for (int jg = 0; jg < NG; ++jg)
      {
        const double2v w = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(cb + loff[jg]));
        o.x[jg][0] = w.x;
        o.x[jg][1] = w.y;
      }
    }
  };
  Ops cur;
  if (item_first < item_end) request(item_first, cur);
  for (unsigned item = item_first; item < item_end; ++item)
  {
    Ops nxt;
    const bool has_next = item + 1u < item_end;
    if (has_next) request(item + 1u, nxt);
    const unsigned n0 = item * 32u + 2u * col;
    const unsigned nc[2] = {min(n0, e.sites - 1u), min(n0 + 1u, e.sites - 1u)};
    double D[NG][2];
    if (e.is_root)
    {
#pragma unroll
      for (int ig = 0; ig < NG; ++ig) D[ig][0] = D[ig][1] = 1.0;
    }
    else if (CTIP)
    {
      const unsigned c0 = CIDX[cur.code[0]], c1 = CIDX[cur.code[1]];
      if (__all(c0 != kCcAmbiguous && c1 != kCcAmbiguous))
      {
        const double *p0 = Mk + c0 * LD + row, *p1 = Mk + c1 * LD + row;
#pragma unroll
        for (int ig = 0; ig < NG; ++ig)
        {
          D[ig][0] = p0[4 * ig];
          D[ig][1] = p1[4 * ig];
        }
      }
      else
      {
        const unsigned long long m0 = tipmap[cur.code[0]], m1 = tipmap[cur.code[1]];
#pragma unroll
        for (int ig = 0; ig < NG; ++ig) D[ig][0] = D[ig][1] = 0.0;
#pragma unroll
        for (int jg = 0; jg < NG; ++jg)
        {
          const double x0 = mfma_x<true>(nullptr, m0, S, 4 * jg + row), x1 = mfma_x<true>(nullptr, m1, S, 4 * jg + row);
#pragma unroll
          for (int ig = 0; ig < NG; ++ig)
          {
            const double a = Mk[afrag + 4 * jg * LD + 4 * ig];
            D[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x0, D[ig][0], 0, 0, 0);
            D[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x1, D[ig][1], 0, 0, 0);
          }
        }
      }
    }
    else
    {
#pragma unroll
      for (int ig = 0; ig < NG; ++ig) D[ig][0] = D[ig][1] = 0.0;
#pragma unroll
      for (int jg = 0; jg < NG; ++jg)
      {
#pragma unroll
        for (int ig = 0; ig < NG; ++ig)
        {
          const double a = Mk[afrag + 4 * jg * LD + 4 * ig];
          D[ig][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, cur.x[jg][0], D[ig][0], 0, 0, 0);
          D[ig][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, cur.x[jg][1], D[ig][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    double *part = PART + (size_t)buf * 2u * R * 32u;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
    {
      double tr = 0.0;
#pragma unroll
      for (int ig = 0; ig < NG; ++ig) tr = fma(cur.p[ig][sg] * pif[ig], D[ig][sg], tr);
      tr += __shfl_xor(tr, 16, 64);
      tr += __shfl_xor(tr, 32, 64);
      if (row == 0)
      {
        if (e.per_rate)
        {
          unsigned mn = 0xFFFFFFFFu, own = 0;
          for (unsigned q = 0; q < R; ++q)
          {
            const unsigned rs = (e.pscaler ? e.pscaler[(size_t)nc[sg] * R + q] : 0u) + (e.cscaler ? e.cscaler[(size_t)nc[sg] * R + q] : 0u);
            mn = min(mn, rs);
            if (q == k) own = rs;
          }
          const unsigned ex = min(own - mn, PLLGPU_RATE_MAXDIFF);
          if (ex) tr *= minlh(ex);
        }
        double ta, ti = 0.0;
        if (pinv > 0.0)
        {
          ta = wk * tr * (1.0 - pinv);
          const int inv = e.invariant ? e.invariant[nc[sg]] : -1;
          if (inv >= 0) ti = wk * e.freqs[(size_t)fi * g.SP + inv] * pinv;
        }
        else
          ta = tr * wk;
        part[(0u * R + k) * 32u + 2u * col + sg] = ta;
        part[(1u * R + k) * 32u + 2u * col + sg] = ti;
      }
    }
    lds_barrier();
    if (k == 0 && row == 0)
    {
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
      {
        const unsigned n = n0 + sg;
        if (n >= e.sites) continue;
        double ta = part[(0u * R) * 32u + 2u * col + sg], ti = part[(1u * R) * 32u + 2u * col + sg];
        for (unsigned q = 1; q < R; ++q)
        {
          ta += part[(0u * R + q) * 32u + 2u * col + sg];
          ti += part[(1u * R + q) * 32u + 2u * col + sg];
        }
        unsigned scal;
        if (e.per_rate)
        {
          scal = 0xFFFFFFFFu;
          for (unsigned q = 0; q < R; ++q)
            scal = min(scal, (e.pscaler ? e.pscaler[(size_t)n * R + q] : 0u) + (e.cscaler ? e.cscaler[(size_t)n * R + q] : 0u));
        }
        else
          scal = (e.pscaler ? e.pscaler[n] : 0u) + (e.cscaler ? e.cscaler[n] : 0u);
        const double site = finish_site(ta, ti, scal, e.is_root) * (double)e.pattern_weights[n];
        if (e.persite) e.persite[n] = site;
        acc += site;
      }
    }
    buf ^= 1u;
    if (has_next) cur = nxt;
  }
  publish_block_sum(e, k == 0 ? wave_sum(acc) : 0.0, 1u);
}
