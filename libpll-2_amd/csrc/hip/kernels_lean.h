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
