// kernels_repeats.h - site-repeats class maps on the device (SURVEY.md section 8 rows a8 / f4;
// src/repeats.c:299-382, pll_update_repeats, and the decision of src/repeats.c:100-110).
//
// A parent's classes are the distinct pairs (left class, right class) of its sites, numbered in order of first
// occurrence; id_site[class] is that first site. The reference walks the sites sequentially through a direct-address
// table (cell = lid + rid * ids_left). The same numbering without the sequential walk, one or two launches per dependency
// level (five where a level may hold large tables), every op of the level in each, NO host round trip between levels:
//
//   k_rep_mark    first[cell] = the lowest site of the cell. A workgroup owns a PART of the op's table (in LDS) and a
//                 RANGE of its sites; what it finds goes to its own copy of the part in the op's slice of the arena -
//                 no atomics on the table, nothing to clear beforehand (round 4 spent 62 % of the update on
//                 device-scope atomicMin). Two builds: k_rep_mark_narrow for small tables over byte maps (few
//                 registers: four workgroups per CU), k_rep_mark for everything else.
//                   small tables (<= kRepSmallCells cells: the levels next to the tips, where ops x sites is largest):
//                     the copies are written through to the coherent level; the op's LAST workgroup (a ticket) folds
//                     them and numbers the classes - a cell's class is the count of cells with a lower first site -
//                     by direct counting in LDS: the op is finished inside this launch;
//                   large tables: plain stores, and three launches (fold, bits | scan, rank) follow:
//   k_rep_fold      first[cell] = the minimum over the ranges' copies
//   k_rep_bits      tables up to kRepBitsCells cells: a workgroup per (op, range of the op's SITES) gathers the folded
//                   table's first sites that fall into its range as a bitmap in LDS, stores its words and their running
//                   bit counts; the op's last workgroup (a ticket) turns the ranges' totals into starts and the class
//                   count - nothing but LDS atomics. Larger tables: the fold sets bit first[cell] with device-scope
//                   atomicOr, and
//   k_rep_scan      one workgroup per op: the running bit count per 32-site word; the class count = all set bits
//   k_rep_rank      a cell's class = the set bits before its first site.
//                 Either way the op leaves table[cell] = class and its class count for the levels above.
//   k_rep_assign  site_id[site] = table[cell(site)] (the table as 16-bit entries in LDS up to 65536 cells); large
//                 tables: the sites the bitmap marks as first also write the class -> first site / child entry maps,
//                 in site order = class order (small tables: the op's last workgroup wrote them).
//
// A small child's site -> class pass runs inside its parent's k_rep_mark instead (kRepFuseLeft, below); what is derived from
// the CONTENTS of the maps is told whether they moved (RepPack::changed).
//
// Whether a parent is compressed at all is decided HERE, by the reference's default rule, from the children's class
// counts as the launches of the levels below left them in device memory: the host enqueues all levels back to back
// and reads every count once at the end (round 4: one blocking hand-off per level). A caller-supplied
// enable_repeats callback keeps the level-by-level form (RepOp::force).
//
// site -> class maps of nodes with at most 256 classes are kept as BYTES (the tips and the first levels above them,
// which is where all sites x ops of the work are): 2 bytes read and 1 written per site and op instead of 8 and 4.
// The 32-bit form the API shows (pll_get_site_id) and the gathering kernels read is produced on demand (k_rep_widen).
// Integer work only: the maps are bit-identical to the reference's.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_common.h"

constexpr int kRepOps = 128;                // ops per launch
constexpr unsigned kRepThreads = 512;       // k_rep_mark workgroup
constexpr unsigned kRepLdsCells = 32768;    // cells of its part of a large table (128 KB of LDS: one workgroup per CU)
constexpr unsigned kRepSmallCells = 1024;   // tables up to here are ranked by direct counting in LDS
constexpr unsigned kRepAssignLds = 65536;   // k_rep_assign keeps tables up to this many cells in LDS, 16 bits per cell
constexpr unsigned kRepAssignThreads = 1024; // ... its workgroup: 16 sites per thread and round, RepPack::assign_iters rounds
constexpr unsigned kRepScanThreads = 1024;                        // k_rep_scan workgroup
constexpr unsigned kRepScanChunk = kRepScanThreads * 32u;         // bitmap words per round of its scan
constexpr unsigned kRepFoldThreads = 256, kRepFoldTiles = 64;     // k_rep_fold: workgroups per op (grid-stride over the cells)
constexpr unsigned kRepBitsCells = 1u << 18;                      // tables up to here: bitmap through k_rep_bits (no atomics on memory)
constexpr unsigned kRepBitsMaxRanges = 32;                        // ... its workgroups per op, at most
constexpr unsigned kRepRankThreads = 256, kRepRankTiles = 64;     // k_rep_rank: workgroups per op (grid-stride over the cells)
constexpr unsigned kRepNarrow = 256;        // up to this many classes: site -> class map in bytes
constexpr unsigned kRepEmpty = 0xFFFFFFFFu;
constexpr unsigned kRepFlag = 0x80000000u;  // counts[]: the op was compressed (low bits: its classes)

struct RepOp
{
  const unsigned char *l8, *r8; // children's site -> class maps, byte form (valid when the child has <= kRepNarrow classes)
  const unsigned *l32, *r32;    // ... 32-bit form (valid otherwise)
  unsigned char *p8;            // out: the parent's map, in the form its class count asks for
  unsigned *p32;
  unsigned *pids;               // out: class -> first site
  unsigned *lent, *rent;        // out: class -> entry of the left / right child (what the gather kernels want)
  unsigned *table;              // this op's slice of the arena: ranges x cells while marking, then cell -> class
  unsigned *final;              // where cell -> class is left: `table`, or - kRepDeferred - a place that outlives the next launches
  unsigned *keep;               // large tables: the node's bitmap of first sites as k_rep_bits last left it + one word: 1 while that is
                                // what the node's maps derive from (RepPack::changed); or null
  unsigned *bitmap;             // large tables: [wstride] bitmap over the sites (ZERO between launches), then [wstride] running bit counts
  int lsrc, rsrc;               // op of this call that produces the child (index into counts[]), or -1: nleft / nright are given
  unsigned nleft, nright;
  unsigned slot;                // this op's index in counts[]
  unsigned force;               // 1: the host decided to compress (enable_repeats callback); 0: the default rule, here
  unsigned slice;               // cells in `table`
  unsigned flags;               // kRepFuseLeft | kRepFuseRight | kRepDeferred
};
typedef const RepOp __attribute__((address_space(4))) *crepop_p;
// A child's site -> class pass inside its parent's mark launch. A child X with a table of at most kRepFuseCells cells over
// byte maps, produced by this call and read by ONE op P of it: k_rep_assign leaves X out (kRepDeferred); P's workgroups -
// which read X's map anyway - form it from X's own children's maps and X's table (bytes in LDS) as they go, and write it
// for whoever comes later (kRepFuseLeft / kRepFuseRight on P). Two launches and one pass over the maps less per level.
constexpr unsigned kRepFuseLeft = 1u, kRepFuseRight = 2u, kRepDeferred = 4u;
constexpr unsigned kRepFuseCells = 256;

struct RepPack
{
  const RepOp *ops;       // device array: this launch's ops
  const RepOp *all_ops;   // ... the call's (RepOp::lsrc / rsrc index it)
  unsigned *counts;       // [ncounts] per op of the CALL: kRepFlag | classes, or 0 (not compressed)
  unsigned *tickets;      // [kRepOps] arrivals of an op's workgroups (small tables; 0 between launches)
  unsigned *launch_ticket; // ops of the launch whose count is known (0 between launches)
  unsigned *host_counts;  // mapped host memory: counts[] for the host, then the sequence word, then an error word
  unsigned ncounts;       // ops in the call
  unsigned host_cap;      // entries before the sequence word
  unsigned nops;          // ops in this launch
  unsigned wgs;           // k_rep_assign: workgroups (tiles of sites) per op
  unsigned mark_wgs;      // k_rep_mark: workgroups per op
  unsigned mark_lds_cells; // ... and the cells of a large table's part (its LDS)
  unsigned sites;
  unsigned lookup;        // pll_repeats_t::lookup_buffer_size: the pair table a compressed parent may use
  unsigned lds_cells;     // k_rep_assign: tables up to this many cells go to LDS
  unsigned assign_iters;  // ... rounds of 16384 sites per workgroup
  unsigned wstride;       // words per bitmap, a multiple of kRepScanChunk
  unsigned sequence;
  unsigned publish;       // the call's last kernel that produces counts: the op that reports last hands the counts to the host
  unsigned has_rank;      // k_rep_fold + k_rep_scan + k_rep_rank follow this k_rep_mark (without them a large table is an error: 2)
  unsigned has_narrow, has_general; // which builds of k_rep_mark this level's launch consists of
  unsigned max_ranges;    // site ranges per part of a large table, at most
  unsigned bit_ranges, bit_words; // k_rep_bits follows k_rep_fold: its workgroups per op and their bitmap words each (0: k_rep_scan)
  unsigned *changed;      // one word: `sequence` of the last call that left a class -> child entry map other than it found it
                          // (what is derived from the CONTENTS of those maps - k_sub_pack - is redone only then)
  int fenced;             // kernels_common.h: handoff_*
};

// classes of a node as the levels above see them (pernode_ids): 0 when it is not compressed - the op was not enabled,
// or it found as many classes as sites (src/repeats.c:364-370)
__device__ __forceinline__ unsigned rep_ids(unsigned count_word, unsigned sites)
{
  const unsigned n = count_word & ~kRepFlag;
  return (count_word & kRepFlag) && n < sites ? n : 0u;
}

struct RepShape
{
  unsigned nl, nr, ncells;
  bool on;
};

// the decision of pll_default_enable_repeats (src/repeats.c:100-110) from the children's counts
__device__ __forceinline__ RepShape rep_shape(const RepPack &p, crepop_p o)
{
  RepShape s;
  s.nl = o->lsrc >= 0 ? rep_ids(p.counts[o->lsrc], p.sites) : o->nleft;
  s.nr = o->rsrc >= 0 ? rep_ids(p.counts[o->rsrc], p.sites) : o->nright;
  const unsigned long long cells = (unsigned long long)s.nl * s.nr;
  s.on = cells != 0ull && cells <= (unsigned long long)o->slice;
  if (!o->force) s.on = s.on && cells < (unsigned long long)p.lookup && s.nl <= p.sites / 2u && s.nr <= p.sites / 2u;
  s.ncells = s.on ? (unsigned)cells : 0u;
  return s;
}

// sixteen consecutive entries of a site -> class map as they come from memory (the buffers are padded to whole groups
// of sixteen), and as numbers
template <bool NARROW>
struct RepRaw
{
  uint4 q[NARROW ? 1 : 4];
};

template <bool NARROW>
__device__ __forceinline__ RepRaw<NARROW> rep_fetch16(const unsigned char *m8, const unsigned *m32, unsigned s)
{
  RepRaw<NARROW> raw;
  if (NARROW)
    raw.q[0] = *reinterpret_cast<const uint4 *>(m8 + s);
  else
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(m32 + s);
#pragma unroll
    for (unsigned q = 0; q < 4u; ++q) raw.q[q] = src[q];
  }
  return raw;
}

template <bool NARROW>
__device__ __forceinline__ void rep_unpack16(const RepRaw<NARROW> &raw, unsigned (&v)[16])
{
  if (NARROW)
  {
    const unsigned w[4] = {raw.q[0].x, raw.q[0].y, raw.q[0].z, raw.q[0].w};
#pragma unroll
    for (unsigned e = 0; e < 16u; ++e) v[e] = (w[e >> 2] >> (8u * (e & 3u))) & 255u;
  }
  else
  {
#pragma unroll
    for (unsigned q = 0; q < 4u; ++q)
    {
      v[4u * q] = raw.q[NARROW ? 0 : q].x;
      v[4u * q + 1u] = raw.q[NARROW ? 0 : q].y;
      v[4u * q + 2u] = raw.q[NARROW ? 0 : q].z;
      v[4u * q + 3u] = raw.q[NARROW ? 0 : q].w;
    }
  }
}

template <bool NARROW>
__device__ __forceinline__ void rep_load16(const unsigned char *m8, const unsigned *m32, unsigned s, unsigned (&v)[16])
{
  rep_unpack16<NARROW>(rep_fetch16<NARROW>(m8, m32, s), v);
}

// sixteen sites against the table part in LDS. All sixteen looks first (a site outside the range or the part looks at
// cell 0 and is dropped afterwards), then the few atomics: one wait for LDS per group instead of one per site. Two
// sites of the group in one cell both see the state before the group - the atomic sorts them out.
// WHOLE: the part is the whole table (no filter); FULL: all sixteen sites lie inside the range.
template <bool WHOLE, bool FULL, unsigned N>
__device__ __forceinline__ void rep_group(const unsigned (&l)[N], const unsigned (&r)[N], unsigned nleft, unsigned s, unsigned s1, unsigned lo, unsigned pcells, unsigned *lds)
{
  unsigned idx[N], seen[N];
#pragma unroll
  for (unsigned e = 0; e < N; ++e)
  {
    const unsigned i = WHOLE ? l[e] + r[e] * nleft : l[e] + r[e] * nleft - lo;
    bool ok = WHOLE || i < pcells;
    if (!FULL) ok = ok && s + e < s1;
    idx[e] = ok ? i : kRepEmpty;
    seen[e] = lds[ok ? i : 0u];
  }
#pragma unroll
  for (unsigned e = 0; e < N; ++e)
    if (((WHOLE && FULL) || idx[e] != kRepEmpty) && seen[e] > s + e) atomicMin(&lds[idx[e]], s + e);
}

// the same for sixteen sites whose maps are both bytes, in two halves straight from the loaded words: half the
// registers (k_rep_mark_narrow is built for 64)
template <bool WHOLE, bool FULL>
__device__ __forceinline__ void rep_group_bytes(const uint4 &lq, const uint4 &rq, unsigned nleft, unsigned s, unsigned s1, unsigned lo, unsigned pcells, unsigned *lds)
{
#pragma unroll
  for (unsigned h = 0; h < 2u; ++h)
  {
    const unsigned lw[2] = {h ? lq.z : lq.x, h ? lq.w : lq.y}, rw[2] = {h ? rq.z : rq.x, h ? rq.w : rq.y};
    unsigned l[8], r[8];
#pragma unroll
    for (unsigned e = 0; e < 8u; ++e)
    {
      l[e] = (lw[e >> 2] >> (8u * (e & 3u))) & 255u;
      r[e] = (rw[e >> 2] >> (8u * (e & 3u))) & 255u;
    }
    rep_group<WHOLE, FULL, 8>(l, r, nleft, s + 8u * h, s1, lo, pcells, lds);
  }
}

// sites [s0, s1) of one op against the table part [lo, lo + pcells) in LDS. Ascending sites per thread and a look
// before the LDS atomic: after its first few sites a thread mostly finds a lower site already there. The next group's
// maps are requested before this one is worked on.
template <bool L8, bool R8, bool WHOLE>
__device__ __forceinline__ void rep_scan(crepop_p o, unsigned nleft, unsigned s0, unsigned s1, unsigned lo, unsigned pcells, unsigned *lds)
{
  const unsigned char *l8 = o->l8, *r8 = o->r8;
  const unsigned *l32 = o->l32, *r32 = o->r32;
  unsigned s = s0 + threadIdx.x * 16u;
  if (s >= s1) return;
  RepRaw<L8> lraw = rep_fetch16<L8>(l8, l32, s);
  RepRaw<R8> rraw = rep_fetch16<R8>(r8, r32, s);
  for (;;)
  {
    const unsigned sn = s + kRepThreads * 16u;
    const bool more = sn < s1;
    RepRaw<L8> lnext = lraw;
    RepRaw<R8> rnext = rraw;
    if (more)
    {
      lnext = rep_fetch16<L8>(l8, l32, sn);
      rnext = rep_fetch16<R8>(r8, r32, sn);
    }
    if (L8 && R8)
    {
      if (s + 16u <= s1) rep_group_bytes<WHOLE, true>(lraw.q[0], rraw.q[0], nleft, s, s1, lo, pcells, lds);
      else rep_group_bytes<WHOLE, false>(lraw.q[0], rraw.q[0], nleft, s, s1, lo, pcells, lds);
    }
    else
    {
      unsigned l[16], r[16];
      rep_unpack16<L8>(lraw, l);
      rep_unpack16<R8>(rraw, r);
      if (s + 16u <= s1) rep_group<WHOLE, true, 16>(l, r, nleft, s, s1, lo, pcells, lds);
      else rep_group<WHOLE, false, 16>(l, r, nleft, s, s1, lo, pcells, lds);
    }
    if (!more) return;
    lraw = lnext;
    rraw = rnext;
    s = sn;
  }
}

// A fused child (kRepFuseLeft / Right): what the scan needs of it.
struct RepFused
{
  const unsigned char *a8, *b8; // the child's own children's maps
  unsigned char *out8;          // the child's map, written here (null: another workgroup of the op writes these sites)
  const unsigned char *tab;     // LDS: the child's cell -> class
  unsigned nl;                  // its left child's classes
};

// sixteen sites of a fused child: its class per site from its children's maps, as the bytes its map holds
__device__ __forceinline__ uint4 rep_fused16(const RepFused &f, const uint4 &a, const uint4 &b, unsigned s)
{
  const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
  unsigned ow[4];
#pragma unroll
  for (unsigned q = 0; q < 4u; ++q)
  {
    unsigned w = 0u;
#pragma unroll
    for (unsigned e = 0; e < 4u; ++e)
    {
      // (the padding behind the last site holds anything: kept inside the table, stored into the padding of the child's map)
      const unsigned cell = (((aw[q] >> (8u * e)) & 255u) + ((bw[q] >> (8u * e)) & 255u) * f.nl) & (kRepFuseCells - 1u);
      w |= (unsigned)f.tab[cell] << (8u * e);
    }
    ow[q] = w;
  }
  const uint4 out = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  if (f.out8) *reinterpret_cast<uint4 *>(f.out8 + s) = out;
  return out;
}

// rep_scan for two byte maps of which one or both are fused children
template <bool LF, bool RF, bool WHOLE>
__device__ __forceinline__ void rep_scan_fused(crepop_p o, const RepFused &fl, const RepFused &fr, unsigned nleft, unsigned s0, unsigned s1, unsigned lo, unsigned pcells,
                                               unsigned *lds)
{
  const unsigned char *la = LF ? fl.a8 : o->l8, *lb = fl.b8, *ra = RF ? fr.a8 : o->r8, *rb = fr.b8;
  unsigned s = s0 + threadIdx.x * 16u;
  if (s >= s1) return;
  uint4 la_q = *reinterpret_cast<const uint4 *>(la + s), ra_q = *reinterpret_cast<const uint4 *>(ra + s);
  uint4 lb_q = la_q, rb_q = ra_q;
  if (LF) lb_q = *reinterpret_cast<const uint4 *>(lb + s);
  if (RF) rb_q = *reinterpret_cast<const uint4 *>(rb + s);
  for (;;)
  {
    const unsigned sn = s + kRepThreads * 16u;
    const bool more = sn < s1;
    uint4 la_n = la_q, lb_n = lb_q, ra_n = ra_q, rb_n = rb_q;
    if (more)
    {
      la_n = *reinterpret_cast<const uint4 *>(la + sn);
      ra_n = *reinterpret_cast<const uint4 *>(ra + sn);
      if (LF) lb_n = *reinterpret_cast<const uint4 *>(lb + sn);
      if (RF) rb_n = *reinterpret_cast<const uint4 *>(rb + sn);
    }
    const uint4 lq = LF ? rep_fused16(fl, la_q, lb_q, s) : la_q;
    const uint4 rq = RF ? rep_fused16(fr, ra_q, rb_q, s) : ra_q;
    if (s + 16u <= s1) rep_group_bytes<WHOLE, true>(lq, rq, nleft, s, s1, lo, pcells, lds);
    else rep_group_bytes<WHOLE, false>(lq, rq, nleft, s, s1, lo, pcells, lds);
    if (!more) return;
    la_q = la_n;
    lb_q = lb_n;
    ra_q = ra_n;
    rb_q = rb_n;
    s = sn;
  }
}

template <bool WHOLE>
__device__ __forceinline__ void rep_scan_fused_forms(crepop_p o, const RepFused &fl, const RepFused &fr, unsigned flags, unsigned nleft, unsigned s0, unsigned s1, unsigned lo,
                                                     unsigned pcells, unsigned *lds)
{
  if ((flags & kRepFuseLeft) && (flags & kRepFuseRight)) rep_scan_fused<true, true, WHOLE>(o, fl, fr, nleft, s0, s1, lo, pcells, lds);
  else if (flags & kRepFuseLeft) rep_scan_fused<true, false, WHOLE>(o, fl, fr, nleft, s0, s1, lo, pcells, lds);
  else rep_scan_fused<false, true, WHOLE>(o, fl, fr, nleft, s0, s1, lo, pcells, lds);
}

// the maps of an op's fused children alone, sites [s0, s1): the op itself stays uncompressed
__device__ __forceinline__ void rep_fused_only(const RepFused &f, unsigned s0, unsigned s1)
{
  for (unsigned s = s0 + threadIdx.x * 16u; s < s1; s += kRepThreads * 16u)
    (void)rep_fused16(f, *reinterpret_cast<const uint4 *>(f.a8 + s), *reinterpret_cast<const uint4 *>(f.b8 + s), s);
}

// what P's workgroup needs of its fused child `side`: the child's table as bytes in `tab` (LDS; the caller's barrier
// follows). live = false: nothing to do for this side (not fused, or the child has no maps).
__device__ __forceinline__ bool rep_fused_open(const RepPack &p, crepop_p o, bool right, unsigned char *tab, RepFused &f)
{
  f.a8 = f.b8 = nullptr;
  f.out8 = nullptr;
  f.tab = tab;
  f.nl = 0u;
  if (!(o->flags & (right ? kRepFuseRight : kRepFuseLeft))) return false;
  crepop_p x = (crepop_p)(uintptr_t)p.all_ops + (right ? o->rsrc : o->lsrc);
  if (!rep_ids(p.counts[x->slot], p.sites)) return false; // (uniform) the child is not compressed: no map
  const RepShape xs = rep_shape(p, x);
  f.a8 = x->l8;
  f.b8 = x->r8;
  f.out8 = x->p8;
  f.nl = xs.nl;
  const unsigned *__restrict__ final = x->final;
  for (unsigned i = threadIdx.x; i < xs.ncells && i < kRepFuseCells; i += kRepThreads) tab[i] = (unsigned char)final[i];
  return true;
}

template <bool WHOLE>
__device__ __forceinline__ void rep_scan_forms(crepop_p o, const RepShape &sh, unsigned s0, unsigned s1, unsigned lo, unsigned pcells, unsigned *lds)
{
  const bool l8 = sh.nl <= kRepNarrow, r8 = sh.nr <= kRepNarrow;
  if (l8 && r8) rep_scan<true, true, WHOLE>(o, sh.nl, s0, s1, lo, pcells, lds);
  else if (l8) rep_scan<true, false, WHOLE>(o, sh.nl, s0, s1, lo, pcells, lds);
  else if (r8) rep_scan<false, true, WHOLE>(o, sh.nl, s0, s1, lo, pcells, lds);
  else rep_scan<false, false, WHOLE>(o, sh.nl, s0, s1, lo, pcells, lds);
}

__device__ __forceinline__ unsigned rep_coherent_load(const unsigned *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The hand-over of the class counts. Called by every thread of ONE workgroup per op of the launch, in the kernel of the
// launch that knows the last counts (p.publish): the workgroup that arrives last copies the counts of the whole call
// to the host and then writes the sequence word the host polls.
__device__ __forceinline__ void rep_arrive(const RepPack &p)
{
  __shared__ unsigned s_lastop;
  if (!p.publish) return; // (uniform over the launch)
  if (threadIdx.x == 0u)
  {
    handoff_before_ticket(p.fenced); // this op's count has been performed
    const unsigned t = __hip_atomic_fetch_add(p.launch_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_lastop = (t == p.nops - 1u) ? 1u : 0u;
    if (s_lastop)
    {
      handoff_after_last_ticket(p.fenced);
      __hip_atomic_store(p.launch_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!s_lastop) return;
  for (unsigned i = threadIdx.x; i < p.ncounts; i += blockDim.x)
    __hip_atomic_store(&p.host_counts[i], rep_coherent_load(&p.counts[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  handoff_before_sequence(p.fenced); // every wave: its counts are in host memory ...
  __syncthreads();
  if (threadIdx.x == 0u) __hip_atomic_store(&p.host_counts[p.host_cap], p.sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // ... before the word the host polls
}

__device__ __forceinline__ void rep_store_count(const RepPack &p, crepop_p o, unsigned count_word, unsigned error)
{
  if (threadIdx.x != 0u) return;
  __hip_atomic_store(&p.counts[o->slot], count_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (error) __hip_atomic_store(&p.host_counts[p.host_cap + 1u], error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the other workgroups' copies of a table part folded into this workgroup's own (still in LDS): `nranges` copies of
// `ncells` cells each, the part is cells [lo, lo + pcells), `mine` is the copy to leave out. The (copy, cell) pairs are
// dealt out over the threads and a thread requests BATCH of them before it looks at the first: every load is a trip to
// the coherent level, and a loop over the copies made the tail of a launch as many trips long as an op had ranges
// (32 ranges of a 16-cell table: 25 us of a 30 us launch).
template <unsigned BATCH>
__device__ __forceinline__ void rep_fold_copies(const unsigned *table, unsigned ncells, unsigned nranges, unsigned mine, unsigned lo, unsigned pcells, unsigned *lds)
{
  const unsigned total = nranges * pcells;
  for (unsigned base = 0; base < total; base += BATCH * kRepThreads)
  {
    unsigned t[BATCH], cell[BATCH];
#pragma unroll
    for (unsigned q = 0; q < BATCH; ++q)
    {
      const unsigned item = base + q * kRepThreads + threadIdx.x;
      const unsigned r = item / pcells;
      cell[q] = item - r * pcells;
      t[q] = item < total && r != mine ? rep_coherent_load(table + (size_t)r * ncells + lo + cell[q]) : kRepEmpty;
    }
#pragma unroll
    for (unsigned q = 0; q < BATCH; ++q)
      if (t[q] != kRepEmpty && lds[cell[q]] > t[q]) atomicMin(&lds[cell[q]], t[q]);
  }
}

// small table, complete in LDS: a cell's class = the cells with a lower first site. Returns the class count (every thread).
// `moved`: nonzero where this thread's entries of the class -> child entry maps differ from what stood there before (read
// ahead of the stores, looked at by the caller once the count is on its way: RepPack::changed).
static_assert(kRepSmallCells <= 2u * kRepThreads, "rep_rank_small: two cells per thread");
__device__ __forceinline__ unsigned rep_rank_small(crepop_p o, unsigned ncells, unsigned nl, const unsigned *lds, unsigned *s_count, unsigned &moved)
{
  unsigned *table = o->final;
  unsigned *pids = o->pids, *lent = o->lent, *rent = o->rent;
  if (threadIdx.x == 0u) *s_count = 0u;
  __syncthreads();
  unsigned was[2][2], is[2][2];
#pragma unroll
  for (unsigned k = 0; k < 2u; ++k)
  {
    was[k][0] = was[k][1] = is[k][0] = is[k][1] = 0u;
    const unsigned c = threadIdx.x + k * kRepThreads;
    if (c >= ncells) continue;
    const unsigned v = lds[c];
    if (v == kRepEmpty)
    {
      table[c] = kRepEmpty;
      continue;
    }
    unsigned rank = 0;
    for (unsigned j = 0; j < ncells; ++j) rank += lds[j] < v ? 1u : 0u; // (an empty cell is never lower)
    table[c] = rank;
    pids[rank] = v;
    was[k][0] = lent[rank];
    was[k][1] = rent[rank];
    lent[rank] = is[k][0] = c % nl;
    rent[rank] = is[k][1] = c / nl;
    atomicAdd(s_count, 1u);
  }
  __syncthreads();
  moved = (was[0][0] ^ is[0][0]) | (was[0][1] ^ is[0][1]) | (was[1][0] ^ is[1][0]) | (was[1][1] ^ is[1][1]);
  return *s_count;
}

// (op, piece) of a workgroup in a 1-D grid of (ops rounded up to eight) x `per` workgroups: the workgroups of one op on
// ONE XCD (workgroups go to the XCDs round-robin by linear id), so that what they share - the op's maps, its table, its
// bitmap - is found in that XCD's L2
__device__ __forceinline__ bool rep_place(unsigned nops, unsigned per, unsigned &opi, unsigned &piece)
{
  const unsigned j = blockIdx.x >> 3, grp = j / per;
  piece = j - grp * per;
  opi = grp * 8u + (blockIdx.x & 7u);
  return opi < nops;
}

// How k_rep_mark's workgroups share an op: `parts` of the table (as few as LDS allows: every part scans the op's sites
// again) x `ranges` of the sites, or - more parts than workgroups - the parts dealt out, one range.
struct RepSplit
{
  unsigned nparts, nranges;
  bool looped;
};
__device__ __forceinline__ RepSplit rep_split(const RepPack &p, crepop_p o, unsigned ncells)
{
  RepSplit sp;
  const bool large = ncells > kRepSmallCells;
  const unsigned lds_cells = large ? p.mark_lds_cells : kRepSmallCells;
  sp.nparts = (ncells + lds_cells - 1u) / lds_cells;
  sp.looped = sp.nparts >= p.mark_wgs;
  sp.nranges = sp.looped ? 1u : p.mark_wgs / sp.nparts;
  // a range is worth its copy of the table: a small table's is a few cells (folded by the op's last workgroup, which
  // pays a trip to memory however few they are), a large one's is written and read once more by k_rep_fold
  const unsigned per_range = large ? 32768u : 8192u;
  const unsigned by_sites = (p.sites + per_range - 1u) / per_range;
  if (sp.nranges > by_sites) sp.nranges = by_sites;
  if (large && sp.nranges > p.max_ranges) sp.nranges = p.max_ranges;
  if (sp.nranges > o->slice / ncells) sp.nranges = o->slice / ncells; // (>= 1: rep_shape; the host sizes the slice for all of them)
  return sp;
}

// k_rep_mark comes in two builds. Where all sites x ops of the work are - next to the tips - the tables are small and
// both children's maps are bytes: that form alone needs few registers, and built by itself (k_rep_mark_narrow, at most 64
// VGPRs) FOUR of its workgroups fit a CU instead of two: the whole grid is resident at once (a launch of 1024
// workgroups took two rounds of them: 23 us for a 10 us chain on a 125k-site shard) and the VALU-bound scan has twice
// the waves to issue from. Everything else - 32-bit maps, large tables - is the general build. A level's launch
// consists of the builds its ops may need (the host knows bounds); an op is taken by the build its actual counts ask
// for; parents that stay uncompressed are reported by the general build if it is there.
template <bool NARROW, bool FUSED>
__device__ __forceinline__ void rep_mark(const RepPack &p, unsigned *rep_lds)
{
  __shared__ unsigned s_count;
  __shared__ unsigned s_last;
  __shared__ unsigned char s_tab[2][kRepFuseCells];
  unsigned opi, w;
  if (!rep_place(p.nops, p.mark_wgs, opi, w)) return;
  crepop_p o = (crepop_p)(uintptr_t)p.ops + opi;
  const RepShape sh = rep_shape(p, o);
  const bool large = sh.ncells > kRepSmallCells;
  const bool publishes = !p.has_rank; // no k_rep_scan behind this launch: the counts are handed over here
  const unsigned fuse = FUSED ? o->flags & (kRepFuseLeft | kRepFuseRight) : 0u; // (a launch with fused children takes the builds that know them)
  RepFused fl = {}, fr = {};
  if (!sh.on || (large && !p.has_rank))
  {
    if (NARROW != !p.has_general) return;
    if (fuse)
    {
      // the maps of its fused children are still this op's to write: its workgroups share the sites
      const bool lv = rep_fused_open(p, o, false, s_tab[0], fl), rv = rep_fused_open(p, o, true, s_tab[1], fr);
      __syncthreads();
      const unsigned rs = ((p.sites + p.mark_wgs - 1u) / p.mark_wgs + 15u) & ~15u;
      const unsigned s0 = w * rs < p.sites ? w * rs : p.sites, s1 = s0 + rs < p.sites ? s0 + rs : p.sites;
      if (lv) rep_fused_only(fl, s0, s1);
      if (rv) rep_fused_only(fr, s0, s1);
    }
    if (w != 0u) return;
    const unsigned long long cells = (unsigned long long)sh.nl * sh.nr;
    // 1: a table the slice cannot hold although the rule admits it - the host's bound was wrong; 2: a large table in
    // a launch that came without the kernels for it - the host's forecast was wrong
    const bool overflow = !sh.on && cells > (unsigned long long)o->slice && (o->force || (cells < (unsigned long long)p.lookup && sh.nl <= p.sites / 2u && sh.nr <= p.sites / 2u));
    rep_store_count(p, o, 0u, overflow ? 1u : sh.on ? 2u : 0u);
    if (publishes) rep_arrive(p);
    return;
  }
  const bool narrow_op = !large && sh.nl <= kRepNarrow && sh.nr <= kRepNarrow;
  if (NARROW ? !narrow_op : (narrow_op && p.has_narrow)) return; // the other build's
  const unsigned ncells = sh.ncells;
  const RepSplit sp = rep_split(p, o, ncells);
  const unsigned used = sp.looped ? (sp.nparts < p.mark_wgs ? sp.nparts : p.mark_wgs) : sp.nparts * sp.nranges;
  if (w >= used) return;
  const unsigned rs = ((p.sites + sp.nranges - 1u) / sp.nranges + 15u) & ~15u; // sites per range, whole groups of sixteen
  const unsigned range = sp.looped ? 0u : w / sp.nparts;
  const unsigned s0 = range * rs < p.sites ? range * rs : p.sites, s1 = s0 + rs < p.sites ? s0 + rs : p.sites;
  if (fuse)
  {
    // (an op that is compressed has compressed children: both fused sides are live; the barriers below come before the scan)
    (void)rep_fused_open(p, o, false, s_tab[0], fl);
    (void)rep_fused_open(p, o, true, s_tab[1], fr);
  }
  if (!large)
  {
    // the whole table in LDS; the op's last workgroup folds the ranges and numbers the classes
    for (unsigned i = threadIdx.x; i < ncells; i += kRepThreads) rep_lds[i] = kRepEmpty;
    __syncthreads();
    if (fuse) rep_scan_fused_forms<true>(o, fl, fr, fuse, sh.nl, s0, s1, 0u, ncells, rep_lds);
    else if (NARROW) rep_scan<true, true, true>(o, sh.nl, s0, s1, 0u, ncells, rep_lds);
    else rep_scan_forms<true>(o, sh, s0, s1, 0u, ncells, rep_lds);
    __syncthreads();
    if (sp.nranges > 1u)
    {
      // this workgroup's copy: written through to the coherent level (kernels_common.h: partial_store); every wave's
      // stores have been performed before the workgroup takes its ticket
      unsigned *dst = o->table + (size_t)range * ncells;
      for (unsigned i = threadIdx.x; i < ncells; i += kRepThreads) __hip_atomic_store(dst + i, rep_lds[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0u)
      {
        if (p.fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned t = __hip_atomic_fetch_add(&p.tickets[opi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == sp.nranges - 1u) ? 1u : 0u;
        if (s_last)
        {
          handoff_after_last_ticket(p.fenced);
          __hip_atomic_store(&p.tickets[opi], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();
      if (!s_last) return;
      rep_fold_copies<NARROW ? 8 : 32>(o->table, ncells, sp.nranges, range, 0u, ncells, rep_lds);
      __syncthreads();
    }
    unsigned moved;
    const unsigned classes = rep_rank_small(o, ncells, sh.nl, rep_lds, &s_count, moved);
    rep_store_count(p, o, kRepFlag | classes, 0u);
    if (publishes) rep_arrive(p);
    if (moved) *p.changed = p.sequence; // (every writer of the call stores the same word)
    if (threadIdx.x == 0u && o->keep) o->keep[(p.sites + 31u) / 32u] = 0u; // (not what k_rep_bits left any more)
    return;
  }
  if (NARROW) return; // (not reached: a large table is never the narrow build's)
  // large: this workgroup's copy of its part(s) for k_rep_fold, plain stores
  const unsigned pc = (ncells + sp.nparts - 1u) / sp.nparts; // cells per part (<= mark_lds_cells)
  unsigned char *const lout = fl.out8, *const rout = fr.out8;
  for (unsigned part = sp.looped ? w : w % sp.nparts; part < sp.nparts; part += sp.looped ? used : sp.nparts)
  {
    const unsigned lo = part * pc, pcells = lo + pc <= ncells ? pc : ncells - lo;
    for (unsigned i = threadIdx.x; i < pcells; i += kRepThreads) rep_lds[i] = kRepEmpty;
    __syncthreads();
    if (fuse)
    {
      // every part scans the range's sites; the fused children's maps are written along with part 0
      fl.out8 = part == 0u ? lout : nullptr;
      fr.out8 = part == 0u ? rout : nullptr;
      if (sp.nparts == 1u) rep_scan_fused_forms<true>(o, fl, fr, fuse, sh.nl, s0, s1, 0u, ncells, rep_lds);
      else rep_scan_fused_forms<false>(o, fl, fr, fuse, sh.nl, s0, s1, lo, pcells, rep_lds);
    }
    else if (sp.nparts == 1u) rep_scan_forms<true>(o, sh, s0, s1, 0u, ncells, rep_lds);
    else rep_scan_forms<false>(o, sh, s0, s1, lo, pcells, rep_lds);
    __syncthreads();
    unsigned *dst = o->table + (size_t)range * ncells + lo;
    for (unsigned i = threadIdx.x; i < pcells; i += kRepThreads) dst[i] = rep_lds[i];
    __syncthreads(); // rep_lds is reused by the next part
  }
}

// Launch: rep_place with mark_wgs workgroups per op.
__global__ __launch_bounds__(kRepThreads, 8) void k_rep_mark_narrow(const RepPack p)
{
  extern __shared__ unsigned rep_lds[];
  rep_mark<true, false>(p, rep_lds);
}

// ... with fused children (twice the maps in flight: 85 registers, three workgroups per CU - still the whole grid at once)
__global__ __launch_bounds__(kRepThreads, 6) void k_rep_mark_narrow_fused(const RepPack p)
{
  extern __shared__ unsigned rep_lds[];
  rep_mark<true, true>(p, rep_lds);
}

__global__ __launch_bounds__(kRepThreads) void k_rep_mark(const RepPack p)
{
  extern __shared__ unsigned rep_lds[];
  rep_mark<false, true>(p, rep_lds);
}

// Large tables, second step: the copies of the ranges folded - first[cell] = the lowest over them, left in copy 0 - and,
// where k_rep_scan follows (tables too large for k_rep_bits' passes over the cells), bit first[cell] of the op's
// bitmap set (shared by the whole launch grid: atomics at the coherent level).
// Launch: rep_place with kRepFoldTiles workgroups per op, which stride over the op's cells.
__global__ __launch_bounds__(kRepFoldThreads) void k_rep_fold(const RepPack p)
{
  unsigned opi, tile;
  if (!rep_place(p.nops, kRepFoldTiles, opi, tile)) return;
  crepop_p o = (crepop_p)(uintptr_t)p.ops + opi;
  const RepShape sh = rep_shape(p, o);
  if (!sh.on || sh.ncells <= kRepSmallCells) return;
  const unsigned ncells = sh.ncells;
  const unsigned nranges = rep_split(p, o, ncells).nranges;
  unsigned *__restrict__ table = o->table, *__restrict__ bitmap = o->bitmap;
  for (unsigned base = tile * kRepFoldThreads * 4u; base < ncells; base += kRepFoldTiles * kRepFoldThreads * 4u)
  {
    unsigned v[4] = {kRepEmpty, kRepEmpty, kRepEmpty, kRepEmpty};
    for (unsigned r = 0; r < nranges; ++r)
    {
      unsigned t[4];
#pragma unroll
      for (unsigned q = 0; q < 4u; ++q)
      {
        const unsigned c = base + q * kRepFoldThreads + threadIdx.x;
        t[q] = c < ncells ? table[(size_t)r * ncells + c] : kRepEmpty;
      }
#pragma unroll
      for (unsigned q = 0; q < 4u; ++q) v[q] = t[q] < v[q] ? t[q] : v[q];
    }
#pragma unroll
    for (unsigned q = 0; q < 4u; ++q)
    {
      const unsigned c = base + q * kRepFoldThreads + threadIdx.x;
      if (c >= ncells) continue;
      if (nranges > 1u) table[c] = v[q];
      if (!p.bit_ranges && v[q] != kRepEmpty) __hip_atomic_fetch_or(&bitmap[v[q] >> 5], 1u << (v[q] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Large tables, the bitmap WITHOUT atomics on memory (tables up to kRepBitsCells cells): workgroup b of an op owns the
// sites of `bit_words` bitmap words, reads the first site of EVERY cell of the op (from that XCD's L2: the workgroups
// of an op share it) and sets the bits of the first sites in its range in LDS; then the running bit count of its words
// (from the range's start) and its total. The op's last workgroup (a ticket) turns the totals into the ranges' starting
// counts and knows the class count. Replaces the device-scope atomicOr of k_rep_fold (0.6 M of them in C4's level 2: 37 us
// at 1M sites, 18 us on a shard - as much as everything else at that level) and k_rep_scan's single workgroup per op.
// Launch: rep_place with bit_ranges workgroups per op; LDS bit_words words. Every op of the launch passes here.
__global__ __launch_bounds__(kRepScanThreads) void k_rep_bits(const RepPack p)
{
  extern __shared__ unsigned rep_lds[];
  __shared__ unsigned wsum[kRepScanThreads / 64u];
  __shared__ unsigned s_last;
  unsigned opi, b;
  if (!rep_place(p.nops, p.bit_ranges, opi, b)) return;
  crepop_p o = (crepop_p)(uintptr_t)p.ops + opi;
  const RepShape sh = rep_shape(p, o);
  if (!sh.on || sh.ncells <= kRepSmallCells)
  {
    if (b == 0u) rep_arrive(p); // (its count was left by k_rep_mark)
    return;
  }
  const unsigned ncells = sh.ncells, words = (p.sites + 31u) / 32u;
  const unsigned w0 = b * p.bit_words, w1 = w0 + p.bit_words < words ? w0 + p.bit_words : words; // (w0 may lie beyond the last word: an empty range)
  const unsigned nw = w1 > w0 ? w1 - w0 : 0u;
  const unsigned *__restrict__ table = o->table;
  unsigned *__restrict__ bitmap = o->bitmap, *__restrict__ wprefix = o->bitmap + p.wstride;
  unsigned *totals = o->bitmap + 2u * (size_t)p.wstride; // [bit_ranges] totals, then [bit_ranges] starting counts
  unsigned *__restrict__ keep = o->keep;
  unsigned moved = keep && keep[words] == 1u ? 0u : 1u; // (the word: written by launches before this one)
  for (unsigned w = threadIdx.x; w < p.bit_words; w += kRepScanThreads) rep_lds[w] = 0u;
  __syncthreads();
  // every cell of the op: sixteen 16-byte loads per thread in flight (65536 cells per round - a round is a trip to memory,
  // and with four cells per trip a 125k-site shard's launch was four of them long: 14 us); the slices of the arena
  // start at multiples of four cells and are padded to one
  for (unsigned base = 0; base < ncells; base += 64u * kRepScanThreads)
  {
    uint4 v[16];
#pragma unroll
    for (unsigned q = 0; q < 16u; ++q)
    {
      const unsigned c = base + (q * kRepScanThreads + threadIdx.x) * 4u;
      v[q] = c < ncells ? *reinterpret_cast<const uint4 *>(table + c) : make_uint4(kRepEmpty, kRepEmpty, kRepEmpty, kRepEmpty);
    }
#pragma unroll
    for (unsigned q = 0; q < 16u; ++q)
    {
      const unsigned c = base + (q * kRepScanThreads + threadIdx.x) * 4u;
      const unsigned f[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
      for (unsigned e = 0; e < 4u; ++e)
      {
        const unsigned w = f[e] >> 5;
        if (c + e < ncells && f[e] != kRepEmpty && w >= w0 && w < w1) atomicOr(&rep_lds[w - w0], 1u << (f[e] & 31u));
      }
    }
  }
  __syncthreads();
  // running count: bit_words / 1024 consecutive words per thread (bit_words is a multiple of 1024)
  const unsigned per = p.bit_words / kRepScanThreads, t0 = threadIdx.x * per;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned n = 0;
  for (unsigned q = 0; q < per; ++q) n += __popc(rep_lds[t0 + q]);
  unsigned inc = n;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
  {
    const unsigned t = __shfl_up(inc, off, 64);
    if ((int)lane >= off) inc += t;
  }
  if (lane == 63u) wsum[wave] = inc;
  __syncthreads();
  unsigned before = 0, total = 0;
  for (unsigned w = 0; w < kRepScanThreads / 64u; ++w)
  {
    before += w < wave ? wsum[w] : 0u;
    total += wsum[w];
  }
  unsigned run = before + inc - n;
  for (unsigned q = 0; q < per; ++q)
  {
    const unsigned w = t0 + q, bits = rep_lds[w];
    if (w < nw)
    {
      bitmap[w0 + w] = bits;
      wprefix[w0 + w] = run;
      // the node's first sites as the last call left them: other ones = other class -> first site / child entry maps
      if (keep)
      {
        moved |= keep[w0 + w] ^ bits;
        keep[w0 + w] = bits;
      }
    }
    run += __popc(bits);
  }
  if (moved) *p.changed = p.sequence;
  // the range's total for the op's last workgroup
  if (threadIdx.x == 0u)
  {
    __hip_atomic_store(&totals[b], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    handoff_before_ticket(p.fenced);
    const unsigned t = __hip_atomic_fetch_add(&p.tickets[opi], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == p.bit_ranges - 1u) ? 1u : 0u;
    if (s_last)
    {
      handoff_after_last_ticket(p.fenced);
      __hip_atomic_store(&p.tickets[opi], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0u)
  {
    unsigned classes = 0;
    for (unsigned r = 0; r < p.bit_ranges; ++r)
    {
      totals[p.bit_ranges + r] = classes; // (plain: read by the next launch)
      classes += rep_coherent_load(&totals[r]);
    }
    s_last = classes;
    if (keep) keep[words] = 1u;
  }
  __syncthreads();
  rep_store_count(p, o, kRepFlag | s_last, 0u);
  rep_arrive(p);
}

// Large tables, second step: ONE workgroup per op of the launch: wprefix[w] = set bits in the words before w, the
// class count = all set bits. Every op of the launch passes here (small tables and parents that stay uncompressed
// only to be counted: this is the kernel of such a launch that hands the counts over).
__global__ __launch_bounds__(kRepScanThreads) void k_rep_scan(const RepPack p)
{
  __shared__ unsigned wsum[kRepScanThreads / 64u];
  crepop_p o = (crepop_p)(uintptr_t)p.ops + blockIdx.x;
  const RepShape sh = rep_shape(p, o);
  if (sh.on && sh.ncells > kRepSmallCells)
  {
    const unsigned *__restrict__ bitmap = o->bitmap;
    unsigned *__restrict__ wprefix = o->bitmap + p.wstride;
    const unsigned words = (p.sites + 31u) / 32u;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0u)
    {
      // (this path keeps no bitmap to compare with: the node's maps count as other ones than before)
      *p.changed = p.sequence;
      if (o->keep) o->keep[words] = 0u;
    }
    unsigned carried = 0; // set bits in the chunks before
    // chunks of 1024 threads x 32 consecutive words (1M sites): eight 16-byte loads per thread, all in flight together
    // (the buffers are allocated in whole chunks, zero beyond the last site)
    for (unsigned chunk0 = 0; chunk0 < words; chunk0 += kRepScanChunk)
    {
      const unsigned w0 = chunk0 + threadIdx.x * 32u;
      uint4 v[8];
#pragma unroll
      for (unsigned q = 0; q < 8u; ++q) v[q] = w0 < words ? *reinterpret_cast<const uint4 *>(bitmap + w0 + 4u * q) : make_uint4(0u, 0u, 0u, 0u);
      unsigned n = 0;
#pragma unroll
      for (unsigned q = 0; q < 8u; ++q) n += __popc(v[q].x) + __popc(v[q].y) + __popc(v[q].z) + __popc(v[q].w);
      unsigned inc = n;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1)
      {
        const unsigned t = __shfl_up(inc, off, 64);
        if ((int)lane >= off) inc += t;
      }
      if (lane == 63u) wsum[wave] = inc;
      __syncthreads();
      unsigned before = 0, total = 0;
      for (unsigned w = 0; w < kRepScanThreads / 64u; ++w)
      {
        before += w < wave ? wsum[w] : 0u;
        total += wsum[w];
      }
      __syncthreads(); // wsum is reused by the next chunk
      unsigned r = carried + before + inc - n;
#pragma unroll
      for (unsigned q = 0; q < 8u; ++q)
      {
        uint4 pre;
        pre.x = r;
        r += __popc(v[q].x);
        pre.y = r;
        r += __popc(v[q].y);
        pre.z = r;
        r += __popc(v[q].z);
        pre.w = r;
        r += __popc(v[q].w);
        if (w0 < words) *reinterpret_cast<uint4 *>(wprefix + w0 + 4u * q) = pre;
      }
      carried += total;
    }
    rep_store_count(p, o, kRepFlag | carried, 0u);
  }
  rep_arrive(p);
}

// Large tables, third step: every non-empty cell takes its class number - the first sites before its own.
// Grid: rep_place with kRepRankTiles workgroups per op, which stride over the op's cells.
__global__ __launch_bounds__(kRepRankThreads) void k_rep_rank(const RepPack p)
{
  unsigned opi, tile;
  if (!rep_place(p.nops, kRepRankTiles, opi, tile)) return;
  crepop_p o = (crepop_p)(uintptr_t)p.ops + opi;
  const RepShape sh = rep_shape(p, o);
  if (!sh.on || sh.ncells <= kRepSmallCells) return;
  unsigned *__restrict__ table = o->table;
  const unsigned *__restrict__ bitmap = o->bitmap, *__restrict__ wprefix = o->bitmap + p.wstride;
  const unsigned *__restrict__ starts = o->bitmap + 2u * (size_t)p.wstride + p.bit_ranges; // k_rep_bits: classes before each range
  const unsigned ncells = sh.ncells;
  for (unsigned base = tile * kRepRankThreads * 4u; base < ncells; base += kRepRankTiles * kRepRankThreads * 4u)
  {
    unsigned v[4], pre[4], bits[4], start[4];
#pragma unroll
    for (unsigned q = 0; q < 4u; ++q)
    {
      const unsigned c = base + q * kRepRankThreads + threadIdx.x;
      v[q] = c < ncells ? table[c] : kRepEmpty;
    }
#pragma unroll
    for (unsigned q = 0; q < 4u; ++q)
    {
      const unsigned w = v[q] != kRepEmpty ? v[q] >> 5 : 0u;
      pre[q] = wprefix[w];
      bits[q] = bitmap[w];
      // (after k_rep_bits the running counts start anew in every range: + the classes of the ranges before)
      start[q] = p.bit_ranges ? starts[w / p.bit_words] : 0u;
    }
#pragma unroll
    for (unsigned q = 0; q < 4u; ++q)
    {
      const unsigned c = base + q * kRepRankThreads + threadIdx.x;
      if (c < ncells && v[q] != kRepEmpty) table[c] = start[q] + pre[q] + __popc(bits[q] & ((1u << (v[q] & 31u)) - 1u));
    }
  }
}

// site -> class. The table sits in LDS as 16-bit class numbers when it has at most 65536 cells (a class number is
// below the cell count) - looked up through global memory a scattered gather costs a cycle of the CU's address unit per
// lane (C4's level 2: 16M look-ups, 52 us; in LDS a few cycles per wave). FIRSTS (large tables): the sites the bitmap
// marks as the first of their cell also write the class -> first site / child entry maps - in site order, which is
// class order: neighbouring stores (the cells, which know the same, would scatter them).
template <bool L8, bool R8, bool P8, bool LDS, bool FIRSTS>
__device__ __forceinline__ void rep_assign_tile(const RepPack &p, crepop_p o, unsigned nleft, const unsigned short *lds, unsigned tile)
{
  const unsigned char *l8 = o->l8, *r8 = o->r8;
  const unsigned *l32 = o->l32, *r32 = o->r32;
  const unsigned *__restrict__ table = o->final;
  const unsigned sites = p.sites;
  const unsigned tile0 = tile * p.assign_iters * kRepAssignThreads * 16u;
  for (unsigned it = 0; it < p.assign_iters; ++it)
  {
    const unsigned s = tile0 + it * kRepAssignThreads * 16u + threadIdx.x * 16u;
    if (s >= sites) continue;
    unsigned l[16], r[16], v[16];
    rep_load16<L8>(l8, l32, s, l);
    rep_load16<R8>(r8, r32, s, r);
    unsigned firsts = 0;
    if (FIRSTS) firsts = (o->bitmap[s >> 5] >> (s & 16u)) & 0xFFFFu;
#pragma unroll
    for (unsigned e = 0; e < 16u; ++e)
    {
      const unsigned cell = s + e < sites ? l[e] + r[e] * nleft : 0u; // (the padding of the maps holds anything: cell 0, stored into the padding of the parent's)
      v[e] = LDS ? (unsigned)lds[cell] : table[cell];
    }
    if (P8)
    {
      uint4 out;
      out.x = (v[0] & 255u) | (v[1] & 255u) << 8 | (v[2] & 255u) << 16 | v[3] << 24;
      out.y = (v[4] & 255u) | (v[5] & 255u) << 8 | (v[6] & 255u) << 16 | v[7] << 24;
      out.z = (v[8] & 255u) | (v[9] & 255u) << 8 | (v[10] & 255u) << 16 | v[11] << 24;
      out.w = (v[12] & 255u) | (v[13] & 255u) << 8 | (v[14] & 255u) << 16 | v[15] << 24;
      *reinterpret_cast<uint4 *>(o->p8 + s) = out;
    }
    else
    {
#pragma unroll
      for (unsigned q = 0; q < 4u; ++q) *reinterpret_cast<uint4 *>(o->p32 + s + 4u * q) = make_uint4(v[4u * q], v[4u * q + 1u], v[4u * q + 2u], v[4u * q + 3u]);
    }
    if (FIRSTS && firsts)
    {
      unsigned *__restrict__ pids = o->pids, *__restrict__ lent = o->lent, *__restrict__ rent = o->rent;
#pragma unroll
      for (unsigned e = 0; e < 16u; ++e)
        if (firsts >> e & 1u)
        {
          pids[v[e]] = s + e;
          lent[v[e]] = l[e];
          rent[v[e]] = r[e];
        }
    }
  }
}

template <bool L8, bool R8, bool P8>
__device__ __forceinline__ void rep_assign_form(const RepPack &p, crepop_p o, unsigned nleft, unsigned ncells, unsigned short *lds, unsigned tile)
{
  if (ncells <= p.lds_cells)
  {
    const unsigned *__restrict__ table = o->final;
    // (an empty cell: no site looks it up; the slices of the arena start at multiples of four cells and are padded to one)
    for (unsigned i = threadIdx.x * 4u; i < ncells; i += kRepAssignThreads * 4u)
    {
      const uint4 t = *reinterpret_cast<const uint4 *>(table + i);
      *reinterpret_cast<uint2 *>(lds + i) = make_uint2((t.x & 0xFFFFu) | t.y << 16, (t.z & 0xFFFFu) | t.w << 16);
    }
    __syncthreads();
    if (ncells > kRepSmallCells) rep_assign_tile<L8, R8, P8, true, true>(p, o, nleft, lds, tile);
    else rep_assign_tile<L8, R8, P8, true, false>(p, o, nleft, lds, tile);
  }
  else
    rep_assign_tile<L8, R8, P8, false, true>(p, o, nleft, lds, tile);
}

// Grid: rep_place with one workgroup per assign_iters x 16384 sites
__global__ __launch_bounds__(kRepAssignThreads) void k_rep_assign(const RepPack p)
{
  extern __shared__ unsigned rep_lds[];
  unsigned short *lds = reinterpret_cast<unsigned short *>(rep_lds);
  unsigned opi, tile;
  if (!rep_place(p.nops, p.wgs, opi, tile)) return;
  crepop_p o = (crepop_p)(uintptr_t)p.ops + opi;
  const unsigned word = p.counts[o->slot];
  if (!(word & kRepFlag)) return; // not compressed: no maps
  if (o->flags & kRepDeferred) return; // its parent's k_rep_mark writes the map
  const RepShape sh = rep_shape(p, o);
  const unsigned form = (sh.nl <= kRepNarrow ? 4u : 0u) | (sh.nr <= kRepNarrow ? 2u : 0u) | ((word & ~kRepFlag) <= kRepNarrow ? 1u : 0u);
  switch (form)
  {
  case 7u: rep_assign_form<true, true, true>(p, o, sh.nl, sh.ncells, lds, tile); break;
  case 6u: rep_assign_form<true, true, false>(p, o, sh.nl, sh.ncells, lds, tile); break;
  case 5u: rep_assign_form<true, false, true>(p, o, sh.nl, sh.ncells, lds, tile); break;
  case 4u: rep_assign_form<true, false, false>(p, o, sh.nl, sh.ncells, lds, tile); break;
  case 3u: rep_assign_form<false, true, true>(p, o, sh.nl, sh.ncells, lds, tile); break;
  case 2u: rep_assign_form<false, true, false>(p, o, sh.nl, sh.ncells, lds, tile); break;
  case 1u: rep_assign_form<false, false, true>(p, o, sh.nl, sh.ncells, lds, tile); break;
  default: rep_assign_form<false, false, false>(p, o, sh.nl, sh.ncells, lds, tile); break;
  }
  if (sh.ncells > kRepSmallCells)
  {
    // the op's bitmap goes back to zero for the next launch that uses the slot: this tile's sites, once every thread
    // of the workgroup has read its bits
    __syncthreads();
    const unsigned tile_words = p.assign_iters * kRepAssignThreads / 2u; // 16 sites per thread and round
    for (unsigned w = threadIdx.x * 4u; w < tile_words; w += kRepAssignThreads * 4u)
      if (tile * tile_words + w < p.wstride) *reinterpret_cast<uint4 *>(o->bitmap + tile * tile_words + w) = make_uint4(0u, 0u, 0u, 0u);
  }
}

// the two forms of a site -> class map into each other (n = entries rounded up to whole groups of sixteen)
__global__ __launch_bounds__(256) void k_rep_widen(const unsigned char *__restrict__ m8, unsigned *__restrict__ m32, unsigned n)
{
  const unsigned s = (blockIdx.x * 256u + threadIdx.x) * 16u;
  if (s >= n) return;
  unsigned v[16];
  rep_load16<true>(m8, nullptr, s, v);
#pragma unroll
  for (unsigned q = 0; q < 4u; ++q) *reinterpret_cast<uint4 *>(m32 + s + 4u * q) = make_uint4(v[4u * q], v[4u * q + 1u], v[4u * q + 2u], v[4u * q + 3u]);
}

__global__ __launch_bounds__(256) void k_rep_narrow(const unsigned *__restrict__ m32, unsigned char *__restrict__ m8, unsigned n)
{
  const unsigned s = (blockIdx.x * 256u + threadIdx.x) * 16u;
  if (s >= n) return;
  unsigned v[16];
  rep_load16<false>(nullptr, m32, s, v);
  uint4 out;
  out.x = (v[0] & 255u) | (v[1] & 255u) << 8 | (v[2] & 255u) << 16 | v[3] << 24;
  out.y = (v[4] & 255u) | (v[5] & 255u) << 8 | (v[6] & 255u) << 16 | v[7] << 24;
  out.z = (v[8] & 255u) | (v[9] & 255u) << 8 | (v[10] & 255u) << 16 | v[11] << 24;
  out.w = (v[12] & 255u) | (v[13] & 255u) << 8 | (v[14] & 255u) << 16 | v[15] << 24;
  *reinterpret_cast<uint4 *>(m8 + s) = out;
}
