// kernels_repeats.h - site-repeats class maps on the device (SURVEY.md section 8 row f4;
// src/repeats.c:299-382, pll_update_repeats).
//
// A parent's classes are the distinct pairs (left class, right class) of its sites, numbered in
// order of first occurrence; id_site[class] is that first site. The reference walks the sites
// sequentially through a direct-address table (cell = lid + rid * ids_left). The same numbering
// without the sequential walk:
//   1. table[cell] = min over the sites that map to the cell            (k_rep_mark, atomicMin)
//   2. a site is a class representative iff it is the lowest site of its cell; the class number of a
//      representative is the count of representatives before it. Counted over the CELLS, not the sites
//      (round 3: the two passes over all sites this step used to be were 40 % of the update): the
//      representatives are set as bits of a bitmap over the sites (k_rep_bitmap, one atomicOr per non-empty
//      cell), one workgroup per op forms the running bit count per 32-site word (k_rep_scan), and every
//      non-empty cell looks its class number up: words before + bits before in its word (k_rep_rank_cells).
//      The cell index IS the pair (left class, right class), so the class -> child entry maps need no
//      site data either.
//   3. site_id[site] = class number of table[cell(site)]                (k_rep_assign)
// Integer work only: the maps are bit-identical to the reference's. All ops of one dependency level
// go through each kernel together (grid.y = op); every op owns a slice of the table, cleared with
// one memset before the level instead of the reference's to-clean list.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_common.h"

constexpr int kRepOps = 128;              // ops per batch (descriptors in device memory, class counts in mapped host memory)
constexpr unsigned kRepBlock = 1024;      // sites (k_rep_assign) or cells (k_rep_bitmap, k_rep_rank_cells) per workgroup: 256 threads x 4
constexpr unsigned kRepMarkSites = 4096;  // sites per workgroup of k_rep_mark: 256 threads x 4 groups of 4 consecutive sites
constexpr unsigned kRepLdsCells = 8192;   // table slices up to this many cells are reduced in LDS first (32 KB)
constexpr unsigned kRepFilterBits = 12, kRepFilter = 1u << kRepFilterBits; // entries of k_rep_mark's LDS filter (8 bytes each)
constexpr unsigned kRepClassFlag = 0x80000000u; // a table cell that holds its class number instead of its first site

struct RepOp
{
  const unsigned *lid;   // site -> class of the left child  [sites]
  const unsigned *rid;
  unsigned *psid;        // out: site -> class of the parent   [sites]
  unsigned *pids;        // out: class -> first site           [<= sites]
  unsigned *lent, *rent; // out: class -> entry of the left / right child (what the gather kernels want)
  unsigned *bitmap;      // scratch [words]: bit s = site s is the lowest site of its cell (zero before k_rep_bitmap)
  unsigned *wprefix;     // scratch [words]: representatives in the words before
  unsigned nleft;        // classes of the left child
  unsigned ncells;       // nleft * classes of the right child: this op's table slice
  unsigned tab_off;      // first cell of the slice
  unsigned pad;
};
typedef const RepOp __attribute__((address_space(4))) *crepop_p;

struct RepPack
{
  const RepOp *ops;      // device array [nops]
  unsigned *table;
  unsigned *counts;      // out [nops]: classes per op
  unsigned *host_counts; // the same in host-mapped memory, followed by ...
  unsigned *host_seq;    // ... the sequence word the host polls (written last)
  unsigned *ticket;      // arrival counter of the ops' last workgroups (0 between calls)
  unsigned sequence;
  unsigned sites;
  unsigned words;        // (sites + 31) / 32; the bitmap / prefix buffers hold `wstride` >= words words per op, a multiple of 32768
  int fenced;            // kernels_common.h: handoff_*
};

__device__ __forceinline__ crepop_p rep_op(const RepPack &p)
{
  return (crepop_p)(uintptr_t)p.ops + blockIdx.y;
}

// Step 1: table[cell] = the lowest site of the cell. Many sites share a cell - that is the point of site
// repeats; near the tips ALL of them share a handful (a DNA cherry: 16 cells for every site of the alignment),
// and one atomic per site on a handful of L2 lines is a queue (C4's shard: 0.2-0.35 ms per level, 80 % of the
// whole class-map update). So a workgroup first reduces its 4096 sites in LDS - a read before the LDS atomic:
// after the first 256 sites nearly every later (higher) site finds a lower one in its cell and moves on - and
// then sends ONE candidate per cell it touched, again after a look at what is there already. Slices too large
// for LDS (deep nodes: many cells, few sites each) keep the direct form, where contention is no issue.
__global__ __launch_bounds__(256) void k_rep_mark(const RepPack p)
{
  extern __shared__ unsigned rep_lds[];
  crepop_p o = rep_op(p);
  const unsigned *__restrict__ lid = o->lid, *__restrict__ rid = o->rid;
  const unsigned nleft = o->nleft, ncells = o->ncells;
  unsigned *table = p.table + o->tab_off;
  const bool in_lds = ncells <= kRepLdsCells; // workgroup-uniform
  // slices too large for LDS get a FILTER there instead: kRepFilter entries {cell, lowest site of this workgroup seen with
  // it}, direct-mapped by a hash of the cell. A site whose cell sits in its slot with a lower site has nothing to tell
  // the table (that lower site's thread does, or was itself told so); anything else - another cell in the slot, a race -
  // just goes to the table as before. What this path costs is its atomics on the table, and in a pattern-sorted
  // alignment the sites of a class come in clusters.
  unsigned long long *filter = reinterpret_cast<unsigned long long *>(rep_lds);
  if (in_lds)
    for (unsigned i = threadIdx.x; i < ncells; i += 256u) rep_lds[i] = 0xFFFFFFFFu;
  else
    for (unsigned i = threadIdx.x; i < kRepFilter; i += 256u) filter[i] = ~0ull;
  __syncthreads();
  // a thread takes four CONSECUTIVE sites at a time (16-byte loads of the two maps; round 2 read them 4 bytes per lane
  // and the kernel ran at 1 TB/s), four such groups 1024 sites apart, all requested before the first is looked at.
  // Consecutive sites of a pattern-sorted alignment mostly share their cell: only the first of a run goes to the table.
  const unsigned wg0 = blockIdx.x * kRepMarkSites;
  uint4 l[4], r[4];
#pragma unroll
  for (unsigned q = 0; q < 4; ++q)
  {
    const unsigned s = wg0 + q * 1024u + threadIdx.x * 4u;
    if (s + 3u < p.sites)
    {
      l[q] = *reinterpret_cast<const uint4 *>(lid + s);
      r[q] = *reinterpret_cast<const uint4 *>(rid + s);
    }
    else
    {
      unsigned lv[4] = {0, 0, 0, 0}, rv[4] = {0, 0, 0, 0};
      for (unsigned e = 0; e < 4; ++e)
        if (s + e < p.sites)
        {
          lv[e] = lid[s + e];
          rv[e] = rid[s + e];
        }
      l[q] = make_uint4(lv[0], lv[1], lv[2], lv[3]);
      r[q] = make_uint4(rv[0], rv[1], rv[2], rv[3]);
    }
  }
#pragma unroll
  for (unsigned q = 0; q < 4; ++q) // ascending sites: later groups mostly find a lower site already there
  {
    const unsigned s = wg0 + q * 1024u + threadIdx.x * 4u;
    const unsigned c[4] = {l[q].x + r[q].x * nleft, l[q].y + r[q].y * nleft, l[q].z + r[q].z * nleft, l[q].w + r[q].w * nleft};
#pragma unroll
    for (unsigned e = 0; e < 4; ++e)
    {
      if (s + e >= p.sites || (e && c[e] == c[e - 1])) continue; // (a lower site of this thread has the cell)
      if (in_lds)
      {
        if (rep_lds[c[e]] > s + e) atomicMin(&rep_lds[c[e]], s + e);
      }
      // slices too large for LDS. A PLAIN (cached) look: cells only ever go down, so a stale value can cost a redundant
      // atomic, never a wrong minimum (an agent-scope load per site was 16M trips to the coherent level per level of
      // C4). One look, then its atomic, site after site: with all looks first a thread's own lower sites no longer
      // shield the later ones and the atomics - what this path costs - tripled (measured: 0.37 -> 1.9 ms).
      else
      {
        const unsigned slot = (c[e] * 2654435761u) >> (32 - kRepFilterBits);
        const unsigned long long ent = filter[slot];
        if ((unsigned)(ent >> 32) == c[e] && (unsigned)ent < s + e) continue;
        filter[slot] = ((unsigned long long)c[e] << 32) | (s + e);
        if (table[c[e]] > s + e) atomicMin(&table[c[e]], s + e);
      }
    }
  }
  if (!in_lds) return;
  __syncthreads();
  for (unsigned i = threadIdx.x; i < ncells; i += 256u)
  {
    const unsigned v = rep_lds[i];
    // a stale value read here can only cause a redundant atomic, never a wrong minimum
    if (v != 0xFFFFFFFFu && __hip_atomic_load(&table[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > v) atomicMin(&table[i], v);
  }
}

// Step 2a: the representatives as a bitmap over the sites
__global__ __launch_bounds__(256) void k_rep_bitmap(const RepPack p)
{
  crepop_p o = rep_op(p);
  const unsigned ncells = o->ncells;
  const unsigned *__restrict__ table = p.table + o->tab_off;
  unsigned *__restrict__ bitmap = o->bitmap;
  const unsigned base = blockIdx.x * kRepBlock + threadIdx.x * 4u;
  if (base >= ncells) return;
#pragma unroll
  for (unsigned q = 0; q < 4; ++q)
  {
    const unsigned cell = base + q;
    if (cell < ncells)
    {
      const unsigned s = table[cell];
      if (s != 0xFFFFFFFFu) atomicOr(&bitmap[s >> 5], 1u << (s & 31u));
    }
  }
}

// Step 2b: ONE workgroup per op: wprefix[w] = set bits in the words before w; the op's class count goes to the device
// array and straight to the host (mapped memory); the op that arrives last publishes the sequence word (hand-off
// without fences by default: kernels_common.h)
__global__ __launch_bounds__(1024) void k_rep_scan(const RepPack p)
{
  __shared__ unsigned wsum[16];
  crepop_p o = rep_op(p);
  const unsigned *__restrict__ bitmap = o->bitmap;
  unsigned *__restrict__ wprefix = o->wprefix;
  const unsigned words = p.words;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned carried = 0; // set bits in the chunks before
  // chunks of 1024 threads x 32 consecutive words (1M sites): eight 16-byte loads per thread, all in flight together
  // (the buffers are allocated in whole chunks: pllgpu_repeats_classes)
  for (unsigned chunk0 = 0; chunk0 < words; chunk0 += 32768u)
  {
    const unsigned w0 = chunk0 + threadIdx.x * 32u;
    uint4 v[8];
#pragma unroll
    for (unsigned q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const uint4 *>(bitmap + w0 + 4u * q);
    unsigned n = 0;
#pragma unroll
    for (unsigned q = 0; q < 8; ++q) n += __popc(v[q].x) + __popc(v[q].y) + __popc(v[q].z) + __popc(v[q].w);
    unsigned inc = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
      const unsigned t = __shfl_up(inc, off, 64);
      if ((int)lane >= off) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned before = 0, total = 0;
    for (unsigned w = 0; w < 16u; ++w)
    {
      before += w < wave ? wsum[w] : 0u;
      total += wsum[w];
    }
    __syncthreads(); // wsum is reused by the next chunk
    unsigned r = carried + before + inc - n;
#pragma unroll
    for (unsigned q = 0; q < 8; ++q)
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
      *reinterpret_cast<uint4 *>(wprefix + w0 + 4u * q) = pre;
    }
    carried += total;
  }
  if (threadIdx.x == 0u)
  {
    p.counts[blockIdx.y] = carried;
    __hip_atomic_store(&p.host_counts[blockIdx.y], carried, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    handoff_before_sequence(p.fenced);
    const unsigned t = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.y - 1u)
    {
      __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.host_seq, p.sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Step 2c: every non-empty cell takes its class number - the representatives before its lowest site - and keeps it
// (flagged: a flagged word can never equal a site), so that step 3 finds the class of any site with one look-up; the
// class -> first site / child entry maps are written here: the cell index is left class + right class * nleft
__global__ __launch_bounds__(256) void k_rep_rank_cells(const RepPack p)
{
  crepop_p o = rep_op(p);
  const unsigned ncells = o->ncells, nleft = o->nleft;
  unsigned *__restrict__ table = p.table + o->tab_off;
  const unsigned *__restrict__ bitmap = o->bitmap, *__restrict__ wprefix = o->wprefix;
  unsigned *__restrict__ pids = o->pids, *__restrict__ lent = o->lent, *__restrict__ rent = o->rent;
  const unsigned base = blockIdx.x * kRepBlock + threadIdx.x * 4u;
  if (base >= ncells) return;
#pragma unroll
  for (unsigned q = 0; q < 4; ++q)
  {
    const unsigned cell = base + q;
    if (cell < ncells)
    {
      const unsigned s = table[cell];
      if (s != 0xFFFFFFFFu)
      {
        const unsigned w = s >> 5;
        const unsigned r = wprefix[w] + __popc(bitmap[w] & ((1u << (s & 31u)) - 1u));
        table[cell] = r | kRepClassFlag;
        pids[r] = s;
        lent[r] = cell % nleft;
        rent[r] = cell / nleft;
      }
    }
  }
}

// Step 3: site -> class
__global__ __launch_bounds__(256) void k_rep_assign(const RepPack p)
{
  crepop_p o = rep_op(p);
  const unsigned *__restrict__ lid = o->lid, *__restrict__ rid = o->rid;
  const unsigned nleft = o->nleft;
  const unsigned *__restrict__ table = p.table + o->tab_off;
  unsigned *__restrict__ psid = o->psid;
  const unsigned base = blockIdx.x * kRepBlock + threadIdx.x * 4u;
  if (base + 3u < p.sites)
  {
    const uint4 l = *reinterpret_cast<const uint4 *>(lid + base), r = *reinterpret_cast<const uint4 *>(rid + base);
    uint4 v;
    v.x = table[l.x + r.x * nleft] & ~kRepClassFlag;
    v.y = table[l.y + r.y * nleft] & ~kRepClassFlag;
    v.z = table[l.z + r.z * nleft] & ~kRepClassFlag;
    v.w = table[l.w + r.w * nleft] & ~kRepClassFlag;
    *reinterpret_cast<uint4 *>(psid + base) = v;
    return;
  }
  for (unsigned q = 0; q < 4; ++q)
  {
    const unsigned s = base + q;
    if (s < p.sites) psid[s] = table[lid[s] + rid[s] * nleft] & ~kRepClassFlag;
  }
}
