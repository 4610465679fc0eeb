// kernels_repeats.h - site-repeats class maps on the device (SURVEY.md section 8 row f4;
// src/repeats.c:299-382, pll_update_repeats).
//
// A parent's classes are the distinct pairs (left class, right class) of its sites, numbered in
// order of first occurrence; id_site[class] is that first site. The reference walks the sites
// sequentially through a direct-address table (cell = lid + rid * ids_left). The same numbering
// without the sequential walk:
//   1. table[cell] = min over the sites that map to the cell            (k_rep_mark, atomicMin)
//   2. a site is a class representative iff table[cell(site)] == site; the class number of a
//      representative is the count of representatives before it         (k_rep_count: per workgroup;
//                                                                        k_rep_rank: prefix over the
//                                                                        workgroups before it + own)
//   3. site_id[site] = class number of table[cell(site)]                (k_rep_assign)
// Integer work only: the maps are bit-identical to the reference's. All ops of one dependency level
// go through each kernel together (grid.y = op); every op owns a slice of the table, cleared with
// one memset before the level instead of the reference's to-clean list.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_common.h"

constexpr int kRepOps = 128;              // ops per batch (descriptors in device memory, class counts in mapped host memory)
constexpr unsigned kRepBlock = 1024;      // sites per workgroup of the count / rank / assign kernels: 256 threads x 4 consecutive sites
constexpr unsigned kRepMarkSites = 4096;  // sites per workgroup of k_rep_mark: 256 threads x 16, strided
constexpr unsigned kRepLdsCells = 8192;   // table slices up to this many cells are reduced in LDS first (32 KB)
constexpr unsigned kRepClassFlag = 0x80000000u; // a table cell that holds its class number instead of its first site

struct RepOp
{
  const unsigned *lid;   // site -> class of the left child  [sites]
  const unsigned *rid;
  unsigned *psid;        // out: site -> class of the parent   [sites]
  unsigned *pids;        // out: class -> first site           [<= sites]
  unsigned *lent, *rent; // out: class -> entry of the left / right child (what the gather kernels want)
  unsigned *blocksum;    // scratch [nblk]: representatives per workgroup
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
  unsigned nblk;
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
  unsigned *__restrict__ table = p.table + o->tab_off;
  const unsigned base = blockIdx.x * kRepMarkSites + threadIdx.x;
  if (ncells <= kRepLdsCells) // workgroup-uniform
  {
    for (unsigned i = threadIdx.x; i < ncells; i += 256u) rep_lds[i] = 0xFFFFFFFFu;
    __syncthreads();
#pragma unroll 4
    for (unsigned q = 0; q < kRepMarkSites / 256u; ++q)
    {
      const unsigned s = base + q * 256u; // ascending in q: later rounds mostly find a lower site already there
      if (s < p.sites)
      {
        const unsigned c = lid[s] + rid[s] * nleft;
        if (rep_lds[c] > s) atomicMin(&rep_lds[c], s);
      }
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < ncells; i += 256u)
    {
      const unsigned v = rep_lds[i];
      // a stale value read here can only cause a redundant atomic, never a wrong minimum
      if (v != 0xFFFFFFFFu && __hip_atomic_load(&table[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > v) atomicMin(&table[i], v);
    }
    return;
  }
#pragma unroll 4
  for (unsigned q = 0; q < kRepMarkSites / 256u; ++q)
  {
    const unsigned s = base + q * 256u;
    if (s < p.sites)
    {
      const unsigned c = lid[s] + rid[s] * nleft;
      if (__hip_atomic_load(&table[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > s) atomicMin(&table[c], s);
    }
  }
}

// inclusive scan of one value per thread over the 256 threads of a workgroup; returns the
// exclusive prefix of the calling thread and, in `total`, the workgroup's sum
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned &total)
{
  __shared__ unsigned wsum[4];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
  {
    const unsigned t = __shfl_up(inc, off, 64);
    if ((int)lane >= off) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  unsigned before = 0;
  for (unsigned w = 0; w < wave; ++w) before += wsum[w];
  total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads(); // wsum is reused by the next call
  return before + inc - v;
}

// Step 2a: representatives (sites that are the minimum of their cell) per workgroup
__global__ __launch_bounds__(256) void k_rep_count(const RepPack p)
{
  crepop_p o = rep_op(p);
  const unsigned *__restrict__ lid = o->lid, *__restrict__ rid = o->rid;
  const unsigned nleft = o->nleft;
  const unsigned *__restrict__ table = p.table + o->tab_off;
  const unsigned base = blockIdx.x * kRepBlock + threadIdx.x * 4u;
  unsigned n = 0;
  if (base + 3u < p.sites)
  {
    const uint4 l = *reinterpret_cast<const uint4 *>(lid + base), r = *reinterpret_cast<const uint4 *>(rid + base);
    n = (table[l.x + r.x * nleft] == base ? 1u : 0u) + (table[l.y + r.y * nleft] == base + 1u ? 1u : 0u) +
        (table[l.z + r.z * nleft] == base + 2u ? 1u : 0u) + (table[l.w + r.w * nleft] == base + 3u ? 1u : 0u);
  }
  else
  {
    for (unsigned q = 0; q < 4; ++q)
    {
      const unsigned s = base + q;
      if (s < p.sites && table[lid[s] + rid[s] * nleft] == s) ++n;
    }
  }
  unsigned total;
  (void)block_exclusive_scan(n, total);
  if (threadIdx.x == 0) o->blocksum[blockIdx.x] = total;
}

// Step 2b: the class number of a representative = the representatives before it. It goes INTO the table cell
// (flagged: a flagged word can never equal a site), so that step 3 finds the class of any site with one look-up;
// the class -> first site / child entry maps are written here
__global__ __launch_bounds__(256) void k_rep_rank(const RepPack p)
{
  crepop_p o = rep_op(p);
  const unsigned *__restrict__ lid = o->lid, *__restrict__ rid = o->rid;
  const unsigned nleft = o->nleft;
  unsigned *__restrict__ table = p.table + o->tab_off;
  const unsigned *__restrict__ blocksum = o->blocksum;
  const unsigned base = blockIdx.x * kRepBlock + threadIdx.x * 4u;
  bool rep[4];
  unsigned cell[4], le[4], re[4];
  unsigned n = 0;
#pragma unroll
  for (unsigned q = 0; q < 4; ++q)
  {
    const unsigned s = base + q;
    rep[q] = false;
    cell[q] = le[q] = re[q] = 0u;
    if (s < p.sites)
    {
      le[q] = lid[s];
      re[q] = rid[s];
      cell[q] = le[q] + re[q] * nleft;
      rep[q] = table[cell[q]] == s;
    }
    n += rep[q] ? 1u : 0u;
  }
  // representatives in the workgroups before this one (nblk is a few hundred: every workgroup adds them up itself)
  __shared__ unsigned before_ws[4];
  unsigned before = 0;
  for (unsigned b = threadIdx.x; b < blockIdx.x; b += 256u) before += blocksum[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
  if ((threadIdx.x & 63u) == 0u) before_ws[threadIdx.x >> 6] = before;
  __syncthreads();
  before = before_ws[0] + before_ws[1] + before_ws[2] + before_ws[3];
  unsigned total;
  unsigned r = before + block_exclusive_scan(n, total);
  if (blockIdx.x == p.nblk - 1u && threadIdx.x == 0u)
  {
    // the op's class count goes to the device array and straight to the host (mapped memory); the op
    // that arrives last publishes the sequence word (hand-off without fences by default: kernels_common.h)
    const unsigned cnt = before + total;
    p.counts[blockIdx.y] = cnt;
    __hip_atomic_store(&p.host_counts[blockIdx.y], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    handoff_before_sequence(p.fenced);
    const unsigned t = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.y - 1u)
    {
      __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.host_seq, p.sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  unsigned *__restrict__ pids = o->pids, *__restrict__ lent = o->lent, *__restrict__ rent = o->rent;
#pragma unroll
  for (unsigned q = 0; q < 4; ++q)
    if (rep[q])
    {
      table[cell[q]] = r | kRepClassFlag; // only this thread ever tests this cell against this site
      pids[r] = base + q;
      lent[r] = le[q];
      rent[r] = re[q];
      ++r;
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
