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

constexpr int kRepOps = 32;           // ops per launch
constexpr unsigned kRepBlock = 1024;  // sites per workgroup: 256 threads x 4 consecutive sites

struct RepOp
{
  const unsigned *lid;   // site -> class of the left child  [sites]
  const unsigned *rid;
  unsigned *psid;        // out: site -> class of the parent   [sites]
  unsigned *pids;        // out: class -> first site           [<= sites]
  unsigned *lent, *rent; // out: class -> entry of the left / right child (what the gather kernels want)
  unsigned *rank;        // scratch [sites]: class number of a representative site
  unsigned *blocksum;    // scratch [nblk]: representatives per workgroup
  unsigned nleft;        // classes of the left child
  unsigned tab_off;      // first cell of this op's table slice
};

struct RepPack
{
  RepOp ops[kRepOps];
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

__device__ __forceinline__ unsigned rep_cell(const RepOp &o, unsigned s)
{
  return o.tab_off + o.lid[s] + o.rid[s] * o.nleft;
}

__global__ __launch_bounds__(256) void k_rep_mark(const RepPack p)
{
  const RepOp &o = p.ops[blockIdx.y];
  const unsigned base = blockIdx.x * kRepBlock + threadIdx.x * 4u;
#pragma unroll
  for (unsigned q = 0; q < 4; ++q)
  {
    const unsigned s = base + q;
    // many sites share a cell (that is the point of site repeats): (1) look before the atomic, so
    // that only sites that could still lower the minimum queue up on the cell's L2 line - a stale
    // value read here can only cause a redundant atomic, never a wrong minimum; (2) of the lanes of
    // a wave that want the same cell only the lowest one (= the lowest site) goes out.
    const unsigned c = s < p.sites ? rep_cell(o, s) : 0u;
    bool pending = s < p.sites && __hip_atomic_load(&p.table[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > s;
    unsigned long long m;
    while ((m = __ballot(pending)) != 0ull)
    {
      const int leader = __ffsll((long long)m) - 1;
      const unsigned lc = __shfl(c, leader, 64);
      if (pending && c == lc)
      {
        if ((int)(threadIdx.x & 63u) == leader) atomicMin(&p.table[c], s);
        pending = false;
      }
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

__global__ __launch_bounds__(256) void k_rep_count(const RepPack p)
{
  const RepOp &o = p.ops[blockIdx.y];
  const unsigned base = blockIdx.x * kRepBlock + threadIdx.x * 4u;
  unsigned n = 0;
#pragma unroll
  for (unsigned q = 0; q < 4; ++q)
  {
    const unsigned s = base + q;
    if (s < p.sites && p.table[rep_cell(o, s)] == s) ++n;
  }
  unsigned total;
  (void)block_exclusive_scan(n, total);
  if (threadIdx.x == 0) o.blocksum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_rep_rank(const RepPack p)
{
  const RepOp &o = p.ops[blockIdx.y];
  const unsigned base = blockIdx.x * kRepBlock + threadIdx.x * 4u;
  bool rep[4];
  unsigned n = 0;
#pragma unroll
  for (unsigned q = 0; q < 4; ++q)
  {
    const unsigned s = base + q;
    rep[q] = s < p.sites && p.table[rep_cell(o, s)] == s;
    n += rep[q] ? 1u : 0u;
  }
  // representatives in the workgroups before this one (nblk is a few hundred: every workgroup adds them up itself)
  __shared__ unsigned before_ws[4];
  unsigned before = 0;
  for (unsigned b = threadIdx.x; b < blockIdx.x; b += 256u) before += o.blocksum[b];
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
    // that arrives last publishes the sequence word (hand-off without fences: kernels_common.h)
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
#pragma unroll
  for (unsigned q = 0; q < 4; ++q)
    if (rep[q])
    {
      o.rank[base + q] = r;
      o.pids[r] = base + q;
      o.lent[r] = o.lid[base + q];
      o.rent[r] = o.rid[base + q];
      ++r;
    }
}

__global__ __launch_bounds__(256) void k_rep_assign(const RepPack p)
{
  const RepOp &o = p.ops[blockIdx.y];
  const unsigned base = blockIdx.x * kRepBlock + threadIdx.x * 4u;
#pragma unroll
  for (unsigned q = 0; q < 4; ++q)
  {
    const unsigned s = base + q;
    if (s < p.sites) o.psid[s] = o.rank[p.table[rep_cell(o, s)]];
  }
}
