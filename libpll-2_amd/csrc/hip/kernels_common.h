// kernels_common.h - device-side structures shared by the gfx950 kernels.
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   CLV      tiled sites-contiguous for every shape: [tile][rate][state][64 lanes], tile =
//            entry >> 6, lane = entry & 63 (kernels_generic.h); state padding dropped. Exception:
//            class-compressed nodes of a 4 x 4 partition are entry-contiguous [entry][16]
//            (kernels_dna.h: scattered entries are read whole).
//   scaler   [entry] or [entry][rate] unsigned (src/pll.c:838-857)
//   P matrix per branch, TRANSPOSED relative to the reference: PT[rate][col j][row i padded to
//            SPT]; a kernel walking the contraction index j then finds the ICH parent-state
//            coefficients it needs contiguous at a wave-uniform address (scalar loads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PLLGPU_SCALE_FACTOR 0x1p256
#define PLLGPU_SCALE_THRESHOLD 0x1p-256
#define PLLGPU_LOG_THRESHOLD (-177.445678223345993274)  /* log(2^-256) = -256 ln 2 */
#define PLLGPU_RATE_MAXDIFF 4u

constexpr int kMaxOpsPerLaunch = 32;
constexpr int kMaxRates = 64;

// one CLV update, device pointers resolved on the host side of the HIP TU
struct DevOp
{
  double *parent;
  const double *left;          // null when the child is a tip given by codes
  const double *right;
  const unsigned char *ltip;   // tip codes or null
  const unsigned char *rtip;
  unsigned *pscaler;           // null = no scaling for this op
  const unsigned *lscaler;
  const unsigned *rscaler;
  const double *lmat;          // PT layout
  const double *rmat;
  const unsigned *id_site;     // parent entry -> representative site, or null
  const unsigned *lsid;        // site -> left entry, or null; with kDirectMaps: PARENT ENTRY -> left entry
  const unsigned *rsid;
  unsigned entries;
  unsigned layout;             // kDirectMaps; 4x4 kernels also kAosLeft | kAosRight | kAosParent (entry-contiguous CLVs of compressed nodes)
};

// DevOp::layout bit: lsid / rsid are indexed by the parent's entry (filled by the class kernels,
// kernels_repeats.h) - two coalesced loads instead of the dependent id_site -> site_id chain
constexpr unsigned kDirectMaps = 8u;

// child entries of parent entry nn for a gathering op
__device__ __forceinline__ void gather_entries(const DevOp &op, unsigned nn, unsigned &le, unsigned &re)
{
  if (op.layout & kDirectMaps)
  {
    le = op.lsid[nn];
    re = op.rsid[nn];
    return;
  }
  const unsigned site = op.id_site ? op.id_site[nn] : nn;
  le = op.lsid ? op.lsid[site] : site;
  re = op.rsid ? op.rsid[site] : site;
}

// passed BY VALUE as a kernel argument: descriptors arrive through the kernarg segment (scalar
// loads, no staging copy, graph-capturable). 32 * 112 B = 3584 B < 4 KiB kernarg limit.
struct OpPack
{
  DevOp ops[kMaxOpsPerLaunch];
};

struct GenGeo
{
  unsigned S, SP, R;  // states, host states_padded (frequencies / host mirror stride), rate categories
  unsigned SPT;       // padded row count of PT = nchunks * ICH
  unsigned nchunks;   // parent-state chunks of ICH
  unsigned tile_sz;   // doubles per 64-entry tile of a tiled CLV = R * S * 64
  int scale_mode;     // 0 none, 1 per site, 2 per rate
};

// ---- XCD-aware workgroup order -----------------------------------------------------------------------------------
// Workgroups go to the eight XCDs round-robin by linear id (workgroup b runs on XCD b % 8), so with the natural mapping
// eight neighbouring pieces of a CLV are written through eight different L2s. tools/store_probe2.hip (round 4,
// profiles/r4_store_probe2.txt): a pure store stream over 0.7 ... 4 GB sustains 5.5-5.75 TB/s that way and 6.3-6.4 TB/s
// when every XCD writes its own CONTIGUOUS eighth of the work - whatever the store's width or cache-policy bits; read
// streams do not care (7.25 TB/s either way), 2 reads + 1 write hardly; runs of 16 workgroups (512 KB) per XCD already
// give all of it. The launches that are store traffic and nothing else - the groups fed from tip codes:
// k_partials_dna_cc, the tip-fed kinds of k_partials_dna_fused, k_partials_mfma_cc - therefore run as a 1-D grid of
// 8 x `per` workgroups and take their logical block from here: XCD x owns logical blocks [x per, (x + 1) per), walked in
// dispatch order. Measured on the launches themselves (same box, alternating): C2's seven-op launch 113-115 -> 108-110 us
// (0.82 -> 0.86 of the HBM peak), at 400k sites 518 -> 498 us; C3's group launch 332 -> 286-327 us. NOT used where it did
// not pay: chains (random trees 0.206 -> 0.217 ms per step: their siblings' reads), plain tip x tip levels of 20 and 61
// states (within the spread). `on` = 0 keeps the natural order on the same grid (PLL_AMD_NO_XCD_ORDER=1: the A/B switch).
__device__ __forceinline__ unsigned xcd_linear(unsigned total, unsigned on) // logical block number, or ~0u for the grid's padding
{
  const unsigned per = gridDim.x >> 3;
  const unsigned l = on ? (blockIdx.x & 7u) * per + (blockIdx.x >> 3) : blockIdx.x;
  return l < total ? l : ~0u; // (the grid is rounded up to a multiple of eight)
}
__device__ __forceinline__ bool xcd_block(unsigned nx, unsigned ny, unsigned on, unsigned &bx, unsigned &by)
{
  const unsigned l = xcd_linear(nx * ny, on);
  if (l == ~0u) return false;
  by = l / nx;
  bx = l - by * nx;
  return true;
}
static inline dim3 xcd_grid(unsigned nx, unsigned ny, unsigned nz = 1) { return dim3((nx * ny * nz + 7u) / 8u * 8u); }

// Workgroup barrier for an exchange through LDS in the middle of a kernel's loop: wait for the wave's own LDS
// accesses, then s_barrier. __syncthreads() carries workgroup-scope release / acquire semantics for GLOBAL memory
// too, i.e. s_waitcnt vmcnt(0): every such barrier drained the wave's prefetched loads and its stores in flight -
// three per item in k_partials_lean3 made an item 19 us long around 2 us of MFMAs.
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// one edge / root evaluation
struct DevEdge
{
  const double *parent;        // CLV end (always present)
  const double *child;         // CLV, or null (tip codes / root)
  const unsigned char *ctip;
  const unsigned *pscaler;
  const unsigned *cscaler;
  const double *mat;           // PT layout; unused for root
  const unsigned *psid;        // site -> entry maps or null
  const unsigned *csid;
  const double *freqs;         // [rate_matrices][SP]
  const double *rate_weights;  // [R]
  const double *prop_invar;    // [rate_matrices]
  const unsigned *pattern_weights;
  const int *invariant;        // or null
  double *persite;             // or null
  double *block_sums;          // [gridDim.x]
  unsigned *counter;           // arrival ticket of the workgroups (0 between calls)
  double *result;              // host-mapped: [0] the total, [1] sequence number of the call
  double sequence;             // written to result[1] after result[0] (host polls it)
  unsigned sites;
  int per_rate;
  int is_root;
  unsigned layout;               // 4x4 kernels: kAosParent | kAosLeft (= child) for entry-contiguous CLVs
  int fenced;                    // PLL_AMD_FENCED_HANDOFF=1: the result hand-off with release / acquire fences (diagnosis)
  unsigned char fidx[kMaxRates]; // freqs_indices
};

// force a scalar (s_load) fetch for a wave-uniform read-only address: constant address space
typedef const double __attribute__((address_space(4))) *cdouble_p;
__device__ __forceinline__ cdouble_p as_const(const double *p)
{
  return (cdouble_p)(uintptr_t)p;
}

// quad (4 adjacent lanes) exchanges through DPP: no LDS traffic
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ double dpp_f64_xor1(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(dpp_i32<0xB1>(hi), dpp_i32<0xB1>(lo)); // quad_perm [1,0,3,2]
}
__device__ __forceinline__ double dpp_f64_xor2(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(dpp_i32<0x4E>(hi), dpp_i32<0x4E>(lo)); // quad_perm [2,3,0,1]
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// log of the site likelihood with the scaling undone (src/core_likelihood.c:1462-1481 for an
// edge, :193-198 for a root)
__device__ __forceinline__ double finish_site(double terma, double terminv, unsigned scalings,
                                              int is_root)
{
  if (is_root) return log(terma + terminv) + (scalings ? scalings * PLLGPU_LOG_THRESHOLD : 0.0);
  if (scalings)
  {
    if (terminv > 0.)
    {
      unsigned c = scalings < PLLGPU_RATE_MAXDIFF ? scalings : PLLGPU_RATE_MAXDIFF;
      return log(ldexp(terma, -256 * (int)c) + terminv);
    }
    return log(terma) + scalings * PLLGPU_LOG_THRESHOLD;
  }
  return log(terma + terminv);
}

// 2^(-256 d) for d = 1..4 (src/core_likelihood.c:1366-1375)
__device__ __forceinline__ double minlh(unsigned d)
{
  return ldexp(1.0, -256 * (int)d);
}

// Every workgroup leaves its partial sum; the workgroup that arrives last adds all partials in
// index order (fixed tree: the result does not depend on arrival order) and writes the total to
// host-mapped memory. One launch, no separate reduction kernel, no D2H copy.
// Hand-off WITHOUT release/acquire fences: on this part a release fence at agent scope is a write-back
// of every dirty line of the XCD's L2 (the eight XCDs do not share one), and a log-likelihood kernel
// usually runs right after - or, with the tail kernels, while - hundreds of MB of CLVs are written:
// one such write-back per workgroup cost more than the kernel's own work. Instead the partial itself
// is stored with an agent-scope atomic (performed at the coherent level, nothing else is flushed), the
// wave waits for that store, then takes its ticket; the last workgroup reads the partials with
// agent-scope atomic loads. nsum_waves: how many of the calling workgroup's waves contribute a value.
__device__ __forceinline__ void partial_store(double *p, double v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double partial_load(const double *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The three ordering points of such a hand-off. Default: wait for the wave's own stores (they were performed at
// the coherent level), nothing else.
// THE ASSUMPTION, stated once: this is ISA-level reasoning, outside the HIP memory model. On gfx950 an
// agent-scope atomic store is performed at the coherent level (the memory side of the XCD's L2: `sc1`, write-through)
// and is acknowledged - the wave's vmcnt decrements - only once it has been; so after "s_waitcnt vmcnt(0)" the
// partial is where an agent-scope atomic load of any other XCD finds it, and the ticket taken afterwards cannot
// overtake it (MI355X_MICROARCH.md, "Valid forms": 8-byte agent atomics on both sides, drained before the counter
// add). Another architecture gets no such promise: the library refuses anything but gfx950 at context creation
// (pllgpu_create), PLL_AMD_FENCED_HANDOFF=1 is the in-model form, and the GPU suite compares the two bit for bit on
// cases that span all eight XCDs (tests/test_gpu_parity.py::test_fenced_handoff_*) and under load
// (tools/handoff_stress.py). fenced (PLL_AMD_FENCED_HANDOFF=1, a diagnosis switch): the textbook form
// inside the HIP memory model - release fence before the ticket / the sequence word, acquire fence in the
// workgroup that arrived last - at the price of an L2 write-back per workgroup.
__device__ __forceinline__ void handoff_before_ticket(int fenced)
{
  if (fenced)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void handoff_after_last_ticket(int fenced)
{
  if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void handoff_before_sequence(int fenced)
{
  if (fenced)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the last workgroup's share of the partials: thread t adds slots t, t + blockDim, ... in that order. All requests of a
// batch of 16 first, then the additions: the loads are agent-scope atomics (each a trip to the coherent level, ~2 us
// under load), and a loop of load / add made the tail of every evaluation as many trips long as a thread had slots.
__device__ __forceinline__ double sum_partials_strided(const double *block_sums, unsigned n)
{
  double a = 0.0;
  const unsigned stride = blockDim.x;
  for (unsigned base = 0; base < n; base += 16u * stride)
  {
    double v[16];
#pragma unroll
    for (unsigned q = 0; q < 16u; ++q)
    {
      const unsigned i = base + q * stride + threadIdx.x;
      v[q] = i < n ? partial_load(&block_sums[i]) : 0.0;
    }
#pragma unroll
    for (unsigned q = 0; q < 16u; ++q)
      if (base + q * stride + threadIdx.x < n) a += v[q];
  }
  return a;
}

// The same when only SOME workgroups hold a value and which ones is decided at run time (k_edge_mfma: the workgroup
// that finishes an item block last): the value goes to the slot of the ITEM BLOCK, not of the workgroup, so that the
// final sum adds the same numbers in the same places whichever workgroup produced them - with one slot per
// workgroup the non-zero entries moved between threads of the last workgroup from run to run and the total with
// them, by an ulp or two.
__device__ __forceinline__ void publish_block_sum_slot(const DevEdge &e, double wave_value, unsigned nsum_waves, bool has_value, unsigned slot,
                                                       unsigned nslots)
{
  __shared__ double ws[4];
  __shared__ unsigned last;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) ws[wave] = wave < nsum_waves ? wave_value : 0.0;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    if (has_value)
    {
      double s = ws[0];
      for (unsigned w = 1; w < nw; ++w) s += ws[w];
      partial_store(&e.block_sums[slot], s);
    }
    handoff_before_ticket(e.fenced);
    const unsigned ticket = __hip_atomic_fetch_add(e.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (ticket == gridDim.x - 1) ? 1u : 0u;
    if (last) handoff_after_last_ticket(e.fenced);
  }
  __syncthreads();
  if (!last) return;
  double a = sum_partials_strided(e.block_sums, nslots);
  a = wave_sum(a);
  __syncthreads();
  if (lane == 0) ws[wave] = a;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double s = ws[0];
    for (unsigned w = 1; w < nw; ++w) s += ws[w];
    __hip_atomic_store(e.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(e.result, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    handoff_before_sequence(e.fenced);
    __hip_atomic_store(e.result + 1, e.sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__device__ __forceinline__ void publish_block_sum(const DevEdge &e, double wave_value, unsigned nsum_waves)
{
  __shared__ double ws[4];
  __shared__ unsigned last;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) ws[wave] = wave < nsum_waves ? wave_value : 0.0;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double s = ws[0];
    for (unsigned w = 1; w < nw; ++w) s += ws[w];
    partial_store(&e.block_sums[blockIdx.x], s);
    handoff_before_ticket(e.fenced); // the partial has been performed before the ticket is taken
    const unsigned ticket = __hip_atomic_fetch_add(e.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (ticket == gridDim.x - 1) ? 1u : 0u;
    if (last) handoff_after_last_ticket(e.fenced);
  }
  __syncthreads();
  if (!last) return;
  double a = sum_partials_strided(e.block_sums, gridDim.x);
  a = wave_sum(a);
  __syncthreads();
  if (lane == 0) ws[wave] = a;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    double s = ws[0];
    for (unsigned w = 1; w < nw; ++w) s += ws[w];
    __hip_atomic_store(e.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // result[0] = value, then result[1] = this call's sequence number with system-scope release:
    // the host polls the sequence word in mapped memory instead of paying a stream synchronise
    __hip_atomic_store(e.result, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    handoff_before_sequence(e.fenced); // the value is in host memory before the sequence word follows
    __hip_atomic_store(e.result + 1, e.sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
