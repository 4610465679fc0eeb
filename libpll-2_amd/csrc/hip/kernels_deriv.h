// kernels_deriv.h - branch-length derivatives (pll_update_sumtable /
// pll_compute_likelihood_derivatives, src/derivatives.c, src/core_derivatives.c).
//
// The sumtable   sum[n][k][j] = (sum_i p_i pi_i Vinv[i][j]) * (sum_i V[j][i] c_i)
// has exactly the shape of a CLV update (two matrix-vector products per (site, rate), multiplied
// element-wise), so it is produced by the CLV-update kernels themselves with the two "transition
// matrices" replaced by M1[j][i] = pi_i Vinv[i][j] and M2[j][i] = V[j][i] (uploaded in the same
// transposed layout) and the table - resident in HBM, tiled like a CLV - as the "parent". Only the
// per-rate-scaler variant needs an extra pass (k_sumtable_excess, src/core_derivatives.c:420-460).
//
// Derivatives at a branch length t: a one-workgroup pre-kernel fills diag[k][j] = (e, l e, l^2 e)
// with e = exp(lambda_j r_k t / (1 - pinv)), l = lambda_j r_k / (1 - pinv) (:757-772); the main kernel
// streams the table once (lane = site, one wave per tile, diag through the scalar path), forms the
// site's (L, L', L'') and accumulates pattern_weight * (-L'/L) and pattern_weight *
// ((L'/L)^2 - L''/L) (:643-694, :843-847); the last workgroup to arrive adds the partials in index
// order and writes {d_f, dd_f, sequence} to mapped host memory.
#pragma once
#include "kernels_common.h"

struct DevDeriv
{
  const double *table;          // tiled [tile][k][j][64]
  const double *diag;           // [k][j][4]
  const double *freqs;          // [rate_matrices][SP]
  const double *rate_weights;   // [R]
  const double *prop_invar;     // [rate_matrices]
  const unsigned *pattern_weights;
  const int *invariant;         // or null
  double *block_sums;           // [2][1024]
  unsigned *counter;
  double *result;               // mapped: [0] d_f, [1] sequence, [2] dd_f
  double sequence;
  unsigned sites;
  int fenced;                   // kernels_common.h: handoff_*
  unsigned char fidx[kMaxRates];
};

struct DevDiag
{
  double *diag;                 // [k][j][4]
  const double *eigenvals;      // [rate_matrices][SP]
  const double *rates;          // [R]
  const double *prop_invar;     // [rate_matrices]
  double branch_length;
  unsigned S, SP, R;
  unsigned char fidx[kMaxRates];
};

__global__ __launch_bounds__(256) void k_diagtable(const DevDiag d)
{
  for (unsigned idx = threadIdx.x; idx < d.R * d.S; idx += blockDim.x)
  {
    const unsigned k = idx / d.S, j = idx % d.S;
    const unsigned fi = d.fidx[k];
    const double ki = d.rates[k] / (1.0 - d.prop_invar[fi]);
    const double lam = d.eigenvals[(size_t)fi * d.SP + j];
    const double e = exp(lam * ki * d.branch_length);
    double *o = d.diag + (size_t)idx * 4;
    o[0] = e;
    o[1] = lam * ki * e;
    o[2] = lam * ki * lam * ki * e;
    o[3] = 0.0;
  }
}

// local_diag: every workgroup forms the (small) diag table itself in LDS instead of waiting for a
// k_diagtable launch - one launch per derivative evaluation, the call a Newton iteration repeats;
// workgroup 0 also leaves the table in d.diag_out for k_asc_deriv_terms.
__global__ __launch_bounds__(256) void k_derivatives(const DevDeriv d, const GenGeo g, unsigned tiles_per_wave, const DevDiag dg,
                                                     unsigned local_diag)
{
  __shared__ double ws[2][4];
  __shared__ unsigned last;
  extern __shared__ double ldiag[]; // [R][S][4] when local_diag
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ntiles = (d.sites + 63u) / 64u;
  if (local_diag)
  {
    for (unsigned idx = threadIdx.x; idx < dg.R * dg.S; idx += blockDim.x)
    {
      const unsigned k = idx / dg.S, j = idx % dg.S;
      const unsigned fi = dg.fidx[k];
      const double ki = dg.rates[k] / (1.0 - dg.prop_invar[fi]);
      const double lam = dg.eigenvals[(size_t)fi * dg.SP + j];
      const double e = exp(lam * ki * dg.branch_length);
      double *o = ldiag + (size_t)idx * 4;
      o[0] = e;
      o[1] = lam * ki * e;
      o[2] = lam * ki * lam * ki * e;
      o[3] = 0.0;
      if (blockIdx.x == 0)
      {
        double *go = dg.diag + (size_t)idx * 4;
        go[0] = o[0];
        go[1] = o[1];
        go[2] = o[2];
        go[3] = 0.0;
      }
    }
    __syncthreads();
  }
  const double *diag = local_diag ? ldiag : d.diag;
  double a1 = 0.0, a2 = 0.0;

  for (unsigned t = 0; t < tiles_per_wave; ++t)
  {
    const unsigned tile = (blockIdx.x * 4u + wave) * tiles_per_wave + t;
    if (tile >= ntiles) break; // wave-uniform
    const unsigned n = tile * 64u + lane;
    const bool valid = n < d.sites;
    const unsigned nn = valid ? n : d.sites - 1;
    const double *x = d.table + tiled_base(nn, g.tile_sz);
    const int inv = d.invariant ? d.invariant[nn] : -1;
    double lk0 = 0.0, lk1 = 0.0, lk2 = 0.0;
    for (unsigned k = 0; k < g.R; ++k)
    {
      double c0 = 0.0, c1 = 0.0, c2 = 0.0;
      const double *dk = diag + (size_t)k * g.S * 4; // wave-uniform addresses: LDS broadcast or scalarised loads
      const double *xk = x + (size_t)k * g.S * 64;
#pragma unroll 4
      for (unsigned j = 0; j < g.S; ++j)
      {
        const double s = __builtin_nontemporal_load(xk + (size_t)j * 64);
        c0 = fma(s, dk[j * 4 + 0], c0);
        c1 = fma(s, dk[j * 4 + 1], c1);
        c2 = fma(s, dk[j * 4 + 2], c2);
      }
      const unsigned fi = d.fidx[k];
      const double pinv = d.prop_invar ? d.prop_invar[fi] : 0.0;
      if (pinv > 0.0)
      {
        const double isl = inv >= 0 ? d.freqs[(size_t)fi * g.SP + inv] * pinv : 0.0;
        c0 = c0 * (1.0 - pinv) + isl;
        c1 = c1 * (1.0 - pinv);
        c2 = c2 * (1.0 - pinv);
      }
      const double w = d.rate_weights[k];
      lk0 += c0 * w;
      lk1 += c1 * w;
      lk2 += c2 * w;
    }
    if (valid)
    {
      const double d1 = -lk1 / lk0;
      const double d2 = d1 * d1 - lk2 / lk0;
      const double pw = (double)d.pattern_weights[n];
      a1 += pw * d1;
      a2 += pw * d2;
    }
  }
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if (lane == 0)
  {
    ws[0][wave] = a1;
    ws[1][wave] = a2;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    // hand-off without fences (kernels_common.h: partial_store)
    partial_store(&d.block_sums[blockIdx.x], (ws[0][0] + ws[0][1]) + (ws[0][2] + ws[0][3]));
    partial_store(&d.block_sums[1024 + blockIdx.x], (ws[1][0] + ws[1][1]) + (ws[1][2] + ws[1][3]));
    handoff_before_ticket(d.fenced);
    const unsigned ticket = __hip_atomic_fetch_add(d.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (ticket == gridDim.x - 1) ? 1u : 0u;
    if (last) handoff_after_last_ticket(d.fenced);
  }
  __syncthreads();
  if (!last) return;
  double b1 = 0.0, b2 = 0.0;
  for (unsigned i = threadIdx.x; i < gridDim.x; i += 256)
  {
    b1 += partial_load(&d.block_sums[i]);
    b2 += partial_load(&d.block_sums[1024 + i]);
  }
  b1 = wave_sum(b1);
  b2 = wave_sum(b2);
  __syncthreads();
  if (lane == 0)
  {
    ws[0][wave] = b1;
    ws[1][wave] = b2;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    __hip_atomic_store(d.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d.result, (ws[0][0] + ws[0][1]) + (ws[0][2] + ws[0][3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(d.result + 2, (ws[1][0] + ws[1][1]) + (ws[1][2] + ws[1][3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    handoff_before_sequence(d.fenced); // the values are in host memory before the sequence word follows
    __hip_atomic_store(d.result + 1, d.sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// per-rate scalers only: multiply column k of every site by 2^(-256 * min(rs_k - min_k rs, 4))
// (src/core_derivatives.c:420-460). One wave per tile.
struct DevExcess
{
  double *table;
  const unsigned *pscaler, *cscaler; // [entry][R] or null
  const unsigned *psid, *csid;       // site -> entry or null
  unsigned sites;
};

__global__ __launch_bounds__(256) void k_sumtable_excess(const DevExcess e, const GenGeo g)
{
  const unsigned lane = threadIdx.x & 63u;
  const unsigned tile = blockIdx.x * 4u + (threadIdx.x >> 6);
  const unsigned n = tile * 64u + lane;
  if (n >= e.sites) return;
  const unsigned pe = e.psid ? e.psid[n] : n;
  const unsigned ce = e.csid ? e.csid[n] : n;
  unsigned mn = 0xFFFFFFFFu;
  for (unsigned k = 0; k < g.R; ++k)
  {
    const unsigned rs = (e.pscaler ? e.pscaler[(size_t)pe * g.R + k] : 0u) + (e.cscaler ? e.cscaler[(size_t)ce * g.R + k] : 0u);
    mn = min(mn, rs);
  }
  double *x = e.table + tiled_base(n, g.tile_sz);
  for (unsigned k = 0; k < g.R; ++k)
  {
    const unsigned rs = (e.pscaler ? e.pscaler[(size_t)pe * g.R + k] : 0u) + (e.cscaler ? e.cscaler[(size_t)ce * g.R + k] : 0u);
    const unsigned ex = min(rs - mn, PLLGPU_RATE_MAXDIFF);
    if (!ex) continue;
    const double f = minlh(ex);
    for (unsigned j = 0; j < g.S; ++j) x[((size_t)k * g.S + j) * 64] *= f;
  }
}

// ---- ascertainment-bias terms (src/likelihood.c:50-120, :191-268, :342-440; src/core_derivatives.c:
// 864-891): the per-state extra entries sites + n. A handful of entries: one wave per state, lanes
// over the parent state j; results go to mapped host memory, the host applies the correction formula.
struct DevAsc
{
  const double *parent;         // tiled CLV
  const double *child;          // tiled CLV or null (tip codes / root)
  const unsigned char *ctip;
  const unsigned *pscaler, *cscaler;
  const double *mat;            // PT layout; unused for root
  const double *freqs;          // [rate_matrices][SP]
  const double *rate_weights;
  double *out;                  // mapped: [S] terms, then [S] scaling counts (as doubles)
  unsigned first;               // entry of state 0 (= sites)
  int per_rate;
  int is_root;
  unsigned char fidx[kMaxRates];
};

__global__ __launch_bounds__(64) void k_asc_terms(const DevAsc a, const GenGeo g, const unsigned long long *__restrict__ tipmap)
{
  const unsigned n = blockIdx.x, j = threadIdx.x;
  const unsigned S = g.S, R = g.R;
  const unsigned e = a.first + n;
  const double *xp = a.parent + tiled_base(e, g.tile_sz);
  const double *xc = a.child ? a.child + tiled_base(e, g.tile_sz) : nullptr;
  unsigned long long mask = 0;
  if (a.ctip) mask = tipmap ? tipmap[a.ctip[e]] : (unsigned long long)a.ctip[e];
  unsigned mn = 0xFFFFFFFFu;
  if (a.per_rate)
    for (unsigned i = 0; i < R; ++i)
      mn = min(mn, (a.pscaler ? a.pscaler[(size_t)e * R + i] : 0u) + (a.cscaler ? a.cscaler[(size_t)e * R + i] : 0u));
  double term = 0.0;
  for (unsigned i = 0; i < R; ++i)
  {
    double v = 0.0;
    if (j < S)
    {
      double termb = 1.0;
      if (!a.is_root)
      {
        termb = 0.0;
        const double *col = a.mat + (size_t)i * S * g.SPT + j; // PT[i][k][j] = P_i[j][k]
        if (a.ctip)
        {
          for (unsigned k = 0; k < S; ++k)
            if ((mask >> k) & 1ull) termb += col[(size_t)k * g.SPT];
        }
        else
          for (unsigned k = 0; k < S; ++k) termb += col[(size_t)k * g.SPT] * xc[((size_t)i * S + k) * 64];
      }
      v = xp[((size_t)i * S + j) * 64] * a.freqs[(size_t)a.fidx[i] * g.SP + j] * termb;
    }
    v = wave_sum(v);
    if (a.per_rate)
    {
      const unsigned rs = (a.pscaler ? a.pscaler[(size_t)e * R + i] : 0u) + (a.cscaler ? a.cscaler[(size_t)e * R + i] : 0u);
      const unsigned d = min(rs - mn, PLLGPU_RATE_MAXDIFF);
      if (d) v *= minlh(d);
    }
    term += v * a.rate_weights[i];
  }
  if (j == 0)
  {
    const unsigned sc = a.per_rate ? mn : (a.pscaler ? a.pscaler[e] : 0u) + (a.cscaler ? a.cscaler[e] : 0u);
    a.out[n] = term;
    a.out[S + n] = (double)sc;
  }
}

struct DevAscDeriv
{
  const double *table;          // tiled sumtable
  const double *diag;           // [k][j][4], left by the last derivative evaluation
  const double *rate_weights;
  const unsigned *pscaler, *cscaler;
  double *out;                  // mapped: [S][3] then [S] scaling counts
  unsigned first;
  int per_rate;
};

__global__ __launch_bounds__(64) void k_asc_deriv_terms(const DevAscDeriv a, const GenGeo g)
{
  const unsigned n = threadIdx.x;
  if (n >= g.S) return;
  const unsigned e = a.first + n;
  const double *x = a.table + tiled_base(e, g.tile_sz);
  double lk0 = 0.0, lk1 = 0.0, lk2 = 0.0;
  unsigned mn = 0xFFFFFFFFu;
  for (unsigned k = 0; k < g.R; ++k)
  {
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    for (unsigned j = 0; j < g.S; ++j)
    {
      const double s = x[((size_t)k * g.S + j) * 64];
      const double *dk = a.diag + ((size_t)k * g.S + j) * 4;
      c0 = fma(s, dk[0], c0);
      c1 = fma(s, dk[1], c1);
      c2 = fma(s, dk[2], c2);
    }
    const double w = a.rate_weights[k];
    lk0 += c0 * w;
    lk1 += c1 * w;
    lk2 += c2 * w;
    if (a.per_rate)
      mn = min(mn, (a.pscaler ? a.pscaler[(size_t)e * g.R + k] : 0u) + (a.cscaler ? a.cscaler[(size_t)e * g.R + k] : 0u));
  }
  // per-rate scalers: the table columns were already brought to the smallest count (k_sumtable_excess)
  const unsigned sc = a.per_rate ? mn : (a.pscaler ? a.pscaler[e] : 0u) + (a.cscaler ? a.cscaler[e] : 0u);
  a.out[n * 3 + 0] = lk0;
  a.out[n * 3 + 1] = lk1;
  a.out[n * 3 + 2] = lk2;
  a.out[3 * g.S + n] = (double)sc;
}

// ---- transition matrices (src/core_pmatrix.c:186-247) -----------------
// One workgroup per (matrix, rate category): A[i][m] = Vinv[i][m] * expm1(lambda_m r t / (1-pinv))
// and B = V staged in LDS, every thread forms entries P[i][j] = delta_ij + sum_m A[i][m] B[m][j] in
// the reference's summation order and stores them transposed (PT[j][i], the kernels' layout).
struct DevPmat
{
  double *pmat;                 // base of the device matrix block
  const double *evecs, *ievecs; // [rate_matrices][S][SP]
  const double *evals;          // [rate_matrices][SP]
  const double *rates, *prop_invar;
  const unsigned *mindex;       // [count]
  const double *brlen;          // [count]
  size_t pm_stride;
  unsigned S, SP, SPT;
  unsigned char fidx[kMaxRates];
};

__device__ __forceinline__ void pmatrix_body(const DevPmat &d, double t, unsigned mi)
{
  extern __shared__ double sm[];
  const unsigned S = d.S, LD = S | 1u; // odd row stride: conflict-free column walks
  double *A = sm, *B = sm + (size_t)S * LD;
  const unsigned n = blockIdx.y;
  const unsigned fi = d.fidx[n];
  double *out = d.pmat + (size_t)mi * d.pm_stride + (size_t)n * S * d.SPT;
  if (t > 0.0)
  {
    const double pinv = d.prop_invar[fi];
    const double scale = pinv > 1e-8 ? d.rates[n] * t / (1.0 - pinv) : d.rates[n] * t; // PLL_MISC_EPSILON
    const double *ev = d.evecs + (size_t)fi * S * d.SP, *iev = d.ievecs + (size_t)fi * S * d.SP;
    const double *lam = d.evals + (size_t)fi * d.SP;
    for (unsigned idx = threadIdx.x; idx < S * S; idx += blockDim.x)
    {
      const unsigned i = idx / S, m = idx % S;
      A[i * LD + m] = iev[(size_t)i * d.SP + m] * expm1(lam[m] * scale);
      B[i * LD + m] = ev[(size_t)i * d.SP + m];
    }
    __syncthreads();
  }
  for (unsigned idx = threadIdx.x; idx < S * d.SPT; idx += blockDim.x)
  {
    const unsigned j = idx / d.SPT, i = idx % d.SPT;
    double acc = 0.0;
    if (i < S)
    {
      acc = (i == j) ? 1.0 : 0.0;
      if (t > 0.0)
        for (unsigned m = 0; m < S; ++m) acc += A[i * LD + m] * B[m * LD + j];
    }
    out[idx] = acc;
  }
}

__global__ __launch_bounds__(256) void k_pmatrix(const DevPmat d)
{
  pmatrix_body(d, d.brlen[blockIdx.x], d.mindex[blockIdx.x]);
}

// a handful of branches (what a tree search changes per move): indices and lengths by value in the kernarg segment - no
// staging copies ahead of the launch (two of them were half of the call's 12 us)
constexpr unsigned kPmatInline = 16;
struct DevPmatFew
{
  DevPmat d;
  double t[kPmatInline];
  unsigned mi[kPmatInline];
};
__global__ __launch_bounds__(256) void k_pmatrix_few(const DevPmatFew f)
{
  pmatrix_body(f.d, f.t[blockIdx.x], f.mi[blockIdx.x]);
}
