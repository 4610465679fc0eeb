/* models.c - host-side producers of the hot path's inputs ("next";
 * O(branches * states^3) scalar work that stays on the host and feeds the device through the
 * dirty flags): discrete-Gamma rates, eigen-decomposition of a reversible rate matrix,
 * transition-probability matrices, invariant-site detection.
 *
 * Own numerical routes, same mathematics as the reference:
 *   rate matrix        src/models.c:182-252 (symmetrised sqrt(pi) Q sqrt(pi)^-1, mean rate 1,
 *                      states with frequency <= 1e-6 dropped from the system, :254-291)
 *   eigensystem        cyclic Jacobi rotations instead of the reference's tred2/tqli (:24-178);
 *                      array conventions of :346-398 kept (eigenvecs = U^T sqrt(pi),
 *                      inv_eigenvecs = sqrt(pi)^-1 U) because callers read them
 *   P(t)               I + inv_eigenvecs * expm1(lambda r t / (1-pinv)) * eigenvecs,
 *                      src/core_pmatrix.c:201-245
 *   Gamma categories   src/gamma.c:220-292 (Yang 1994); the incomplete-gamma integral is
 *                      evaluated to ~1e-15 here, the reference's AS32 routine stops at 1e-8, so
 *                      category rates agree with the reference to ~1e-7 relative, not 1e-10.
 */
#include <math.h>

#include "pll_internal.h"

#define MINFREQ 1e-6 /* PLL_EIGEN_MINFREQ */

/* ---- regularised lower incomplete gamma P(a, x) and its inverse -------------------------------- */
static double gamma_p(double a, double x)
{
  if (x <= 0) return 0.0;
  const double lg = lgamma(a);
  if (x < a + 1.0)
  {
    /* series */
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 10000; ++n)
    {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (fabs(del) < fabs(sum) * 1e-17) break;
    }
    return sum * exp(-x + a * log(x) - lg);
  }
  /* Lentz continued fraction for Q = 1 - P */
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 10000; ++i)
  {
    const double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < tiny) d = tiny;
    c = b + an / c;
    if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return 1.0 - exp(-x + a * log(x) - lg) * h;
}

/* x with P(a, x) = p: bracketing + bisection/Newton hybrid, monotone function */
static double gamma_p_inv(double a, double p)
{
  if (p <= 0) return 0.0;
  double lo = 0.0, hi = a > 1 ? a : 1.0;
  while (gamma_p(a, hi) < p) hi *= 2.0;
  double x = 0.5 * (lo + hi);
  const double lg = lgamma(a);
  for (int it = 0; it < 200; ++it)
  {
    const double f = gamma_p(a, x) - p;
    if (f > 0) hi = x; else lo = x;
    const double dens = exp(-x + (a - 1.0) * log(x) - lg);
    double nx = dens > 0 ? x - f / dens : 0.5 * (lo + hi);
    if (!(nx > lo && nx < hi)) nx = 0.5 * (lo + hi);
    if (fabs(nx - x) <= 1e-16 * fabs(x)) { x = nx; break; }
    x = nx;
  }
  return x;
}

int pll_compute_gamma_cats(double alpha, unsigned int categories, double *rates, int rates_mode)
{
  unsigned int i;
  if (alpha < 0.02 /* ALPHA_MIN, src/gamma.c:25 */ || categories < 1)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "Invalid alpha value (%f)", alpha);
    return PLL_FAILURE;
  }
  if (categories == 1)
  {
    rates[0] = 1.0;
    return PLL_SUCCESS;
  }
  if (rates_mode == PLL_GAMMA_RATES_MEDIAN)
  {
    double sum = 0;
    for (i = 0; i < categories; ++i)
    {
      rates[i] = gamma_p_inv(alpha, (2.0 * i + 1.0) / (2.0 * categories)) / alpha;
      sum += rates[i];
    }
    for (i = 0; i < categories; ++i) rates[i] *= categories / sum;
    return PLL_SUCCESS;
  }
  if (rates_mode != PLL_GAMMA_RATES_MEAN)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "Invalid GAMMA discretization mode (%d)", rates_mode);
    return PLL_FAILURE;
  }
  /* mean of each equal-probability slice of Gamma(alpha, alpha) */
  double prev = 0.0;
  for (i = 0; i < categories; ++i)
  {
    double upper = 1.0;
    if (i + 1 < categories)
    {
      const double cut = gamma_p_inv(alpha, (double)(i + 1) / categories); /* in units of alpha*x */
      upper = gamma_p(alpha + 1.0, cut);
    }
    rates[i] = (upper - prev) * categories;
    prev = upper;
  }
  return PLL_SUCCESS;
}

/* ---- eigensystem of a symmetric matrix: cyclic Jacobi ------------------------------------------ */
/* a: n x n symmetric (destroyed), v: n x n, columns become orthonormal eigenvectors, d: eigenvalues */
static void jacobi_eigen(double *a, unsigned int n, double *d, double *v)
{
  unsigned int i, j, p, q;
  for (i = 0; i < n; ++i)
    for (j = 0; j < n; ++j) v[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 100; ++sweep)
  {
    double off = 0, diag = 0;
    for (i = 0; i < n; ++i)
    {
      diag += a[i * n + i] * a[i * n + i];
      for (j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j];
    }
    if (off <= 1e-40 * (diag + off) || off == 0.0) break;
    for (p = 0; p + 1 < n; ++p)
      for (q = p + 1; q < n; ++q)
      {
        const double apq = a[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (i = 0; i < n; ++i)
        {
          const double aip = a[i * n + p], aiq = a[i * n + q];
          a[i * n + p] = c * aip - s * aiq;
          a[i * n + q] = s * aip + c * aiq;
        }
        for (i = 0; i < n; ++i)
        {
          const double api = a[p * n + i], aqi = a[q * n + i];
          a[p * n + i] = c * api - s * aqi;
          a[q * n + i] = s * api + c * aqi;
        }
        for (i = 0; i < n; ++i)
        {
          const double vip = v[i * n + p], viq = v[i * n + q];
          v[i * n + p] = c * vip - s * viq;
          v[i * n + q] = s * vip + c * viq;
        }
      }
  }
  for (i = 0; i < n; ++i) d[i] = a[i * n + i];
}

int pll_update_eigen(pll_partition_t *p, unsigned int idx)
{
  const unsigned int s = p->states, sp = p->states_padded;
  const double *freqs = p->frequencies[idx];
  const double *params = p->subst_params[idx];
  unsigned int i, j, k;
  unsigned int *keep = (unsigned int *)malloc(sizeof(unsigned int) * s);
  double *a = (double *)calloc((size_t)s * s, sizeof(double));
  double *v = (double *)malloc(sizeof(double) * s * s);
  double *d = (double *)malloc(sizeof(double) * s);
  double *full = (double *)calloc((size_t)s * s, sizeof(double));
  if (!keep || !a || !v || !d || !full)
  {
    free(keep); free(a); free(v); free(d); free(full);
    pll_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
    return PLL_FAILURE;
  }
  /* exchangeabilities relative to the last one (src/models.c:199-203) */
  const unsigned int np = s * (s - 1) / 2;
  const double last = params[np - 1] > 0.0 ? params[np - 1] : 1.0;
  k = 0;
  for (i = 0; i < s; ++i)
    for (j = i + 1; j < s; ++j, ++k)
    {
      const double f = (freqs[i] <= MINFREQ || freqs[j] <= MINFREQ) ? 0.0 : params[k] / last;
      full[i * s + j] = full[j * s + i] = f * sqrt(freqs[i] * freqs[j]);
      full[i * s + i] -= f * freqs[j];
      full[j * s + j] -= f * freqs[i];
    }
  double mean = 0;
  for (i = 0; i < s; ++i) mean += freqs[i] * -full[i * s + i];
  for (i = 0; i < s * s; ++i) full[i] /= mean;
  /* reduced system over the states that actually occur */
  unsigned int n = 0;
  for (i = 0; i < s; ++i)
    if (freqs[i] > MINFREQ) keep[n++] = i;
  for (i = 0; i < n; ++i)
    for (j = 0; j < n; ++j) a[i * n + j] = full[keep[i] * s + keep[j]];
  jacobi_eigen(a, n, d, v);

  double *evecs = p->eigenvecs[idx], *ievecs = p->inv_eigenvecs[idx], *evals = p->eigenvals[idx];
  memset(evecs, 0, sizeof(double) * s * sp);
  memset(ievecs, 0, sizeof(double) * s * sp);
  memset(evals, 0, sizeof(double) * sp);
  for (i = 0; i < s; ++i) evecs[i * sp + i] = ievecs[i * sp + i] = 1.0; /* dropped states: identity */
  for (i = 0; i < n; ++i)
  {
    evals[keep[i]] = d[i];
    for (j = 0; j < n; ++j)
    {
      /* row = eigen index, column = state (and the transpose for the inverse) */
      evecs[keep[i] * sp + keep[j]] = v[j * n + i] * sqrt(freqs[keep[j]]);
      ievecs[keep[i] * sp + keep[j]] = v[i * n + j] / sqrt(freqs[keep[i]]);
    }
  }
  p->eigen_decomp_valid[idx] = 1;
  {
    pll_amd_ext_t *x = pll_ext(p);
    if (x)
    {
      x->eigen_dirty[idx] = 1;
      x->eigen_version++;
      x->model_foreign[idx] = 0; /* computed here from this set's own rates and frequencies */
    }
  }
  free(keep); free(a); free(v); free(d); free(full);
  return PLL_SUCCESS;
}

int pll_update_prob_matrices(pll_partition_t *p, const unsigned int *params_indices,
                             const unsigned int *matrix_indices, const double *branch_lengths,
                             unsigned int count)
{
  /* the matrices are formed ON THE DEVICE (k_pmatrix,
   * src/core_pmatrix.c:186-247) and stay there; partition->pmatrix is a host mirror refreshed by
   * pll_gpu_sync_pmatrix() or after every call under PLL_AMD_EAGER_MIRROR=1. Only the eigensystem
   * (a few kB per rate matrix) crosses PCIe, and only when it changed. */
  unsigned int b, n;
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  if (!x || !x->ctx)
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "pll_update_prob_matrices: no MI355X context behind this partition; this library has no CPU path");
    fprintf(stderr, "libpll_amd: pll_update_prob_matrices: [%d] %s\n", pll_errno, pll_errmsg);
    return PLL_FAILURE;
  }
  for (n = 0; n < p->rate_cats; ++n)
  {
    if (params_indices[n] >= p->rate_matrices)
    {
      pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_update_prob_matrices: params_indices[%u] out of range", n);
      return PLL_FAILURE;
    }
    if (!p->eigen_decomp_valid[params_indices[n]] && !pll_update_eigen(p, params_indices[n])) return PLL_FAILURE;
  }
  for (b = 0; b < count; ++b)
    if (matrix_indices[b] >= p->prob_matrices || !(branch_lengths[b] >= 0))
    {
      pll_set_error(PLL_ERROR_PARAM_INVALID, "invalid matrix index or negative branch length");
      return PLL_FAILURE;
    }
  if (!pll_flush_eigen(p, x)) return PLL_FAILURE;
  /* The reference forms the matrices one after another (src/core_pmatrix.c:62), so an index that comes twice ends with
   * the LAST of its branch lengths - and lists like that are what callers hand over (an SPR's three changed branches plus
   * a re-optimised one next to it: found by tests/test_gpu_tree_search.py, the device wrote both in one launch and either
   * could win). Only the last occurrence of an index goes to the device. */
  unsigned int *uidx = NULL;
  double *ubl = NULL;
  unsigned int ucount = count;
  {
    int dup = 0;
    if (count <= 32)
    {
      for (b = 1; b < count && !dup; ++b)
        for (n = 0; n < b; ++n)
          if (matrix_indices[n] == matrix_indices[b])
          {
            dup = 1;
            break;
          }
    }
    else
      dup = -1; /* not looked at yet: the marks below tell */
    if (dup)
    {
      unsigned int *last = (unsigned int *)malloc(sizeof(unsigned int) * (p->prob_matrices ? p->prob_matrices : 1));
      uidx = (unsigned int *)malloc(sizeof(unsigned int) * count);
      ubl = (double *)malloc(sizeof(double) * count);
      if (!last || !uidx || !ubl)
      {
        free(last);
        free(uidx);
        free(ubl);
        pll_set_error(PLL_ERROR_MEM_ALLOC, "pll_update_prob_matrices: out of memory");
        return PLL_FAILURE;
      }
      for (b = 0; b < count; ++b) last[matrix_indices[b]] = b;
      ucount = 0;
      for (b = 0; b < count; ++b)
        if (last[matrix_indices[b]] == b)
        {
          uidx[ucount] = matrix_indices[b];
          ubl[ucount++] = branch_lengths[b];
        }
      free(last);
    }
  }
  const int rc = pllgpu_update_pmatrices(x->ctx, params_indices, uidx ? uidx : matrix_indices, ubl ? ubl : branch_lengths, ucount);
  free(uidx);
  free(ubl);
  if (rc != 0)
  {
    pll_set_gpu_error("pll_update_prob_matrices");
    return PLL_FAILURE;
  }
  for (b = 0; b < count; ++b)
  {
    x->pmatrix_dirty[matrix_indices[b]] = 0;
    x->pmatrix_stale[matrix_indices[b]] = 1;
    for (n = 0; n < p->rate_cats; ++n)
    {
      const size_t at = (size_t)matrix_indices[b] * p->rate_cats + n;
      x->pmatrix_params[at] = (params_indices[n] < 0xFFu && !x->model_foreign[params_indices[n]]) ? (unsigned char)params_indices[n] : 0xFFu;
      x->pmatrix_version[at] = x->model_version[params_indices[n]];
    }
  }
  if (x->eager_mirror) return pll_gpu_sync_pmatrix(p, -1);
  return PLL_SUCCESS;
}

/* ---- invariant sites (src/models.c:495-544, :651-752) ------------------------------------------ */
/* per site: index of the single state shared by all tips, or -1 (src/models.c:651-752) */
static int invariant_states(const pll_partition_t *p, int *out)
{
  const unsigned int s = p->states, n = p->sites;
  unsigned int i, j, k;
  pll_state_t all = (s >= 64) ? ~0ull : ((1ull << s) - 1ull);
  pll_state_t *acc = (pll_state_t *)malloc(sizeof(pll_state_t) * (n ? n : 1));
  if (!acc) return 0;
  for (j = 0; j < n; ++j) acc[j] = all;
  if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
  {
    for (i = 0; i < p->tips; ++i)
      for (j = 0; j < n; ++j)
        acc[j] &= (s == 4) ? (pll_state_t)p->tipchars[i][j] : p->tipmap[p->tipchars[i][j]];
  }
  else
  {
    const size_t span = (size_t)p->rate_cats * p->states_padded;
    for (i = 0; i < p->tips; ++i)
    {
      const unsigned int *sid = (p->repeats && p->repeats->pernode_ids[i]) ? p->repeats->pernode_site_id[i] : NULL;
      /* a tip CLV edited on the device side only would be stale here; tips are host-authored */
      for (j = 0; j < n; ++j)
      {
        const double *c = p->clv[i] + span * (sid ? sid[j] : j);
        pll_state_t m = 0;
        for (k = 0; k < s; ++k) m |= ((pll_state_t)c[k]) << k;
        acc[j] &= m;
      }
    }
  }
  for (j = 0; j < n; ++j) out[j] = (acc[j] && !(acc[j] & (acc[j] - 1))) ? __builtin_ctzll(acc[j]) : -1;
  free(acc);
  return 1;
}

int pll_update_invariant_sites(pll_partition_t *p)
{
  pll_amd_ext_t *x = pll_ext(p);
  const unsigned int n = p->sites;
  if (!p->invariant) p->invariant = (int *)malloc(sizeof(int) * (n ? n : 1));
  if (!p->invariant || !invariant_states(p, p->invariant))
  {
    pll_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate charmap for invariant sites array.");
    return PLL_FAILURE;
  }
  if (x) x->invariant_dirty = 1;
  return PLL_SUCCESS;
}

/* src/models.c:546-649: weighted number of invariant sites, and (unweighted, as in the reference)
 * how many site patterns are invariant for each state. Host bookkeeping over the tip data. The
 * reference's PATTERN_TIP branch ANDs the tip CODES (:597), which are masks only for 4 states; here
 * the masks behind the codes are used for every state count. */
unsigned int pll_count_invariant_sites(pll_partition_t *p, unsigned int *state_inv_count)
{
  unsigned int j, total = 0;
  const int *inv = p->invariant;
  int *tmp = NULL;
  if (state_inv_count) memset(state_inv_count, 0, p->states * sizeof(unsigned int));
  if (!inv)
  {
    tmp = (int *)malloc(sizeof(int) * (p->sites ? p->sites : 1));
    if (!tmp || !invariant_states(p, tmp))
    {
      free(tmp);
      pll_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate memory for counting invariant sites.");
      return 0;
    }
    inv = tmp;
  }
  for (j = 0; j < p->sites; ++j)
    if (inv[j] > -1)
    {
      total += p->pattern_weights[j];
      if (state_inv_count) state_inv_count[inv[j]]++;
    }
  free(tmp);
  return total;
}

int pll_update_invariant_sites_proportion(pll_partition_t *p, unsigned int idx, double pinv)
{
  pll_amd_ext_t *x = pll_ext(p);
  if (pinv != 0.0 && (p->attributes & PLL_ATTRIB_AB_MASK)) /* src/models.c:500-508 */
  {
    pll_set_error(PLL_ERROR_INVAR_INCOMPAT, "Invariant sites are not compatible with asc bias correction");
    return PLL_FAILURE;
  }
  if (pinv < 0 || pinv >= 1)
  {
    pll_set_error(PLL_ERROR_INVAR_PROPORTION, "Invalid proportion of invariant sites (%f)", pinv);
    return PLL_FAILURE;
  }
  if (idx >= p->rate_matrices)
  {
    pll_set_error(PLL_ERROR_INVAR_PARAMINDEX, "Invalid params index (%u)", idx);
    return PLL_FAILURE;
  }
  if (pinv > 0.0 && !p->invariant && !pll_update_invariant_sites(p))
  {
    pll_set_error(PLL_ERROR_INVAR_NONEFOUND, "No invariant sites found");
    return PLL_FAILURE;
  }
  p->prop_invar[idx] = pinv;
  if (x) x->prop_invar_dirty = 1;
  return PLL_SUCCESS;
}
