/* likelihood.c - pll_compute_edge_loglikelihood / pll_compute_root_loglikelihood.
 *
 * Case selection mirrors src/likelihood.c:586-636: site repeats on either end -> gather maps;
 * PATTERN_TIP with a tip end -> the inner node becomes the "parent" and the tip is read as codes;
 * otherwise two CLVs. One kernel evaluates all sites; the last workgroup to arrive adds the
 * per-workgroup partials in a fixed order and leaves {lnL, call sequence} in mapped host memory,
 * which the call polls (or in device memory for pll_gpu_edge_loglikelihood_async).
 */
#include <math.h>

#include "pll_internal.h"

static double fail_lnl(const char *what)
{
  fprintf(stderr, "libpll_amd: %s: [%d] %s\n", what, pll_errno, pll_errmsg);
  return -INFINITY; /* the reference's failure value, src/core_likelihood.c:1384 */
}

/* Ascertainment-bias correction (src/likelihood.c:24-48 and the three *_asc_bias functions): the
 * device returns, per state n, the likelihood of the extra entry sites + n and its scaling count;
 * the correction formula runs here in the reference's order.
 * Deliberate differences from the reference, both in code it gets wrong:
 *  - with PLL_ATTRIB_RATE_SCALERS it indexes the [entry][rate] scaler array as if it were per site
 *    (:219,241,376-379,411-412); here the per-rate counts are honoured (min + capped differences,
 *    like the main kernel);
 *  - pll_compute_root_loglikelihood locates the extra entries at sites_number - rate_cats (:180),
 *    which is only right when rate_cats == states; here they are taken where they are stored.
 * Kept as is: the Stamatakis term adds scale_factors * log(2^-256) unweighted (:98-100). */
static double asc_correction(pll_partition_t *p, pll_amd_ext_t *x, const pllgpu_edge_t *e, int is_root)
{
  double terms[64];
  unsigned int sc[64];
  unsigned int n, sum_w_inv = 0;
  const int type = (int)(p->attributes & PLL_ATTRIB_AB_MASK);
  const unsigned int *w = p->pattern_weights + p->sites;
  double base = 0.0;
  if (pllgpu_asc_terms(x->ctx, e, is_root, terms, sc) != 0)
  {
    pll_set_gpu_error("ascertainment bias terms");
    return -INFINITY;
  }
  for (n = 0; n < p->states; ++n)
  {
    double site_lk;
    sum_w_inv += w[n];
    if (type == PLL_ATTRIB_AB_STAMATAKIS)
    {
      site_lk = log(terms[n]) * w[n];
      if (sc[n]) site_lk += sc[n] * log(PLL_SCALE_THRESHOLD);
    }
    else
      site_lk = terms[n] * pow(PLL_SCALE_THRESHOLD, (double)sc[n]);
    base += site_lk;
  }
  switch (type)
  {
    case PLL_ATTRIB_AB_LEWIS: return -(p->pattern_weight_sum * log(1 - base));
    case PLL_ATTRIB_AB_STAMATAKIS: return base;
    case PLL_ATTRIB_AB_FELSENSTEIN: return sum_w_inv * log(base);
    default:
      pll_set_error(PLL_ERROR_AB_INVALIDMETHOD, "Illegal ascertainment bias algorithm");
      return -INFINITY;
  }
}

static int prepare_end(pll_partition_t *p, pll_amd_ext_t *x, unsigned int clv, int scaler)
{
  if (!pll_flush_clv(p, x, clv)) return 0;
  if (!pll_tip_by_codes(p, clv) && !pll_flush_scaler(p, x, scaler)) return 0;
  if (!pll_flush_repeats(p, x, clv)) return 0;
  return 1;
}

/* A parent end that the device reads as tip codes is evaluated with the two ends swapped (the tip
 * kernels apply P on the tip side). The reference does that only for PLL_ATTRIB_PATTERN_TIP tips
 * (src/likelihood.c:612-624); compact indicator tips are this library's own device format, so for
 * them the swap must not change the value: sum_i p_i pi_i sum_j P_ij c_j is symmetric in (p, c) iff
 * pi_i P_ij = pi_j P_ji for the frequency set of every rate category. That is known to hold when
 * pll_update_prob_matrices formed the matrix from an eigensystem this library computed for the very
 * parameter set whose frequencies the caller names in freqs_indices, and neither the frequencies nor
 * the substitution parameters of that set have changed since: every set carries a version counter
 * (pll_set_frequencies, pll_set_subst_params, pll_gpu_invalidate FREQS / EIGEN move it) and the matrix
 * remembers the version it was formed at. The CURRENT eigen_decomp_valid flag says nothing about
 * that: it is set again by any later pll_update_prob_matrices / pll_update_sumtable on the new
 * frequencies, and a caller who writes p->frequencies and invalidates never clears it. A matrix the
 * caller wrote, an eigensystem the caller wrote, or other frequency sets: the tip is given a dense
 * CLV and the caller's orientation is evaluated as is. */
static int swap_is_exact(const pll_partition_t *p, const pll_amd_ext_t *x, unsigned int matrix_index,
                         const unsigned int *freqs_indices)
{
  unsigned int k;
  const unsigned char *formed = x->pmatrix_params + (size_t)matrix_index * p->rate_cats;
  const unsigned int *version = x->pmatrix_version + (size_t)matrix_index * p->rate_cats;
  for (k = 0; k < p->rate_cats; ++k)
    if (formed[k] == 0xFFu || formed[k] != freqs_indices[k] || version[k] != x->model_version[formed[k]]) return 0;
  return 1;
}

static double edge_lnl(pll_partition_t *p, unsigned int parent_clv_index, int parent_scaler_index,
                       unsigned int child_clv_index, int child_scaler_index, unsigned int matrix_index,
                       const unsigned int *freqs_indices, double *persite_lnl, double *device_result, double sequence)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  if (!x || !x->ctx)
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "pll_compute_edge_loglikelihood: no MI355X context behind this partition; this library has no CPU path");
    return fail_lnl("pll_compute_edge_loglikelihood");
  }
  if (parent_clv_index >= p->nodes || child_clv_index >= p->nodes || matrix_index >= p->prob_matrices ||
      parent_scaler_index >= (int)p->scale_buffers || child_scaler_index >= (int)p->scale_buffers)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_compute_edge_loglikelihood: index out of range");
    return fail_lnl("pll_compute_edge_loglikelihood");
  }
  if (pll_tip_by_codes(p, parent_clv_index) && pll_tip_by_codes(p, child_clv_index))
  {
    if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
    {
      pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_compute_edge_loglikelihood: both ends are pattern tips");
      return fail_lnl("pll_compute_edge_loglikelihood");
    }
    pll_tip_densify(p, parent_clv_index); /* two compact tips: one of them becomes a dense CLV */
  }
  if (pll_tip_by_codes(p, parent_clv_index) && !(p->attributes & PLL_ATTRIB_PATTERN_TIP) &&
      !swap_is_exact(p, x, matrix_index, freqs_indices))
    pll_tip_densify(p, parent_clv_index); /* keep the caller's orientation (src/likelihood.c:626-634) */
  const int ptip = pll_tip_by_codes(p, parent_clv_index);
  const int ctip = pll_tip_by_codes(p, child_clv_index);
  if (!pll_flush_model(p, x) || !pll_flush_pmatrix(p, x, matrix_index, matrix_index) ||
      !prepare_end(p, x, parent_clv_index, parent_scaler_index) ||
      !prepare_end(p, x, child_clv_index, child_scaler_index))
    return fail_lnl("pll_compute_edge_loglikelihood");

  pllgpu_edge_t e;
  memset(&e, 0, sizeof e);
  /* the inner node plays "parent" when the other end is a tip given by codes
   * (src/likelihood.c:612-624); P is symmetric in the reversible sense used there */
  e.parent_clv = ptip ? child_clv_index : parent_clv_index;
  e.parent_scaler = ptip ? child_scaler_index : parent_scaler_index;
  e.child_clv = ptip ? parent_clv_index : child_clv_index;
  e.child_scaler = (ptip || ctip) ? PLL_SCALE_BUFFER_NONE : child_scaler_index;
  e.child_is_tip = (ptip || ctip);
  e.matrix = matrix_index;
  e.gather = pll_repeats_enabled(p) &&
             (p->repeats->pernode_ids[parent_clv_index] || p->repeats->pernode_ids[child_clv_index]);
  e.freqs_indices = freqs_indices;
  e.want_persite = persite_lnl != NULL;
  e.device_result = device_result;
  e.sequence = sequence;
  if (device_result && (p->attributes & PLL_ATTRIB_AB_MASK))
  {
    pll_set_error(PLL_ERROR_GPU_UNSUPPORTED, "the ascertainment-bias correction needs the synchronous call");
    return fail_lnl("pll_gpu_edge_loglikelihood_async");
  }
  double lnl = 0;
  if (pllgpu_edge_loglikelihood(x->ctx, &e, persite_lnl, &lnl) != 0)
  {
    pll_set_gpu_error("pll_compute_edge_loglikelihood");
    return -INFINITY;
  }
  if (p->attributes & PLL_ATTRIB_AB_MASK) lnl += asc_correction(p, x, &e, 0);
  return lnl;
}

double pll_compute_edge_loglikelihood(pll_partition_t *p, unsigned int parent_clv_index,
                                      int parent_scaler_index, unsigned int child_clv_index,
                                      int child_scaler_index, unsigned int matrix_index,
                                      const unsigned int *freqs_indices, double *persite_lnl)
{
  return edge_lnl(p, parent_clv_index, parent_scaler_index, child_clv_index, child_scaler_index, matrix_index,
                  freqs_indices, persite_lnl, NULL, 0.0);
}

int pll_gpu_edge_loglikelihood_async(pll_partition_t *p, unsigned int parent_clv_index, int parent_scaler_index,
                                     unsigned int child_clv_index, int child_scaler_index,
                                     unsigned int matrix_index, const unsigned int *freqs_indices,
                                     double *device_result)
{
  if (!device_result)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_edge_loglikelihood_async: device_result is NULL");
    return PLL_FAILURE;
  }
  const double v = edge_lnl(p, parent_clv_index, parent_scaler_index, child_clv_index, child_scaler_index,
                            matrix_index, freqs_indices, NULL, device_result, 0.0);
  return v == 0.0 ? PLL_SUCCESS : PLL_FAILURE; /* 0 = enqueued; failures come back as -inf */
}

int pll_gpu_edge_loglikelihood_numbered(pll_partition_t *p, unsigned int parent_clv_index, int parent_scaler_index,
                                        unsigned int child_clv_index, int child_scaler_index, unsigned int matrix_index,
                                        const unsigned int *freqs_indices, double *device_result, double sequence)
{
  const double v = edge_lnl(p, parent_clv_index, parent_scaler_index, child_clv_index, child_scaler_index,
                            matrix_index, freqs_indices, NULL, device_result, sequence);
  return v == 0.0 ? PLL_SUCCESS : PLL_FAILURE;
}

double pll_compute_root_loglikelihood(pll_partition_t *p, unsigned int clv_index, int scaler_index,
                                      const unsigned int *freqs_indices, double *persite_lnl)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  if (!x || !x->ctx)
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "pll_compute_root_loglikelihood: no MI355X context behind this partition; this library has no CPU path");
    return fail_lnl("pll_compute_root_loglikelihood");
  }
  pll_tip_densify(p, clv_index); /* a root at a compact tip needs the dense CLV */
  if (clv_index >= p->nodes || scaler_index >= (int)p->scale_buffers || pll_is_pattern_tip(p, clv_index))
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_compute_root_loglikelihood: invalid CLV/scaler index");
    return fail_lnl("pll_compute_root_loglikelihood");
  }
  if (!pll_flush_model(p, x) || !prepare_end(p, x, clv_index, scaler_index))
    return fail_lnl("pll_compute_root_loglikelihood");
  const unsigned int gather = pll_repeats_enabled(p) && p->repeats->pernode_ids[clv_index];
  double lnl = 0;
  if (pllgpu_root_loglikelihood(x->ctx, clv_index, scaler_index, gather, freqs_indices, persite_lnl, &lnl) != 0)
  {
    pll_set_gpu_error("pll_compute_root_loglikelihood");
    return -INFINITY;
  }
  if (p->attributes & PLL_ATTRIB_AB_MASK)
  {
    pllgpu_edge_t e;
    memset(&e, 0, sizeof e);
    e.parent_clv = clv_index;
    e.parent_scaler = scaler_index;
    e.freqs_indices = freqs_indices;
    lnl += asc_correction(p, x, &e, 1);
  }
  return lnl;
}
