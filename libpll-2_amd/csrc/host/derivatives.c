/* derivatives.c - pll_update_sumtable / pll_compute_likelihood_derivatives (SURVEY.md section 8
 * row f1; reference: src/derivatives.c:239-418 over src/core_derivatives.c).
 *
 * Case selection as in the reference (:264-321): site repeats on either end -> gather maps;
 * PATTERN_TIP with one tip end -> the tip is read as codes and plays the "left" role (:67-79);
 * tip-tip is refused (:278-286). The table is computed into HBM by the CLV-update kernels with the
 * two contraction matrices
 *     M1[k][j][i] = pi_i * inv_eigenvecs[i][j]      (left end)
 *     M2[k][j][i] = eigenvecs[j][i]                 (right end)
 * per rate category k = params_indices[k]; they are rebuilt and uploaded only when an
 * eigensystem, a frequency vector or params_indices changed. The caller's `sumtable` pointer is a
 * handle to one of PLLGPU_SUMTABLE_SLOTS device tables (allocated on first use; beyond that the least
 * recently used is recycled and its handle remembered: an evaluation on a recycled handle fails
 * loudly instead of reading the caller's never-written buffer). pll_gpu_release_sumtable() gives a
 * table's HBM back.
 */
#include <math.h>

#include "pll_internal.h"

static int fail_loudly(const char *what)
{
  fprintf(stderr, "libpll_amd: %s: [%d] %s\n", what, pll_errno, pll_errmsg);
  return PLL_FAILURE;
}

static int was_evicted(const pll_amd_ext_t *x, const double *key)
{
  int i;
  for (i = 0; i < PLLGPU_SUMTABLE_SLOTS; ++i)
    if (x->sumtable_evicted[i] == key) return 1;
  return 0;
}

static int slot_of(pll_amd_ext_t *x, const double *key, int create)
{
  int i, victim = 0;
  for (i = 0; i < PLLGPU_SUMTABLE_SLOTS; ++i)
    if (x->sumtable_key[i] == key)
    {
      x->sumtable_age[i] = ++x->sumtable_clock;
      return i;
    }
  if (!create) return -1;
  for (i = 0; i < PLLGPU_SUMTABLE_SLOTS; ++i)
  {
    if (!x->sumtable_key[i])
    {
      victim = i;
      break;
    }
    if (x->sumtable_age[i] < x->sumtable_age[victim]) victim = i;
  }
  if (x->sumtable_key[victim])
  {
    x->sumtable_evicted[x->sumtable_evicted_next] = x->sumtable_key[victim];
    x->sumtable_evicted_next = (x->sumtable_evicted_next + 1u) % PLLGPU_SUMTABLE_SLOTS;
  }
  for (i = 0; i < PLLGPU_SUMTABLE_SLOTS; ++i) /* the handle is live again */
    if (x->sumtable_evicted[i] == key) x->sumtable_evicted[i] = NULL;
  x->sumtable_key[victim] = key;
  x->sumtable_age[victim] = ++x->sumtable_clock;
  return victim;
}

/* eigensystems, category rates, prop_invar, frequencies: whatever is stale */
static int flush_deriv_model(pll_partition_t *p, pll_amd_ext_t *x) { return pll_flush_eigen(p, x); }

static int flush_aux_matrices(pll_partition_t *p, pll_amd_ext_t *x, const unsigned int *params_indices)
{
  const unsigned int s = p->states, sp = p->states_padded, r = p->rate_cats;
  unsigned int k, i, j;
  int same = (x->aux_version == x->eigen_version) && !x->always_upload;
  for (k = 0; same && k < r; ++k) same = (x->aux_params[k] == params_indices[k]);
  if (same) return PLL_SUCCESS;

  double *m = (double *)calloc((size_t)2 * r * s * sp, sizeof(double));
  if (!m)
  {
    pll_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory.");
    return PLL_FAILURE;
  }
  double *m1 = m, *m2 = m + (size_t)r * s * sp;
  for (k = 0; k < r; ++k)
  {
    const double *ev = p->eigenvecs[params_indices[k]];
    const double *iev = p->inv_eigenvecs[params_indices[k]];
    const double *pi = p->frequencies[params_indices[k]];
    for (j = 0; j < s; ++j)
      for (i = 0; i < s; ++i)
      {
        m1[((size_t)k * s + j) * sp + i] = pi[i] * iev[(size_t)i * sp + j];
        m2[((size_t)k * s + j) * sp + i] = ev[(size_t)j * sp + i];
      }
  }
  int rc = pllgpu_aux_matrix_upload(x->ctx, 0, m1) || pllgpu_aux_matrix_upload(x->ctx, 1, m2);
  free(m);
  if (rc)
  {
    pll_set_gpu_error("sumtable matrices upload");
    return PLL_FAILURE;
  }
  x->aux_version = x->eigen_version;
  memcpy(x->aux_params, params_indices, sizeof(unsigned int) * r);
  return PLL_SUCCESS;
}

int pll_update_sumtable(pll_partition_t *p, unsigned int parent_clv_index, unsigned int child_clv_index,
                        int parent_scaler_index, int child_scaler_index,
                        const unsigned int *params_indices, double *sumtable)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  unsigned int k;
  if (!x || !x->ctx)
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "pll_update_sumtable: no MI355X context behind this partition; this library has no CPU path");
    return fail_loudly("pll_update_sumtable");
  }
  if (parent_clv_index >= p->nodes || child_clv_index >= p->nodes || parent_scaler_index >= (int)p->scale_buffers ||
      child_scaler_index >= (int)p->scale_buffers || !sumtable)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_update_sumtable: index out of range");
    return fail_loudly("pll_update_sumtable");
  }
  if (pll_tip_by_codes(p, parent_clv_index) && pll_tip_by_codes(p, child_clv_index))
  {
    if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
    {
      pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_update_sumtable() was called for the tip-tip case!");
      return PLL_FAILURE;
    }
    pll_tip_densify(p, parent_clv_index);
  }
  const int ptip = pll_tip_by_codes(p, parent_clv_index);
  const int ctip = pll_tip_by_codes(p, child_clv_index);
  for (k = 0; k < p->rate_cats; ++k)
  {
    if (params_indices[k] >= p->rate_matrices)
    {
      pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_update_sumtable: params_indices[%u] out of range", k);
      return fail_loudly("pll_update_sumtable");
    }
    /* the reference reads whatever eigensystem is stored; an invalid one is computed here */
    if (!p->eigen_decomp_valid[params_indices[k]] && !pll_update_eigen(p, params_indices[k])) return PLL_FAILURE;
  }
  if (!flush_deriv_model(p, x) || !flush_aux_matrices(p, x, params_indices)) return fail_loudly("pll_update_sumtable");
  if (!pll_flush_clv(p, x, parent_clv_index) || !pll_flush_clv(p, x, child_clv_index) ||
      (!ptip && !pll_flush_scaler(p, x, parent_scaler_index)) || (!ctip && !pll_flush_scaler(p, x, child_scaler_index)) ||
      !pll_flush_repeats(p, x, parent_clv_index) || !pll_flush_repeats(p, x, child_clv_index))
    return fail_loudly("pll_update_sumtable");

  pllgpu_sumtable_t st;
  memset(&st, 0, sizeof st);
  /* a tip given by codes is the left end; otherwise parent left, child right (:142-153) */
  st.left_clv = ctip ? child_clv_index : parent_clv_index;
  st.right_clv = ctip ? parent_clv_index : child_clv_index;
  st.left_scaler = (ptip || ctip) ? PLL_SCALE_BUFFER_NONE : parent_scaler_index;
  st.right_scaler = ctip ? parent_scaler_index : child_scaler_index;
  st.left_is_tip = (ptip || ctip);
  st.gather = pll_repeats_enabled(p) &&
              (p->repeats->pernode_ids[parent_clv_index] || p->repeats->pernode_ids[child_clv_index]);
  const int slot = slot_of(x, sumtable, 1);
  if (pllgpu_update_sumtable(x->ctx, &st, (unsigned)slot) != 0)
  {
    x->sumtable_key[slot] = NULL;
    pll_set_gpu_error("pll_update_sumtable");
    return PLL_FAILURE;
  }
  if (x->eager_mirror && pllgpu_sumtable_download(x->ctx, (unsigned)slot, sumtable) != 0)
  {
    pll_set_gpu_error("pll_update_sumtable (mirror)");
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

int pll_gpu_sync_sumtable(pll_partition_t *p, double *sumtable)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  if (!x || !x->ctx)
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "pll_gpu_sync_sumtable: no MI355X context behind this partition");
    return fail_loudly("pll_gpu_sync_sumtable");
  }
  const int slot = slot_of(x, sumtable, 0);
  if (slot < 0)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_sync_sumtable: no device table stands for this buffer");
    return PLL_FAILURE;
  }
  if (pllgpu_sumtable_download(x->ctx, (unsigned)slot, sumtable) != 0)
  {
    pll_set_gpu_error("pll_gpu_sync_sumtable");
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

int pll_gpu_release_sumtable(pll_partition_t *p, const double *sumtable)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  int i;
  if (!x || !x->ctx)
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "pll_gpu_release_sumtable: no MI355X context behind this partition");
    return fail_loudly("pll_gpu_release_sumtable");
  }
  for (i = 0; i < PLLGPU_SUMTABLE_SLOTS; ++i)
    if (x->sumtable_evicted[i] == sumtable) x->sumtable_evicted[i] = NULL;
  const int slot = slot_of(x, sumtable, 0);
  if (slot < 0) return PLL_SUCCESS; /* nothing on the device stands for it (any more) */
  x->sumtable_key[slot] = NULL;
  if (pllgpu_sumtable_release(x->ctx, (unsigned)slot) != 0)
  {
    pll_set_gpu_error("pll_gpu_release_sumtable");
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

int pll_compute_likelihood_derivatives(pll_partition_t *p, int parent_scaler_index, int child_scaler_index,
                                       double branch_length, const unsigned int *params_indices,
                                       const double *sumtable, double *d_f, double *dd_f)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  const int asc_type = (int)(p->attributes & PLL_ATTRIB_AB_MASK);
  if (!x || !x->ctx)
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "pll_compute_likelihood_derivatives: no MI355X context behind this partition; this library has no CPU path");
    return fail_loudly("pll_compute_likelihood_derivatives");
  }
  if (!sumtable || !d_f || !dd_f || !(branch_length >= 0))
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_compute_likelihood_derivatives: invalid argument");
    return fail_loudly("pll_compute_likelihood_derivatives");
  }
  if (!flush_deriv_model(p, x)) return fail_loudly("pll_compute_likelihood_derivatives");
  int slot = slot_of(x, sumtable, 0);
  if (slot < 0)
  {
    if (was_evicted(x, sumtable))
    {
      pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_compute_likelihood_derivatives: the device table of this sumtable was recycled "
                    "(more than %d live sumtables in one partition); call pll_update_sumtable again", PLLGPU_SUMTABLE_SLOTS);
      return fail_loudly("pll_compute_likelihood_derivatives");
    }
    /* a table this library did not produce: the caller's buffer is the truth */
    slot = slot_of(x, sumtable, 1);
    if (pllgpu_sumtable_upload(x->ctx, (unsigned)slot, sumtable) != 0)
    {
      x->sumtable_key[slot] = NULL;
      pll_set_gpu_error("pll_compute_likelihood_derivatives (table upload)");
      return PLL_FAILURE;
    }
  }
  /* Stamatakis: the per-state extra entries are ordinary weighted sites (src/core_derivatives.c:733-742) */
  const unsigned int eval_sites = p->sites + (asc_type == PLL_ATTRIB_AB_STAMATAKIS ? p->states : 0);
  if (pllgpu_likelihood_derivatives(x->ctx, (unsigned)slot, branch_length, params_indices, eval_sites, d_f, dd_f) != 0)
  {
    pll_set_gpu_error("pll_compute_likelihood_derivatives");
    return PLL_FAILURE;
  }
  if (asc_type && asc_type != PLL_ATTRIB_AB_STAMATAKIS)
  {
    /* Lewis / Felsenstein (src/core_derivatives.c:851-924): (L, L', L'') of the extra entries with
     * their scaling undone, summed over the states. pattern_weight_sum stands for the reference's
     * on-the-spot sum of pattern_weights[0..sites) (:900-902): equal whenever the weights were set
     * through pll_set_pattern_weights. */
    double lk[3 * 64], asc[3] = {0.0, 0.0, 0.0};
    unsigned int sc[64], n, sum_w_inv = 0;
    if (parent_scaler_index >= (int)p->scale_buffers || child_scaler_index >= (int)p->scale_buffers ||
        !pll_flush_scaler(p, x, parent_scaler_index) || !pll_flush_scaler(p, x, child_scaler_index))
      return fail_loudly("pll_compute_likelihood_derivatives");
    if (pllgpu_asc_derivative_terms(x->ctx, (unsigned)slot, parent_scaler_index, child_scaler_index, params_indices, lk, sc) != 0)
    {
      pll_set_gpu_error("pll_compute_likelihood_derivatives (ascertainment terms)");
      return PLL_FAILURE;
    }
    for (n = 0; n < p->states; ++n)
    {
      const double f = pow(PLL_SCALE_THRESHOLD, (double)sc[n]);
      asc[0] += lk[3 * n + 0] * f;
      asc[1] += lk[3 * n + 1] * f;
      asc[2] += lk[3 * n + 2] * f;
      sum_w_inv += p->pattern_weights[p->sites + n];
    }
    if (asc_type == PLL_ATTRIB_AB_LEWIS)
    {
      const double w = (double)p->pattern_weight_sum;
      *d_f += w * (asc[1] / (asc[0] - 1.0));
      *dd_f += w * (((asc[0] - 1.0) * asc[2] - asc[1] * asc[1]) / ((asc[0] - 1.0) * (asc[0] - 1.0)));
    }
    else if (asc_type == PLL_ATTRIB_AB_FELSENSTEIN)
    {
      *d_f -= sum_w_inv * (asc[1] / asc[0]);
      *dd_f -= sum_w_inv * (((asc[2] * asc[0]) - asc[1] * asc[1]) / (asc[0] * asc[0]));
    }
    else
    {
      pll_set_error(PLL_ERROR_AB_INVALIDMETHOD, "Illegal ascertainment bias algorithm");
      return PLL_FAILURE;
    }
  }
  return PLL_SUCCESS;
}
