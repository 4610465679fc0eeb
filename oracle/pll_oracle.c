/*
 * pll_oracle.c - CPU restatement of the libpll-2 partial-likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see pll_oracle.h). Parity status: PINNED against the reference
 * (oracle/_ref) and the golden vectors in tests/golden/.
 *
 * Written from the mathematics of SURVEY.md section 8a and the generic-C bodies of the
 * reference it cites; one routine covers every child-kind combination instead of the reference's
 * one-function-per-case layout. Deliberately scalar and slow.
 */
#include "pll_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_SCALE_FACTOR 0x1p256     /* src/pll.h:96 */
#define ORC_SCALE_THRESHOLD 0x1p-256 /* src/pll.h:97 */
#define ORC_RATE_MAXDIFF 4           /* src/pll.h:104 */

static unsigned int entry_of(const unsigned int *site_id, unsigned int site)
{
  return site_id ? site_id[site] : site; /* PLL_GET_ID, src/pll.h:682 */
}

/* conditional likelihood of child state j at (entry, rate): a CLV value, or the 0/1 indicator
 * of a tip code (src/core_partials.c:304-312: bit j of the code selects column j) */
static orc_state_t tip_mask(const orc_child_t *c, unsigned int entry)
{
  unsigned int code = c->tipchars[entry];
  return c->tipmap ? c->tipmap[code] : (orc_state_t)code;
}

/* sum_j P[i][j] * x[j] for one child, one (entry, rate), all parent states i */
static void branch_term(const orc_child_t *c, unsigned int entry, unsigned int rate,
                        const double *pmat_rate, unsigned int states, unsigned int sp,
                        unsigned int span, double *out)
{
  unsigned int i, j;
  if (c->clv)
  {
    const double *x = c->clv + (size_t)entry * span + (size_t)rate * sp;
    for (i = 0; i < states; ++i)
    {
      double t = 0;
      for (j = 0; j < states; ++j) t += pmat_rate[(size_t)i * sp + j] * x[j];
      out[i] = t;
    }
  }
  else
  {
    orc_state_t m = tip_mask(c, entry);
    for (i = 0; i < states; ++i)
    {
      double t = 0;
      orc_state_t b = m;
      for (j = 0; j < states; ++j, b >>= 1)
        if (b & 1) t += pmat_rate[(size_t)i * sp + j];
      out[i] = t;
    }
  }
}

void orc_update_partial(unsigned int states, unsigned int sp, unsigned int rate_cats,
                        unsigned int parent_entries, double *parent_clv,
                        unsigned int *parent_scaler, const unsigned int *parent_id_site,
                        const orc_child_t *left, const double *left_matrix,
                        const orc_child_t *right, const double *right_matrix, int scale_mode)
{
  const unsigned int span = rate_cats * sp;
  double *ta = (double *)malloc(sizeof(double) * states);
  double *tb = (double *)malloc(sizeof(double) * states);
  unsigned int n, k, i;

  if (!parent_scaler) scale_mode = 0;

  for (n = 0; n < parent_entries; ++n)
  {
    const unsigned int site = parent_id_site ? parent_id_site[n] : n; /* PLL_GET_SITE */
    const unsigned int le = entry_of(left->site_id, site);
    const unsigned int re = entry_of(right->site_id, site);
    double *p = parent_clv + (size_t)n * span;
    int site_small = 1;

    /* parent scaler starts as the sum of the children's (src/core_partials.c:24-46,
     * src/repeats.c:392-540) */
    if (scale_mode == 1)
      parent_scaler[n] = (left->scaler ? left->scaler[le] : 0) +
                         (right->scaler ? right->scaler[re] : 0);
    else if (scale_mode == 2)
      for (k = 0; k < rate_cats; ++k)
        parent_scaler[(size_t)n * rate_cats + k] =
            (left->scaler ? left->scaler[(size_t)le * rate_cats + k] : 0) +
            (right->scaler ? right->scaler[(size_t)re * rate_cats + k] : 0);

    for (k = 0; k < rate_cats; ++k)
    {
      int rate_small = 1;
      branch_term(left, le, k, left_matrix + (size_t)k * states * sp, states, sp, span, ta);
      branch_term(right, re, k, right_matrix + (size_t)k * states * sp, states, sp, span, tb);
      for (i = 0; i < states; ++i)
      {
        double v = ta[i] * tb[i];
        p[(size_t)k * sp + i] = v;
        rate_small &= (v < ORC_SCALE_THRESHOLD);
      }
      for (i = states; i < sp; ++i) p[(size_t)k * sp + i] = 0.0; /* padding lanes: defined */
      if (scale_mode == 2 && rate_small)
      {
        for (i = 0; i < states; ++i) p[(size_t)k * sp + i] *= ORC_SCALE_FACTOR;
        parent_scaler[(size_t)n * rate_cats + k] += 1;
      }
      site_small &= rate_small;
    }
    if (scale_mode == 1 && site_small)
    {
      for (k = 0; k < rate_cats; ++k)
        for (i = 0; i < states; ++i) p[(size_t)k * sp + i] *= ORC_SCALE_FACTOR;
      parent_scaler[n] += 1;
    }
  }
  free(ta);
  free(tb);
}

/* scale_minlh[d-1] = 2^(-256 d), d = 1..4 (src/core_likelihood.c:1366-1375) */
static void fill_minlh(double *m)
{
  double f = 1.0;
  int i;
  for (i = 0; i < ORC_RATE_MAXDIFF; ++i)
  {
    f *= ORC_SCALE_THRESHOLD;
    m[i] = f;
  }
}

/* combine the two ends' scalers for one site; returns the common per-site count and fills the
 * capped per-rate excess (src/core_likelihood.c:1390-1414) */
static unsigned int site_scalings(const orc_child_t *a, unsigned int ae, const orc_child_t *b,
                                  unsigned int be, unsigned int rate_cats, int per_rate,
                                  unsigned int *excess)
{
  unsigned int k, s;
  if (!per_rate)
  {
    s = (a && a->scaler) ? a->scaler[ae] : 0;
    s += (b && b->scaler) ? b->scaler[be] : 0;
    return s;
  }
  s = UINT_MAX;
  for (k = 0; k < rate_cats; ++k)
  {
    excess[k] = (a && a->scaler) ? a->scaler[(size_t)ae * rate_cats + k] : 0;
    excess[k] += (b && b->scaler) ? b->scaler[(size_t)be * rate_cats + k] : 0;
    if (excess[k] < s) s = excess[k];
  }
  for (k = 0; k < rate_cats; ++k)
  {
    excess[k] -= s;
    if (excess[k] > ORC_RATE_MAXDIFF) excess[k] = ORC_RATE_MAXDIFF;
  }
  return s;
}

/* log of the site likelihood with the scaling undone (src/core_likelihood.c:1462-1481) */
static double finish_site(double terma, double terminv, unsigned int scalings,
                          const double *minlh)
{
  if (scalings)
  {
    if (terminv > 0.)
    {
      unsigned int c = scalings < ORC_RATE_MAXDIFF ? scalings : ORC_RATE_MAXDIFF;
      return log(terma * minlh[c - 1] + terminv);
    }
    return log(terma) + scalings * log(ORC_SCALE_THRESHOLD);
  }
  return log(terma + terminv);
}

double orc_edge_loglikelihood(unsigned int states, unsigned int sp, unsigned int rate_cats,
                              unsigned int sites, const orc_child_t *parent,
                              const orc_child_t *child, const double *pmatrix,
                              const double *const *frequencies, const double *rate_weights,
                              const unsigned int *pattern_weights, const double *prop_invar,
                              const int *invariant, const unsigned int *freqs_indices,
                              double *persite_lnl, int per_rate)
{
  const unsigned int span = rate_cats * sp;
  double minlh[ORC_RATE_MAXDIFF];
  unsigned int *excess = (unsigned int *)calloc(rate_cats ? rate_cats : 1, sizeof(unsigned int));
  double *tb = (double *)malloc(sizeof(double) * states);
  double logl = 0;
  unsigned int n, k, i;

  fill_minlh(minlh);

  for (n = 0; n < sites; ++n)
  {
    const unsigned int pe = entry_of(parent->site_id, n);
    const unsigned int ce = entry_of(child->site_id, n);
    const unsigned int scal = site_scalings(parent, pe, child, ce, rate_cats, per_rate, excess);
    double terma = 0, terminv = 0, site_lk;

    for (k = 0; k < rate_cats; ++k)
    {
      const double *freqs = frequencies[freqs_indices[k]];
      const double pinv = prop_invar ? prop_invar[freqs_indices[k]] : 0;
      const double *xp = parent->clv + (size_t)pe * span + (size_t)k * sp;
      double terma_r = 0;

      branch_term(child, ce, k, pmatrix + (size_t)k * states * sp, states, sp, span, tb);
      for (i = 0; i < states; ++i) terma_r += xp[i] * freqs[i] * tb[i];

      if (per_rate && excess[k] > 0) terma_r *= minlh[excess[k] - 1];

      if (pinv > 0)
      {
        terma += rate_weights[k] * terma_r * (1. - pinv);
        if (invariant && invariant[n] != -1)
          terminv += rate_weights[k] * freqs[invariant[n]] * pinv;
      }
      else
        terma += terma_r * rate_weights[k];
    }

    site_lk = finish_site(terma, terminv, scal, minlh) * pattern_weights[n];
    if (persite_lnl) persite_lnl[n] = site_lk;
    logl += site_lk;
  }
  free(excess);
  free(tb);
  return logl;
}

double orc_root_loglikelihood(unsigned int states, unsigned int sp, unsigned int rate_cats,
                              unsigned int sites, const orc_child_t *node,
                              const double *const *frequencies, const double *rate_weights,
                              const unsigned int *pattern_weights, const double *prop_invar,
                              const int *invariant, const unsigned int *freqs_indices,
                              double *persite_lnl, int per_rate)
{
  const unsigned int span = rate_cats * sp;
  double minlh[ORC_RATE_MAXDIFF];
  unsigned int *excess = (unsigned int *)calloc(rate_cats ? rate_cats : 1, sizeof(unsigned int));
  double logl = 0;
  unsigned int n, k, i;

  fill_minlh(minlh);

  for (n = 0; n < sites; ++n)
  {
    const unsigned int e = entry_of(node->site_id, n);
    const unsigned int scal = site_scalings(node, e, NULL, 0, rate_cats, per_rate, excess);
    double term = 0, site_lk;

    for (k = 0; k < rate_cats; ++k)
    {
      const double *freqs = frequencies[freqs_indices[k]];
      const double pinv = prop_invar ? prop_invar[freqs_indices[k]] : 0;
      const double *x = node->clv + (size_t)e * span + (size_t)k * sp;
      double term_r = 0;
      for (i = 0; i < states; ++i) term_r += x[i] * freqs[i];
      if (per_rate && excess[k] > 0) term_r *= minlh[excess[k] - 1];
      /* src/core_likelihood.c:176-188: the invariant share is mixed in before the log and is
       * NOT protected from the scaler (unlike the edge routine) */
      if (pinv > 0)
      {
        double inv = (invariant && invariant[n] != -1) ? freqs[invariant[n]] : 0;
        term += rate_weights[k] * (term_r * (1 - pinv) + inv * pinv);
      }
      else
        term += term_r * rate_weights[k];
    }
    site_lk = log(term);
    if (scal) site_lk += scal * log(ORC_SCALE_THRESHOLD);
    site_lk *= pattern_weights[n];
    if (persite_lnl) persite_lnl[n] = site_lk;
    logl += site_lk;
  }
  free(excess);
  return logl;
}

unsigned int orc_repeat_classes(unsigned int sites, const unsigned int *site_id_left,
                                unsigned int ids_left, const unsigned int *site_id_right,
                                unsigned int ids_right, unsigned int *site_id_parent,
                                unsigned int *id_site_parent)
{
  /* direct-address table over (left id, right id), first occurrence gets the next class number
   * (src/repeats.c:334-347) */
  const size_t cells = (size_t)ids_left * ids_right;
  unsigned int *table = (unsigned int *)malloc(sizeof(unsigned int) * (cells ? cells : 1));
  unsigned int next = 0, s;
  memset(table, 0xFF, sizeof(unsigned int) * (cells ? cells : 1));
  for (s = 0; s < sites; ++s)
  {
    size_t cell = (size_t)site_id_left[s] + (size_t)site_id_right[s] * ids_left;
    if (table[cell] == UINT_MAX)
    {
      id_site_parent[next] = s;
      table[cell] = next++;
    }
    site_id_parent[s] = table[cell];
  }
  free(table);
  return next;
}

/* ---- branch-length derivatives -------------------------------------- */

/* sumtable[n][k][j] = (sum_i p_i pi_i Vinv[i][j]) * (sum_i V[j][i] c_i), times the capped per-rate
 * scaler excess (src/core_derivatives.c:417-465 ii, :587-635 ti, :215-319 repeats). The arrays
 * follow the reference's naming: eigenvecs[j*sp+i], inv_eigenvecs[i*sp+j], one set per category. */
void orc_update_sumtable(unsigned int states, unsigned int sp, unsigned int rate_cats, unsigned int sites,
                         const orc_child_t *parent, const orc_child_t *child,
                         const double *const *eigenvecs, const double *const *inv_eigenvecs,
                         const double *const *freqs, double *sumtable, int per_rate)
{
  const unsigned int span = rate_cats * sp;
  unsigned int *excess = (unsigned int *)calloc(rate_cats ? rate_cats : 1, sizeof(unsigned int));
  double minlh[ORC_RATE_MAXDIFF];
  unsigned int n, k, i, j;
  fill_minlh(minlh);
  for (n = 0; n < sites; ++n)
  {
    const unsigned int pe = entry_of(parent->site_id, n);
    const unsigned int ce = entry_of(child->site_id, n);
    if (per_rate) (void)site_scalings(parent, pe, child, ce, rate_cats, 1, excess);
    for (k = 0; k < rate_cats; ++k)
    {
      const double *V = eigenvecs[k], *Vi = inv_eigenvecs[k], *pi = freqs[k];
      double *sum = sumtable + (size_t)n * span + (size_t)k * sp;
      orc_state_t pm = parent->clv ? 0 : tip_mask(parent, pe);
      orc_state_t cm = child->clv ? 0 : tip_mask(child, ce);
      const double *xp = parent->clv ? parent->clv + (size_t)pe * span + (size_t)k * sp : NULL;
      const double *xc = child->clv ? child->clv + (size_t)ce * span + (size_t)k * sp : NULL;
      for (j = 0; j < states; ++j)
      {
        double l = 0, r = 0;
        for (i = 0; i < states; ++i)
        {
          const double p = xp ? xp[i] : (double)((pm >> i) & 1);
          const double c = xc ? xc[i] : (double)((cm >> i) & 1);
          l += p * pi[i] * Vi[(size_t)i * sp + j];
          r += V[(size_t)j * sp + i] * c;
        }
        sum[j] = l * r;
        if (per_rate && excess[k] > 0) sum[j] *= minlh[excess[k] - 1];
      }
      for (j = states; j < sp; ++j) sum[j] = 0.0;
    }
  }
  free(excess);
}

/* first and second derivative of -lnL with respect to the branch length
 * (src/core_derivatives.c:643-694, :757-772, :825-848) */
void orc_likelihood_derivatives(unsigned int states, unsigned int sp, unsigned int rate_cats,
                                unsigned int sites, const double *rate_weights, const int *invariant,
                                const unsigned int *pattern_weights, double branch_length,
                                const double *prop_invar /* per category */,
                                const double *const *freqs, const double *rates,
                                const double *const *eigenvals, const double *sumtable, double *d_f,
                                double *dd_f)
{
  double *diag = (double *)malloc(sizeof(double) * 3 * rate_cats * states);
  unsigned int n, k, j;
  *d_f = 0;
  *dd_f = 0;
  for (k = 0; k < rate_cats; ++k)
  {
    const double ki = rates[k] / (1.0 - prop_invar[k]);
    for (j = 0; j < states; ++j)
    {
      double *d = diag + 3 * ((size_t)k * states + j);
      d[0] = exp(eigenvals[k][j] * ki * branch_length);
      d[1] = eigenvals[k][j] * ki * d[0];
      d[2] = eigenvals[k][j] * ki * eigenvals[k][j] * ki * d[0];
    }
  }
  for (n = 0; n < sites; ++n)
  {
    double lk[3] = {0, 0, 0};
    for (k = 0; k < rate_cats; ++k)
    {
      const double *sum = sumtable + ((size_t)n * rate_cats + k) * sp;
      double c[3] = {0, 0, 0};
      for (j = 0; j < states; ++j)
      {
        const double *d = diag + 3 * ((size_t)k * states + j);
        c[0] += sum[j] * d[0];
        c[1] += sum[j] * d[1];
        c[2] += sum[j] * d[2];
      }
      if (prop_invar[k] > 0)
      {
        const double inv = (invariant && invariant[n] != -1) ? freqs[k][invariant[n]] * prop_invar[k] : 0;
        c[0] = c[0] * (1. - prop_invar[k]) + inv;
        c[1] = c[1] * (1. - prop_invar[k]);
        c[2] = c[2] * (1. - prop_invar[k]);
      }
      lk[0] += c[0] * rate_weights[k];
      lk[1] += c[1] * rate_weights[k];
      lk[2] += c[2] * rate_weights[k];
    }
    {
      const double d1 = -lk[1] / lk[0];
      const double d2 = d1 * d1 - lk[2] / lk[0];
      *d_f += pattern_weights[n] * d1;
      *dd_f += pattern_weights[n] * d2;
    }
  }
  free(diag);
}

/* ---- ascertainment-bias correction ------------------------------------------------------------
 * src/likelihood.c:24-48 (formula), :50-120 (root), :191-268 (tip-inner), :342-440 (inner-inner):
 * the per-state extra entries first + n of both ends, state n's weight asc_weights[n]. The
 * tip-inner variant multiplies by P[j][n] because the tip's extra entry n is state n (:232); here
 * that falls out of the tip's code at the extra entry. Per-rate scalers: the reference indexes the
 * [entry][rate] array per site there; this restatement brings the rates to the smallest count like
 * the main routine (identical whenever the extra entries were never rescaled). */
double orc_asc_bias_correction(unsigned int states, unsigned int sp, unsigned int rate_cats,
                               unsigned int first, const orc_child_t *parent, const orc_child_t *child,
                               const double *pmatrix, const double *const *frequencies,
                               const double *rate_weights, const unsigned int *asc_weights,
                               unsigned int pattern_weight_sum, const unsigned int *freqs_indices,
                               int asc_type, int per_rate)
{
  const unsigned int span = rate_cats * sp;
  double minlh[ORC_RATE_MAXDIFF];
  unsigned int *excess = (unsigned int *)calloc(rate_cats ? rate_cats : 1, sizeof(unsigned int));
  double *tb = (double *)malloc(sizeof(double) * states);
  double base = 0;
  unsigned int n, k, i, sum_w_inv = 0;
  fill_minlh(minlh);
  for (n = 0; n < states; ++n)
  {
    const unsigned int e = first + n;
    const unsigned int scal = site_scalings(parent, e, child, e, rate_cats, per_rate, excess);
    double terma = 0, site_lk;
    for (k = 0; k < rate_cats; ++k)
    {
      const double *freqs = frequencies[freqs_indices[k]];
      const double *xp = parent->clv + (size_t)e * span + (size_t)k * sp;
      double terma_r = 0;
      if (child)
        branch_term(child, e, k, pmatrix + (size_t)k * states * sp, states, sp, span, tb);
      else
        for (i = 0; i < states; ++i) tb[i] = 1.0;
      for (i = 0; i < states; ++i) terma_r += xp[i] * freqs[i] * tb[i];
      if (per_rate && excess[k] > 0) terma_r *= minlh[excess[k] - 1];
      terma += terma_r * rate_weights[k];
    }
    sum_w_inv += asc_weights[n];
    if (asc_type == 3)
    {
      site_lk = log(terma) * asc_weights[n];
      if (scal) site_lk += scal * log(ORC_SCALE_THRESHOLD); /* unweighted, as in :98-100 */
    }
    else
      site_lk = terma * pow(ORC_SCALE_THRESHOLD, (double)scal);
    base += site_lk;
  }
  free(excess);
  free(tb);
  switch (asc_type)
  {
    case 1: return -(pattern_weight_sum * log(1 - base));
    case 2: return sum_w_inv * log(base);
    case 3: return base;
  }
  return -INFINITY;
}

/* src/core_derivatives.c:851-924 (Lewis, Felsenstein; Stamatakis is the ordinary loop over
 * sites + states entries, :733-742): contributions to d_f / dd_f from the extra table entries. */
void orc_asc_bias_derivatives(unsigned int states, unsigned int sp, unsigned int rate_cats, unsigned int first,
                              const unsigned int *parent_scaler, const unsigned int *child_scaler,
                              const double *rate_weights, const unsigned int *asc_weights,
                              unsigned int pattern_weight_sum, double branch_length, const double *prop_invar,
                              const double *rates, const double *const *eigenvals, const double *sumtable,
                              int asc_type, int per_rate, double *d_f, double *dd_f)
{
  double asc[3] = {0, 0, 0};
  unsigned int n, k, j, sum_w_inv = 0;
  for (n = 0; n < states; ++n)
  {
    double lk[3] = {0, 0, 0}, f;
    unsigned int scal;
    for (k = 0; k < rate_cats; ++k)
    {
      const double *sum = sumtable + ((size_t)(first + n) * rate_cats + k) * sp;
      const double ki = rates[k] / (1.0 - prop_invar[k]);
      double c[3] = {0, 0, 0};
      for (j = 0; j < states; ++j)
      {
        const double e = exp(eigenvals[k][j] * ki * branch_length);
        c[0] += sum[j] * e;
        c[1] += sum[j] * (eigenvals[k][j] * ki * e);
        c[2] += sum[j] * (eigenvals[k][j] * ki * eigenvals[k][j] * ki * e);
      }
      lk[0] += c[0] * rate_weights[k];
      lk[1] += c[1] * rate_weights[k];
      lk[2] += c[2] * rate_weights[k];
    }
    if (per_rate)
    {
      scal = UINT_MAX;
      for (k = 0; k < rate_cats; ++k)
      {
        unsigned int s = (parent_scaler ? parent_scaler[(size_t)(first + n) * rate_cats + k] : 0) +
                         (child_scaler ? child_scaler[(size_t)(first + n) * rate_cats + k] : 0);
        if (s < scal) scal = s;
      }
    }
    else
      scal = (parent_scaler ? parent_scaler[first + n] : 0) + (child_scaler ? child_scaler[first + n] : 0);
    f = pow(ORC_SCALE_THRESHOLD, (double)scal);
    asc[0] += lk[0] * f;
    asc[1] += lk[1] * f;
    asc[2] += lk[2] * f;
    sum_w_inv += asc_weights[n];
  }
  if (asc_type == 1)
  {
    *d_f += pattern_weight_sum * (asc[1] / (asc[0] - 1.0));
    *dd_f += pattern_weight_sum * (((asc[0] - 1.0) * asc[2] - asc[1] * asc[1]) / ((asc[0] - 1.0) * (asc[0] - 1.0)));
  }
  else if (asc_type == 2)
  {
    *d_f -= sum_w_inv * (asc[1] / asc[0]);
    *dd_f -= sum_w_inv * (((asc[2] * asc[0]) - asc[1] * asc[1]) / (asc[0] * asc[0]));
  }
}
