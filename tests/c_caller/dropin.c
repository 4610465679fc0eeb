/* A C caller of the drop-in boundary, written against include/pll_amd.h only: the call sequence of
 * the reference's own 5-taxon likelihood test (test/src/00010_NMDU_lkcalc.c:96-175: GTR + Gamma4,
 * 12 sites; expected log-likelihood -58.887310 on the edge (6,7), test/out/00010_NMDU_lkcalc.out),
 * then a branch-length derivative and a re-evaluation after changing that branch through the
 * model API. Built and run by tests/test_gpu_c_caller.py:
 *     gcc -O2 -Iinclude tests/c_caller/dropin.c -Llibpll-2_amd/csrc -lpll_amd -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#ifdef USE_REFERENCE_HEADER
#include "pll.h" /* the reference's own header: same source, same binary interface */
int pll_gpu_sync_clv(pll_partition_t *partition, unsigned int clv_index); /* the one extension used below */
#else
#include "pll_amd.h"
#endif

int main(void)
{
  const char *seq[5] = {"WAC-CTA-ATCT", "CCC-TTA-ATGT", "A-C-TAG-CTCT", "CTCTTAA-A-CG", "CAC-TCA-A-TG"};
  const double freqs[4] = {0.3, 0.4, 0.1, 0.2};
  const double subst[6] = {1, 2.5, 1, 1, 2.5, 1};
  const double brlen[4] = {0.1, 0.2, 1.0, 1.0};
  const unsigned int matrix_indices[4] = {0, 1, 2, 3};
  const unsigned int params[4] = {0, 0, 0, 0};
  double rates[4], persite[12], d_f, dd_f;
  unsigned int i;

  pll_partition_t *p = pll_partition_create(5, 4, 4, 12, 1, 5, 4, 0, PLL_ATTRIB_ARCH_AVX2);
  if (!p)
  {
    fprintf(stderr, "pll_partition_create: [%d] %s\n", pll_errno, pll_errmsg);
    return 2;
  }
  pll_set_frequencies(p, 0, freqs);
  pll_set_subst_params(p, 0, subst);
  pll_compute_gamma_cats(0.5, 4, rates, PLL_GAMMA_RATES_MEAN);
  pll_set_category_rates(p, rates);
  for (i = 0; i < 5; ++i)
    if (!pll_set_tip_states(p, i, pll_map_nt, seq[i])) return 3;
  if (!pll_update_prob_matrices(p, params, matrix_indices, brlen, 4)) return 4;

  pll_operation_t ops[3] = {
      {5, PLL_SCALE_BUFFER_NONE, 0, 1, PLL_SCALE_BUFFER_NONE, 1, 1, PLL_SCALE_BUFFER_NONE},
      {6, PLL_SCALE_BUFFER_NONE, 5, 0, PLL_SCALE_BUFFER_NONE, 2, 1, PLL_SCALE_BUFFER_NONE},
      {7, PLL_SCALE_BUFFER_NONE, 3, 1, PLL_SCALE_BUFFER_NONE, 4, 1, PLL_SCALE_BUFFER_NONE}};
  pll_update_partials(p, ops, 3);
  double lnl = pll_compute_edge_loglikelihood(p, 6, PLL_SCALE_BUFFER_NONE, 7, PLL_SCALE_BUFFER_NONE, 0, params, persite);
  double sum = 0;
  for (i = 0; i < 12; ++i) sum += persite[i];
  printf("lnl %.6f persite_sum %.6f\n", lnl, sum);

  /* Newton-style step on that edge: table, derivatives, new branch length, new matrix, new lnL */
  double *sumtable = (double *)pll_aligned_alloc(12 * 4 * p->states_padded * sizeof(double), p->alignment);
  if (!pll_update_sumtable(p, 6, 7, PLL_SCALE_BUFFER_NONE, PLL_SCALE_BUFFER_NONE, params, sumtable)) return 5;
  if (!pll_compute_likelihood_derivatives(p, PLL_SCALE_BUFFER_NONE, PLL_SCALE_BUFFER_NONE, brlen[0], params, sumtable, &d_f, &dd_f)) return 6;
  printf("d_f %.6e dd_f %.6e\n", d_f, dd_f);
  double t = brlen[0] - d_f / dd_f;
  if (!(t > 1e-6)) t = 1e-6;
  const unsigned int m0 = 0;
  pll_update_prob_matrices(p, params, &m0, &t, 1);
  double lnl2 = pll_compute_edge_loglikelihood(p, 6, PLL_SCALE_BUFFER_NONE, 7, PLL_SCALE_BUFFER_NONE, 0, params, NULL);
  printf("t %.6f lnl2 %.6f\n", t, lnl2);

  /* the host mirror on request */
  if (!pll_gpu_sync_clv(p, 6)) return 7;
  printf("clv6[0] %.6e\n", p->clv[6][0]);
  pll_aligned_free(sumtable);
  pll_partition_destroy(p);
  return (fabs(lnl + 58.887310) < 5.1e-7 && fabs(sum - lnl) < 1e-9 && lnl2 >= lnl - 1e-9) ? 0 : 1;
}
