/* output.c - pll_show_pmatrix / pll_show_clv (reference: src/output.c:26-101; src/pll.h:2590-2600), the
 * printers the reference's examples and tests call. The numbers live on the device: each call first
 * brings the host mirror of what it prints up to date, then prints it in the reference's format
 * (matrix rows "%+2.*f   " with a blank line per rate category; CLVs as
 * [ {(v,...,v),(...)} {...} ] with the scaling undone by repeated multiplication with 2^-256). */
#include "pll_internal.h"

void pll_show_pmatrix(const pll_partition_t *partition, unsigned int index, unsigned int float_precision)
{
  pll_partition_t *p = (pll_partition_t *)partition;
  if (!pll_gpu_sync_pmatrix(p, (int)index)) fprintf(stderr, "libpll_amd: pll_show_pmatrix: [%d] %s\n", pll_errno, pll_errmsg);
  const unsigned int s = p->states, sp = p->states_padded;
  for (unsigned int k = 0; k < p->rate_cats; ++k)
  {
    const double *m = p->pmatrix[index] + (size_t)k * s * sp;
    for (unsigned int i = 0; i < s; ++i)
    {
      for (unsigned int j = 0; j < s; ++j) printf("%+2.*f   ", (int)float_precision, m[(size_t)i * sp + j]);
      printf("\n");
    }
    printf("\n");
  }
}

void pll_show_clv(const pll_partition_t *partition, unsigned int clv_index, int scaler_index, unsigned int float_precision)
{
  pll_partition_t *p = (pll_partition_t *)partition;
  if (clv_index < p->tips && (p->attributes & PLL_ATTRIB_PATTERN_TIP)) return; /* tips are characters there, not CLVs */
  int ok = pll_gpu_sync_clv(p, clv_index);
  if (ok && scaler_index != PLL_SCALE_BUFFER_NONE) ok = pll_gpu_sync_scaler(p, (unsigned int)scaler_index);
  const unsigned int *site_id = NULL;
  if (ok && pll_repeats_enabled(p) && p->repeats->pernode_ids[clv_index])
  {
    ok = pll_gpu_sync_repeats(p, (int)clv_index);
    site_id = p->repeats->pernode_site_id[clv_index];
  }
  if (!ok) fprintf(stderr, "libpll_amd: pll_show_clv: [%d] %s\n", pll_errno, pll_errmsg);
  const double *clv = p->clv[clv_index];
  const unsigned int *scaler = scaler_index == PLL_SCALE_BUFFER_NONE ? NULL : p->scale_buffer[scaler_index];
  const unsigned int s = p->states, sp = p->states_padded, r = p->rate_cats;
  printf("[ ");
  for (unsigned int n = 0; n < p->sites; ++n)
  {
    const unsigned int e = site_id ? site_id[n] : n;
    printf("{");
    for (unsigned int k = 0; k < r; ++k)
    {
      printf("(");
      for (unsigned int j = 0; j < s; ++j)
      {
        double v = clv[((size_t)e * r + k) * sp + j];
        if (scaler) /* the reference indexes the scaler by entry only (src/output.c:90), per-rate scalers included */
          for (unsigned int t = 0; t < scaler[e]; ++t) v *= PLL_SCALE_THRESHOLD;
        printf(j + 1 < s ? "%.*f," : "%.*f)", (int)float_precision, v);
      }
      if (k + 1 < r) printf(",");
    }
    printf("} ");
  }
  printf("]\n");
}
