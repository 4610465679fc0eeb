/* partition.c - pll_partition_t lifecycle, model/weight setters, host<->device freshness.
 *
 * Layout contract: every array the reference's pll_partition_create
 * allocates (src/pll.c:424-868) exists here with the same shape, padding and initial values,
 * because callers read and write them directly. What differs is ownership of the numbers: CLVs
 * and scalers are computed in HBM and partition->clv / ->scale_buffer are a host MIRROR that is
 * refreshed on request (pll_gpu_sync_*), see DESIGN.md "Residency".
 */
#include <stdarg.h>

#include "pll_internal.h"

__thread int pll_errno;
__thread char pll_errmsg[200] = {0};

void pll_set_error(int code, const char *fmt, ...)
{
  va_list ap;
  pll_errno = code;
  va_start(ap, fmt);
  vsnprintf(pll_errmsg, sizeof pll_errmsg, fmt, ap);
  va_end(ap);
}

void pll_set_gpu_error(const char *where)
{
  pll_set_error(PLL_ERROR_GPU_RUNTIME, "%s: %s", where, pllgpu_last_error());
  /* void entry points of the reference API cannot return a status: be loud as well */
  fprintf(stderr, "libpll_amd: %s\n", pll_errmsg);
}

void *pll_aligned_alloc(size_t size, size_t alignment)
{
  void *mem = NULL;
  if (alignment < sizeof(void *)) alignment = sizeof(void *);
  if (posix_memalign(&mem, alignment, size ? size : alignment)) return NULL;
  return mem;
}

void pll_aligned_free(void *ptr) { free(ptr); }

unsigned int pll_sites_alloc(const pll_partition_t *p)
{
  return p->sites + (p->asc_bias_alloc ? p->states : 0);
}

/* zero-filled aligned block. Large blocks come straight from the kernel as untouched zero pages,
 * so the host mirror of a CLV costs no resident memory until somebody syncs it. */
static void *zalloc_aligned(size_t bytes, size_t alignment)
{
  void *m = pll_aligned_alloc(bytes, alignment);
  if (m && bytes < ((size_t)1 << 20)) memset(m, 0, bytes);
  return m;
}

int pll_is_pattern_tip(const pll_partition_t *p, unsigned int clv_index)
{
  return (p->attributes & PLL_ATTRIB_PATTERN_TIP) && clv_index < p->tips;
}

int pll_tip_by_codes(const pll_partition_t *p, unsigned int clv_index)
{
  if (clv_index >= p->tips) return 0;
  if (p->attributes & PLL_ATTRIB_PATTERN_TIP) return 1;
  const pll_amd_ext_t *x = pll_ext(p);
  return x && x->tip_compact && x->tip_compact[clv_index];
}

void pll_tip_densify(pll_partition_t *p, unsigned int clv_index)
{
  pll_amd_ext_t *x = pll_ext(p);
  if (!x || clv_index >= p->tips || (p->attributes & PLL_ATTRIB_PATTERN_TIP)) return;
  if (x->tip_compact[clv_index])
  {
    x->tip_compact[clv_index] = 0;
    x->fast_valid = 0;
    x->clv_side[clv_index] = SIDE_HOST; /* the host mirror holds the indicator CLV */
  }
}

static void free_ext(pll_amd_ext_t *x)
{
  if (x->ctx) pllgpu_destroy(x->ctx);
  free(x->tip_compact);
  free(x->ctipmap);
  free(x->clv_side);
  free(x->scaler_side);
  free(x->scaler_entries);
  free(x->tipchars_dirty);
  free(x->repeats_dirty);
  free(x->pmatrix_dirty);
  free(x->freqs_dirty);
  free(x->eigen_dirty);
  free(x->pmatrix_stale);
  free(x->pmatrix_params);
  free(x->fast_ops);
  free(x->model_version);
  free(x->model_foreign);
  free(x->pmatrix_version);
  free(x->repeats_stale);
  free(x->repeats_count);
  free(x->map_version);
  free(x->map_stamp);
  free(x->aux_params);
  free(x->gops);
  free(x->lvl_clv_w);
  free(x->lvl_clv_r);
  free(x->lvl_sc_w);
  free(x->lvl_sc_r);
  x->magic = 0;
}

void pll_partition_destroy(pll_partition_t *p)
{
  unsigned int i;
  if (!p) return;
  pll_amd_ext_t *x = pll_ext(p);
  if (x && x->tipcodes)
  {
    for (i = 0; i < p->tips; ++i) free(x->tipcodes[i]);
    free(x->tipcodes);
    x->tipcodes = NULL;
  }
  if (x) free_ext(x);

  free(p->rates);
  free(p->rate_weights);
  free(p->eigen_decomp_valid);
  free(p->prop_invar);
  free(p->invariant);
  free(p->pattern_weights);
  if (p->scale_buffer)
    for (i = 0; i < p->scale_buffers; ++i) free(p->scale_buffer[i]);
  free(p->scale_buffer);
  if (p->tipchars)
    for (i = 0; i < p->tips; ++i) free(p->tipchars[i]);
  free(p->tipchars);
  free(p->ttlookup);
  free(p->charmap);
  free(p->tipmap);
  if (p->clv)
    for (i = 0; i < p->nodes; ++i) free(p->clv[i]);
  free(p->clv);
  if (p->pmatrix) free(p->pmatrix[0]);
  free(p->pmatrix);
#define FREE_PER_MATRIX(arr)                                   \
  if (p->arr)                                                  \
    for (i = 0; i < p->rate_matrices; ++i) free(p->arr[i]);    \
  free(p->arr)
  FREE_PER_MATRIX(subst_params);
  FREE_PER_MATRIX(eigenvecs);
  FREE_PER_MATRIX(inv_eigenvecs);
  FREE_PER_MATRIX(eigenvals);
  FREE_PER_MATRIX(frequencies);
#undef FREE_PER_MATRIX
  if (p->repeats)
  {
    pll_repeats_t *r = p->repeats;
    for (i = 0; i < p->nodes; ++i)
    {
      if (r->pernode_site_id) free(r->pernode_site_id[i]);
      if (r->pernode_id_site) free(r->pernode_id_site[i]);
    }
    free(r->pernode_site_id);
    free(r->pernode_id_site);
    free(r->pernode_ids);
    free(r->perscale_ids);
    free(r->pernode_allocated_clvs);
    free(r->lookup_buffer);
    free(r->toclean_buffer);
    free(r->id_site_buffer);
    free(r->bclv_buffer);
    free(r->charmap);
    free(r);
  }
  free(p);
}

static int env_flag(const char *name)
{
  const char *v = getenv(name);
  return v && *v && strcmp(v, "0") != 0;
}

pll_partition_t *pll_partition_create(unsigned int tips, unsigned int clv_buffers, unsigned int states,
                                      unsigned int sites, unsigned int rate_matrices,
                                      unsigned int prob_matrices, unsigned int rate_cats,
                                      unsigned int scale_buffers, unsigned int attributes)
{
  unsigned int i;

  if (__builtin_popcount(attributes & PLL_ATTRIB_ARCH_MASK) > 1)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "Multiple architecture flags specified.");
    return NULL;
  }
  /* too few sites: repeats are switched off, as in src/pll.c:445-449 */
  if (sites < 16) attributes &= ~PLL_ATTRIB_SITE_REPEATS;
  if ((attributes & PLL_ATTRIB_SITE_REPEATS) && (attributes & PLL_ATTRIB_PATTERN_TIP))
  {
    /* "only one may be set" (docs/pll_partition_t.md:126-128); the reference silently returns
     * -inf for the combination, here it is refused */
    pll_set_error(PLL_ERROR_PARAM_INVALID, "PLL_ATTRIB_PATTERN_TIP and PLL_ATTRIB_SITE_REPEATS are mutually exclusive.");
    return NULL;
  }
  if ((attributes & (PLL_ATTRIB_AB_MASK | PLL_ATTRIB_AB_FLAG)) && (attributes & PLL_ATTRIB_SITE_REPEATS))
  {
    /* the reference's repeats update never computes the per-state extra sites
     * (src/partials.c:183-235 has no asc_bias_alloc branch): the combination has no defined result */
    pll_set_error(PLL_ERROR_GPU_UNSUPPORTED, "ascertainment-bias correction cannot be combined with PLL_ATTRIB_SITE_REPEATS.");
    return NULL;
  }
  if ((attributes & PLL_ATTRIB_AB_MASK) > PLL_ATTRIB_AB_STAMATAKIS)
  {
    pll_set_error(PLL_ERROR_AB_INVALIDMETHOD, "Illegal ascertainment bias algorithm \"%d\"", (int)(attributes & PLL_ATTRIB_AB_MASK));
    return NULL;
  }
  if (states < 2 || states > 64 || rate_cats < 1)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "Unsupported shape: %u states (2..64), %u rate categories.", states, rate_cats);
    return NULL;
  }

  pll_partition_t *p = (pll_partition_t *)calloc(1, sizeof(pll_partition_t) + sizeof(pll_amd_ext_t));
  if (!p)
  {
    pll_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate memory for partition.");
    return NULL;
  }
  pll_amd_ext_t *x = (pll_amd_ext_t *)(p + 1);
  x->magic = PLL_AMD_MAGIC;

  /* host-visible padding follows the ARCH bits exactly as the reference does (src/pll.c:462-485) */
  p->alignment = PLL_ALIGNMENT_CPU;
  p->states_padded = states;
  if (attributes & PLL_ATTRIB_ARCH_SSE)
  {
    p->alignment = PLL_ALIGNMENT_SSE;
    p->states_padded = (states + 1) & ~1u;
  }
  if (attributes & (PLL_ATTRIB_ARCH_AVX | PLL_ATTRIB_ARCH_AVX2))
  {
    p->alignment = PLL_ALIGNMENT_AVX;
    p->states_padded = (states + 3) & ~3u;
  }
  const unsigned int sp = p->states_padded;
  p->attributes = attributes;
  p->tips = tips;
  p->clv_buffers = clv_buffers;
  p->nodes = tips + clv_buffers;
  p->states = states;
  p->sites = sites;
  p->pattern_weight_sum = sites;
  p->rate_matrices = rate_matrices;
  p->prob_matrices = prob_matrices;
  p->rate_cats = rate_cats;
  p->scale_buffers = scale_buffers;
  /* one extra "site" per state behind the real ones (src/pll.c:525-531) */
  p->asc_bias_alloc = (attributes & (PLL_ATTRIB_AB_MASK | PLL_ATTRIB_AB_FLAG)) != 0;
  p->asc_additional_sites = p->asc_bias_alloc ? (int)states : 0;
  const unsigned int sites_alloc = sites + (unsigned int)p->asc_additional_sites;
  x->sites_alloc = sites_alloc;
  const int repeats = (attributes & PLL_ATTRIB_SITE_REPEATS) != 0;
  const size_t span = (size_t)rate_cats * sp;

#define NEED(ptr)                                                        \
  if (!(ptr))                                                            \
  {                                                                      \
    pll_partition_destroy(p);                                            \
    pll_set_error(PLL_ERROR_MEM_ALLOC, "Unable to allocate enough memory."); \
    return NULL;                                                         \
  }

  p->eigen_decomp_valid = (int *)calloc(rate_matrices ? rate_matrices : 1, sizeof(int));
  NEED(p->eigen_decomp_valid);
  p->clv = (double **)calloc(p->nodes ? p->nodes : 1, sizeof(double *));
  NEED(p->clv);
  if (!repeats)
  {
    /* tips keep NULL CLVs under PATTERN_TIP (src/pll.c:558-563) */
    const unsigned int start = (attributes & PLL_ATTRIB_PATTERN_TIP) ? tips : 0;
    for (i = start; i < p->nodes; ++i)
    {
      p->clv[i] = (double *)zalloc_aligned((size_t)sites_alloc * span * sizeof(double), p->alignment);
      NEED(p->clv[i]);
    }
  }
  /* one contiguous block for all transition matrices plus the displacement tail the
   * reference's SIMD kernels read past the last matrix (src/pll.c:593-617) */
  p->pmatrix = (double **)calloc(prob_matrices ? prob_matrices : 1, sizeof(double *));
  NEED(p->pmatrix);
  {
    const size_t per = (size_t)states * sp * rate_cats;
    const size_t tail = (size_t)(sp - states) * sp;
    double *blk = (double *)pll_aligned_alloc((prob_matrices * per + tail) * sizeof(double), p->alignment);
    NEED(blk);
    memset(blk, 0, (prob_matrices * per + tail) * sizeof(double));
    for (i = 0; i < prob_matrices; ++i) p->pmatrix[i] = blk + i * per;
  }
#define PER_MATRIX(arr, count)                                                             \
  p->arr = (double **)calloc(rate_matrices ? rate_matrices : 1, sizeof(double *));         \
  NEED(p->arr);                                                                            \
  for (i = 0; i < rate_matrices; ++i)                                                      \
  {                                                                                        \
    p->arr[i] = (double *)pll_aligned_alloc((count) * sizeof(double), p->alignment);       \
    NEED(p->arr[i]);                                                                       \
    memset(p->arr[i], 0, (count) * sizeof(double));                                        \
  }
  PER_MATRIX(eigenvecs, (size_t)states * sp);
  PER_MATRIX(inv_eigenvecs, (size_t)states * sp);
  PER_MATRIX(eigenvals, (size_t)sp);
  PER_MATRIX(subst_params, (size_t)(states * states - states) / 2 + 1);
  PER_MATRIX(frequencies, (size_t)sp);
#undef PER_MATRIX
  p->rates = (double *)calloc(rate_cats, sizeof(double));
  NEED(p->rates);
  p->rate_weights = (double *)calloc(rate_cats, sizeof(double));
  NEED(p->rate_weights);
  for (i = 0; i < rate_cats; ++i) p->rate_weights[i] = 1.0 / rate_cats;
  p->prop_invar = (double *)calloc(rate_matrices ? rate_matrices : 1, sizeof(double));
  NEED(p->prop_invar);
  p->pattern_weights = (unsigned int *)malloc((sites_alloc ? sites_alloc : 1) * sizeof(unsigned int));
  NEED(p->pattern_weights);
  for (i = 0; i < sites; ++i) p->pattern_weights[i] = 1;
  for (i = sites; i < sites_alloc; ++i) p->pattern_weights[i] = 0; /* src/pll.c:822-824 */
  p->scale_buffer = (unsigned int **)calloc(scale_buffers ? scale_buffers : 1, sizeof(unsigned int *));
  NEED(p->scale_buffer);
  if (!repeats)
  {
    const size_t n = (attributes & PLL_ATTRIB_RATE_SCALERS) ? (size_t)sites_alloc * rate_cats : sites_alloc;
    for (i = 0; i < scale_buffers; ++i)
    {
      p->scale_buffer[i] = (unsigned int *)calloc(n ? n : 1, sizeof(unsigned int));
      NEED(p->scale_buffer[i]);
    }
  }

  /* freshness bookkeeping */
  x->clv_side = (unsigned char *)calloc(p->nodes ? p->nodes : 1, 1);
  x->scaler_side = (unsigned char *)calloc(scale_buffers ? scale_buffers : 1, 1);
  x->scaler_entries = (unsigned int *)calloc(scale_buffers ? scale_buffers : 1, sizeof(unsigned int));
  x->tipchars_dirty = (unsigned char *)calloc(tips ? tips : 1, 1);
  x->repeats_dirty = (unsigned char *)calloc(p->nodes ? p->nodes : 1, 1);
  x->pmatrix_dirty = (unsigned char *)malloc(prob_matrices ? prob_matrices : 1);
  x->freqs_dirty = (unsigned char *)malloc(rate_matrices ? rate_matrices : 1);
  x->tip_compact = (unsigned char *)calloc(tips ? tips : 1, 1);
  x->tipcodes = (unsigned char **)calloc(tips ? tips : 1, sizeof(unsigned char *));
  x->ctipmap = (pll_state_t *)calloc(PLL_ASCII_SIZE, sizeof(pll_state_t));
  x->no_tip_codes = env_flag("PLL_AMD_NO_TIP_CODES");
  NEED(x->tip_compact && x->tipcodes && x->ctipmap);
  x->eigen_dirty = (unsigned char *)malloc(rate_matrices ? rate_matrices : 1);
  x->pmatrix_stale = (unsigned char *)calloc(prob_matrices ? prob_matrices : 1, 1);
  NEED(x->pmatrix_stale);
  x->pmatrix_params = (unsigned char *)malloc((size_t)(prob_matrices ? prob_matrices : 1) * rate_cats);
  NEED(x->pmatrix_params);
  memset(x->pmatrix_params, 0xFF, (size_t)(prob_matrices ? prob_matrices : 1) * rate_cats);
  x->model_version = (unsigned int *)calloc(rate_matrices ? rate_matrices : 1, sizeof(unsigned int));
  x->model_foreign = (unsigned char *)calloc(rate_matrices ? rate_matrices : 1, 1);
  x->pmatrix_version = (unsigned int *)calloc((size_t)(prob_matrices ? prob_matrices : 1) * rate_cats, sizeof(unsigned int));
  NEED(x->model_version && x->model_foreign && x->pmatrix_version);
  x->repeats_stale = (unsigned char *)calloc(p->nodes ? p->nodes : 1, 1);
  x->repeats_count = (unsigned int *)calloc(p->nodes ? p->nodes : 1, sizeof(unsigned int));
  NEED(x->repeats_stale && x->repeats_count);
  x->map_version = (unsigned long long *)calloc(p->nodes ? p->nodes : 1, sizeof(unsigned long long));
  x->map_stamp = (pll_map_stamp_t *)calloc(p->nodes ? p->nodes : 1, sizeof(pll_map_stamp_t));
  NEED(x->map_version && x->map_stamp);
  {
    const char *v = getenv("PLL_AMD_REP_STAMPS");
    x->map_stamps = !(v && *v == '0');
  }
  x->aux_params = (unsigned int *)malloc(sizeof(unsigned int) * rate_cats);
  NEED(x->clv_side && x->scaler_side && x->scaler_entries && x->tipchars_dirty && x->repeats_dirty &&
       x->pmatrix_dirty && x->freqs_dirty && x->eigen_dirty && x->aux_params);
  memset(x->eigen_dirty, 1, rate_matrices ? rate_matrices : 1);
  x->rates_dirty = 1;
  x->eigen_version = 1;
  x->aux_version = 0;
  memset(x->pmatrix_dirty, 1, prob_matrices ? prob_matrices : 1);
  memset(x->freqs_dirty, 1, rate_matrices ? rate_matrices : 1);
  x->rate_weights_dirty = x->pattern_weights_dirty = x->prop_invar_dirty = 1;
  x->invariant_dirty = 0;
  x->tipmap_dirty = 0;
  x->eager_mirror = env_flag("PLL_AMD_EAGER_MIRROR");
  x->always_upload = env_flag("PLL_AMD_ALWAYS_UPLOAD");

  if (repeats && !pll_repeats_initialize(p))
  {
    pll_partition_destroy(p);
    return NULL;
  }

  /* device context. There is no CPU arithmetic behind this library: without a device the
   * partition is refused, unless the caller explicitly asks for a host-only shell (CPU tests
   * of the bookkeeping), in which case every compute call fails loudly. */
  if (env_flag("PLL_AMD_HOST_ONLY"))
    x->ctx = NULL;
  else
  {
    pllgpu_geometry_t g;
    memset(&g, 0, sizeof g);
    g.tips = tips;
    g.nodes = p->nodes;
    g.states = states;
    g.states_padded = sp;
    g.rate_cats = rate_cats;
    g.sites = sites;
    g.sites_alloc = sites_alloc;
    g.prob_matrices = prob_matrices;
    g.rate_matrices = rate_matrices;
    g.scale_buffers = scale_buffers;
    g.per_rate_scalers = (attributes & PLL_ATTRIB_RATE_SCALERS) ? 1 : 0;
    g.pattern_tip = (attributes & PLL_ATTRIB_PATTERN_TIP) ? 1 : 0;
    x->ctx = pllgpu_create(&g, -1);
    if (!x->ctx)
    {
      pll_partition_destroy(p);
      pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "MI355X context: %s", pllgpu_last_error());
      return NULL;
    }
    if (!repeats)
    {
      /* reserve HBM for every inner CLV and scaler now: no allocation inside the hot path */
      for (i = tips; i < p->nodes; ++i)
        if (pllgpu_clv_reserve(x->ctx, i, sites_alloc))
        {
          pll_partition_destroy(p);
          pll_set_error(PLL_ERROR_MEM_ALLOC, "MI355X: %s", pllgpu_last_error());
          return NULL;
        }
      for (i = 0; i < scale_buffers; ++i)
        if (pllgpu_scaler_reserve(x->ctx, i, sites_alloc))
        {
          pll_partition_destroy(p);
          pll_set_error(PLL_ERROR_MEM_ALLOC, "MI355X: %s", pllgpu_last_error());
          return NULL;
        }
    }
  }
#undef NEED
  return p;
}

/* ---- setters (src/pll.c:1131-1143, src/models.c:445-493) -------------------------------------- */
void pll_set_pattern_weights(pll_partition_t *p, const unsigned int *w)
{
  unsigned int i, sum = 0;
  memcpy(p->pattern_weights, w, sizeof(unsigned int) * p->sites);
  for (i = 0; i < p->sites; ++i) sum += w[i];
  p->pattern_weight_sum = sum;
  pll_amd_ext_t *x = pll_ext(p);
  if (x) x->pattern_weights_dirty = 1;
}

/* src/pll.c:1145-1191 */
int pll_set_asc_bias_type(pll_partition_t *p, int asc_bias_type)
{
  unsigned int i;
  int prop_invar = 0;
  const int attr = asc_bias_type & PLL_ATTRIB_AB_MASK;
  if (!p->asc_bias_alloc)
  {
    pll_set_error(PLL_ERROR_AB_NOSUPPORT, "Partition was not created with ascertainment bias support");
    return PLL_FAILURE;
  }
  for (i = 0; i < p->rate_matrices; ++i) prop_invar |= (p->prop_invar[i] > 0);
  if (asc_bias_type != 0 && prop_invar)
  {
    pll_set_error(PLL_ERROR_INVAR_INCOMPAT, "Invariant sites are not compatible with asc bias correction");
    return PLL_FAILURE;
  }
  if (attr != asc_bias_type || (unsigned int)attr > PLL_ATTRIB_AB_STAMATAKIS)
  {
    pll_set_error(PLL_ERROR_AB_INVALIDMETHOD, "Illegal ascertainment bias algorithm \"%d\"", asc_bias_type);
    return PLL_FAILURE;
  }
  p->attributes &= ~(unsigned int)PLL_ATTRIB_AB_MASK;
  p->attributes |= (unsigned int)attr;
  return PLL_SUCCESS;
}

/* src/pll.c:1193-1200 */
void pll_set_asc_state_weights(pll_partition_t *p, const unsigned int *state_weights)
{
  if (!p->asc_bias_alloc) return; /* the reference asserts */
  memcpy(p->pattern_weights + p->sites, state_weights, sizeof(unsigned int) * p->states);
  pll_amd_ext_t *x = pll_ext(p);
  if (x) x->pattern_weights_dirty = 1;
}

void pll_set_frequencies(pll_partition_t *p, unsigned int idx, const double *f)
{
  unsigned int i;
  double sum = 0;
  double *dst = p->frequencies[idx];
  memcpy(dst, f, p->states * sizeof(double));
  for (i = 0; i < p->states; ++i) sum += dst[i];
  if (sum - 1.0 > 1e-8 || 1.0 - sum > 1e-8) /* PLL_MISC_EPSILON, src/models.c:460 */
    for (i = 0; i < p->states; ++i) dst[i] /= sum;
  p->eigen_decomp_valid[idx] = 0;
  pll_amd_ext_t *x = pll_ext(p);
  if (x)
  {
    x->freqs_dirty[idx] = 1;
    x->eigen_version++;
    x->model_version[idx]++;
  }
}

void pll_set_subst_params(pll_partition_t *p, unsigned int idx, const double *params)
{
  memcpy(p->subst_params[idx], params, (size_t)p->states * (p->states - 1) / 2 * sizeof(double));
  p->eigen_decomp_valid[idx] = 0;
  pll_amd_ext_t *x = pll_ext(p);
  if (x) x->model_version[idx]++;
}

void pll_set_category_rates(pll_partition_t *p, const double *rates)
{
  memcpy(p->rates, rates, p->rate_cats * sizeof(double));
  pll_amd_ext_t *x = pll_ext(p);
  if (x) x->rates_dirty = 1;
}

void pll_set_category_weights(pll_partition_t *p, const double *w)
{
  memcpy(p->rate_weights, w, p->rate_cats * sizeof(double));
  pll_amd_ext_t *x = pll_ext(p);
  if (x) x->rate_weights_dirty = 1;
}

void pll_fill_parent_scaler(unsigned int n, unsigned int *parent, const unsigned int *left,
                            const unsigned int *right)
{
  /* src/pll.c:1202-1224: host utility kept for callers; the kernels fuse this step */
  unsigned int i;
  if (!left && !right)
    memset(parent, 0, sizeof(unsigned int) * n);
  else if (left && right)
    for (i = 0; i < n; ++i) parent[i] = left[i] + right[i];
  else
    memcpy(parent, left ? left : right, sizeof(unsigned int) * n);
}

/* ---- host -> device freshness ------------------------------------------------------------------ */
#define GPU_TRY(call, where)     \
  if ((call) != 0)               \
  {                              \
    pll_set_gpu_error(where);    \
    return PLL_FAILURE;          \
  }

int pll_flush_model(pll_partition_t *p, pll_amd_ext_t *x)
{
  unsigned int i;
  if (x->always_upload)
  {
    memset(x->freqs_dirty, 1, p->rate_matrices);
    x->rate_weights_dirty = x->pattern_weights_dirty = x->prop_invar_dirty = 1;
    if (p->invariant) x->invariant_dirty = 1;
  }
  for (i = 0; i < p->rate_matrices; ++i)
    if (x->freqs_dirty[i])
    {
      GPU_TRY(pllgpu_frequencies_upload(x->ctx, i, p->frequencies[i]), "frequencies upload");
      x->freqs_dirty[i] = 0;
    }
  if (x->rate_weights_dirty)
  {
    GPU_TRY(pllgpu_rate_weights_upload(x->ctx, p->rate_weights), "rate weights upload");
    x->rate_weights_dirty = 0;
  }
  if (x->prop_invar_dirty)
  {
    GPU_TRY(pllgpu_prop_invar_upload(x->ctx, p->prop_invar), "prop_invar upload");
    x->prop_invar_dirty = 0;
  }
  if (x->pattern_weights_dirty)
  {
    GPU_TRY(pllgpu_pattern_weights_upload(x->ctx, p->pattern_weights, x->sites_alloc), "pattern weights upload");
    x->pattern_weights_dirty = 0;
  }
  if (x->invariant_dirty)
  {
    GPU_TRY(pllgpu_invariant_upload(x->ctx, p->invariant, p->sites), "invariant upload");
    x->invariant_dirty = 0;
  }
  if (x->tipmap_dirty)
  {
    /* 4-state codes are the masks themselves (src/pll.c:893-895) */
    const pll_state_t *tm = (p->attributes & PLL_ATTRIB_PATTERN_TIP) ? p->tipmap : x->ctipmap;
    GPU_TRY(pllgpu_tipmap_upload(x->ctx, p->states == 4 ? NULL : tm, PLL_ASCII_SIZE), "tipmap upload");
    x->tipmap_dirty = 0;
  }
  return PLL_SUCCESS;
}

int pll_flush_clv(pll_partition_t *p, pll_amd_ext_t *x, unsigned int idx)
{
  if (pll_tip_by_codes(p, idx))
  {
    if (x->tipchars_dirty[idx])
    {
      const int pattern = (p->attributes & PLL_ATTRIB_PATTERN_TIP) != 0;
      GPU_TRY(pllgpu_tipchars_upload(x->ctx, idx, pattern ? p->tipchars[idx] : x->tipcodes[idx],
                                     pattern ? x->sites_alloc : pll_get_sites_number(p, idx)),
              "tip codes upload");
      x->tipchars_dirty[idx] = 0;
    }
    return PLL_SUCCESS;
  }
  if (x->clv_side[idx] == SIDE_HOST)
  {
    GPU_TRY(pllgpu_clv_upload(x->ctx, idx, p->clv[idx], pll_get_sites_number(p, idx)), "CLV upload");
    x->clv_side[idx] = SIDE_BOTH;
  }
  else if (x->clv_side[idx] == SIDE_NONE)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "CLV %u is read before it was set or computed", idx);
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

int pll_flush_scaler(pll_partition_t *p, pll_amd_ext_t *x, int idx)
{
  if (idx == PLL_SCALE_BUFFER_NONE) return PLL_SUCCESS;
  if (x->scaler_side[idx] == SIDE_HOST)
  {
    /* a buffer the caller filled before the device ever saw it holds one entry per site */
    if (!x->scaler_entries[idx]) x->scaler_entries[idx] = x->sites_alloc;
    GPU_TRY(pllgpu_scaler_upload(x->ctx, (unsigned)idx, p->scale_buffer[idx], x->scaler_entries[idx]), "scaler upload");
    x->scaler_side[idx] = SIDE_BOTH;
  }
  else if (x->scaler_side[idx] == SIDE_NONE)
  {
    /* never written: the reference hands out zero-initialised scalers (src/pll.c:845) */
    unsigned int n = x->scaler_entries[idx] ? x->scaler_entries[idx] : x->sites_alloc;
    if (!p->scale_buffer[idx]) return PLL_SUCCESS;
    GPU_TRY(pllgpu_scaler_upload(x->ctx, (unsigned)idx, p->scale_buffer[idx], n), "scaler upload");
    x->scaler_entries[idx] = n;
    x->scaler_side[idx] = SIDE_BOTH;
  }
  return PLL_SUCCESS;
}

int pll_flush_pmatrix(pll_partition_t *p, pll_amd_ext_t *x, unsigned int first, unsigned int last)
{
  /* upload maximal dirty runs inside [first, last] */
  unsigned int i = first;
  if (x->always_upload) memset(x->pmatrix_dirty + first, 1, last - first + 1);
  while (i <= last)
  {
    /* a matrix computed on the device is newer than its host mirror: never overwrite it */
    if (!x->pmatrix_dirty[i] || x->pmatrix_stale[i])
    {
      x->pmatrix_dirty[i] = 0;
      ++i;
      continue;
    }
    unsigned int j = i;
    while (j + 1 <= last && x->pmatrix_dirty[j + 1] && !x->pmatrix_stale[j + 1]) ++j;
    GPU_TRY(pllgpu_pmatrix_upload(x->ctx, i, j - i + 1, p->pmatrix[i]), "p-matrix upload");
    memset(x->pmatrix_dirty + i, 0, j - i + 1);
    i = j + 1;
  }
  return PLL_SUCCESS;
}

int pll_flush_eigen(pll_partition_t *p, pll_amd_ext_t *x)
{
  unsigned int i;
  if (!pll_flush_model(p, x)) return PLL_FAILURE;
  for (i = 0; i < p->rate_matrices; ++i)
    if (x->eigen_dirty[i] || x->always_upload)
    {
      GPU_TRY(pllgpu_eigen_upload(x->ctx, i, p->eigenvecs[i], p->inv_eigenvecs[i], p->eigenvals[i]), "eigensystem upload");
      x->eigen_dirty[i] = 0;
    }
  if (x->rates_dirty || x->always_upload)
  {
    GPU_TRY(pllgpu_rates_upload(x->ctx, p->rates), "category rates upload");
    x->rates_dirty = 0;
  }
  return PLL_SUCCESS;
}

int pll_flush_repeats(pll_partition_t *p, pll_amd_ext_t *x, unsigned int node)
{
  if (!pll_repeats_enabled(p) || !x->repeats_dirty[node]) return PLL_SUCCESS;
  pll_repeats_t *r = p->repeats;
  GPU_TRY(pllgpu_repeats_upload(x->ctx, node, r->pernode_site_id[node], r->pernode_id_site[node], r->pernode_ids[node]),
          "repeats maps upload");
  x->repeats_dirty[node] = 0;
  return PLL_SUCCESS;
}

/* ---- device-residency API (include/pll_amd.h) ------------------------------------------------ */
static pll_amd_ext_t *need_ctx(const pll_partition_t *p, const char *what)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  if (!x)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "%s: partition was not created by libpll_amd", what);
    return NULL;
  }
  if (!x->ctx)
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "%s: partition has no MI355X context (PLL_AMD_HOST_ONLY)", what);
    fprintf(stderr, "libpll_amd: %s\n", pll_errmsg);
    return NULL;
  }
  return x;
}

int pll_gpu_sync_clv(pll_partition_t *p, unsigned int idx)
{
  pll_amd_ext_t *x = need_ctx(p, "pll_gpu_sync_clv");
  if (!x) return PLL_FAILURE;
  if (idx >= p->nodes || pll_tip_by_codes(p, idx)) return PLL_SUCCESS; /* host copy is the original */
  if (x->clv_side[idx] == SIDE_DEVICE)
  {
    if (!p->clv[idx])
    {
      pll_set_error(PLL_ERROR_MEM_ALLOC, "CLV %u has no host buffer", idx);
      return PLL_FAILURE;
    }
    GPU_TRY(pllgpu_clv_download(x->ctx, idx, p->clv[idx], pll_get_sites_number(p, idx)), "CLV download");
    x->clv_side[idx] = SIDE_BOTH;
  }
  return PLL_SUCCESS;
}

int pll_gpu_sync_scaler(pll_partition_t *p, unsigned int idx)
{
  pll_amd_ext_t *x = need_ctx(p, "pll_gpu_sync_scaler");
  if (!x) return PLL_FAILURE;
  if (idx >= p->scale_buffers) return PLL_SUCCESS;
  if (x->scaler_side[idx] == SIDE_DEVICE && p->scale_buffer[idx])
  {
    GPU_TRY(pllgpu_scaler_download(x->ctx, idx, p->scale_buffer[idx], x->scaler_entries[idx]), "scaler download");
    x->scaler_side[idx] = SIDE_BOTH;
  }
  return PLL_SUCCESS;
}

int pll_gpu_sync_pmatrix(pll_partition_t *p, int index)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  unsigned int i;
  if (!x || !x->ctx)
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "pll_gpu_sync_pmatrix: no MI355X context behind this partition");
    return PLL_FAILURE;
  }
  if (index >= (int)p->prob_matrices)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_sync_pmatrix: matrix index %d out of range", index);
    return PLL_FAILURE;
  }
  for (i = 0; i < p->prob_matrices; ++i)
  {
    if ((index >= 0 && i != (unsigned int)index) || !x->pmatrix_stale[i]) continue;
    GPU_TRY(pllgpu_pmatrix_download(x->ctx, i, p->pmatrix[i]), "pll_gpu_sync_pmatrix");
    x->pmatrix_stale[i] = 0;
  }
  return PLL_SUCCESS;
}

int pll_gpu_sync_repeats(pll_partition_t *p, int node)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  unsigned int i;
  if (!x || !pll_repeats_enabled(p)) return PLL_SUCCESS;
  if (node >= (int)p->nodes)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_sync_repeats: node %d out of range", node);
    return PLL_FAILURE;
  }
  for (i = 0; i < p->nodes; ++i)
  {
    if ((node >= 0 && i != (unsigned int)node) || !x->repeats_stale[i]) continue;
    GPU_TRY(pllgpu_repeats_download(x->ctx, i, p->repeats->pernode_site_id[i], p->repeats->pernode_id_site[i], x->repeats_count[i]),
            "pll_gpu_sync_repeats");
    x->repeats_stale[i] = 0;
  }
  return PLL_SUCCESS;
}

int pll_gpu_sync_all(pll_partition_t *p)
{
  unsigned int i;
  int ok = PLL_SUCCESS;
  ok &= pll_gpu_sync_pmatrix(p, -1);
  ok &= pll_gpu_sync_repeats(p, -1);
  /* small CLVs and scaler vectors: enqueued one after another, waited for together */
  pll_amd_ext_t *x = pll_ext(p);
  /* ... so a mirror is only in step once that wait has succeeded: what was marked on the way is taken back otherwise */
  unsigned char *was = (x && x->ctx) ? (unsigned char *)malloc((size_t)p->nodes + p->scale_buffers + 1) : NULL;
  if (was)
  {
    memcpy(was, x->clv_side, p->nodes);
    memcpy(was + p->nodes, x->scaler_side, p->scale_buffers);
    (void)pllgpu_download_defer(x->ctx, 1);
  }
  for (i = 0; i < p->nodes; ++i) ok &= pll_gpu_sync_clv(p, i);
  for (i = 0; i < p->scale_buffers; ++i) ok &= pll_gpu_sync_scaler(p, i);
  if (was && pllgpu_download_defer(x->ctx, 0) != 0)
  {
    pll_set_gpu_error("pll_gpu_sync_all");
    for (i = 0; i < p->nodes; ++i)
      if (was[i] == SIDE_DEVICE) x->clv_side[i] = SIDE_DEVICE;
    for (i = 0; i < p->scale_buffers; ++i)
      if (was[p->nodes + i] == SIDE_DEVICE) x->scaler_side[i] = SIDE_DEVICE;
    ok = PLL_FAILURE;
  }
  free(was);
  return ok;
}

void pll_gpu_invalidate(pll_partition_t *p, unsigned int what, int index)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  unsigned int i;
  if (!x) return;
  /* the caller wrote something: the next traversal takes the whole path again (forgetting what the class maps were
   * computed from writes nothing: if they come out as they were, the classified list stands - repeats.c) */
  if (what & ~PLL_GPU_FORGET_REPEATS) x->fast_valid = 0;
#define MARK(arr, n, val)                                             \
  do                                                                  \
  {                                                                   \
    if (index < 0)                                                    \
      for (i = 0; i < (n); ++i) x->arr[i] = (val);                    \
    else if ((unsigned)index < (n))                                   \
      x->arr[index] = (val);                                          \
  } while (0)
  if (what & PLL_GPU_DIRTY_PMATRIX)
  {
    MARK(pmatrix_dirty, p->prob_matrices, 1);
    MARK(pmatrix_stale, p->prob_matrices, 0); /* the caller wrote the host copy: it is the truth now */
    for (i = 0; i < p->prob_matrices; ++i)    /* ... and nothing is known about how it was formed */
      if (index < 0 || i == (unsigned int)index) memset(x->pmatrix_params + (size_t)i * p->rate_cats, 0xFF, p->rate_cats);
  }
  if (what & PLL_GPU_DIRTY_FREQS)
  {
    MARK(freqs_dirty, p->rate_matrices, 1);
    for (i = 0; i < p->rate_matrices; ++i) /* matrices formed with the old frequencies are not reversible for the new */
      if (index < 0 || i == (unsigned int)index) x->model_version[i]++;
    /* ... and neither is a matrix formed LATER from the eigensystem that still stands: a direct write leaves
     * eigen_decomp_valid set (as in the reference, whose pll_update_prob_matrices then keeps using the old
     * eigensystem, src/models.c:412-443), so until pll_update_eigen recomputes it from the new frequencies the
     * eigensystem is foreign to its set and swap_is_exact() (likelihood.c) must not trust matrices made from it */
    MARK(model_foreign, p->rate_matrices, 1);
  }
  if (what & PLL_GPU_DIRTY_RATE_WEIGHTS) x->rate_weights_dirty = x->prop_invar_dirty = 1;
  if (what & PLL_GPU_DIRTY_PATTERN_WEIGHTS) x->pattern_weights_dirty = 1;
  if (what & PLL_GPU_DIRTY_INVARIANT) x->invariant_dirty = x->prop_invar_dirty = 1;
  if (what & PLL_GPU_DIRTY_CLV)
  {
    MARK(clv_side, p->nodes, SIDE_HOST);
    if (!(p->attributes & PLL_ATTRIB_PATTERN_TIP)) MARK(tip_compact, p->tips, 0); /* arbitrary values now */
  }
  if (what & PLL_GPU_DIRTY_SCALER) MARK(scaler_side, p->scale_buffers, SIDE_HOST);
  if (what & PLL_GPU_DIRTY_TIPCHARS)
  {
    MARK(tipchars_dirty, p->tips, 1);
    x->tipmap_dirty = 1;
  }
  if (what & PLL_GPU_DIRTY_REPEATS)
  {
    MARK(repeats_dirty, p->nodes, 1);
    pll_maps_touched(x, p, index);
  }
  if (what & PLL_GPU_FORGET_REPEATS) pll_maps_touched(x, p, index);
  if (what & PLL_GPU_DIRTY_EIGEN)
  {
    MARK(eigen_dirty, p->rate_matrices, 1);
    MARK(model_foreign, p->rate_matrices, 1); /* the caller's own eigensystem: nothing is known about it */
    for (i = 0; i < p->rate_matrices; ++i)
      if (index < 0 || i == (unsigned int)index) x->model_version[i]++;
    x->eigen_version++;
    x->rates_dirty = 1;
  }
#undef MARK
}

int pll_gpu_set_stream(pll_partition_t *p, void *s)
{
  pll_amd_ext_t *x = need_ctx(p, "pll_gpu_set_stream");
  if (!x) return PLL_FAILURE;
  GPU_TRY(pllgpu_set_stream(x->ctx, s), "set stream");
  return PLL_SUCCESS;
}

void *pll_gpu_get_stream(const pll_partition_t *p)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  return (x && x->ctx) ? pllgpu_get_stream(x->ctx) : NULL;
}

int pll_gpu_synchronize(pll_partition_t *p)
{
  pll_amd_ext_t *x = need_ctx(p, "pll_gpu_synchronize");
  if (!x) return PLL_FAILURE;
  GPU_TRY(pllgpu_synchronize(x->ctx), "synchronize");
  return PLL_SUCCESS;
}

int pll_gpu_timer_start(pll_partition_t *p)
{
  pll_amd_ext_t *x = need_ctx(p, "pll_gpu_timer_start");
  if (!x) return PLL_FAILURE;
  GPU_TRY(pllgpu_timer_start(x->ctx), "timer start");
  return PLL_SUCCESS;
}

double pll_gpu_timer_stop(pll_partition_t *p)
{
  pll_amd_ext_t *x = need_ctx(p, "pll_gpu_timer_stop");
  if (!x) return -1.0;
  return pllgpu_timer_stop(x->ctx);
}

double pll_gpu_last_algorithmic_bytes(const pll_partition_t *p)
{
  const pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  return (x && x->ctx) ? pllgpu_last_algorithmic_bytes(x->ctx) : 0.0;
}

int pll_gpu_last_update_replayed(const pll_partition_t *p)
{
  const pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  return x ? x->fast_taken : 0;
}

unsigned long long pll_gpu_plan_replays(const pll_partition_t *p)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  return (x && x->ctx) ? pllgpu_plan_replays(x->ctx) : 0ull;
}

unsigned long long pll_gpu_class_map_work(const pll_partition_t *p, int launches)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  return (x && x->ctx) ? pllgpu_class_map_work(x->ctx, launches) : 0ull;
}

unsigned int pll_gpu_last_launch_count(const pll_partition_t *p)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  return (x && x->ctx) ? pllgpu_last_launch_count(x->ctx) : 0;
}

int pll_gpu_device_count(void) { return pllgpu_device_count(); }

int pll_gpu_available(void) { return pllgpu_device_count() > 0; }
