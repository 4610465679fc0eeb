/* core_seam.c - the flat pll_core_* entry points of the hot path (reference: src/pll.h:1049-1177 CLV
 * updates, :1295-1414 log-likelihoods; bodies in src/core_partials.c and src/core_likelihood.c).
 *
 * The reference's partition-level functions call these with pointers into the partition; a caller
 * that uses them directly hands over raw HOST arrays. Each call here wraps the arrays in a partition of
 * the call's shape (layout from the PLL_ATTRIB_ARCH_* bits of `attrib`, scaler mode from
 * PLL_ATTRIB_RATE_SCALERS; kept per thread and shape: seam_open), runs the same device path as the
 * partition-level API and copies the result back. Functionally complete; priced at a PCIe round trip
 * per call - the partition-level entry points are the ones to use for speed.
 *
 * Tip children (the `ti` / `tt` forms) arrive as encoded characters: their 0/1 indicator vectors are
 * formed on the host and the update runs as inner x inner - the same arithmetic (a tip's vector is what
 * the tip kernels multiply with too). The lookup table of the `tt` form is private to the pair
 * pll_core_create_lookup / pll_core_update_partial_tt in the reference as well: here it carries the two
 * matrices. Class-compressed operands (the `repeats` forms) are expanded through their id maps on
 * the host. No CPU arithmetic on CLVs happens here: without a device every call fails loudly. */
#include "pll_internal.h"
#include <math.h>

typedef struct
{
  pll_partition_t *p;
  unsigned int span; /* doubles per entry in the caller's layout */
  int slot;          /* where the partition lives in this thread's cache, or -1: destroyed by seam_close */
} seam_t;

/* A caller that loops over these entry points - the reference's own partition-level functions do, once per operation -
 * comes with the same shapes again and again: the partition behind a call (device context, stream, buffers) is kept,
 * per thread, and the next call of the same shape finds it (round 4 paid an allocation, a stream and their release
 * per call). A few shapes at a time (update + evaluation + derivative calls of one loop); PLL_AMD_SEAM_CACHE=0: none.
 * A thread's partitions are destroyed when it exits or calls pll_core_seam_release(); the main thread's otherwise stay
 * until the process ends. What is kept is BOUNDED BY BYTES, not only by count (round-5 advice: the API looks stateless,
 * and a 1M-site call would have left several GB of host mirror, HBM and pinned memory behind): a partition above
 * SEAM_KEEP_ONE is never kept, the idle ones together stay within SEAM_KEEP_ALL (least recently used out first). A
 * kept partition is bound to the device it was created on, so the device a new one WOULD get (PLL_AMD_DEVICE / the
 * thread's current device) is part of the match. */
#include <pthread.h>
#define SEAM_SLOTS 6
#define SEAM_KEEP_ONE ((size_t)32 << 20)
#define SEAM_KEEP_ALL ((size_t)128 << 20)
typedef struct
{
  pll_partition_t *p;
  unsigned int key[7];
  unsigned long long used;
  size_t bytes;
  int device;
  int busy;
} seam_slot_t;
static __thread seam_slot_t seam_slots[SEAM_SLOTS];
static __thread unsigned long long seam_clock;
static pthread_key_t seam_key;
static pthread_once_t seam_once = PTHREAD_ONCE_INIT;

static void seam_thread_exit(void *slots_)
{
  seam_slot_t *slots = (seam_slot_t *)slots_;
  for (int i = 0; i < SEAM_SLOTS; ++i)
    if (slots[i].p && !slots[i].busy)
    {
      pll_partition_destroy(slots[i].p);
      slots[i].p = NULL;
    }
}
static void seam_make_key(void) { (void)pthread_key_create(&seam_key, seam_thread_exit); }

static int seam_cache_on(void)
{
  static int on = -1;
  if (on < 0)
  {
    const char *v = getenv("PLL_AMD_SEAM_CACHE");
    on = !(v && *v == '0');
  }
  return on;
}

static int seam_open(seam_t *s, unsigned int states, unsigned int entries, unsigned int rate_cats, unsigned int clvs,
                     unsigned int matrices, unsigned int freq_sets, unsigned int attrib)
{
  const unsigned int attrs = (attrib & PLL_ATTRIB_ARCH_MASK) | (attrib & PLL_ATTRIB_RATE_SCALERS);
  const unsigned int key[7] = {states, entries, rate_cats, clvs, matrices, freq_sets ? freq_sets : 1, attrs};
  s->slot = -1;
  s->p = NULL;
  const int want = pllgpu_default_device(); /* -2: any (PLL_AMD_DEVICE=auto) */
  if (seam_cache_on())
    for (int i = 0; i < SEAM_SLOTS; ++i)
      if (seam_slots[i].p && !seam_slots[i].busy && memcmp(seam_slots[i].key, key, sizeof key) == 0 &&
          (want == -2 || seam_slots[i].device == want))
      {
        pll_partition_t *p = seam_slots[i].p;
        /* what a former call may have left behind and this one may not set: no invariant sites, no invariant
         * proportion, pattern weights of one - touched (and sent to the device again) only where they are not so */
        if (p->invariant)
        {
          free(p->invariant);
          p->invariant = NULL;
          pll_gpu_invalidate(p, PLL_GPU_DIRTY_INVARIANT, -1);
        }
        unsigned int what = 0;
        for (unsigned int f = 0; f < p->rate_matrices; ++f)
          if (p->prop_invar[f] != 0.0)
          {
            p->prop_invar[f] = 0.0;
            what |= PLL_GPU_DIRTY_FREQS;
          }
        for (unsigned int k = 0; k < p->sites; ++k)
          if (p->pattern_weights[k] != 1)
          {
            p->pattern_weights[k] = 1;
            what |= PLL_GPU_DIRTY_PATTERN_WEIGHTS;
          }
        if (what) pll_gpu_invalidate(p, what, -1);
        seam_slots[i].busy = 1;
        seam_slots[i].used = ++seam_clock;
        s->p = p;
        s->slot = i;
        s->span = p->states_padded * rate_cats;
        return 1;
      }
  /* one (unused) tip: CLV k of the seam is clv[k + 1] of the partition, scale buffer k is scale_buffer[k] */
  s->p = pll_partition_create(1, clvs, states, entries, key[5], matrices, rate_cats, clvs, attrs);
  if (!s->p)
  {
    fprintf(stderr, "libpll_amd: pll_core_*: [%d] %s\n", pll_errno, pll_errmsg);
    return 0;
  }
  s->span = s->p->states_padded * rate_cats;
  /* host mirror + HBM of the CLVs and scalers (what grows with the call's size; the context's pinned block is 8 MB) */
  const size_t bytes = 2u * (size_t)(clvs + 1u) * entries * ((size_t)s->span * sizeof(double) + rate_cats * sizeof(unsigned int)) + ((size_t)8 << 20);
  if (seam_cache_on() && bytes <= SEAM_KEEP_ONE)
  {
    /* within the byte bound: idle partitions go, least recently used first, until this one fits */
    for (;;)
    {
      size_t held = bytes;
      int lru = -1;
      for (int i = 0; i < SEAM_SLOTS; ++i)
      {
        if (!seam_slots[i].p) continue;
        held += seam_slots[i].bytes;
        if (!seam_slots[i].busy && (lru < 0 || seam_slots[i].used < seam_slots[lru].used)) lru = i;
      }
      if (held <= SEAM_KEEP_ALL || lru < 0) break;
      pll_partition_destroy(seam_slots[lru].p);
      seam_slots[lru].p = NULL;
    }
    /* a free slot, or the least recently used idle one */
    int at = -1;
    for (int i = 0; i < SEAM_SLOTS; ++i)
      if (!seam_slots[i].busy && (at < 0 || !seam_slots[i].p || (seam_slots[at].p && seam_slots[i].used < seam_slots[at].used))) at = i;
    if (at >= 0)
    {
      if (seam_slots[at].p) pll_partition_destroy(seam_slots[at].p);
      seam_slots[at].p = s->p;
      seam_slots[at].bytes = bytes;
      seam_slots[at].device = pllgpu_context_device(pll_ext(s->p)->ctx);
      memcpy(seam_slots[at].key, key, sizeof key);
      seam_slots[at].busy = 1;
      seam_slots[at].used = ++seam_clock;
      s->slot = at;
      pthread_once(&seam_once, seam_make_key);
      (void)pthread_setspecific(seam_key, seam_slots);
    }
  }
  return 1;
}

/* the calling thread's kept partitions (host mirrors, HBM, streams, pinned blocks) are given back now */
void pll_core_seam_release(void)
{
  for (int i = 0; i < SEAM_SLOTS; ++i)
    if (seam_slots[i].p && !seam_slots[i].busy)
    {
      pll_partition_destroy(seam_slots[i].p);
      seam_slots[i].p = NULL;
    }
}

static void seam_close(seam_t *s)
{
  if (s->p && s->slot >= 0)
    seam_slots[s->slot].busy = 0;
  else if (s->p)
    pll_partition_destroy(s->p);
  s->p = NULL;
}

static size_t scaler_words(const pll_partition_t *p) { return (p->attributes & PLL_ATTRIB_RATE_SCALERS) ? p->rate_cats : 1u; }

/* entry e of CLV `idx` <- entry src_entry of `clv` (or the whole vector when map == NULL) */
static void seam_put_clv(seam_t *s, unsigned int idx, const double *clv, const unsigned int *scaler, unsigned int entries,
                         const unsigned int *outer /* entry -> site or NULL */, const unsigned int *inner /* site -> source entry or NULL */)
{
  pll_partition_t *p = s->p;
  pll_amd_ext_t *x = pll_ext(p);
  const size_t sw = scaler_words(p);
  if (!outer && !inner && x && x->ctx && entries == pll_get_sites_number(p, idx + 1))
  {
    /* the caller's arrays as they are: straight to the device, the partition's host mirror left out (a 128 KB CLV
     * copied into the mirror and from there to where the device reads it was a fifth of a call at 1k sites) */
    if (pllgpu_clv_upload(x->ctx, idx + 1, clv, entries) == 0 &&
        (!scaler || pllgpu_scaler_upload(x->ctx, idx, scaler, entries) == 0))
    {
      x->clv_side[idx + 1] = SIDE_DEVICE;
      x->fast_valid = 0;
      if (scaler)
      {
        x->scaler_side[idx] = SIDE_DEVICE;
        x->scaler_entries[idx] = entries;
      }
      return;
    }
    /* (an error of the device layer: the mirror path below reports it) */
  }
  for (unsigned int e = 0; e < entries; ++e)
  {
    unsigned int src = outer ? outer[e] : e;
    if (inner) src = inner[src];
    memcpy(p->clv[idx + 1] + (size_t)e * s->span, clv + (size_t)src * s->span, s->span * sizeof(double));
    if (scaler) memcpy(p->scale_buffer[idx] + (size_t)e * sw, scaler + (size_t)src * sw, sw * sizeof(unsigned int));
  }
  pll_gpu_invalidate(p, PLL_GPU_DIRTY_CLV, (int)idx + 1);
  if (scaler) pll_gpu_invalidate(p, PLL_GPU_DIRTY_SCALER, (int)idx);
}

/* CLV `idx` <- the indicator vectors of encoded tip characters */
static void seam_put_tip(seam_t *s, unsigned int idx, const unsigned char *chars, const pll_state_t *tipmap, unsigned int entries)
{
  pll_partition_t *p = s->p;
  const unsigned int sp = p->states_padded;
  memset(p->clv[idx + 1], 0, (size_t)entries * s->span * sizeof(double));
  for (unsigned int e = 0; e < entries; ++e)
  {
    const pll_state_t mask = tipmap ? tipmap[chars[e]] : (pll_state_t)chars[e];
    for (unsigned int k = 0; k < p->rate_cats; ++k)
      for (unsigned int j = 0; j < p->states; ++j)
        if ((mask >> j) & 1) p->clv[idx + 1][(size_t)e * s->span + (size_t)k * sp + j] = 1.0;
  }
  pll_gpu_invalidate(p, PLL_GPU_DIRTY_CLV, (int)idx + 1);
}

static void seam_put_matrix(seam_t *s, unsigned int idx, const double *m)
{
  pll_partition_t *p = s->p;
  memcpy(p->pmatrix[idx], m, (size_t)p->rate_cats * p->states * p->states_padded * sizeof(double));
  pll_gpu_invalidate(p, PLL_GPU_DIRTY_PMATRIX, (int)idx);
}

/* clv[0] <- op(clv[1], clv[2]) and back to the caller */
static void seam_update(seam_t *s, double *parent_clv, unsigned int *parent_scaler, int lscal, int rscal, unsigned int entries)
{
  pll_partition_t *p = s->p;
  pll_operation_t op;
  op.parent_clv_index = 1;
  op.parent_scaler_index = parent_scaler ? 0 : PLL_SCALE_BUFFER_NONE;
  op.child1_clv_index = 2;
  op.child1_matrix_index = 0;
  op.child1_scaler_index = lscal ? 1 : PLL_SCALE_BUFFER_NONE;
  op.child2_clv_index = 3;
  op.child2_matrix_index = 1;
  op.child2_scaler_index = rscal ? 2 : PLL_SCALE_BUFFER_NONE;
  pll_errno = 0;
  pll_update_partials(p, &op, 1);
  /* the CLV and its scaler vector come back behind ONE wait, into the caller's arrays */
  pll_amd_ext_t *x = pll_ext(p);
  int ok = !pll_errno && x && x->ctx && pllgpu_download_defer(x->ctx, 1) == 0;
  ok = ok && pllgpu_clv_download(x->ctx, 1, parent_clv, entries) == 0 &&
       (!parent_scaler || pllgpu_scaler_download(x->ctx, 0, parent_scaler, entries) == 0);
  if (x && x->ctx && pllgpu_download_defer(x->ctx, 0) != 0) ok = 0;
  if (!ok)
  {
    if (!pll_errno) pll_set_gpu_error("pll_core_update_partial_*");
    fprintf(stderr, "libpll_amd: pll_core_update_partial_*: [%d] %s\n", pll_errno, pll_errmsg);
  }
}

void pll_core_update_partial_ii(unsigned int states, unsigned int sites, unsigned int rate_cats, double *parent_clv,
                                unsigned int *parent_scaler, const double *left_clv, const double *right_clv,
                                const double *left_matrix, const double *right_matrix, const unsigned int *left_scaler,
                                const unsigned int *right_scaler, unsigned int attrib)
{
  seam_t s;
  if (!seam_open(&s, states, sites, rate_cats, 3, 2, 1, attrib)) return;
  seam_put_clv(&s, 1, left_clv, left_scaler, sites, NULL, NULL);
  seam_put_clv(&s, 2, right_clv, right_scaler, sites, NULL, NULL);
  seam_put_matrix(&s, 0, left_matrix);
  seam_put_matrix(&s, 1, right_matrix);
  seam_update(&s, parent_clv, parent_scaler, left_scaler != NULL, right_scaler != NULL, sites);
  seam_close(&s);
}

void pll_core_update_partial_ti(unsigned int states, unsigned int sites, unsigned int rate_cats, double *parent_clv,
                                unsigned int *parent_scaler, const unsigned char *left_tipchars, const double *right_clv,
                                const double *left_matrix, const double *right_matrix, const unsigned int *right_scaler,
                                const pll_state_t *tipmap, unsigned int tipmap_size, unsigned int attrib)
{
  (void)tipmap_size;
  seam_t s;
  if (!seam_open(&s, states, sites, rate_cats, 3, 2, 1, attrib)) return;
  seam_put_tip(&s, 1, left_tipchars, tipmap, sites);
  seam_put_clv(&s, 2, right_clv, right_scaler, sites, NULL, NULL);
  seam_put_matrix(&s, 0, left_matrix);
  seam_put_matrix(&s, 1, right_matrix);
  seam_update(&s, parent_clv, parent_scaler, 0, right_scaler != NULL, sites);
  seam_close(&s);
}

void pll_core_update_partial_ti_4x4(unsigned int sites, unsigned int rate_cats, double *parent_clv, unsigned int *parent_scaler,
                                    const unsigned char *left_tipchars, const double *right_clv, const double *left_matrix,
                                    const double *right_matrix, const unsigned int *right_scaler, unsigned int attrib)
{
  pll_core_update_partial_ti(4, sites, rate_cats, parent_clv, parent_scaler, left_tipchars, right_clv, left_matrix, right_matrix,
                             right_scaler, NULL, 16, attrib);
}

/* doubles of one matrix in the caller's layout */
static size_t matrix_doubles(unsigned int states, unsigned int rate_cats, unsigned int attrib)
{
  unsigned int sp = states;
  const unsigned int arch = attrib & PLL_ATTRIB_ARCH_MASK;
  if (arch == PLL_ATTRIB_ARCH_SSE) sp = (states + 1) & ~1u;
  if (arch == PLL_ATTRIB_ARCH_AVX || arch == PLL_ATTRIB_ARCH_AVX2 || arch == PLL_ATTRIB_ARCH_AVX512) sp = (states + 3) & ~3u;
  return (size_t)rate_cats * states * sp;
}

/* The tip-tip pair of the flat seam (src/pll.h:1049-1071). pll_core_create_lookup fills the CALLER's table in the
 * reference's own layout (src/core_partials.c:1013-1071 for 4 states, :1149-1209 otherwise; stride states_padded of
 * `attrib` as the vectorised forms, src/core_partials_avx.c:59-118): entry (j, k) of two tip codes at index
 * (j << ceil(log2(tipmap_size))) + k - 16 j + k for 4 states - holds the parent entry of a cherry showing j and k.
 * pll_core_update_partial_tt copies one entry per site and zeroes the scaler (:180-199). A table made by the reference
 * works with this library's tt and vice versa; both run on the device (k_create_lookup, k_tt_from_lookup). */
void pll_core_create_lookup(unsigned int states, unsigned int rate_cats, double *lookup, const double *left_matrix,
                            const double *right_matrix, const pll_state_t *tipmap, unsigned int tipmap_size, unsigned int attrib)
{
  seam_t s;
  if (!seam_open(&s, states, 1, rate_cats, 1, 1, 1, attrib)) return;
  pll_amd_ext_t *x = pll_ext(s.p);
  if (!x || !x->ctx || pllgpu_create_lookup(x->ctx, lookup, left_matrix, right_matrix, tipmap, tipmap_size) != 0)
  {
    pll_set_gpu_error("pll_core_create_lookup");
    fprintf(stderr, "libpll_amd: pll_core_create_lookup: [%d] %s\n", pll_errno, pll_errmsg);
  }
  seam_close(&s);
}

void pll_core_create_lookup_4x4(unsigned int rate_cats, double *lookup, const double *left_matrix, const double *right_matrix)
{
  /* the 4x4 form has no attrib argument: the table holds 4-double rows whatever the architecture
   * (src/core_partials.c:1015-1071); so do the matrices of a 4-state partition */
  pll_core_create_lookup(4, rate_cats, lookup, left_matrix, right_matrix, NULL, 16, PLL_ATTRIB_ARCH_CPU);
}

void pll_core_update_partial_tt(unsigned int states, unsigned int sites, unsigned int rate_cats, double *parent_clv,
                                unsigned int *parent_scaler, const unsigned char *left_tipchars,
                                const unsigned char *right_tipchars, const pll_state_t *tipmap, unsigned int tipmap_size,
                                const double *lookup, unsigned int attrib)
{
  (void)tipmap;
  seam_t s;
  if (!seam_open(&s, states, 1, rate_cats, 1, 1, 1, attrib)) return;
  pll_amd_ext_t *x = pll_ext(s.p);
  if (!x || !x->ctx || pllgpu_tt_from_lookup(x->ctx, parent_clv, left_tipchars, right_tipchars, lookup, sites, tipmap_size) != 0)
  {
    pll_set_gpu_error("pll_core_update_partial_tt");
    fprintf(stderr, "libpll_amd: pll_core_update_partial_tt: [%d] %s\n", pll_errno, pll_errmsg);
  }
  else if (parent_scaler)
    memset(parent_scaler, 0, sizeof(unsigned int) * ((attrib & PLL_ATTRIB_RATE_SCALERS) ? (size_t)sites * rate_cats : sites));
  seam_close(&s);
}

void pll_core_update_partial_tt_4x4(unsigned int sites, unsigned int rate_cats, double *parent_clv, unsigned int *parent_scaler,
                                    const unsigned char *left_tipchars, const unsigned char *right_tipchars, const double *lookup,
                                    unsigned int attrib)
{
  pll_core_update_partial_tt(4, sites, rate_cats, parent_clv, parent_scaler, left_tipchars, right_tipchars, NULL, 16, lookup, attrib);
}

void pll_core_update_partial_repeats_generic(unsigned int states, unsigned int parent_sites, unsigned int left_sites,
                                             unsigned int right_sites, unsigned int rate_cats, double *parent_clv,
                                             unsigned int *parent_scaler, const double *left_clv, const double *right_clv,
                                             const double *left_matrix, const double *right_matrix, const unsigned int *left_scaler,
                                             const unsigned int *right_scaler, const unsigned int *parent_id_site,
                                             const unsigned int *left_site_id, const unsigned int *right_site_id,
                                             double *bclv_buffer, unsigned int attrib)
{
  (void)left_sites;
  (void)right_sites;
  (void)bclv_buffer;
  seam_t s;
  if (!seam_open(&s, states, parent_sites, rate_cats, 3, 2, 1, attrib)) return;
  /* parent entry n stands for site id_site[n]; its children's entries are site_id[site] (PLL_GET_SITE / PLL_GET_ID, src/pll.h:682-683) */
  seam_put_clv(&s, 1, left_clv, left_scaler, parent_sites, parent_id_site, left_site_id);
  seam_put_clv(&s, 2, right_clv, right_scaler, parent_sites, parent_id_site, right_site_id);
  seam_put_matrix(&s, 0, left_matrix);
  seam_put_matrix(&s, 1, right_matrix);
  seam_update(&s, parent_clv, parent_scaler, left_scaler != NULL, right_scaler != NULL, parent_sites);
  seam_close(&s);
}

void pll_core_update_partial_repeats(unsigned int states, unsigned int parent_sites, unsigned int left_sites, unsigned int right_sites,
                                     unsigned int rate_cats, double *parent_clv, unsigned int *parent_scaler, const double *left_clv,
                                     const double *right_clv, const double *left_matrix, const double *right_matrix,
                                     const unsigned int *left_scaler, const unsigned int *right_scaler,
                                     const unsigned int *parent_id_site, const unsigned int *left_site_id,
                                     const unsigned int *right_site_id, double *bclv_buffer, unsigned int attrib)
{
  pll_core_update_partial_repeats_generic(states, parent_sites, left_sites, right_sites, rate_cats, parent_clv, parent_scaler, left_clv,
                                          right_clv, left_matrix, right_matrix, left_scaler, right_scaler, parent_id_site, left_site_id,
                                          right_site_id, bclv_buffer, attrib);
}

void pll_core_update_partial_repeatsbclv_generic(unsigned int states, unsigned int parent_sites, unsigned int left_sites,
                                                 unsigned int right_sites, unsigned int rate_cats, double *parent_clv,
                                                 unsigned int *parent_scaler, const double *left_clv, const double *right_clv,
                                                 const double *left_matrix, const double *right_matrix,
                                                 const unsigned int *left_scaler, const unsigned int *right_scaler,
                                                 const unsigned int *parent_id_site, const unsigned int *left_site_id,
                                                 const unsigned int *right_site_id, double *bclv_buffer, unsigned int attrib)
{
  pll_core_update_partial_repeats_generic(states, parent_sites, left_sites, right_sites, rate_cats, parent_clv, parent_scaler, left_clv,
                                          right_clv, left_matrix, right_matrix, left_scaler, right_scaler, parent_id_site, left_site_id,
                                          right_site_id, bclv_buffer, attrib);
}

/* ---- log-likelihoods --------------------------------------------------------------------------- */

static unsigned int freq_sets(const unsigned int *freqs_indices, unsigned int rate_cats)
{
  unsigned int n = 0;
  for (unsigned int k = 0; k < rate_cats; ++k)
    if (freqs_indices[k] + 1 > n) n = freqs_indices[k] + 1;
  return n;
}

static int seam_model(seam_t *s, double *const *frequencies, const double *rate_weights, const unsigned int *pattern_weights,
                      const double *invar_proportion, const int *invar_indices, const unsigned int *freqs_indices, unsigned int sites)
{
  pll_partition_t *p = s->p;
  const unsigned int sets = freq_sets(freqs_indices, p->rate_cats);
  for (unsigned int f = 0; f < sets; ++f)
  {
    pll_set_frequencies(p, f, frequencies[f]);
    if (invar_proportion) p->prop_invar[f] = invar_proportion[f];
  }
  pll_gpu_invalidate(p, PLL_GPU_DIRTY_FREQS, -1);
  pll_set_category_weights(p, rate_weights);
  pll_set_pattern_weights(p, pattern_weights);
  if (invar_indices)
  {
    if (!p->invariant) p->invariant = (int *)malloc((size_t)sites * sizeof(int));
    if (!p->invariant)
    {
      pll_set_error(PLL_ERROR_MEM_ALLOC, "pll_core_*_loglikelihood: out of memory");
      return 0;
    }
    memcpy(p->invariant, invar_indices, (size_t)sites * sizeof(int));
    pll_gpu_invalidate(p, PLL_GPU_DIRTY_INVARIANT, -1);
  }
  return 1;
}

static double seam_edge(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                        const unsigned int *parent_scaler, const unsigned int *parent_site_id, const double *child_clv,
                        const unsigned int *child_scaler, const unsigned int *child_site_id, const unsigned char *tipchars,
                        const pll_state_t *tipmap, const double *pmatrix, double *const *frequencies, const double *rate_weights,
                        const unsigned int *pattern_weights, const double *invar_proportion, const int *invar_indices,
                        const unsigned int *freqs_indices, double *persite_lnl, unsigned int attrib, int root)
{
  seam_t s;
  double lnl = -INFINITY;
  if (!seam_open(&s, states, sites, rate_cats, 2, 1, freq_sets(freqs_indices, rate_cats), attrib)) return lnl;
  if (seam_model(&s, frequencies, rate_weights, pattern_weights, invar_proportion, invar_indices, freqs_indices, sites))
  {
    seam_put_clv(&s, 0, parent_clv, parent_scaler, sites, NULL, parent_site_id);
    if (root)
      lnl = pll_compute_root_loglikelihood(s.p, 1, parent_scaler ? 0 : PLL_SCALE_BUFFER_NONE, freqs_indices, persite_lnl);
    else
    {
      if (tipchars)
        seam_put_tip(&s, 1, tipchars, tipmap, sites);
      else
        seam_put_clv(&s, 1, child_clv, child_scaler, sites, NULL, child_site_id);
      seam_put_matrix(&s, 0, pmatrix);
      lnl = pll_compute_edge_loglikelihood(s.p, 1, parent_scaler ? 0 : PLL_SCALE_BUFFER_NONE, 2,
                                           (!tipchars && child_scaler) ? 1 : PLL_SCALE_BUFFER_NONE, 0, freqs_indices, persite_lnl);
    }
  }
  seam_close(&s);
  return lnl;
}

double pll_core_edge_loglikelihood_ii(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                      const unsigned int *parent_scaler, const double *child_clv, const unsigned int *child_scaler,
                                      const double *pmatrix, double *const *frequencies, const double *rate_weights,
                                      const unsigned int *pattern_weights, const double *invar_proportion, const int *invar_indices,
                                      const unsigned int *freqs_indices, double *persite_lnl, unsigned int attrib)
{
  return seam_edge(states, sites, rate_cats, parent_clv, parent_scaler, NULL, child_clv, child_scaler, NULL, NULL, NULL, pmatrix,
                   frequencies, rate_weights, pattern_weights, invar_proportion, invar_indices, freqs_indices, persite_lnl, attrib, 0);
}

double pll_core_edge_loglikelihood_ti(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                      const unsigned int *parent_scaler, const unsigned char *tipchars, const pll_state_t *tipmap,
                                      unsigned int tipmap_size, const double *pmatrix, double *const *frequencies,
                                      const double *rate_weights, const unsigned int *pattern_weights, const double *invar_proportion,
                                      const int *invar_indices, const unsigned int *freqs_indices, double *persite_lnl,
                                      unsigned int attrib)
{
  (void)tipmap_size;
  return seam_edge(states, sites, rate_cats, parent_clv, parent_scaler, NULL, NULL, NULL, NULL, tipchars, tipmap, pmatrix, frequencies,
                   rate_weights, pattern_weights, invar_proportion, invar_indices, freqs_indices, persite_lnl, attrib, 0);
}

double pll_core_edge_loglikelihood_ti_4x4(unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                          const unsigned int *parent_scaler, const unsigned char *tipchars, const double *pmatrix,
                                          double *const *frequencies, const double *rate_weights, const unsigned int *pattern_weights,
                                          const double *invar_proportion, const int *invar_indices, const unsigned int *freqs_indices,
                                          double *persite_lnl, unsigned int attrib)
{
  return seam_edge(4, sites, rate_cats, parent_clv, parent_scaler, NULL, NULL, NULL, NULL, tipchars, NULL, pmatrix, frequencies,
                   rate_weights, pattern_weights, invar_proportion, invar_indices, freqs_indices, persite_lnl, attrib, 0);
}

double pll_core_edge_loglikelihood_repeats_generic(unsigned int states, unsigned int sites, const unsigned int child_sites,
                                                   unsigned int rate_cats, const double *parent_clv, const unsigned int *parent_scaler,
                                                   const double *child_clv, const unsigned int *child_scaler, const double *pmatrix,
                                                   double **frequencies, const double *rate_weights, const unsigned int *pattern_weights,
                                                   const double *invar_proportion, const int *invar_indices,
                                                   const unsigned int *freqs_indices, double *persite_lnl,
                                                   const unsigned int *parent_site_id, const unsigned int *child_site_id, double *bclv,
                                                   unsigned int attrib)
{
  (void)child_sites;
  (void)bclv;
  return seam_edge(states, sites, rate_cats, parent_clv, parent_scaler, parent_site_id, child_clv, child_scaler, child_site_id, NULL, NULL,
                   pmatrix, frequencies, rate_weights, pattern_weights, invar_proportion, invar_indices, freqs_indices, persite_lnl, attrib, 0);
}

double pll_core_edge_loglikelihood_repeats(unsigned int states, unsigned int sites, const unsigned int child_sites, unsigned int rate_cats,
                                           const double *parent_clv, const unsigned int *parent_scaler, const double *child_clv,
                                           const unsigned int *child_scaler, const double *pmatrix, double **frequencies,
                                           const double *rate_weights, const unsigned int *pattern_weights, const double *invar_proportion,
                                           const int *invar_indices, const unsigned int *freqs_indices, double *persite_lnl,
                                           const unsigned int *parent_site_id, const unsigned int *child_site_id, double *bclv,
                                           unsigned int attrib)
{
  return pll_core_edge_loglikelihood_repeats_generic(states, sites, child_sites, rate_cats, parent_clv, parent_scaler, child_clv,
                                                     child_scaler, pmatrix, frequencies, rate_weights, pattern_weights, invar_proportion,
                                                     invar_indices, freqs_indices, persite_lnl, parent_site_id, child_site_id, bclv, attrib);
}

double pll_core_root_loglikelihood(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *clv,
                                   const unsigned int *scaler, double *const *frequencies, const double *rate_weights,
                                   const unsigned int *pattern_weights, const double *invar_proportion, const int *invar_indices,
                                   const unsigned int *freqs_indices, double *persite_lnl, unsigned int attrib)
{
  return seam_edge(states, sites, rate_cats, clv, scaler, NULL, NULL, NULL, NULL, NULL, NULL, NULL, frequencies, rate_weights,
                   pattern_weights, invar_proportion, invar_indices, freqs_indices, persite_lnl, attrib, 1);
}

double pll_core_root_loglikelihood_repeats(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *clv,
                                           const unsigned int *site_id, const unsigned int *scaler, double *const *frequencies,
                                           const double *rate_weights, const unsigned int *pattern_weights,
                                           const double *invar_proportion, const int *invar_indices, const unsigned int *freqs_indices,
                                           double *persite_lnl, unsigned int attrib)
{
  return seam_edge(states, sites, rate_cats, clv, scaler, site_id, NULL, NULL, NULL, NULL, NULL, NULL, frequencies, rate_weights,
                   pattern_weights, invar_proportion, invar_indices, freqs_indices, persite_lnl, attrib, 1);
}

/* ---- derivatives and transition matrices (reference: src/pll.h:1181-1273, :2400-2412; bodies in
 * src/core_derivatives.c and src/core_pmatrix.c). The flat functions take the model per RATE CATEGORY
 * (eigenvecs[k], freqs[k], prop_invar[k]: the partition-level callers have applied params_indices
 * already, src/derivatives.c:56-66), so the seam partition holds one rate matrix per category and the
 * identity for params_indices. ------------------------------------------------------------------- */
static unsigned int *seam_identity(unsigned int n)
{
  unsigned int *idx = (unsigned int *)malloc((n ? n : 1) * sizeof(unsigned int));
  if (!idx) pll_set_error(PLL_ERROR_MEM_ALLOC, "pll_core_*: out of memory");
  for (unsigned int k = 0; idx && k < n; ++k) idx[k] = k;
  return idx;
}

/* eigensystem of rate matrix `set` <- the caller's arrays (any of them may be NULL: not needed by the call) */
static void seam_put_eigen(seam_t *s, unsigned int set, const double *eigenvecs, const double *inv_eigenvecs, const double *eigenvals)
{
  pll_partition_t *p = s->p;
  const size_t sp = p->states_padded;
  if (eigenvecs) memcpy(p->eigenvecs[set], eigenvecs, p->states * sp * sizeof(double));
  if (inv_eigenvecs) memcpy(p->inv_eigenvecs[set], inv_eigenvecs, p->states * sp * sizeof(double));
  if (eigenvals) memcpy(p->eigenvals[set], eigenvals, p->states * sizeof(double));
  p->eigen_decomp_valid[set] = 1;
  pll_gpu_invalidate(p, PLL_GPU_DIRTY_EIGEN, (int)set);
}

static int seam_sumtable(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *left_clv,
                         const unsigned int *left_scaler, const unsigned int *left_site_id, const unsigned char *left_tipchars,
                         const pll_state_t *tipmap, const double *right_clv, const unsigned int *right_scaler,
                         const unsigned int *right_site_id, double *const *eigenvecs, double *const *inv_eigenvecs,
                         double *const *freqs, double *sumtable, unsigned int attrib)
{
  seam_t s;
  unsigned int *ident = seam_identity(rate_cats);
  int rc = PLL_FAILURE;
  if (!ident) return PLL_FAILURE;
  if (!seam_open(&s, states, sites, rate_cats, 2, 1, rate_cats, attrib))
  {
    free(ident);
    return PLL_FAILURE;
  }
  for (unsigned int k = 0; k < rate_cats; ++k)
  {
    pll_set_frequencies(s.p, k, freqs[k]);
    seam_put_eigen(&s, k, eigenvecs[k], inv_eigenvecs[k], NULL);
  }
  /* the left operand meets freqs x inv_eigenvecs, the right one eigenvecs (src/core_derivatives.c:440-456);
   * a tip given by characters is the left one (:608-627) */
  if (left_tipchars)
    seam_put_tip(&s, 0, left_tipchars, tipmap, sites);
  else
    seam_put_clv(&s, 0, left_clv, left_scaler, sites, NULL, left_site_id);
  seam_put_clv(&s, 1, right_clv, right_scaler, sites, NULL, right_site_id);
  rc = pll_update_sumtable(s.p, 1, 2, (!left_tipchars && left_scaler) ? 0 : PLL_SCALE_BUFFER_NONE,
                           right_scaler ? 1 : PLL_SCALE_BUFFER_NONE, ident, sumtable);
  if (rc) rc = pll_gpu_sync_sumtable(s.p, sumtable);
  if (!rc) fprintf(stderr, "libpll_amd: pll_core_update_sumtable_*: [%d] %s\n", pll_errno, pll_errmsg);
  seam_close(&s);
  free(ident);
  return rc;
}

int pll_core_update_sumtable_ii(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                const double *child_clv, const unsigned int *parent_scaler, const unsigned int *child_scaler,
                                double *const *eigenvecs, double *const *inv_eigenvecs, double *const *freqs, double *sumtable,
                                unsigned int attrib)
{
  return seam_sumtable(states, sites, rate_cats, parent_clv, parent_scaler, NULL, NULL, NULL, child_clv, child_scaler, NULL,
                       eigenvecs, inv_eigenvecs, freqs, sumtable, attrib);
}

int pll_core_update_sumtable_ti(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                const unsigned char *left_tipchars, const unsigned int *parent_scaler, double *const *eigenvecs,
                                double *const *inv_eigenvecs, double *const *freqs, const pll_state_t *tipmap,
                                unsigned int tipmap_size, double *sumtable, unsigned int attrib)
{
  (void)tipmap_size;
  return seam_sumtable(states, sites, rate_cats, NULL, NULL, NULL, left_tipchars, tipmap, parent_clv, parent_scaler, NULL,
                       eigenvecs, inv_eigenvecs, freqs, sumtable, attrib);
}

int pll_core_update_sumtable_repeats_generic(unsigned int states, unsigned int sites, unsigned int parent_sites,
                                             unsigned int rate_cats, const double *clvp, const double *clvc,
                                             const unsigned int *parent_scaler, const unsigned int *child_scaler,
                                             double *const *eigenvecs, double *const *inv_eigenvecs, double *const *freqs,
                                             double *sumtable, const unsigned int *parent_site_id,
                                             const unsigned int *child_site_id, double *bclv_buffer, unsigned int inv,
                                             unsigned int attrib)
{
  /* parent_sites / bclv_buffer / inv steer the reference's choice of loop (a per-class pre-product of the
   * smaller operand, src/core_derivatives.c:60); the table is the same, the operands are expanded here */
  (void)parent_sites; (void)bclv_buffer; (void)inv;
  return seam_sumtable(states, sites, rate_cats, clvp, parent_scaler, parent_site_id, NULL, NULL, clvc, child_scaler,
                       child_site_id, eigenvecs, inv_eigenvecs, freqs, sumtable, attrib);
}

int pll_core_update_sumtable_repeats(unsigned int states, unsigned int sites, unsigned int parent_sites, unsigned int rate_cats,
                                     const double *clvp, const double *clvc, const unsigned int *parent_scaler,
                                     const unsigned int *child_scaler, double *const *eigenvecs, double *const *inv_eigenvecs,
                                     double *const *freqs, double *sumtable, const unsigned int *parent_site_id,
                                     const unsigned int *child_site_id, double *bclv_buffer, unsigned int inv, unsigned int attrib)
{
  return pll_core_update_sumtable_repeats_generic(states, sites, parent_sites, rate_cats, clvp, clvc, parent_scaler, child_scaler,
                                                  eigenvecs, inv_eigenvecs, freqs, sumtable, parent_site_id, child_site_id,
                                                  bclv_buffer, inv, attrib);
}

int pll_core_likelihood_derivatives(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *rate_weights,
                                    const unsigned int *parent_scaler, const unsigned int *child_scaler, unsigned int parent_ids,
                                    unsigned int child_ids, const int *invariant, const unsigned int *pattern_weights,
                                    double branch_length, const double *prop_invar, double *const *freqs, const double *rates,
                                    double *const *eigenvals, const double *sumtable, double *d_f, double *dd_f,
                                    unsigned int attrib)
{
  seam_t s;
  unsigned int *ident;
  int rc = PLL_FAILURE;
  /* the scalers only enter the Lewis / Felsenstein ascertainment terms (src/core_derivatives.c:851-924),
   * which need the partition's extra entries: that form goes through pll_compute_likelihood_derivatives */
  (void)parent_scaler; (void)child_scaler; (void)parent_ids; (void)child_ids;
  if (attrib & PLL_ATTRIB_AB_MASK)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_core_likelihood_derivatives: the ascertainment bias correction is served by "
                  "pll_compute_likelihood_derivatives only");
    fprintf(stderr, "libpll_amd: pll_core_likelihood_derivatives: [%d] %s\n", pll_errno, pll_errmsg);
    return PLL_FAILURE;
  }
  if (!(ident = seam_identity(rate_cats))) return PLL_FAILURE;
  if (!seam_open(&s, states, sites, rate_cats, 1, 1, rate_cats, attrib))
  {
    free(ident);
    return PLL_FAILURE;
  }
  if (seam_model(&s, freqs, rate_weights, pattern_weights, prop_invar, invariant, ident, sites))
  {
    pll_set_category_rates(s.p, rates);
    for (unsigned int k = 0; k < rate_cats; ++k) seam_put_eigen(&s, k, NULL, NULL, eigenvals[k]);
    rc = pll_compute_likelihood_derivatives(s.p, PLL_SCALE_BUFFER_NONE, PLL_SCALE_BUFFER_NONE, branch_length, ident, sumtable,
                                            d_f, dd_f);
  }
  if (!rc) fprintf(stderr, "libpll_amd: pll_core_likelihood_derivatives: [%d] %s\n", pll_errno, pll_errmsg);
  seam_close(&s);
  free(ident);
  return rc;
}

int pll_core_update_pmatrix(double **pmatrix, unsigned int states, unsigned int rate_cats, const double *rates,
                            const double *branch_lengths, const unsigned int *matrix_indices, const unsigned int *params_indices,
                            const double *prop_invar, double *const *eigenvals, double *const *eigenvecs,
                            double *const *inv_eigenvecs, unsigned int count, unsigned int attrib)
{
  /* here the arrays ARE indexed through params_indices (src/core_pmatrix.c:205-215), and pmatrix[] through
   * matrix_indices: the seam partition mirrors both index spaces */
  seam_t s;
  unsigned int sets = 0, matrices = 0, n;
  int rc = PLL_FAILURE;
  for (n = 0; n < rate_cats; ++n)
    if (params_indices[n] + 1 > sets) sets = params_indices[n] + 1;
  for (n = 0; n < count; ++n)
    if (matrix_indices[n] + 1 > matrices) matrices = matrix_indices[n] + 1;
  if (!count) return PLL_SUCCESS;
  if (!seam_open(&s, states, 1, rate_cats, 1, matrices, sets, attrib)) return PLL_FAILURE;
  pll_set_category_rates(s.p, rates);
  for (n = 0; n < rate_cats; ++n)
  {
    const unsigned int set = params_indices[n];
    seam_put_eigen(&s, set, eigenvecs[set], inv_eigenvecs[set], eigenvals[set]);
    s.p->prop_invar[set] = prop_invar[set];
  }
  rc = pll_update_prob_matrices(s.p, params_indices, matrix_indices, branch_lengths, count);
  for (n = 0; rc && n < count; ++n)
  {
    rc = pll_gpu_sync_pmatrix(s.p, (int)matrix_indices[n]);
    if (rc) memcpy(pmatrix[matrix_indices[n]], s.p->pmatrix[matrix_indices[n]], matrix_doubles(states, rate_cats, attrib) * sizeof(double));
  }
  if (!rc) fprintf(stderr, "libpll_amd: pll_core_update_pmatrix: [%d] %s\n", pll_errno, pll_errmsg);
  seam_close(&s);
  return rc;
}

/* 4-state form of the tip case: the characters ARE the state masks (src/core_derivatives.c:474-560) */
int pll_core_update_sumtable_ti_4x4(unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                    const unsigned char *left_tipchars, const unsigned int *parent_scaler,
                                    double *const *eigenvecs, double *const *inv_eigenvecs, double *const *freqs,
                                    double *sumtable, unsigned int attrib)
{
  return seam_sumtable(4, sites, rate_cats, NULL, NULL, NULL, left_tipchars, NULL, parent_clv, parent_scaler, NULL, eigenvecs,
                       inv_eigenvecs, freqs, sumtable, attrib);
}

/* exported by the reference without a declaration in pll.h (src/core_likelihood.c:211-223): unpadded layout */
double pll_core_root_loglikelihood_repeats_generic(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                                   const double *clv, const unsigned int *site_id, const unsigned int *scaler,
                                                   double *const *frequencies, const double *rate_weights,
                                                   const unsigned int *pattern_weights, const double *invar_proportion,
                                                   const int *invar_indices, const unsigned int *freqs_indices, double *persite_lnl)
{
  return pll_core_root_loglikelihood_repeats(states, sites, rate_cats, clv, site_id, scaler, frequencies, rate_weights,
                                             pattern_weights, invar_proportion, invar_indices, freqs_indices, persite_lnl,
                                             PLL_ATTRIB_ARCH_CPU);
}
