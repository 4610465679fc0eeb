/* tips.c - tip data: character-state codes (PATTERN_TIP) or 0/1 tip CLVs.
 *
 * Semantics follow src/pll.c:157-422 (code tables) and :875-1129 (encoders):
 *   - PATTERN_TIP, 4 states: the stored code IS the 4-bit state mask; maxstates = max mask + 1.
 *   - PATTERN_TIP, other:    every distinct state mask gets a small integer code in order of
 *     first appearance over the ASCII table (charmap: char -> code, tipmap: code -> mask);
 *     a later sequence with a different map extends the tables.
 *   - otherwise a tip is an ordinary CLV holding the mask's bits as 0.0/1.0, replicated per rate.
 * The codes / CLVs are written to the host arrays (callers read them, and
 * pll_update_invariant_sites needs them) and marked for upload; the device consumes them on the
 * next pll_update_partials.
 */
#include "pll_internal.h"

static unsigned int ceil_log2(unsigned int v)
{
  unsigned int l = 0;
  while ((1u << l) < v) ++l;
  return l;
}

/* (re)build charmap/tipmap for `map`; first call allocates the tables and the per-tip arrays */
static int register_map(pll_partition_t *p, const pll_state_t *map)
{
  unsigned int i, j, k;
  const int first = (p->tipchars == NULL);
  pll_state_t local[PLL_ASCII_SIZE];
  memcpy(local, map, sizeof local);

  if (first)
  {
    p->charmap = (unsigned char *)calloc(PLL_ASCII_SIZE, 1);
    p->tipmap = (pll_state_t *)calloc(PLL_ASCII_SIZE, sizeof(pll_state_t));
    if (!p->charmap || !p->tipmap)
    {
      pll_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate charmap for tip-tip precomputation.");
      return PLL_FAILURE;
    }
  }
  /* number of codes already handed out */
  k = 0;
  while (k < PLL_ASCII_SIZE && p->tipmap[k]) ++k;
  const unsigned int known = k;

  /* how many masks of this map are new? */
  unsigned int fresh = 0;
  for (i = 0; i < PLL_ASCII_SIZE; ++i)
  {
    if (!local[i]) continue;
    for (j = 0; j < known; ++j)
      if (p->tipmap[j] == local[i]) break;
    if (j < known) continue;
    for (j = 0; j < i; ++j)
      if (local[j] == local[i]) break;
    if (j == i) ++fresh;
  }
  if (known + fresh >= PLL_ASCII_SIZE)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "Cannot specify 256 or more states with PLL_ATTRIB_PATTERN_TIP.");
    return PLL_FAILURE;
  }

  memset(p->charmap, 0, PLL_ASCII_SIZE);
  pll_state_t maxmask = 0;
  for (i = 0; i < PLL_ASCII_SIZE; ++i)
  {
    if (!local[i]) continue;
    if (local[i] > maxmask) maxmask = local[i];
    unsigned int code;
    for (code = 0; code < k; ++code)
      if (p->tipmap[code] == local[i]) break;
    if (code == k) p->tipmap[k++] = local[i];
    /* every character carrying this mask shares the code */
    for (j = i; j < PLL_ASCII_SIZE; ++j)
      if (local[j] == local[i])
      {
        p->charmap[j] = (unsigned char)code;
        if (j != i) local[j] = 0;
      }
  }

  if (first || fresh)
  {
    if (p->states == 4)
    {
      /* no remapping for DNA: codes are masks, so the table must span the largest mask */
      pll_state_t m = 0;
      for (i = 0; i < k; ++i)
        if (p->tipmap[i] > m) m = p->tipmap[i];
      p->maxstates = (unsigned int)m + 1;
    }
    else
      p->maxstates = k;
  }
  (void)maxmask;
  (void)ceil_log2;

  if (first)
  {
    /* ttlookup stays NULL: the device kernels evaluate tip-tip products directly
     * (src/pll.c:360-395 sizes a table the MI355X path has no use for) */
    const unsigned int n = pll_sites_alloc(p);
    p->tipchars = (unsigned char **)calloc(p->tips ? p->tips : 1, sizeof(unsigned char *));
    if (!p->tipchars)
    {
      pll_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate space for storing tip characters.");
      return PLL_FAILURE;
    }
    for (i = 0; i < p->tips; ++i)
    {
      p->tipchars[i] = (unsigned char *)malloc(n ? n : 1);
      if (!p->tipchars[i])
      {
        pll_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate space for storing tip characters.");
        return PLL_FAILURE;
      }
    }
  }
  return PLL_SUCCESS;
}

static int illegal_state(char c)
{
  pll_set_error(PLL_ERROR_TIPDATA_ILLEGALSTATE, "Illegal state code in tip \"%c\"", c);
  return PLL_FAILURE;
}

static int encode_tipchars(pll_partition_t *p, unsigned int tip, const pll_state_t *map, const char *seq)
{
  unsigned int i;
  unsigned char *dst = p->tipchars[tip];
  for (i = 0; i < p->sites; ++i)
  {
    const pll_state_t m = map[(unsigned char)seq[i]];
    if (!m) return illegal_state(seq[i]);
    dst[i] = (p->states == 4) ? (unsigned char)m : p->charmap[(unsigned char)seq[i]];
  }
  if (p->asc_bias_alloc)
  {
    /* the per-state extra sites: state i at position sites + i (src/pll.c:897-905, :935-953) */
    dst += p->sites;
    memset(dst, 0, p->states);
    if (p->states == 4)
      for (i = 0; i < 4; ++i) dst[i] = (unsigned char)(1u << i);
    else
      for (i = 0; i < p->maxstates; ++i)
      {
        const pll_state_t st = p->tipmap[i];
        if (__builtin_popcountll(st) == 1 && (unsigned int)__builtin_ctzll(st) < p->states)
          dst[__builtin_ctzll(st)] = (unsigned char)i;
      }
  }
  return PLL_SUCCESS;
}

/* the per-state extra sites of a tip CLV: indicator of state i at entry first + i
 * (src/pll.c:1003-1021, :1113-1126) */
static void asc_tip_entries(const pll_partition_t *p, double *clv, unsigned int first)
{
  const size_t span = (size_t)p->rate_cats * p->states_padded;
  unsigned int i, j, k;
  for (i = 0; i < p->states; ++i)
    for (k = 0; k < p->rate_cats; ++k)
      for (j = 0; j < p->states; ++j) clv[(first + i) * span + (size_t)k * p->states_padded + j] = (j == i) ? 1.0 : 0.0;
}

static void spread_mask(const pll_partition_t *p, pll_state_t m, double *dst)
{
  /* one (site) entry: the mask's bits for rate 0, copied to the other rates (src/pll.c:984-1000) */
  unsigned int j, k;
  for (j = 0; j < p->states; ++j, m >>= 1) dst[j] = (double)(m & 1);
  for (k = 1; k < p->rate_cats; ++k) memcpy(dst + (size_t)k * p->states_padded, dst, p->states * sizeof(double));
}

/* one-byte code of a state mask for the device: the mask itself for 4 states, otherwise its
 * index in a per-partition table of the masks seen so far; -1 when the table is full */
static int compact_code(pll_partition_t *p, pll_amd_ext_t *x, pll_state_t m)
{
  unsigned int c;
  if (p->states == 4) return (int)m;
  for (c = 0; c < x->ctip_count; ++c)
    if (x->ctipmap[c] == m) return (int)c;
  if (x->ctip_count >= PLL_ASCII_SIZE) return -1;
  x->ctipmap[x->ctip_count] = m;
  return (int)x->ctip_count++;
}

static int encode_tipclv(pll_partition_t *p, unsigned int tip, const pll_state_t *map, const char *seq)
{
  const int rep = pll_repeats_enabled(p);
  const unsigned int n = rep ? p->repeats->pernode_ids[tip] : p->sites;
  const size_t span = (size_t)p->rate_cats * p->states_padded;
  unsigned int i;
  double *clv = p->clv[tip];
  pll_amd_ext_t *x = pll_ext(p);
  int compact = x && !x->no_tip_codes;
  if (compact)
  {
    free(x->tipcodes[tip]);
    x->tipcodes[tip] = (unsigned char *)malloc((size_t)n + p->states);
    compact = x->tipcodes[tip] != NULL;
  }
  for (i = 0; i < n; ++i)
  {
    const unsigned int site = rep ? p->repeats->pernode_id_site[tip][i] : i;
    const pll_state_t m = map[(unsigned char)seq[site]];
    if (!m) return illegal_state(seq[site]);
    spread_mask(p, m, clv + i * span);
    if (compact)
    {
      const int code = compact_code(p, x, m);
      if (code < 0) compact = 0;
      else x->tipcodes[tip][i] = (unsigned char)code;
    }
  }
  if (p->asc_bias_alloc)
  {
    asc_tip_entries(p, clv, n);
    for (i = 0; compact && i < p->states; ++i)
    {
      const int code = compact_code(p, x, (pll_state_t)1 << i);
      if (code < 0) compact = 0;
      else x->tipcodes[tip][n + i] = (unsigned char)code;
    }
  }
  if (x) x->tip_compact[tip] = (unsigned char)compact;
  if (x) x->fast_valid = 0;
  return PLL_SUCCESS;
}

int pll_set_tip_states(pll_partition_t *p, unsigned int tip, const pll_state_t *map, const char *seq)
{
  pll_amd_ext_t *x = pll_ext(p);
  int rc;
  if (tip >= p->tips)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "tip index %u out of range", tip);
    return PLL_FAILURE;
  }
  if (pll_repeats_enabled(p) && !pll_update_repeats_tips(p, tip, map, seq)) return PLL_FAILURE;

  if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
  {
    if (!register_map(p, map)) return PLL_FAILURE;
    rc = encode_tipchars(p, tip, map, seq);
    if (rc && x)
    {
      x->tipchars_dirty[tip] = 1;
      x->tipmap_dirty = 1;
    }
  }
  else
  {
    rc = encode_tipclv(p, tip, map, seq);
    if (rc && x)
    {
      x->clv_side[tip] = SIDE_HOST; /* the dense indicator CLV lives in the host mirror either way */
      if (x->tip_compact[tip])
      {
        x->tipchars_dirty[tip] = 1;
        x->tipmap_dirty = 1;
      }
    }
  }
  return rc;
}

int pll_set_tip_clv(pll_partition_t *p, unsigned int tip, const double *clv, int padding)
{
  /* src/pll.c:1066-1129 */
  pll_amd_ext_t *x = pll_ext(p);
  unsigned int i, k;
  if (p->attributes & PLL_ATTRIB_PATTERN_TIP)
  {
    pll_set_error(PLL_ERROR_TIPDATA_ILLEGALFUNCTION, "Cannot use pll_set_tip_clv with PLL_ATTRIB_PATTERN_TIP.");
    return PLL_FAILURE;
  }
  const unsigned int in_stride = padding ? p->states_padded : p->states;
  const int rep = pll_repeats_enabled(p);
  const unsigned int n = rep ? p->repeats->pernode_ids[tip] : p->sites;
  double *dst = p->clv[tip];
  for (i = 0; i < n; ++i)
  {
    const unsigned int site = rep ? p->repeats->pernode_id_site[tip][i] : i;
    const double *src = clv + (size_t)site * in_stride;
    for (k = 0; k < p->rate_cats; ++k, dst += p->states_padded) memcpy(dst, src, p->states * sizeof(double));
  }
  if (p->asc_bias_alloc) asc_tip_entries(p, p->clv[tip], n);
  if (x)
  {
    x->clv_side[tip] = SIDE_HOST;
    /* An INDICATOR vector per site - every value exactly 0 or 1, at least one 1: what a caller gets from one-hot
     * encoding its sequences itself (SURVEY 8d's C5) - is a state mask in CLV form: the device then reads one-byte
     * codes and runs the tip kernels, as it does for pll_set_tip_states without PLL_ATTRIB_PATTERN_TIP. The dense CLV
     * stays in the host mirror for whoever reads or edits it (pll_tip_densify). Anything else: a dense CLV. */
    int compact = !x->no_tip_codes && !rep && p->states <= 8 * sizeof(pll_state_t);
    if (compact)
    {
      free(x->tipcodes[tip]);
      x->tipcodes[tip] = (unsigned char *)malloc((size_t)n + p->states);
      compact = x->tipcodes[tip] != NULL;
    }
    for (i = 0; compact && i < n; ++i)
    {
      const double *src = clv + (size_t)i * in_stride;
      pll_state_t m = 0;
      unsigned int j;
      for (j = 0; j < p->states; ++j)
      {
        if (src[j] == 1.0) m |= (pll_state_t)1 << j;
        else if (src[j] != 0.0) compact = 0;
      }
      if (!m) compact = 0;
      if (compact)
      {
        const int code = compact_code(p, x, m);
        if (code < 0) compact = 0;
        else x->tipcodes[tip][i] = (unsigned char)code;
      }
    }
    if (p->asc_bias_alloc)
      for (i = 0; compact && i < p->states; ++i)
      {
        const int code = compact_code(p, x, (pll_state_t)1 << i);
        if (code < 0) compact = 0;
        else x->tipcodes[tip][n + i] = (unsigned char)code;
      }
    x->tip_compact[tip] = (unsigned char)compact;
    x->fast_valid = 0;
    if (compact)
    {
      x->tipchars_dirty[tip] = 1;
      x->tipmap_dirty = 1;
    }
  }
  return PLL_SUCCESS;
}
