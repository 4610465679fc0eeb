/* compress.c - pll_compress_site_patterns / pll_compress_site_patterns_msa (SURVEY.md section 8
 * row f4; reference: src/compress.c:171-410).
 *
 * Host side: argument checks and error codes, the character recoding (states out of the byte range
 * are renumbered 1, 2, ... - :101-126; decoding prefers '-' for a gap and the lowest ASCII code
 * otherwise - :228-237), moving the sequences to and from the caller's strings. The work itself -
 * ordering the alignment columns, merging equal ones, counting - runs on the device
 * (csrc/hip/compress.hip); the outputs are those of the reference's sort: unique columns in
 * lexicographic order of the encoded (signed) characters, weights, site -> pattern map.
 */
#include "pll_internal.h"

int pllgpu_compress_patterns(const unsigned char *encoded, unsigned count, unsigned length, unsigned char *compressed,
                             unsigned *weights, unsigned *site_pattern_map, unsigned *patterns_out, int device);
const char *pllgpu_compress_last_error(void);

static unsigned int *compress(char **sequence, const pll_state_t *map, int count, int *length, unsigned int *site_pattern_map)
{
  unsigned char charmap[PLL_ASCII_SIZE], inv_charmap[PLL_ASCII_SIZE];
  int i, j;
  if (!count)
  {
    pll_set_error(PLL_ERROR_MSA_EMPTY, "Number of sequences must be greater than 0.");
    return NULL;
  }
  if (!map)
  {
    pll_set_error(PLL_ERROR_MSA_MAP_INVALID, "Map is undefined.");
    return NULL;
  }
  if (map[0])
  {
    pll_set_error(PLL_ERROR_MSA_MAP_INVALID, "'0' cannot be used as a state.");
    return NULL;
  }
  /* recode: states that do not fit a byte are renumbered in order of first appearance */
  pll_state_t maxv = 0;
  for (i = 0; i < PLL_ASCII_SIZE; ++i)
    if (map[i] > maxv) maxv = map[i];
  if (maxv >= PLL_ASCII_SIZE)
  {
    unsigned char k = 1;
    memset(charmap, 0, sizeof charmap);
    for (i = 0; i < PLL_ASCII_SIZE; ++i)
    {
      if (!map[i] || charmap[i]) continue;
      for (j = i; j < PLL_ASCII_SIZE; ++j)
        if (map[j] == map[i]) charmap[j] = k;
      ++k;
    }
  }
  else
    for (i = 0; i < PLL_ASCII_SIZE; ++i) charmap[i] = (unsigned char)map[i];
  memset(inv_charmap, 0, sizeof inv_charmap);
  for (i = 0; i < PLL_ASCII_SIZE; ++i)
    if (map[i] && (!inv_charmap[charmap[i]] || i == '-')) inv_charmap[charmap[i]] = (unsigned char)i;

  const int len = *length;
  if (len <= 0)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_compress_site_patterns: empty alignment");
    return NULL;
  }
  /* encode in place, like the reference (a failure leaves the sequences partly recoded there too) */
  pll_errno = 0;
  for (i = 0; i < count; ++i)
    for (j = 0; j < len; ++j)
    {
      const unsigned char c = charmap[(unsigned char)sequence[i][j]];
      if (!c)
      {
        pll_set_error(PLL_ERROR_TIPDATA_ILLEGALSTATE, "Cannot encode character %c at sequence %d position %d.", sequence[i][j],
                      i + 1, j + 1);
        return NULL;
      }
      sequence[i][j] = (char)c;
    }

  const size_t cells = (size_t)(unsigned int)count * (size_t)(unsigned int)len;
  unsigned char *flat = NULL, *comp = NULL;
  unsigned int *weight = NULL, *result = NULL;
  unsigned int patterns = 0;
  flat = (unsigned char *)calloc(cells ? cells : 1, 1);
  comp = (unsigned char *)malloc(cells);
  weight = (unsigned int *)malloc((size_t)(unsigned int)len * sizeof(unsigned int));
  if (!flat || !comp || !weight)
  {
    pll_set_error(PLL_ERROR_MEM_ALLOC, "Cannot allocate space for matrix data.");
    goto out;
  }
  for (i = 0; i < count; ++i) memcpy(flat + (size_t)i * len, sequence[i], (size_t)len);
  {
    const int rc = pllgpu_compress_patterns(flat, (unsigned)count, (unsigned)len, comp, weight, site_pattern_map, &patterns, -1);
    if (rc != 0)
    {
      /* no CPU fallback: without a device the call fails (the sequences stay recoded, as after any
       * other failure of this function) */
      pll_set_error(rc == -1 ? PLL_ERROR_GPU_UNAVAILABLE : PLL_ERROR_GPU_RUNTIME, "pll_compress_site_patterns: %s",
                    pllgpu_compress_last_error());
      fprintf(stderr, "libpll_amd: %s\n", pll_errmsg);
      goto out;
    }
  }
  for (i = 0; i < count; ++i)
  {
    for (j = 0; j < (int)patterns; ++j) sequence[i][j] = (char)inv_charmap[comp[(size_t)i * patterns + j]];
    sequence[i][patterns] = 0;
  }
  result = (unsigned int *)malloc((size_t)patterns * sizeof(unsigned int));
  if (result)
    memcpy(result, weight, (size_t)patterns * sizeof(unsigned int));
  else
  {
    result = weight; /* the reference keeps the over-long vector in that case */
    weight = NULL;
  }
  *length = (int)patterns;
out:
  free(flat);
  free(comp);
  free(weight);
  return result;
}

unsigned int *pll_compress_site_patterns(char **sequence, const pll_state_t *map, int count, int *length)
{
  return compress(sequence, map, count, length, NULL);
}

unsigned int *pll_compress_site_patterns_msa(pll_msa_t *msa, const pll_state_t *map, unsigned int *site_pattern_map)
{
  return compress(msa->sequence, map, msa->count, &msa->length, site_pattern_map);
}
