/* synth_alignment.c - the synthetic alignment of SURVEY.md section 8d: benchmark and test INPUT
 * generation only (nothing here is on the product path; built as libpll_workload.so next to
 * libpll_amd.so so that bench.py and the tests get the same bytes on every host, at 100M+ draws).
 *
 * xorshift64 (x ^= x << 13; x ^= x >> 7; x ^= x << 17; seed 88172645463325252, output x >> 32).
 * Per site: an ancestral state u mod states; then every tip in turn copies it unless a draw
 * u mod 100 < mutate_pct replaces it by a uniform random state (one more draw, u mod states).
 */
#include <stddef.h>
#include <stdint.h>

static inline uint32_t next_u32(uint64_t *x)
{
  uint64_t v = *x;
  v ^= v << 13;
  v ^= v >> 7;
  v ^= v << 17;
  *x = v;
  return (uint32_t)(v >> 32);
}

/* out[tip * stride + site] = state of `tip` at `site` for sites [0, sites). *state is the generator
 * state: in = the seed (or what an earlier call left), out = the state after the last draw, so a long
 * alignment may be produced in column blocks. */
void pllwl_xorshift_alignment(unsigned tips, size_t sites, unsigned states, unsigned mutate_pct,
                              uint64_t *state, unsigned char *out, size_t stride)
{
  uint64_t x = *state;
  size_t n;
  unsigned t;
  for (n = 0; n < sites; ++n)
  {
    const unsigned char anc = (unsigned char)(next_u32(&x) % states);
    for (t = 0; t < tips; ++t)
    {
      unsigned char s = anc;
      if (next_u32(&x) % 100u < mutate_pct) s = (unsigned char)(next_u32(&x) % states);
      out[t * stride + n] = s;
    }
  }
  *state = x;
}
