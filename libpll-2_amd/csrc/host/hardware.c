/* hardware.c - pll_hardware and its three functions (reference: src/hardware.c, src/pll.h:220-237,
 * :555, :2694-2698). The record describes the HOST CPU: callers consult it to choose a
 * PLL_ATTRIB_ARCH_* layout for the arrays they index themselves; nothing in this library computes with
 * those instruction sets. */
#include "pll_internal.h"

__thread pll_hardware_t pll_hardware;

int pll_hardware_probe(void)
{
  memset(&pll_hardware, 0, sizeof pll_hardware);
  pll_hardware.init = 1;
#if defined(__x86_64__) || defined(__i386__)
  __builtin_cpu_init();
  pll_hardware.mmx_present = __builtin_cpu_supports("mmx") != 0;
  pll_hardware.sse_present = __builtin_cpu_supports("sse") != 0;
  pll_hardware.sse2_present = __builtin_cpu_supports("sse2") != 0;
  pll_hardware.sse3_present = __builtin_cpu_supports("sse3") != 0;
  pll_hardware.ssse3_present = __builtin_cpu_supports("ssse3") != 0;
  pll_hardware.sse41_present = __builtin_cpu_supports("sse4.1") != 0;
  pll_hardware.sse42_present = __builtin_cpu_supports("sse4.2") != 0;
  pll_hardware.popcnt_present = __builtin_cpu_supports("popcnt") != 0;
  pll_hardware.avx_present = __builtin_cpu_supports("avx") != 0;
  pll_hardware.avx2_present = __builtin_cpu_supports("avx2") != 0;
#endif
  return PLL_SUCCESS;
}

void pll_hardware_dump(void)
{
  if (!pll_hardware.init) pll_hardware_probe();
  fprintf(stderr, "Host CPU features:%s%s%s%s%s%s%s%s%s%s%s\n", pll_hardware.altivec_present ? " altivec" : "",
          pll_hardware.mmx_present ? " mmx" : "", pll_hardware.sse_present ? " sse" : "", pll_hardware.sse2_present ? " sse2" : "",
          pll_hardware.sse3_present ? " sse3" : "", pll_hardware.ssse3_present ? " ssse3" : "",
          pll_hardware.sse41_present ? " sse4.1" : "", pll_hardware.sse42_present ? " sse4.2" : "",
          pll_hardware.popcnt_present ? " popcnt" : "", pll_hardware.avx_present ? " avx" : "", pll_hardware.avx2_present ? " avx2" : "");
}

void pll_hardware_ignore(void)
{
  pll_hardware.init = 1;
  pll_hardware.altivec_present = pll_hardware.mmx_present = pll_hardware.sse_present = pll_hardware.sse2_present = 1;
  pll_hardware.sse3_present = pll_hardware.ssse3_present = pll_hardware.sse41_present = pll_hardware.sse42_present = 1;
  pll_hardware.popcnt_present = pll_hardware.avx_present = pll_hardware.avx2_present = 1;
}
