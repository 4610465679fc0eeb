/* maps.c - character -> state-mask tables that callers pass to pll_set_tip_states.
 *
 * Library-owned data in the reference (src/pll.h:557-559, defined in src/maps.c:26-262); a
 * drop-in must export them with the same contents. Written from the IUPAC definitions
 * (SURVEY.md section 8b), not transcribed: nucleotides A=1 C=2 G=4 T/U=8 and the ambiguity
 * codes as unions, gap-like characters fully ambiguous; amino acids one bit each in the order
 * ARNDCQEGHILKMFPSTWYV with B=N|D, Z=Q|E, J=I|L; binary 0/1. Every other byte is 0 = illegal.
 */
#include "pll_internal.h"

#define BOTH(c, v) [c] = (v), [(c) + 32] = (v) /* upper and lower case letter */

const pll_state_t pll_map_bin[256] = {['0'] = 1, ['1'] = 2, ['-'] = 3, ['.'] = 3, ['?'] = 3};

const pll_state_t pll_map_nt[256] = {
    BOTH('A', 1),  BOTH('C', 2),  BOTH('G', 4),  BOTH('T', 8),  BOTH('U', 8),
    BOTH('M', 3),  /* A|C */
    BOTH('R', 5),  /* A|G */
    BOTH('S', 6),  /* C|G */
    BOTH('V', 7),  /* A|C|G */
    BOTH('W', 9),  /* A|T */
    BOTH('Y', 10), /* C|T */
    BOTH('H', 11), /* A|C|T */
    BOTH('K', 12), /* G|T */
    BOTH('D', 13), /* A|G|T */
    BOTH('B', 14), /* C|G|T */
    BOTH('N', 15), BOTH('O', 15), BOTH('X', 15),
    ['-'] = 15, ['.'] = 15, ['?'] = 15};

#define AA(i) (1ull << (i))
#define AA_ALL ((1ull << 20) - 1)
const pll_state_t pll_map_aa[256] = {
    BOTH('A', AA(0)),  BOTH('R', AA(1)),  BOTH('N', AA(2)),  BOTH('D', AA(3)),  BOTH('C', AA(4)),
    BOTH('Q', AA(5)),  BOTH('E', AA(6)),  BOTH('G', AA(7)),  BOTH('H', AA(8)),  BOTH('I', AA(9)),
    BOTH('L', AA(10)), BOTH('K', AA(11)), BOTH('M', AA(12)), BOTH('F', AA(13)), BOTH('P', AA(14)),
    BOTH('S', AA(15)), BOTH('T', AA(16)), BOTH('W', AA(17)), BOTH('Y', AA(18)), BOTH('V', AA(19)),
    BOTH('B', AA(2) | AA(3)), BOTH('Z', AA(5) | AA(6)), BOTH('J', AA(9) | AA(10)),
    BOTH('X', AA_ALL), ['*'] = AA_ALL, ['-'] = AA_ALL, ['.'] = AA_ALL, ['?'] = AA_ALL};

/* Diploid genotypes (src/pll.h:560-561): the IUPAC letter of an unphased call names the genotype.
 * 10 states: the homozygotes AA CC GG TT (bits 0-3, letters A C G T/U), then the heterozygotes in the
 * order AC AG AT CG CT GT (bits 4-9, letters M R W S Y K); N, O, X, '-' and '?' are fully ambiguous.
 * 16 states: ordered (phased) pairs - a heterozygote letter stands for both orders, bit b and bit b + 6. */
#define GT10_ALL ((1ull << 10) - 1)
const pll_state_t pll_map_gt10[256] = {
    BOTH('A', 1ull << 0), BOTH('C', 1ull << 1), BOTH('G', 1ull << 2), BOTH('T', 1ull << 3), BOTH('U', 1ull << 3),
    BOTH('M', 1ull << 4), BOTH('R', 1ull << 5), BOTH('W', 1ull << 6), BOTH('S', 1ull << 7), BOTH('Y', 1ull << 8), BOTH('K', 1ull << 9),
    BOTH('N', GT10_ALL), BOTH('O', GT10_ALL), BOTH('X', GT10_ALL), ['-'] = GT10_ALL, ['?'] = GT10_ALL};

#define GT16_HET(b) ((1ull << (b)) | (1ull << ((b) + 6)))
#define GT16_ALL ((1ull << 16) - 1)
const pll_state_t pll_map_gt16[256] = {
    BOTH('A', 1ull << 0), BOTH('C', 1ull << 1), BOTH('G', 1ull << 2), BOTH('T', 1ull << 3), BOTH('U', 1ull << 3),
    BOTH('M', GT16_HET(4)), BOTH('R', GT16_HET(5)), BOTH('W', GT16_HET(6)), BOTH('S', GT16_HET(7)), BOTH('Y', GT16_HET(8)),
    BOTH('K', GT16_HET(9)),
    BOTH('N', GT16_ALL), BOTH('O', GT16_ALL), BOTH('X', GT16_ALL), ['-'] = GT16_ALL, ['?'] = GT16_ALL};
