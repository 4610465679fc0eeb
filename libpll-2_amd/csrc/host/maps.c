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
