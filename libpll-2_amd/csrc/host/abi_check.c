/* abi_check.c - compile-time proof that the public structs are laid out like the reference's
 * (x86-64 LP64 offsets recorded in SURVEY.md section 8b against src/pll.h:241-335). The GPU box
 * has no reference header to compare with, so the record lives here as static assertions. */
#include <stddef.h>

#include "pll_internal.h"

#define AT(type, field, off) _Static_assert(offsetof(type, field) == (off), #type "." #field " moved")

_Static_assert(sizeof(pll_partition_t) == 232, "pll_partition_t size");
AT(pll_partition_t, tips, 0);
AT(pll_partition_t, clv_buffers, 4);
AT(pll_partition_t, nodes, 8);
AT(pll_partition_t, states, 12);
AT(pll_partition_t, sites, 16);
AT(pll_partition_t, pattern_weight_sum, 20);
AT(pll_partition_t, rate_matrices, 24);
AT(pll_partition_t, prob_matrices, 28);
AT(pll_partition_t, rate_cats, 32);
AT(pll_partition_t, scale_buffers, 36);
AT(pll_partition_t, attributes, 40);
AT(pll_partition_t, alignment, 48);
AT(pll_partition_t, states_padded, 56);
AT(pll_partition_t, clv, 64);
AT(pll_partition_t, pmatrix, 72);
AT(pll_partition_t, rates, 80);
AT(pll_partition_t, rate_weights, 88);
AT(pll_partition_t, subst_params, 96);
AT(pll_partition_t, scale_buffer, 104);
AT(pll_partition_t, frequencies, 112);
AT(pll_partition_t, prop_invar, 120);
AT(pll_partition_t, invariant, 128);
AT(pll_partition_t, pattern_weights, 136);
AT(pll_partition_t, eigen_decomp_valid, 144);
AT(pll_partition_t, eigenvecs, 152);
AT(pll_partition_t, inv_eigenvecs, 160);
AT(pll_partition_t, eigenvals, 168);
AT(pll_partition_t, maxstates, 176);
AT(pll_partition_t, tipchars, 184);
AT(pll_partition_t, charmap, 192);
AT(pll_partition_t, ttlookup, 200);
AT(pll_partition_t, tipmap, 208);
AT(pll_partition_t, asc_bias_alloc, 216);
AT(pll_partition_t, asc_additional_sites, 220);
AT(pll_partition_t, repeats, 224);

_Static_assert(sizeof(pll_repeats_t) == 104, "pll_repeats_t size");
AT(pll_repeats_t, pernode_site_id, 0);
AT(pll_repeats_t, pernode_id_site, 8);
AT(pll_repeats_t, pernode_ids, 16);
AT(pll_repeats_t, perscale_ids, 24);
AT(pll_repeats_t, pernode_allocated_clvs, 32);
AT(pll_repeats_t, enable_repeats, 40);
AT(pll_repeats_t, reallocate_repeats, 48);
AT(pll_repeats_t, lookup_buffer, 56);
AT(pll_repeats_t, toclean_buffer, 64);
AT(pll_repeats_t, id_site_buffer, 72);
AT(pll_repeats_t, bclv_buffer, 80);
AT(pll_repeats_t, lookup_buffer_size, 88);
AT(pll_repeats_t, charmap, 96);

_Static_assert(sizeof(pll_operation_t) == 32, "pll_operation_t size");
AT(pll_operation_t, parent_clv_index, 0);
AT(pll_operation_t, parent_scaler_index, 4);
AT(pll_operation_t, child1_clv_index, 8);
AT(pll_operation_t, child1_matrix_index, 12);
AT(pll_operation_t, child1_scaler_index, 16);
AT(pll_operation_t, child2_clv_index, 20);
AT(pll_operation_t, child2_matrix_index, 24);
AT(pll_operation_t, child2_scaler_index, 28);
_Static_assert(sizeof(pll_state_t) == 8, "pll_state_t size");

/* the extension block must start 8-byte aligned directly behind the public struct */
_Static_assert(sizeof(pll_partition_t) % 8 == 0, "extension block alignment");
