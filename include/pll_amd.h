/*
 * pll_amd.h - public C interface of the MI355X (gfx950) build of the libpll-2 partial-likelihood
 * hot path.
 *
 * This header is written from the ABI record in SURVEY.md section 8b, not transcribed from the
 * reference header. It declares ONLY what the hot path needs, with struct layouts that are
 * byte-identical to the reference (x86-64 LP64) so that a caller compiled against the
 * reference's own pll.h can link against libpll_amd.so unchanged. Every declaration cites the
 * reference declaration it replaces (paths relative to the reference checkout).
 *
 * The library computes on the GPU only. There is no CPU kernel behind these entry points: if no
 * gfx950 device (or the HIP code object) is available, pll_partition_create() fails with
 * pll_errno = PLL_ERROR_GPU_UNAVAILABLE instead of silently falling back.
 */
#ifndef PLL_AMD_H_
#define PLL_AMD_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status / limits (src/pll.h:82-110) --------------------------------------------------- */
#define PLL_FAILURE 0
#define PLL_SUCCESS 1
#define PLL_FALSE 0
#define PLL_TRUE 1

#define PLL_ALIGNMENT_CPU 8
#define PLL_ALIGNMENT_SSE 16
#define PLL_ALIGNMENT_AVX 32
#define PLL_ASCII_SIZE 256

/* 2^256 and its inverse, exact in binary64 (src/pll.h:96-97) */
#define PLL_SCALE_FACTOR 0x1p256
#define PLL_SCALE_THRESHOLD 0x1p-256
#define PLL_SCALE_BUFFER_NONE (-1)
#define PLL_SCALE_RATE_MAXDIFF 4 /* src/pll.h:104 */

/* ---- attribute word (src/pll.h:112-137) --------------------------------------------------- */
/* The ARCH bits only select the *host-visible layout* (states_padded, alignment) so existing
 * callers that pass ARCH_AVX2 keep seeing the padding they expect; arithmetic always runs on
 * the MI355X. */
#define PLL_ATTRIB_ARCH_CPU 0u
#define PLL_ATTRIB_ARCH_SSE (1u << 0)
#define PLL_ATTRIB_ARCH_AVX (1u << 1)
#define PLL_ATTRIB_ARCH_AVX2 (1u << 2)
#define PLL_ATTRIB_ARCH_AVX512 (1u << 3)
#define PLL_ATTRIB_ARCH_MASK 0xFu
#define PLL_ATTRIB_PATTERN_TIP (1u << 4)
#define PLL_ATTRIB_AB_LEWIS (1u << 5)
#define PLL_ATTRIB_AB_FELSENSTEIN (2u << 5)
#define PLL_ATTRIB_AB_STAMATAKIS (3u << 5)
#define PLL_ATTRIB_AB_MASK (7u << 5)
#define PLL_ATTRIB_AB_FLAG (1u << 8)
#define PLL_ATTRIB_RATE_SCALERS (1u << 9)
#define PLL_ATTRIB_SITE_REPEATS (1u << 10)
#define PLL_REPEATS_LOOKUP_SIZE 2000000u
#define PLL_ATTRIB_MASK ((1u << 11) - 1)

/* ---- error codes (subset of src/pll.h:154-190 that this path can raise) -------------------- */
#define PLL_ERROR_MEM_ALLOC 112
#define PLL_ERROR_PARAM_INVALID 113
#define PLL_ERROR_TIPDATA_ILLEGALSTATE 114
#define PLL_ERROR_TIPDATA_ILLEGALFUNCTION 115
#define PLL_ERROR_INVAR_INCOMPAT 117
#define PLL_ERROR_INVAR_PROPORTION 118
#define PLL_ERROR_INVAR_PARAMINDEX 119
#define PLL_ERROR_INVAR_NONEFOUND 120
#define PLL_ERROR_AB_INVALIDMETHOD 121
#define PLL_ERROR_AB_NOSUPPORT 122
#define PLL_ERROR_MSA_EMPTY 131
#define PLL_ERROR_MSA_MAP_INVALID 132
/* new, outside the reference's range: device problems are reported through the same
 * pll_errno / pll_errmsg convention */
#define PLL_ERROR_GPU_UNAVAILABLE 900
#define PLL_ERROR_GPU_RUNTIME 901
#define PLL_ERROR_GPU_UNSUPPORTED 902

#define PLL_GAMMA_RATES_MEAN 0
#define PLL_GAMMA_RATES_MEDIAN 1

/* ---- types ------------------------------------------------------------------------------- */
typedef unsigned long long pll_state_t; /* src/pll.h:217: one bit per state, <= 64 states */

struct pll_repeats;

/* src/pll.h:241-288; sizeof == 232, offsets asserted in csrc/host/abi_check.c */
typedef struct pll_partition
{
  unsigned int tips;
  unsigned int clv_buffers;
  unsigned int nodes;
  unsigned int states;
  unsigned int sites;
  unsigned int pattern_weight_sum;
  unsigned int rate_matrices;
  unsigned int prob_matrices;
  unsigned int rate_cats;
  unsigned int scale_buffers;
  unsigned int attributes;
  size_t alignment;
  unsigned int states_padded;
  double **clv;                  /* host mirror of the device CLVs, see pll_gpu_sync_* below */
  double **pmatrix;
  double *rates;
  double *rate_weights;
  double **subst_params;
  unsigned int **scale_buffer;   /* host mirror of the device scalers */
  double **frequencies;
  double *prop_invar;
  int *invariant;
  unsigned int *pattern_weights;
  int *eigen_decomp_valid;
  double **eigenvecs;
  double **inv_eigenvecs;
  double **eigenvals;
  unsigned int maxstates;
  unsigned char **tipchars;
  unsigned char *charmap;
  double *ttlookup;              /* kept NULL: the device kernels need no tip-tip table */
  pll_state_t *tipmap;
  int asc_bias_alloc;
  int asc_additional_sites;
  struct pll_repeats *repeats;
} pll_partition_t;

/* src/pll.h:290-321; sizeof == 104 */
typedef struct pll_repeats
{
  unsigned int **pernode_site_id;
  unsigned int **pernode_id_site;
  unsigned int *pernode_ids;
  unsigned int *perscale_ids;
  unsigned int *pernode_allocated_clvs;
  unsigned int (*enable_repeats)(struct pll_partition *partition, unsigned int left_clv,
                                 unsigned int right_clv);
  void (*reallocate_repeats)(struct pll_partition *partition, unsigned int parent,
                             int scaler_index, unsigned int sites_to_alloc);
  unsigned int *lookup_buffer;
  unsigned int *toclean_buffer;
  unsigned int *id_site_buffer;
  double *bclv_buffer;
  unsigned int lookup_buffer_size;
  char *charmap;
} pll_repeats_t;

/* src/pll.h:325-335; eight 4-byte fields, sizeof == 32 */
typedef struct pll_operation
{
  unsigned int parent_clv_index;
  int parent_scaler_index;
  unsigned int child1_clv_index;
  unsigned int child1_matrix_index;
  int child1_scaler_index;
  unsigned int child2_clv_index;
  unsigned int child2_matrix_index;
  int child2_scaler_index;
} pll_operation_t;

/* src/pll.h:347-354 */
typedef struct pll_msa_s
{
  int count;
  int length;
  char **sequence;
  char **label;
} pll_msa_t;

/* ---- printers used by the reference's examples and tests (src/pll.h:2590-2600, src/output.c) -- */
void pll_show_pmatrix(const pll_partition_t *partition, unsigned int index, unsigned int float_precision);
void pll_show_clv(const pll_partition_t *partition, unsigned int clv_index, int scaler_index, unsigned int float_precision);

/* ---- host feature record (src/pll.h:220-237, :555, :2694-2698; src/hardware.c) -------------- */
/* Callers test it (PLL_STAT(avx2_present), src/pll.h:77-78) before they ask for a PLL_ATTRIB_ARCH_* layout.
 * Here the bits describe the host CPU as the reference's probe does; they only ever select a host
 * LAYOUT (states_padded), the arithmetic runs on the device whatever they say. */
typedef struct pll_hardware_s
{
  int init;
  int altivec_present, mmx_present, sse_present, sse2_present, sse3_present, ssse3_present, sse41_present,
      sse42_present, popcnt_present, avx_present, avx2_present;
} pll_hardware_t;
extern __thread pll_hardware_t pll_hardware;
int pll_hardware_probe(void);   /* fills pll_hardware; returns PLL_SUCCESS */
void pll_hardware_dump(void);   /* prints the record */
void pll_hardware_ignore(void); /* marks every feature present */

/* ---- thread-local error state (src/pll.h:553-555, src/pll.c:24-25) ------------------------- */
extern __thread int pll_errno;
extern __thread char pll_errmsg[200];

/* ---- character maps callers hand to pll_set_tip_states (src/pll.h:557-560, src/maps.c) ----- */
extern const pll_state_t pll_map_bin[256];
extern const pll_state_t pll_map_nt[256];
extern const pll_state_t pll_map_gt10[256]; /* diploid genotypes, 10 unordered / 16 ordered states */
extern const pll_state_t pll_map_gt16[256];
extern const pll_state_t pll_map_aa[256];

/* ---- lifecycle (src/pll.h:638-648, src/pll.c:424-873) -------------------------------------- */
pll_partition_t *pll_partition_create(unsigned int tips, unsigned int clv_buffers,
                                      unsigned int states, unsigned int sites,
                                      unsigned int rate_matrices, unsigned int prob_matrices,
                                      unsigned int rate_cats, unsigned int scale_buffers,
                                      unsigned int attributes);
void pll_partition_destroy(pll_partition_t *partition);
void *pll_aligned_alloc(size_t size, size_t alignment); /* src/pll.h:778 */
void pll_aligned_free(void *ptr);                       /* src/pll.h:780 */

/* ---- site-pattern compression (src/pll.h:2499-2510, src/compress.c:171-410) ----------------- */
/* Merges identical alignment columns: the sequences are rewritten in place with the unique columns
 * (sorted lexicographically by encoded character, NUL-terminated), *length becomes their number,
 * the returned vector (malloc'ed, caller frees) holds their multiplicities; the _msa variant also
 * fills site_pattern_map[original site] = pattern index. The ordering and counting run on the
 * MI355X (csrc/hip/compress.hip); outputs are identical to the reference's. */
unsigned int *pll_compress_site_patterns(char **sequence, const pll_state_t *map, int count, int *length);
unsigned int *pll_compress_site_patterns_msa(pll_msa_t *msa, const pll_state_t *map,
                                             unsigned int *site_pattern_map);

/* ---- inputs (src/pll.h:650-661,746-758; src/pll.c:1026-1143; src/models.c:445-493) --------- */
int pll_set_tip_states(pll_partition_t *partition, unsigned int tip_index,
                       const pll_state_t *map, const char *sequence);
int pll_set_tip_clv(pll_partition_t *partition, unsigned int tip_index, const double *clv,
                    int padding);
void pll_set_pattern_weights(pll_partition_t *partition, const unsigned int *pattern_weights);
/* ascertainment-bias correction (src/pll.h:663-667, src/pll.c:1145-1200): the partition must have
 * been created with PLL_ATTRIB_AB_FLAG or an AB type; type = 0 | PLL_ATTRIB_AB_{LEWIS,FELSENSTEIN,
 * STAMATAKIS}. The correction enters pll_compute_{edge,root}_loglikelihood and
 * pll_compute_likelihood_derivatives (src/likelihood.c:24-120,191-268,342-440;
 * src/core_derivatives.c:851-924). Not combinable with PLL_ATTRIB_SITE_REPEATS (refused at creation). */
int pll_set_asc_bias_type(pll_partition_t *partition, int asc_bias_type);
void pll_set_asc_state_weights(pll_partition_t *partition, const unsigned int *state_weights);
void pll_set_frequencies(pll_partition_t *partition, unsigned int params_index,
                         const double *frequencies);
void pll_set_subst_params(pll_partition_t *partition, unsigned int params_index,
                          const double *params);
void pll_set_category_rates(pll_partition_t *partition, const double *rates);
void pll_set_category_weights(pll_partition_t *partition, const double *rate_weights);
int pll_update_invariant_sites_proportion(pll_partition_t *partition, unsigned int params_index,
                                          double prop_invar); /* src/models.c:495-540 */
int pll_update_invariant_sites(pll_partition_t *partition);  /* src/models.c:651-752 */
/* src/models.c:546-649: pattern-weighted number of invariant sites; state_inv_count[states] (or
 * NULL) receives the number of invariant patterns per state */
unsigned int pll_count_invariant_sites(pll_partition_t *partition, unsigned int *state_inv_count);

/* model side ("next" rows f2 of SURVEY section 8; host code, feeds the path) */
int pll_update_eigen(pll_partition_t *partition, unsigned int params_index); /* models.c:293 */
int pll_update_prob_matrices(pll_partition_t *partition, const unsigned int *params_indices,
                             const unsigned int *matrix_indices, const double *branch_lengths,
                             unsigned int count); /* src/models.c:412-443 */
int pll_compute_gamma_cats(double alpha, unsigned int categories, double *output_rates,
                           int rates_mode); /* src/gamma.c:220-292 */

/* ---- THE HOT PATH (src/pll.h:790-797,823-830) ---------------------------------------------- */
/* src/partials.c:237-291. Asynchronous: kernels are enqueued on the partition's HIP stream and
 * the call returns; results stay resident in HBM. */
void pll_update_partials(pll_partition_t *partition, const pll_operation_t *operations,
                         unsigned int count);
void pll_update_partials_rep(pll_partition_t *partition, const pll_operation_t *operations,
                             unsigned int count, unsigned int update_repeats);
/* src/likelihood.c:586-636. Synchronises the stream; returns -INFINITY (and sets pll_errno) on
 * a device error. persite_lnl may be NULL. */
double pll_compute_edge_loglikelihood(pll_partition_t *partition, unsigned int parent_clv_index,
                                      int parent_scaler_index, unsigned int child_clv_index,
                                      int child_scaler_index, unsigned int matrix_index,
                                      const unsigned int *freqs_indices, double *persite_lnl);
/* src/likelihood.c:122-189 ("next" row f3) */
double pll_compute_root_loglikelihood(pll_partition_t *partition, unsigned int clv_index,
                                      int scaler_index, const unsigned int *freqs_indices,
                                      double *persite_lnl);

/* ---- the flat core seam of the hot path (src/pll.h:1049-1177 and :1295-1414; bodies in
 * src/core_partials.c:48-1210, src/core_likelihood.c:24-1496) ----------------------------------
 * Same signatures as the reference: raw HOST arrays in the layout `attrib` describes (PLL_ATTRIB_ARCH_*
 * -> states_padded; PLL_ATTRIB_RATE_SCALERS -> [entry][rate] scalers). Every call wraps its arrays in a
 * partition of the call's shape and runs the device path (csrc/host/core_seam.c): complete, but priced at a
 * PCIe round trip per call - use the partition-level functions above for speed. Small shapes are kept per
 * thread, shape and device for the next call (at most 32 MB each, 128 MB together, least recently used out
 * first; PLL_AMD_SEAM_CACHE=0: none); pll_core_seam_release() gives the calling thread's back at once.
 * The lookup table of pll_core_create_lookup is private to its pair with pll_core_update_partial_tt,
 * as in the reference (there: products per pair of tip states; here: the two matrices). */
void pll_core_seam_release(void); /* (no counterpart in the reference, whose core functions hold no state) */
void pll_core_create_lookup(unsigned int states, unsigned int rate_cats, double *lookup, const double *left_matrix,
                            const double *right_matrix, const pll_state_t *tipmap, unsigned int tipmap_size, unsigned int attrib);
void pll_core_create_lookup_4x4(unsigned int rate_cats, double *lookup, const double *left_matrix, const double *right_matrix);
void pll_core_update_partial_tt(unsigned int states, unsigned int sites, unsigned int rate_cats, double *parent_clv,
                                unsigned int *parent_scaler, const unsigned char *left_tipchars, const unsigned char *right_tipchars,
                                const pll_state_t *tipmap, unsigned int tipmap_size, const double *lookup, unsigned int attrib);
void pll_core_update_partial_tt_4x4(unsigned int sites, unsigned int rate_cats, double *parent_clv, unsigned int *parent_scaler,
                                    const unsigned char *left_tipchars, const unsigned char *right_tipchars, const double *lookup,
                                    unsigned int attrib);
void pll_core_update_partial_ti(unsigned int states, unsigned int sites, unsigned int rate_cats, double *parent_clv,
                                unsigned int *parent_scaler, const unsigned char *left_tipchars, const double *right_clv,
                                const double *left_matrix, const double *right_matrix, const unsigned int *right_scaler,
                                const pll_state_t *tipmap, unsigned int tipmap_size, unsigned int attrib);
void pll_core_update_partial_ti_4x4(unsigned int sites, unsigned int rate_cats, double *parent_clv, unsigned int *parent_scaler,
                                    const unsigned char *left_tipchars, const double *right_clv, const double *left_matrix,
                                    const double *right_matrix, const unsigned int *right_scaler, unsigned int attrib);
void pll_core_update_partial_ii(unsigned int states, unsigned int sites, unsigned int rate_cats, double *parent_clv,
                                unsigned int *parent_scaler, const double *left_clv, const double *right_clv,
                                const double *left_matrix, const double *right_matrix, const unsigned int *left_scaler,
                                const unsigned int *right_scaler, unsigned int attrib);
void pll_core_update_partial_repeats(unsigned int states, unsigned int parent_sites, unsigned int left_sites, unsigned int right_sites,
                                     unsigned int rate_cats, double *parent_clv, unsigned int *parent_scaler, const double *left_clv,
                                     const double *right_clv, const double *left_matrix, const double *right_matrix,
                                     const unsigned int *left_scaler, const unsigned int *right_scaler,
                                     const unsigned int *parent_id_site, const unsigned int *left_site_id,
                                     const unsigned int *right_site_id, double *bclv_buffer, unsigned int attrib);
void pll_core_update_partial_repeats_generic(unsigned int states, unsigned int parent_sites, unsigned int left_sites,
                                             unsigned int right_sites, unsigned int rate_cats, double *parent_clv,
                                             unsigned int *parent_scaler, const double *left_clv, const double *right_clv,
                                             const double *left_matrix, const double *right_matrix, const unsigned int *left_scaler,
                                             const unsigned int *right_scaler, const unsigned int *parent_id_site,
                                             const unsigned int *left_site_id, const unsigned int *right_site_id,
                                             double *bclv_buffer, unsigned int attrib);
void pll_core_update_partial_repeatsbclv_generic(unsigned int states, unsigned int parent_sites, unsigned int left_sites,
                                                 unsigned int right_sites, unsigned int rate_cats, double *parent_clv,
                                                 unsigned int *parent_scaler, const double *left_clv, const double *right_clv,
                                                 const double *left_matrix, const double *right_matrix,
                                                 const unsigned int *left_scaler, const unsigned int *right_scaler,
                                                 const unsigned int *parent_id_site, const unsigned int *left_site_id,
                                                 const unsigned int *right_site_id, double *bclv_buffer, unsigned int attrib);
double pll_core_edge_loglikelihood_ii(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                      const unsigned int *parent_scaler, const double *child_clv, const unsigned int *child_scaler,
                                      const double *pmatrix, double *const *frequencies, const double *rate_weights,
                                      const unsigned int *pattern_weights, const double *invar_proportion, const int *invar_indices,
                                      const unsigned int *freqs_indices, double *persite_lnl, unsigned int attrib);
double pll_core_edge_loglikelihood_ti(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                      const unsigned int *parent_scaler, const unsigned char *tipchars, const pll_state_t *tipmap,
                                      unsigned int tipmap_size, const double *pmatrix, double *const *frequencies,
                                      const double *rate_weights, const unsigned int *pattern_weights, const double *invar_proportion,
                                      const int *invar_indices, const unsigned int *freqs_indices, double *persite_lnl,
                                      unsigned int attrib);
double pll_core_edge_loglikelihood_ti_4x4(unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                          const unsigned int *parent_scaler, const unsigned char *tipchars, const double *pmatrix,
                                          double *const *frequencies, const double *rate_weights, const unsigned int *pattern_weights,
                                          const double *invar_proportion, const int *invar_indices, const unsigned int *freqs_indices,
                                          double *persite_lnl, unsigned int attrib);
double pll_core_edge_loglikelihood_repeats(unsigned int states, unsigned int sites, const unsigned int child_sites, unsigned int rate_cats,
                                           const double *parent_clv, const unsigned int *parent_scaler, const double *child_clv,
                                           const unsigned int *child_scaler, const double *pmatrix, double **frequencies,
                                           const double *rate_weights, const unsigned int *pattern_weights, const double *invar_proportion,
                                           const int *invar_indices, const unsigned int *freqs_indices, double *persite_lnl,
                                           const unsigned int *parent_site_id, const unsigned int *child_site_id, double *bclv,
                                           unsigned int attrib);
double pll_core_edge_loglikelihood_repeats_generic(unsigned int states, unsigned int sites, const unsigned int child_sites,
                                                   unsigned int rate_cats, const double *parent_clv, const unsigned int *parent_scaler,
                                                   const double *child_clv, const unsigned int *child_scaler, const double *pmatrix,
                                                   double **frequencies, const double *rate_weights, const unsigned int *pattern_weights,
                                                   const double *invar_proportion, const int *invar_indices,
                                                   const unsigned int *freqs_indices, double *persite_lnl,
                                                   const unsigned int *parent_site_id, const unsigned int *child_site_id, double *bclv,
                                                   unsigned int attrib);
double pll_core_root_loglikelihood(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *clv,
                                   const unsigned int *scaler, double *const *frequencies, const double *rate_weights,
                                   const unsigned int *pattern_weights, const double *invar_proportion, const int *invar_indices,
                                   const unsigned int *freqs_indices, double *persite_lnl, unsigned int attrib);
double pll_core_root_loglikelihood_repeats(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *clv,
                                           const unsigned int *site_id, const unsigned int *scaler, double *const *frequencies,
                                           const double *rate_weights, const unsigned int *pattern_weights,
                                           const double *invar_proportion, const int *invar_indices, const unsigned int *freqs_indices,
                                           double *persite_lnl, unsigned int attrib);

/* flat forms of the derivative path and of the transition matrices (src/pll.h:1181-1273, :2400-2412;
 * bodies src/core_derivatives.c:26-118, :219-320, :324-470, :474-640, :695-930 and src/core_pmatrix.c:186-247).
 * Same seam as above: raw host arrays in, a throw-away partition, the device path, the result copied
 * back - `sumtable` here is a real table in the reference's layout, not a handle. The model arrives per
 * rate category (eigenvecs[k], freqs[k], prop_invar[k]); pll_core_update_pmatrix indexes its arrays
 * through params_indices and pmatrix[] through matrix_indices, as the reference does.
 * pll_core_likelihood_derivatives refuses PLL_ATTRIB_AB_* (the correction needs the partition's extra
 * entries: pll_compute_likelihood_derivatives serves it). */
int pll_core_update_sumtable_ii(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                const double *child_clv, const unsigned int *parent_scaler, const unsigned int *child_scaler,
                                double *const *eigenvecs, double *const *inv_eigenvecs, double *const *freqs, double *sumtable,
                                unsigned int attrib);
int pll_core_update_sumtable_ti(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                const unsigned char *left_tipchars, const unsigned int *parent_scaler, double *const *eigenvecs,
                                double *const *inv_eigenvecs, double *const *freqs, const pll_state_t *tipmap,
                                unsigned int tipmap_size, double *sumtable, unsigned int attrib);
/* src/pll.h:1217-1226 */
int pll_core_update_sumtable_ti_4x4(unsigned int sites, unsigned int rate_cats, const double *parent_clv,
                                    const unsigned char *left_tipchars, const unsigned int *parent_scaler,
                                    double *const *eigenvecs, double *const *inv_eigenvecs, double *const *freqs,
                                    double *sumtable, unsigned int attrib);
/* src/core_likelihood.c:211-223 (exported there, not declared in src/pll.h): unpadded layout */
double pll_core_root_loglikelihood_repeats_generic(unsigned int states, unsigned int sites, unsigned int rate_cats,
                                                   const double *clv, const unsigned int *site_id, const unsigned int *scaler,
                                                   double *const *frequencies, const double *rate_weights,
                                                   const unsigned int *pattern_weights, const double *invar_proportion,
                                                   const int *invar_indices, const unsigned int *freqs_indices, double *persite_lnl);
int pll_core_update_sumtable_repeats(unsigned int states, unsigned int sites, unsigned int parent_sites, unsigned int rate_cats,
                                     const double *clvp, const double *clvc, const unsigned int *parent_scaler,
                                     const unsigned int *child_scaler, double *const *eigenvecs, double *const *inv_eigenvecs,
                                     double *const *freqs, double *sumtable, const unsigned int *parent_site_id,
                                     const unsigned int *child_site_id, double *bclv_buffer, unsigned int inv, unsigned int attrib);
int pll_core_update_sumtable_repeats_generic(unsigned int states, unsigned int sites, unsigned int parent_sites,
                                             unsigned int rate_cats, const double *clvp, const double *clvc,
                                             const unsigned int *parent_scaler, const unsigned int *child_scaler,
                                             double *const *eigenvecs, double *const *inv_eigenvecs, double *const *freqs,
                                             double *sumtable, const unsigned int *parent_site_id,
                                             const unsigned int *child_site_id, double *bclv_buffer, unsigned int inv,
                                             unsigned int attrib);
int pll_core_likelihood_derivatives(unsigned int states, unsigned int sites, unsigned int rate_cats, const double *rate_weights,
                                    const unsigned int *parent_scaler, const unsigned int *child_scaler, unsigned int parent_ids,
                                    unsigned int child_ids, const int *invariant, const unsigned int *pattern_weights,
                                    double branch_length, const double *prop_invar, double *const *freqs, const double *rates,
                                    double *const *eigenvals, const double *sumtable, double *d_f, double *dd_f,
                                    unsigned int attrib);
int pll_core_update_pmatrix(double **pmatrix, unsigned int states, unsigned int rate_cats, const double *rates,
                            const double *branch_lengths, const unsigned int *matrix_indices, const unsigned int *params_indices,
                            const double *prop_invar, double *const *eigenvals, double *const *eigenvecs,
                            double *const *inv_eigenvecs, unsigned int count, unsigned int attrib);

/* ---- branch-length derivatives (src/pll.h:834-852, src/derivatives.c:239-418; SURVEY section 8
 * row f1). `sumtable` is the caller's buffer of sites*rate_cats*states_padded doubles, as in the
 * reference, but it is used as a HANDLE: pll_update_sumtable computes the table into HBM and
 * remembers which host buffer it stands for; pll_compute_likelihood_derivatives given the same
 * pointer streams the device copy. The host buffer itself is only filled by
 * pll_gpu_sync_sumtable() (or under PLL_AMD_EAGER_MIRROR=1); a table the library has never seen
 * (written by the caller) is uploaded from the host buffer. Up to 16 tables per partition stay
 * resident (pll_gpu_release_sumtable). */
int pll_update_sumtable(pll_partition_t *partition, unsigned int parent_clv_index,
                        unsigned int child_clv_index, int parent_scaler_index,
                        int child_scaler_index, const unsigned int *params_indices, double *sumtable);
int pll_compute_likelihood_derivatives(pll_partition_t *partition, int parent_scaler_index,
                                       int child_scaler_index, double branch_length,
                                       const unsigned int *params_indices, const double *sumtable,
                                       double *d_f, double *dd_f);

/* ---- site repeats bookkeeping (src/pll.h:682-742, src/repeats.c) --------------------------- */
/* scaler vector of a parent whose children are class-compressed (src/pll.h:727-742,
 * src/repeats.c:392-540): parent[i] = left[lids[site]] + right[rids[site]] with site = psites[i]; a NULL
 * map is the identity, a NULL scaler contributes 0. Integer utilities on host arrays, like
 * pll_fill_parent_scaler; the kernels fold this step into the update. */
void pll_fill_parent_scaler_repeats(unsigned int sites, unsigned int *parent_scaler, const unsigned int *psites,
                                    const unsigned int *left_scaler, const unsigned int *lids, const unsigned int *right_scaler,
                                    const unsigned int *rids);
void pll_fill_parent_scaler_repeats_per_rate(unsigned int sites, unsigned int rates, unsigned int *parent_scaler,
                                             const unsigned int *psites, const unsigned int *left_scaler, const unsigned int *lids,
                                             const unsigned int *right_scaler, const unsigned int *rids);
#define PLL_GET_ID(site_id, site) ((site_id) ? ((site_id)[(site)]) : (site))
#define PLL_GET_SITE(id_site, site) ((id_site) ? ((id_site)[(site)]) : (site))
int pll_repeats_enabled(const pll_partition_t *partition);
void pll_resize_repeats_lookup(pll_partition_t *partition, unsigned int size);
unsigned int pll_get_sites_number(const pll_partition_t *partition, unsigned int clv_index);
unsigned int *pll_get_site_id(const pll_partition_t *partition, unsigned int clv_index);
unsigned int *pll_get_id_site(const pll_partition_t *partition, unsigned int clv_index);
unsigned int pll_get_clv_size(const pll_partition_t *partition, unsigned int clv_index);
unsigned int pll_default_enable_repeats(pll_partition_t *partition, unsigned int left_clv,
                                        unsigned int right_clv);
unsigned int pll_no_enable_repeats(pll_partition_t *partition, unsigned int left_clv,
                                   unsigned int right_clv);
void pll_default_reallocate_repeats(pll_partition_t *partition, unsigned int parent,
                                    int scaler_index, unsigned int sites_to_alloc);
int pll_repeats_initialize(pll_partition_t *partition);
int pll_update_repeats_tips(pll_partition_t *partition, unsigned int tip_index,
                            const pll_state_t *map, const char *sequence);
void pll_update_repeats(pll_partition_t *partition, const pll_operation_t *op);
void pll_disable_bclv(pll_partition_t *partition);
void pll_fill_parent_scaler(unsigned int scaler_size, unsigned int *parent_scaler,
                            const unsigned int *left_scaler, const unsigned int *right_scaler);

/* ---- device-residency contract (new; SURVEY section 7 "hard parts" 1 and 2) ---------------- */
/* CLVs and scalers live in HBM; partition->clv[i] / scale_buffer[i] are a lazily refreshed host
 * mirror. Callers that read those arrays directly call one of these first. */
int pll_gpu_sync_clv(pll_partition_t *partition, unsigned int clv_index);     /* D2H one CLV */
int pll_gpu_sync_scaler(pll_partition_t *partition, unsigned int scaler_index);
/* transition matrices computed by pll_update_prob_matrices live on the device; this refreshes
 * partition->pmatrix[index] (index < 0: every matrix that is newer on the device) */
int pll_gpu_sync_pmatrix(pll_partition_t *partition, int index);
/* site-repeats class maps of inner nodes are computed on the device; pll_get_site_id() /
 * pll_get_id_site() refresh the host arrays they return, this call does it explicitly for callers
 * that read partition->repeats->pernode_* directly (node < 0: all nodes) */
int pll_gpu_sync_repeats(pll_partition_t *partition, int node);
int pll_gpu_sync_all(pll_partition_t *partition);
/* Callers that WRITE partition arrays directly (instead of through the setters above) tell the
 * library which device copies are stale. what = bitwise OR of PLL_GPU_DIRTY_*; index = array
 * slot or -1 for "all". */
#define PLL_GPU_DIRTY_PMATRIX 1u
#define PLL_GPU_DIRTY_FREQS 2u
#define PLL_GPU_DIRTY_RATE_WEIGHTS 4u
#define PLL_GPU_DIRTY_PATTERN_WEIGHTS 8u
#define PLL_GPU_DIRTY_INVARIANT 16u
#define PLL_GPU_DIRTY_CLV 32u    /* host copy of clv[index] is newer than the device copy */
#define PLL_GPU_DIRTY_SCALER 64u
#define PLL_GPU_DIRTY_TIPCHARS 128u
#define PLL_GPU_DIRTY_REPEATS 256u
#define PLL_GPU_DIRTY_EIGEN 512u /* eigenvecs / inv_eigenvecs / eigenvals / rates written directly */
/* Site repeats: pll_update_repeats / pll_update_partials keep, per node, what the class map standing on the device was
 * computed from (the two children, the versions of their maps, the lookup size) and launch nothing for an op whose
 * inputs have not moved - the maps are a function of the children's maps alone (src/repeats.c:299-382), so a
 * re-evaluation of the same tree after new branch lengths recomputes none and a topology move only the ancestors of
 * the moved edge. FORGET_REPEATS drops that knowledge for node `index` (-1: every node) without marking any host map
 * as newer: the next update computes the maps derived from it again (benchmarks of the recomputation itself; callers
 * that changed pernode_ids or a map behind the library's back use PLL_GPU_DIRTY_REPEATS, which implies it). */
#define PLL_GPU_FORGET_REPEATS 1024u
void pll_gpu_invalidate(pll_partition_t *partition, unsigned int what, int index);
/* download the device sumtable that stands for this host buffer into it (reference layout) */
int pll_gpu_sync_sumtable(pll_partition_t *partition, double *sumtable);
/* a partition keeps up to 16 device sumtables alive, one per host buffer handed to
 * pll_update_sumtable; beyond that the least recently used is recycled and an evaluation on ITS
 * handle fails with PLL_ERROR_GPU_RUNTIME (never a silent read of the unwritten host buffer). A
 * caller that is done with a table (about to free the host buffer) gives its HBM back here.
 * A recycled handle stays marked until pll_update_sumtable or this call names it again: a caller that frees an
 * evicted table and later fills a NEW buffer that malloc happened to place at the same address must announce it
 * (pll_gpu_release_sumtable(partition, buffer) before the first pll_compute_likelihood_derivatives on it) - otherwise
 * the evaluation fails with PLL_ERROR_GPU_RUNTIME instead of uploading the buffer. Resident tables are CLV-sized
 * (1M-site DNA: 128 MB each, up to 16 per partition): release what is no longer needed. */
int pll_gpu_release_sumtable(pll_partition_t *partition, const double *sumtable);
/* stream plumbing: by default each partition owns a stream; a harness may substitute its own
 * (a hipStream_t passed as void*) so that its events see the kernels. pll_update_partials is
 * asynchronous and may hold its last one or two operations back until the next call on the
 * partition (they are evaluated inside the edge log-likelihood kernel if that is the next call,
 * DESIGN.md "Tail fusion"); pll_gpu_synchronize(), pll_gpu_get_stream() and the timer calls launch
 * whatever is held, so a harness that brackets work with its own events calls one of them first. */
int pll_gpu_set_stream(pll_partition_t *partition, void *hip_stream);
void *pll_gpu_get_stream(const pll_partition_t *partition);
int pll_gpu_synchronize(pll_partition_t *partition);
/* Multi-GPU building block (SURVEY section 8 row e): pll_compute_edge_loglikelihood
 * (src/pll.h:790-797) without the host round trip. The evaluation is enqueued on the partition's
 * stream and leaves {lnL of this partition's sites, call sequence number} in the two doubles of
 * DEVICE memory at device_result; nothing is copied back and the call does not wait. A site-sharded
 * run hands device_result[0] of every rank to one RCCL all-reduce on the same stream
 * (pll_gpu_set_stream) and reads the sum once. Returns PLL_SUCCESS when enqueued. Not available
 * with an ascertainment-bias correction (its formula runs on the host). */
int pll_gpu_edge_loglikelihood_async(pll_partition_t *partition, unsigned int parent_clv_index,
                                     int parent_scaler_index, unsigned int child_clv_index,
                                     int child_scaler_index, unsigned int matrix_index,
                                     const unsigned int *freqs_indices, double *device_result);
/* ---- the ONE exchange of a site-sharded run (SURVEY section 8 row e) --------------------------
 * Sites are independent through every CLV update; the only cross-site operation of the path is the sum
 * of the per-site log-likelihoods (src/core_likelihood.c:1489, the sequential `logl += site_lk`). A run
 * that gives every GPU its own partition over a contiguous site range therefore needs exactly one
 * exchange per evaluation: the sum of one double per rank. Two forms, both plain C:
 *
 * (1) ranks of ONE node, fixed order - pll_gpu_group_*. The ranks (processes or threads) meet in a named
 *     POSIX shared-memory segment with two alternating slots per rank; every rank leaves {value, step}
 *     in its slot and adds the slots of all ranks IN RANK ORDER, so every rank returns the same bits and
 *     the sum is reproducible run to run whatever the arrival order (an all-reduce tree is not). Cost:
 *     a cache-line hand-off between host cores behind the result the device has already written to host
 *     memory - no kernel, no collective library. `name` must be unique per run and start with '/'.
 * (2) any communicator - pll_gpu_allreduce_lnl / pll_gpu_edge_loglikelihood_allreduce: one
 *     ncclAllReduce(sum, ncclDouble) on the partition's stream. librccl is opened with dlopen() at the
 *     first call (PLL_AMD_RCCL_LIB overrides the name), so the library loads and works without RCCL;
 *     the caller creates the ncclComm_t (ncclCommInitRank) and passes it as void *. */
typedef struct pll_gpu_group pll_gpu_group_t;
/* join (and, whoever comes first, create) the segment `name` as rank `rank` of `size`; waits until all
 * `size` ranks have joined (timeout_ms <= 0: 60 s). NULL + pll_errno on failure. */
pll_gpu_group_t *pll_gpu_group_join(const char *name, unsigned int rank, unsigned int size, int timeout_ms);
void pll_gpu_group_leave(pll_gpu_group_t *group);
unsigned int pll_gpu_group_rank(const pll_gpu_group_t *group);
unsigned int pll_gpu_group_size(const pll_gpu_group_t *group);
/* global[i] = local[i] of rank 0 + rank 1 + ... + rank size-1, in that order, i < count <= 6. Every
 * rank must call it the same number of times. PLL_FAILURE (pll_errno PLL_ERROR_GPU_RUNTIME) when a rank
 * does not arrive within the group's timeout. */
int pll_gpu_group_sum(pll_gpu_group_t *group, const double *local, unsigned int count, double *global);
/* pll_compute_edge_loglikelihood (src/pll.h:790-797) on this rank's partition followed by the exchange:
 * the log-likelihood of the WHOLE alignment on every rank (-inf on every rank if any rank failed).
 * persite_lnl, if given, receives this rank's sites only (per-site values stay sharded). */
double pll_gpu_group_edge_loglikelihood(pll_partition_t *partition, pll_gpu_group_t *group,
                                        unsigned int parent_clv_index, int parent_scaler_index,
                                        unsigned int child_clv_index, int child_scaler_index,
                                        unsigned int matrix_index, const unsigned int *freqs_indices,
                                        double *persite_lnl);
/* pll_compute_likelihood_derivatives (src/pll.h:2400-2412 region; src/derivatives.c:296-418) on this rank's partition
 * followed by the exchange of {d_f, dd_f}: the derivatives of the WHOLE alignment's log-likelihood on every rank,
 * the same bits everywhere (rank order), so that all ranks of a sharded branch-length optimisation take the same
 * Newton step. Every rank calls it with the same branch length. PLL_FAILURE on every rank if any rank failed
 * (that rank keeps its own pll_errno). group == NULL: the plain evaluation. */
int pll_gpu_group_likelihood_derivatives(pll_partition_t *partition, pll_gpu_group_t *group,
                                         int parent_scaler_index, int child_scaler_index, double branch_length,
                                         const unsigned int *params_indices, const double *sumtable,
                                         double *d_f, double *dd_f);
/* enqueue ncclAllReduce(device_values, device_values, count, ncclDouble, ncclSum, comm) on the
 * partition's stream (count doubles of DEVICE memory, e.g. what pll_gpu_edge_loglikelihood_async left) */
int pll_gpu_allreduce_lnl(pll_partition_t *partition, void *nccl_comm, double *device_values, unsigned int count);
/* set-up of a (partition, communicator) pair, once: binds librccl, asks ncclCommCount, reserves the 16-byte operand
 * in device memory. Everything that can fail before a collective is enqueued fails HERE - call it on every rank
 * after creating the communicator and agree on the results before the first collective evaluation
 * (pll_gpu_edge_loglikelihood_allreduce calls it itself when it meets a new communicator, but a rank that fails
 * there has not joined the collective its peers are in). PLL_SUCCESS / PLL_FAILURE + pll_errno. */
int pll_gpu_allreduce_prepare(pll_partition_t *partition, void *nccl_comm);
/* the whole step without a host round trip before the exchange: the shard's log-likelihood stays in
 * device memory, is all-reduced there and only the sum comes back. Collective: every rank of the
 * communicator calls it, the same number of times. Returns the sum, -inf on failure. Once the pair is
 * prepared, a rank whose own evaluation fails still takes part (its operand is -inf): every rank returns
 * -inf - that rank with its own pll_errno, the others with PLL_ERROR_GPU_RUNTIME - and nobody is left
 * waiting inside the collective for a rank that merely failed to evaluate. A rank that never calls (it died,
 * or returned from a failed set-up) cannot be helped by the callers that did: their all-reduce never completes;
 * they get -inf + PLL_ERROR_GPU_RUNTIME after PLL_AMD_REDUCE_TIMEOUT_MS (default 60 000) instead of blocking
 * for ever, with the partition's stream still stuck behind the collective - fatal for the job, but reported. */
double pll_gpu_edge_loglikelihood_allreduce(pll_partition_t *partition, void *nccl_comm,
                                            unsigned int parent_clv_index, int parent_scaler_index,
                                            unsigned int child_clv_index, int child_scaler_index,
                                            unsigned int matrix_index, const unsigned int *freqs_indices);
/* 1 if a RCCL library could be opened (pll_gpu_allreduce_* usable), 0 otherwise */
int pll_gpu_rccl_available(void);
/* HIP-event stopwatch on the partition's stream (bench.py's roofline leg): start .. stop
 * brackets whatever was enqueued in between; returns elapsed milliseconds from stop(). */
int pll_gpu_timer_start(pll_partition_t *partition);
double pll_gpu_timer_stop(pll_partition_t *partition);
/* number of kernel launches issued by the last pll_update_partials call (bench bookkeeping) */
unsigned int pll_gpu_last_launch_count(const pll_partition_t *partition);
/* site repeats: class-map operations computed on the device (launches = 0) / class kernels launched (launches != 0)
 * since the partition was created. An unchanged tree adds nothing, a topology move the ops of its partial traversal */
unsigned long long pll_gpu_class_map_work(const pll_partition_t *partition, int launches);
/* pll_update_partials calls whose launches came from a plan the partition's device context had kept, since it was
 * created. A context's plans depend on ITS device blocks only: other partitions coming, growing and going leave them */
unsigned long long pll_gpu_plan_replays(const pll_partition_t *partition);
/* 1 if the last pll_update_partials call found its operation list, and everything its classification rests on, as the
 * call before left them and went straight to the launches (a re-evaluation of one tree); 0 if it took the whole path */
int pll_gpu_last_update_replayed(const pll_partition_t *partition);
/* HBM bytes the kernels of the last pll_update_partials call had to move by construction: child
 * reads + parent and scaler writes of every launch AS IT WAS GROUPED (an op evaluated together with
 * the producers of its children does not read those children back) - bench.py's roofline numerator */
double pll_gpu_last_algorithmic_bytes(const pll_partition_t *partition);
int pll_gpu_device_count(void);
/* 1 if a usable gfx950 device is present, 0 otherwise (the analogue of src/hardware.c's probe) */
int pll_gpu_available(void);

#ifdef __cplusplus
}
#endif
#endif /* PLL_AMD_H_ */
