/*
 * pll_amd_device.h - the thin C ABI between the C host code (libpll-2_amd/csrc/host) and the
 * HIP translation unit (libpll-2_amd/csrc/hip/pllgpu.hip) that owns the gfx950 kernels.
 *
 * Plain C: opaque context handle, indices, pointers and sizes only. The host side keeps every
 * libpll-2 semantic (pll_partition_t bookkeeping, op classification, dependency levels, site
 * repeats, error convention); this layer only moves bytes and launches kernels. Each entry point
 * names the reference routine whose arithmetic it replaces.
 *
 * All functions return 0 on success or a negative pllgpu_status; pllgpu_last_error() gives text.
 */
#ifndef PLL_AMD_DEVICE_H_
#define PLL_AMD_DEVICE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pllgpu_ctx pllgpu_ctx_t;

enum pllgpu_status
{
  PLLGPU_OK = 0,
  PLLGPU_ENODEVICE = -1, /* no gfx950 device / HIP runtime unusable */
  PLLGPU_ENOMEM = -2,
  PLLGPU_ERUNTIME = -3,  /* a HIP call failed */
  PLLGPU_EINVAL = -4,
  PLLGPU_EUNSUPPORTED = -5
};

/* shape of one partition (immutable for the life of the context) */
typedef struct pllgpu_geometry
{
  unsigned int tips;
  unsigned int nodes;          /* tips + clv_buffers */
  unsigned int states;
  unsigned int states_padded;  /* host stride, kept on the device: mirror copies are memcpy */
  unsigned int rate_cats;
  unsigned int sites;          /* alignment sites */
  unsigned int sites_alloc;    /* sites + asc-bias extra sites */
  unsigned int prob_matrices;
  unsigned int rate_matrices;
  unsigned int scale_buffers;
  unsigned int per_rate_scalers; /* PLL_ATTRIB_RATE_SCALERS */
  unsigned int pattern_tip;      /* PLL_ATTRIB_PATTERN_TIP */
} pllgpu_geometry_t;

/* one CLV update after host-side classification (src/partials.c:24-235) */
#define PLLGPU_OP_LEFT_TIP 1u   /* left child is given by tip codes, not a CLV */
#define PLLGPU_OP_RIGHT_TIP 2u
#define PLLGPU_OP_GATHER 4u     /* at least one of the three nodes is class-compressed */
typedef struct pllgpu_op
{
  unsigned int parent_clv, left_clv, right_clv;       /* node indices */
  int parent_scaler, left_scaler, right_scaler;       /* scale buffer indices or -1 */
  unsigned int left_matrix, right_matrix;
  unsigned int parent_entries;                        /* sites, or class count under repeats */
  unsigned int flags;
  unsigned int level;                                 /* dependency level, 0-based */
  int war_level;                                      /* latest level at which an EARLIER op of the list still
                                                         reads or writes this op's parent CLV / scaler
                                                         (-1: none): the op may not run before it */
} pllgpu_op_t;

typedef struct pllgpu_edge
{
  unsigned int parent_clv, child_clv; /* parent is always a CLV node; child may be a tip */
  int parent_scaler, child_scaler;
  unsigned int matrix;
  unsigned int child_is_tip;
  unsigned int gather;                /* either end class-compressed */
  const unsigned int *freqs_indices;  /* host, [rate_cats] */
  int want_persite;
  double *device_result;              /* NULL: synchronous call. Otherwise 2 doubles of DEVICE memory:
                                         the kernel leaves {lnL, call sequence number} there and the
                                         call returns without waiting (multi-GPU: the sum over shards
                                         is reduced on the device, pll_gpu_edge_loglikelihood_async) */
  double sequence;                    /* 0: the context numbers the call itself. Otherwise the sequence word the
                                         kernel leaves next to the value (ranks of a collective number their
                                         evaluations in step, so that the reduced word identifies the step) */
} pllgpu_edge_t;

int pllgpu_device_count(void);
/* the device a context created now with device = -1 would live on (PLL_AMD_DEVICE=<n>, else the calling thread's
 * current HIP device); -2: any of them in turn (PLL_AMD_DEVICE=auto); and the device a context does live on */
/* site repeats, since the context was created: class-map ops handed to the device (launches = 0) or class kernels
 * launched (launches = 1) - what the version stamps of repeats.c save is visible here */
unsigned long long pllgpu_class_map_work(const pllgpu_ctx_t *ctx, int launches);
/* op lists launched from a plan the context had kept (the same list over blocks that have not moved), ever */
unsigned long long pllgpu_plan_replays(const pllgpu_ctx_t *ctx);
int pllgpu_default_device(void);
int pllgpu_context_device(const pllgpu_ctx_t *ctx);
const char *pllgpu_last_error(void);

pllgpu_ctx_t *pllgpu_create(const pllgpu_geometry_t *geo, int device);
void pllgpu_destroy(pllgpu_ctx_t *ctx);

/* ---- data movement ------------------------------------------------------------------------ */
/* entries = number of site entries (each entry is rate_cats*states_padded doubles) */
int pllgpu_clv_reserve(pllgpu_ctx_t *ctx, unsigned int node, unsigned int entries);
int pllgpu_clv_upload(pllgpu_ctx_t *ctx, unsigned int node, const double *host, unsigned int entries);
int pllgpu_clv_download(pllgpu_ctx_t *ctx, unsigned int node, double *host, unsigned int entries);
int pllgpu_scaler_reserve(pllgpu_ctx_t *ctx, unsigned int index, unsigned int entries);
int pllgpu_scaler_upload(pllgpu_ctx_t *ctx, unsigned int index, const unsigned int *host,
                         unsigned int entries);
int pllgpu_scaler_download(pllgpu_ctx_t *ctx, unsigned int index, unsigned int *host,
                           unsigned int entries);
/* on = 1: CLV and scaler downloads of up to 2 MB are enqueued and return before their bytes have arrived; on = 0: one
 * wait for all of them, after which every `host` buffer handed over since holds its data (a caller that wants a CLV and
 * its scaler vector pays one wait instead of two). Larger downloads wait as always. */
int pllgpu_download_defer(pllgpu_ctx_t *ctx, int on);
int pllgpu_tipchars_upload(pllgpu_ctx_t *ctx, unsigned int tip, const unsigned char *host,
                           unsigned int count);
/* code -> state mask table; NULL selects "code is the mask" (4-state, src/pll.c:875-910) */
int pllgpu_tipmap_upload(pllgpu_ctx_t *ctx, const unsigned long long *host, unsigned int count);
/* host_block: count matrices in the reference layout [rate][row][states_padded], contiguous
 * (src/pll.c:593-611); re-laid-out for the kernels during the upload */
int pllgpu_pmatrix_upload(pllgpu_ctx_t *ctx, unsigned int first, unsigned int count,
                          const double *host_block);
int pllgpu_frequencies_upload(pllgpu_ctx_t *ctx, unsigned int index, const double *host);
int pllgpu_rate_weights_upload(pllgpu_ctx_t *ctx, const double *host);
int pllgpu_prop_invar_upload(pllgpu_ctx_t *ctx, const double *host);
int pllgpu_pattern_weights_upload(pllgpu_ctx_t *ctx, const unsigned int *host, unsigned int count);
int pllgpu_invariant_upload(pllgpu_ctx_t *ctx, const int *host /* or NULL */, unsigned int count);
/* site repeats maps of one node (src/pll.h:292-297): site_id[sites_alloc] site -> class,
 * id_site[ids] class -> representative site; ids == 0 clears (node not compressed) */
int pllgpu_repeats_upload(pllgpu_ctx_t *ctx, unsigned int node, const unsigned int *site_id,
                          const unsigned int *id_site, unsigned int ids);

/* ---- compute ------------------------------------------------------------------------------ */
/* class maps computed ON the device (replaces the table walk of
 * pll_update_repeats, src/repeats.c:299-382, and - for ops with force == 0 - the decision of
 * pll_default_enable_repeats, src/repeats.c:100-110): for every op the parent's site -> class and
 * class -> first-site maps from the children's maps. The ops of a whole traversal go in ONE call,
 * sorted by dependency level (ops of one level are mutually independent): a child produced by an
 * earlier op of the call is named by that op's index (lsrc / rsrc), its class count is taken from
 * device memory; a child from outside the call (lsrc / rsrc = -1) must have its maps on the device,
 * nleft / nright are its class counts (0: not compressed). force = 1: the host has decided to
 * compress this parent (a caller-supplied enable_repeats callback), both children from outside.
 * lookup_size = pll_repeats_t::lookup_buffer_size. counts_out[i] = PLLGPU_REPEATS_COMPRESSED |
 * classes of ops[i].parent, or 0 where the rule said no. Synchronises once, at the end. */
#define PLLGPU_REPEATS_COMPRESSED 0x80000000u
#define PLLGPU_REPEATS_MAX_OPS 65536u
typedef struct pllgpu_repop
{
  unsigned int parent, left, right;
  unsigned int nleft, nright;
  int lsrc, rsrc;
  unsigned int level;
  unsigned int force;
} pllgpu_repop_t;
int pllgpu_repeats_classes(pllgpu_ctx_t *ctx, const pllgpu_repop_t *ops, unsigned int count,
                           unsigned int lookup_size, unsigned int *counts_out);
/* how many classes the kernels shall assume for `node` (0 = one entry per site, maps unused) */
int pllgpu_repeats_set_ids(pllgpu_ctx_t *ctx, unsigned int node, unsigned int ids);
/* device maps of `node` back to the host: site_id[sites], id_site[ids]. Synchronises. */
int pllgpu_repeats_download(pllgpu_ctx_t *ctx, unsigned int node, unsigned int *site_id,
                            unsigned int *id_site, unsigned int ids);

/* replaces pll_core_update_partial_{ii,ti,tt,repeats} + pll_core_create_lookup
 * (src/core_partials.c:48-1210). Asynchronous on the context's stream. ops must be sorted by
 * level; ops of one level are independent. */
int pllgpu_update_partials(pllgpu_ctx_t *ctx, const pllgpu_op_t *ops, unsigned int count);
/* replaces pll_core_edge_loglikelihood_{ii,ti,ti_4x4,repeats} (src/core_likelihood.c:351-1496).
 * Synchronises. persite_host may be NULL. */
int pllgpu_edge_loglikelihood(pllgpu_ctx_t *ctx, const pllgpu_edge_t *edge, double *persite_host,
                              double *lnl_out);
/* replaces pll_core_root_loglikelihood[_repeats] (src/core_likelihood.c:25-349) */
int pllgpu_root_loglikelihood(pllgpu_ctx_t *ctx, unsigned int clv, int scaler, unsigned int gather,
                              const unsigned int *freqs_indices, double *persite_host,
                              double *lnl_out);

/* ---- transition matrices on the device ---------------------------- */
/* eigensystem of rate matrix `index` in the reference's layouts: eigenvecs[j*sp+i],
 * inv_eigenvecs[i*sp+j] ([states][states_padded] each), eigenvals[states_padded] */
int pllgpu_eigen_upload(pllgpu_ctx_t *ctx, unsigned int index, const double *eigenvecs,
                        const double *inv_eigenvecs, const double *eigenvals);
/* replaces pll_core_update_pmatrix (src/core_pmatrix.c:24-258): P = I + Vinv' diag(expm1(lambda r t /
 * (1 - pinv))) V per (matrix, rate category), identity for t = 0, written straight into the
 * device's transposed layout. Needs the eigensystems, category rates (pllgpu_rates_upload) and
 * prop_invar on the device. Asynchronous. */
int pllgpu_update_pmatrices(pllgpu_ctx_t *ctx, const unsigned int *params_indices /* [rate_cats] */,
                            const unsigned int *matrix_indices, const double *branch_lengths,
                            unsigned int count);
/* device matrix `index` back in the reference's host layout [rate][row][states_padded]. Synchronises. */
int pllgpu_pmatrix_download(pllgpu_ctx_t *ctx, unsigned int index, double *host);

/* ascertainment-bias correction (src/likelihood.c:50-120, :191-268,
 * :342-440): for each state n the likelihood of the per-state extra entry `sites + n` of the same
 * edge (is_root = 0; edge->child_is_tip honoured) or root (is_root = 1; only parent_clv /
 * parent_scaler / freqs_indices of `edge` are read) and the number of scalings it carries.
 * With per-rate scalers the term is already brought to the smallest per-rate count, which is the
 * count reported. Synchronises. */
int pllgpu_asc_terms(pllgpu_ctx_t *ctx, const pllgpu_edge_t *edge, int is_root, double *terms /* [states] */,
                     unsigned int *scalings /* [states] */);

/* ---- branch-length derivatives ---------------------------------- */
/* the two contraction matrices of the sumtable, one block [rate][row j][states_padded] each:
 * slot 0: M1[j][i] = pi_i * inv_eigenvecs[i][j] (applied to the left end), slot 1: M2[j][i] =
 * eigenvecs[j][i] (right end) - src/core_derivatives.c:446-456 */
int pllgpu_aux_matrix_upload(pllgpu_ctx_t *ctx, unsigned int slot, const double *host_block);
int pllgpu_eigenvals_upload(pllgpu_ctx_t *ctx, unsigned int index, const double *host);
int pllgpu_rates_upload(pllgpu_ctx_t *ctx, const double *host); /* category rates [rate_cats] */
typedef struct pllgpu_sumtable
{
  unsigned int left_clv, right_clv; /* a tip given by codes must be the left end */
  int left_scaler, right_scaler;
  unsigned int left_is_tip;
  unsigned int gather;
} pllgpu_sumtable_t;
/* replaces pll_core_update_sumtable_{ii,ti,repeats} (src/core_derivatives.c:25-641); the table
 * stays in HBM in one of PLLGPU_SUMTABLE_SLOTS slots (allocated on first use) */
#define PLLGPU_SUMTABLE_SLOTS 16
int pllgpu_update_sumtable(pllgpu_ctx_t *ctx, const pllgpu_sumtable_t *st, unsigned int slot);
int pllgpu_sumtable_upload(pllgpu_ctx_t *ctx, unsigned int slot, const double *host);
int pllgpu_sumtable_download(pllgpu_ctx_t *ctx, unsigned int slot, double *host);
/* give the slot's HBM back (pll_gpu_release_sumtable) */
int pllgpu_sumtable_release(pllgpu_ctx_t *ctx, unsigned int slot);
/* replaces pll_core_likelihood_derivatives (src/core_derivatives.c:696-849). Synchronises. */
/* eval_sites: how many leading table entries enter the sums (sites, or sites + states for the
 * Stamatakis correction, src/core_derivatives.c:733-742) */
int pllgpu_likelihood_derivatives(pllgpu_ctx_t *ctx, unsigned int slot, double branch_length,
                                  const unsigned int *params_indices, unsigned int eval_sites, double *d_f,
                                  double *dd_f);
/* (L, L', L'') of the per-state extra entries at the branch length of the LAST
 * pllgpu_likelihood_derivatives call on this context, and their scaling counts
 * (src/core_derivatives.c:864-891). Synchronises. */
int pllgpu_asc_derivative_terms(pllgpu_ctx_t *ctx, unsigned int slot, int parent_scaler, int child_scaler,
                                const unsigned int *params_indices, double *lk /* [states][3] */,
                                unsigned int *scalings /* [states] */);

/* ---- site-pattern compression (src/compress.c:171-410) ------------- */
/* encoded: [count][length] bytes, one row per sequence (characters already recoded). Outputs:
 * compressed [count][*patterns_out] (row stride = *patterns_out) = the unique columns in
 * lexicographic order of the characters taken as signed char, weights[*patterns_out] their
 * multiplicities, site_pattern_map[length] (or NULL) the pattern of every original site. No
 * context: the call owns a stream on `device` (-1: PLL_AMD_DEVICE or 0). Synchronous. */
int pllgpu_compress_patterns(const unsigned char *encoded, unsigned int count, unsigned int length,
                             unsigned char *compressed, unsigned int *weights,
                             unsigned int *site_pattern_map, unsigned int *patterns_out, int device);
const char *pllgpu_compress_last_error(void);

/* ---- the flat seam's tip-tip pair (src/pll.h:1049-1071, src/core_partials.c:1013-1210, :82-200) --------------
 * The caller's lookup table in the REFERENCE's layout: entry (j, k) of two tip codes at index (j << ceil(log2(ncodes)))
 * + k (16 j + k for 4 states, where the code is the state mask) of rate_cats x states_padded doubles = the parent entry
 * of a cherry showing j and k. Matrices in the caller's layout [rate][row][states_padded]. Both calls synchronise. */
int pllgpu_create_lookup(pllgpu_ctx_t *ctx, double *lookup_host, const double *left_host, const double *right_host,
                         const unsigned long long *tipmap_host, unsigned int ncodes);
int pllgpu_tt_from_lookup(pllgpu_ctx_t *ctx, double *parent_host, const unsigned char *left_codes,
                          const unsigned char *right_codes, const double *lookup_host, unsigned int sites, unsigned int ncodes);

/* ---- the exchange of a site-sharded run (pll_gpu_edge_loglikelihood_allreduce) ---------------- */
/* make the context's device the calling thread's current one while a collective library enqueues on the context's
 * stream from the host side of this boundary; *previous (-1: nothing changed) goes to pllgpu_leave_device afterwards */
int pllgpu_enter_device(pllgpu_ctx_t *ctx, int *previous);
void pllgpu_leave_device(int previous);
/* two doubles of device memory owned by the context: {lnL, sequence}, the operand of the all-reduce */
double *pllgpu_reduce_buffer(pllgpu_ctx_t *ctx);
/* after the collective has been enqueued on the context's stream: a one-lane kernel copies the reduced
 * pair to mapped host memory (value first, sequence word behind it) and the call polls for
 * expected_sequence (= ranks x the sequence every rank used). Never synchronises: a collective that a peer does not
 * join never completes, so after 50 ms the poll turns to hipStreamQuery and gives up with PLLGPU_ERUNTIME after
 * timeout_ms (<= 0: 60 s); a drained stream with another sequence word = the ranks are out of step. */
int pllgpu_reduce_fetch(pllgpu_ctx_t *ctx, double expected_sequence, double *value_out, int timeout_ms);
/* a rank whose evaluation failed: {-inf, sequence} as its operand, so that it still takes part in the collective */
int pllgpu_reduce_poison(pllgpu_ctx_t *ctx, double sequence);

/* ---- stream / timing ---------------------------------------------------------------------- */
int pllgpu_set_stream(pllgpu_ctx_t *ctx, void *hip_stream);
void *pllgpu_get_stream(const pllgpu_ctx_t *ctx);
int pllgpu_synchronize(pllgpu_ctx_t *ctx);
int pllgpu_timer_start(pllgpu_ctx_t *ctx);
double pllgpu_timer_stop(pllgpu_ctx_t *ctx); /* ms, < 0 on error */
unsigned int pllgpu_last_launch_count(const pllgpu_ctx_t *ctx);
/* HBM bytes the kernels of the last pllgpu_update_partials call had to move by construction
 * (child reads + parent and scaler writes of every launch as it was grouped) */
double pllgpu_last_algorithmic_bytes(const pllgpu_ctx_t *ctx);

/* Test hook, host logic only (no device is touched): how pllgpu_update_partials partitions a
 * 4-state x 4-rate op list - classified and level-sorted like the ones the host layer hands over -
 * into chains. Returns the number of launch stages, or 0 when the list does not
 * qualify (anything but producer -> consumer dependencies: it goes through the level scheduler).
 * Per op (arrays of `count`, any may be NULL): stage = the launch it runs in (1-based);
 * chain = its chain's number, -1 for a member of a seven-op group, -2 for an op formed on the fly as
 * another chain's sibling; form = 0 top of a chain, 1 lower step, 2 formed on the fly, 3 group member. */
int pllgpu_debug_chain_plan(const pllgpu_op_t *ops, unsigned int count, unsigned int nodes, unsigned int scale_buffers, int fuse_cc,
                            unsigned int *stage, int *chain, unsigned char *form);

#ifdef __cplusplus
}
#endif
#endif
