/* pll_internal.h - host-side private state of libpll_amd.so.
 *
 * pll_partition_t has no spare field for a device handle, so the partition is
 * over-allocated and the extension block below sits directly behind the public struct. */
#ifndef PLL_INTERNAL_H_
#define PLL_INTERNAL_H_

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/pll_amd.h"
#include "../../../include/pll_amd_device.h"

/* everything declared below is internal to libpll_amd.so: not part of its exported surface */
#pragma GCC visibility push(hidden)

#define PLL_AMD_MAGIC 0x504c4c414d443335ull /* "PLLAMD35" */

/* which side holds the current copy of a mirrored buffer */
enum
{
  SIDE_NONE = 0,   /* never written */
  SIDE_HOST = 1,   /* host copy is newer: upload before the device reads it */
  SIDE_DEVICE = 2, /* device copy is newer: download before the host reads it */
  SIDE_BOTH = 3    /* in sync */
};

typedef struct pll_amd_ext
{
  unsigned long long magic;
  pllgpu_ctx_t *ctx;            /* NULL only in host-only mode (PLL_AMD_HOST_ONLY=1, CPU tests) */
  unsigned char *clv_side;      /* [nodes] */
  unsigned char *scaler_side;   /* [scale_buffers] */
  unsigned int *scaler_entries; /* [scale_buffers] entries last written */
  unsigned char *tipchars_dirty; /* [tips] host codes not yet on the device */
  unsigned char *repeats_dirty;  /* [nodes] class maps not yet on the device */
  unsigned char *pmatrix_dirty;  /* [prob_matrices] */
  unsigned char *freqs_dirty;    /* [rate_matrices] */
  int rate_weights_dirty, pattern_weights_dirty, invariant_dirty, prop_invar_dirty, tipmap_dirty;
  /* tips set through pll_set_tip_states WITHOUT PLL_ATTRIB_PATTERN_TIP are 0/1 indicator CLVs of a
   * state mask: the device is handed the one-byte code per entry instead of the dense CLV and the
   * tip-inner / tip-tip kernels are used for them (PLL_AMD_NO_TIP_CODES=1 switches this off) */
  unsigned char *tip_compact;   /* [tips] 1 = device holds codes, not a dense CLV */
  unsigned char **tipcodes;     /* [tips] code per entry (NULL until set) */
  pll_state_t *ctipmap;         /* code -> mask for compact tips (unused for 4 states: code = mask) */
  unsigned int ctip_count;
  int no_tip_codes;
  int eager_mirror;             /* PLL_AMD_EAGER_MIRROR=1: copy results back after every call */
  int always_upload;            /* PLL_AMD_ALWAYS_UPLOAD=1: treat model arrays as dirty on every call */
  unsigned int sites_alloc;
  /* derivatives: eigenvalues / category rates on the device, the two sumtable contraction
   * matrices (rebuilt when the eigensystem, the frequencies or params_indices change), and which
   * caller-owned host sumtable each device slot stands for */
  unsigned char *eigen_dirty;   /* [rate_matrices] eigensystem (vectors + values) not yet on the device */
  unsigned char *pmatrix_stale; /* [prob_matrices] computed on the device, host mirror not refreshed */
  unsigned char *pmatrix_params; /* [prob_matrices][rate_cats] params_indices pll_update_prob_matrices formed the
                                    matrix with; 0xFF = unknown (written by the caller) */
  /* reversibility bookkeeping for swap_is_exact() (likelihood.c): model_version[set] moves whenever the
   * frequencies or substitution parameters of a set change (setters, pll_gpu_invalidate FREQS / EIGEN);
   * model_foreign[set] = the eigensystem was written by the caller, not computed here from (rates, pi);
   * pmatrix_version = model_version of each category's set at the moment the matrix was formed */
  unsigned int *model_version;   /* [rate_matrices] */
  unsigned char *model_foreign;  /* [rate_matrices] */
  unsigned int *pmatrix_version; /* [prob_matrices][rate_cats] */
  unsigned char *repeats_stale; /* [nodes] class maps computed on the device, host mirror not refreshed */
  unsigned int *repeats_count;  /* [nodes] classes the device found (kept even when the node stays uncompressed) */
  /* A node's class map is a function of its children's maps and the enable rule alone (src/repeats.c:299-382) - not of
   * branch lengths or the model. map_version[node] moves whenever the node's map is (or may have been) written;
   * map_stamp[node] records what the map standing on the device was computed FROM. pll_update_repeats for an op whose
   * stamp still describes its inputs launches nothing and leaves the version alone (repeats.c: stamp_holds) */
  unsigned long long *map_version; /* [nodes] */
  struct pll_map_stamp *map_stamp; /* [nodes] */
  unsigned long long map_clock;
  int map_stamps;                  /* PLL_AMD_REP_STAMPS=0: every call recomputes every map (bit-identity control) */
  int rates_dirty;
  unsigned int eigen_version;   /* bumped whenever an eigensystem or frequency vector changes */
  unsigned int aux_version;     /* eigen_version the device contraction matrices were built from */
  unsigned int *aux_params;     /* [rate_cats] params_indices they were built for */
  const double *sumtable_key[PLLGPU_SUMTABLE_SLOTS];
  unsigned int sumtable_age[PLLGPU_SUMTABLE_SLOTS];
  unsigned int sumtable_clock;
  /* handles whose device table was recycled: a derivative evaluation on one of them must fail, not
   * upload the caller's (never written) buffer */
  const double *sumtable_evicted[PLLGPU_SUMTABLE_SLOTS];
  unsigned int sumtable_evicted_next;
  /* the last operation list that went through the whole of pll_update_partials_rep (partials.c): the same list again,
   * with nothing dirty and nothing that its classification depends on changed (fast_valid is cleared wherever tip forms,
   * class counts or caller-written arrays change), goes straight to the device layer */
  pll_operation_t *fast_ops;
  unsigned int fast_count, fast_cap, fast_lo, fast_hi;
  int fast_taken; /* the last pll_update_partials went straight to the launches (pll_gpu_last_update_replayed) */
  int fast_valid;
  double reduce_step;           /* collective evaluations issued so far (group.c: the ranks count in step) */
  /* what pll_gpu_allreduce_prepare established for the communicator last used with this partition: everything that
   * can fail before a collective is enqueued is done once, there, not on the per-step path */
  void *reduce_comm;
  int reduce_ranks;
  int reduce_timeout_ms;
  double *reduce_pair;
  /* scheduler scratch (grown on demand) */
  pllgpu_op_t *gops;
  unsigned int gops_cap;
  int *lvl_clv_w, *lvl_clv_r, *lvl_sc_w, *lvl_sc_r;
} pll_amd_ext_t;

typedef struct pll_map_stamp
{
  unsigned int left, right;         /* the children the map was computed from, in the op's order ... */
  unsigned long long lver, rver;    /* ... and the versions of their maps at that moment */
  unsigned int lookup;              /* lookup_buffer_size of the enable rule (src/repeats.c:100-110) */
  int valid;
} pll_map_stamp_t;

static inline pll_amd_ext_t *pll_ext(const pll_partition_t *p)
{
  pll_amd_ext_t *x = (pll_amd_ext_t *)(p + 1);
  return x->magic == PLL_AMD_MAGIC ? x : NULL;
}

/* error convention of the reference: code + message in thread-locals (src/pll.c:24-25) */
void pll_set_error(int code, const char *fmt, ...);
/* map the device layer's status + text onto pll_errno / pll_errmsg */
void pll_set_gpu_error(const char *where);

unsigned int pll_sites_alloc(const pll_partition_t *p);
/* bring every input of the hot path that is stale on the device up to date; returns PLL_SUCCESS */
int pll_flush_model(pll_partition_t *p, pll_amd_ext_t *x);
int pll_flush_clv(pll_partition_t *p, pll_amd_ext_t *x, unsigned int clv_index);
int pll_flush_scaler(pll_partition_t *p, pll_amd_ext_t *x, int scaler_index);
/* model arrays + eigensystems + category rates: what the derivative and P-matrix kernels read */
int pll_flush_eigen(pll_partition_t *p, pll_amd_ext_t *x);
/* class maps of the parents of `ops` on the device, dependency level by level (repeats.c) */
int pll_update_repeats_device(pll_partition_t *p, pll_amd_ext_t *x, const pll_operation_t *ops,
                              unsigned int count, const unsigned int *level, unsigned int nlevels);
int pll_flush_pmatrix(pll_partition_t *p, pll_amd_ext_t *x, unsigned int first, unsigned int last);
int pll_flush_repeats(pll_partition_t *p, pll_amd_ext_t *x, unsigned int node);
/* the class map of `node` (all nodes: node < 0) was written by something other than the class kernels, or must be
 * taken as such: whatever was derived from it is computed again by the next pll_update_repeats */
void pll_maps_touched(pll_amd_ext_t *x, const pll_partition_t *p, int node);
int pll_is_pattern_tip(const pll_partition_t *p, unsigned int clv_index);
/* the device reads this node as tip codes (PATTERN_TIP tip, or a compact indicator tip) */
int pll_tip_by_codes(const pll_partition_t *p, unsigned int clv_index);
/* give a compact tip a dense device CLV again (someone needs it as an ordinary CLV) */
void pll_tip_densify(pll_partition_t *p, unsigned int clv_index);

/* pll_gpu_edge_loglikelihood_async with the sequence word chosen by the caller (group.c) */
int pll_gpu_edge_loglikelihood_numbered(pll_partition_t *p, unsigned int parent_clv_index, int parent_scaler_index,
                                        unsigned int child_clv_index, int child_scaler_index, unsigned int matrix_index,
                                        const unsigned int *freqs_indices, double *device_result, double sequence);

#pragma GCC visibility pop

#endif
