/* step_loop.c - the CALLER side of bench.py's timed region, in C: K times
 *     pll_update_partials_rep(partition, ops, count, update_repeats);
 *     lnL = pll_compute_edge_loglikelihood(partition, edge...)      (or the group form of a sharded run)
 * which is how an application (RAxML-NG, ModelTest-NG: C / C++) drives the library. Nothing here is on the
 * product path and nothing is computed here: the two entry points are passed in as function pointers from
 * the loaded libpll_amd.so (this file links against nothing), so the loop measures the library's C ABI without
 * the ~4 us per call that Python's ctypes adds between the result of one step and the launches of the next.
 */
#include <time.h>

typedef void (*update_fn)(void *partition, const void *ops, unsigned int count, unsigned int update_repeats);
typedef double (*edge_fn)(void *partition, unsigned int parent_clv, int parent_scaler, unsigned int child_clv,
                          int child_scaler, unsigned int matrix, const unsigned int *freqs_indices, double *persite);
typedef double (*group_edge_fn)(void *partition, void *group, unsigned int parent_clv, int parent_scaler,
                                unsigned int child_clv, int child_scaler, unsigned int matrix,
                                const unsigned int *freqs_indices, double *persite);
typedef void (*invalidate_fn)(void *partition, unsigned int what, int index);
#define FORGET_REPEATS 1024u /* include/pll_amd.h: PLL_GPU_FORGET_REPEATS */

/* edge = {parent_clv, parent_scaler, child_clv, child_scaler, matrix}; group == NULL: the plain evaluation.
 * first_update_repeats: 0 = every step with update_repeats = 0; 1 = the first step with update_repeats = 1, the
 * others with 0 (class maps of a site-repeats partition formed once and re-used until the topology changes);
 * 2 = EVERY step with update_repeats = 1, which is what the reference's pll_update_partials is
 * (src/partials.c:237-242) - on an unchanged tree the library finds every map's inputs as they were and computes none;
 * 3 = the same with invalidate(partition, PLL_GPU_FORGET_REPEATS, -1) ahead of every step: every class map of the
 * traversal is computed again by every step (what the reference does on every call, and what a step after a change
 * of all tips would cost). Returns the seconds the K steps took on the calling thread; *lnl = the last step's
 * value. */
double pllwl_step_loop(update_fn update, edge_fn edge_lnl, group_edge_fn group_edge_lnl, void *partition, void *group,
                       const void *ops, unsigned int count, unsigned int first_update_repeats, const int *edge,
                       const unsigned int *freqs_indices, unsigned int steps, double *lnl, invalidate_fn invalidate)
{
  struct timespec a, b;
  double v = 0.0;
  unsigned int k, ur = first_update_repeats ? 1u : 0u;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (k = 0; k < steps; ++k)
  {
    if (first_update_repeats == 3u && invalidate) invalidate(partition, FORGET_REPEATS, -1);
    update(partition, ops, count, ur);
    if (first_update_repeats == 1u) ur = 0u; /* 1 -> 0 after the first step; 0 stays 0, 2 stays 1 */
    v = group ? group_edge_lnl(partition, group, (unsigned int)edge[0], edge[1], (unsigned int)edge[2], edge[3], (unsigned int)edge[4], freqs_indices, 0)
              : edge_lnl(partition, (unsigned int)edge[0], edge[1], (unsigned int)edge[2], edge[3], (unsigned int)edge[4], freqs_indices, 0);
  }
  clock_gettime(CLOCK_MONOTONIC, &b);
  *lnl = v;
  return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

/* the same loop for a run whose exchange is the library's RCCL form (include/pll_amd.h:
 * pll_gpu_edge_loglikelihood_allreduce - evaluation, one ncclAllReduce on the partition's stream, the sum back) */
typedef double (*allreduce_edge_fn)(void *partition, void *nccl_comm, unsigned int parent_clv, int parent_scaler,
                                    unsigned int child_clv, int child_scaler, unsigned int matrix,
                                    const unsigned int *freqs_indices);

double pllwl_step_loop_allreduce(update_fn update, allreduce_edge_fn allreduce_edge_lnl, void *partition, void *nccl_comm,
                                 const void *ops, unsigned int count, unsigned int first_update_repeats, const int *edge,
                                 const unsigned int *freqs_indices, unsigned int steps, double *lnl, invalidate_fn invalidate)
{
  struct timespec a, b;
  double v = 0.0;
  unsigned int k, ur = first_update_repeats ? 1u : 0u;
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (k = 0; k < steps; ++k)
  {
    if (first_update_repeats == 3u && invalidate) invalidate(partition, FORGET_REPEATS, -1);
    update(partition, ops, count, ur);
    if (first_update_repeats == 1u) ur = 0u;
    v = allreduce_edge_lnl(partition, nccl_comm, (unsigned int)edge[0], edge[1], (unsigned int)edge[2], edge[3], (unsigned int)edge[4], freqs_indices);
  }
  clock_gettime(CLOCK_MONOTONIC, &b);
  *lnl = v;
  return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}
