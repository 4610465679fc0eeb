/* A site-sharded C caller (SURVEY.md section 8 row e): the reference's 5-taxon likelihood test
 * (test/src/00010_NMDU_lkcalc.c:96-175; log-likelihood -58.887310 on edge (6,7), test/out/00010_NMDU_lkcalc.out)
 * with its 12 sites cut into contiguous ranges, one partition per rank, and the ONE exchange of the path - the
 * sum of the ranks' log-likelihoods (src/core_likelihood.c:1489) - done by the library:
 *
 *     sharded peer N      N processes (forked before anything touches the GPU) meet in the fixed-order
 *                         shared-memory exchange: pll_gpu_group_join / pll_gpu_group_edge_loglikelihood
 *     sharded rccl        one rank, a real RCCL communicator (ncclCommInitAll) handed to
 *                         pll_gpu_edge_loglikelihood_allreduce - the all-reduce reachable from C without Python
 *
 * Built and run by tests/test_gpu_c_caller.py:
 *     gcc -O2 -Iinclude tests/c_caller/sharded.c -Llibpll-2_amd/csrc -lpll_amd -ldl -lm
 */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include "pll_amd.h"

static const char *SEQ[5] = {"WAC-CTA-ATCT", "CCC-TTA-ATGT", "A-C-TAG-CTCT", "CTCTTAA-A-CG", "CAC-TCA-A-TG"};

static pll_partition_t *make_shard(unsigned lo, unsigned hi)
{
  const double freqs[4] = {0.3, 0.4, 0.1, 0.2};
  const double subst[6] = {1, 2.5, 1, 1, 2.5, 1};
  const double brlen[4] = {0.1, 0.2, 1.0, 1.0};
  const unsigned int matrix_indices[4] = {0, 1, 2, 3};
  const unsigned int params[4] = {0, 0, 0, 0};
  double rates[4];
  char buf[16];
  pll_partition_t *p = pll_partition_create(5, 4, 4, hi - lo, 1, 5, 4, 0, PLL_ATTRIB_ARCH_AVX2);
  if (!p) return NULL;
  pll_set_frequencies(p, 0, freqs);
  pll_set_subst_params(p, 0, subst);
  pll_compute_gamma_cats(0.5, 4, rates, PLL_GAMMA_RATES_MEAN);
  pll_set_category_rates(p, rates);
  for (unsigned i = 0; i < 5; ++i)
  {
    memcpy(buf, SEQ[i] + lo, hi - lo);
    buf[hi - lo] = 0;
    if (!pll_set_tip_states(p, i, pll_map_nt, buf)) return NULL;
  }
  if (!pll_update_prob_matrices(p, params, matrix_indices, brlen, 4)) return NULL;
  pll_operation_t ops[3] = {
      {5, PLL_SCALE_BUFFER_NONE, 0, 1, PLL_SCALE_BUFFER_NONE, 1, 1, PLL_SCALE_BUFFER_NONE},
      {6, PLL_SCALE_BUFFER_NONE, 5, 0, PLL_SCALE_BUFFER_NONE, 2, 1, PLL_SCALE_BUFFER_NONE},
      {7, PLL_SCALE_BUFFER_NONE, 3, 1, PLL_SCALE_BUFFER_NONE, 4, 1, PLL_SCALE_BUFFER_NONE}};
  pll_update_partials(p, ops, 3);
  return p;
}

static int rank_main(const char *name, unsigned rank, unsigned world)
{
  const unsigned int params[4] = {0, 0, 0, 0};
  const unsigned lo = 12u * rank / world, hi = 12u * (rank + 1) / world;
  pll_partition_t *p = make_shard(lo, hi);
  if (!p)
  {
    fprintf(stderr, "rank %u: [%d] %s\n", rank, pll_errno, pll_errmsg);
    return 2;
  }
  pll_gpu_group_t *g = pll_gpu_group_join(name, rank, world, 30000);
  if (!g)
  {
    fprintf(stderr, "rank %u: pll_gpu_group_join: [%d] %s\n", rank, pll_errno, pll_errmsg);
    return 3;
  }
  double first = 0;
  int bad = 0;
  for (int step = 0; step < 50; ++step) /* evaluations in step: every rank sees the same bits every time */
  {
    const double lnl = pll_gpu_group_edge_loglikelihood(p, g, 6, PLL_SCALE_BUFFER_NONE, 7, PLL_SCALE_BUFFER_NONE, 0, params, NULL);
    if (step == 0) first = lnl;
    bad |= memcmp(&lnl, &first, sizeof lnl) != 0;
  }
  printf("rank %u sites [%u,%u) lnl %.6f %a\n", rank, lo, hi, first, first);
  fflush(stdout);
  pll_gpu_group_leave(g);
  pll_partition_destroy(p);
  return (!bad && fabs(first + 58.887310) < 5.1e-7) ? 0 : 1;
}

typedef int (*init_all_fn)(void **, int, const int *);
typedef int (*destroy_fn)(void *);

static int rccl_main(void)
{
  const unsigned int params[4] = {0, 0, 0, 0};
  if (!pll_gpu_rccl_available())
  {
    printf("rccl unavailable\n");
    return 77;
  }
  /* the caller owns the communicator; here one rank on device 0 (the same library the product bound) */
  void *h = dlopen(getenv("PLL_AMD_RCCL_LIB") ? getenv("PLL_AMD_RCCL_LIB") : "librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return 4;
  init_all_fn init_all = (init_all_fn)dlsym(h, "ncclCommInitAll");
  destroy_fn destroy = (destroy_fn)dlsym(h, "ncclCommDestroy");
  void *comm = NULL;
  const int dev = 0;
  if (!init_all || init_all(&comm, 1, &dev) != 0 || !comm) return 5;
  pll_partition_t *p = make_shard(0, 12);
  if (!p) return 2;
  double lnl = 0;
  for (int step = 0; step < 5; ++step)
    lnl = pll_gpu_edge_loglikelihood_allreduce(p, comm, 6, PLL_SCALE_BUFFER_NONE, 7, PLL_SCALE_BUFFER_NONE, 0, params);
  /* a failing evaluation still takes part in the collective (operand -inf), reports the rank's own error, and the
   * next collective evaluation is in step again */
  const double failed = pll_gpu_edge_loglikelihood_allreduce(p, comm, 6, PLL_SCALE_BUFFER_NONE, 99, PLL_SCALE_BUFFER_NONE, 0, params);
  const int failed_errno = pll_errno;
  if (isfinite(failed) || failed_errno != PLL_ERROR_PARAM_INVALID)
  {
    fprintf(stderr, "a bad CLV index gave %g, errno %d\n", failed, failed_errno);
    return 6;
  }
  lnl = pll_gpu_edge_loglikelihood_allreduce(p, comm, 6, PLL_SCALE_BUFFER_NONE, 7, PLL_SCALE_BUFFER_NONE, 0, params);
  const double plain = pll_compute_edge_loglikelihood(p, 6, PLL_SCALE_BUFFER_NONE, 7, PLL_SCALE_BUFFER_NONE, 0, params, NULL);
  printf("rccl lnl %.6f plain %.6f\n", lnl, plain);
  if (!isfinite(lnl)) fprintf(stderr, "[%d] %s\n", pll_errno, pll_errmsg);
  pll_partition_destroy(p);
  if (destroy) destroy(comm);
  return (fabs(lnl + 58.887310) < 5.1e-7 && lnl == plain) ? 0 : 1;
}

int main(int argc, char **argv)
{
  if (argc >= 2 && strcmp(argv[1], "rccl") == 0) return rccl_main();
  const unsigned world = argc >= 3 ? (unsigned)atoi(argv[2]) : 2u;
  char name[64];
  snprintf(name, sizeof name, "/pllamd-ccaller-%d", (int)getpid());
  for (unsigned r = 0; r < world; ++r)
    if (fork() == 0) _exit(rank_main(name, r, world));
  int bad = 0, st;
  while (wait(&st) > 0) bad |= !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
  return bad;
}
