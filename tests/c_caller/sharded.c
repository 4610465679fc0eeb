/* A site-sharded C caller (SURVEY.md section 8 row e): the reference's 5-taxon likelihood test
 * (test/src/00010_NMDU_lkcalc.c:96-175; log-likelihood -58.887310 on edge (6,7), test/out/00010_NMDU_lkcalc.out)
 * with its 12 sites cut into contiguous ranges, one partition per rank, and the ONE exchange of the path - the
 * sum of the ranks' log-likelihoods (src/core_likelihood.c:1489) - done by the library:
 *
 *     sharded peer N      N processes (forked before anything touches the GPU) meet in the fixed-order
 *                         shared-memory exchange: pll_gpu_group_join / pll_gpu_group_edge_loglikelihood
 *     sharded rccl        one rank, a real RCCL communicator (ncclCommInitAll) handed to
 *                         pll_gpu_edge_loglikelihood_allreduce - the all-reduce reachable from C without Python
 *     sharded double N LIB   N forked ranks on ONE device through pll_gpu_edge_loglikelihood_allreduce with the
 *                         stream-ordered stand-in for librccl (tests/c_caller/rccl_double.c = LIB; real RCCL refuses
 *                         two ranks on one device): ranks x step sequence words, a step in which the last rank's
 *                         evaluation fails (-inf on every rank, each with the right pll_errno), in step again after it
 *     sharded missing N LIB  the same, but the last rank sits one collective out: the others get -inf and an error
 *                         after PLL_AMD_REDUCE_TIMEOUT_MS instead of blocking for ever in hipStreamSynchronize
 *
 * Built and run by tests/test_gpu_c_caller.py:
 *     gcc -O2 -Iinclude tests/c_caller/sharded.c -Llibpll-2_amd/csrc -lpll_amd -ldl -lm
 */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "pll_amd.h"
#include "rccl_double.h"

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

static double wall_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* one rank of the runs through the RCCL stand-in; `missing`: the last rank skips the second collective */
static int double_rank_main(rccl_double_shared_t *shared, unsigned rank, unsigned world, int missing)
{
  const unsigned int params[4] = {0, 0, 0, 0};
  const unsigned lo = 12u * rank / world, hi = 12u * (rank + 1) / world;
  const int last = rank == world - 1;
  pll_partition_t *p = make_shard(lo, hi);
  if (!p)
  {
    fprintf(stderr, "rank %u: [%d] %s\n", rank, pll_errno, pll_errmsg);
    return 2;
  }
  rccl_double_comm_t comm = {RCCL_DOUBLE_MAGIC, (int)rank, (int)world, shared, missing ? 6000 : 30000, 0, NULL};
  /* set-up errors surface here, before the first collective: a NULL communicator, then the real one */
  if (pll_gpu_allreduce_prepare(p, NULL) || pll_errno != PLL_ERROR_PARAM_INVALID) return 3;
  if (!pll_gpu_allreduce_prepare(p, &comm))
  {
    fprintf(stderr, "rank %u: pll_gpu_allreduce_prepare: [%d] %s\n", rank, pll_errno, pll_errmsg);
    return 3;
  }
  double first = pll_gpu_edge_loglikelihood_allreduce(p, &comm, 6, PLL_SCALE_BUFFER_NONE, 7, PLL_SCALE_BUFFER_NONE, 0, params);
  if (!isfinite(first))
  {
    fprintf(stderr, "rank %u: first collective: [%d] %s\n", rank, pll_errno, pll_errmsg);
    return 4;
  }
  if (missing)
  {
    if (last)
    { /* sits the second collective out (but stays alive until the others have given up) */
      sleep(5);
      pll_partition_destroy(p);
      return 0;
    }
    const double t0 = wall_ms();
    const double v = pll_gpu_edge_loglikelihood_allreduce(p, &comm, 6, PLL_SCALE_BUFFER_NONE, 7, PLL_SCALE_BUFFER_NONE, 0, params);
    const double waited = wall_ms() - t0;
    printf("rank %u missing-peer value %g errno %d after %.0f ms: %s\n", rank, v, pll_errno, waited, pll_errmsg);
    fflush(stdout);
    const int ok = !isfinite(v) && pll_errno == PLL_ERROR_GPU_RUNTIME && strstr(pll_errmsg, "did not complete") && waited < 5000;
    pll_partition_destroy(p); /* (the stand-in's exchange gives up after 6 s: the stream drains, the destroy returns) */
    return ok ? 0 : 5;
  }
  int bad = 0;
  for (int step = 0; step < 20; ++step)
  {
    const double v = pll_gpu_edge_loglikelihood_allreduce(p, &comm, 6, PLL_SCALE_BUFFER_NONE, 7, PLL_SCALE_BUFFER_NONE, 0, params);
    bad |= memcmp(&v, &first, sizeof v) != 0;
  }
  /* the last rank's evaluation fails (a CLV index out of range): it still brings an operand, everybody returns -inf */
  const double failed = pll_gpu_edge_loglikelihood_allreduce(p, &comm, 6, PLL_SCALE_BUFFER_NONE, last ? 99 : 7, PLL_SCALE_BUFFER_NONE, 0, params);
  const int failed_errno = pll_errno;
  const int want_errno = last ? PLL_ERROR_PARAM_INVALID : PLL_ERROR_GPU_RUNTIME;
  if (isfinite(failed) || failed_errno != want_errno || (!last && !strstr(pll_errmsg, "another rank")))
  {
    fprintf(stderr, "rank %u: failing step gave %g, errno %d (%s), expected -inf and %d\n", rank, failed, failed_errno, pll_errmsg, want_errno);
    bad = 1;
  }
  /* ... and the ranks are in step again afterwards */
  const double again = pll_gpu_edge_loglikelihood_allreduce(p, &comm, 6, PLL_SCALE_BUFFER_NONE, 7, PLL_SCALE_BUFFER_NONE, 0, params);
  bad |= memcmp(&again, &first, sizeof again) != 0;
  printf("rank %u sites [%u,%u) double lnl %.6f %a\n", rank, lo, hi, first, first);
  fflush(stdout);
  pll_partition_destroy(p);
  return (!bad && fabs(first + 58.887310) < 5.1e-7) ? 0 : 1;
}

static int double_main(int argc, char **argv, int missing)
{
  if (argc < 4) return 64;
  const unsigned world = (unsigned)atoi(argv[2]);
  if (world < 1 || world > RCCL_DOUBLE_MAX_RANKS) return 64;
  setenv("PLL_AMD_RCCL_LIB", argv[3], 1);
  if (missing) setenv("PLL_AMD_REDUCE_TIMEOUT_MS", "1500", 1);
  rccl_double_shared_t *shared = (rccl_double_shared_t *)mmap(NULL, sizeof *shared, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (shared == MAP_FAILED) return 65;
  for (unsigned r = 0; r < world; ++r) /* forked before anything touches the GPU */
    if (fork() == 0) _exit(double_rank_main(shared, r, world, missing));
  int bad = 0, st;
  while (wait(&st) > 0) bad |= !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
  return bad;
}

int main(int argc, char **argv)
{
  if (argc >= 2 && strcmp(argv[1], "rccl") == 0) return rccl_main();
  if (argc >= 2 && strcmp(argv[1], "double") == 0) return double_main(argc, argv, 0);
  if (argc >= 2 && strcmp(argv[1], "missing") == 0) return double_main(argc, argv, 1);
  const unsigned world = argc >= 3 ? (unsigned)atoi(argv[2]) : 2u;
  char name[64];
  snprintf(name, sizeof name, "/pllamd-ccaller-%d", (int)getpid());
  for (unsigned r = 0; r < world; ++r)
    if (fork() == 0) _exit(rank_main(name, r, world));
  int bad = 0, st;
  while (wait(&st) > 0) bad |= !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
  return bad;
}
