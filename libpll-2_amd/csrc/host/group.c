/* group.c - the one exchange of a site-sharded run (SURVEY.md section 8 row e).
 *
 * Sites are independent through every CLV update; the single cross-site operation of the path is the
 * sum of the site log-likelihoods, in the reference a sequential `logl += site_lk` over all sites
 * (src/core_likelihood.c:1489). With one partition per GPU over a contiguous site range the sum over
 * ranks of one double is all that is ever exchanged. Two forms (include/pll_amd.h):
 *
 *  pll_gpu_group_*   ranks of one node meet in a POSIX shared-memory segment and add the slots in RANK
 *                    ORDER: same bits on every rank, every run. Pure host code: the device has already
 *                    left {lnL, sequence} in host memory (csrc/hip/kernels_common.h: publish_block_sum),
 *                    what is added here is a cache-line hand-off between cores.
 *  pll_gpu_*allreduce*  one ncclAllReduce on the partition's stream, librccl bound with dlopen().
 */
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "pll_internal.h"

#define GROUP_MAGIC 0x504c4c4752503032ull /* "PLLGRP02" */
#define GROUP_MAX_VALUES 6u
#define GROUP_JOIN_ATTEMPTS 200u /* x 1 ms: how long a rank keeps looking for the segment that replaces a stale one */

/* one cache line per (rank, parity): the step word is written last (release) and read first (acquire) */
typedef struct group_slot
{
  _Alignas(64) volatile unsigned long long step;
  double v[GROUP_MAX_VALUES];
  unsigned int count;
} group_slot_t;

typedef struct group_header
{
  _Alignas(64) volatile unsigned long long magic;
  volatile unsigned int size;
  volatile unsigned int joined;   /* ranks that have cleared their slots */
  volatile unsigned int left;     /* ranks that have gone: the last one unlinks the name */
  volatile unsigned int poisoned; /* a rank found the segment stale, was refused or gave up waiting: whoever still
                                   * waits in it stops at once (whoever set the word has removed the name) */
} group_header_t;

/* segment = header | owner[size] (one word per rank: who sits there, 0 = free) | slots[size][2] */
static size_t owners_bytes(unsigned int size) { return (((size_t)size * sizeof(unsigned int)) + 63u) & ~(size_t)63u; }

struct pll_gpu_group
{
  unsigned int rank, size;
  unsigned long long step;
  group_header_t *hdr;
  volatile unsigned int *owner; /* [size] */
  group_slot_t *slots;          /* [size][2] */
  size_t bytes;
  int timeout_ms;
  char name[96];
};

static double now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static inline void cpu_relax(void)
{
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

/* The first to call a segment unusable removes its name - exactly once, so that the fresh segment another rank may
 * already have created under the name is never the one removed - and everybody who still waits in it learns of it. */
static void group_poison(pll_gpu_group_t *g)
{
  if (__atomic_exchange_n(&g->hdr->poisoned, 1u, __ATOMIC_ACQ_REL) == 0) shm_unlink(g->name);
}

static void group_unmap(pll_gpu_group_t *g)
{
  if (g->hdr) munmap((void *)g->hdr, g->bytes);
  g->hdr = NULL;
}

/* A seat holds the process id of the rank that took it. A run that was killed while joining leaves seats of processes
 * that no longer exist: the ranks of the next run under the name would otherwise count them as present and walk through
 * the barrier (ranks of one group share a PID namespace: processes or threads of one node). */
static int seated_by_the_dead(const pll_gpu_group_t *g)
{
  for (unsigned int r = 0; r < g->size; ++r)
  {
    const unsigned int o = __atomic_load_n(&g->owner[r], __ATOMIC_ACQUIRE);
    if (o && kill((pid_t)(o & 0x7FFFFFFFu), 0) != 0 && errno == ESRCH) return 1;
  }
  return 0;
}

/* one attempt: 1 joined, 0 failed for good (pll_errno set), -1 the segment under the name was not this run's (a run
 * that crashed or gave up left it behind; it has been poisoned and its name removed): look again */
static int group_join_once(pll_gpu_group_t *g, double t0)
{
  const unsigned int size = g->size, rank = g->rank;
  /* whoever comes first creates the segment (zero-filled by the kernel) and sizes it; the others find it */
  int fd = shm_open(g->name, O_RDWR | O_CREAT, 0600);
  if (fd < 0)
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_join: shm_open(%s): %s", g->name, strerror(errno));
    return 0;
  }
  struct stat st;
  if (fstat(fd, &st) != 0 || ((size_t)st.st_size < g->bytes && ftruncate(fd, (off_t)g->bytes) != 0))
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_join: sizing %s: %s", g->name, strerror(errno));
    close(fd);
    return 0;
  }
  void *mem = mmap(NULL, g->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (mem == MAP_FAILED)
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_join: mmap(%s): %s", g->name, strerror(errno));
    return 0;
  }
  g->hdr = (group_header_t *)mem;
  g->owner = (volatile unsigned int *)((char *)mem + sizeof(group_header_t));
  g->slots = (group_slot_t *)((char *)mem + sizeof(group_header_t) + owners_bytes(size));
  /* a segment of a run of another size under the same name is refused, not reused (and not removed: it may be alive) */
  unsigned int expect = 0;
  if (!__atomic_compare_exchange_n(&g->hdr->size, &expect, size, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE) && expect != size)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_group_join: %s exists with %u ranks, not %u (names must be unique per run)", g->name, expect, size);
    group_unmap(g);
    return 0;
  }
  /* Not this run's segment: somebody already gave it up, a rank of it has already left (every rank of a run joins
   * before the first one can leave), all its ranks are there already, or this rank's seat is taken. (a
   * segment a failed or killed run left behind let the first ranks of the next run through the barrier at once and
   * the rest spin until their time-out.) */
  unsigned int seat = 0;
  if (__atomic_load_n(&g->hdr->poisoned, __ATOMIC_ACQUIRE) || __atomic_load_n(&g->hdr->left, __ATOMIC_ACQUIRE) != 0 ||
      __atomic_load_n(&g->hdr->joined, __ATOMIC_ACQUIRE) >= size || seated_by_the_dead(g) ||
      !__atomic_compare_exchange_n(&g->owner[rank], &seat, (unsigned int)getpid() | 0x80000000u, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE))
  {
    group_poison(g);
    group_unmap(g);
    return -1;
  }
  memset((void *)&g->slots[2 * rank], 0, 2 * sizeof(group_slot_t));
  __atomic_store_n(&g->hdr->magic, GROUP_MAGIC, __ATOMIC_RELEASE);
  __atomic_fetch_add(&g->hdr->joined, 1u, __ATOMIC_ACQ_REL);
  unsigned int spins = 0;
  while (__atomic_load_n(&g->hdr->joined, __ATOMIC_ACQUIRE) < size)
  {
    cpu_relax();
    if ((++spins & 0xFFu) != 0) continue;
    if (__atomic_load_n(&g->hdr->poisoned, __ATOMIC_ACQUIRE))
    {
      group_unmap(g);
      return -1;
    }
    if (now_ms() - t0 > g->timeout_ms)
    {
      pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_join: %u of %u ranks joined %s within %d ms", g->hdr->joined, size, g->name, g->timeout_ms);
      group_poison(g); /* (and the name is gone: the next run under it starts from nothing) */
      group_unmap(g);
      return 0;
    }
  }
  if (__atomic_load_n(&g->hdr->poisoned, __ATOMIC_ACQUIRE))
  {
    group_unmap(g);
    return -1;
  }
  return 1;
}

pll_gpu_group_t *pll_gpu_group_join(const char *name, unsigned int rank, unsigned int size, int timeout_ms)
{
  if (!name || name[0] != '/' || strlen(name) >= sizeof(((pll_gpu_group_t *)0)->name) || size == 0 || rank >= size || size > 4096)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_group_join: name must start with '/', rank < size <= 4096");
    return NULL;
  }
  pll_gpu_group_t *g = (pll_gpu_group_t *)calloc(1, sizeof *g);
  if (!g)
  {
    pll_set_error(PLL_ERROR_MEM_ALLOC, "pll_gpu_group_join: out of memory");
    return NULL;
  }
  g->rank = rank;
  g->size = size;
  g->bytes = sizeof(group_header_t) + owners_bytes(size) + (size_t)size * 2 * sizeof(group_slot_t);
  g->timeout_ms = timeout_ms > 0 ? timeout_ms : 60000;
  strcpy(g->name, name);
  const double t0 = now_ms();
  int rc = -1;
  for (unsigned int attempt = 0; rc < 0 && attempt < GROUP_JOIN_ATTEMPTS && now_ms() - t0 <= g->timeout_ms; ++attempt)
  {
    if (attempt)
    {
      const struct timespec ms = {0, 1000000};
      nanosleep(&ms, NULL); /* whoever poisoned the stale segment is about to remove its name */
    }
    rc = group_join_once(g, t0);
  }
  if (rc < 0)
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_join: %s stayed a stale or foreign segment (rank %u seated twice? names must be unique per run)", name, rank);
  if (rc <= 0)
  {
    free(g);
    return NULL;
  }
  return g;
}

void pll_gpu_group_leave(pll_gpu_group_t *g)
{
  if (!g) return;
  /* the last rank to leave removes the name (unless the segment was given up: then the name is gone already, and a
   * later run may own it); a killed run leaves it behind, for the next run under the name to find stale and replace */
  /* "removed exactly once": whoever flips `poisoned` first owns the unlink - the last leaver here, or a rank of a later
   * run that found the segment stale (group_poison). Decided with ONE atomic: a read of the flag followed by the unlink
   * let a later run poison, unlink and re-create the name in between, and this call then removed the NEW segment's name */
  if (__atomic_add_fetch(&g->hdr->left, 1u, __ATOMIC_ACQ_REL) >= g->size && __atomic_exchange_n(&g->hdr->poisoned, 1u, __ATOMIC_ACQ_REL) == 0u)
    shm_unlink(g->name);
  group_unmap(g);
  free(g);
}

unsigned int pll_gpu_group_rank(const pll_gpu_group_t *g) { return g ? g->rank : 0; }
unsigned int pll_gpu_group_size(const pll_gpu_group_t *g) { return g ? g->size : 0; }

/* Two slots per rank suffice: a rank writes step k + 2 into the slot of step k only after every rank has
 * published step k + 1, and a rank publishes step k + 1 only after it has read every slot of step k. */
int pll_gpu_group_sum(pll_gpu_group_t *g, const double *local, unsigned int count, double *global)
{
  unsigned int r, i;
  if (!g || !local || !global || count == 0 || count > GROUP_MAX_VALUES)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_group_sum: 1 <= count <= %u", GROUP_MAX_VALUES);
    return PLL_FAILURE;
  }
  const unsigned long long step = ++g->step;
  const unsigned int par = (unsigned int)(step & 1u);
  group_slot_t *mine = &g->slots[2 * g->rank + par];
  for (i = 0; i < count; ++i) mine->v[i] = local[i];
  mine->count = count;
  __atomic_store_n(&mine->step, step, __ATOMIC_RELEASE);
  double acc[GROUP_MAX_VALUES] = {0};
  double t0 = 0;
  for (r = 0; r < g->size; ++r)
  {
    const group_slot_t *s = &g->slots[2 * r + par];
    unsigned int spins = 0;
    while (__atomic_load_n(&s->step, __ATOMIC_ACQUIRE) != step)
    {
      cpu_relax();
      if ((++spins & 0xFFFu) == 0)
      {
        const double t = now_ms();
        if (t0 == 0) t0 = t;
        if (__atomic_load_n(&g->hdr->poisoned, __ATOMIC_ACQUIRE))
        {
          pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_sum: the group was given up at step %llu (a rank timed out, or another run claimed %s)", step, g->name);
          return PLL_FAILURE;
        }
        if (t - t0 > g->timeout_ms)
        {
          pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_sum: rank %u did not reach step %llu within %d ms", r, step, g->timeout_ms);
          group_poison(g); /* the others stop waiting too */
          return PLL_FAILURE;
        }
      }
    }
    if (s->count != count)
    {
      pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_group_sum: rank %u brought %u values to step %llu, rank %u brings %u", r, s->count, step, g->rank, count);
      return PLL_FAILURE;
    }
    for (i = 0; i < count; ++i) acc[i] = r ? acc[i] + s->v[i] : s->v[i]; /* rank order, on every rank */
  }
  for (i = 0; i < count; ++i) global[i] = acc[i];
  return PLL_SUCCESS;
}

double pll_gpu_group_edge_loglikelihood(pll_partition_t *p, pll_gpu_group_t *g, unsigned int parent_clv_index,
                                        int parent_scaler_index, unsigned int child_clv_index, int child_scaler_index,
                                        unsigned int matrix_index, const unsigned int *freqs_indices, double *persite_lnl)
{
  /* a rank whose evaluation failed still takes part (-inf poisons the sum): nobody waits for a time-out */
  double mine = pll_compute_edge_loglikelihood(p, parent_clv_index, parent_scaler_index, child_clv_index,
                                               child_scaler_index, matrix_index, freqs_indices, persite_lnl);
  double all = -INFINITY;
  if (!g) return mine;
  const int my_errno = pll_errno;
  if (!pll_gpu_group_sum(g, &mine, 1, &all)) return -INFINITY;
  if (!isfinite(all) && isfinite(mine))
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_edge_loglikelihood: another rank's evaluation failed");
  else if (!isfinite(mine))
    pll_errno = my_errno;
  return isfinite(all) ? all : -INFINITY;
}

/* Branch-length optimisation of a sharded run (SURVEY section 8 row f1 next to row e): every rank evaluates the two
 * derivatives of ITS sites' log-likelihood at the same branch length; the whole alignment's are their sums
 * (src/core_derivatives.c:643-849 accumulates over sites exactly like the log-likelihood does). Added in rank order, so
 * every rank holds the same bits and takes the same Newton step: the ranks' branch lengths cannot drift apart by an
 * ulp, which an arrival-order reduction would allow. A rank whose evaluation failed says so in a third value. */
int pll_gpu_group_likelihood_derivatives(pll_partition_t *p, pll_gpu_group_t *g, int parent_scaler_index, int child_scaler_index,
                                         double branch_length, const unsigned int *params_indices, const double *sumtable,
                                         double *d_f, double *dd_f)
{
  /* {d_f, dd_f, failed}: the third value counts the ranks whose evaluation failed - an explicit word, not a NaN in the
   * sums: a derivative that IS NaN on one shard (a site of zero likelihood) is a result, and every rank shall see it as
   * one instead of "another rank failed" */
  double mine[3] = {0.0, 0.0, 0.0}, all[3] = {NAN, NAN, 0.0};
  const int ok = pll_compute_likelihood_derivatives(p, parent_scaler_index, child_scaler_index, branch_length, params_indices, sumtable,
                                                    &mine[0], &mine[1]);
  if (!g)
  {
    if (ok) *d_f = mine[0], *dd_f = mine[1];
    return ok;
  }
  const int my_errno = pll_errno;
  if (!ok) mine[0] = mine[1] = 0.0, mine[2] = 1.0;
  if (!pll_gpu_group_sum(g, mine, 3, all)) return PLL_FAILURE;
  if (!ok)
  {
    pll_errno = my_errno;
    return PLL_FAILURE;
  }
  if (all[2] != 0.0)
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_likelihood_derivatives: another rank's evaluation failed");
    return PLL_FAILURE;
  }
  *d_f = all[0];
  *dd_f = all[1];
  return PLL_SUCCESS;
}

/* ---- RCCL, bound at run time ------------------------------------------------------------------ */
typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, void *);
typedef int (*nccl_count_fn)(const void *, int *);
typedef const char *(*nccl_errstr_fn)(int);
static struct
{
  void *handle;
  nccl_allreduce_fn allreduce;
  nccl_count_fn count;
  nccl_errstr_fn errstr;
} g_rccl;

static void rccl_bind_once(void)
{
  const char *names[] = {getenv("PLL_AMD_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (size_t i = 0; i < sizeof names / sizeof *names && !g_rccl.handle; ++i)
    if (names[i] && names[i][0]) g_rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl.handle) return;
  g_rccl.allreduce = (nccl_allreduce_fn)dlsym(g_rccl.handle, "ncclAllReduce");
  g_rccl.count = (nccl_count_fn)dlsym(g_rccl.handle, "ncclCommCount");
  g_rccl.errstr = (nccl_errstr_fn)dlsym(g_rccl.handle, "ncclGetErrorString");
  if (!g_rccl.allreduce || !g_rccl.count) g_rccl.allreduce = NULL;
}

/* (partitions may be driven from concurrent threads: the library is bound once) */
static int rccl_bind(void)
{
  static pthread_once_t once = PTHREAD_ONCE_INIT;
  pthread_once(&once, rccl_bind_once);
  return g_rccl.allreduce != NULL;
}

int pll_gpu_rccl_available(void) { return rccl_bind(); }

#define NCCL_DOUBLE 8 /* ncclFloat64, rccl.h:467 */
#define NCCL_SUM 0    /* ncclSum, rccl.h:448 */

static pll_amd_ext_t *reduce_ctx(pll_partition_t *p, void *comm, const char *who)
{
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  if (!x || !x->ctx)
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "%s: no MI355X context behind this partition", who);
    return NULL;
  }
  if (!comm)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "%s: the communicator is NULL", who);
    return NULL;
  }
  if (!rccl_bind())
  {
    pll_set_error(PLL_ERROR_GPU_UNAVAILABLE, "%s: no RCCL library could be opened (librccl.so.1; PLL_AMD_RCCL_LIB names another)", who);
    return NULL;
  }
  return x;
}

int pll_gpu_allreduce_lnl(pll_partition_t *p, void *comm, double *device_values, unsigned int count)
{
  pll_amd_ext_t *x = reduce_ctx(p, comm, "pll_gpu_allreduce_lnl");
  if (!x) return PLL_FAILURE;
  if (!device_values || !count)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_allreduce_lnl: nothing to reduce");
    return PLL_FAILURE;
  }
  void *stream = pllgpu_get_stream(x->ctx); /* launches whatever the partition still holds back */
  /* the communicator's rank lives on the partition's device; the calling thread's current device may be another
   * (PLL_AMD_DEVICE=auto, a caller that drives several devices from one thread) */
  int previous = -1;
  if (pllgpu_enter_device(x->ctx, &previous) != 0)
  {
    pll_set_gpu_error("pll_gpu_allreduce_lnl");
    return PLL_FAILURE;
  }
  const int rc = g_rccl.allreduce(device_values, device_values, count, NCCL_DOUBLE, NCCL_SUM, comm, stream);
  pllgpu_leave_device(previous);
  if (rc != 0)
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "ncclAllReduce: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "failed");
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

/* Everything about a (partition, communicator) pair that can fail BEFORE a collective is enqueued - no context, no
 * RCCL library, ncclCommCount, the 16 bytes of device memory for the operand - is established here, once. A rank that
 * failed at one of these inside a collective step would return without joining the all-reduce and leave its peers
 * blocked in it; failing here, before the rank's first collective, is a set-up error the job sees while
 * it can still agree on it (call it on every rank after creating the communicator and compare the results). */
int pll_gpu_allreduce_prepare(pll_partition_t *p, void *comm)
{
  pll_amd_ext_t *x = reduce_ctx(p, comm, "pll_gpu_allreduce_prepare");
  if (!x) return PLL_FAILURE;
  /* (always asked again, also for the communicator prepared last: a new ncclComm_t may live at the address of a destroyed
   * one with another number of ranks - the per-step entry point only comes here when it has nothing prepared) */
  int ranks = 0;
  const int rc = g_rccl.count(comm, &ranks);
  if (rc != 0 || ranks < 1)
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "ncclCommCount: %s", (rc && g_rccl.errstr) ? g_rccl.errstr(rc) : "failed");
    return PLL_FAILURE;
  }
  double *pair = pllgpu_reduce_buffer(x->ctx);
  if (!pair)
  {
    pll_set_gpu_error("pll_gpu_allreduce_prepare");
    return PLL_FAILURE;
  }
  const char *t = getenv("PLL_AMD_REDUCE_TIMEOUT_MS");
  x->reduce_timeout_ms = (t && atoi(t) > 0) ? atoi(t) : 60000;
  x->reduce_ranks = ranks;
  x->reduce_pair = pair;
  x->reduce_comm = comm;
  return PLL_SUCCESS;
}

double pll_gpu_edge_loglikelihood_allreduce(pll_partition_t *p, void *comm, unsigned int parent_clv_index,
                                            int parent_scaler_index, unsigned int child_clv_index, int child_scaler_index,
                                            unsigned int matrix_index, const unsigned int *freqs_indices)
{
  /* first call with this communicator: the set-up (a failure here is before this rank's first collective) */
  pll_amd_ext_t *x = p ? pll_ext(p) : NULL;
  if ((!x || x->reduce_comm != comm || !x->reduce_pair) && !pll_gpu_allreduce_prepare(p, comm)) return -INFINITY;
  x = pll_ext(p);
  double *pair = x->reduce_pair;
  const int ranks = x->reduce_ranks;
  /* the ranks number their collective evaluations in step: the reduced sequence word is ranks x step */
  x->reduce_step += 1.0;
  /* From here on the rank always reaches the all-reduce. A rank whose evaluation failed takes part as in
   * pll_gpu_group_edge_loglikelihood: its operand is -inf and every rank returns -inf. (The one exception is a device
   * that cannot launch the one-lane kernel that writes that operand: then no collective can be enqueued on it either.) */
  const int mine_ok = pll_gpu_edge_loglikelihood_numbered(p, parent_clv_index, parent_scaler_index, child_clv_index,
                                                          child_scaler_index, matrix_index, freqs_indices, pair, x->reduce_step);
  const int my_errno = pll_errno;
  char my_errmsg[sizeof pll_errmsg];
  memcpy(my_errmsg, pll_errmsg, sizeof my_errmsg);
  if (!mine_ok && pllgpu_reduce_poison(x->ctx, x->reduce_step) != 0) return -INFINITY; /* (the device is gone: nothing to send) */
  if (!pll_gpu_allreduce_lnl(p, comm, pair, 2)) return -INFINITY;
  double sum = -INFINITY;
  if (pllgpu_reduce_fetch(x->ctx, x->reduce_step * ranks, &sum, x->reduce_timeout_ms) != 0)
  {
    pll_set_gpu_error("pll_gpu_edge_loglikelihood_allreduce");
    return -INFINITY;
  }
  if (!mine_ok)
  {
    pll_errno = my_errno;
    memcpy(pll_errmsg, my_errmsg, sizeof my_errmsg);
    return -INFINITY;
  }
  if (!isfinite(sum))
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_edge_loglikelihood_allreduce: another rank's evaluation failed");
    return -INFINITY;
  }
  return sum;
}
