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
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "pll_internal.h"

#define GROUP_MAGIC 0x504c4c4752503031ull /* "PLLGRP01" */
#define GROUP_MAX_VALUES 6u

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
  volatile unsigned int joined; /* ranks that have cleared their slots */
  volatile unsigned int left;   /* ranks that have gone: the last one unlinks the name */
} group_header_t;

struct pll_gpu_group
{
  unsigned int rank, size;
  unsigned long long step;
  group_header_t *hdr;
  group_slot_t *slots; /* [size][2] */
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

pll_gpu_group_t *pll_gpu_group_join(const char *name, unsigned int rank, unsigned int size, int timeout_ms)
{
  if (!name || name[0] != '/' || strlen(name) >= sizeof(((pll_gpu_group_t *)0)->name) || size == 0 || rank >= size || size > 4096)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_group_join: name must start with '/', rank < size <= 4096");
    return NULL;
  }
  if (timeout_ms <= 0) timeout_ms = 60000;
  const size_t bytes = sizeof(group_header_t) + (size_t)size * 2 * sizeof(group_slot_t);
  /* whoever comes first creates the segment (zero-filled by the kernel) and sizes it; the others find it */
  int fd = shm_open(name, O_RDWR | O_CREAT, 0600);
  if (fd < 0)
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_join: shm_open(%s): %s", name, strerror(errno));
    return NULL;
  }
  struct stat st;
  if (fstat(fd, &st) != 0 || ((size_t)st.st_size < bytes && ftruncate(fd, (off_t)bytes) != 0))
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_join: sizing %s: %s", name, strerror(errno));
    close(fd);
    return NULL;
  }
  void *mem = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (mem == MAP_FAILED)
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_join: mmap(%s): %s", name, strerror(errno));
    return NULL;
  }
  pll_gpu_group_t *g = (pll_gpu_group_t *)calloc(1, sizeof *g);
  if (!g)
  {
    munmap(mem, bytes);
    pll_set_error(PLL_ERROR_MEM_ALLOC, "pll_gpu_group_join: out of memory");
    return NULL;
  }
  g->rank = rank;
  g->size = size;
  g->hdr = (group_header_t *)mem;
  g->slots = (group_slot_t *)((char *)mem + sizeof(group_header_t));
  g->bytes = bytes;
  g->timeout_ms = timeout_ms;
  strcpy(g->name, name);
  /* a segment left behind by a run of another size under the same name is refused, not reused */
  unsigned int expect = 0;
  if (!__atomic_compare_exchange_n(&g->hdr->size, &expect, size, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE) && expect != size)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_group_join: %s exists with %u ranks, not %u (names must be unique per run)", name, expect, size);
    munmap(mem, bytes);
    free(g);
    return NULL;
  }
  memset((void *)&g->slots[2 * rank], 0, 2 * sizeof(group_slot_t));
  __atomic_store_n(&g->hdr->magic, GROUP_MAGIC, __ATOMIC_RELEASE);
  __atomic_fetch_add(&g->hdr->joined, 1u, __ATOMIC_ACQ_REL);
  const double t0 = now_ms();
  while (__atomic_load_n(&g->hdr->joined, __ATOMIC_ACQUIRE) < size)
  {
    cpu_relax();
    if (now_ms() - t0 > timeout_ms)
    {
      pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_join: %u of %u ranks joined %s within %d ms", g->hdr->joined, size, name, timeout_ms);
      pll_gpu_group_leave(g);
      return NULL;
    }
  }
  if (__atomic_load_n(&g->hdr->joined, __ATOMIC_ACQUIRE) > size)
  {
    pll_set_error(PLL_ERROR_PARAM_INVALID, "pll_gpu_group_join: more than %u ranks joined %s (a stale segment? names must be unique per run)", size, name);
    pll_gpu_group_leave(g);
    return NULL;
  }
  return g;
}

void pll_gpu_group_leave(pll_gpu_group_t *g)
{
  if (!g) return;
  /* the last rank to leave removes the name; a crashed run leaves it behind (hence: unique names) */
  if (__atomic_add_fetch(&g->hdr->left, 1u, __ATOMIC_ACQ_REL) >= g->size) shm_unlink(g->name);
  munmap((void *)g->hdr, g->bytes);
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
        if (t - t0 > g->timeout_ms)
        {
          pll_set_error(PLL_ERROR_GPU_RUNTIME, "pll_gpu_group_sum: rank %u did not reach step %llu within %d ms", r, step, g->timeout_ms);
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
  const int rc = g_rccl.allreduce(device_values, device_values, count, NCCL_DOUBLE, NCCL_SUM, comm, stream);
  if (rc != 0)
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "ncclAllReduce: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "failed");
    return PLL_FAILURE;
  }
  return PLL_SUCCESS;
}

double pll_gpu_edge_loglikelihood_allreduce(pll_partition_t *p, void *comm, unsigned int parent_clv_index,
                                            int parent_scaler_index, unsigned int child_clv_index, int child_scaler_index,
                                            unsigned int matrix_index, const unsigned int *freqs_indices)
{
  pll_amd_ext_t *x = reduce_ctx(p, comm, "pll_gpu_edge_loglikelihood_allreduce");
  if (!x) return -INFINITY;
  int ranks = 0;
  if (g_rccl.count(comm, &ranks) != 0 || ranks < 1)
  {
    pll_set_error(PLL_ERROR_GPU_RUNTIME, "ncclCommCount failed");
    return -INFINITY;
  }
  double *pair = pllgpu_reduce_buffer(x->ctx);
  if (!pair)
  {
    pll_set_gpu_error("pll_gpu_edge_loglikelihood_allreduce");
    return -INFINITY;
  }
  /* the ranks number their collective evaluations in step: the reduced sequence word is ranks x step */
  x->reduce_step += 1.0;
  /* a rank whose evaluation failed still takes part, as in pll_gpu_group_edge_loglikelihood: its operand is -inf and
   * every rank returns -inf, nobody is left waiting inside the collective */
  const int mine_ok = pll_gpu_edge_loglikelihood_numbered(p, parent_clv_index, parent_scaler_index, child_clv_index,
                                                          child_scaler_index, matrix_index, freqs_indices, pair, x->reduce_step);
  const int my_errno = pll_errno;
  char my_errmsg[sizeof pll_errmsg];
  memcpy(my_errmsg, pll_errmsg, sizeof my_errmsg);
  if (!mine_ok && pllgpu_reduce_poison(x->ctx, x->reduce_step) != 0) return -INFINITY; /* (the device is gone: nothing to send) */
  if (!pll_gpu_allreduce_lnl(p, comm, pair, 2)) return -INFINITY;
  double sum = -INFINITY;
  if (pllgpu_reduce_fetch(x->ctx, x->reduce_step * ranks, &sum) != 0)
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
