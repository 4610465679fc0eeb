/* TEST INFRASTRUCTURE - a stand-in for the three RCCL entry points the product binds with dlopen()
 * (csrc/host/group.c: ncclAllReduce, ncclCommCount, ncclGetErrorString; PLL_AMD_RCCL_LIB names this file's .so).
 *
 * Why: the library's collective evaluation has rank-count logic - the reduced sequence word is ranks x step, a rank
 * whose evaluation failed sends -inf, every rank must return the same bits - that a one-rank communicator does not
 * exercise, and real RCCL refuses two ranks on the one device a test box has. This double keeps RCCL's CONTRACT towards
 * the library - asynchronous, ordered on the caller's stream, operand and result in device memory, sum over the ranks
 * of the communicator - and replaces the transport by a shared-memory hand-off between the forked ranks:
 *
 *     D2H copy of the operand (stream)  ->  host function on the stream: slots added in rank order  ->  H2D copy
 *
 * A peer that never calls leaves the host function waiting, i.e. the stream blocked behind the collective - what a real
 * all-reduce does - until comm->timeout_ms, after which everybody gets NaN. Not a product path; nothing under
 * libpll-2_amd/ knows it exists.
 *
 *     gcc -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/c_caller/rccl_double.c -L/opt/rocm/lib -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rccl_double.h"

static double now_ms(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

typedef struct pending
{
  rccl_double_comm_t *comm;
  unsigned int count;
  unsigned long long step;
} pending_t;

/* runs on the stream, after the D2H copy of the operand and before the H2D copy of the result */
static void exchange(void *arg)
{
  pending_t *job = (pending_t *)arg;
  rccl_double_comm_t *c = job->comm;
  const unsigned int par = (unsigned int)(job->step & 1u);
  const double *in = c->pinned;
  double *out = c->pinned + RCCL_DOUBLE_MAX_COUNT;
  rccl_double_slot_t *mine = &c->shared->slot[c->rank][par];
  for (unsigned int i = 0; i < job->count; ++i) mine->v[i] = in[i];
  mine->count = job->count;
  __atomic_store_n(&mine->step, job->step, __ATOMIC_RELEASE);
  const double t0 = now_ms();
  int missing = 0;
  double acc[RCCL_DOUBLE_MAX_COUNT] = {0};
  for (int r = 0; r < c->size && !missing; ++r)
  {
    const rccl_double_slot_t *s = &c->shared->slot[r][par];
    while (__atomic_load_n(&s->step, __ATOMIC_ACQUIRE) < job->step)
      if (now_ms() - t0 > c->timeout_ms)
      {
        missing = 1;
        break;
      }
    for (unsigned int i = 0; i < job->count && !missing; ++i) acc[i] = r ? acc[i] + s->v[i] : s->v[i];
  }
  for (unsigned int i = 0; i < job->count; ++i) out[i] = missing ? NAN : acc[i];
  free(job);
}

int ncclCommCount(const void *comm, int *count)
{
  const rccl_double_comm_t *c = (const rccl_double_comm_t *)comm;
  if (!c || c->magic != RCCL_DOUBLE_MAGIC || !count) return 4; /* ncclInvalidArgument */
  *count = c->size;
  return 0;
}

int ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, int datatype, int op, void *comm, void *stream)
{
  rccl_double_comm_t *c = (rccl_double_comm_t *)comm;
  if (!c || c->magic != RCCL_DOUBLE_MAGIC || datatype != 8 /* ncclFloat64 */ || op != 0 /* ncclSum */ || count == 0 ||
      count > RCCL_DOUBLE_MAX_COUNT || c->size < 1 || c->size > RCCL_DOUBLE_MAX_RANKS)
    return 4;
  if (!c->pinned && hipHostMalloc((void **)&c->pinned, 2 * RCCL_DOUBLE_MAX_COUNT * sizeof(double), hipHostMallocDefault) != hipSuccess) return 1;
  pending_t *job = (pending_t *)malloc(sizeof *job);
  if (!job) return 2;
  job->comm = c;
  job->count = (unsigned int)count;
  job->step = ++c->step;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyAsync(c->pinned, sendbuff, count * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipLaunchHostFunc(s, exchange, job) != hipSuccess ||
      hipMemcpyAsync(recvbuff, c->pinned + RCCL_DOUBLE_MAX_COUNT, count * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess)
    return 1; /* ncclUnhandledCudaError */
  return 0;
}

const char *ncclGetErrorString(int code)
{
  switch (code)
  {
    case 0: return "no error";
    case 1: return "unhandled HIP error (rccl double)";
    case 2: return "out of memory (rccl double)";
    case 4: return "invalid argument (rccl double)";
    default: return "error (rccl double)";
  }
}
