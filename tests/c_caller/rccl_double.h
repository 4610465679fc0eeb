/* TEST INFRASTRUCTURE: the communicator of tests/c_caller/rccl_double.c - a stand-in for librccl that lets the
 * MULTI-RANK logic of pll_gpu_edge_loglikelihood_allreduce (csrc/host/group.c) run on a one-GPU box, where real RCCL
 * refuses two ranks on one device. The caller (tests/c_caller/sharded.c) maps `rccl_double_shared_t` before it forks
 * its ranks and hands every rank's own `rccl_double_comm_t` to the library as the opaque ncclComm_t. */
#ifndef RCCL_DOUBLE_H
#define RCCL_DOUBLE_H

#define RCCL_DOUBLE_MAGIC 0x52434c44u /* "RCLD" */
#define RCCL_DOUBLE_MAX_RANKS 8
#define RCCL_DOUBLE_MAX_COUNT 6

typedef struct rccl_double_slot
{
  _Alignas(64) volatile unsigned long long step;
  double v[RCCL_DOUBLE_MAX_COUNT];
  unsigned int count;
} rccl_double_slot_t;

typedef struct rccl_double_shared
{
  rccl_double_slot_t slot[RCCL_DOUBLE_MAX_RANKS][2];
} rccl_double_shared_t;

typedef struct rccl_double_comm
{
  unsigned int magic;
  int rank, size;
  rccl_double_shared_t *shared;
  int timeout_ms;          /* how long the stream-ordered exchange waits for a peer before it gives NaN to everyone */
  unsigned long long step; /* collectives issued by this rank */
  double *pinned;          /* [2][RCCL_DOUBLE_MAX_COUNT] host staging, allocated at the first collective */
} rccl_double_comm_t;

#endif
