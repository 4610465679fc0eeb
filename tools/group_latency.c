/* group_latency.c - what the fixed-order exchange of a site-sharded run costs (pll_gpu_group_sum,
 * csrc/host/group.c): N processes on one host, each adds one double per step; prints the time per step.
 * No GPU involved: in the path the device has already written its result to host memory when the exchange
 * starts.   cc -O2 tools/group_latency.c -o /tmp/group_latency -ldl && /tmp/group_latency <libpll_amd.so> N [steps] */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

typedef void *(*join_fn)(const char *, unsigned, unsigned, int);
typedef int (*sum_fn)(void *, const double *, unsigned, double *);
typedef void (*leave_fn)(void *);

int main(int argc, char **argv)
{
  if (argc < 3) return 2;
  const unsigned n = (unsigned)atoi(argv[2]);
  const long steps = argc > 3 ? atol(argv[3]) : 200000;
  char name[64];
  snprintf(name, sizeof name, "/pllamd-lat-%d", (int)getpid());
  setenv("PLL_AMD_HOST_ONLY", "1", 1);
  for (unsigned r = 0; r < n; ++r)
  {
    if (fork() != 0) continue;
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); _exit(3); }
    join_fn join = (join_fn)dlsym(h, "pll_gpu_group_join");
    sum_fn sum = (sum_fn)dlsym(h, "pll_gpu_group_sum");
    leave_fn leave = (leave_fn)dlsym(h, "pll_gpu_group_leave");
    void *g = join(name, r, n, 20000);
    if (!g) _exit(4);
    double v = 1.0 + r, out = 0;
    for (long i = 0; i < 1000; ++i) sum(g, &v, 1, &out);
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (long i = 0; i < steps; ++i) sum(g, &v, 1, &out);
    clock_gettime(CLOCK_MONOTONIC, &b);
    const double us = ((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec)) / 1e3 / steps;
    if (r == 0) printf("{\"ranks\": %u, \"steps\": %ld, \"us_per_exchange\": %.3f, \"sum\": %.1f}\n", n, steps, us, out);
    fflush(stdout);
    leave(g);
    _exit(0);
  }
  int bad = 0, st;
  while (wait(&st) > 0) bad |= st;
  return bad != 0;
}
