"""Host-side cost of one hot-path step (C2): time to ENQUEUE a traversal (pll_update_partials returns
before the kernels finish), the traversal's GPU time, and the blocking lnL call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from pllamd import api, driver, workload as W  # noqa: E402

lib = api.PllLib()
if len(sys.argv) > 1 and sys.argv[1] == "c4shard":  # shard 1 of 8 of the pattern-sorted 1M-site alignment, site repeats
    import bench
    from pllamd import sharding
    full = bench.build_case(bench.CONFIGS["c4"], 1000000, api.SITE_REPEATS)
    case = sharding.shard_case(sharding.sort_columns(lib, full), 1, 8)
else:
    case = W.make_case("c2", 4, 64, 100000, seed=1000)
ops = api.make_ops(case.op_batches[0])
n = len(case.op_batches[0])
fi = np.zeros(4, dtype=np.uint32)
e = case.edges[0]
with driver.Session(lib, case, api.ARCH_AVX2) as s:
    lib.pll_update_partials(s.p, ops, n)
    upd = lib.pll_update_partials
    if case.attributes & api.SITE_REPEATS:  # class maps once, then re-used
        upd = lambda p_, o_, n_: lib.pll_update_partials_rep(p_, o_, n_, 0)
    for _ in range(5):
        upd(s.p, ops, n)
        lib.pll_compute_edge_loglikelihood(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(fi), None)
    enq, tot, lnl = [], [], []
    for _ in range(200):
        lib.pll_gpu_synchronize(s.p)
        t0 = time.perf_counter()
        upd(s.p, ops, n)
        t1 = time.perf_counter()
        lib.pll_gpu_synchronize(s.p)
        t2 = time.perf_counter()
        lib.pll_compute_edge_loglikelihood(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(fi), None)
        t3 = time.perf_counter()
        enq.append(t1 - t0)
        tot.append(t2 - t0)
        lnl.append(t3 - t2)
    med = lambda v: sorted(v)[len(v) // 2] * 1e6
    print(dict(enqueue_us=round(med(enq), 1), traversal_wall_us=round(med(tot), 1), lnl_call_us=round(med(lnl), 1)))
    # back-to-back steps as bench.py runs them
    t0 = time.perf_counter()
    for _ in range(200):
        upd(s.p, ops, n)
        lib.pll_compute_edge_loglikelihood(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(fi), None)
    print(dict(step_us=round((time.perf_counter() - t0) / 200 * 1e6, 1)))
