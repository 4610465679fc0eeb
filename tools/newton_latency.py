"""Latency of the branch-length optimisation calls: pll_update_sumtable,
pll_compute_likelihood_derivatives, pll_update_prob_matrices + 1-op update + edge lnL - what a
Newton-Raphson step of a tree search issues per branch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
import numpy as np  # noqa: E402
from pllamd import api, driver, workload as W  # noqa: E402

lib = api.PllLib()
for states, sites in ((4, 1000), (4, 100000), (20, 10000), (61, 2000)):
    case = W.make_case("nr", states, 16, sites, seed=2)
    e = case.edges[0]
    with driver.Session(lib, case, api.ARCH_AVX2) as s:
        s.set_model(case.model["exch"], case.freqs, case.model["rates"])
        pi = np.zeros(case.rate_cats, dtype=np.uint32)
        mi = np.arange(case.prob_matrices, dtype=np.uint32)
        brl = np.ascontiguousarray(W.branch_lengths(case.prob_matrices))
        assert lib.pll_update_prob_matrices(s.p, api.uptr(pi), api.uptr(mi), api.dptr(brl), len(mi))
        s.update_partials()
        s.edge_lnl(e, persite=False)
        st = s.new_sumtable()
        for _ in range(5):
            s.update_sumtable(e, st)
            s.derivatives(e, st, 0.1)
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            s.update_sumtable(e, st)
        lib.pll_gpu_synchronize(s.p)
        t1 = time.perf_counter()
        for i in range(n):
            s.derivatives(e, st, 0.05 + 1e-4 * i)
        t2 = time.perf_counter()
        one = np.array([e[4]], dtype=np.uint32)
        last = api.make_ops(case.op_batches[0][-1:])
        for i in range(n):
            bl = np.array([0.05 + 1e-4 * i])
            lib.pll_update_prob_matrices(s.p, api.uptr(pi), api.uptr(one), api.dptr(bl), 1)
            lib.pll_update_partials(s.p, last, 1)
            lib.pll_compute_edge_loglikelihood(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(pi), None)
        t3 = time.perf_counter()
        print(f"states={states} sites={sites}: sumtable {1e6*(t1-t0)/n:6.1f} us (async), derivatives {1e6*(t2-t1)/n:6.1f} us, "
              f"pmatrix + 1-op update + lnL {1e6*(t3-t2)/n:6.1f} us")
