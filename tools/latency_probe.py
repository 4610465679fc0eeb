#!/usr/bin/env python3
"""Latency of short hot-path calls (partial traversals of 1-3 ops + edge lnL), the pattern tree-search
applications issue thousands of times per second (SURVEY 8b "what calls the boundary")."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
import numpy as np
from pllamd import api, driver, workload as W

lib = api.PllLib()
for states, sites in ((4, 1000), (4, 100000), (20, 1000), (20, 50000)):
    case = W.make_case("lat", states, 64, sites, seed=1)
    ops = case.op_batches[0]
    with driver.Session(lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        s.edge_lnl(case.edges[0], persite=False)
        for nops in (1, 3):
            sub = api.make_ops(ops[-nops:])
            fi = np.zeros(4, dtype=np.uint32)
            e = case.edges[0]
            # warm
            for _ in range(20):
                lib.pll_update_partials(s.p, sub, nops)
                lib.pll_compute_edge_loglikelihood(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(fi), None)
            n = 500
            t0 = time.perf_counter()
            for _ in range(n):
                lib.pll_update_partials(s.p, sub, nops)
            lib.pll_gpu_synchronize(s.p)
            t1 = time.perf_counter()
            for _ in range(n):
                lib.pll_update_partials(s.p, sub, nops)
                lib.pll_compute_edge_loglikelihood(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(fi), None)
            t2 = time.perf_counter()
            for _ in range(n):
                lib.pll_compute_edge_loglikelihood(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(fi), None)
            t3 = time.perf_counter()
            print(f"states={states} sites={sites} ops={nops}: update_partials {1e6*(t1-t0)/n:7.1f} us/call (async, amortised), "
                  f"update+lnl {1e6*(t2-t1)/n:7.1f} us, lnl alone {1e6*(t3-t2)/n:7.1f} us")

# ---- derivatives: sumtable once, then a Newton-style series of evaluations ----------------------
print()
for states, sites in ((4, 1000), (4, 100000), (20, 50000)):
    case = W.make_case("lat", states, 64, sites, seed=1)
    eig = W.eigensystem(case.model["exch"], case.freqs[0])
    e = case.edges[0]
    edge = (e[0], e[1], e[2], e[3])
    with driver.Session(lib, case, api.ARCH_AVX2) as s:
        s.inject_eigen(eig, case.model["rates"])
        s.update_partials()
        st = s.new_sumtable()
        for _ in range(5):
            s.update_sumtable(edge, st)
            s.derivatives(edge, st, 0.1)
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            s.update_sumtable(edge, st)
        lib.pll_gpu_synchronize(s.p)
        t1 = time.perf_counter()
        for i in range(n):
            s.derivatives(edge, st, 0.05 + 0.001 * i)
        t2 = time.perf_counter()
        print(f"states={states} sites={sites}: pll_update_sumtable {1e6*(t1-t0)/n:7.1f} us/call, "
              f"pll_compute_likelihood_derivatives {1e6*(t2-t1)/n:7.1f} us/call "
              f"({sites*states*4*8/((t2-t1)/n)/1e9:.0f} GB/s of table)")
