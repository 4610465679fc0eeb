"""Tree-search pattern: after a topology move the CLVs on a path towards the (virtual) root are
recomputed - k ops, each consuming the previous one - and the edge log-likelihood is asked for.
Wall time per (pll_update_partials(k ops) + pll_compute_edge_loglikelihood) on a 64-taxon ladder, the
op list changing from call to call (no plan is re-used), chains on and off."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
import numpy as np  # noqa: E402
from pllamd import api, driver, workload as W  # noqa: E402

lib = api.PllLib()
rows = []
for sites in (1000, 100000):
    case = W.make_case("path", 4, 64, sites, tree="caterpillar", seed=7)
    ops = case.op_batches[0]
    e = case.edges[0]
    fi = np.zeros(4, dtype=np.uint32)
    for chains in (True, False):
        if chains:
            os.environ.pop("PLL_AMD_NO_CHAINS", None)
        else:
            os.environ["PLL_AMD_NO_CHAINS"] = "1"
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            s.update_partials()
            ref, _ = s.edge_lnl(e, persite=False)
            for k in (1, 2, 3, 5, 8, 16):
                # two alternating lists of k ops: the last k, and the last k with the first one dropped + re-added
                a = api.make_ops(ops[-k:])
                b = api.make_ops(ops[-k - 1:]) if k < len(ops) else a
                for _ in range(5):
                    lib.pll_update_partials(s.p, a, k)
                    lib.pll_compute_edge_loglikelihood(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(fi), None)
                n = 200
                t0 = time.perf_counter()
                for i in range(n):
                    if i & 1:
                        lib.pll_update_partials(s.p, b, k + 1)
                    else:
                        lib.pll_update_partials(s.p, a, k)
                    v = lib.pll_compute_edge_loglikelihood(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(fi), None)
                dt = (time.perf_counter() - t0) / n * 1e6
                assert v == ref, (v, ref)
                rows.append((sites, "chains" if chains else "levels", k, round(dt, 1)))
for r in rows:
    print(*r)
