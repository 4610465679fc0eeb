"""The result hand-off of the reduction kernels (partial sums through agent-scope atomics, no
fences - kernels_common.h) under load: thousands of traversal + log-likelihood / derivative calls on
large and small inputs, every value must be bit-identical to the first one."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from pllamd import api, driver, workload as W  # noqa: E402

lib = api.PllLib()
bad = 0
for states, taxa, sites, tree, reps in ((4, 64, 100000, "balanced", 3000), (4, 64, 100000, "random", 2000), (4, 32, 250000, "caterpillar", 500),
                                        (4, 16, 1000, "balanced", 5000), (20, 16, 20000, "balanced", 500), (4, 64, 100000, "nochains", 1500),
                                        # 61 states: k_edge_mfma's per-rate partials and the ticket per item block (round 2)
                                        (61, 32, 20000, "balanced", 1500), (61, 8, 3000, "balanced", 3000), (40, 8, 70000, "balanced", 300)):
    if tree == "nochains":
        os.environ["PLL_AMD_NO_CHAINS"] = "1"
        tree = "balanced"
    case = W.make_case("stress", states, taxa, sites, tree=tree, seed=5)
    e = case.edges[0]
    with driver.Session(lib, case, api.ARCH_AVX2) as s:
        s.set_model(case.model["exch"], case.freqs, case.model["rates"])
        s.update_partials()
        ref, _ = s.edge_lnl(e, persite=False)
        st = s.new_sumtable()
        s.update_sumtable(e, st)
        dref = s.derivatives(e, st, 0.1)
        for i in range(reps):
            s.update_partials()
            v, _ = s.edge_lnl(e, persite=False)
            if v != ref:
                bad += 1
            if i % 8 == 0:
                s.update_sumtable(e, st)
                d = s.derivatives(e, st, 0.1)
                if d != dref:
                    bad += 1
        print(states, taxa, sites, tree, reps, "lnL", ref, "mismatches so far", bad, flush=True)
sys.exit(1 if bad else 0)
