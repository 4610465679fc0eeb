"""Where does run-to-run variation come from? N repetitions of traversal + log-likelihood (+ derivatives) on one
partition: distinct values with counts, and whether the two root-side CLVs ever differ between repetitions."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd")); sys.path.insert(0, ROOT)
import numpy as np
from pllamd import api, driver, workload as W
states, taxa, sites, reps = (int(x) for x in sys.argv[1:5])
lib = api.PllLib()
case = W.make_case("det", states, taxa, sites, seed=5)
e = case.edges[0]
with driver.Session(lib, case, api.ARCH_AVX2) as s:
    s.set_model(case.model["exch"], case.freqs, case.model["rates"])
    s.update_partials()
    ref_clv = [s.read_clv(e[0]), s.read_clv(e[2])]
    vals, dvals, clv_diff = collections.Counter(), collections.Counter(), 0
    st = s.new_sumtable()
    for i in range(reps):
        s.update_partials()
        v, _ = s.edge_lnl(e, persite=False)
        vals[v] += 1
        if i % 4 == 0:
            s.update_sumtable(e, st)
            dvals[s.derivatives(e, st, 0.1)] += 1
        if i % 16 == 0:
            for a, b in zip(ref_clv, [s.read_clv(e[0]), s.read_clv(e[2])]):
                if not np.array_equal(a, b): clv_diff += 1
    print("lnL values:", vals.most_common(5))
    print("derivative values:", dvals.most_common(4))
    print("root CLV reads that differed:", clv_diff)
