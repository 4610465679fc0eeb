"""What the per-site scaling exchange costs a chain: the 64-taxon ladder / a random tree (100k sites) with per-site scalers on
every inner node (one ballot exchange + workgroup barrier per chain step), with per-rate scalers (no exchange: a rate's
decision is its wave's own) and without scalers. ms per (traversal + log-likelihood)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
import numpy as np  # noqa: E402
from pllamd import api, driver, workload as W  # noqa: E402

lib = api.PllLib()
for tree in ("caterpillar", "random"):
    for label, kw in (("per-site scalers", dict()), ("per-rate scalers", dict(attributes=api.RATE_SCALERS)), ("no scalers", dict(scalers=False))):
        case = W.make_case("probe", 4, 64, 100000, tree=tree, seed=11, **kw)
        ops = api.make_ops(case.op_batches[0])
        n = len(case.op_batches[0])
        fi = np.zeros(4, dtype=np.uint32)
        e = case.edges[0]
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            for _ in range(10):
                lib.pll_update_partials(s.p, ops, n)
                lib.pll_compute_edge_loglikelihood(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(fi), None)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(50):
                    lib.pll_update_partials(s.p, ops, n)
                    lib.pll_compute_edge_loglikelihood(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(fi), None)
                best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
            print(f"{tree:12s} {label:18s} {best:.4f} ms  launches {lib.pll_gpu_last_launch_count(s.p)}")
