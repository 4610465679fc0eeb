"""Class maps under repetition (round 5): N traversals with update_repeats = 1 on one partition - every count of every
node, the site -> class and class -> site maps of a sample of nodes, and the log-likelihood must be the same bits every
time (the tickets, the fence-free hand-off of the copies and of the counts, the bitmap ranges: kernels_repeats.h).
    python tools/rep_stress.py [iterations] [sites] [bench|mutated]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
sys.path.insert(0, ROOT)
from pllamd import api, driver, workload as W  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    sites = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
    kind = sys.argv[3] if len(sys.argv) > 3 else "bench"
    if kind == "bench":
        case = W.make_case("bench", 4, 128, sites, attributes=api.SITE_REPEATS, generator="xorshift64")
    else:
        case = W.make_case("c4", 4, 128, sites, attributes=api.SITE_REPEATS, mutate_pct=4, seed=4)
    ops = api.make_ops(case.op_batches[0])
    n = len(case.op_batches[0])
    lib = api.PllLib(os.environ.get("PLL_AMD_LIB", os.path.join(ROOT, "libpll-2_amd", "csrc", "libpll_amd.so")))
    seen = {}
    t0 = time.time()
    with driver.Session(lib, case, api.ARCH_AVX2) as s:
        rep = s.part.repeats.contents
        sample = [case.tips + i for i in (0, 1, 40, 70, 100, 111)]  # nodes of levels 0 .. 2 (and above: uncompressed)
        for it in range(iters):
            lib.pll_gpu_invalidate(s.p, api.FORGET_REPEATS, -1)  # round 6: every map computed again (an unchanged tree computes none)
            lib.pll_update_partials(s.p, ops, n)
            lnl = s.edge_lnl(case.edges[0], persite=False)[0]
            h = hashlib.sha256()
            h.update(np.array([rep.pernode_ids[i] for i in range(s.part.nodes)], dtype=np.uint32).tobytes())
            if it % 16 == 0:  # the maps themselves (a download each): every 16th round
                for node in sample:
                    ids = rep.pernode_ids[node]
                    if ids:
                        h.update(api.as_np(lib.pll_get_site_id(s.p, node), case.sites, np.uint32).tobytes())
                        h.update(api.as_np(lib.pll_get_id_site(s.p, node), ids, np.uint32).tobytes())
            key = ("maps" if it % 16 == 0 else "counts", h.hexdigest(), lnl)
            seen[key] = seen.get(key, 0) + 1
    kinds = {}
    for (what, dig, lnl), cnt in seen.items():
        kinds.setdefault(what, []).append((cnt, dig[:16], lnl))
    print("%s alignment, %d sites, %d traversals with class maps in %.1f s" % (kind, sites, iters, time.time() - t0))
    for what, rows in kinds.items():
        print("  %-6s distinct outcomes: %d  %s" % (what, len(rows), rows))
    ok = all(len(rows) == 1 for rows in kinds.values())
    print("DETERMINISTIC" if ok else "DIFFERENT OUTCOMES")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
