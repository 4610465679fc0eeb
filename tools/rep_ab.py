"""A/B of the class-map path (kernels_repeats.h) on bench.py's C4 alignment or a mutated one (deeper compression):
what a traversal with update_repeats = 1 costs over one that re-uses the maps, per setting of the PLL_AMD_REP_*
switches (read at partition creation). Usage:
    python tools/rep_ab.py <sites> <bench|mutated> "ENV=V ENV2=V" "ENV=W" ...   ('' = the defaults)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
sys.path.insert(0, ROOT)
from pllamd import api, driver, workload as W  # noqa: E402


def main():
    sites, kind = int(sys.argv[1]), sys.argv[2]
    settings = sys.argv[3:] or [""]
    if kind == "bench":
        case = W.make_case("bench", 4, 128, sites, attributes=api.SITE_REPEATS, generator="xorshift64")
    else:
        case = W.make_case("c4", 4, 128, sites, attributes=api.SITE_REPEATS, mutate_pct=4, seed=4)
    ops = api.make_ops(case.op_batches[0])
    n = len(case.op_batches[0])
    lib = api.PllLib(os.environ.get("PLL_AMD_LIB", os.path.join(ROOT, "libpll-2_amd", "csrc", "libpll_amd.so")))
    keys = set()
    for st in settings:
        for kv in st.split():
            keys.add(kv.split("=")[0])
    for st in settings:
        for k in keys:
            os.environ.pop(k, None)
        for kv in st.split():
            k, v = kv.split("=")
            os.environ[k] = v
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            lib.pll_update_partials(s.p, ops, n)
            lib.pll_gpu_synchronize(s.p)
            res = {}
            for upd in (0, 1):
                best = []
                for _ in range(7):
                    t0 = time.perf_counter()
                    for _ in range(5):
                        if upd:  # round 6: an unchanged tree computes no maps - every map's inputs are forgotten first
                            lib.pll_gpu_invalidate(s.p, api.FORGET_REPEATS, -1)
                        lib.pll_update_partials_rep(s.p, ops, n, upd)
                    lib.pll_gpu_synchronize(s.p)
                    best.append((time.perf_counter() - t0) / 5 * 1e3)
                res[upd] = sorted(best)[len(best) // 2]
            lnl = s.edge_lnl(case.edges[0], persite=False)[0]
        print("%-40s maps reused %.3f ms  with class maps %.3f ms  class maps cost %.3f ms  lnL %.6f" % (st or "(defaults)", res[0], res[1], res[1] - res[0], lnl), flush=True)


if __name__ == "__main__":
    main()
