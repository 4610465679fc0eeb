"""Site repeats in a tree search: after a move a handful of ops along a path are recomputed with
update_repeats = 1 (their class maps change). Time per pll_update_partials_rep call for k ops with and
without the class-map update, device library against the reference's host walk (oracle/_ref).
python tools/repeats_path_timing.py [sites]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
sys.path.insert(0, ROOT)
from pllamd import api, driver, workload as W  # noqa: E402

sites = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
case = W.make_case("c4", 4, 128, sites, attributes=api.SITE_REPEATS, mutate_pct=4, seed=4)
allops = case.op_batches[0]
# a path: the last op and its chain of producers (one child each)
prod = {op[0]: op for op in allops}
path = [allops[-1]]
while path[-1][2] in prod or path[-1][5] in prod:
    nxt = prod.get(path[-1][2]) or prod.get(path[-1][5])
    path.append(nxt)
path = path[::-1]
libs = [("amd", api.PllLib())]
ref = os.path.join(ROOT, "oracle", "_ref", "libpll_ref.so")
if os.path.exists(ref):
    libs.append(("ref", api.PllLib(ref)))
for name, lib in libs:
    with driver.Session(lib, case, api.ARCH_AVX2) as s:
        sync = (lambda: lib.pll_gpu_synchronize(s.p)) if lib.is_amd else (lambda: None)
        full = api.make_ops(allops)
        lib.pll_update_partials(s.p, full, len(allops))
        sync()
        for k in (1, 2, 3, len(path)):
            ops = api.make_ops(path[-k:])
            row = []
            for upd in (1, 0):
                for _ in range(3):
                    lib.pll_update_partials_rep(s.p, ops, k, upd)
                sync()
                t0 = time.perf_counter()
                for _ in range(20):
                    if upd and lib.is_amd:  # round 6: the path's maps forgotten first (an unchanged tree computes none)
                        for o in path[-k:]:
                            lib.pll_gpu_invalidate(s.p, api.FORGET_REPEATS, int(o[0]))
                    lib.pll_update_partials_rep(s.p, ops, k, upd)
                sync()
                row.append((time.perf_counter() - t0) / 20 * 1e6)
            print(name, "sites", sites, "ops", k, "with class maps %.1f us" % row[0], "reusing them %.1f us" % row[1], flush=True)
