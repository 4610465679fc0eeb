"""Time the site-repeats class computation (device, kernels_repeats.h) inside pll_update_partials:
a traversal with update_repeats = 1 against the same traversal reusing the maps, and the
reference's host walk (oracle/_ref) for the same op list. Usage: python tools/repeats_timing.py [sites]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
sys.path.insert(0, ROOT)
from pllamd import api, driver, workload as W  # noqa: E402


def main():
    sites = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
    case = W.make_case("c4", 4, 128, sites, attributes=api.SITE_REPEATS, mutate_pct=4, seed=4)
    ops = api.make_ops(case.op_batches[0])
    n = len(case.op_batches[0])
    out = {}
    libs = [("amd", api.PllLib(os.environ.get("PLL_AMD_LIB", os.path.join(ROOT, "libpll-2_amd", "csrc", "libpll_amd.so"))))]
    ref = os.path.join(ROOT, "oracle", "_ref", "libpll_ref.so")
    if os.path.exists(ref):
        libs.append(("ref", api.PllLib(ref)))
    for name, lib in libs:
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            sync = (lambda: lib.pll_gpu_synchronize(s.p)) if lib.is_amd else (lambda: None)
            lib.pll_update_partials(s.p, ops, n)
            sync()
            t = []
            for upd in (1, 0, 1, 0):
                if upd and lib.is_amd:  # round 6: an unchanged tree computes no maps unless their inputs are forgotten
                    lib.pll_gpu_invalidate(s.p, api.FORGET_REPEATS, -1)
                t0 = time.perf_counter()
                lib.pll_update_partials_rep(s.p, ops, n, upd)
                sync()
                t.append((upd, (time.perf_counter() - t0) * 1e3))
            rep = s.part.repeats.contents
            classes = sum(rep.pernode_ids[i] or sites for i in range(case.tips, s.part.nodes))
            out[name] = dict(ms=[(u, round(v, 3)) for u, v in t], class_entries=classes, of=sites * (s.part.nodes - case.tips))
    print(out)


if __name__ == "__main__":
    main()
