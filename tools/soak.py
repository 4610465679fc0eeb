"""Soak: many partition lifecycles and hot-path calls in one process; device memory before / after
(leaks), results stable. python tools/soak.py [cycles]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from pllamd import api, driver, workload as W  # noqa: E402

hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")


def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value / 1e6


lib = api.PllLib()
cases = [W.make_case("soak", 4, 16, 5000, seed=1), W.make_case("soak", 4, 32, 3000, seed=2, attributes=api.SITE_REPEATS, mutate_pct=5),
         W.make_case("soak", 20, 8, 1500, seed=3), W.make_case("soak", 61, 8, 300, seed=4),
         W.make_case("soak", 4, 16, 2000, seed=5, asc_type=1)]
with driver.Session(lib, cases[0], api.ARCH_AVX2) as s:  # warm the runtime
    s.update_partials()
    s.edge_lnl(cases[0].edges[0])
start = free_mb()
ref = {}
t0 = time.perf_counter()
CYCLES = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for it in range(CYCLES):
    for ci, case in enumerate(cases):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            for _ in range(20):
                s.update_partials()
                v, _ = s.edge_lnl(case.edges[0], persite=False)
            st = s.new_sumtable()
            s.set_model(case.model["exch"], case.freqs, case.model["rates"])
            s.update_sumtable(case.edges[0], st)
            d = s.derivatives(case.edges[0], st, 0.2)
            assert ref.setdefault(ci, (v, d)) == (v, d), (ci, it)
    if it % 20 == 19:
        print(f"cycle {it + 1}: free {free_mb():.0f} MB (start {start:.0f}), {time.perf_counter() - t0:.1f} s", flush=True)
end = free_mb()
print("leak MB:", round(start - end, 1))
assert start - end < 64, "device memory is not coming back"
