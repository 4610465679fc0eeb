"""One rank of a pll_gpu_group_* test (tests/test_group_exchange.py): joins the named segment, runs `steps`
fixed-order sums of values whose order of addition matters, prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))

import numpy as np  # noqa: E402

from pllamd import api  # noqa: E402


def values(rank, step, count):
    """magnitudes spread over 30 decades with signs that cancel: any other order of addition gives other bits"""
    rng = np.random.Generator(np.random.PCG64(1000 * step + rank))
    return rng.standard_normal(count) * 10.0 ** rng.integers(-15, 15, count)


def main():
    name, rank, size, steps, count, timeout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    os.environ["PLL_AMD_HOST_ONLY"] = "1"
    lib = api.PllLib()
    g = lib.pll_gpu_group_join(name.encode(), rank, size, timeout)
    if not g:
        print(json.dumps({"rank": rank, "error": lib.errmsg(), "errno": lib.errno()}))
        return
    out = np.zeros(count)
    sums = []
    t0 = time.perf_counter()
    for step in range(1, steps + 1):
        v = np.ascontiguousarray(values(rank, step, count))
        if not lib.pll_gpu_group_sum(g, api.dptr(v), count, api.dptr(out)):
            print(json.dumps({"rank": rank, "error": lib.errmsg(), "errno": lib.errno(), "step": step}))
            return
        sums.append(out.copy())
    dt = time.perf_counter() - t0
    # the exchange alone (no value generation in between): what one step adds to the path
    v = np.ascontiguousarray(values(rank, 0, 1))
    t0 = time.perf_counter()
    for _ in range(2000):
        lib.pll_gpu_group_sum(g, api.dptr(v), 1, api.dptr(out))
    us = (time.perf_counter() - t0) / 2000 * 1e6
    lib.pll_gpu_group_leave(g)
    print(json.dumps({"rank": rank, "sums": [[float.hex(float(x)) for x in s] for s in sums], "loop_s": dt, "exchange_us": us}))


if __name__ == "__main__":
    main()
