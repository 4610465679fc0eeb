"""The flat seam in a loop: pll_core_update_partial_ii at 1k sites (4 states x 4 rates), N calls, mean time per call.
    python tools/seam_loop.py [calls] [sites]      (under rocprofv3 --hip-trace --stats: where a call's time goes)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "libpll-2_amd"), os.path.join(ROOT, "tests")]
import test_gpu_core_seam as T  # noqa: E402
from pllamd import api  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    lib = api.PllLib()
    f = lib.dll.pll_core_update_partial_ii
    f.restype = None
    f.argtypes = [C.c_uint] * 3 + [T.D, T.U, T.D, T.D, T.D, T.D, T.U, T.U, C.c_uint]
    rng = np.random.default_rng(5)
    sp = 4
    lm, _ = T.pmat(4, 4, sp, 0.1, 1)
    rm, _ = T.pmat(4, 4, sp, 0.23, 2)
    pc = T.aligned(np.zeros((n, 4, sp)))
    ps = np.zeros((n, 1), dtype=np.uint32)
    l = T.rand_clv(rng, n, 4, 4, sp)
    r = T.rand_clv(rng, n, 4, 4, sp)
    args = (4, n, 4, T.dp(pc), T.up(ps), T.dp(l), T.dp(r), T.dp(lm), T.dp(rm), None, None, api.ARCH_AVX2)  # (the pointers once: numpy's ctypes views cost a microsecond each)
    for _ in range(5):
        f(*args)
    t0 = time.perf_counter()
    for _ in range(calls):
        f(*args)
    print("pll_core_update_partial_ii, %d sites: %.1f us per call (%d calls), checksum %.17g" % (n, (time.perf_counter() - t0) / calls * 1e6, calls, float(pc.sum())))


if __name__ == "__main__":
    main()
