"""tree search (tests/test_gpu_tree_search.py) of one configuration under sets of PLL_AMD_* switches: which step is the
first whose log-likelihood differs from the reference's, and what the call was. Usage:
  python tools/tree_search_debug.py <attrs> <states> <tips> <sites> <seed> <moves> [SWITCH=VALUE,SWITCH=VALUE ...]..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "libpll-2_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from pllamd import api  # noqa: E402
import test_gpu_tree_search as T  # noqa: E402


class Stop(Exception):
    pass


def main():
    attrs, states, tips, sites, seed, moves = sys.argv[1], *map(int, sys.argv[2:7])
    sets = sys.argv[7:] or [""]
    amd = api.PllLib()
    ref = api.PllLib(os.path.join(ROOT, "oracle", "_ref", "libpll_ref.so"))
    for sw in sets:
        pairs = [kv.split("=") for kv in sw.split(",") if kv]
        for k, v in pairs:
            os.environ[k] = v
        seen = []

        def check(step, what, vals, ctx):
            if what in ("maps", "derivatives"):
                return
            a, r = vals
            rel = abs(a - r) / abs(r)
            seen.append((step, what, ctx["ops"], rel))
            if not rel <= 1e-10:
                raise Stop()
        try:
            T._search([amd, ref], states, tips, sites, T.ATTRS[attrs], seed, moves, check=check)
            print(f"[{sw or 'default'}] all {len(seen)} evaluations within 1e-10 (worst {max(x[3] for x in seen):.2e})", flush=True)
        except Stop:
            print(f"[{sw or 'default'}] FIRST MISMATCH at {seen[-1]}; before it: {seen[-4:-1]}", flush=True)
        for k, _ in pairs:
            del os.environ[k]


if __name__ == "__main__":
    main()
