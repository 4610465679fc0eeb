"""Many seeds of tests/test_gpu_tree_search.py's search (round 6): python tools/tree_search_soak.py [seeds] [moves]
Every configuration of the test matrix (64 taxa; every 4th seed 300 taxa) under fresh seeds against the reference; prints the
worst relative log-likelihood error per configuration and stops at the first mismatch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "libpll-2_amd"), ROOT, os.path.join(ROOT, "tests")]
from pllamd import api  # noqa: E402
import test_gpu_tree_search as T  # noqa: E402


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    moves = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    amd = api.PllLib()
    ref = api.PllLib(os.path.join(ROOT, "oracle", "_ref", "libpll_ref.so"))
    total = 0
    for seed in range(seeds):
        for attrs in T.ATTRS:
            for states, sites in ((4, 1500), (20, 400), (7, 600)):
                tips = 300 if seed % 4 == 3 else 64
                check, worst = T._against_reference(sites)
                T._search([amd, ref], states, tips, sites if tips == 64 else sites // 3, T.ATTRS[attrs], seed=100000 + 977 * seed + states, moves=moves, check=check,
                          spr_radius=3 + seed % 8, full_every=25 + seed % 7, tip_every=19 + seed % 5)
                total += worst["n"]
                print(f"seed {seed} {attrs} {states} states {tips} taxa: {worst['n']} evaluations, worst {worst['lnl']:.2e}", flush=True)
    print(f"OK: {total} evaluations against the reference")


if __name__ == "__main__":
    main()
