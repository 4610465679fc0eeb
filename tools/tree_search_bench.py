"""What a topology move costs (round 6; profiles/r6_tree_search.json): the calling pattern of a tree search - thousands
of short calls, 1-20 operations each - on the shapes of BASELINE configs[1] (C2: 64 taxa x 100k sites,
DNA) and of a configs[3] shard (128 taxa x 125k sites, PLL_ATTRIB_SITE_REPEATS), libpll_amd.so next to the reference's
AVX2 build on ONE host core, the same random NNI / SPR sequence for both.

A move = pll_update_prob_matrices for the branches it changed + the partial traversal's pll_update_partials (the
reference's default call: class maps of the touched nodes recomputed) + pll_compute_edge_loglikelihood at the moved
edge. Timed per move around exactly these three library calls (ctypes on both sides); the tree bookkeeping (tests/utree.py)
is outside. Usage: python tools/tree_search_bench.py [--moves 300] [--out file.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "libpll-2_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from pllamd import api, workload as W  # noqa: E402
from test_gpu_tree_search import Driven  # noqa: E402
from utree import UTree, random_move  # noqa: E402

SHAPES = {
    "c2_shape": dict(tips=64, sites=100000, attrs=0, note="64 taxa x 100k sites, 4 states x 4 rates (BASELINE configs[1]'s shape, random topology)"),
    "c4_shard_shape": dict(tips=128, sites=125000, attrs=api.SITE_REPEATS,
                           note="128 taxa x 125k sites, SITE_REPEATS (the shape of one of the eight shards of configs[3]; SURVEY 8d alignment, random topology)"),
}


def run(lib, shape, moves, seed, radius):
    rng = np.random.Generator(np.random.PCG64(seed))
    tree = UTree(shape["tips"], rng)
    st = W.section8d_states(shape["tips"], shape["sites"], 4)
    seqs = W.states_to_sequences(st, W.NT_CHARS)
    d = Driven(lib, tree, 4, shape["sites"], shape["attrs"], seqs, W.map_nt(), W.GTR_DNA["exch"], W.GTR_DNA["freqs"], W.gamma_rates_mean(0.5, 4))
    sync = (lambda: lib.pll_gpu_synchronize(d.p)) if lib.is_amd else (lambda: None)
    per_move, nops, lnls, phases = [], [], [], []
    try:
        rec = tree.inner_edges()[0]
        d.update(tree.ops_for(rec))
        d.lnl(tree.edge_args(rec))
        sync()
        for step in range(moves):
            rec, changed = random_move(tree, rng, radius=radius)
            ops = tree.ops_for(rec)
            arr = api.make_ops(ops)
            edge = tree.edge_args(rec)
            idx = np.ascontiguousarray([m for m, _ in changed], dtype=np.uint32)
            bl = np.ascontiguousarray([x for _, x in changed], dtype=np.float64)
            t0 = time.perf_counter()
            if changed:
                lib.pll_update_prob_matrices(d.p, api.uptr(d.params), api.uptr(idx), api.dptr(bl), len(changed))
            t1 = time.perf_counter()
            lib.pll_update_partials(d.p, arr, len(ops))
            t2 = time.perf_counter()
            v = lib.pll_compute_edge_loglikelihood(d.p, edge[0], edge[1], edge[2], edge[3], edge[4], api.uptr(d.params), None)
            t3 = time.perf_counter()
            per_move.append((t3 - t0) * 1e6)
            phases.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
            nops.append(len(ops))
            lnls.append(v)
    finally:
        d.close()
    a = np.array(per_move)
    return dict(us_per_move_median=round(float(np.median(a)), 1), us_per_move_mean=round(float(a.mean()), 1),
                us_per_move_p90=round(float(np.quantile(a, 0.9)), 1), ops_per_move_mean=round(float(np.mean(nops)), 2),
                ops_per_move_max=int(max(nops)), moves=moves,
                us_median_by_call=dict(zip(("pll_update_prob_matrices", "pll_update_partials (returns when enqueued)", "pll_compute_edge_loglikelihood (waits)"),
                                           [round(float(x), 1) for x in np.median(np.array(phases), axis=0)]))), lnls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--moves", type=int, default=300)
    ap.add_argument("--radius", type=int, default=5)
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="", help="one shape")
    ap.add_argument("--no-ref", action="store_true")
    args = ap.parse_args()
    amd = api.PllLib()
    refp = os.path.join(ROOT, "oracle", "_ref", "libpll_ref.so")
    ref = api.PllLib(refp) if os.path.exists(refp) and not args.no_ref else None
    out = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "spr_radius": args.radius}
    for name, shape in SHAPES.items():
        if args.only and name != args.only:
            continue
        row = {"shape": shape["note"]}
        row["libpll_amd"], la = run(amd, shape, args.moves, 11, args.radius)
        if ref is not None:
            row["reference_avx2_one_core"], lr = run(ref, shape, args.moves, 11, args.radius)
            row["speedup_median"] = round(row["reference_avx2_one_core"]["us_per_move_median"] / row["libpll_amd"]["us_per_move_median"], 1)
            row["worst_lnl_rel_err"] = float(max(abs(a - b) / abs(b) for a, b in zip(la, lr)))
        out[name] = row
        print(name, json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
