"""Round 6 evidence for the class-map stamps (profiles/r6_stamps_trace.txt): run under `rocprofv3 --kernel-trace --stats`.
One 128-taxon x 125k-site SITE_REPEATS partition (random topology, SURVEY 8d alignment):
  phase 1  the first full traversal (pll_update_partials: every class map computed)
  phase 2  50 x the same traversal again, the reference's default call, new branch lengths in between
  phase 3  20 x (a random NNI + pll_update_prob_matrices + the partial traversal's pll_update_partials + edge lnL)
The library's own counters (pll_gpu_class_map_work: map operations / class kernels that reached the device) are printed per
phase; the k_rep_* call counts of the rocprofv3 summary are their sum."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "libpll-2_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from pllamd import api, workload as W  # noqa: E402
from test_gpu_tree_search import Driven  # noqa: E402
from utree import UTree, random_move  # noqa: E402


def main():
    lib = api.PllLib()
    rng = np.random.Generator(np.random.PCG64(3))
    tips, sites = 128, 125000
    tree = UTree(tips, rng)
    seqs = W.states_to_sequences(W.section8d_states(tips, sites, 4), W.NT_CHARS)
    d = Driven(lib, tree, 4, sites, api.SITE_REPEATS, seqs, W.map_nt(), W.GTR_DNA["exch"], W.GTR_DNA["freqs"], W.gamma_rates_mean(0.5, 4))
    work = lambda: (lib.pll_gpu_class_map_work(d.p, 0), lib.pll_gpu_class_map_work(d.p, 1))
    rec = tree.inner_edges()[0]
    ops = tree.ops_for(rec)
    d.update(ops)
    v0 = d.lnl(tree.edge_args(rec))
    w1 = work()
    print(f"phase 1: first traversal, {len(ops)} ops: {w1[0]} map operations on the device, {w1[1]} class kernels; lnL {v0:.6f}")
    for k in range(50):
        d.matrices([(m, x * (1.0 + 0.001 * (k + 1))) for m, x in tree.branches()])
        d.update(ops)
        d.lnl(tree.edge_args(rec))
    w2 = work()
    print(f"phase 2: 50 x pll_update_partials on the unchanged tree (new branch lengths each time): {w2[0] - w1[0]} map operations, {w2[1] - w1[1]} class kernels")
    nops = 0
    for k in range(20):
        rec, changed = random_move(tree, rng, spr_share=0.0)
        d.matrices(changed)
        o = tree.ops_for(rec)
        nops += len(o)
        d.update(o)
        d.lnl(tree.edge_args(rec))
    w3 = work()
    print(f"phase 3: 20 NNI moves, {nops} ops in their partial traversals: {w3[0] - w2[0]} map operations on the device "
          f"(the others cannot be compressed: settled on the host), {w3[1] - w2[1]} class kernels")
    print(f"total class kernels: {w3[1]}")
    d.close()


if __name__ == "__main__":
    main()
