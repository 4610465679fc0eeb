"""CPU: the tree-search helper itself (tests/utree.py) - checked on the reference build alone, so that what
tests/test_gpu_tree_search.py feeds both libraries is a correct calling sequence: after random NNI / SPR moves the
log-likelihood reached through PARTIAL traversals on one long-lived partition equals the one a fresh partition gets from
a full traversal of the same tree, and moves keep the tree a tree."""
import numpy as np
import pytest

from pllamd import api, workload as W
from test_gpu_tree_search import Driven, _alignment
from utree import UTree, random_move


def test_moves_keep_the_tree_a_tree():
    rng = np.random.Generator(np.random.PCG64(1))
    tree = UTree(40, rng)
    for _ in range(300):
        rec, changed = random_move(tree, rng)
        tree.check()
        assert len(tree.inner_edges()) == 40 - 3
        for pm, length in changed:
            assert length > 0 and 0 <= pm < 2 * 40 - 3


def test_a_full_traversal_lists_every_inner_node_once_and_a_second_one_nothing():
    rng = np.random.Generator(np.random.PCG64(2))
    tree = UTree(25, rng)
    rec = tree.inner_edges()[3]
    ops = tree.ops_for(rec)
    assert sorted(o[0] for o in ops) == list(range(25, 2 * 25 - 2))
    done = set(range(25))
    for o in ops:  # producers first
        assert o[2] in done and o[5] in done
        done.add(o[0])
    assert tree.ops_for(rec) == []
    tree.nni(rec, 1)
    again = tree.ops_for(rec)
    assert sorted(o[0] for o in again) == sorted([rec.clv, rec.back.clv])


@pytest.mark.parametrize("attrs", [0, api.SITE_REPEATS])
def test_partial_traversals_reach_what_a_full_traversal_of_a_fresh_partition_gives(ref_lib, attrs):
    rng = np.random.Generator(np.random.PCG64(3))
    tips, sites = 24, 300
    tree = UTree(tips, rng)
    seqs, cmap, exch, freqs = _alignment(4, tips, sites, 9, 20)
    rates = W.gamma_rates_mean(0.7, 4)
    live = Driven(ref_lib, tree, 4, sites, attrs, seqs, cmap, exch, freqs, rates)
    try:
        rec = tree.inner_edges()[0]
        live.update(tree.ops_for(rec))
        partial_sizes = []
        for step in range(60):
            rec, changed = random_move(tree, rng)
            live.matrices(changed)
            ops = tree.ops_for(rec)
            partial_sizes.append(len(ops))
            live.update(ops)
            v = live.lnl(tree.edge_args(rec))
            if step % 6 == 5:
                keep = dict(tree.computed)
                tree.forget()
                fresh = Driven(ref_lib, tree, 4, sites, attrs, seqs, cmap, exch, freqs, rates)
                try:
                    fresh.update(tree.ops_for(rec))
                    w = fresh.lnl(tree.edge_args(rec))
                finally:
                    fresh.close()
                tree.computed = keep
                assert abs(v - w) <= 1e-12 * abs(w), (step, v, w)
        assert max(partial_sizes) < tips - 2 and min(partial_sizes) >= 1
    finally:
        live.close()
