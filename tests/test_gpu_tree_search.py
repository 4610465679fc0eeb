"""GPU: the hot path called the way a TREE SEARCH calls it, against the reference (round-5 verdict, missing #1 / #2).

ONE partition per library (libpll_amd.so, the reference build) lives through a few hundred random NNI / SPR moves.
After every move: the branch lengths the move changed go through pll_update_prob_matrices, a PARTIAL traversal
re-orients just the CLVs the evaluation at the moved edge needs (the three records of an inner node share one
clv_index: same parent index, other children than the last time - `/root/reference/src/utree.c:317-366`,
`/root/reference/examples/partial-traversal/partial.c:374-432`, moves as `/root/reference/src/utree_moves.c:72-229`),
pll_update_partials (the reference's default: class maps recomputed, update_repeats = 1) and the edge log-likelihood;
then the evaluated branch gets a new length and is evaluated again; every 10th move a sumtable and two derivative
evaluations; every 25th a full traversal. Everything the device layer keeps between calls is warm the whole time - plan
caches, class-map stamps and versions, packed sub-tree look-ups, kept bitmaps, descriptors on the device, the level
forecast, the host's short path - and none of it may show: log-likelihoods within 1e-10 of the reference's AVX2 path,
under site repeats the class counts and the maps of every node a call touched bit for bit.

Trees: 64 taxa (random) and 2000 taxa at 200 sites (the size the reference pins scaling on,
`/root/reference/test/src/scaling.c:30`; crosses 128 ops per class-map launch and 32 descriptors per kernarg many times
over on a random topology, and actually rescales)."""
import ctypes as C
import os

import numpy as np
import pytest

from compare import RTOL
from deriv_common import close
from pllamd import api, workload as W
from utree import UTree, random_move

pytestmark = pytest.mark.gpu

MOVES = int(os.environ.get("PLL_TREE_SEARCH_MOVES", "200"))


class Driven:
    """one library's partition, driven by the search"""

    def __init__(self, lib, tree, states, sites, attrs, seqs, cmap, exch, freqs, rates):
        self.lib, self.states, self.sites, self.attrs = lib, states, sites, attrs
        t = tree.tips
        self.p = lib.pll_partition_create(t, t - 2, states, sites, 1, 2 * t - 3, len(rates), t - 2, attrs | api.ARCH_AVX2)
        assert self.p, (lib.errno(), lib.errmsg())
        self.part = self.p.contents
        self.rate_cats = len(rates)
        f = np.ascontiguousarray(freqs, dtype=np.float64)
        e = np.ascontiguousarray(exch, dtype=np.float64)
        r = np.ascontiguousarray(rates, dtype=np.float64)
        lib.pll_set_frequencies(self.p, 0, api.dptr(f))
        lib.pll_set_subst_params(self.p, 0, api.dptr(e))
        lib.pll_set_category_rates(self.p, api.dptr(r))
        self.cmap = (C.c_ulonglong * 256)(*[int(x) for x in cmap])
        for i, s in enumerate(seqs):
            assert lib.pll_set_tip_states(self.p, i, self.cmap, s), (lib.errno(), lib.errmsg())
        self.params = np.zeros(self.rate_cats, dtype=np.uint32)
        self.matrices(tree.branches())
        sp = self.part.states_padded
        raw = np.zeros((sites + states) * self.rate_cats * sp + 8, dtype=np.float64)
        off = (-raw.ctypes.data // 8) % 8
        self._sumtable_raw = raw
        self.sumtable = raw[off:off + (sites + states) * self.rate_cats * sp]

    def close(self):
        self.lib.pll_partition_destroy(self.p)

    def matrices(self, pairs):
        if not pairs:
            return
        idx = np.ascontiguousarray([m for m, _ in pairs], dtype=np.uint32)
        bl = np.ascontiguousarray([x for _, x in pairs], dtype=np.float64)
        assert self.lib.pll_update_prob_matrices(self.p, api.uptr(self.params), api.uptr(idx), api.dptr(bl), len(pairs))

    def update(self, ops, update_repeats=1):
        if not ops:
            return
        arr = api.make_ops(ops)
        if update_repeats == 1:
            self.lib.pll_update_partials(self.p, arr, len(ops))
        else:
            self.lib.pll_update_partials_rep(self.p, arr, len(ops), update_repeats)

    def lnl(self, edge):
        return self.lib.pll_compute_edge_loglikelihood(self.p, edge[0], edge[1], edge[2], edge[3], edge[4], api.uptr(self.params), None)

    def derivatives(self, edge, lengths):
        assert self.lib.pll_update_sumtable(self.p, edge[0], edge[2], edge[1], edge[3], api.uptr(self.params), api.dptr(self.sumtable))
        out = []
        for t in lengths:
            d1, d2 = C.c_double(0), C.c_double(0)
            assert self.lib.pll_compute_likelihood_derivatives(self.p, edge[1], edge[3], float(t), api.uptr(self.params),
                                                               api.dptr(self.sumtable), C.byref(d1), C.byref(d2))
            out.append((d1.value, d2.value))
        return out

    def maps(self, node):
        """(classes, site -> class, class -> first site) of a node as the library's accessors give them"""
        ids = int(self.part.repeats.contents.pernode_ids[node])
        if not ids:
            return 0, None, None
        sid = api.as_np(self.lib.pll_get_site_id(self.p, node), self.sites, np.uint32).copy()
        ist = api.as_np(self.lib.pll_get_id_site(self.p, node), ids, np.uint32).copy()
        return ids, sid, ist


def _alignment(states, tips, sites, seed, mutate_pct):
    st = W.random_states(tips, sites, states, seed, mutate_pct)
    if states == 4:
        return W.states_to_sequences(st, W.NT_CHARS), W.map_nt(), W.GTR_DNA["exch"], W.GTR_DNA["freqs"]
    ex, fr = W.synthetic_exch(states)
    if states == 20:
        return W.states_to_sequences(st, W.AA_CHARS), W.map_aa(), ex, fr
    return W.states_to_sequences(st, bytes(range(48, 48 + states))), W.map_generic(states), ex, fr


def _search(libs, states, tips, sites, attrs, seed, moves, check=None, full_every=25, deriv_every=10, spr_radius=6, tip_every=23):
    """drive every library of `libs` through the same search; `check(step, what, values per library, context)` after
    every evaluation. Returns the per-library list of everything evaluated (bit-identity controls compare those)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tree = UTree(tips, rng)
    tree.check()
    seqs, cmap, exch, freqs = _alignment(states, tips, sites, seed + 1, mutate_pct=int(rng.choice([5, 15, 30])))
    rates = W.gamma_rates_mean(0.7, 4)
    drv = [Driven(lib, tree, states, sites, attrs, seqs, cmap, exch, freqs, rates) for lib in libs]
    log = [[] for _ in libs]
    repeats = bool(attrs & api.SITE_REPEATS)

    def evaluate(step, what, rec, ops):
        edge = tree.edge_args(rec)
        vals = []
        for d, lg in zip(drv, log):
            d.update(ops)
            v = d.lnl(edge)
            vals.append(v)
            lg.append(v)
        if check:
            check(step, what, vals, dict(ops=len(ops)))
        if repeats and ops:
            for o in ops:
                got = [d.maps(o[0]) for d in drv]
                for d, lg, g in zip(drv, log, got):
                    lg.append((g[0], None if g[1] is None else g[1].tobytes(), None if g[2] is None else g[2].tobytes()))
                if check:
                    check(step, "maps", got, dict(node=o[0]))

    try:
        rec = tree.inner_edges()[0]
        evaluate(-1, "first full traversal", rec, tree.ops_for(rec))
        for step in range(moves):
            rec, changed = random_move(tree, rng, radius=spr_radius)
            if step % 3 == 0:  # a search also re-optimises a branch next to the move
                side = rec.next
                tree.set_length(side, float(rng.uniform(0.02, 0.5)))
                changed = changed + [(side.pm, side.length)]
            for d in drv:
                d.matrices(changed)
            if full_every and step % full_every == full_every - 1:
                tree.forget()
            evaluate(step, "after the move", rec, tree.ops_for(rec))
            # the evaluated branch gets a new length: both ends stay valid, only the matrix moves
            tree.set_length(rec, float(rng.uniform(0.02, 0.5)))
            for d in drv:
                d.matrices([(rec.pm, rec.length)])
            evaluate(step, "new length on the evaluated branch", rec, tree.ops_for(rec))
            if deriv_every and step % deriv_every == deriv_every - 1:
                ds = [d.derivatives(tree.edge_args(rec), (rec.length, 0.11)) for d in drv]
                for lg, x in zip(log, ds):
                    lg.append(x)
                if check:
                    check(step, "derivatives", ds, {})
            if step % 5 == 4:
                # an evaluation somewhere else in the tree: a long partial traversal that turns many CLVs round
                edges = tree.edges()
                far = edges[int(rng.integers(0, len(edges)))]
                far = far if far.inner else far.back
                evaluate(step, "at a random edge", far, tree.ops_for(far))
            if step % 11 == 10:
                # the traversals of two evaluations handed over as ONE list: the second turns CLVs round that the first has
                # read (the three records of a node share a clv_index - write-after-read inside one call)
                edges = tree.inner_edges()
                e1, e2 = (edges[int(i)] for i in rng.choice(len(edges), size=2, replace=False))
                evaluate(step, "two traversals in one call", e2, tree.ops_for(e1) + tree.ops_for(e2))
            if tip_every and step % tip_every == tip_every - 1:
                # new data for a tip (a search over alignments with missing data re-reads sequences rarely; the class maps of
                # everything above the tip move)
                t = int(rng.integers(0, tips))
                seqs[t] = seqs[int(rng.integers(0, tips))]
                for d in drv:
                    assert d.lib.pll_set_tip_states(d.p, t, d.cmap, seqs[t])
                tree.tip_changed(t)
                evaluate(step, "new tip sequence", rec, tree.ops_for(rec))
            if step % 17 == 16 and rec.inner and rec.back.inner:
                # a rejected move: the same swap again gives the old topology back (pll_utree_nni's rollback)
                tree.nni(rec, 0)
                evaluate(step, "nni", rec, tree.ops_for(rec))
                tree.nni(rec, 0)
                evaluate(step, "nni undone", rec, tree.ops_for(rec))
        tree.check()
    finally:
        for d in drv:
            d.close()
    return log


def _against_reference(sites):
    worst = {"lnl": 0.0, "n": 0}

    def check(step, what, vals, ctx):
        if what == "maps":
            (ia, sa, fa), (ir, sr, fr) = vals
            assert ia == ir, (step, ctx, "class count", ia, ir)
            if ia:
                assert np.array_equal(sa, sr), (step, ctx, "site -> class map")
                assert np.array_equal(fa, fr), (step, ctx, "class -> first site map")
        elif what == "derivatives":
            for (a1, a2), (r1, r2) in zip(*vals):
                assert close(a1, r1, sites=sites) and close(a2, r2, sites=sites), (step, vals)
        else:
            a, r = vals
            assert np.isfinite(a) and np.isfinite(r), (step, what, vals)
            rel = abs(a - r) / abs(r)
            worst["lnl"] = max(worst["lnl"], rel)
            worst["n"] += 1
            assert rel <= RTOL, (step, what, ctx, a, r, rel)
    return check, worst


ATTRS = {"plain": 0, "pattern_tip": api.PATTERN_TIP, "site_repeats": api.SITE_REPEATS,
         "site_repeats_rate_scalers": api.SITE_REPEATS | api.RATE_SCALERS}


@pytest.mark.parametrize("shape", [(64, 1500), (2000, 200)], ids=["64taxa", "2000taxa"])
@pytest.mark.parametrize("states", [4, 20])
@pytest.mark.parametrize("attrs", list(ATTRS), ids=list(ATTRS))
def test_a_tree_search_against_the_reference(amd_lib, ref_lib, attrs, states, shape):
    tips, sites = shape
    check, worst = _against_reference(sites)
    moves = MOVES if tips <= 64 else max(40, MOVES // 2) if states == 20 else MOVES
    _search([amd_lib, ref_lib], states, tips, sites, ATTRS[attrs], seed=4100 + 7 * states + tips, moves=moves, check=check)
    print(f"tree search {attrs} {states} states {tips} taxa: {worst['n']} evaluations, worst lnL rel err {worst['lnl']:.2e}")


@pytest.mark.parametrize("states,sites", [(7, 900), (61, 250)], ids=["7states", "61states"])
@pytest.mark.parametrize("attrs", ["plain", "site_repeats"])
def test_a_tree_search_in_other_state_counts(amd_lib, ref_lib, attrs, states, sites):
    """odd state counts (padding) and the fp64 matrix pipe of 33...64 states (cherry tables per pair of tip matrices, the wide
    inner x inner kernel, scaling epilogues) through the same search, 64 taxa"""
    check, worst = _against_reference(sites)
    _search([amd_lib, ref_lib], states, 64, sites, ATTRS[attrs], seed=5200 + states, moves=max(30, MOVES // 4), check=check)
    print(f"tree search {attrs} {states} states 64 taxa: {worst['n']} evaluations, worst lnL rel err {worst['lnl']:.2e}")


@pytest.mark.parametrize("switch", ["PLL_AMD_NO_PLAN_CACHE", "PLL_AMD_REP_STAMPS"])
@pytest.mark.parametrize("attrs", ["plain", "site_repeats"])
def test_caches_do_not_show(amd_lib, monkeypatch, attrs, switch):
    """the same search with and without (a) cached launch plans, (b) class-map stamps: every value and every map the same bits"""
    args = dict(states=4, tips=64, sites=1500, attrs=ATTRS[attrs], seed=977, moves=60)
    warm = _search([amd_lib], **args)[0]
    monkeypatch.setenv(switch, "0" if switch == "PLL_AMD_REP_STAMPS" else "1")
    cold = _search([amd_lib], **args)[0]
    assert len(warm) == len(cold)
    for i, (a, b) in enumerate(zip(warm, cold)):
        assert a == b, (i, a if not isinstance(a, tuple) else a[0], b if not isinstance(b, tuple) else b[0])


def test_an_unchanged_tree_computes_no_class_maps_and_a_move_only_the_path_above_it(amd_lib, ref_lib):
    """what the stamps are for: the launch count of a repeated traversal is the CLV launches alone, a move relaunches
    class kernels only for the ops of its partial traversal"""
    rng = np.random.Generator(np.random.PCG64(5))
    tree = UTree(32, rng)
    seqs, cmap, exch, freqs = _alignment(4, 32, 4000, 3, 10)
    rates = W.gamma_rates_mean(0.7, 4)
    d = Driven(amd_lib, tree, 4, 4000, api.SITE_REPEATS, seqs, cmap, exch, freqs, rates)
    r = Driven(ref_lib, tree, 4, 4000, api.SITE_REPEATS, seqs, cmap, exch, freqs, rates)
    try:
        rec = tree.inner_edges()[0]
        ops = tree.ops_for(rec)
        for x in (d, r):
            x.update(ops)
        assert abs(d.lnl(tree.edge_args(rec)) - r.lnl(tree.edge_args(rec))) <= RTOL * abs(r.lnl(tree.edge_args(rec)))
        work = lambda: (amd_lib.pll_gpu_class_map_work(d.p, 0), amd_lib.pll_gpu_class_map_work(d.p, 1))
        first = work()                                  # (parents that cannot be compressed - a child is not - never reach the device)
        assert 0 < first[0] <= len(ops) and first[1] > 0
        d.update(ops)                                   # the same tree again, the reference's default call:
        assert work() == first                          # ... nothing to compute, nothing launched
        d.matrices([(m, 0.5 * x) for m, x in tree.branches()])
        d.update(ops)                                   # new branch lengths move no map either
        assert work() == first
        r.matrices([(m, 0.5 * x) for m, x in tree.branches()])
        r.update(ops)
        rec2, changed = random_move(tree, rng, spr_share=0.0)   # an NNI: the ops of its partial traversal, no more
        ops2 = tree.ops_for(rec2)
        assert 0 < len(ops2) < len(ops) // 2
        for x in (d, r):
            x.matrices(changed)
            x.update(ops2)
        after_nni = work()
        assert after_nni[0] <= first[0] + len(ops2)
        rec = rec2
        assert not tree.ops_for(rec2)
        tree.forget()
        ops = tree.ops_for(rec)
        for x in (d, r):
            x.update(ops)                               # a full traversal of the moved tree: only what is NOT as the
        moved = work()                                  # partial traversals left it is computed (other orientations)
        assert moved[0] - after_nni[0] < first[0]
        amd_lib.pll_gpu_invalidate(d.p, api.FORGET_REPEATS, -1)  # everything is computed again
        d.update(ops)
        again = work()
        assert again[0] - moved[0] > moved[0] - after_nni[0] and again[0] - moved[0] <= len(ops) and again[1] > moved[1]
        d.update(ops)
        assert work() == again
        # a caller that wrote ONE tip's maps behind the library's back says so: that tip's ancestors are computed again, no more
        v_before = d.lnl(tree.edge_args(rec))
        amd_lib.pll_gpu_invalidate(d.p, api.DIRTY_REPEATS, 5)
        d.update(ops)
        one_tip = work()
        assert 0 < one_tip[0] - again[0] < again[0] - moved[0]
        assert d.lnl(tree.edge_args(rec)) == v_before
        for o in ops:
            a, b = d.maps(o[0]), r.maps(o[0])
            assert a[0] == b[0] and (a[0] == 0 or (np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])))
        assert abs(d.lnl(tree.edge_args(rec)) - r.lnl(tree.edge_args(rec))) <= RTOL * abs(r.lnl(tree.edge_args(rec)))
    finally:
        d.close()
        r.close()


def test_searches_in_concurrent_threads(amd_lib, ref_lib):
    """partitions are independent: two threads drive a search each (a site-repeats and a plain partition per library) at the same
    time - the device blocks a context recycles, the epoch of its plans and the stream they are ordered by belong to the context
    whose entry point is running on the calling thread"""
    import threading
    errors = []

    def run(attrs, states, sites, seed):
        try:
            check, worst = _against_reference(sites)
            _search([amd_lib, ref_lib], states, 48, sites, ATTRS[attrs], seed=seed, moves=60, check=check)
            assert worst["n"] > 100
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors.append((attrs, states, repr(e)))

    threads = [threading.Thread(target=run, args=a) for a in (("site_repeats", 4, 2000, 7001), ("plain", 4, 3000, 7002),
                                                              ("site_repeats_rate_scalers", 20, 300, 7003))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
