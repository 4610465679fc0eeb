"""An unrooted binary tree the way a tree search holds it - TEST INFRASTRUCTURE (tree structures and moves are
out of scope for the product: SURVEY 2 / 8; this is the caller the hot path is exercised by).

Semantics follow the reference's data structure without its code: an inner node is a ring of three records that
share ONE clv_index and ONE scaler_index (pll_unode_t, src/pll.h), a record's `back` is the other end of
its edge, both ends of an edge carry the edge's pmatrix_index and length. The CLV standing at clv_index is oriented
towards exactly one of the three records at any time, so a partial traversal re-orients (overwrites) CLVs that an
earlier evaluation left behind - the calling pattern of `examples/partial-traversal/partial.c:374-432`.

* `ops_for(record)`       = pll_utree_traverse(partial) + pll_utree_create_operations (src/utree.c:317-366): post
                            order, child1 = node.next.back, child2 = node.next.next.back, only the CLVs that are not
                            valid in the orientation the evaluation at `record` needs.
* `nni(record, kind)`     = pll_utree_nni (src/utree_moves.c:72-119): the subtrees behind record.next and behind
                            record.back.next (kind 0) or record.back.next.next (kind 1) change places; edges keep
                            their pmatrix_index and length.
* `spr(p, r)`             = pll_utree_spr (src/utree_moves.c:121-229): prune the subtree behind p.back together with
                            p's node, regraft into the edge r <-> r.back; returns the three (pmatrix_index, length)
                            pairs that changed.

Whether a CLV is valid is decided from FIRST PRINCIPLES, not by bookkeeping that mirrors a library's caches: a
record's subtree has a signature (its shape, the tips, the edges and the version of every branch length in it), each
clv_index remembers the signature it was last computed for, and it is valid iff that is the signature the tree gives
the wanted orientation now."""
import sys

import numpy as np

sys.setrecursionlimit(20000)


class Rec:
    __slots__ = ("next", "back", "clv", "scaler", "pm", "length", "uid")

    def __init__(self, clv, scaler, uid):
        self.next = None
        self.back = None
        self.clv = clv
        self.scaler = scaler
        self.pm = -1
        self.length = 0.0
        self.uid = uid

    @property
    def inner(self):
        return self.next is not None


def link(a, b, length, pm):
    a.back, b.back = b, a
    a.length = b.length = length
    a.pm = b.pm = pm


class UTree:
    def __init__(self, tips, rng, lo=0.02, hi=0.4):
        """random topology by stepwise addition of the tips in index order into a random edge"""
        assert tips >= 4
        self.tips = tips
        self.rng = rng
        self.lo, self.hi = lo, hi
        self._uid = 0
        self.tip_recs = [self._rec(t, -1) for t in range(tips)]
        self.inner_nodes = []  # one record of every ring
        self.brver = [0] * (2 * tips - 3)
        self.tipver = [0] * tips
        self.computed = {}  # clv_index -> signature it holds
        self._sig = {}
        next_pm = 0
        # three tips around the first inner node
        ring = self._ring(tips)
        for k, r in enumerate(ring):
            link(r, self.tip_recs[k], self._brlen(), next_pm)
            next_pm += 1
        for t in range(3, tips):
            edges = self.edges()
            a = edges[int(rng.integers(0, len(edges)))]
            b = a.back
            ring = self._ring(tips + t - 2)
            old_len, old_pm = a.length, a.pm
            link(ring[0], self.tip_recs[t], self._brlen(), next_pm)
            link(ring[1], a, old_len, old_pm)
            link(ring[2], b, self._brlen(), next_pm + 1)
            next_pm += 2
        assert next_pm == 2 * tips - 3

    # ---- construction ---------------------------------------------------------------------------
    def _rec(self, clv, scaler):
        self._uid += 1
        return Rec(clv, scaler, self._uid)

    def _ring(self, clv):
        a, b, c = (self._rec(clv, clv - self.tips) for _ in range(3))
        a.next, b.next, c.next = b, c, a
        self.inner_nodes.append(a)
        return a, b, c

    def _brlen(self):
        return float(self.rng.uniform(self.lo, self.hi))

    # ---- views ----------------------------------------------------------------------------------
    def records(self):
        for t in self.tip_recs:
            yield t
        for n in self.inner_nodes:
            yield n
            yield n.next
            yield n.next.next

    def edges(self):
        """one record per edge (the one with the smaller uid)"""
        return [r for r in self.records() if r.back is not None and r.uid < r.back.uid]

    def inner_edges(self):
        return [r for r in self.edges() if r.inner and r.back.inner]

    def branches(self):
        """(pmatrix_index, length) of every edge"""
        return sorted((r.pm, r.length) for r in self.edges())

    def check(self):
        seen = set()
        for r in self.records():
            assert r.back.back is r and r.pm == r.back.pm and r.length == r.back.length
            seen.add(r.pm)
        assert seen == set(range(2 * self.tips - 3))
        reach, todo = set(), [self.tip_recs[0]]
        while todo:
            r = todo.pop()
            if r.uid in reach:
                continue
            reach.add(r.uid)
            todo.append(r.back)
            if r.inner:
                todo += [r.next, r.next.next]
        assert len(reach) == self.tips + 3 * (self.tips - 2)

    # ---- validity -------------------------------------------------------------------------------
    def touched(self):
        """the topology, a branch length or a tip changed: signatures are formed anew"""
        self._sig = {}

    def sig(self, r):
        """signature of the CLV at record r oriented towards r.back"""
        s = self._sig.get(r.uid)
        if s is None:
            if not r.inner:
                s = hash(("tip", r.clv, self.tipver[r.clv]))
            else:
                a, b = r.next, r.next.next
                s = hash((self.sig(a.back), a.pm, self.brver[a.pm], self.sig(b.back), b.pm, self.brver[b.pm]))
            self._sig[r.uid] = s
        return s

    def valid(self, r):
        return (not r.inner) or self.computed.get(r.clv) == self.sig(r)

    def forget(self):
        self.computed = {}

    def set_length(self, r, length):
        r.length = r.back.length = float(length)
        self.brver[r.pm] += 1
        self.touched()

    def tip_changed(self, t):
        self.tipver[t] += 1
        self.touched()

    # ---- traversal -> operations ------------------------------------------------------------------
    def _post(self, r, out):
        if self.valid(r):
            return
        self._post(r.next.back, out)
        self._post(r.next.next.back, out)
        c1, c2 = r.next.back, r.next.next.back
        out.append((r.clv, r.scaler, c1.clv, c1.pm, c1.scaler, c2.clv, c2.pm, c2.scaler))
        self.computed[r.clv] = self.sig(r)

    def ops_for(self, r):
        """operations (8-tuples in pll_operation_t field order) that make both ends of the edge at r valid"""
        out = []
        self._post(r, out)
        self._post(r.back, out)
        return out

    @staticmethod
    def edge_args(r):
        """(parent_clv, parent_scaler, child_clv, child_scaler, matrix) for pll_compute_edge_loglikelihood"""
        return (r.clv, r.scaler, r.back.clv, r.back.scaler, r.pm)

    # ---- moves ----------------------------------------------------------------------------------
    def nni(self, p, kind):
        assert p.inner and p.back.inner
        t1 = p.next
        t2 = p.back.next if kind == 0 else p.back.next.next
        x1, x2 = t1.back, t2.back
        l1, m1, l2, m2 = x1.length, x1.pm, x2.length, x2.pm
        link(t1, x2, l2, m2)
        link(t2, x1, l1, m1)
        self.touched()

    def subtree_records(self, r):
        """every record of the subtree behind r (r's own node and what hangs off its other two records)"""
        out, todo = [], [r]
        while todo:
            q = todo.pop()
            out.append(q)
            if q.inner:
                for k in (q.next, q.next.next):
                    out.append(k)
                    todo.append(k.back)
        return out

    def spr_targets(self, p, radius):
        """records r (one per edge) outside the pruned part, at most `radius` nodes away from the pruning point, that
        give a different tree"""
        u, v = p.next.back, p.next.next.back
        out = []
        todo = [(u, 1), (v, 1)]
        while todo:
            q, d = todo.pop()
            if d >= 2:  # the two edges next to the pruning point give the same tree back
                out.append(q)
            if q.inner and d < radius:
                todo.append((q.next.back, d + 1))
                todo.append((q.next.next.back, d + 1))
        return out

    def spr(self, p, r):
        assert p.inner
        u, v = p.next.back, p.next.next.back
        changed = []
        link(u, v, u.length + v.length, u.pm)
        self.brver[u.pm] += 1
        changed.append((u.pm, u.length))
        rb = r.back
        half = r.length / 2.0
        rpm = r.pm
        link(rb, p.next.next, half, p.next.next.pm)
        self.brver[p.next.next.pm] += 1
        changed.append((p.next.next.pm, half))
        link(r, p.next, half, rpm)
        self.brver[rpm] += 1
        changed.append((rpm, half))
        self.touched()
        return changed


def random_move(tree, rng, spr_share=0.5, radius=6):
    """one random topology move; returns (record of the edge to evaluate at, [(pmatrix_index, new length)...])"""
    if rng.random() < spr_share:
        for _ in range(50):
            cands = [r for n in tree.inner_nodes for r in (n, n.next, n.next.next)]
            p = cands[int(rng.integers(0, len(cands)))]
            targets = tree.spr_targets(p, radius)
            if targets:
                r = targets[int(rng.integers(0, len(targets)))]
                return p, tree.spr(p, r)
    edges = tree.inner_edges()
    p = edges[int(rng.integers(0, len(edges)))]
    tree.nni(p, int(rng.integers(0, 2)))
    return p, []
