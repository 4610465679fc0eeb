"""CPU: the chain partition of pllgpu_update_partials (host logic of the device layer, DESIGN.md
section 4) through its test hook pllgpu_debug_chain_plan - no device is touched. Checked on ladders,
balanced and random trees: every op is placed exactly once, a chain is a path (each step consumes the
step below it), everything a chain reads from HBM is produced in an earlier stage, the stage counts
are the ones the design promises, and lists with anything but producer -> consumer dependencies are
refused (they belong to the level scheduler)."""
import ctypes as C

import numpy as np
import pytest

from pllamd import api, workload as W


class GpuOp(C.Structure):
    _fields_ = [("parent_clv", C.c_uint), ("left_clv", C.c_uint), ("right_clv", C.c_uint),
                ("parent_scaler", C.c_int), ("left_scaler", C.c_int), ("right_scaler", C.c_int),
                ("left_matrix", C.c_uint), ("right_matrix", C.c_uint), ("parent_entries", C.c_uint),
                ("flags", C.c_uint), ("level", C.c_uint), ("war_level", C.c_int)]


LEFT_TIP, RIGHT_TIP = 1, 2


def classify(ops, tips, entries=1000):
    """what host/partials.c hands to the device layer: tip flags (tip on the left in a tip-inner
    pair), dependency levels, stable sort by level; war_level = -1 (no output is touched earlier)"""
    level = {}
    rows = []
    for (p, ps, c1, m1, s1, c2, m2, s2) in ops:
        lv = 1 + max(level.get(c1, -1), level.get(c2, -1))
        level[p] = lv
        t1, t2 = c1 < tips, c2 < tips
        if (not t1) and t2:
            c1, m1, s1, c2, m2, s2, t1, t2 = c2, m2, s2, c1, m1, s1, t2, t1
        rows.append((lv, GpuOp(p, c1, c2, ps, -1 if t1 else s1, -1 if t2 else s2, m1, m2, entries,
                               (LEFT_TIP if t1 else 0) | (RIGHT_TIP if t2 else 0), lv, -1)))
    rows.sort(key=lambda r: r[0])
    arr = (GpuOp * len(rows))(*[r[1] for r in rows])
    return arr


@pytest.fixture(scope="module")
def hook(amd_lib):
    fn = amd_lib.dll.pllgpu_debug_chain_plan
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(GpuOp), C.c_uint, C.c_uint, C.c_uint, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_int), C.POINTER(C.c_ubyte)]
    return fn


def plan(hook, arr, tips, fuse_cc=1):
    n = len(arr)
    stage = (C.c_uint * n)()
    chain = (C.c_int * n)()
    form = (C.c_ubyte * n)()
    stages = hook(arr, n, 2 * tips, tips, fuse_cc, stage, chain, form)
    return stages, list(stage), list(chain), list(form)


def check_invariants(arr, tips, stages, stage, chain, form):
    n = len(arr)
    producer = {arr[i].parent_clv: i for i in range(n)}
    consumer = {}
    for i in range(n):
        for side, tipflag in ((arr[i].left_clv, LEFT_TIP), (arr[i].right_clv, RIGHT_TIP)):
            if not (arr[i].flags & tipflag) and side in producer:
                consumer[producer[side]] = i
    assert all(1 <= st <= stages for st in stage)
    for i in range(n):
        kids = [producer.get(c) for c, f in ((arr[i].left_clv, LEFT_TIP), (arr[i].right_clv, RIGHT_TIP)) if not (arr[i].flags & f)]
        kids = [k for k in kids if k is not None]
        if form[i] == 3:  # seven-op group: stage 1, everything below it is tips or fellow members
            assert stage[i] == 1 and all(form[k] == 3 for k in kids)
            continue
        in_regs = [k for k in kids if form[k] == 1 and chain[k] == chain[i]] if form[i] != 2 else []
        assert len(in_regs) <= 1  # at most one child rides in registers
        for k in kids:
            if k in in_regs:
                assert stage[k] == stage[i]  # the chain runs as one launch
            elif form[k] == 2:  # formed on the fly by this op's step
                assert form[i] != 2 and stage[k] == stage[i] and consumer[k] == i
            else:  # read from HBM: produced by an earlier launch
                assert stage[k] < stage[i], (i, k, stage[i], stage[k])
        if form[i] == 2:
            assert all(form[k] in (0, 3) for k in kids)  # an op formed on the fly reads leaves only
        if form[i] == 1:
            assert i in consumer and chain[consumer[i]] == chain[i]
        if form[i] == 0 and i in consumer:
            assert chain[consumer[i]] != chain[i] or form[consumer[i]] == 2 or True
    # chains are paths: per chain exactly one top, every other member is a lower step
    by_chain = {}
    for i in range(n):
        if chain[i] >= 0:
            by_chain.setdefault(chain[i], []).append(i)
    for members in by_chain.values():
        assert sum(1 for i in members if form[i] == 0) == 1
        assert len({stage[i] for i in members}) == 1


@pytest.mark.parametrize("tips", [8, 64, 200])
def test_ladder_is_one_chain(hook, tips):
    ops, _, _ = W.caterpillar_ops(tips)
    arr = classify(ops, tips)
    stages, stage, chain, form = plan(hook, arr, tips)
    assert stages == 1
    # one chain; its first cherry may be the op that the second step forms on the fly (equal cost)
    assert len(set(c for c in chain if c >= 0)) == 1 and form.count(0) == 1 and form.count(2) <= 1
    check_invariants(arr, tips, stages, stage, chain, form)


@pytest.mark.parametrize("tips,with_cc,without_cc", [(16, 1, 2), (32, 2, 2), (64, 2, 3), (256, 3, 4), (1024, 4, 5)])
def test_balanced_tree_stage_counts(hook, tips, with_cc, without_cc):
    ops, _, _ = W.balanced_ops(tips)
    arr = classify(ops, tips)
    stages, stage, chain, form = plan(hook, arr, tips, fuse_cc=1)
    assert stages == with_cc
    assert form.count(3) == 7 * (tips // 8)  # every complete 8-tip subtree is a seven-op group (the root edge splits a 16-tip tree into two)
    check_invariants(arr, tips, stages, stage, chain, form)
    stages, stage, chain, form = plan(hook, arr, tips, fuse_cc=0)
    assert stages == without_cc and form.count(3) == 0
    check_invariants(arr, tips, stages, stage, chain, form)


@pytest.mark.parametrize("tips,stages16,members", [(16, 1, 14), (32, 1, 30), (64, 2, 60), (128, 2, 120), (256, 3, 240), (1024, 4, 960)])
def test_balanced_tree_with_fifteen_op_groups(hook, tips, stages16, members):
    """fuse_cc bit 1 (round 4, k_partials_dna_cc16): a parent over two complete 8-tip groups takes both - complete
    16-tip subtrees are fifteen-op groups of stage 1; what is left of a balanced tree is shallower by one level (the
    un-rooted 16-tip tree is two 8-tip groups joined by the evaluated edge: nothing above them to take)"""
    ops, _, _ = W.balanced_ops(tips)
    arr = classify(ops, tips)
    stages, stage, chain, form = plan(hook, arr, tips, fuse_cc=3)
    assert stages == stages16 and form.count(3) == members
    check_invariants(arr, tips, stages, stage, chain, form)
    s7, _, _, form7 = plan(hook, arr, tips, fuse_cc=1)
    assert s7 >= stages and form7.count(3) <= members


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("tips", [5, 33, 64, 300])
def test_random_trees_with_fifteen_op_groups(hook, tips, seed):
    ops, _, _ = W.random_tree_ops(tips, seed=seed)
    arr = classify(ops, tips)
    if len(arr) < 1:
        return
    stages, stage, chain, form = plan(hook, arr, tips, fuse_cc=3)
    assert stages >= 1
    check_invariants(arr, tips, stages, stage, chain, form)


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("tips", [5, 33, 64, 300])
def test_random_trees(hook, tips, seed):
    ops, _, _ = W.random_tree_ops(tips, seed=seed)
    arr = classify(ops, tips)
    if len(arr) < 1:
        return
    stages, stage, chain, form = plan(hook, arr, tips)
    assert stages >= 1
    check_invariants(arr, tips, stages, stage, chain, form)
    # far fewer launches than dependency levels
    levels = 1 + max(o.level for o in arr)
    assert stages <= max(2, int(np.ceil(np.log2(tips))) + 1) and stages <= levels


def test_lists_with_other_dependencies_are_refused(hook):
    tips = 16
    ops, _, _ = W.balanced_ops(tips)
    arr = classify(ops, tips)
    assert plan(hook, arr, tips)[0] > 0
    # an op that overwrites a CLV an earlier op of the list touches
    bad = classify(ops, tips)
    bad[5].war_level = 0
    assert plan(hook, bad, tips)[0] == 0
    # a child scaler that is not the one its producer writes
    bad = classify(ops, tips)
    k = next(i for i in range(len(bad)) if not (bad[i].flags & (LEFT_TIP | RIGHT_TIP)))
    bad[k].left_scaler = -1
    assert plan(hook, bad, tips)[0] == 0
    # class-compressed node (site repeats)
    bad = classify(ops, tips)
    bad[0].flags |= 4
    assert plan(hook, bad, tips)[0] == 0
    # the same CLV on both sides
    bad = classify(ops, tips)
    bad[k].right_clv = bad[k].left_clv
    assert plan(hook, bad, tips)[0] == 0
    # a CLV consumed twice
    ops2 = list(ops) + [(2 * tips - 2, -1, ops[-1][0], 0, ops[-1][1], ops[-2][0], 1, ops[-2][1]),
                        (2 * tips - 1, -1, ops[-1][0], 0, ops[-1][1], 0, 1, -1)]
    assert plan(hook, classify(ops2, tips), tips)[0] == 0
