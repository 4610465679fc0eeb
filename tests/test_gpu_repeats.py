"""GPU: site-repeats class maps computed on the device (SURVEY section 8 row f4,
kernels_repeats.h) against the reference's sequential table walk (src/repeats.c:299-382).
Integer work: bit-exact, including the class counts and the fall-back to uncompressed nodes."""
import ctypes as C

import numpy as np
import pytest

from pllamd import api, driver, workload as W

pytestmark = pytest.mark.gpu


def _maps(lib, s, sites, through_accessors):
    rep = s.part.repeats.contents
    rows = []
    if lib.is_amd and not through_accessors:
        assert lib.pll_gpu_sync_repeats(s.p, -1)
    for node in range(s.part.nodes):
        ids = rep.pernode_ids[node]
        if through_accessors:
            sid, ids_ptr = lib.pll_get_site_id(s.p, node), lib.pll_get_id_site(s.p, node)
            assert bool(sid) == bool(ids) == bool(ids_ptr)
        else:
            sid, ids_ptr = rep.pernode_site_id[node], rep.pernode_id_site[node]
        rows.append((ids, api.as_np(sid, sites, np.uint32).copy() if ids else None,
                     api.as_np(ids_ptr, ids, np.uint32).copy() if ids else None,
                     lib.pll_get_sites_number(s.p, node), lib.pll_get_clv_size(s.p, node)))
    return rows


CASES = [
    dict(states=4, tips=16, sites=300, mutate_pct=2, seed=7),
    dict(states=4, tips=16, sites=300, mutate_pct=30, seed=8),          # compression stops paying near the root
    dict(states=4, tips=64, sites=5000, mutate_pct=4, seed=9),
    dict(states=4, tips=128, sites=1025, mutate_pct=1, seed=10),        # one site past a workgroup boundary
    dict(states=4, tips=200, sites=700, tree="caterpillar", mutate_pct=2, seed=11),
    dict(states=20, tips=32, sites=2000, mutate_pct=3, seed=12),
    dict(states=61, tips=16, sites=999, mutate_pct=2, seed=13),
    dict(states=4, tips=8, sites=70000, mutate_pct=6, seed=14),         # several scan chunks per op
    dict(states=4, tips=8, sites=16, mutate_pct=20, seed=15),           # the fewest sites that keep site repeats on (src/pll.c:445-449),
    dict(states=4, tips=8, sites=17, mutate_pct=20, seed=16),           # one past a group of sixteen,
    dict(states=4, tips=8, sites=33, mutate_pct=0, seed=17),            # one past a bitmap word, one class per node,
    dict(states=4, tips=16, sites=4097, mutate_pct=0, seed=18),         # every tip the same sequence: at most four classes anywhere
    dict(states=4, tips=8, sites=2000, mutate_pct=100, seed=19),        # sequences of noise: compression ends at once
    dict(states=20, tips=8, sites=40000, mutate_pct=10, seed=20),       # tip maps of up to 21 classes: level-2 tables beyond the narrow form
]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: "s%d-t%d-n%d-m%d" % (k["states"], k["tips"], k["sites"], k["mutate_pct"]))
@pytest.mark.parametrize("how", ["traversal", "op-by-op"])
def test_device_class_maps_match_reference(amd_lib, ref_lib, kw, how):
    case = W.make_case("rep", attributes=api.SITE_REPEATS, **kw)
    ops = api.make_ops(case.op_batches[0])
    res = {}
    for lib in (amd_lib, ref_lib):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            if how == "traversal":
                lib.pll_update_partials(s.p, ops, len(case.op_batches[0]))  # levels of independent ops at once
            else:
                for i in range(len(case.op_batches[0])):
                    lib.pll_update_repeats(s.p, C.byref(ops[i]))
            res[lib.is_amd] = _maps(lib, s, case.sites, through_accessors=(how == "traversal"))
    for node, (a, b) in enumerate(zip(res[True], res[False])):
        assert a[0] == b[0] and a[3] == b[3] and a[4] == b[4], (node, a[0], b[0])
        if a[0]:
            assert (a[1] == b[1]).all() and (a[2] == b[2]).all(), node
    if kw["mutate_pct"] < 100:
        assert any(r[0] for r in res[True][case.tips:]), "no inner node was compressed - test is vacuous"


def test_lookup_size_bounds_the_pair_table(amd_lib, ref_lib):
    """pll_resize_repeats_lookup: a parent whose children have ids_left * ids_right >= the table size
    stays uncompressed on both sides (src/repeats.c:100-110)"""
    case = W.make_case("rep", 4, 16, 3000, attributes=api.SITE_REPEATS, mutate_pct=8, seed=21)
    ops = api.make_ops(case.op_batches[0])
    out = {}
    for lib in (amd_lib, ref_lib):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            lib.pll_resize_repeats_lookup(s.p, 200)
            lib.pll_update_partials(s.p, ops, len(case.op_batches[0]))
            rep = s.part.repeats.contents
            out[lib.is_amd] = [rep.pernode_ids[n] for n in range(s.part.nodes)]
            if lib.is_amd:
                v, _ = s.edge_lnl(case.edges[0], persite=False)
                assert np.isfinite(v)
    assert out[True] == out[False]
    assert 0 in out[True][16:] and any(out[True][16:])


def test_maps_survive_partial_traversals(amd_lib, ref_lib):
    """update_repeats = 0 reuses the device maps; a later partial update recomputes only its parents"""
    case = W.make_case("rep", 4, 16, 800, attributes=api.SITE_REPEATS, mutate_pct=3, seed=22)
    batch = case.op_batches[0]
    ops = api.make_ops(batch)
    tail = api.make_ops(batch[-3:])
    vals = {}
    for lib in (amd_lib, ref_lib):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            lib.pll_update_partials(s.p, ops, len(batch))
            a, _ = s.edge_lnl(case.edges[0], persite=False)
            lib.pll_update_partials_rep(s.p, ops, len(batch), 0)
            b, _ = s.edge_lnl(case.edges[0], persite=False)
            lib.pll_update_partials(s.p, tail, 3)
            c, _ = s.edge_lnl(case.edges[0], persite=False)
            vals[lib.is_amd] = (a, b, c)
    assert vals[True][0] == vals[True][1] == vals[True][2]
    assert abs(vals[True][0] - vals[False][0]) <= 1e-10 * abs(vals[False][0])


@pytest.mark.parametrize("kw", [dict(tips=64, sites=5000, mutate_pct=4, seed=31),                                   # balanced: levels 1-3 collapse
                                dict(tips=128, sites=20000, mutate_pct=30, seed=32),                                 # C4's shape, L3 keeps thousands of classes
                                dict(tips=33, sites=3000, tree="random", mutate_pct=3, seed=33),                     # cherries, (tip, cherry), deeper clades
                                dict(tips=40, sites=2000, tree="random", mutate_pct=2, seed=34, attributes=api.RATE_SCALERS),
                                dict(tips=300, sites=500, tree="caterpillar", mutate_pct=1, seed=35, brlen_scale=4),  # deep: scaling above the subtrees
                                dict(tips=16, sites=700, mutate_pct=50, seed=36),                                    # poor compression: uncompressed parents over compressed children
                                dict(tips=24, sites=900, tree="random", mutate_pct=5, seed=37, ambiguity_pct=10, partial_pct=5)],
                         ids=lambda k: "t%d-n%d-%s" % (k["tips"], k["sites"], k.get("tree", "balanced")))
def test_tip_rooted_subtrees_are_bit_identical(amd_lib, monkeypatch, kw):
    """site repeats: ops whose subtree is tips only (up to three ops deep) are evaluated from the tip
    codes in one launch (k_partials_dna_sub) - every CLV, scaler vector and the log-likelihood must equal,
    bit for bit, what the level-by-level gather kernels produce, and agree with the oracle"""
    from compare import assert_results_match
    from oracle import oracle as O
    kw = dict(kw)
    attrs = api.SITE_REPEATS | kw.pop("attributes", 0)
    case = W.make_case("sub", 4, attributes=attrs, **kw)
    fast = driver.run_case(amd_lib, case, api.ARCH_AVX2)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        launches_fast = amd_lib.pll_gpu_last_launch_count(s.p)
        s.update_partials(update_repeats=0)                 # cached descriptors
        again = s.edge_lnl(case.edges[0], persite=False)[0]
    monkeypatch.setenv("PLL_AMD_NO_SUBTREES", "1")
    plain = driver.run_case(amd_lib, case, api.ARCH_AVX2)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        launches_plain = amd_lib.pll_gpu_last_launch_count(s.p)
    assert launches_fast <= launches_plain
    assert fast["lnl"] == plain["lnl"] and again == plain["lnl"][0]
    for k in plain["clv"]:
        assert np.array_equal(fast["clv"][k], plain["clv"][k]), k
        if k in plain["scaler"]:
            assert np.array_equal(fast["scaler"][k], plain["scaler"][k]), k
    assert_results_match(fast, O.run_case(case), what="subtrees")


@pytest.mark.parametrize("kw", [dict(tips=128, sites=20000, mutate_pct=30, seed=41),                                   # C4's shape: levels 4+5 in groups, level 6 inside the edge kernel
                                dict(tips=64, sites=3000, mutate_pct=40, seed=42, attributes=api.RATE_SCALERS),
                                dict(tips=48, sites=2500, tree="random", mutate_pct=35, seed=43),
                                dict(tips=300, sites=900, tree="caterpillar", mutate_pct=20, seed=44, brlen_scale=4),    # scaling above the groups
                                dict(tips=32, sites=1111, mutate_pct=50, seed=45, ambiguity_pct=10)],
                         ids=lambda k: "t%d-n%d-%s" % (k["tips"], k["sites"], k.get("tree", "balanced")))
def test_groups_over_gathering_producers_are_bit_identical(amd_lib, monkeypatch, kw):
    """site repeats, where compression ends: an uncompressed op over two gathering inner x inner ops is evaluated with
    them (k_partials_dna_gg) - every CLV, scaler vector and the log-likelihood equal, bit for bit, what the launches
    per level give (PLL_AMD_NO_FUSE_GG=1), and agree with the oracle"""
    from compare import assert_results_match
    from oracle import oracle as O
    kw = dict(kw)
    attrs = api.SITE_REPEATS | kw.pop("attributes", 0)
    case = W.make_case("gg", 4, attributes=attrs, **kw)
    fused = driver.run_case(amd_lib, case, api.ARCH_AVX2)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        nf = amd_lib.pll_gpu_last_launch_count(s.p)
        s.update_partials(update_repeats=0)                 # the cached plan
        again = s.edge_lnl(case.edges[0], persite=False)[0]
    monkeypatch.setenv("PLL_AMD_NO_FUSE_GG", "1")
    plain = driver.run_case(amd_lib, case, api.ARCH_AVX2)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        npl = amd_lib.pll_gpu_last_launch_count(s.p)
    assert nf <= npl
    if kw["tips"] == 128:
        assert nf < npl
    assert fused["lnl"] == plain["lnl"] and again == plain["lnl"][0]
    for k in plain["clv"]:
        assert np.array_equal(fused["clv"][k], plain["clv"][k]), k
        if k in plain["scaler"]:
            assert np.array_equal(fused["scaler"][k], plain["scaler"][k]), k
    assert_results_match(fused, O.run_case(case), what="gather groups")


@pytest.mark.parametrize("kw", [dict(states=20, tips=32, sites=2000, mutate_pct=3, seed=51),
                                dict(states=20, tips=32, sites=1999, mutate_pct=3, seed=52, rate_cats=1),
                                dict(states=20, tips=16, sites=777, mutate_pct=5, seed=53, rate_cats=2, attributes=api.RATE_SCALERS),
                                dict(states=20, tips=24, sites=1500, mutate_pct=4, seed=54, rate_cats=3, tree="random"),
                                dict(states=17, tips=16, sites=900, mutate_pct=4, seed=55),
                                dict(states=19, tips=16, sites=333, mutate_pct=6, seed=56, ambiguity_pct=10, partial_pct=5),
                                dict(states=20, tips=150, sites=400, mutate_pct=30, seed=57, tree="caterpillar", brlen_scale=6),  # scaling, per site: the LDS exchange
                                dict(states=20, tips=150, sites=400, mutate_pct=30, seed=58, tree="caterpillar", brlen_scale=6, attributes=api.RATE_SCALERS)],
                         ids=lambda k: "s%d-t%d-n%d-r%d" % (k["states"], k["tips"], k["sites"], k.get("rate_cats", 4)))
def test_matrix_pipe_gather_kernel_against_the_fma_kernel(amd_lib, monkeypatch, kw):
    """17..20 states under site repeats: the gathering launches run on the matrix pipe (k_partials_lean) - class maps
    and scaler vectors equal those of the FMA kernels (PLL_AMD_NO_LEAN=1), CLVs and log-likelihood agree within the
    tolerance (the two pipes sum a contraction in different orders) and with the oracle"""
    from compare import assert_results_match
    from oracle import oracle as O
    kw = dict(kw)
    attrs = api.SITE_REPEATS | kw.pop("attributes", 0)
    case = W.make_case("lean", attributes=attrs, **kw)
    lean = driver.run_case(amd_lib, case, api.ARCH_AVX2)
    monkeypatch.setenv("PLL_AMD_NO_LEAN", "1")
    fma = driver.run_case(amd_lib, case, api.ARCH_AVX2)
    exp = O.run_case(case)
    assert_results_match(lean, exp, what="lean")
    assert_results_match(fma, exp, what="fma")
    for k in fma["scaler"]:
        assert np.array_equal(lean["scaler"][k], fma["scaler"][k]), k
    if "caterpillar" == kw.get("tree"):
        assert sum(int(v.sum()) for v in fma["scaler"].values()) > 0  # the case does rescale


@pytest.mark.parametrize("always", ["0", "1"], ids=["as-reported", "always"])
def test_packed_subtree_lookups_follow_maps_and_tips(amd_lib, monkeypatch, always):
    """the all-tip subtrees are evaluated from per-entry words of packed tip codes that k_sub_pack forms once per set
    of class maps, tip data and descriptors: new tip sequences in a living partition (class maps recomputed) must give
    what a fresh partition gives, and so must going back. Class maps recomputed over unchanged tips (the reference's
    pll_update_partials on every call) leave the words standing - the class kernels report whether an entry map moved
    (PLL_AMD_SUB_PACK_ALWAYS=1: formed anew after every class-map call)"""
    import ctypes
    monkeypatch.setenv("PLL_AMD_SUB_PACK_ALWAYS", always)
    case_a = W.make_case("pk", 4, tips=32, sites=4000, attributes=api.SITE_REPEATS, mutate_pct=4, seed=61)
    case_b = W.make_case("pk", 4, tips=32, sites=4000, attributes=api.SITE_REPEATS, mutate_pct=4, seed=62)
    case_b.pmatrix, case_b.freqs = case_a.pmatrix, case_a.freqs  # the same model and tree, other sequences
    fresh = {}
    for tag, c in (("a", case_a), ("b", case_b)):
        c2 = W.make_case("pk", 4, tips=32, sites=4000, attributes=api.SITE_REPEATS, mutate_pct=4, seed=61)
        c2.sequences = c.sequences
        with driver.Session(amd_lib, c2, api.ARCH_AVX2) as s:
            s.update_partials()
            fresh[tag] = (s.edge_lnl(c2.edges[0], persite=False)[0], s.read_clv(c2.edges[0][0]))
    with driver.Session(amd_lib, case_a, api.ARCH_AVX2) as s:
        cmap = (ctypes.c_ulonglong * 256)(*[int(x) for x in case_a.charmap])
        for tag, c in (("a", case_a), ("b", case_b), ("a", case_a)):
            for t, seq in enumerate(c.sequences):
                assert amd_lib.pll_set_tip_states(s.p, t, cmap, seq)
            s.update_partials(update_repeats=1)
            v = s.edge_lnl(case_a.edges[0], persite=False)[0]
            assert v == fresh[tag][0], tag
            assert np.array_equal(s.read_clv(case_a.edges[0][0]), fresh[tag][1]), tag
            s.update_partials(update_repeats=0)  # the cached plan and the packed words as they are
            assert s.edge_lnl(case_a.edges[0], persite=False)[0] == v
            for k in range(3):  # the maps again, as they were: recomputed (round 6: only when their inputs are forgotten) or recognised
                if k < 2:
                    amd_lib.pll_gpu_invalidate(s.p, api.FORGET_REPEATS, -1)
                s.update_partials(update_repeats=1)
                assert s.edge_lnl(case_a.edges[0], persite=False)[0] == v
                assert np.array_equal(s.read_clv(case_a.edges[0][0]), fresh[tag][1]), tag


@pytest.mark.parametrize("env", [{"PLL_AMD_REP_LEVEL_SYNC": "1"}, {"PLL_AMD_REP_HINTS": "0"}, {"PLL_AMD_REP_WGS": "1"}, {"PLL_AMD_REP_WGS": "64"},
                                 {"PLL_AMD_REP_RANGES": "1"}, {"PLL_AMD_REP_RANGES": "16", "PLL_AMD_REP_WGS": "64"}, {"PLL_AMD_FENCED_HANDOFF": "1"},
                                 {"PLL_AMD_REP_FUSE": "0"}, {"PLL_AMD_REP_BITS": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
@pytest.mark.parametrize("kw", [dict(states=4, tips=64, sites=70000, mutate_pct=4, seed=81),     # small and large tables, several ranges and parts
                                dict(states=4, tips=16, sites=300, mutate_pct=30, seed=82),
                                dict(states=4, tips=90, sites=33333, mutate_pct=2, seed=83, tree="random"),       # levels of mixed table sizes and map forms
                                dict(states=20, tips=120, sites=5000, mutate_pct=1, seed=84, tree="caterpillar")],  # (tip, inner) all the way: one op per level
                         ids=lambda k: "t%d-n%d" % (k["tips"], k["sites"]))
def test_class_maps_do_not_depend_on_how_the_launches_are_cut(amd_lib, ref_lib, monkeypatch, kw, env):
    """round 5: all levels in one call with the decisions on the device, against the level-by-level form with the
    decisions on the host (PLL_AMD_REP_LEVEL_SYNC=1), without the level forecast, with other numbers of workgroups per
    op and of site ranges per table part, with the in-model hand-off: the same maps as the reference's table walk"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    case = W.make_case("rep", attributes=api.SITE_REPEATS, **kw)
    ops = api.make_ops(case.op_batches[0])
    res = {}
    for lib in (amd_lib, ref_lib):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            lib.pll_update_partials(s.p, ops, len(case.op_batches[0]))
            if lib.is_amd:  # (round 6: an unchanged tree computes no maps - everything known about their inputs is dropped first)
                lib.pll_gpu_invalidate(s.p, api.FORGET_REPEATS, -1)
            lib.pll_update_partials(s.p, ops, len(case.op_batches[0]))  # (the forecast of the first call, the cached launches)
            res[lib.is_amd] = (_maps(lib, s, case.sites, through_accessors=True), s.edge_lnl(case.edges[0], persite=False)[0])
    for node, (a, b) in enumerate(zip(res[True][0], res[False][0])):
        assert a[0] == b[0] and a[3] == b[3] and a[4] == b[4], (node, a[0], b[0])
        if a[0]:
            assert (a[1] == b[1]).all() and (a[2] == b[2]).all(), node
    assert abs(res[True][1] - res[False][1]) <= 1e-10 * abs(res[False][1])


def test_a_wrong_level_forecast_is_repaired(amd_lib, ref_lib):
    """the levels a class-map call launches follow where compression ended the last time (pllgpu_repeats_classes);
    sequences that compress deeper than the last ones did must still get the reference's maps - the call notices that
    the rule admits a parent above the launched levels and runs every level - and so must going back"""
    import ctypes
    shallow = W.make_case("fc", 4, tips=64, sites=4000, attributes=api.SITE_REPEATS, mutate_pct=45, seed=91)
    deep = W.make_case("fc", 4, tips=64, sites=4000, attributes=api.SITE_REPEATS, mutate_pct=1, seed=92)
    ops = api.make_ops(shallow.op_batches[0])
    n = len(shallow.op_batches[0])
    want = {}
    for tag, c in (("shallow", shallow), ("deep", deep)):
        with driver.Session(ref_lib, c, api.ARCH_AVX2) as s:
            ref_lib.pll_update_partials(s.p, ops, n)
            want[tag] = [s.part.repeats.contents.pernode_ids[i] for i in range(s.part.nodes)]
    assert sum(1 for v in want["deep"][64:] if v) > sum(1 for v in want["shallow"][64:] if v)  # the second tree compresses further up
    with driver.Session(amd_lib, shallow, api.ARCH_AVX2) as s:
        cmap = (ctypes.c_ulonglong * 256)(*[int(x) for x in shallow.charmap])
        for tag, c in (("shallow", shallow), ("deep", deep), ("shallow", shallow), ("deep", deep)):
            for t, seq in enumerate(c.sequences):
                assert amd_lib.pll_set_tip_states(s.p, t, cmap, seq)
            amd_lib.pll_update_partials(s.p, ops, n)
            got = [s.part.repeats.contents.pernode_ids[i] for i in range(s.part.nodes)]
            assert got == want[tag], tag
            assert np.isfinite(s.edge_lnl(shallow.edges[0], persite=False)[0])


def test_a_callers_enable_repeats_callback_is_asked_level_by_level(amd_lib, ref_lib):
    """pll_repeats_t::enable_repeats supplied by the caller (src/pll.h:292-297): the decision stays with the callback -
    it sees the class counts of the levels below in pernode_ids, as in the reference's op loop - and the maps are the
    reference's under the same callback; pll_no_enable_repeats switches compression off altogether"""
    case = W.make_case("cb", 4, tips=32, sites=3000, attributes=api.SITE_REPEATS, mutate_pct=4, seed=95)
    ops = api.make_ops(case.op_batches[0])
    n = len(case.op_batches[0])
    asked = {True: [], False: []}

    def make_cb(is_amd):
        @C.CFUNCTYPE(C.c_uint, C.POINTER(api.Partition), C.c_uint, C.c_uint)
        def cb(p, left, right):
            ids = p.contents.repeats.contents.pernode_ids
            asked[is_amd].append((left, right, ids[left], ids[right]))
            return 1 if (0 < ids[left] <= 40 and 0 < ids[right] <= 40) else 0  # an own rule: small children only
        return cb

    res = {}
    for lib in (amd_lib, ref_lib):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            cb = make_cb(lib.is_amd)
            s.part.repeats.contents.enable_repeats = C.cast(cb, C.c_void_p).value
            lib.pll_update_partials(s.p, ops, n)
            res[lib.is_amd] = (_maps(lib, s, case.sites, through_accessors=True), s.edge_lnl(case.edges[0], persite=False)[0])
    assert sorted(asked[True]) == sorted(asked[False]) and len(asked[True]) == n  # the same questions, with the same counts
    for node, (a, b) in enumerate(zip(res[True][0], res[False][0])):
        assert a[0] == b[0] and a[3] == b[3], (node, a[0], b[0])
        if a[0]:
            assert (a[1] == b[1]).all() and (a[2] == b[2]).all(), node
    assert any(r[0] for r in res[True][0][case.tips:]) and not all(r[0] for r in res[True][0][case.tips:])
    assert abs(res[True][1] - res[False][1]) <= 1e-10 * abs(res[False][1])
    # the library's own "never" callback
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.part.repeats.contents.enable_repeats = C.cast(amd_lib.dll.pll_no_enable_repeats, C.c_void_p).value
        amd_lib.pll_update_partials(s.p, ops, n)
        assert not any(s.part.repeats.contents.pernode_ids[i] for i in range(case.tips, s.part.nodes))
        assert abs(s.edge_lnl(case.edges[0], persite=False)[0] - res[False][1]) <= 1e-10 * abs(res[False][1])


@pytest.mark.parametrize("attributes", [api.SITE_REPEATS, 0, api.PATTERN_TIP], ids=["site-repeats", "plain", "pattern-tip"])
def test_the_same_list_again_goes_straight_to_the_launches(amd_lib, attributes):
    """a re-evaluation of one tree - the same operation list, nothing edited in between - skips levels, flushes and
    classification (partials.c: nothing_dirty). Tips the device reads as one-byte codes keep their indicator CLV in
    the host mirror for good; that must not look like a pending upload (it did until round 5: every partition without
    PLL_ATTRIB_PATTERN_TIP took the whole path on every call). With site repeats the class maps recomputed by the call
    (update_repeats = 1) leave the short path standing as long as they come out as they were."""
    case = W.make_case("again", 4, tips=32, sites=3000, attributes=attributes, mutate_pct=3, seed=91)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        assert amd_lib.pll_gpu_last_update_replayed(s.p) == 0
        first = s.edge_lnl(case.edges[0], persite=False)[0]
        for update_repeats in ((0, 1, 2, 2, 0) if attributes & api.SITE_REPEATS else (1, 1)):
            if update_repeats == 2:  # every class map computed again: they come out as they were, the short path stands
                amd_lib.pll_gpu_invalidate(s.p, api.FORGET_REPEATS, -1)
                update_repeats = 1
            s.update_partials(update_repeats=update_repeats)
            assert amd_lib.pll_gpu_last_update_replayed(s.p) == 1, update_repeats
            assert s.edge_lnl(case.edges[0], persite=False)[0] == first
        # an edit in between: the whole path once, then the short one again
        import ctypes
        cmap = (ctypes.c_ulonglong * 256)(*[int(x) for x in case.charmap])
        assert amd_lib.pll_set_tip_states(s.p, 3, cmap, case.sequences[3])
        s.update_partials()
        assert amd_lib.pll_gpu_last_update_replayed(s.p) == 0
        assert s.edge_lnl(case.edges[0], persite=False)[0] == first
        s.update_partials()
        assert amd_lib.pll_gpu_last_update_replayed(s.p) == 1
        assert s.edge_lnl(case.edges[0], persite=False)[0] == first
