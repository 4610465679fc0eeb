"""GPU, BASELINE.json's full configurations. Where the prebuilt reference (oracle/_ref, shipped
with the repo snapshot) is present the HIP path is compared with the reference's AVX2 path directly
at full size; independent of that, size-independent properties are checked: agreement between the
three data paths (tip CLVs / PATTERN_TIP / SITE_REPEATS), linearity in the pattern weights,
invariance under a permutation of the sites, per-site values summing to the total."""
import os

import numpy as np
import pytest

from compare import RTOL
from oracle import oracle as O
from pllamd import api, driver, workload as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lg():
    z = np.load(os.path.join(ROOT, "tests", "golden", "model_lg.npz"))
    return dict(exch=z["rates"], freqs=z["freqs"])


CONFIGS = {
    "C2-dna-64x100k": dict(states=4, tips=64, sites=100000, seed=1000),
    "C3-lg-64x50k": dict(states=20, tips=64, sites=50000, seed=1000, **lg()),
    "C5-codon61-32x20k": dict(states=61, tips=32, sites=20000, seed=1000),
    "C4shard-dna-128x125k": dict(states=4, tips=128, sites=125000, seed=1000),
    # BASELINE's largest site count on ONE device: offsets beyond 2^31 bytes inside a traversal
    "C4sites-dna-16x1M": dict(states=4, tips=16, sites=1000000, seed=1001),
    "C4sites-dna-16x1M-repeats": dict(states=4, tips=16, sites=1000000, seed=1001, mutate_pct=4),
}


def run(lib, case, arch=api.ARCH_AVX2, clvs=()):
    with driver.Session(lib, case, arch) as s:
        s.update_partials()
        lnl, ps = s.edge_lnl(case.edges[0])
        got = {c: (s.read_clv(c), s.read_scaler(c - case.tips, c)) for c in clvs}
        ids = [s.entries(c) for c in range(case.tips, case.tips + case.clv_buffers)]
    return lnl, ps, got, ids


@pytest.fixture(scope="module")
def reference():
    if not os.path.exists(O.REF_LIB):
        pytest.skip("oracle/_ref/libpll_ref.so not shipped")
    return api.PllLib(O.REF_LIB)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_against_reference_avx2(amd_lib, reference, name):
    kw = CONFIGS[name]
    attrs = api.SITE_REPEATS if (name.startswith("C4shard") or name.endswith("repeats")) else 0
    case = W.make_case(name, attributes=attrs, **kw)
    a, b = case.edges[0][0], case.edges[0][2]
    r_lnl, r_ps, r_clv, r_ids = run(reference, case, clvs=(a, b))
    g_lnl, g_ps, g_clv, g_ids = run(amd_lib, case, clvs=(a, b))
    assert abs(g_lnl - r_lnl) <= RTOL * abs(r_lnl), (g_lnl, r_lnl)
    assert np.all(np.abs(g_ps - r_ps) <= RTOL * np.maximum(np.abs(r_ps), 1.0))
    for c in (a, b):
        err = driver.rel_err_normalised(g_clv[c][0], g_clv[c][1], r_clv[c][0], r_clv[c][1])
        assert err <= RTOL, (name, c, err)
    assert g_ids == r_ids  # class counts per inner node (site repeats) are integers: exact


@pytest.mark.parametrize("name", ["C2-dna-64x100k", "C3-lg-64x50k", "C5-codon61-32x20k"])
def test_data_paths_agree_at_full_size(amd_lib, name):
    """tip CLVs, tip codes and site repeats are three routes to the same numbers"""
    kw = CONFIGS[name]
    vals = {}
    for tag, attrs in (("clv", 0), ("tip", api.PATTERN_TIP), ("rep", api.SITE_REPEATS), ("tip+rs", api.PATTERN_TIP | api.RATE_SCALERS)):
        case = W.make_case(name, attributes=attrs, **kw)
        lnl, ps, _, ids = run(amd_lib, case)
        vals[tag] = (lnl, ps)
        if tag == "rep":
            assert min(ids) < kw["sites"], "site repeats compressed nothing"
    base = vals["clv"]
    for tag, (lnl, ps) in vals.items():
        assert abs(lnl - base[0]) <= 1e-12 * abs(base[0]), tag
        assert np.all(np.abs(ps - base[1]) <= 1e-11 * np.maximum(np.abs(base[1]), 1.0)), tag
    assert abs(base[1].sum() - base[0]) <= 1e-11 * abs(base[0])


def test_linearity_and_permutation_at_full_size(amd_lib):
    kw = CONFIGS["C2-dna-64x100k"]
    case = W.make_case("w1", **kw)
    lnl1, ps1, _, _ = run(amd_lib, case)
    rng = np.random.Generator(np.random.PCG64(4))
    w = rng.integers(1, 9, size=kw["sites"]).astype(np.uint32)
    case_w = W.make_case("w", pattern_weights=w, **kw)
    lnl_w, ps_w, _, _ = run(amd_lib, case_w)
    assert np.all(np.abs(ps_w - ps1 * w) <= 1e-12 * np.abs(ps_w) + 1e-300)
    assert abs(lnl_w - float(np.dot(ps1, w))) <= 1e-11 * abs(lnl_w)
    # permute the sites: per-site values permute, the total is unchanged up to summation order
    perm = rng.permutation(kw["sites"])
    case_p = W.make_case("p", **kw)
    case_p.sequences = [np.frombuffer(s, dtype=np.uint8)[perm].tobytes() for s in case.sequences]
    lnl_p, ps_p, _, _ = run(amd_lib, case_p)
    assert np.array_equal(ps_p, ps1[perm])
    assert abs(lnl_p - lnl1) <= 1e-12 * abs(lnl1)


# ---- bench.py's own inputs (SURVEY 8d to the letter) -------------------------------------------------
def _bench():
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def _pinned(key):
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "section8d_lnl.json")))[key]


@pytest.mark.parametrize("key", ["c2", "c3", "c5"])
def test_bench_inputs_against_the_pinned_reference_value(amd_lib, key):
    """the xorshift64 alignment of SURVEY 8d through the HIP path == what the reference's AVX2 path gave
    for the same bytes in the authoring container (tools/gen_section8d_lnl.py)"""
    b = _bench()
    cfg = b.CONFIGS[key]
    case = b.build_case(cfg, cfg["sites"], 0)
    lnl, ps, _, _ = run(amd_lib, case)
    pin = _pinned(key)
    assert abs(lnl - pin) <= RTOL * abs(pin), (lnl, pin)
    assert abs(ps.sum() - lnl) <= 1e-11 * abs(lnl)


@pytest.mark.parametrize("groups", ["seven-op", "fifteen-op"])
def test_c2_intermediate_clvs_at_full_size(amd_lib, reference, groups, monkeypatch):
    """C2 as bench.py runs it: 56 of the 62 CLVs leave the seven-op kernel through non-temporal stores and
    are never read back by the traversal - so the root-edge check above cannot see an addressing error in
    them. Whole CLVs and scalers of a cherry, a level-2 and a level-3 node at both ends of the node range
    (every tile: first, middle, last) and of the top levels against the reference. fifteen-op (round 4): the same
    through k_partials_dna_cc16, which larger alignments and trees take by default - 60 of the 62 CLVs out of one launch,
    the level-4 nodes its plainly stored group parents."""
    monkeypatch.setenv("PLL_AMD_FUSE_CC16", "1" if groups == "fifteen-op" else "0")
    b = _bench()
    cfg = b.CONFIGS["c2"]
    case = b.build_case(cfg, cfg["sites"], 0)
    T = case.tips
    # level 1: T .. T+31, level 2: T+32 .. T+47, level 3: T+48 .. T+55, level 4: T+56 .. T+59, level 5: T+60, T+61
    nodes = (T, T + 17, T + 31, T + 32, T + 47, T + 48, T + 55, T + 56, T + 59, T + 60, T + 61)
    r_lnl, _, r_clv, _ = run(reference, case, clvs=nodes)
    g_lnl, _, g_clv, _ = run(amd_lib, case, clvs=nodes)
    for c in nodes:
        assert g_clv[c][0].shape == (cfg["sites"], 4, 4)
        err = driver.rel_err_normalised(g_clv[c][0], g_clv[c][1], r_clv[c][0], r_clv[c][1])
        assert err <= RTOL, (c, err)
        assert np.array_equal(g_clv[c][1], r_clv[c][1])
    assert abs(g_lnl - r_lnl) <= RTOL * abs(r_lnl)


def _check_nodes(cfg_sites, span, case, nodes, reference, amd_lib):
    r_lnl, _, r_clv, r_ids = run(reference, case, clvs=nodes)
    g_lnl, _, g_clv, g_ids = run(amd_lib, case, clvs=nodes)
    for c in nodes:
        assert g_clv[c][0].shape == (cfg_sites,) + span
        err = driver.rel_err_normalised(g_clv[c][0], g_clv[c][1], r_clv[c][0], r_clv[c][1])
        assert err <= RTOL, (c, err)
        assert np.array_equal(g_clv[c][1], r_clv[c][1]), c  # scaler vectors: exact
    assert abs(g_lnl - r_lnl) <= RTOL * abs(r_lnl)
    assert g_ids == r_ids


def test_c3_intermediate_clvs_at_full_size(amd_lib, reference):
    """C3 as bench.py runs it: the bottom two levels are ONE launch of (tip x tip, tip x tip -> inner x inner)
    groups on the matrix pipe (k_partials_mfma_cc<5>: 48 CLVs per launch, kernels_mfma.h) whose cherries nobody in
    the traversal reads back; the small cases stop at a few hundred entries. Whole CLVs and scalers of the
    first / middle / last node of every level at 50k sites against the reference."""
    b = _bench()
    cfg = b.CONFIGS["c3"]
    case = b.build_case(cfg, cfg["sites"], 0)
    T = case.tips
    nodes = (T, T + 15, T + 31,          # cherries
             T + 32, T + 40, T + 47,     # level 2: the groups' parents
             T + 48, T + 51, T + 55, T + 56, T + 59, T + 60, T + 61)
    _check_nodes(cfg["sites"], (4, 20), case, nodes, reference, amd_lib)


def test_c4_shard_intermediate_clvs_at_full_size(amd_lib, reference):
    """One 125k-site shard of configs[3] as bench.py --gpus 8 runs it (pattern-sorted alignment, site repeats):
    levels 1-3 come out of ONE launch over packed sub-tree look-ups (k_partials_dna_sub), levels 4 + 5 out of
    groups over gathering producers (k_partials_dna_gg: the producers A / B leave through streaming stores and are
    not read back), level 6 inside the edge kernel. Nodes of every level incl. both producers and the parent of a
    group, expanded through the class maps, CLVs <= 1e-10 and scalers exact against the reference; class counts
    exact. The small cases of tests/test_gpu_repeats.py stop at 700 entries."""
    b = _bench()
    from pllamd import sharding
    cfg = b.CONFIGS["c4"]
    full = sharding.sort_columns(amd_lib, b.build_case(cfg, cfg["sites"], api.SITE_REPEATS))
    case = sharding.shard_case(full, 1, 8, sharding.balanced_bounds(full, 8))  # the shard with the most classes
    T = case.tips
    # level 1: T .. T+63, 2: T+64 .. T+95, 3: T+96 .. T+111, 4: T+112 .. T+119, 5: T+120 .. T+123, 6: T+124, T+125
    nodes = (T, T + 63, T + 64, T + 95, T + 96, T + 103, T + 111,
             T + 112, T + 113, T + 120,   # producers A, B and the parent P of the first group
             T + 118, T + 119, T + 123,   # ... and of the last
             T + 124, T + 125)
    _check_nodes(case.sites, (4, 4), case, nodes, reference, amd_lib)
