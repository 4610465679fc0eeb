"""GPU: small transfers through the context's block of pinned host memory (pllgpu.hip: stage_take, round 5) - uploads
read out of host memory by the layout kernel, downloads written into it, several downloads behind one wait
(pllgpu_download_defer) - against the runtime's pageable copies (PLL_AMD_PINNED_STAGING=0) and the oracle. Sizes chosen
so that one flush hands out more than the block holds (it starts over after a wait) and so that downloads are still
pending when it does."""
import numpy as np
import pytest

from compare import assert_results_match, scalers_equal
from oracle import oracle as O
from pllamd import api, driver, workload as W

pytestmark = pytest.mark.gpu


def _dense_case(tips, sites, seed, states=4, rate_cats=4, **kw):
    """tips as CLVs that are NOT 0/1 indicators: they go up as dense CLVs (sites x rates x states doubles each)"""
    case = W.make_case("dense", states, tips, sites, rate_cats=rate_cats, tips_as="clv", seed=seed, **kw)
    rng = np.random.default_rng(seed)
    case.tip_clvs = case.tip_clvs * 0.75 + rng.uniform(0.01, 0.2, size=case.tip_clvs.shape)
    return case


def _all(lib, case, sync_all):
    out = {}
    with driver.Session(lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        if sync_all:
            assert lib.pll_gpu_sync_all(s.p)
        parents = {op[0]: op[1] for batch in case.op_batches for op in batch}
        out["clv"] = {i: s.read_clv(i) for i in sorted(parents)}
        out["scaler"] = {i: s.read_scaler(sc, i) for i, sc in parents.items() if sc >= 0}
        v, ps = s.edge_lnl(case.edges[0])
        out["lnl"], out["persite"] = [v], [ps]
        out["root_lnl"], out["root_persite"] = [], []
    return out


@pytest.mark.parametrize("sync_all", [False, True], ids=["one-by-one", "sync-all"])
def test_more_than_the_block_holds_in_one_flush(amd_lib, monkeypatch, sync_all):
    """32 dense tips of 1 MB each go up in one pll_update_partials (four times the block), 30 inner CLVs of 1 MB come
    back - one by one, or enqueued together by pll_gpu_sync_all with the block starting over while downloads are
    pending; the same bits as with the runtime's copies, the oracle's numbers"""
    case = _dense_case(32, 8192, seed=5)
    staged = _all(amd_lib, case, sync_all)
    monkeypatch.setenv("PLL_AMD_PINNED_STAGING", "0")
    plain = _all(amd_lib, case, sync_all)
    for i in staged["clv"]:
        assert np.array_equal(staged["clv"][i], plain["clv"][i]), i
    for i in staged["scaler"]:
        assert np.array_equal(staged["scaler"][i], plain["scaler"][i]), i
    assert staged["lnl"] == plain["lnl"] and np.array_equal(staged["persite"][0], plain["persite"][0])
    small = _dense_case(32, 300, seed=5)  # (the oracle is a scalar loop: the same construction at a size it finishes)
    exp = O.run_case(small)
    got = driver.run_case(amd_lib, small, api.ARCH_AVX2)
    assert_results_match(got, exp, what="dense tips")
    assert scalers_equal(got, exp)


@pytest.mark.parametrize("kw", [dict(states=20, tips=16, sites=3000, rate_cats=4), dict(states=4, tips=8, sites=20001, rate_cats=8, attributes=api.RATE_SCALERS),
                                dict(states=61, tips=8, sites=500, rate_cats=2)], ids=lambda k: "s%d-n%d" % (k["states"], k["sites"]))
def test_other_shapes_through_the_block(amd_lib, monkeypatch, kw):
    """ragged tiles, per-rate scaler vectors (4 x the words), transfers above the 2 MB limit beside ones below it"""
    case = _dense_case(seed=9, **kw)
    staged = _all(amd_lib, case, True)
    monkeypatch.setenv("PLL_AMD_PINNED_STAGING", "0")
    plain = _all(amd_lib, case, True)
    for i in staged["clv"]:
        assert np.array_equal(staged["clv"][i], plain["clv"][i]), i
    for i in staged["scaler"]:
        assert np.array_equal(staged["scaler"][i], plain["scaler"][i]), i
    assert staged["lnl"] == plain["lnl"]


def test_a_host_edit_between_two_downloads(amd_lib):
    """download, edit on the host, invalidate, update, download again: the second download must not hand back what the
    first left in the block"""
    case = _dense_case(8, 4096, seed=3)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        top = next(op[0] for op in case.op_batches[0] if 0 in (op[2], op[5]))  # the parent of tip 0
        first = s.read_clv(top).copy()
        a = np.ascontiguousarray(case.tip_clvs[0] * 0.5, dtype=np.float64)
        assert amd_lib.pll_set_tip_clv(s.p, 0, api.dptr(a), 0)
        s.update_partials()
        second = s.read_clv(top)
        assert not np.array_equal(first, second)
    edited = _dense_case(8, 4096, seed=3)
    edited.tip_clvs[0] = edited.tip_clvs[0] * 0.5
    with driver.Session(amd_lib, edited, api.ARCH_AVX2) as s:
        s.update_partials()
        assert np.array_equal(s.read_clv(top), second)
