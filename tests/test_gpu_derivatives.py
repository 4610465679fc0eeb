"""GPU: pll_update_sumtable / pll_compute_likelihood_derivatives through
the C ABI against the reference-generated golden vectors, the values pinned in the reference's
test/out/derivatives.out, and the restatement on seeded cases."""
import numpy as np
import pytest

from conftest import DERIV_GOLDEN
from deriv_common import assert_sumtable, close, load, run_session
from oracle import oracle_deriv as OD
from pllamd import api, driver, workload as W

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", DERIV_GOLDEN, ids=lambda p: p.split("/")[-1][:-4])
def test_golden_same_eigenbasis(amd_lib, path):
    """the reference's eigensystem written into the partition: sumtable comparable entry by entry"""
    case, eig, rates, edges, brlens, exp_d, exp_st, extra = load(path)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        got_d, got_st = run_session(s, case, eig, rates, edges, brlens, inject=True)
    for i in range(len(edges)):
        assert_sumtable(got_st[i], exp_st[i], case.name)
        for (g1, g2), (e1, e2) in zip(got_d[i], exp_d[i]):
            assert close(g1, e1, sites=case.sites) and close(g2, e2, sites=case.sites), (case.name, g1, e1, g2, e2)
        if "kat" in extra:
            for (g1, g2), (_, p1, p2) in zip(got_d[i], extra["kat"][i]):
                assert abs(g1 - p1) <= 6e-5 * abs(p1) + 1e-13 and abs(g2 - p2) <= 6e-5 * abs(p2) + 1e-13


@pytest.mark.parametrize("path", DERIV_GOLDEN, ids=lambda p: p.split("/")[-1][:-4])
def test_golden_own_eigensystem(amd_lib, path):
    """model through the setters, eigensystem by this library (Jacobi): the derivatives do not
    depend on the eigenbasis"""
    case, eig, rates, edges, brlens, exp_d, exp_st, extra = load(path)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        got_d, _ = run_session(s, case, eig, rates, edges, brlens, inject=False, exch=np.array(extra["exch"]))
    for i in range(len(edges)):
        for (g1, g2), (e1, e2) in zip(got_d[i], exp_d[i]):
            assert close(g1, e1, tol=1e-9, sites=case.sites) and close(g2, e2, tol=1e-9, sites=case.sites), (case.name, g1, e1, g2, e2)


SEEDED = [
    dict(states=4, tips=32, sites=1000, seed=71),
    dict(states=4, tips=32, sites=777, attributes=api.PATTERN_TIP, ambiguity_pct=5, seed=72),
    dict(states=4, tips=32, sites=900, attributes=api.SITE_REPEATS, mutate_pct=5, seed=73),
    dict(states=4, tips=16, sites=130, rate_cats=8, attributes=api.RATE_SCALERS, seed=74),
    dict(states=4, tips=256, sites=70, tree="caterpillar", brlen_scale=4, attributes=api.RATE_SCALERS, seed=75),
    dict(states=4, tips=16, sites=400, pinv=0.3, mutate_pct=4, seed=76),
    dict(states=20, tips=16, sites=200, seed=77),
    dict(states=20, tips=16, sites=130, attributes=api.PATTERN_TIP | api.RATE_SCALERS, seed=78),
    dict(states=9, tips=8, sites=100, rate_cats=3, seed=79),
    dict(states=61, tips=8, sites=100, seed=80),
]


@pytest.mark.parametrize("kw", SEEDED, ids=lambda k: "s%d-t%d-a%d" % (k["states"], k["tips"], k.get("attributes", 0)))
def test_against_oracle(amd_lib, kw):
    case = W.make_case("d", **kw)
    eig = W.eigensystem(case.model["exch"], case.freqs[0])
    e = case.edges[0]
    edges = [((e[0], e[1], e[2], e[3]), 0)]
    brlens = [0.002, 0.07, 0.4, 1.3, 9.0]
    exp = OD.run_derivatives(case, eig, case.model["rates"], [edges[0][0]], brlens)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        got_d, got_st = run_session(s, case, eig, case.model["rates"], edges, brlens, inject=True)
    assert_sumtable(got_st[0], exp["sumtable"][0], str(kw))
    for (g1, g2), (e1, e2) in zip(got_d[0], exp["d"][0]):
        assert close(g1, e1, sites=case.sites) and close(g2, e2, sites=case.sites), (kw, g1, e1, g2, e2)


def test_derivatives_agree_with_finite_differences_of_lnl(amd_lib):
    """what the functions are for (examples/newton/newton.c): d_f and dd_f are the first and second
    derivative of -lnL along the edge; check them against central differences of the edge
    log-likelihood evaluated independently through the P-matrix path"""
    case = W.make_case("fd", 4, 16, 2000, seed=5)
    eig = W.eigensystem(case.model["exch"], case.freqs[0])
    e = case.edges[0]
    edge = (e[0], e[1], e[2], e[3])

    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.inject_eigen(eig, case.model["rates"])
        s.update_partials()
        st = s.new_sumtable()
        s.update_sumtable(edge, st)

        def lnl_at(tt):
            pm = W.pmatrices(case.model["exch"], case.freqs[0], case.model["rates"], [tt])
            dst = api.as_np(s.part.pmatrix[e[4]], 4 * 4 * s.sp, np.float64).reshape(4, 4, s.sp)
            dst[:, :, :4] = pm[0]
            amd_lib.pll_gpu_invalidate(s.p, api.DIRTY_PMATRIX, e[4])
            return s.edge_lnl(e, persite=False)[0]

        for t in (0.02, 0.15, 0.8):
            h = 1e-4 * t
            lm, l0, lp = lnl_at(t - h), lnl_at(t), lnl_at(t + h)
            d1, d2 = s.derivatives(edge, st, t)
            fd1 = -(lp - lm) / (2 * h)
            fd2 = -(lp - 2 * l0 + lm) / (h * h)
            assert abs(d1 - fd1) <= 1e-6 * abs(d1) + 1e-6, (t, d1, fd1)
            assert abs(d2 - fd2) <= 1e-3 * abs(d2) + 1e-2, (t, d2, fd2)


def test_sumtable_handles(amd_lib):
    """the host pointer is a handle: five live tables in one partition all keep their device table
    (the fifth used to recycle the first one's slot silently), a table the library never
    produced is uploaded from the caller's buffer, sync fills the host buffer; beyond 16 live tables
    the least recently used is recycled and an evaluation on ITS handle fails loudly"""
    case = W.make_case("h", 4, 16, 300, seed=9)
    eig = W.eigensystem(case.model["exch"], case.freqs[0])
    e = case.edges[0]
    edge = (e[0], e[1], e[2], e[3])
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.inject_eigen(eig, case.model["rates"])
        s.update_partials()
        tabs = [s.new_sumtable() for _ in range(5)]
        for t in tabs:
            s.update_sumtable(edge, t)
        ref = s.derivatives(edge, tabs[4], 0.3)
        for t in tabs:                              # every one of the five is still resident
            assert s.derivatives(edge, t, 0.3) == ref
        host = s.read_sumtable(tabs[4])          # fills tabs[4] on the host
        assert np.abs(host).max() > 0
        foreign = s.new_sumtable()               # "a caller-written table": the numbers come from the host buffer
        foreign[:] = tabs[4]
        got = s.derivatives(edge, foreign, 0.3)
        assert close(got[0], ref[0], sites=300) and close(got[1], ref[1], sites=300)
        assert not amd_lib.pll_gpu_sync_sumtable(s.p, api.dptr(s.new_sumtable()))
        # 6 tables are live (5 + foreign); 11 more overflow the 16 slots by one: tabs[0] (least recently
        # used) is recycled, and using it must fail - its host buffer was never written
        more = [s.new_sumtable() for _ in range(11)]
        for t in more:
            s.update_sumtable(edge, t)
        with pytest.raises(RuntimeError, match="recycled"):
            s.derivatives(edge, tabs[0], 0.3)
        assert s.derivatives(edge, tabs[1], 0.3) == ref
        s.update_sumtable(edge, tabs[0])         # computing it again makes the handle live again
        assert s.derivatives(edge, tabs[0], 0.3) == ref
        # a released table frees its slot; the handle then means "caller-written" again
        assert amd_lib.pll_gpu_release_sumtable(s.p, api.dptr(more[0]))
        assert not amd_lib.pll_gpu_sync_sumtable(s.p, api.dptr(more[0]))
