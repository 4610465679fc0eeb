"""GPU: per-category model indices (mixture models such as LG4X, examples/lg4/lg4.c:367 in the
reference): freqs_indices / params_indices select a different frequency vector, proportion of
invariant sites and eigensystem for every rate category, with non-uniform category weights.
Edge and root lnL, sumtable + derivatives and device-formed transition matrices against the
reference library and the restatement."""
import numpy as np
import pytest

from compare import RTOL, assert_results_match
from deriv_common import assert_sumtable, close
from oracle import oracle as O
from oracle import oracle_deriv as OD
from pllamd import api, driver, workload as W

pytestmark = pytest.mark.gpu


def mixture_case(states, tips, sites, seed, attributes=0, pinv=(0.0, 0.0, 0.0, 0.0), tree="balanced"):
    """four rate categories, each with its own model"""
    base = W.make_case("mix", states, tips, sites, seed=seed, attributes=attributes, tree=tree)
    rng = np.random.default_rng(seed)
    nex = states * (states - 1) // 2
    models = []
    for k in range(4):
        fr = rng.uniform(0.5, 1.5, states)
        models.append((rng.uniform(0.3, 3.0, nex), fr / fr.sum()))
    rates = np.array([0.3, 0.8, 1.2, 2.1])
    weights = np.array([0.1, 0.4, 0.3, 0.2])
    brl = W.branch_lengths(base.prob_matrices)
    pm = np.empty_like(base.pmatrix)
    for k, (ex, fr) in enumerate(models):
        pm[:, k] = W.pmatrices(ex, fr, rates, brl, pinv[k])[:, k]
    case = driver.Case(name="mix", states=states, rate_cats=4, tips=tips, sites=sites, pmatrix=pm,
                       freqs=np.stack([m[1] for m in models]), op_batches=base.op_batches, edges=base.edges,
                       roots=[(base.edges[0][0], base.edges[0][1])], charmap=base.charmap, sequences=base.sequences,
                       attributes=attributes, clv_buffers=base.clv_buffers, scale_buffers=base.scale_buffers,
                       rate_weights=weights, prop_invar=np.array(pinv), freqs_indices=np.arange(4, dtype=np.uint32))
    return case, models, rates, brl


@pytest.mark.parametrize("states,attrs,pinv", [(4, 0, (0, 0, 0, 0)), (4, api.PATTERN_TIP | api.RATE_SCALERS, (0.1, 0.0, 0.3, 0.2)),
                                               (4, api.SITE_REPEATS, (0, 0, 0, 0)), (20, 0, (0.0, 0.2, 0.0, 0.1)), (61, 0, (0, 0, 0, 0))])
def test_mixture_lnl(amd_lib, ref_lib, states, attrs, pinv):
    case, _, _, _ = mixture_case(states, 16, 400 if states < 61 else 120, seed=200 + states, attributes=attrs, pinv=pinv)
    got = driver.run_case(amd_lib, case, api.ARCH_AVX2)
    ref = driver.run_case(ref_lib, case, api.ARCH_AVX2)
    exp = O.run_case(case)
    assert_results_match(got, exp, what="mixture vs oracle")
    assert abs(got["lnl"][0] - ref["lnl"][0]) <= RTOL * abs(ref["lnl"][0])
    assert abs(got["root_lnl"][0] - ref["root_lnl"][0]) <= RTOL * abs(ref["root_lnl"][0])
    assert np.all(np.abs(got["persite"][0] - ref["persite"][0]) <= RTOL * np.maximum(np.abs(ref["persite"][0]), 1.0))


@pytest.mark.parametrize("states", [4, 20])
def test_mixture_model_api(amd_lib, ref_lib, states):
    """the same through the model setters: per-category eigensystems feed pll_update_prob_matrices,
    pll_update_sumtable and pll_compute_likelihood_derivatives"""
    case, models, rates, brl = mixture_case(states, 8, 300, seed=300 + states)
    e = case.edges[0]
    out = {}
    for lib in (ref_lib, amd_lib):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            for m, (ex, fr) in enumerate(models):
                lib.pll_set_frequencies(s.p, m, api.dptr(np.ascontiguousarray(fr)))
                lib.pll_set_subst_params(s.p, m, api.dptr(np.ascontiguousarray(ex)))
            lib.pll_set_category_rates(s.p, api.dptr(np.ascontiguousarray(rates)))
            pi = np.arange(4, dtype=np.uint32)
            mi = np.arange(case.prob_matrices, dtype=np.uint32)
            assert lib.pll_update_prob_matrices(s.p, api.uptr(pi), api.uptr(mi), api.dptr(np.ascontiguousarray(brl)), len(mi))
            s.update_partials()
            lnl, _ = s.edge_lnl(e, persite=False)
            st = s.new_sumtable()
            s.update_sumtable(e, st)
            d = [s.derivatives(e, st, t) for t in (0.01, 0.2, 1.5)]
            out[lib.is_amd] = (lnl, d)
    assert abs(out[True][0] - out[False][0]) <= RTOL * abs(out[False][0])
    for (g1, g2), (r1, r2) in zip(out[True][1], out[False][1]):
        assert close(g1, r1, tol=1e-9, sites=300) and close(g2, r2, tol=1e-9, sites=300), (g1, r1, g2, r2)
