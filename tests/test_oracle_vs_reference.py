"""CPU, authoring container only: the restatement against the real reference on randomised
cases beyond the committed fixtures (skipped where oracle/_ref is absent)."""
import numpy as np
import pytest

from compare import assert_results_match, scalers_equal
from oracle import oracle as O
from pllamd import api, driver, workload as W

CASES = [
    dict(states=4, tips=16, sites=333, attributes=0, seed=11),
    dict(states=4, tips=16, sites=333, attributes=api.PATTERN_TIP, seed=12, ambiguity_pct=10),
    dict(states=4, tips=32, sites=500, attributes=api.SITE_REPEATS, seed=13, mutate_pct=8),
    dict(states=4, tips=16, sites=100, rate_cats=16, seed=14),
    dict(states=20, tips=16, sites=77, attributes=api.PATTERN_TIP | api.RATE_SCALERS, seed=15),
    dict(states=20, tips=150, sites=40, tree="caterpillar", brlen_scale=3, attributes=api.RATE_SCALERS, seed=16),
    dict(states=7, tips=8, sites=50, attributes=api.SITE_REPEATS, seed=17),
    dict(states=61, tips=8, sites=20, seed=18),
    dict(states=4, tips=200, sites=64, tree="caterpillar", brlen_scale=4, pinv=0.25, mutate_pct=2, seed=19),
]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: f"s{k['states']}-t{k['tips']}-a{k.get('attributes', 0)}")
@pytest.mark.parametrize("arch", [api.ARCH_CPU, api.ARCH_AVX2], ids=["cpu", "avx2"])
def test_restatement_matches_reference(ref_lib, kw, arch):
    case = W.make_case("rnd", **kw)
    exp = driver.run_case(ref_lib, case, arch)
    got = O.run_case(case)
    assert_results_match(got, exp, rtol=1e-12, what=str(kw))
    assert scalers_equal(got, exp)


def test_repeat_classes_match_reference(ref_lib):
    """integer bookkeeping: class ids are bit-identical to src/repeats.c:334-347"""
    case = W.make_case("rep", 4, 16, 400, attributes=api.SITE_REPEATS, mutate_pct=6, seed=3)
    with driver.Session(ref_lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        rep = s.part.repeats.contents
        for (pc, _, c1, _, _, c2, _, _) in case.op_batches[0]:
            ids_p = rep.pernode_ids[pc]
            if not ids_p:
                continue
            l = api.as_np(rep.pernode_site_id[c1], case.sites, np.uint32)
            r = api.as_np(rep.pernode_site_id[c2], case.sites, np.uint32)
            sid, ids = O.repeat_classes(l, rep.pernode_ids[c1], r, rep.pernode_ids[c2])
            assert len(ids) == ids_p
            assert (sid == api.as_np(rep.pernode_site_id[pc], case.sites, np.uint32)).all()
            assert (ids == api.as_np(rep.pernode_id_site[pc], ids_p, np.uint32)).all()


@pytest.mark.parametrize("states,attrs,pinv", [(4, 0, (0, 0, 0, 0)), (4, api.PATTERN_TIP | api.RATE_SCALERS, (0.1, 0.0, 0.3, 0.2)),
                                               (20, 0, (0.0, 0.2, 0.0, 0.1))])
def test_restatement_matches_reference_on_mixtures(ref_lib, states, attrs, pinv):
    """per-category model indices (freqs_indices selects frequencies / prop_invar per rate category,
    src/core_likelihood.c:1421,1442) with non-uniform category weights; asc-bias cases as well"""
    from test_gpu_mixture import mixture_case
    case, _, _, _ = mixture_case(states, 16, 200, seed=400 + states, attributes=attrs, pinv=pinv)
    exp = driver.run_case(ref_lib, case, api.ARCH_AVX2)
    got = O.run_case(case)
    assert_results_match(got, exp, rtol=1e-12, what="mixture")


@pytest.mark.parametrize("kw", [dict(states=4, tips=16, sites=150, asc_type=1, seed=31),
                                dict(states=4, tips=64, sites=64, tree="caterpillar", brlen_scale=4, asc_type=2, asc_weights=[3, 1, 4, 1], seed=32),
                                dict(states=20, tips=8, sites=60, asc_type=3, asc_weights=list(range(2, 22)), attributes=api.PATTERN_TIP, seed=33)],
                         ids=lambda k: f"s{k['states']}-asc{k['asc_type']}")
def test_restatement_matches_reference_with_ascertainment_bias(ref_lib, kw):
    case = W.make_case("asc", **kw)
    exp = driver.run_case(ref_lib, case, api.ARCH_AVX2)
    got = O.run_case(case)
    assert_results_match(got, exp, rtol=1e-12, what=str(kw))
