"""GPU: transition matrices formed on the device by pll_update_prob_matrices (SURVEY section 8
row f2, k_pmatrix) against the reference's own pll_update_prob_matrices, and their use by the hot
path without ever crossing PCIe."""
import numpy as np
import pytest

from pllamd import api, driver, workload as W

pytestmark = pytest.mark.gpu


def _partition(lib, states, exch, freqs, cats, nmat, arch, pinv, rates):
    p = lib.pll_partition_create(2, 1, states, 16, 1, nmat, cats, 0, arch)
    assert p, lib.errmsg()
    lib.pll_set_frequencies(p, 0, api.dptr(np.ascontiguousarray(freqs, dtype=np.float64)))
    lib.pll_set_subst_params(p, 0, api.dptr(np.ascontiguousarray(exch, dtype=np.float64)))
    p.contents.prop_invar[0] = pinv
    if lib.is_amd:
        lib.pll_gpu_invalidate(p, api.DIRTY_INVARIANT, -1)
    lib.pll_set_category_rates(p, api.dptr(np.ascontiguousarray(rates, dtype=np.float64)))
    return p


def _read(lib, p, n, cats, states):
    part = p.contents
    sp = part.states_padded
    if lib.is_amd:
        assert lib.pll_gpu_sync_pmatrix(p, -1), lib.errmsg()
    return np.stack([api.as_np(part.pmatrix[i], cats * states * sp, np.float64).reshape(cats, states, sp)[:, :, :states].copy()
                     for i in range(n)])


@pytest.mark.parametrize("states,arch,cats,pinv", [(4, api.ARCH_AVX2, 4, 0.1), (4, api.ARCH_CPU, 1, 0.0), (7, api.ARCH_AVX2, 3, 0.0),
                                                   (20, api.ARCH_CPU, 4, 0.25), (20, api.ARCH_AVX2, 8, 0.0), (61, api.ARCH_AVX2, 4, 0.1),
                                                   (64, api.ARCH_SSE, 2, 0.0)])
def test_device_prob_matrices_match_reference(amd_lib, ref_lib, states, arch, cats, pinv):
    exch, freqs = (W.GTR_DNA["exch"], W.GTR_DNA["freqs"]) if states == 4 else W.synthetic_exch(states)
    brlens = np.array([0.0, 1e-9, 1e-6, 0.05, 0.5, 3.0, 40.0])
    rates = W.gamma_rates_mean(0.7, cats)
    out = {}
    for lib in (ref_lib, amd_lib):
        p = _partition(lib, states, exch, freqs, cats, len(brlens), arch, pinv, rates)
        pi = np.zeros(cats, dtype=np.uint32)
        # matrices out of order, one of them written twice with DIFFERENT lengths: the last one must win (the reference
        # forms them one after another; a tree search hands over such lists - tests/test_gpu_tree_search.py)
        mi = np.array([6, 5, 4, 3, 2, 1, 0, 3], dtype=np.uint32)
        bl = np.concatenate([brlens[::-1], [brlens[3]]])
        bl[3] = 1.234
        assert lib.pll_update_prob_matrices(p, api.uptr(pi), api.uptr(mi), api.dptr(np.ascontiguousarray(bl)), len(mi)), lib.errmsg()
        out[lib.is_amd] = _read(lib, p, len(brlens), cats, states)
        assert p.contents.eigen_decomp_valid[0] == 1
        lib.pll_partition_destroy(p)
    a, b = out[True], out[False]
    assert np.allclose(a.sum(-1), 1.0, atol=1e-12)
    assert np.max(np.abs(a - b)) < 1e-12
    assert (a[0] == np.eye(states)[None]).all()


def test_device_matrices_feed_the_hot_path(amd_lib, ref_lib):
    """full sequence through the model API on both libraries: same lnL; then a branch-length change
    re-evaluated without touching the host copy of the matrices"""
    case = W.make_case("pm", 20, 16, 200, seed=91)
    rates = case.model["rates"]
    nmat = case.prob_matrices
    brl = W.branch_lengths(nmat)
    vals = {}
    for lib in (ref_lib, amd_lib):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            s.set_model(case.model["exch"], case.freqs, rates)
            pi = np.zeros(case.rate_cats, dtype=np.uint32)
            mi = np.arange(nmat, dtype=np.uint32)
            assert lib.pll_update_prob_matrices(s.p, api.uptr(pi), api.uptr(mi), api.dptr(np.ascontiguousarray(brl)), nmat)
            s.update_partials()
            v0, _ = s.edge_lnl(case.edges[0], persite=False)
            # lengthen the evaluated branch only
            m = np.array([case.edges[0][4]], dtype=np.uint32)
            assert lib.pll_update_prob_matrices(s.p, api.uptr(pi), api.uptr(m), api.dptr(np.array([0.77])), 1)
            v1, _ = s.edge_lnl(case.edges[0], persite=False)
            vals[lib.is_amd] = (v0, v1)
    for g, e in zip(vals[True], vals[False]):
        assert abs(g - e) <= 1e-10 * abs(e)
    assert vals[True][0] != vals[True][1]


def test_host_written_matrix_overrides_device_one(amd_lib):
    """a caller may still write partition->pmatrix and invalidate: the host copy wins again"""
    case = W.make_case("pm2", 4, 8, 100, seed=92)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.set_model(case.model["exch"], case.freqs, case.model["rates"])
        s.update_partials()
        ref0, _ = s.edge_lnl(case.edges[0], persite=False)
        mi = case.edges[0][4]
        pi = np.zeros(4, dtype=np.uint32)
        assert amd_lib.pll_update_prob_matrices(s.p, api.uptr(pi), api.uptr(np.array([mi], dtype=np.uint32)), api.dptr(np.array([5.0])), 1)
        v1, _ = s.edge_lnl(case.edges[0], persite=False)
        assert abs(v1 - ref0) > 1e-3
        # put the original matrix back through the host array
        sp = s.sp
        dst = api.as_np(s.part.pmatrix[mi], 4 * 4 * sp, np.float64).reshape(4, 4, sp)
        dst[:, :, :4] = case.pmatrix[mi]
        amd_lib.pll_gpu_invalidate(s.p, api.DIRTY_PMATRIX, mi)
        v2, _ = s.edge_lnl(case.edges[0], persite=False)
        assert abs(v2 - ref0) <= 1e-12 * abs(ref0)


@pytest.mark.parametrize("how", ["setter", "direct-write", "direct-write-reform"])
def test_stale_matrix_with_a_revalidated_eigensystem_keeps_the_orientation(amd_lib, ref_lib, how):
    """a matrix formed by pll_update_prob_matrices with frequencies A, then the frequencies of the
    same set become B and the eigensystem is valid again (another matrix was formed, or the caller wrote
    p->frequencies and invalidated - which never clears eigen_decomp_valid). The stale matrix is reversible
    for A, the evaluation weighs with B: swapping the ends of a tip-parent edge would change the value, so the
    caller's orientation must be evaluated (src/likelihood.c:626-634 - the reference never swaps without
    PLL_ATTRIB_PATTERN_TIP). The set's version counter, not the current eigen_decomp_valid flag, decides.
    ADVICE r3 ("direct-write-reform"): after the direct write the caller forms matrix m AGAIN - eigen_decomp_valid was
    never cleared, so both libraries form it from the OLD eigensystem (reversible for A) and it carries the set's NEW
    version; the eigensystem is foreign to its set until pll_update_eigen recomputes it, so no swap either."""
    case = W.make_case("orient-stale", 4, 12, 500, tree="caterpillar", seed=62, ambiguity_pct=5)
    p, ps, c, cs, m = case.edges[0]
    assert c < case.tips  # the child end of the caterpillar's root edge is a tip
    tip_parent = (c, cs, p, ps, m)
    nmat = case.prob_matrices
    brl = np.ascontiguousarray(W.branch_lengths(nmat))
    fb = np.array([0.05, 0.45, 0.1, 0.4])
    vals = {}
    for lib in (ref_lib, amd_lib):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            s.set_model(case.model["exch"], case.freqs, case.model["rates"])
            pi = np.zeros(case.rate_cats, dtype=np.uint32)
            mi = np.arange(nmat, dtype=np.uint32)
            assert lib.pll_update_prob_matrices(s.p, api.uptr(pi), api.uptr(mi), api.dptr(brl), nmat)
            s.update_partials()
            before = s.edge_lnl(tip_parent, persite=False)[0]
            if how == "setter":
                lib.pll_set_frequencies(s.p, 0, api.dptr(fb))
                other = np.array([(m + 1) % nmat], dtype=np.uint32)  # revalidates the eigensystem; matrix m stays stale
                assert lib.pll_update_prob_matrices(s.p, api.uptr(pi), api.uptr(other), api.dptr(np.array([0.3])), 1)
                assert s.part.eigen_decomp_valid[0] == 1
            else:
                api.as_np(s.part.frequencies[0], 4, np.float64)[:] = fb
                if lib.is_amd:
                    lib.pll_gpu_invalidate(s.p, api.DIRTY_FREQS, 0)
                if how == "direct-write-reform":
                    assert s.part.eigen_decomp_valid[0] == 1
                    again = np.array([m], dtype=np.uint32)
                    assert lib.pll_update_prob_matrices(s.p, api.uptr(pi), api.uptr(again), api.dptr(brl[m:m + 1].copy()), 1)
            vals[lib.is_amd] = (before, s.edge_lnl(tip_parent, persite=False)[0], s.edge_lnl((p, ps, c, cs, m), persite=False)[0])
    (b0, g_tp, g_usual), (r0, e_tp, e_usual) = vals[True], vals[False]
    assert abs(b0 - r0) <= 1e-10 * abs(r0)
    assert abs(e_tp - e_usual) > 1e-6 * abs(e_usual)  # the orientation matters for the stale matrix
    assert abs(g_tp - e_tp) <= 1e-10 * abs(e_tp) and abs(g_usual - e_usual) <= 1e-10 * abs(e_usual)
