"""CPU (no GPU): the C-ABI library loads, exports every symbol its headers declare, lays its
structs out like the reference, and its host-side bookkeeping (tip encodings, site-repeats class
maps, model matrices, invariant sites, error behaviour) matches the reference. Partitions are
created as host-only shells (PLL_AMD_HOST_ONLY=1): no compute call is made - and the compute entry
points are checked to FAIL LOUDLY rather than fall back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pllamd import api, driver, workload as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def host_only(monkeypatch):
    monkeypatch.setenv("PLL_AMD_HOST_ONLY", "1")


def declared_symbols():
    names = set()
    for hdr in ("pll_amd.h", "pll_amd_device.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(pll(?:gpu)?_[a-z0-9_]+)\s*\(", txt))
        names |= set(re.findall(r"extern\s+[^;]*?\b(pll_[a-z0-9_]+)\s*(?:\[|;)", txt))
    names -= {"pll_state_t", "pll_partition", "pll_repeats", "pll_operation"}
    return sorted(names)


def test_exports_every_declared_symbol(amd_lib):
    missing = []
    for name in declared_symbols():
        try:
            getattr(amd_lib.dll, name)
        except AttributeError:
            missing.append(name)
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    assert len(declared_symbols()) > 70


def test_no_cpu_fallback_compute_fails_loudly(amd_lib, capfd):
    case = W.make_case("t", 4, 4, 32)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        assert amd_lib.errno() == 900  # PLL_ERROR_GPU_UNAVAILABLE
        v, _ = s.edge_lnl(case.edges[0])
        assert v == -np.inf and amd_lib.errno() == 900
        assert not amd_lib.pll_gpu_sync_clv(s.p, 5)
    err = capfd.readouterr().err
    assert "no MI355X context" in err


def test_create_without_device_is_refused(amd_lib, monkeypatch):
    if amd_lib.pll_gpu_device_count() > 0:
        pytest.skip("a GPU is present")
    monkeypatch.delenv("PLL_AMD_HOST_ONLY")
    p = amd_lib.pll_partition_create(4, 2, 4, 10, 1, 5, 4, 2, api.ARCH_AVX2)
    assert not p
    assert amd_lib.errno() == 900 and "MI355X" in amd_lib.errmsg()


def test_invalid_attribute_combinations(amd_lib):
    assert not amd_lib.pll_partition_create(4, 2, 4, 100, 1, 5, 4, 2, api.ARCH_AVX | api.ARCH_SSE)
    assert amd_lib.errno() == 113
    assert not amd_lib.pll_partition_create(4, 2, 4, 100, 1, 5, 4, 2, api.PATTERN_TIP | api.SITE_REPEATS)
    assert amd_lib.errno() == 113
    # the reference's repeats update never computes the per-state extra entries: refused
    assert not amd_lib.pll_partition_create(4, 2, 4, 100, 1, 5, 4, 2, api.AB_LEWIS | api.AB_FLAG | api.SITE_REPEATS)
    assert amd_lib.errno() == 902
    assert not amd_lib.pll_partition_create(4, 2, 4, 100, 1, 5, 4, 2, 5 << 5)
    assert amd_lib.errno() == 121


def test_asc_bias_partition_bookkeeping(amd_lib, ref_lib):
    """extra per-state entries: allocation sizes, weights, tip codes / tip CLVs, setter errors -
    same observable state as the reference (src/pll.c:525-531, 822-824, 897-905, 935-953,
    1003-1021, 1145-1200; src/models.c:500-508)"""
    nt = (C.c_ulonglong * 256)(*[int(v) for v in W.map_nt()])
    for attr in (api.AB_FLAG, api.AB_FLAG | api.PATTERN_TIP):
        st = []
        for lib in (amd_lib, ref_lib):
            p = lib.pll_partition_create(4, 2, 4, 20, 1, 5, 2, 2, api.ARCH_AVX2 | attr)
            assert p
            part = p.contents
            assert (part.asc_bias_alloc, part.asc_additional_sites) == (1, 4)
            assert lib.pll_set_tip_states(p, 1, nt, b"ACGTACGTAC-NRYACGTAC")
            w = api.as_np(part.pattern_weights, 24, np.uint32).copy()
            assert list(w[20:]) == [0, 0, 0, 0]
            sw = np.array([5, 4, 6, 2], dtype=np.uint32)
            lib.pll_set_asc_state_weights(p, api.uptr(sw))
            assert list(api.as_np(part.pattern_weights, 24, np.uint32)[20:]) == [5, 4, 6, 2]
            assert part.pattern_weight_sum == 20
            assert lib.pll_get_sites_number(p, 4) == 24
            if attr & api.PATTERN_TIP:
                obs = bytes(api.as_np(part.tipchars[1], 24, np.uint8))
            else:
                obs = api.as_np(part.clv[1], 24 * 2 * 4, np.float64).copy().tobytes()
            assert lib.pll_set_asc_bias_type(p, api.AB_FELSENSTEIN)
            assert part.attributes & (7 << 5) == api.AB_FELSENSTEIN
            assert not lib.pll_update_invariant_sites_proportion(p, 0, 0.3) and lib.errno() == 117
            assert not lib.pll_set_asc_bias_type(p, api.AB_LEWIS | 1) and lib.errno() == 121
            assert lib.pll_set_asc_bias_type(p, 0) and part.attributes & (7 << 5) == 0
            st.append(obs)
            lib.pll_partition_destroy(p)
        assert st[0] == st[1]
    for lib in (amd_lib, ref_lib):
        p = lib.pll_partition_create(4, 2, 4, 20, 1, 5, 2, 2, api.ARCH_AVX2)
        assert not lib.pll_set_asc_bias_type(p, api.AB_LEWIS) and lib.errno() == 122
        lib.pll_partition_destroy(p)


@pytest.mark.parametrize("states,arch,sp,align", [(4, 0, 4, 8), (5, api.ARCH_SSE, 6, 16), (7, api.ARCH_AVX, 8, 32),
                                                    (20, api.ARCH_AVX2, 20, 32), (61, api.ARCH_AVX2, 64, 32),
                                                    (61, 0, 61, 8)])
def test_layout_follows_arch_bits(amd_lib, ref_lib, states, arch, sp, align):
    for lib in (amd_lib, ref_lib):
        p = lib.pll_partition_create(5, 3, states, 20, 2, 7, 4, 3, arch | api.RATE_SCALERS)
        part = p.contents
        assert (part.states_padded, part.alignment) == (sp, align)
        assert (part.nodes, part.pattern_weight_sum) == (8, 20)
        assert part.pmatrix[1] and (C.addressof(part.pmatrix[1].contents) - C.addressof(part.pmatrix[0].contents)) == 8 * 4 * states * sp
        assert list(api.as_np(part.rate_weights, 4, np.float64)) == [0.25] * 4
        assert list(api.as_np(part.pattern_weights, 20, np.uint32)) == [1] * 20
        assert C.addressof(part.clv[7].contents) % align == 0
        lib.pll_partition_destroy(p)


def test_repeats_disabled_below_16_sites(amd_lib):
    p = amd_lib.pll_partition_create(4, 2, 4, 15, 1, 5, 4, 2, api.SITE_REPEATS)
    assert p and not amd_lib.pll_repeats_enabled(p) and not p.contents.repeats
    amd_lib.pll_partition_destroy(p)


@pytest.mark.parametrize("states", [4, 7, 20, 61])
def test_tip_codes_match_reference(amd_lib, ref_lib, states):
    case = W.make_case("t", states, 8, 80, attributes=api.PATTERN_TIP, ambiguity_pct=15, seed=states)
    got = {}
    for lib in (amd_lib, ref_lib):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            part = s.part
            got[lib.is_amd] = (part.maxstates, [api.as_np(part.tipchars[t], 80, np.uint8).copy() for t in range(8)],
                               api.as_np(part.tipmap, 256, np.uint64).copy(), api.as_np(part.charmap, 256, np.uint8).copy(),
                               [bool(part.clv[t]) for t in range(8)])
    a, b = got[True], got[False]
    assert a[0] == b[0]
    assert all((x == y).all() for x, y in zip(a[1], b[1]))
    assert (a[2] == b[2]).all() and (a[3] == b[3]).all() and a[4] == b[4] == [False] * 8


def test_second_map_extends_code_table(amd_lib, ref_lib):
    """src/pll.c:157-286: a later sequence with another map adds codes without disturbing old ones"""
    m1 = W.map_generic(7)
    m2 = m1.copy()
    m2[ord("z")] = 0x3 | 0x40
    m2[ord("y")] = 0x11
    out = {}
    for lib in (amd_lib, ref_lib):
        p = lib.pll_partition_create(3, 1, 7, 16, 1, 3, 2, 0, api.PATTERN_TIP)
        a1 = (C.c_ulonglong * 256)(*[int(v) for v in m1])
        a2 = (C.c_ulonglong * 256)(*[int(v) for v in m2])
        assert lib.pll_set_tip_states(p, 0, a1, b"0123456-01234560")
        assert lib.pll_set_tip_states(p, 1, a2, b"zy23456-0123456z")
        part = p.contents
        out[lib.is_amd] = (part.maxstates, api.as_np(part.tipchars[0], 16, np.uint8).copy(),
                           api.as_np(part.tipchars[1], 16, np.uint8).copy(), api.as_np(part.tipmap, 16, np.uint64).copy())
        lib.pll_partition_destroy(p)
    assert out[True][0] == out[False][0]
    for i in (1, 2, 3):
        assert (out[True][i] == out[False][i]).all()


def test_illegal_state_is_reported(amd_lib):
    p = amd_lib.pll_partition_create(2, 1, 4, 8, 1, 1, 1, 0, 0)
    assert not amd_lib.pll_set_tip_states(p, 0, amd_lib.state_map("pll_map_nt"), b"ACGT!CGT")
    assert amd_lib.errno() == 114
    amd_lib.pll_partition_destroy(p)
    p = amd_lib.pll_partition_create(2, 1, 4, 8, 1, 1, 1, 0, api.PATTERN_TIP)
    clv = np.zeros(32)
    assert not amd_lib.pll_set_tip_clv(p, 0, api.dptr(clv), 0)
    assert amd_lib.errno() == 115
    amd_lib.pll_partition_destroy(p)


@pytest.mark.parametrize("name", ["pll_map_nt", "pll_map_aa", "pll_map_bin", "pll_map_gt10", "pll_map_gt16"])
def test_state_maps_equal_reference(amd_lib, ref_lib, name):
    assert list(amd_lib.state_map(name)) == list(ref_lib.state_map(name))


def test_tip_clvs_match_reference(amd_lib, ref_lib):
    case = W.make_case("t", 7, 8, 33, ambiguity_pct=20)
    clv = {}
    for lib in (amd_lib, ref_lib):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            n = 33 * 4 * s.sp
            clv[lib.is_amd] = [api.as_np(s.part.clv[t], n, np.float64).reshape(33, 4, s.sp)[:, :, :7].copy() for t in range(8)]
    assert all((a == b).all() for a, b in zip(clv[True], clv[False]))


@pytest.mark.parametrize("states,mut", [(4, 5), (4, 40), (20, 5)])
def test_repeat_class_maps_match_reference(amd_lib, ref_lib, states, mut):
    """integer work: bit-exact (src/repeats.c:189-254, :299-382), including the fall-back to
    uncompressed nodes when compression does not pay"""
    case = W.make_case("t", states, 16, 300, attributes=api.SITE_REPEATS, mutate_pct=mut, seed=7)
    ops = api.make_ops(case.op_batches[0])
    res = {}
    for lib in (amd_lib, ref_lib):
        with driver.Session(lib, case, api.ARCH_AVX2) as s:
            for i in range(len(case.op_batches[0])):
                lib.pll_update_repeats(s.p, C.byref(ops[i]))
            rep = s.part.repeats.contents
            rows = []
            for node in range(s.part.nodes):
                ids = rep.pernode_ids[node]
                rows.append((ids, api.as_np(rep.pernode_site_id[node], 300, np.uint32).copy() if ids else None,
                             api.as_np(rep.pernode_id_site[node], ids, np.uint32).copy() if ids else None,
                             lib.pll_get_sites_number(s.p, node), lib.pll_get_clv_size(s.p, node)))
            res[lib.is_amd] = rows
    for a, b in zip(res[True], res[False]):
        assert a[0] == b[0] and a[3] == b[3] and a[4] == b[4]
        if a[0]:
            assert (a[1] == b[1]).all() and (a[2] == b[2]).all()
    assert any(r[0] for r in res[True][16:]), "no inner node was compressed - test is vacuous"


def test_invariant_sites_match_reference(amd_lib, ref_lib):
    for attr in (0, api.PATTERN_TIP):
        case = W.make_case("t", 4, 8, 200, attributes=attr, mutate_pct=3, ambiguity_pct=10, pinv=0.2)
        inv = {}
        for lib in (amd_lib, ref_lib):
            with driver.Session(lib, case, api.ARCH_AVX2) as s:
                inv[lib.is_amd] = api.as_np(s.part.invariant, 200, np.int32).copy()
                assert abs(s.part.prop_invar[0] - 0.2) < 1e-15
        assert (inv[True] == inv[False]).all() and (inv[True] >= 0).any() and (inv[True] < 0).any()


def test_count_invariant_sites_matches_reference(amd_lib, ref_lib):
    """src/models.c:546-649: with and without a stored invariant array, weighted total and per-state
    pattern counts (4 states with tip codes and tip CLVs, 20 states with tip CLVs)"""
    rng = np.random.default_rng(3)
    for states, attr, pinv in ((4, 0, 0.0), (4, api.PATTERN_TIP, 0.0), (4, 0, 0.2), (20, 0, 0.0)):
        w = rng.integers(1, 9, 200).astype(np.uint32)
        case = W.make_case("t", states, 8, 200, attributes=attr, mutate_pct=3, ambiguity_pct=10, pinv=pinv, pattern_weights=w)
        got = {}
        for lib in (amd_lib, ref_lib):
            with driver.Session(lib, case, api.ARCH_AVX2) as s:
                per = np.zeros(states, dtype=np.uint32)
                total = lib.pll_count_invariant_sites(s.p, api.uptr(per))
                got[lib.is_amd] = (total, per.copy(), lib.pll_count_invariant_sites(s.p, None))
        assert got[True][0] == got[False][0] > 0 and got[True][2] == got[False][2]
        assert (got[True][1] == got[False][1]).all()


def model_pmatrices(lib, states, exch, freqs, alpha, cats, brlens, arch, pinv=0.0):
    p = lib.pll_partition_create(2, 1, states, 16, 1, len(brlens), cats, 0, arch)
    part = p.contents
    sp = part.states_padded
    rates = np.zeros(cats)
    assert lib.pll_compute_gamma_cats(alpha, cats, api.dptr(rates), 0)
    f = np.ascontiguousarray(freqs, dtype=np.float64)
    e = np.ascontiguousarray(exch, dtype=np.float64)
    lib.pll_set_frequencies(p, 0, api.dptr(f))
    lib.pll_set_subst_params(p, 0, api.dptr(e))
    part.prop_invar[0] = pinv
    bl = np.ascontiguousarray(brlens, dtype=np.float64)
    return p, part, sp, rates, bl


def _p_from_eigen(part, states, sp, rates, brlens, pinv):
    """P(t) = I + Vinv' diag(expm1(lambda r t / (1 - pinv))) V from the partition's eigen arrays
    (src/core_pmatrix.c:186-232)"""
    ev = api.as_np(part.eigenvecs[0], states * sp, np.float64).reshape(states, sp)[:, :states]
    iev = api.as_np(part.inv_eigenvecs[0], states * sp, np.float64).reshape(states, sp)[:, :states]
    lam = api.as_np(part.eigenvals[0], states, np.float64)
    out = np.zeros((len(brlens), len(rates), states, states))
    for b, t in enumerate(brlens):
        for k, r in enumerate(rates):
            out[b, k] = np.eye(states) + (iev * np.expm1(lam * r * t / (1.0 - pinv))[None, :]) @ ev if t > 0 else np.eye(states)
    return out


@pytest.mark.parametrize("states,arch", [(4, api.ARCH_AVX2), (7, api.ARCH_AVX2), (20, api.ARCH_CPU), (61, api.ARCH_AVX2)])
def test_eigensystem_reproduces_reference_prob_matrices(amd_lib, ref_lib, states, arch):
    """the eigen decomposition stays on the host (pll_update_eigen); the transition matrices it
    implies must agree with the reference's pll_update_prob_matrices to 1e-12. (The matrices
    themselves are formed on the device: tests/test_gpu_models.py.)"""
    exch, freqs = (W.GTR_DNA["exch"], W.GTR_DNA["freqs"]) if states == 4 else W.synthetic_exch(states)
    brlens = [0.0, 1e-6, 0.05, 0.5, 3.0]
    p, part, sp, rates, bl = model_pmatrices(ref_lib, states, exch, freqs, 0.7, 4, brlens, arch, pinv=0.1)
    ref_lib.pll_set_category_rates(p, api.dptr(rates))
    pi = np.zeros(4, dtype=np.uint32)
    mi = np.arange(len(brlens), dtype=np.uint32)
    assert ref_lib.pll_update_prob_matrices(p, api.uptr(pi), api.uptr(mi), api.dptr(bl), len(brlens))
    exp = np.stack([api.as_np(part.pmatrix[i], 4 * states * sp, np.float64).reshape(4, states, sp)[:, :, :states].copy()
                    for i in range(len(brlens))])
    ref_lib.pll_partition_destroy(p)
    p, part, sp, _, bl = model_pmatrices(amd_lib, states, exch, freqs, 0.7, 4, brlens, arch, pinv=0.1)
    assert amd_lib.pll_update_eigen(p, 0) and part.eigen_decomp_valid[0] == 1
    got = _p_from_eigen(part, states, sp, rates, brlens, 0.1)
    amd_lib.pll_partition_destroy(p)
    assert np.allclose(got.sum(-1), 1.0, atol=1e-12)
    assert np.max(np.abs(got - exp)) < 1e-12


def test_zero_frequency_states_are_dropped(amd_lib, ref_lib):
    """src/models.c:254-291,346-385"""
    freqs = np.array([0.4, 0.0, 0.35, 0.25])
    p, part, sp, rates, bl = model_pmatrices(ref_lib, 4, [1, 2, 3, 4, 5, 6], freqs, 1.0, 2, [0.3], api.ARCH_AVX2)
    ref_lib.pll_set_category_rates(p, api.dptr(np.array([0.5, 1.5])))
    assert ref_lib.pll_update_prob_matrices(p, api.uptr(np.zeros(2, dtype=np.uint32)), api.uptr(np.zeros(1, dtype=np.uint32)), api.dptr(bl), 1)
    exp = api.as_np(part.pmatrix[0], 2 * 4 * sp, np.float64).reshape(2, 4, sp)[:, :, :4].copy()
    ref_lib.pll_partition_destroy(p)
    p, part, sp, _, bl = model_pmatrices(amd_lib, 4, [1, 2, 3, 4, 5, 6], freqs, 1.0, 2, [0.3], api.ARCH_AVX2)
    assert amd_lib.pll_update_eigen(p, 0)
    got = _p_from_eigen(part, 4, sp, [0.5, 1.5], [0.3], 0.0)[0]
    amd_lib.pll_partition_destroy(p)
    assert np.max(np.abs(got - exp)) < 1e-12


def test_prob_matrices_need_the_device(amd_lib):
    """pll_update_prob_matrices forms the matrices on the MI355X; a host-only shell refuses loudly"""
    p, part, sp, rates, bl = model_pmatrices(amd_lib, 4, W.GTR_DNA["exch"], W.GTR_DNA["freqs"], 0.7, 4, [0.1], api.ARCH_AVX2)
    amd_lib.pll_set_category_rates(p, api.dptr(rates))
    ok = amd_lib.pll_update_prob_matrices(p, api.uptr(np.zeros(4, dtype=np.uint32)), api.uptr(np.zeros(1, dtype=np.uint32)), api.dptr(bl), 1)
    assert not ok and amd_lib.errno() == 900
    amd_lib.pll_partition_destroy(p)


@pytest.mark.parametrize("alpha", [0.05, 0.5, 1.0, 4.2, 50.0])
@pytest.mark.parametrize("cats", [1, 2, 4, 8, 16])
def test_gamma_categories(amd_lib, ref_lib, alpha, cats):
    """mean/median category rates: the reference's incomplete-gamma routine stops at 1e-8
    (src/gamma.c:47), ours does not - agreement is to 1e-6, and against scipy to 1e-12"""
    for mode in (0, 1):
        a = np.zeros(cats)
        b = np.zeros(cats)
        assert amd_lib.pll_compute_gamma_cats(alpha, cats, api.dptr(a), mode)
        assert ref_lib.pll_compute_gamma_cats(alpha, cats, api.dptr(b), mode)
        assert np.allclose(a, b, rtol=2e-6, atol=1e-12)
        assert abs(a.mean() - 1.0) < 1e-12
    a = np.zeros(cats)
    amd_lib.pll_compute_gamma_cats(alpha, cats, api.dptr(a), 0)
    assert np.allclose(a, W.gamma_rates_mean(alpha, cats), rtol=1e-11, atol=1e-14)


def test_gamma_rejects_bad_alpha(amd_lib):
    a = np.zeros(4)
    assert not amd_lib.pll_compute_gamma_cats(0.0, 4, api.dptr(a), 0)
    assert amd_lib.errno() == 113


def test_frequencies_are_normalised_like_reference(amd_lib, ref_lib):
    for f in ([0.1, 0.2, 0.3, 0.4], [1.0, 2.0, 3.0, 4.0], [0.25, 0.25, 0.25, 0.25 + 5e-9]):
        got = []
        for lib in (amd_lib, ref_lib):
            p = lib.pll_partition_create(2, 1, 4, 16, 1, 1, 1, 0, 0)
            lib.pll_set_frequencies(p, 0, api.dptr(np.array(f)))
            got.append(api.as_np(p.contents.frequencies[0], 4, np.float64).copy())
            lib.pll_partition_destroy(p)
        assert (got[0] == got[1]).all()


def test_caller_built_against_the_reference_header_links(tmp_path):
    """authoring container only: tests/c_caller/dropin.c compiled against the REFERENCE's pll.h links
    against libpll_amd.so with no unresolved symbol - the drop-in claim at the C level (it runs on
    the GPU box built against include/pll_amd.h: tests/test_gpu_c_caller.py)"""
    import subprocess
    ref_inc = "/root/reference/src"
    if not os.path.exists(os.path.join(ref_inc, "pll.h")):
        pytest.skip("reference header not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "libpll-2_amd", "csrc")
    exe = str(tmp_path / "dropin_ref")
    subprocess.check_call(["gcc", "-O2", "-DUSE_REFERENCE_HEADER", "-I" + ref_inc, os.path.join(root, "tests", "c_caller", "dropin.c"),
                           "-L" + libdir, "-lpll_amd", "-lm", "-Wl,-rpath," + libdir, "-Wl,--no-undefined", "-o", exe])
    assert os.path.getsize(exe) > 0


def test_repeats_scaler_utilities(amd_lib):
    """pll_fill_parent_scaler_repeats[_per_rate] (src/repeats.c:392-540): integer gather-adds on host
    arrays, against the definition - and against the reference binary where it is present"""
    import ctypes as C
    import os
    rng = np.random.default_rng(4)
    n, nl, nr, npar, rates = 200, 30, 45, 80, 4
    U = C.POINTER(C.c_uint)
    up = lambda a: None if a is None else a.ctypes.data_as(U)  # noqa: E731
    lids = np.ascontiguousarray(rng.integers(0, nl, size=n).astype(np.uint32))
    rids = np.ascontiguousarray(rng.integers(0, nr, size=n).astype(np.uint32))
    psites = np.ascontiguousarray(rng.integers(0, n, size=npar).astype(np.uint32))
    libs = [amd_lib]
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libpll_ref.so")
    if os.path.exists(ref):
        libs.append(api.PllLib(ref))
    for per_rate in (1, rates):
        ls = np.ascontiguousarray(rng.integers(0, 5, size=(nl, per_rate)).astype(np.uint32))
        rs = np.ascontiguousarray(rng.integers(0, 5, size=(nr, per_rate)).astype(np.uint32))
        for use_ps, use_l, use_r in ((True, True, True), (True, True, False), (True, False, True), (False, True, True)):
            ps = psites if use_ps else None
            m = npar if use_ps else n
            site = psites if use_ps else np.arange(n)
            want = (ls[lids[site]] if use_l else 0) + (rs[rids[site]] if use_r else 0)
            for lib in libs:
                out = np.full((m, per_rate), 77, dtype=np.uint32)
                if per_rate == 1:
                    f = lib.dll.pll_fill_parent_scaler_repeats
                    f.restype = None
                    f.argtypes = [C.c_uint, U, U, U, U, U, U]
                    f(m, up(out), up(ps), up(ls if use_l else None), up(lids), up(rs if use_r else None), up(rids))
                else:
                    f = lib.dll.pll_fill_parent_scaler_repeats_per_rate
                    f.restype = None
                    f.argtypes = [C.c_uint, C.c_uint, U, U, U, U, U, U]
                    f(m, per_rate, up(out), up(ps), up(ls if use_l else None), up(lids), up(rs if use_r else None), up(rids))
                assert (out == want).all(), (per_rate, use_ps, use_l, use_r)


def test_hardware_record(amd_lib):
    """pll_hardware (src/pll.h:220-237,555): 48 bytes, probe / ignore behave like the reference's"""
    import ctypes as C

    class HW(C.Structure):
        _fields_ = [(k, C.c_int) for k in ("init", "altivec", "mmx", "sse", "sse2", "sse3", "ssse3", "sse41", "sse42", "popcnt", "avx", "avx2")]

    assert C.sizeof(HW) == 48
    amd_lib.dll.pll_hardware_probe.restype = C.c_int
    assert amd_lib.dll.pll_hardware_probe() == 1
    hw = HW.in_dll(amd_lib.dll, "pll_hardware")
    assert hw.init == 1 and hw.sse2 == 1  # every x86-64 host has SSE2
    amd_lib.dll.pll_hardware_ignore()
    assert all(getattr(hw, k) == 1 for k, _ in HW._fields_)


def test_collective_entry_points_fail_loudly_without_a_device(amd_lib, capfd):
    """round 4: the set-up of the RCCL form and the sharded derivative evaluation on a host-only shell - an error and
    the reference's failure values, never a wait for a collective that cannot happen"""
    import ctypes as C
    case = W.make_case("t", 4, 4, 32)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        fake_comm = C.c_void_p(0x1234)
        assert not amd_lib.pll_gpu_allreduce_prepare(s.p, fake_comm)
        assert amd_lib.errno() == 900 and "no MI355X context" in amd_lib.errmsg()
        assert not amd_lib.pll_gpu_allreduce_prepare(s.p, None)
        fi = np.zeros(4, dtype=np.uint32)
        v = amd_lib.pll_gpu_edge_loglikelihood_allreduce(s.p, fake_comm, case.edges[0][0], case.edges[0][1], case.edges[0][2],
                                                          case.edges[0][3], case.edges[0][4], api.uptr(fi))
        assert v == -np.inf and amd_lib.errno() == 900
        st = np.zeros(32 * 4 * 4)
        d1, d2 = C.c_double(7.0), C.c_double(7.0)
        ok = amd_lib.pll_gpu_group_likelihood_derivatives(s.p, None, -1, -1, 0.1, api.uptr(fi), api.dptr(st), C.byref(d1), C.byref(d2))
        assert not ok and amd_lib.errno() == 900 and d1.value == 7.0  # outputs untouched on failure
    assert "no MI355X context" in capfd.readouterr().err
