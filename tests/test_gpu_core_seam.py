"""GPU: the flat pll_core_* seam (include/pll_amd.h; reference src/pll.h:1049-1177, :1295-1414) against
the same functions of the reference library (oracle/_ref, built from the reference's sources in the
authoring container and shipped as a binary) on seeded raw arrays: CLVs within 1e-10 relative, scaler
vectors exact, log-likelihoods within 1e-10 relative. Skipped where the reference binary is absent."""
import ctypes as C

import numpy as np
import pytest

from pllamd import api, workload as W

pytestmark = pytest.mark.gpu
RTOL = 1e-10
D = C.POINTER(C.c_double)
U = C.POINTER(C.c_uint)
B = C.POINTER(C.c_ubyte)
I = C.POINTER(C.c_int)
S64 = C.POINTER(C.c_ulonglong)
DP = C.POINTER(D)


def aligned(a):
    """a copy of `a` on a 64-byte boundary (the reference's AVX kernels use aligned loads)"""
    a = np.ascontiguousarray(a)
    raw = np.zeros(a.nbytes + 64, dtype=np.uint8)
    off = (-raw.ctypes.data) % 64
    out = raw[off:off + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def dp(a):
    return None if a is None else a.ctypes.data_as(D)


def up(a):
    return None if a is None else a.ctypes.data_as(U)


def bp(a):
    return a.ctypes.data_as(B)


def sp_of(states, arch):
    return {api.ARCH_CPU: states, api.ARCH_SSE: (states + 1) & ~1, api.ARCH_AVX: (states + 3) & ~3, api.ARCH_AVX2: (states + 3) & ~3}[arch]


def rand_clv(rng, entries, rates, states, sp, scale=1.0):
    a = np.zeros((entries, rates, sp))
    a[:, :, :states] = rng.random((entries, rates, states)) * scale + 1e-3 * scale
    return aligned(a)


def pmat(states, rates, sp, t, seed):
    exch, freqs = (W.GTR_DNA["exch"], W.GTR_DNA["freqs"]) if states == 4 else W.synthetic_exch(states)
    r = W.gamma_rates_mean(0.7, rates)
    pm = W.pmatrices(exch, np.asarray(freqs, dtype=np.float64), r, np.array([t]), 0.0)  # [1][rate][state][state]
    out = np.zeros((rates, states, sp))
    out[:, :, :states] = pm[0]
    return aligned(out), np.asarray(freqs, dtype=np.float64)


def freq_ptrs(freqs, sp, sets):
    rows = [aligned(np.concatenate([freqs, np.zeros(sp - len(freqs))])) for _ in range(sets)]
    arr = (D * sets)(*[r.ctypes.data_as(D) for r in rows])
    return arr, rows


def both(amd_lib, ref_lib, name):
    return getattr(amd_lib.dll, name), getattr(ref_lib.dll, name)


@pytest.mark.parametrize("states,rates,arch,per_rate", [(4, 4, api.ARCH_AVX2, False), (4, 4, api.ARCH_CPU, True), (20, 4, api.ARCH_AVX2, False),
                                                        (5, 2, api.ARCH_SSE, False), (7, 3, api.ARCH_AVX, True)])
def test_update_partial_ii_and_edge(amd_lib, ref_lib, states, rates, arch, per_rate):
    rng = np.random.default_rng(states * 100 + rates)
    n = 333
    sp = sp_of(states, arch)
    attrib = arch | (api.RATE_SCALERS if per_rate else 0)
    sw = rates if per_rate else 1
    # small values so that some sites rescale
    l = rand_clv(rng, n, rates, states, sp, 1e-40)
    r = rand_clv(rng, n, rates, states, sp, 1e-40)
    l[::7] *= 1e-30
    r[::7] *= 1e-30
    lm, freqs = pmat(states, rates, sp, 0.1, 1)
    rm, _ = pmat(states, rates, sp, 0.23, 2)
    ls = np.ascontiguousarray(rng.integers(0, 3, size=(n, sw)).astype(np.uint32))
    rs = np.ascontiguousarray(rng.integers(0, 3, size=(n, sw)).astype(np.uint32))
    res = {}
    for tag, lib in (("amd", amd_lib), ("ref", ref_lib)):
        f = lib.dll.pll_core_update_partial_ii
        f.restype = None
        f.argtypes = [C.c_uint] * 3 + [D, U, D, D, D, D, U, U, C.c_uint]
        pc = aligned(np.zeros((n, rates, sp)))
        ps = np.zeros((n, sw), dtype=np.uint32)
        f(states, n, rates, dp(pc), up(ps), dp(l), dp(r), dp(lm), dp(rm), up(ls), up(rs), attrib)
        res[tag] = (pc, ps)
    assert (res["amd"][1] == res["ref"][1]).all()
    assert res["ref"][1].sum() > ls.sum() + rs.sum()  # something was rescaled
    np.testing.assert_allclose(res["amd"][0][..., :states], res["ref"][0][..., :states], rtol=RTOL, atol=0)  # padding entries: anything
    # edge log-likelihood between the parent just formed and a fresh child
    pc, ps = res["ref"]
    child = rand_clv(rng, n, rates, states, sp)
    cs = np.ascontiguousarray(rng.integers(0, 2, size=(n, sw)).astype(np.uint32))
    fp, keep = freq_ptrs(freqs, sp, 2)
    rw = rng.random(rates) + 0.1
    rw = aligned(rw / rw.sum())
    pw = np.ascontiguousarray(rng.integers(1, 5, size=n).astype(np.uint32))
    pinv = np.ascontiguousarray(np.array([0.0, 0.0]))
    fi = np.ascontiguousarray((np.arange(rates) % 2).astype(np.uint32))
    out = {}
    for tag, lib in (("amd", amd_lib), ("ref", ref_lib)):
        f = lib.dll.pll_core_edge_loglikelihood_ii
        f.restype = C.c_double
        f.argtypes = [C.c_uint] * 3 + [D, U, D, U, D, DP, D, U, D, I, U, D, C.c_uint]
        per = aligned(np.zeros(n))
        v = f(states, n, rates, dp(pc), up(ps), dp(child), up(cs), dp(lm), fp, dp(rw), up(pw), dp(pinv), None, up(fi), dp(per), attrib)
        out[tag] = (v, per)
    assert abs(out["amd"][0] - out["ref"][0]) <= RTOL * abs(out["ref"][0])
    np.testing.assert_allclose(out["amd"][1], out["ref"][1], rtol=RTOL, atol=0)
    # root: against the reference and against the formula itself (src/core_likelihood.c:163-207). The
    # reference's SSE root kernel is not a usable yardstick for every shape: with an odd number of states or
    # with rate categories that use different frequency sets it returns NaN or values that disagree with its
    # own generic kernel (5 states: NaN; 6 states, two sets: -37.36 against -35.11 on a 50-site check) - there
    # the formula alone decides.
    scal = None if per_rate else cs
    for tag, lib in (("amd", amd_lib), ("ref", ref_lib)):
        f = lib.dll.pll_core_root_loglikelihood
        f.restype = C.c_double
        f.argtypes = [C.c_uint] * 3 + [D, U, DP, D, U, D, I, U, D, C.c_uint]
        out[tag] = f(states, n, rates, dp(child), up(scal), fp, dp(rw), up(pw), dp(pinv), None, up(fi), None, attrib)
    site = np.log(((child[:, :, :states] * freqs[None, None, :]).sum(axis=2) * rw[None, :]).sum(axis=1))
    if scal is not None:
        site = site + scal.ravel() * np.log(2.0 ** -256)
    manual = float((site * pw).sum())
    assert abs(out["amd"] - manual) <= 1e-12 * abs(manual)
    if arch != api.ARCH_SSE:
        assert abs(out["amd"] - out["ref"]) <= RTOL * abs(out["ref"])


@pytest.mark.parametrize("states,arch", [(4, api.ARCH_AVX2), (20, api.ARCH_AVX2), (7, api.ARCH_CPU)])
def test_tip_forms(amd_lib, ref_lib, states, arch):
    """ti / tt / edge ti with encoded tip characters and a tipmap (4 states: the 4x4 forms, code = mask)"""
    rng = np.random.default_rng(states)
    n, rates = 257, 4
    sp = sp_of(states, arch)
    attrib = arch
    nchar = 16 if states == 4 else states + 3
    tipmap = np.zeros(256, dtype=np.uint64)
    if states == 4:
        tipmap[:16] = np.arange(16)
        codes = rng.integers(1, 16, size=(2, n)).astype(np.uint8)
    else:
        for c in range(states):
            tipmap[c] = 1 << c
        tipmap[states] = (1 << states) - 1
        tipmap[states + 1] = 0b101
        tipmap[states + 2] = 0b11000
        codes = rng.integers(0, nchar, size=(2, n)).astype(np.uint8)
    codes = np.ascontiguousarray(codes)
    lm, freqs = pmat(states, rates, sp, 0.15, 3)
    rm, _ = pmat(states, rates, sp, 0.3, 4)
    inner = rand_clv(rng, n, rates, states, sp)
    isc = np.ascontiguousarray(rng.integers(0, 2, size=n).astype(np.uint32))
    tm = tipmap.ctypes.data_as(S64)
    res = {}
    for tag, lib in (("amd", amd_lib), ("ref", ref_lib)):
        d = lib.dll
        # ti
        pc = aligned(np.zeros((n, rates, sp)))
        ps = np.zeros(n, dtype=np.uint32)
        if states == 4:
            d.pll_core_update_partial_ti_4x4.restype = None
            d.pll_core_update_partial_ti_4x4.argtypes = [C.c_uint] * 2 + [D, U, B, D, D, D, U, C.c_uint]
            d.pll_core_update_partial_ti_4x4(n, rates, dp(pc), up(ps), bp(codes[0]), dp(inner), dp(lm), dp(rm), up(isc), attrib)
        else:
            d.pll_core_update_partial_ti.restype = None
            d.pll_core_update_partial_ti.argtypes = [C.c_uint] * 3 + [D, U, B, D, D, D, U, S64, C.c_uint, C.c_uint]
            d.pll_core_update_partial_ti(states, n, rates, dp(pc), up(ps), bp(codes[0]), dp(inner), dp(lm), dp(rm), up(isc), tm, nchar, attrib)
        # tt through the lookup pair
        pow2 = 1 << int(np.ceil(np.log2(nchar)))  # the reference addresses its table with shifts (src/core_partials.c:1149-1209)
        look = aligned(np.zeros(max(1024 * rates, pow2 * pow2 * rates * sp)))
        tc = aligned(np.zeros((n, rates, sp)))
        ts = np.zeros(n, dtype=np.uint32)
        if states == 4:
            d.pll_core_create_lookup_4x4.restype = None
            d.pll_core_create_lookup_4x4.argtypes = [C.c_uint, D, D, D]
            d.pll_core_create_lookup_4x4(rates, dp(look), dp(lm), dp(rm))
            d.pll_core_update_partial_tt_4x4.restype = None
            d.pll_core_update_partial_tt_4x4.argtypes = [C.c_uint] * 2 + [D, U, B, B, D, C.c_uint]
            d.pll_core_update_partial_tt_4x4(n, rates, dp(tc), up(ts), bp(codes[0]), bp(codes[1]), dp(look), attrib)
        else:
            d.pll_core_create_lookup.restype = None
            d.pll_core_create_lookup.argtypes = [C.c_uint, C.c_uint, D, D, D, S64, C.c_uint, C.c_uint]
            d.pll_core_create_lookup(states, rates, dp(look), dp(lm), dp(rm), tm, nchar, attrib)
            d.pll_core_update_partial_tt.restype = None
            d.pll_core_update_partial_tt.argtypes = [C.c_uint] * 3 + [D, U, B, B, S64, C.c_uint, D, C.c_uint]
            d.pll_core_update_partial_tt(states, n, rates, dp(tc), up(ts), bp(codes[0]), bp(codes[1]), tm, nchar, dp(look), attrib)
        # edge ti
        fp, keep = freq_ptrs(freqs, sp, 1)
        rw = aligned(np.full(rates, 1.0 / rates))
        pw = np.ones(n, dtype=np.uint32)
        pinv = np.zeros(1)
        fi = np.zeros(rates, dtype=np.uint32)
        if states == 4:
            d.pll_core_edge_loglikelihood_ti_4x4.restype = C.c_double
            d.pll_core_edge_loglikelihood_ti_4x4.argtypes = [C.c_uint] * 2 + [D, U, B, D, DP, D, U, D, I, U, D, C.c_uint]
            v = d.pll_core_edge_loglikelihood_ti_4x4(n, rates, dp(inner), up(isc), bp(codes[1]), dp(lm), fp, dp(rw), up(pw), dp(pinv), None, up(fi), None, attrib)
        else:
            d.pll_core_edge_loglikelihood_ti.restype = C.c_double
            d.pll_core_edge_loglikelihood_ti.argtypes = [C.c_uint] * 3 + [D, U, B, S64, C.c_uint, D, DP, D, U, D, I, U, D, C.c_uint]
            v = d.pll_core_edge_loglikelihood_ti(states, n, rates, dp(inner), up(isc), bp(codes[1]), tm, nchar, dp(lm), fp, dp(rw), up(pw), dp(pinv), None, up(fi), None, attrib)
        res[tag] = (pc, ps, tc, ts, v, look)
    a, r = res["amd"], res["ref"]
    # the lookup table itself is the reference's (a caller may inspect it, or hand a table made by one
    # library to the other's tt): same entries at the same places ...
    used = pow2 * pow2 * rates * sp if states != 4 else 256 * rates * 4
    stride = rates * sp if states != 4 else rates * 4
    tab_a, tab_r = a[5][:used].reshape(-1, stride), r[5][:used].reshape(-1, stride)
    idx = [(j << int(np.ceil(np.log2(nchar)))) + k if states != 4 else 16 * j + k
           for j in range(nchar if states != 4 else 16) for k in range(nchar if states != 4 else 16)]
    lanes = np.arange(stride).reshape(rates, -1)[:, :states].ravel()
    np.testing.assert_allclose(tab_a[idx][:, lanes], tab_r[idx][:, lanes], rtol=1e-13, atol=0)
    # ... and each library's tt reads the other's table
    for lib, table, want in ((amd_lib, r[5], r[2]), (ref_lib, a[5], a[2])):
        d = lib.dll
        tc = aligned(np.zeros((n, rates, sp)))
        ts = np.ones(n, dtype=np.uint32)
        if states == 4:
            d.pll_core_update_partial_tt_4x4(n, rates, dp(tc), up(ts), bp(codes[0]), bp(codes[1]), dp(table), attrib)
        else:
            d.pll_core_update_partial_tt(states, n, rates, dp(tc), up(ts), bp(codes[0]), bp(codes[1]), tm, nchar, dp(table), attrib)
        np.testing.assert_allclose(tc[..., :states], want[..., :states], rtol=1e-13, atol=0)
        assert (ts == 0).all()
    np.testing.assert_allclose(a[0][..., :states], r[0][..., :states], rtol=RTOL, atol=0)
    assert (a[1] == r[1]).all()
    np.testing.assert_allclose(a[2][..., :states], r[2][..., :states], rtol=RTOL, atol=0)
    assert (a[3] == r[3]).all()
    assert abs(a[4] - r[4]) <= RTOL * abs(r[4])


def test_repeats_forms(amd_lib, ref_lib):
    """class-compressed operands addressed through id_site / site_id maps"""
    rng = np.random.default_rng(9)
    states, rates, arch = 4, 4, api.ARCH_AVX2
    sp = 4
    n, nl, nr, npar = 300, 40, 55, 120
    lsid = np.ascontiguousarray(rng.integers(0, nl, size=n).astype(np.uint32))
    rsid = np.ascontiguousarray(rng.integers(0, nr, size=n).astype(np.uint32))
    pids = np.ascontiguousarray(np.sort(rng.choice(n, size=npar, replace=False)).astype(np.uint32))
    psid = np.ascontiguousarray(rng.integers(0, npar, size=n).astype(np.uint32))
    l = rand_clv(rng, nl, rates, states, sp)
    r = rand_clv(rng, nr, rates, states, sp)
    ls = np.ascontiguousarray(rng.integers(0, 2, size=nl).astype(np.uint32))
    rs = np.ascontiguousarray(rng.integers(0, 2, size=nr).astype(np.uint32))
    lm, freqs = pmat(states, rates, sp, 0.1, 5)
    rm, _ = pmat(states, rates, sp, 0.2, 6)
    res = {}
    for tag, lib in (("amd", amd_lib), ("ref", ref_lib)):
        d = lib.dll
        f = d.pll_core_update_partial_repeats
        f.restype = None
        f.argtypes = [C.c_uint] * 5 + [D, U, D, D, D, D, U, U, U, U, U, D, C.c_uint]
        pc = aligned(np.zeros((npar, rates, sp)))
        ps = np.zeros(npar, dtype=np.uint32)
        bclv = aligned(np.zeros((nl, rates, sp)))
        f(states, npar, nl, nr, rates, dp(pc), up(ps), dp(l), dp(r), dp(lm), dp(rm), up(ls), up(rs), up(pids), up(lsid), up(rsid), dp(bclv), arch)
        g = d.pll_core_edge_loglikelihood_repeats
        g.restype = C.c_double
        g.argtypes = [C.c_uint] * 4 + [D, U, D, U, D, DP, D, U, D, I, U, D, U, U, D, C.c_uint]
        fp, keep = freq_ptrs(freqs, sp, 1)
        rw = aligned(np.full(rates, 0.25))
        pw = np.ascontiguousarray(rng.integers(1, 3, size=n).astype(np.uint32)) if tag == "amd" else res["pw"]
        res["pw"] = pw
        pinv = np.zeros(1)
        fi = np.zeros(rates, dtype=np.uint32)
        v = g(states, n, nl, rates, dp(pc), up(ps), dp(l), up(ls), dp(lm), fp, dp(rw), up(pw), dp(pinv), None, up(fi), None, up(psid), up(lsid), dp(bclv), arch)
        res[tag] = (pc, ps, v)
    np.testing.assert_allclose(res["amd"][0], res["ref"][0], rtol=RTOL, atol=0)
    assert (res["amd"][1] == res["ref"][1]).all()
    assert abs(res["amd"][2] - res["ref"][2]) <= RTOL * abs(res["ref"][2])


# ---- flat derivative / transition-matrix functions (reference src/pll.h:1181-1273, :2400-2412) ------
def padded_rows(mat, sp):
    out = np.zeros((mat.shape[0], sp))
    out[:, :mat.shape[1]] = mat
    return aligned(out)


def eigen_arrays(states, sp, sets, seed=0):
    """per set: (eigenvecs, inv_eigenvecs, eigenvals, freqs) in the partition's array conventions"""
    rows = []
    for s in range(sets):
        exch, freqs = (W.GTR_DNA["exch"], W.GTR_DNA["freqs"]) if states == 4 else W.synthetic_exch(states)
        freqs = np.asarray(freqs, dtype=np.float64)
        if s:  # a different model per set
            rng = np.random.default_rng(seed + s)
            freqs = freqs * (0.5 + rng.random(states))
            freqs /= freqs.sum()
        e = W.eigensystem(exch, freqs)
        rows.append((padded_rows(e["eigenvecs"][0], sp), padded_rows(e["inv_eigenvecs"][0], sp),
                     aligned(np.concatenate([e["eigenvals"][0], np.zeros(sp - states)])),
                     aligned(np.concatenate([freqs, np.zeros(sp - states)]))))
    return rows


def ptr_array(rows, which, index):
    return (D * len(index))(*[rows[i][which].ctypes.data_as(D) for i in index])


@pytest.mark.parametrize("states,rates,arch,per_rate", [(4, 4, api.ARCH_AVX2, False), (4, 4, api.ARCH_CPU, True),
                                                       (20, 4, api.ARCH_AVX2, False), (7, 3, api.ARCH_CPU, False)])
def test_flat_sumtable_and_derivatives(amd_lib, ref_lib, states, rates, arch, per_rate):
    rng = np.random.default_rng(77 + states)
    n = 257
    sp = sp_of(states, arch)
    attrib = arch | (api.RATE_SCALERS if per_rate else 0)
    sw = rates if per_rate else 1
    rows = eigen_arrays(states, sp, rates)          # one model per rate category, as the flat functions take them
    idx = list(range(rates))
    evec, ievec, evals, fr = (ptr_array(rows, k, idx) for k in range(4))
    pclv = rand_clv(rng, n, rates, states, sp)
    cclv = rand_clv(rng, n, rates, states, sp)
    psc = np.ascontiguousarray(rng.integers(0, 3, size=(n, sw)).astype(np.uint32))
    csc = np.ascontiguousarray(rng.integers(0, 3, size=(n, sw)).astype(np.uint32))
    tipmap = np.array([(1 << (c % states)) | (1 << ((c * 3 + 1) % states)) for c in range(256)], dtype=np.uint64)
    if states == 4:
        tipmap = np.arange(256, dtype=np.uint64) & 15
        tipmap[tipmap == 0] = 15
    chars = np.ascontiguousarray(rng.integers(1, 16 if states == 4 else 200, size=n).astype(np.uint8))
    # class-compressed operands
    pid = np.ascontiguousarray(rng.integers(0, 40, size=n).astype(np.uint32)); pid[:40] = np.arange(40)
    cid = np.ascontiguousarray(rng.integers(0, 90, size=n).astype(np.uint32)); cid[:90] = np.arange(90)
    weights = np.ascontiguousarray(rng.integers(1, 5, size=n).astype(np.uint32))
    rate_w = aligned(np.full(rates, 1.0 / rates))
    cat_rates = aligned(W.gamma_rates_mean(0.6, rates))
    pinv = aligned(np.full(rates, 0.15))
    invariant = np.ascontiguousarray(np.where(rng.random(n) < 0.3, rng.integers(0, states, size=n), -1).astype(np.int32))

    tables = {}
    for tag, lib in (("amd", amd_lib), ("ref", ref_lib)):
        t = {}
        f = lib.dll.pll_core_update_sumtable_ii
        f.restype = C.c_int
        f.argtypes = [C.c_uint] * 3 + [D, D, U, U, DP, DP, DP, D, C.c_uint]
        t["ii"] = aligned(np.zeros((n, rates, sp)))
        assert f(states, n, rates, dp(pclv), dp(cclv), up(psc), up(csc), evec, ievec, fr, dp(t["ii"]), attrib)
        f = lib.dll.pll_core_update_sumtable_ti
        f.restype = C.c_int
        f.argtypes = [C.c_uint] * 3 + [D, B, U, DP, DP, DP, S64, C.c_uint, D, C.c_uint]
        t["ti"] = aligned(np.zeros((n, rates, sp)))
        assert f(states, n, rates, dp(pclv), bp(chars), up(psc), evec, ievec, fr, tipmap.ctypes.data_as(S64), 256,
                 dp(t["ti"]), attrib)
        t["ti4"] = aligned(np.zeros((n, rates, sp)))
        if states == 4:
            f = lib.dll.pll_core_update_sumtable_ti_4x4
            f.restype = C.c_int
            f.argtypes = [C.c_uint] * 2 + [D, B, U, DP, DP, DP, D, C.c_uint]
            assert f(n, rates, dp(pclv), bp(chars), up(psc), evec, ievec, fr, dp(t["ti4"]), attrib)
        f = lib.dll.pll_core_update_sumtable_repeats_generic
        f.restype = C.c_int
        f.argtypes = [C.c_uint] * 4 + [D, D, U, U, DP, DP, DP, D, U, U, D, C.c_uint, C.c_uint]
        t["rep"] = aligned(np.zeros((n, rates, sp)))
        # the generic form reads operands with the unpadded span (src/core_derivatives.c:241-242): CPU layout only
        if sp == states:
            assert f(states, n, 40, rates, dp(pclv), dp(cclv), up(psc), up(csc), evec, ievec, fr, dp(t["rep"]),
                     up(pid), up(cid), None, 0, attrib)
        tables[tag] = t
    if states == 4:
        assert (tables["ref"]["ti4"] == tables["ref"]["ti"]).all()
    for key in ("ii", "ti", "ti4", "rep"):
        a, r = tables["amd"][key][..., :states], tables["ref"][key][..., :states]
        assert np.allclose(a, r, rtol=RTOL, atol=RTOL * np.abs(r).max()), key

    # derivatives from the REFERENCE's table (a table this library did not produce is uploaded as it is)
    table = tables["ref"]["ii"]
    for t_len in (0.05, 0.7):
        out = {}
        for tag, lib in (("amd", amd_lib), ("ref", ref_lib)):
            f = lib.dll.pll_core_likelihood_derivatives
            f.restype = C.c_int
            f.argtypes = [C.c_uint] * 3 + [D, U, U, C.c_uint, C.c_uint, I, U, C.c_double, D, DP, D, DP, D, D, D, C.c_uint]
            d1, d2 = C.c_double(0), C.c_double(0)
            assert f(states, n, rates, dp(rate_w), up(psc), up(csc), n, n, invariant.ctypes.data_as(I), up(weights), t_len,
                     dp(pinv), fr, dp(cat_rates), evals, dp(table), C.byref(d1), C.byref(d2), attrib)
            out[tag] = (d1.value, d2.value)
        assert out["amd"][0] == pytest.approx(out["ref"][0], rel=1e-9, abs=1e-9)
        assert out["amd"][1] == pytest.approx(out["ref"][1], rel=1e-9, abs=1e-9)


@pytest.mark.parametrize("states,rates,arch", [(4, 4, api.ARCH_AVX2), (20, 4, api.ARCH_AVX), (7, 3, api.ARCH_CPU)])
def test_flat_update_pmatrix(amd_lib, ref_lib, states, rates, arch):
    sp = sp_of(states, arch)
    rows = eigen_arrays(states, sp, 2)
    params = np.ascontiguousarray(np.array([k % 2 for k in range(rates)], dtype=np.uint32))   # two models, mixed over the rates
    evec, ievec, evals = (ptr_array(rows, k, [0, 1]) for k in range(3))
    cat_rates = aligned(W.gamma_rates_mean(0.8, rates))
    pinv = aligned(np.array([0.0, 0.2]))
    matrix_indices = np.ascontiguousarray(np.array([3, 0, 5], dtype=np.uint32))
    lengths = aligned(np.array([0.01, 0.3, 2.5]))
    res = {}
    for tag, lib in (("amd", amd_lib), ("ref", ref_lib)):
        f = lib.dll.pll_core_update_pmatrix
        f.restype = C.c_int
        f.argtypes = [DP, C.c_uint, C.c_uint, D, D, U, U, D, DP, DP, DP, C.c_uint, C.c_uint]
        mats = [aligned(np.full((rates, states, sp), -1.0)) for _ in range(6)]
        arr = (D * 6)(*[m.ctypes.data_as(D) for m in mats])
        assert f(arr, states, rates, dp(cat_rates), dp(lengths), up(matrix_indices), up(params), dp(pinv), evals, evec, ievec,
                 3, arch)
        res[tag] = mats
    for i in range(6):
        a, r = res["amd"][i][..., :states], res["ref"][i][..., :states]
        if i in (0, 3, 5):
            assert np.allclose(a, r, rtol=RTOL, atol=1e-14), i
            assert np.allclose(r.sum(axis=-1), 1.0, atol=1e-9)
        else:
            assert (a == -1.0).all() and (r == -1.0).all()   # untouched


def test_root_loglikelihood_repeats_generic(amd_lib, ref_lib):
    """the undeclared export of src/core_likelihood.c:211-223 (unpadded layout, no attrib argument)"""
    rng = np.random.default_rng(5)
    states, rates, n, classes = 7, 3, 200, 60
    clv = rand_clv(rng, classes, rates, states, states)
    site_id = np.ascontiguousarray(rng.integers(0, classes, size=n).astype(np.uint32))
    scaler = np.ascontiguousarray(rng.integers(0, 2, size=classes).astype(np.uint32))
    _, freqs = W.synthetic_exch(states)
    fr, keep = freq_ptrs(np.asarray(freqs, dtype=np.float64), states, 1)
    rate_w = aligned(np.full(rates, 1.0 / rates))
    weights = np.ascontiguousarray(rng.integers(1, 4, size=n).astype(np.uint32))
    fidx = np.zeros(rates, dtype=np.uint32)
    out = {}
    for tag, lib in (("amd", amd_lib), ("ref", ref_lib)):
        f = lib.dll.pll_core_root_loglikelihood_repeats_generic
        f.restype = C.c_double
        f.argtypes = [C.c_uint] * 3 + [D, U, U, DP, D, U, D, I, U, D]
        per = aligned(np.zeros(n))
        out[tag] = (f(states, n, rates, dp(clv), up(site_id), up(scaler), fr, dp(rate_w), up(weights), None, None, up(fidx), dp(per)), per)
    assert out["amd"][0] == pytest.approx(out["ref"][0], rel=RTOL)
    assert np.allclose(out["amd"][1], out["ref"][1], rtol=RTOL, atol=0)


def test_a_loop_over_the_flat_update_reuses_its_partition(amd_lib, ref_lib):
    """a caller that loops over pll_core_update_partial_ii (src/pll.h:1049-1177) must not pay an allocation,
    a stream and their release per call: the partition of a shape is kept per thread (core_seam.c: seam_open). 1000
    calls at 1k sites, the last one's result against the reference's, the mean time per call, and results that do
    not depend on what an earlier call of the same shape left behind (pattern weights, invariant sites)."""
    import time
    states, rates, n, arch = 4, 4, 1000, api.ARCH_AVX2
    sp = sp_of(states, arch)
    rng = np.random.default_rng(5)
    lm, freqs = pmat(states, rates, sp, 0.1, 1)
    rm, _ = pmat(states, rates, sp, 0.23, 2)
    f = amd_lib.dll.pll_core_update_partial_ii
    f.restype = None
    f.argtypes = [C.c_uint] * 3 + [D, U, D, D, D, D, U, U, C.c_uint]
    pc = aligned(np.zeros((n, rates, sp)))
    ps = np.zeros((n, 1), dtype=np.uint32)
    l = rand_clv(rng, n, rates, states, sp)
    r = rand_clv(rng, n, rates, states, sp)
    args = (states, n, rates, dp(pc), up(ps), dp(l), dp(r), dp(lm), dp(rm), None, None, arch)  # (numpy's ctypes views cost a microsecond each: once)
    for _ in range(5):
        f(*args)
    t0 = time.perf_counter()
    for i in range(1000):
        f(*args)
    per_call_us = (time.perf_counter() - t0) / 1000 * 1e6
    g = ref_lib.dll.pll_core_update_partial_ii
    g.restype = None
    g.argtypes = f.argtypes
    pc2 = aligned(np.zeros((n, rates, sp)))
    ps2 = np.zeros((n, 1), dtype=np.uint32)
    g(states, n, rates, dp(pc2), up(ps2), dp(l), dp(r), dp(lm), dp(rm), None, None, arch)
    np.testing.assert_allclose(pc[..., :states], pc2[..., :states], rtol=RTOL, atol=0)
    assert (ps == ps2).all()
    print("pll_core_update_partial_ii, 1k sites, cached partition: %.1f us per call" % per_call_us)
    # (round 5: the call's transfers - two CLVs and two matrices up, CLV and scaler vector back - go through pinned host
    # memory, the layout kernels reading / writing it directly, with ONE wait: 38-42 us per call, 130 with the runtime's
    # pageable copies; without the kept partition a call is the creation and release of a device context on top:
    # measured below through the A/B switch, in a child process - the switch is read once)
    import subprocess, sys, os
    code = ("import os,sys,time,ctypes as C,numpy as np;sys.path[:0]=[%r,%r];import test_gpu_core_seam as T;from pllamd import api;"
            "lib=api.PllLib();f=lib.dll.pll_core_update_partial_ii;f.restype=None;f.argtypes=[C.c_uint]*3+[T.D,T.U,T.D,T.D,T.D,T.D,T.U,T.U,C.c_uint];"
            "rng=np.random.default_rng(5);sp=4;lm,_=T.pmat(4,4,sp,0.1,1);rm,_=T.pmat(4,4,sp,0.23,2);pc=T.aligned(np.zeros((1000,4,sp)));ps=np.zeros((1000,1),dtype=np.uint32);"
            "l=T.rand_clv(rng,1000,4,4,sp);r=T.rand_clv(rng,1000,4,4,sp);\n"
            "for _ in range(3): f(4,1000,4,T.dp(pc),T.up(ps),T.dp(l),T.dp(r),T.dp(lm),T.dp(rm),None,None,api.ARCH_AVX2)\n"
            "t0=time.perf_counter()\n"
            "for _ in range(50): f(4,1000,4,T.dp(pc),T.up(ps),T.dp(l),T.dp(r),T.dp(lm),T.dp(rm),None,None,api.ARCH_AVX2)\n"
            "print((time.perf_counter()-t0)/50*1e6)") % (os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libpll-2_amd"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PLL_AMD_SEAM_CACHE="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    uncached_us = float(out.stdout.strip().splitlines()[-1])
    print("the same without the kept partition (PLL_AMD_SEAM_CACHE=0): %.1f us per call" % uncached_us)
    assert per_call_us < 70.0 and per_call_us * 3.0 < uncached_us, (per_call_us, uncached_us)
    # an evaluation with pattern weights and invariant sites, then one without: the second must not see the first's
    e = amd_lib.dll.pll_core_edge_loglikelihood_ii
    e.restype = C.c_double
    e.argtypes = [C.c_uint] * 3 + [D, U, D, U, D, DP, D, U, D, I, U, D, C.c_uint]
    fp, keep = freq_ptrs(freqs, sp, 1)
    rw = aligned(np.full(rates, 1.0 / rates))
    fi = np.zeros(rates, dtype=np.uint32)
    ones = np.ones(n, dtype=np.uint32)
    heavy = np.ascontiguousarray(rng.integers(1, 9, size=n).astype(np.uint32))
    inv = np.ascontiguousarray(rng.integers(-1, 4, size=n).astype(np.int32))
    pinv0, pinv = np.zeros(1), np.array([0.3])
    plain = e(states, n, rates, dp(l), None, dp(r), None, dp(lm), fp, dp(rw), up(ones), dp(pinv0), None, up(fi), None, arch)
    other = e(states, n, rates, dp(l), None, dp(r), None, dp(lm), fp, dp(rw), up(heavy), dp(pinv), inv.ctypes.data_as(I), up(fi), None, arch)
    again = e(states, n, rates, dp(l), None, dp(r), None, dp(lm), fp, dp(rw), up(ones), dp(pinv0), None, up(fi), None, arch)
    assert other != plain and again == plain
