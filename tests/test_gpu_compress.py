"""GPU: site-pattern compression on the device (csrc/hip/compress.hip)
against the reference's pll_compress_site_patterns[_msa] (src/compress.c:171-410): compressed
sequences, weights, new length and the site -> pattern map must be identical."""
import ctypes as C

import numpy as np
import pytest

from pllamd import api, workload as W

pytestmark = pytest.mark.gpu


def _run(lib, seqs, cmap, msa):
    n, length = len(seqs), len(seqs[0])
    bufs = [C.create_string_buffer(s, length + 1) for s in seqs]
    arr = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    m = (C.c_ulonglong * 256)(*[int(v) for v in cmap])
    smap = None
    if msa:
        st = api.Msa(count=n, length=length, sequence=arr, label=None)
        smap = np.full(length, 0xFFFFFFFF, dtype=np.uint32)
        w = lib.pll_compress_site_patterns_msa(C.byref(st), m, api.uptr(smap))
        newlen = st.length
    else:
        ln = C.c_int(length)
        w = lib.pll_compress_site_patterns(arr, m, n, C.byref(ln))
        newlen = ln.value
    if not w:
        return None, lib.errno(), None, None
    weights = api.as_np(w, newlen, np.uint32).copy()
    out = [b.value for b in bufs]
    assert all(len(o) == newlen for o in out)
    return out, newlen, weights, smap


def _alignment(states, tips, sites, seed, mutate_pct, ambiguity_pct=0, partial_pct=0):
    tree = "balanced" if tips & (tips - 1) == 0 else "caterpillar"
    case = W.make_case("cmp", states, tips, sites, seed=seed, mutate_pct=mutate_pct, ambiguity_pct=ambiguity_pct,
                       partial_pct=partial_pct, tree=tree)
    return case.sequences, case.charmap


@pytest.mark.parametrize("states,tips,sites,mut,amb", [(4, 8, 500, 5, 5), (4, 64, 20000, 3, 2), (4, 130, 3000, 2, 0), (20, 16, 4000, 2, 3),
                                                       (20, 33, 1000, 6, 0), (7, 8, 300, 20, 5), (61, 16, 2000, 1, 2), (4, 4, 17, 50, 0)])
@pytest.mark.parametrize("msa", [False, True], ids=["plain", "msa"])
def test_compression_matches_reference(amd_lib, ref_lib, states, tips, sites, mut, amb, msa):
    seqs, cmap = _alignment(states, tips, sites, seed=states * 1000 + tips, mutate_pct=mut, ambiguity_pct=amb, partial_pct=amb)
    a = _run(amd_lib, seqs, cmap, msa)
    b = _run(ref_lib, seqs, cmap, msa)
    assert a[1] == b[1] and a[1] < sites, (a[1], b[1])
    assert a[0] == b[0]
    assert (a[2] == b[2]).all() and int(a[2].sum()) == sites
    if msa:
        assert (a[3] == b[3]).all()


def test_codes_with_the_sign_bit(amd_lib, ref_lib):
    """a map whose states are byte values >= 128: the reference compares encoded characters as
    (signed) char, so those sort before the small ones"""
    cmap = np.zeros(256, dtype=np.uint64)
    for ch, v in zip(b"ABCDEF", (1, 2, 200, 130, 7, 255)):
        cmap[ch] = v
    rng = np.random.default_rng(5)
    seqs = [bytes(rng.choice(list(b"ABCDEF"), size=400).astype(np.uint8)) for _ in range(5)]
    a = _run(amd_lib, seqs, cmap, True)
    b = _run(ref_lib, seqs, cmap, True)
    assert a[0] == b[0] and (a[2] == b[2]).all() and (a[3] == b[3]).all()


def test_compression_errors(amd_lib, ref_lib):
    seqs, cmap = _alignment(4, 4, 50, seed=1, mutate_pct=10)
    bad = [seqs[0], seqs[1][:10] + b"!" + seqs[1][11:], seqs[2], seqs[3]]
    for lib in (amd_lib, ref_lib):
        out, code, _, _ = _run(lib, bad, cmap, False)
        assert out is None and code == 114  # PLL_ERROR_TIPDATA_ILLEGALSTATE
        zero = cmap.copy()
        zero[0] = 1
        out, code, _, _ = _run(lib, seqs, zero, False)
        assert out is None and code == 132  # PLL_ERROR_MSA_MAP_INVALID


def test_compressed_alignment_gives_the_same_lnl(amd_lib):
    """end to end: lnL of the compressed alignment with its weights == lnL of the original"""
    from pllamd import driver
    case = W.make_case("cmp_lnl", 4, 16, 3000, seed=77, mutate_pct=3)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        full, _ = s.edge_lnl(case.edges[0], persite=False)
    out, newlen, weights, _ = _run(amd_lib, case.sequences, case.charmap, False)
    small = driver.Case(name="cmp_small", states=4, rate_cats=4, tips=16, sites=newlen, pmatrix=case.pmatrix, freqs=case.freqs,
                        op_batches=case.op_batches, edges=case.edges, charmap=case.charmap, sequences=out,
                        clv_buffers=case.clv_buffers, scale_buffers=case.scale_buffers, pattern_weights=weights)
    with driver.Session(amd_lib, small, api.ARCH_AVX2) as s:
        s.update_partials()
        comp, _ = s.edge_lnl(small.edges[0], persite=False)
    assert newlen < 3000 and abs(comp - full) <= 1e-10 * abs(full)
