"""ctypes front end of the derivative restatement in oracle/pll_oracle.c (orc_update_sumtable,
orc_likelihood_derivatives). TEST INFRASTRUCTURE ONLY, like oracle.py.

`run_derivatives(case, eig, rates, edges, brlens)` evaluates, for every edge, the sumtable and the
(d_f, dd_f) pair at every branch length, from dense CLVs computed by oracle.run_case's machinery.
"""
import ctypes as C

import numpy as np

from . import oracle as O

c_double_p = C.POINTER(C.c_double)
c_uint_p = C.POINTER(C.c_uint)
_bound = False


def _dll():
    global _bound
    d = O.dll()
    if not _bound:
        d.orc_update_sumtable.restype = None
        d.orc_update_sumtable.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.POINTER(O.Child), C.POINTER(O.Child),
                                          C.POINTER(c_double_p), C.POINTER(c_double_p), C.POINTER(c_double_p),
                                          c_double_p, C.c_int]
        d.orc_likelihood_derivatives.restype = None
        d.orc_likelihood_derivatives.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, c_double_p, C.POINTER(C.c_int),
                                                 c_uint_p, C.c_double, c_double_p, C.POINTER(c_double_p), c_double_p,
                                                 C.POINTER(c_double_p), c_double_p, c_double_p, c_double_p]
        d.orc_asc_bias_derivatives.restype = None
        d.orc_asc_bias_derivatives.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, c_uint_p, c_uint_p, c_double_p,
                                               c_uint_p, C.c_uint, C.c_double, c_double_p, c_double_p,
                                               C.POINTER(c_double_p), c_double_p, C.c_int, C.c_int, c_double_p,
                                               c_double_p]
        _bound = True
    return d


def _ptrs(rows):
    return (c_double_p * len(rows))(*[r.ctypes.data_as(c_double_p) for r in rows])


def run_derivatives(case, eig, rates, edges, brlens):
    """eig: dict eigenvecs/inv_eigenvecs [rate_matrices][s][s], eigenvals [rate_matrices][s]
    (reference conventions: eigenvecs[j][i], inv_eigenvecs[i][j]). Returns
    {'sumtable': [per edge [sites][r][s]], 'd': [[(d_f, dd_f) per brlen] per edge]}"""
    from pllamd import api
    d = _dll()
    s, r, n = case.states, case.rate_cats, case.sites
    na = case.entries_alloc  # the table covers the per-state extra entries of an asc-bias partition
    per_rate = bool(case.attributes & api.RATE_SCALERS)
    nodes, scalers = O.dense_nodes(case)
    fi = np.asarray(case.freqs_indices, dtype=np.int64)
    fsum = case.freqs.sum(1, keepdims=True)
    fr = np.where(np.abs(fsum - 1.0) > 1e-8, case.freqs / fsum, case.freqs)
    ev = [np.ascontiguousarray(eig["eigenvecs"][fi[k]]) for k in range(r)]
    iev = [np.ascontiguousarray(eig["inv_eigenvecs"][fi[k]]) for k in range(r)]
    evals = [np.ascontiguousarray(eig["eigenvals"][fi[k]]) for k in range(r)]
    frk = [np.ascontiguousarray(fr[fi[k]]) for k in range(r)]
    pinv = np.ascontiguousarray(np.asarray(case.prop_invar)[fi], dtype=np.float64)
    inv = O.invariant_sites(case) if (pinv > 0).any() else None
    rw = np.ascontiguousarray(case.rate_weights, dtype=np.float64)
    aw = np.ascontiguousarray(case.asc_weights if case.asc_weights is not None else np.zeros(s), dtype=np.uint32)
    pw_sum = int(np.asarray(case.pattern_weights, dtype=np.uint64).sum())
    pw = np.ascontiguousarray(np.concatenate([np.asarray(case.pattern_weights, dtype=np.uint32), aw]), dtype=np.uint32)
    # Stamatakis: the extra entries are ordinary weighted sites (src/core_derivatives.c:733-742)
    ef = n + (s if case.asc_type == 3 else 0)
    rt = np.ascontiguousarray(rates, dtype=np.float64)
    out = {"sumtable": [], "d": []}
    for (pc, psc, cc, csc) in edges:
        a, b = nodes[pc].child(), nodes[cc].child()
        a.scaler = scalers[psc].ctypes.data_as(c_uint_p) if psc >= 0 else None
        b.scaler = scalers[csc].ctypes.data_as(c_uint_p) if csc >= 0 else None
        st = np.zeros((na, r, s))
        d.orc_update_sumtable(s, s, r, na, C.byref(a), C.byref(b), _ptrs(ev), _ptrs(iev), _ptrs(frk),
                              st.ctypes.data_as(c_double_p), int(per_rate))
        out["sumtable"].append(st)
        row = []
        for t in brlens:
            d1, d2 = C.c_double(0), C.c_double(0)
            d.orc_likelihood_derivatives(s, s, r, ef, rw.ctypes.data_as(c_double_p),
                                         inv.ctypes.data_as(C.POINTER(C.c_int)) if inv is not None else None,
                                         pw.ctypes.data_as(c_uint_p), float(t), pinv.ctypes.data_as(c_double_p),
                                         _ptrs(frk), rt.ctypes.data_as(c_double_p), _ptrs(evals),
                                         st.ctypes.data_as(c_double_p), C.byref(d1), C.byref(d2))
            if case.asc_type in (1, 2):
                d.orc_asc_bias_derivatives(s, s, r, n, a.scaler, b.scaler, rw.ctypes.data_as(c_double_p),
                                           aw.ctypes.data_as(c_uint_p), pw_sum, float(t), pinv.ctypes.data_as(c_double_p),
                                           rt.ctypes.data_as(c_double_p), _ptrs(evals), st.ctypes.data_as(c_double_p),
                                           case.asc_type, int(per_rate), C.byref(d1), C.byref(d2))
            row.append((d1.value, d2.value))
        out["d"].append(row)
    return out
