"""shared pieces of the derivative parity tests"""
import numpy as np

from pllamd import fixtures

DTOL = 1e-10  # north_star tolerance, applied to d_f / dd_f and (scaled) sumtable entries


def load(path):
    case, _, extra = fixtures.load(path)
    a = extra["arrays"]
    eig = dict(eigenvecs=a["eigenvecs"], inv_eigenvecs=a["inv_eigenvecs"], eigenvals=a["eigenvals"])
    edges = [(tuple(e), after) for e, after in extra["deriv_edges"]]
    return case, eig, a["rates"], edges, list(extra["brlens"]), a["d"], a["sumtable"], extra


def close(a, b, tol=DTOL, sites=1):
    """relative tolerance `tol`, plus an absolute floor: d_f is a sum over sites of -L'/L with L' a
    sum of lambda_j-weighted terms of size O(1) - on long branches everything but the (numerically
    not exactly) zero eigenvalue has died out and each site carries ~1e-15 of rounding noise"""
    return abs(a - b) <= tol * abs(b) + 4e-15 * sites


def assert_sumtable(got, exp, what=""):
    scale = np.abs(exp).max(axis=(1, 2), keepdims=True)
    err = np.abs(got - exp) / np.maximum(scale, 1e-300)
    assert err.max() <= DTOL, f"{what}: sumtable err {err.max():.3e} (relative to the site's largest entry)"


def run_session(s, case, eig, rates, edges, brlens, inject=True, exch=None):
    """replay: [partials up to batch] -> sumtable -> derivatives, per edge"""
    import numpy as np
    if inject:
        s.inject_eigen(eig, rates)
    else:
        s.set_model(exch, case.freqs, rates)
    done = 0
    out_d, out_st = [], []
    for (edge, after) in edges:
        while done <= after:
            arr, batch = s._op_arrays[done], case.op_batches[done]
            s.lib.pll_update_partials(s.p, arr, len(batch))
            done += 1
        st = s.new_sumtable()
        s.update_sumtable(edge, st)
        out_d.append([s.derivatives(edge, st, t) for t in brlens])
        out_st.append(s.read_sumtable(st))
    return np.array(out_d), np.stack(out_st)
