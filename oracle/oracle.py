"""ctypes front end of oracle/liboracle.so (the CPU restatement, pll_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg - never by the product (libpll-2_amd/). `run_case` evaluates a driver.Case with dense
(uncompressed) CLVs and returns the same result dictionary as pllamd.driver.run_case, so the
two can be compared key by key.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_LIB = os.path.join(HERE, "_ref", "libpll_ref.so")

c_uint_p = C.POINTER(C.c_uint)
c_double_p = C.POINTER(C.c_double)


class Child(C.Structure):
    _fields_ = [("clv", c_double_p), ("tipchars", C.POINTER(C.c_ubyte)),
                ("tipmap", C.POINTER(C.c_ulonglong)), ("scaler", c_uint_p), ("site_id", c_uint_p)]


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


_dll = None


def dll():
    global _dll
    if _dll is None:
        if not os.path.exists(LIB):
            build()
        d = C.CDLL(LIB)
        d.orc_update_partial.restype = None
        d.orc_update_partial.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, c_double_p, c_uint_p,
                                         c_uint_p, C.POINTER(Child), c_double_p, C.POINTER(Child),
                                         c_double_p, C.c_int]
        d.orc_edge_loglikelihood.restype = C.c_double
        d.orc_edge_loglikelihood.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.POINTER(Child),
                                             C.POINTER(Child), c_double_p, C.POINTER(c_double_p),
                                             c_double_p, c_uint_p, c_double_p, C.POINTER(C.c_int),
                                             c_uint_p, c_double_p, C.c_int]
        d.orc_root_loglikelihood.restype = C.c_double
        d.orc_root_loglikelihood.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.POINTER(Child),
                                             C.POINTER(c_double_p), c_double_p, c_uint_p, c_double_p,
                                             C.POINTER(C.c_int), c_uint_p, c_double_p, C.c_int]
        d.orc_asc_bias_correction.restype = C.c_double
        d.orc_asc_bias_correction.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.POINTER(Child),
                                              C.POINTER(Child), c_double_p, C.POINTER(c_double_p), c_double_p,
                                              c_uint_p, C.c_uint, c_uint_p, C.c_int, C.c_int]
        d.orc_repeat_classes.restype = C.c_uint
        d.orc_repeat_classes.argtypes = [C.c_uint, c_uint_p, C.c_uint, c_uint_p, C.c_uint, c_uint_p,
                                         c_uint_p]
        _dll = d
    return _dll


def _dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def _up(a):
    return a.ctypes.data_as(c_uint_p) if a is not None else None


def tip_masks(case):
    """[tips][sites] uint64 state masks of the tip data (src/pll.c:875-1024 semantics)."""
    if case.sequences is not None:
        cmap = np.asarray(case.charmap, dtype=np.uint64)
        out = np.stack([cmap[np.frombuffer(s, dtype=np.uint8)] for s in case.sequences])
        if (out == 0).any():
            raise ValueError("illegal state code in tip data")
        return out
    return None


def invariant_sites(case):
    """src/models.c:651-752: index of the single state shared by every tip, else -1."""
    m = tip_masks(case)
    if m is None:
        bits = (np.asarray(case.tip_clvs) != 0)
        w = (np.uint64(1) << np.arange(case.states, dtype=np.uint64))
        m = (bits * w).sum(-1).astype(np.uint64)
    inter = np.bitwise_and.reduce(m, axis=0)
    pop = np.array([bin(int(x)).count("1") for x in inter])
    ctz = np.array([(int(x) & -int(x)).bit_length() - 1 if x else -1 for x in inter])
    return np.where(pop == 1, ctz, -1).astype(np.int32)


def repeat_classes(site_id_left, ids_left, site_id_right, ids_right):
    n = len(site_id_left)
    sid = np.zeros(n, dtype=np.uint32)
    ids = np.zeros(n, dtype=np.uint32)
    l = np.ascontiguousarray(site_id_left, dtype=np.uint32)
    r = np.ascontiguousarray(site_id_right, dtype=np.uint32)
    k = dll().orc_repeat_classes(n, _up(l), ids_left, _up(r), ids_right, _up(sid), _up(ids))
    return sid, ids[:k].copy()


class _Node:
    """dense per-site representation of one node for the oracle"""

    def __init__(self):
        self.clv = None  # [sites][r][sp]
        self.codes = None  # uint8[sites] (pattern-tip)
        self.tipmap = None
        self.scaler = None

    def child(self):
        c = Child()
        if self.clv is not None:
            c.clv = _dp(self.clv)
        else:
            c.tipchars = self.codes.ctypes.data_as(C.POINTER(C.c_ubyte))
            if self.tipmap is not None:
                c.tipmap = self.tipmap.ctypes.data_as(C.POINTER(C.c_ulonglong))
        if self.scaler is not None:
            c.scaler = _up(self.scaler)
        return c


def dense_nodes(case):
    """(nodes, scalers) after all op batches: per node a dense [sites][r][s] CLV or tip codes"""
    return run_case(case)["_nodes"]


def run_case(case, states_padded=None, pattern_tip=None):
    """Evaluate `case` with the restatement. pattern_tip: None -> follow case.attributes."""
    from pllamd import api  # constants only

    d = dll()
    s, r, n = case.states, case.rate_cats, case.sites
    # ascertainment bias: one extra entry per state behind the sites (src/pll.c:525-531); every CLV
    # update covers them, the lnL loops do not
    na = n + (s if case.asc_alloc else 0)
    sp = states_padded or s
    per_rate = bool(case.attributes & api.RATE_SCALERS)
    if pattern_tip is None:
        pattern_tip = bool(case.attributes & api.PATTERN_TIP)
    span = r * sp

    masks = tip_masks(case)
    if masks is not None and na > n:
        onehot = np.uint64(1) << np.arange(s, dtype=np.uint64)
        masks = np.concatenate([masks, np.broadcast_to(onehot, (case.tips, s))], axis=1)
    nodes = {}
    for t in range(case.tips):
        nd = _Node()
        if pattern_tip:
            if s == 4:
                nd.codes = masks[t].astype(np.uint8)
            else:
                uniq, inv = np.unique(masks[t], return_inverse=True)
                nd.codes = inv.astype(np.uint8)
                nd.tipmap = np.zeros(256, dtype=np.uint64)
                nd.tipmap[:len(uniq)] = uniq
        else:
            clv = np.zeros((na, r, sp))
            if masks is not None:
                bits = ((masks[t][:, None] >> np.arange(s, dtype=np.uint64)[None, :]) & np.uint64(1))
                clv[:, :, :s] = bits[:, None, :].astype(np.float64)
            else:
                clv[:n, :, :s] = np.asarray(case.tip_clvs[t])[:, None, :]
                if na > n:
                    clv[n:, :, :s] = np.eye(s)[:, None, :]
            nd.clv = np.ascontiguousarray(clv)
        nodes[t] = nd

    pm = np.zeros((case.prob_matrices, r, s, sp))
    pm[:, :, :, :s] = case.pmatrix
    scalers = {}

    out = {"clv": {}, "scaler": {}, "lnl": [], "persite": [], "root_lnl": [], "root_persite": []}
    parents = {}
    for batch in case.op_batches:
        for (pc, psc, c1, m1, s1, c2, m2, s2) in batch:
            left, right = nodes[c1], nodes[c2]
            # scaler buffers are addressed by the op, not owned by the node (src/pll.h:325-335)
            lch, rch = left.child(), right.child()
            lch.scaler = _up(scalers[s1]) if s1 >= 0 else None
            rch.scaler = _up(scalers[s2]) if s2 >= 0 else None
            par = _Node()
            par.clv = np.zeros((na, r, sp))
            mode = 0
            pscal = None
            if psc >= 0:
                mode = 2 if per_rate else 1
                pscal = np.zeros(na * (r if per_rate else 1), dtype=np.uint32)
            d.orc_update_partial(s, sp, r, na, _dp(par.clv), _up(pscal), None, C.byref(lch),
                                 _dp(pm[m1]), C.byref(rch), _dp(pm[m2]), mode)
            nodes[pc] = par
            if psc >= 0:
                scalers[psc] = pscal
            parents[pc] = psc

    out["_nodes"] = (nodes, scalers)  # dense per-node state, for oracle_deriv
    dump = case.dump_clvs if case.dump_clvs is not None else sorted(parents)
    for idx in dump:
        out["clv"][idx] = nodes[idx].clv[:, :, :s].copy()
        psc = parents.get(idx, -1)
        if psc >= 0:
            out["scaler"][idx] = scalers[psc].reshape(na, -1).copy()

    fr = np.zeros((case.rate_matrices, sp))
    # pll_set_frequencies renormalises only when the sum is off by more than 1e-8
    # (src/models.c:456-464)
    fsum = case.freqs.sum(1, keepdims=True)
    fr[:, :s] = np.where(np.abs(fsum - 1.0) > 1e-8, case.freqs / fsum, case.freqs)
    fptrs = (c_double_p * case.rate_matrices)(*[_dp(fr[i]) for i in range(case.rate_matrices)])
    rw = np.ascontiguousarray(case.rate_weights, dtype=np.float64)
    pw = np.ascontiguousarray(case.pattern_weights, dtype=np.uint32)
    pinv = np.ascontiguousarray(case.prop_invar, dtype=np.float64)
    inv = invariant_sites(case) if (pinv > 0).any() else None
    fi = np.ascontiguousarray(case.freqs_indices, dtype=np.uint32)
    invp = inv.ctypes.data_as(C.POINTER(C.c_int)) if inv is not None else None
    aw = np.ascontiguousarray(case.asc_weights if case.asc_weights is not None else np.zeros(s), dtype=np.uint32)
    pw_sum = int(np.asarray(case.pattern_weights, dtype=np.uint64).sum())

    def asc(ach, bch, pmat):
        if not case.asc_type:
            return 0.0
        return d.orc_asc_bias_correction(s, sp, r, n, ach, bch, pmat, fptrs, _dp(rw), _up(aw), pw_sum, _up(fi),
                                         case.asc_type, int(per_rate))

    for (pc, psc, cc, csc, mi) in case.edges:
        a, b = nodes[pc], nodes[cc]
        if a.clv is None:  # the reference makes the inner node the "parent" (src/likelihood.c:612-624)
            a, b, psc, csc = b, a, csc, psc
        ach, bch = a.child(), b.child()
        ach.scaler = _up(scalers[psc]) if psc >= 0 else None
        bch.scaler = _up(scalers[csc]) if csc >= 0 else None
        ps = np.zeros(n)
        v = d.orc_edge_loglikelihood(s, sp, r, n, C.byref(ach), C.byref(bch), _dp(pm[mi]), fptrs,
                                     _dp(rw), _up(pw), _dp(pinv), invp, _up(fi), _dp(ps),
                                     int(per_rate))
        out["lnl"].append(v + asc(C.byref(ach), C.byref(bch), _dp(pm[mi])))
        out["persite"].append(ps)
    for (rc, rsc) in case.roots:
        ch = nodes[rc].child()
        ch.scaler = _up(scalers[rsc]) if rsc >= 0 else None
        ps = np.zeros(n)
        v = d.orc_root_loglikelihood(s, sp, r, n, C.byref(ch), fptrs, _dp(rw), _up(pw), _dp(pinv),
                                     invp, _up(fi), _dp(ps), int(per_rate))
        out["root_lnl"].append(v + asc(C.byref(ch), None, None))
        out["root_persite"].append(ps)
    return out
