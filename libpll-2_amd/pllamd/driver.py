"""Issue the libpll call sequence of one test case against a library with the libpll ABI.

`run_case` is the Python rendering of what the reference's own tests do in C
(test/src/00010_NMDU_lkcalc.c:33-175): create partition, set model arrays and tip data, update
partials, evaluate edge log-likelihoods, read CLVs back. Because it talks ABI only, parity
tests call it once per library under comparison.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import api


@dataclass
class Case:
    name: str
    states: int
    rate_cats: int
    tips: int
    sites: int
    pmatrix: np.ndarray  # [prob_matrices][rate][states][states] float64, row = parent state
    freqs: np.ndarray  # [rate_matrices][states]
    op_batches: List[List[Tuple[int, ...]]]  # one inner list per pll_update_partials call
    edges: List[Tuple[int, int, int, int, int]]  # (parent, pscaler, child, cscaler, matrix)
    charmap: Optional[np.ndarray] = None  # uint64[256] char -> state mask
    sequences: Optional[List[bytes]] = None
    tip_clvs: Optional[np.ndarray] = None  # [tips][sites][states]
    attributes: int = 0  # PATTERN_TIP / RATE_SCALERS / SITE_REPEATS / AB_* (no arch bits)
    clv_buffers: int = 0
    scale_buffers: int = 0
    rate_weights: Optional[np.ndarray] = None
    pattern_weights: Optional[np.ndarray] = None
    prop_invar: Optional[np.ndarray] = None
    freqs_indices: Optional[np.ndarray] = None
    roots: List[Tuple[int, int]] = field(default_factory=list)  # (clv, scaler) root lnL
    dump_clvs: Optional[Sequence[int]] = None  # default: every op parent
    update_repeats: int = 1
    # optional model description for the derivative tests: exch (upper triangle), rates (category
    # rates), and the list of branch lengths at which derivatives are evaluated
    model: Optional[dict] = None
    # ascertainment-bias correction: attributes must carry api.AB_FLAG (extra per-state entries);
    # asc_type 0 none, 1 Lewis, 2 Felsenstein, 3 Stamatakis (PLL_ATTRIB_AB_* >> 5), set through
    # pll_set_asc_bias_type; asc_weights[states] through pll_set_asc_state_weights
    asc_type: int = 0
    asc_weights: Optional[np.ndarray] = None

    def __post_init__(self):
        self.pmatrix = np.ascontiguousarray(self.pmatrix, dtype=np.float64)
        self.freqs = np.ascontiguousarray(np.atleast_2d(self.freqs), dtype=np.float64)
        if not self.clv_buffers:
            self.clv_buffers = max(self.tips - 2, 1)
        if self.rate_weights is None:
            self.rate_weights = np.full(self.rate_cats, 1.0 / self.rate_cats)
        if self.pattern_weights is None:
            self.pattern_weights = np.ones(self.sites, dtype=np.uint32)
        if self.prop_invar is None:
            self.prop_invar = np.zeros(self.freqs.shape[0])
        if self.freqs_indices is None:
            self.freqs_indices = np.zeros(self.rate_cats, dtype=np.uint32)

    @property
    def asc_alloc(self):
        return bool(self.attributes & (api.AB_FLAG | (7 << 5)))

    @property
    def entries_alloc(self):
        """site entries per (uncompressed) CLV: the sites plus one extra entry per state"""
        return self.sites + (self.states if self.asc_alloc else 0)

    @property
    def prob_matrices(self):
        return self.pmatrix.shape[0]

    @property
    def rate_matrices(self):
        return self.freqs.shape[0]


def states_padded_for(states, arch):
    if arch & (api.ARCH_AVX | api.ARCH_AVX2):
        return (states + 3) & ~3
    if arch & api.ARCH_SSE:
        return (states + 1) & ~1
    return states


class Session:
    """A live partition built from a Case (kept open so benches can re-run the hot path)."""

    def __init__(self, lib: api.PllLib, case: Case, arch: int = api.ARCH_AVX2):
        self.lib, self.case, self.arch = lib, case, arch
        c = case
        p = lib.pll_partition_create(c.tips, c.clv_buffers, c.states, c.sites, c.rate_matrices,
                                     c.prob_matrices, c.rate_cats, c.scale_buffers,
                                     c.attributes | arch)
        if not p:
            raise RuntimeError(f"pll_partition_create failed: [{lib.errno()}] {lib.errmsg()}")
        self.p = p
        self.part = p.contents
        self.sp = self.part.states_padded
        try:
            self._load_inputs()
        except Exception:
            lib.pll_partition_destroy(p)
            self.p = None
            raise
        self._op_arrays = [api.make_ops(b) for b in c.op_batches]
        self._fi = np.ascontiguousarray(c.freqs_indices, dtype=np.uint32)

    # ------------------------------------------------------------------------------
    def _load_inputs(self):
        lib, c, part, sp = self.lib, self.case, self.part, self.sp
        for m in range(c.rate_matrices):
            f = np.ascontiguousarray(c.freqs[m], dtype=np.float64)
            lib.pll_set_frequencies(self.p, m, api.dptr(f))
        rw = np.ascontiguousarray(c.rate_weights, dtype=np.float64)
        lib.pll_set_category_weights(self.p, api.dptr(rw))
        pw = np.ascontiguousarray(c.pattern_weights, dtype=np.uint32)
        lib.pll_set_pattern_weights(self.p, api.uptr(pw))
        # transition matrices are written straight into the partition's block, padded stride
        r, s = c.rate_cats, c.states
        for i in range(c.prob_matrices):
            dst = api.as_np(part.pmatrix[i], r * s * sp, np.float64).reshape(r, s, sp)
            dst[:, :, :s] = c.pmatrix[i]
            dst[:, :, s:] = 0.0
        if lib.is_amd:
            lib.pll_gpu_invalidate(self.p, api.DIRTY_PMATRIX, -1)
        # tip data
        if c.sequences is not None:
            cmap = (C.c_ulonglong * 256)(*[int(x) for x in c.charmap])
            for t, seq in enumerate(c.sequences):
                assert len(seq) == c.sites
                if not lib.pll_set_tip_states(self.p, t, cmap, seq):
                    raise RuntimeError(f"pll_set_tip_states: [{lib.errno()}] {lib.errmsg()}")
        else:
            for t in range(c.tips):
                a = np.ascontiguousarray(c.tip_clvs[t], dtype=np.float64)
                if not lib.pll_set_tip_clv(self.p, t, api.dptr(a), 0):
                    raise RuntimeError(f"pll_set_tip_clv: [{lib.errno()}] {lib.errmsg()}")
        if c.asc_weights is not None:
            w = np.ascontiguousarray(c.asc_weights, dtype=np.uint32)
            lib.pll_set_asc_state_weights(self.p, api.uptr(w))
        if c.asc_type:
            if not lib.pll_set_asc_bias_type(self.p, c.asc_type << 5):
                raise RuntimeError(f"pll_set_asc_bias_type: [{lib.errno()}] {lib.errmsg()}")
        for m in range(c.rate_matrices):
            if c.prop_invar[m] > 0:
                if not lib.pll_update_invariant_sites_proportion(self.p, m, float(c.prop_invar[m])):
                    raise RuntimeError(f"invariant sites: [{lib.errno()}] {lib.errmsg()}")

    # ------------------------------------------------------------------------------
    def update_partials(self, update_repeats=None):
        ur = self.case.update_repeats if update_repeats is None else update_repeats
        for arr, batch in zip(self._op_arrays, self.case.op_batches):
            if ur == 1:
                self.lib.pll_update_partials(self.p, arr, len(batch))
            else:
                self.lib.pll_update_partials_rep(self.p, arr, len(batch), ur)

    def edge_lnl(self, edge, persite=True):
        c = self.case
        ps = np.zeros(c.sites) if persite else None
        v = self.lib.pll_compute_edge_loglikelihood(
            self.p, edge[0], edge[1], edge[2], edge[3], edge[4], api.uptr(self._fi),
            api.dptr(ps) if persite else None)
        return v, ps

    def root_lnl(self, root, persite=True):
        c = self.case
        ps = np.zeros(c.sites) if persite else None
        v = self.lib.pll_compute_root_loglikelihood(
            self.p, root[0], root[1], api.uptr(self._fi), api.dptr(ps) if persite else None)
        return v, ps

    def entries(self, clv_index):
        return self.lib.pll_get_sites_number(self.p, clv_index)

    def read_clv(self, clv_index, expand=True):
        """CLV as [sites or entries][rate][states] (padding stripped). With site repeats and
        expand=True the class-compressed CLV is expanded through site_id."""
        c, sp = self.case, self.sp
        if self.lib.is_amd:
            if not self.lib.pll_gpu_sync_clv(self.p, clv_index):
                raise RuntimeError(f"pll_gpu_sync_clv: [{self.lib.errno()}] {self.lib.errmsg()}")
        n = self.entries(clv_index)
        raw = api.as_np(self.part.clv[clv_index], n * c.rate_cats * sp, np.float64)
        a = raw.reshape(n, c.rate_cats, sp)[:, :, :c.states].copy()
        sid = self.lib.pll_get_site_id(self.p, clv_index)
        if expand and sid:
            ids = api.as_np(sid, c.sites, np.uint32)
            a = a[ids]
        return a

    def set_asc_type(self, asc_type):
        if not self.lib.pll_set_asc_bias_type(self.p, asc_type << 5):
            raise RuntimeError(f"pll_set_asc_bias_type: [{self.lib.errno()}] {self.lib.errmsg()}")

    def read_scaler(self, scaler_index, clv_index=None, expand=True):
        c = self.case
        if scaler_index < 0:
            return None
        if self.lib.is_amd:
            self.lib.pll_gpu_sync_scaler(self.p, scaler_index)
        n = self.entries(clv_index) if clv_index is not None else c.entries_alloc
        per = c.rate_cats if (c.attributes & api.RATE_SCALERS) else 1
        a = api.as_np(self.part.scale_buffer[scaler_index], n * per, np.uint32).reshape(n, per).copy()
        if clv_index is not None and expand:
            sid = self.lib.pll_get_site_id(self.p, clv_index)
            if sid:
                a = a[api.as_np(sid, c.sites, np.uint32)]
        return a

    # ---- branch-length derivatives ---------------------------------
    def set_model(self, exch, freqs, rates):
        """substitution parameters, frequencies and category rates through the model setters; the
        library computes its own eigensystem on demand (pll_update_eigen)"""
        c = self.case
        e = np.ascontiguousarray(exch, dtype=np.float64)
        for m in range(c.rate_matrices):
            f = np.ascontiguousarray(np.atleast_2d(freqs)[m], dtype=np.float64)
            self.lib.pll_set_frequencies(self.p, m, api.dptr(f))
            self.lib.pll_set_subst_params(self.p, m, api.dptr(e))
        r = np.ascontiguousarray(rates, dtype=np.float64)
        self.lib.pll_set_category_rates(self.p, api.dptr(r))

    def update_eigen(self):
        for m in range(self.case.rate_matrices):
            if not self.lib.pll_update_eigen(self.p, m):
                raise RuntimeError(f"pll_update_eigen: [{self.lib.errno()}] {self.lib.errmsg()}")

    def read_eigen(self):
        c, sp, part = self.case, self.sp, self.part
        out = {"eigenvecs": [], "inv_eigenvecs": [], "eigenvals": []}
        for m in range(c.rate_matrices):
            out["eigenvecs"].append(api.as_np(part.eigenvecs[m], c.states * sp, np.float64).reshape(c.states, sp)[:, :c.states].copy())
            out["inv_eigenvecs"].append(api.as_np(part.inv_eigenvecs[m], c.states * sp, np.float64).reshape(c.states, sp)[:, :c.states].copy())
            out["eigenvals"].append(api.as_np(part.eigenvals[m], c.states, np.float64).copy())
        return {k: np.stack(v) for k, v in out.items()}

    def inject_eigen(self, eig, rates):
        """write a given eigensystem into the partition's arrays (callers may do that: the arrays are
        public) and mark it valid, so both libraries work in the same eigenbasis"""
        c, sp, part = self.case, self.sp, self.part
        for m in range(c.rate_matrices):
            for name in ("eigenvecs", "inv_eigenvecs"):
                dst = api.as_np(getattr(part, name)[m], c.states * sp, np.float64).reshape(c.states, sp)
                dst[:, :] = 0.0
                dst[:, :c.states] = eig[name][m]
            api.as_np(part.eigenvals[m], sp, np.float64)[:c.states] = eig["eigenvals"][m]
            part.eigen_decomp_valid[m] = 1
        r = np.ascontiguousarray(rates, dtype=np.float64)
        self.lib.pll_set_category_rates(self.p, api.dptr(r))
        if self.lib.is_amd:
            self.lib.pll_gpu_invalidate(self.p, api.DIRTY_EIGEN, -1)

    def new_sumtable(self):
        """caller-owned table, aligned like pll_aligned_alloc(.., partition->alignment) in the
        reference's tests (its AVX kernels use aligned stores)"""
        n = self.case.entries_alloc * self.case.rate_cats * self.sp
        raw = np.zeros(n + 8, dtype=np.float64)
        off = (-raw.ctypes.data // 8) % 8  # doubles up to the next 64-byte boundary
        return raw[off:off + n]

    def update_sumtable(self, edge, sumtable):
        ok = self.lib.pll_update_sumtable(self.p, edge[0], edge[2], edge[1], edge[3], api.uptr(self._fi), api.dptr(sumtable))
        if not ok:
            raise RuntimeError(f"pll_update_sumtable: [{self.lib.errno()}] {self.lib.errmsg()}")

    def read_sumtable(self, sumtable):
        """[sites (+ states with ascertainment bias)][rate][states]; the AMD library keeps the table
        in HBM until asked for it"""
        c = self.case
        if self.lib.is_amd:
            if not self.lib.pll_gpu_sync_sumtable(self.p, api.dptr(sumtable)):
                raise RuntimeError(f"pll_gpu_sync_sumtable: [{self.lib.errno()}] {self.lib.errmsg()}")
        return sumtable.reshape(c.entries_alloc, c.rate_cats, self.sp)[:, :, :c.states].copy()

    def derivatives(self, edge, sumtable, t):
        d1, d2 = C.c_double(0), C.c_double(0)
        ok = self.lib.pll_compute_likelihood_derivatives(self.p, edge[1], edge[3], float(t), api.uptr(self._fi),
                                                         api.dptr(sumtable), C.byref(d1), C.byref(d2))
        if not ok:
            raise RuntimeError(f"pll_compute_likelihood_derivatives: [{self.lib.errno()}] {self.lib.errmsg()}")
        return d1.value, d2.value

    def close(self):
        if self.p:
            self.lib.pll_partition_destroy(self.p)
            self.p = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def run_case(lib: api.PllLib, case: Case, arch: int = api.ARCH_AVX2):
    """Full sequence; returns {'clv': {idx: arr}, 'scaler': {idx: arr}, 'lnl': [...],
    'persite': [...], 'root_lnl': [...], 'root_persite': [...]}. CLVs are scaler-free raw values;
    use `normalised` to compare across implementations whose scaling decisions may differ."""
    out = {"clv": {}, "scaler": {}, "lnl": [], "persite": [], "root_lnl": [], "root_persite": []}
    with Session(lib, case, arch) as s:
        s.update_partials()
        parents = {}
        for batch in case.op_batches:
            for op in batch:
                parents[op[0]] = op[1]
        dump = case.dump_clvs if case.dump_clvs is not None else sorted(parents)
        for idx in dump:
            out["clv"][idx] = s.read_clv(idx)
            sc = parents.get(idx, -1)
            if sc >= 0:
                out["scaler"][idx] = s.read_scaler(sc, idx)
        for e in case.edges:
            v, ps = s.edge_lnl(e)
            out["lnl"].append(v)
            out["persite"].append(ps)
        for r in case.roots:
            v, ps = s.root_lnl(r)
            out["root_lnl"].append(v)
            out["root_persite"].append(ps)
    return out


def normalised(clv, scaler):
    """(mantissa-like value, exponent) pairs that are invariant to WHEN a site was rescaled:
    returns clv * 2^(-256*scaler) as (frexp mantissa, integer exponent) arrays."""
    m, e = np.frexp(clv)
    e = e.astype(np.int64)
    if scaler is not None:
        sc = scaler.astype(np.int64)
        if sc.shape[1] == 1:
            e = e - 256 * sc[:, :, None]
        else:
            e = e - 256 * sc[:, :, None]
    e = np.where(clv == 0, 0, e)
    return m, e


def rel_err_normalised(a, sa, b, sb):
    """max relative difference between two (clv, scaler) pairs after undoing the scaling."""
    ma, ea = normalised(a, sa)
    mb, eb = normalised(b, sb)
    # bring to common exponent: values equal iff ma*2^ea == mb*2^eb
    d = np.clip(ea - eb, -64, 64)
    va = np.ldexp(ma, d)  # a expressed at b's exponent
    denom = np.maximum(np.abs(mb), 1e-300)
    err = np.abs(va - mb) / denom
    err = np.where((a == 0) & (b == 0), 0.0, err)
    return float(err.max()) if err.size else 0.0
