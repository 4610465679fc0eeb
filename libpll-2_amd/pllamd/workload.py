"""Synthetic workloads for tests and bench.py: tree shapes -> op lists,
seeded alignments, reversible substitution models -> transition matrices. Numpy only; nothing
here is on the product path.
"""
import numpy as np

from . import api
from .driver import Case

NT_CHARS = b"ACGT"
AA_CHARS = b"ARNDCQEGHILKMFPSTWYV"


# ---- character maps (IUPAC; SURVEY.md section 8b "globals") ------------------------------------
def map_nt():
    m = np.zeros(256, dtype=np.uint64)
    codes = {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "R": 5, "Y": 10, "S": 6, "W": 9, "K": 12,
             "M": 3, "B": 14, "D": 13, "H": 11, "V": 7, "N": 15, "X": 15, "O": 15, "-": 15,
             ".": 15, "?": 15}
    for ch, v in codes.items():
        m[ord(ch)] = v
        m[ord(ch.lower())] = v
    return m


def map_aa():
    m = np.zeros(256, dtype=np.uint64)
    for i, ch in enumerate(AA_CHARS.decode()):
        m[ord(ch)] = m[ord(ch.lower())] = 1 << i
    bit = {ch: 1 << i for i, ch in enumerate(AA_CHARS.decode())}
    amb = {"B": bit["N"] | bit["D"], "Z": bit["Q"] | bit["E"], "J": bit["I"] | bit["L"]}
    allbits = (1 << 20) - 1
    for ch in "X*-.?":
        amb[ch] = allbits
    for ch, v in amb.items():
        m[ord(ch)] = v
        if ch.isalpha():
            m[ord(ch.lower())] = v
    return m


def map_generic(states, first=48):
    """one printable character per state (chars first, first+1, ...), plus '-' = fully ambiguous."""
    assert states <= 64 and first + states < 127
    m = np.zeros(256, dtype=np.uint64)
    for i in range(states):
        m[first + i] = np.uint64(1) << np.uint64(i)
    m[ord("-")] = np.uint64((1 << states) - 1) if states < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    # partial ambiguities: '!' = first two states, '#' = upper half, '$' = second and last state
    m[ord("!")] = np.uint64(3)
    m[ord("#")] = np.uint64(sum(1 << i for i in range(states // 2, states)))
    m[ord("$")] = np.uint64((1 << 1) | (1 << (states - 1)))
    return m


# ---- trees -> operation lists -----------------------------------------------------------------
def balanced_ops(tips, with_scalers=True):
    """pair adjacent nodes level by level until two remain. Returns (ops, edge,
    levels) with ops as 8-tuples in pll_operation_t order; edge = (A, scA, B, scB, matrix A)."""
    assert tips >= 4 and tips & (tips - 1) == 0
    cur = list(range(tips))
    nxt_id = tips
    ops, levels = [], []

    def sc(idx):
        return (idx - tips) if (with_scalers and idx >= tips) else -1

    while len(cur) > 2:
        new, lvl = [], []
        for i in range(0, len(cur), 2):
            a, b = cur[i], cur[i + 1]
            ops.append((nxt_id, sc(nxt_id), a, a, sc(a), b, b, sc(b)))
            lvl.append(len(ops) - 1)
            new.append(nxt_id)
            nxt_id += 1
        levels.append(lvl)
        cur = new
    a, b = cur
    return ops, (a, sc(a), b, sc(b), a), levels


def caterpillar_ops(tips, with_scalers=True):
    """ladder tree ((((t0,t1),t2),t3)...): tips-2 ops, every op depends on the previous one;
    the deep chain drives CLVs below 2^-256 so the scaling code is exercised."""
    ops = []

    def sc(idx):
        return (idx - tips) if (with_scalers and idx >= tips) else -1

    prev = 0
    nxt_id = tips
    for t in range(1, tips - 1):
        ops.append((nxt_id, sc(nxt_id), prev, prev, sc(prev), t, t, sc(t)))
        prev = nxt_id
        nxt_id += 1
    last = tips - 1
    return ops, (prev, sc(prev), last, -1, last), [[i] for i in range(len(ops))]


def random_tree_ops(tips, seed=1, scaler_pct=100):
    """random unrooted binary tree: join two random subtrees until two remain (the evaluated edge).
    Ops come out in an order in which producers precede consumers but levels are interleaved; a
    node keeps a scaler buffer with probability scaler_pct % (children without one pass -1 on)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pool = list(range(tips))
    has_sc = {}
    ops = []
    nxt = tips

    def sc(idx):
        return (idx - tips) if has_sc.get(idx, False) else -1

    while len(pool) > 2:
        i, j = sorted(rng.choice(len(pool), size=2, replace=False))
        b = pool.pop(j)
        a = pool.pop(i)
        has_sc[nxt] = bool(rng.integers(0, 100) < scaler_pct)
        ops.append((nxt, sc(nxt), a, a, sc(a), b, b, sc(b)))
        pool.append(nxt)
        nxt += 1
    a, b = pool
    if a < tips:  # the inner node plays the parent end
        a, b = b, a
    return ops, (a, sc(a), b, sc(b), min(a, b)), None  # the last node is never a child: its own matrix index does not exist


def branch_lengths(n, lo=0.05, step=0.01, period=10):
    return lo + step * (np.arange(n) % period)


# ---- alignments -------------------------------------------------------------------------------
def random_states(tips, sites, states, seed=1, mutate_pct=30):
    """SURVEY 8d in spirit: per site an ancestral state; every tip copies it unless a draw
    (mutate_pct %) replaces it by a uniform random state. Vectorised (PCG64), deterministic."""
    rng = np.random.Generator(np.random.PCG64(seed))
    anc = rng.integers(0, states, size=sites, dtype=np.int64)
    mut = rng.integers(0, 100, size=(tips, sites)) < mutate_pct
    rnd = rng.integers(0, states, size=(tips, sites), dtype=np.int64)
    return np.where(mut, rnd, anc[None, :]).astype(np.uint8)


SECTION_8D_SEED = 88172645463325252


def section8d_states(tips, sites, states, mutate_pct=30, seed=SECTION_8D_SEED):
    """SURVEY 8d to the letter: xorshift64 (13/7/17, output x >> 32) seeded with 88172645463325252;
    per site an ancestral state u mod states, then every tip in turn copies it unless a draw
    u mod 100 < 30 replaces it by the state of one more draw. The stream is sequential (170M draws
    for C4), so it comes from the small C helper built next to the library
    (csrc/workload/synth_alignment.c -> libpll_workload.so)."""
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc", "libpll_workload.so")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found - build it first (python -c 'import __graft_entry__ as g; g.build()')")
    dll = C.CDLL(path)
    fn = dll.pllwl_xorshift_alignment
    fn.restype = None
    fn.argtypes = [C.c_uint, C.c_size_t, C.c_uint, C.c_uint, C.POINTER(C.c_uint64), C.c_void_p, C.c_size_t]
    out = np.zeros((tips, sites), dtype=np.uint8)
    st = C.c_uint64(seed)
    fn(tips, sites, states, mutate_pct, C.byref(st), out.ctypes.data, sites)
    return out


def states_to_sequences(st, alphabet):
    lut = np.frombuffer(bytes(alphabet), dtype=np.uint8)
    return [lut[row].tobytes() for row in st]


def onehot_clvs(st, states):
    tips, sites = st.shape
    out = np.zeros((tips, sites, states))
    out[np.arange(tips)[:, None], np.arange(sites)[None, :], st] = 1.0
    return out


# ---- models -----------------------------------------------------------------------------------
def gamma_rates_mean(alpha, cats):
    """discrete-Gamma category rates, mean-of-category variant (Yang 1994): the quantity
    src/gamma.c:220-292 produces for PLL_GAMMA_RATES_MEAN."""
    from scipy.special import gammainc, gammaincinv
    if cats == 1:
        return np.ones(1)
    # boundaries of equal-probability categories of Gamma(alpha, beta=alpha)
    qs = gammaincinv(alpha, np.arange(1, cats) / cats) / alpha
    upper = np.concatenate([gammainc(alpha + 1, qs * alpha), [1.0]])
    lower = np.concatenate([[0.0], upper[:-1]])
    return (upper - lower) * cats


def reversible_q(exch, freqs):
    """exch: upper-triangle exchangeabilities (row-major, s(s-1)/2); returns the rate matrix Q
    normalised to one expected substitution per unit time (src/models.c create_ratematrix)."""
    s = len(freqs)
    r = np.zeros((s, s))
    r[np.triu_indices(s, 1)] = exch
    r = r + r.T
    q = r * freqs[None, :]
    np.fill_diagonal(q, 0)
    np.fill_diagonal(q, -q.sum(1))
    q /= -(np.diag(q) * freqs).sum()
    return q


def pmatrices(exch, freqs, rates, brlens, pinv=0.0):
    """[branch][rate][i][j] = expm(Q * rate * t / (1-pinv)) via the symmetrised eigensystem."""
    freqs = np.asarray(freqs, dtype=np.float64)
    q = reversible_q(np.asarray(exch, dtype=np.float64), freqs)
    sq = np.sqrt(freqs)
    sym = q * sq[:, None] / sq[None, :]
    sym = 0.5 * (sym + sym.T)
    w, v = np.linalg.eigh(sym)
    left = v / sq[:, None] * 1.0  # D^-1/2 V
    right = v.T * sq[None, :]  # V^T D^1/2
    out = np.empty((len(brlens), len(rates), len(freqs), len(freqs)))
    for b, t in enumerate(brlens):
        for k, r in enumerate(rates):
            e = np.exp(w * r * t / (1.0 - pinv))
            out[b, k] = (left * e[None, :]) @ right
    return np.clip(out, 0.0, None)


def eigensystem(exch, freqs):
    """eigen-decomposition of the reversible rate matrix in the array conventions of
    pll_partition_t (src/models.c:346-398): eigenvecs[m][k] = U[k][m] sqrt(pi_k),
    inv_eigenvecs[j][m] = U[j][m] / sqrt(pi_j), so that P(t) = inv_eigenvecs . diag(exp(lambda t)) .
    eigenvecs. Returns arrays with a leading rate-matrix axis of length 1."""
    freqs = np.asarray(freqs, dtype=np.float64)
    q = reversible_q(np.asarray(exch, dtype=np.float64), freqs)
    sq = np.sqrt(freqs)
    sym = q * sq[:, None] / sq[None, :]
    sym = 0.5 * (sym + sym.T)
    w, u = np.linalg.eigh(sym)
    return dict(eigenvecs=(u.T * sq[None, :])[None], inv_eigenvecs=(u / sq[:, None])[None], eigenvals=w[None])


GTR_DNA = dict(exch=[1, 2, 1, 1, 2, 1], freqs=[0.3, 0.2, 0.2, 0.3])


def synthetic_exch(states):
    """SURVEY 8d codon stand-in: rate[i] = 1 + (i mod 3) (last = 1), pi_i ~ 1 + 0.25 (i mod 4)."""
    n = states * (states - 1) // 2
    ex = 1.0 + (np.arange(n) % 3)
    ex[-1] = 1.0
    fr = 1.0 + 0.25 * (np.arange(states) % 4)
    return ex, fr / fr.sum()


def make_case(name, states, tips, sites, rate_cats=4, tree="balanced", attributes=0, seed=1,
              mutate_pct=30, alpha=0.5, scalers=True, tips_as="states", exch=None, freqs=None,
              brlen_scale=1.0, pinv=0.0, pattern_weights=None, ambiguity_pct=0, partial_pct=0,
              asc_type=None, asc_weights=None, generator="pcg64", states_matrix=None):
    """One synthetic configuration of SURVEY 8d (C2: states=4,tips=64,sites=100000; C3: 20/64/
    50000; C5: 61/32/20000)."""
    if exch is None:
        if states == 4:
            exch, freqs = GTR_DNA["exch"], GTR_DNA["freqs"]
        else:
            exch, freqs = synthetic_exch(states)
    freqs = np.asarray(freqs, dtype=np.float64)
    if tree == "balanced":
        ops, edge, _ = balanced_ops(tips, scalers)
    elif tree == "random":
        ops, edge, _ = random_tree_ops(tips, seed=seed + 500, scaler_pct=100 if scalers is True else (int(scalers) if scalers else 0))
    else:
        ops, edge, _ = caterpillar_ops(tips, scalers)
    nmat = 2 * tips - 3
    rates = gamma_rates_mean(alpha, rate_cats)
    pm = pmatrices(exch, freqs, rates, branch_lengths(nmat) * brlen_scale, pinv)
    if states_matrix is not None:          # the caller's own [tips][sites] states (e.g. a shard of a sorted alignment)
        st = np.ascontiguousarray(states_matrix, dtype=np.uint8)
        assert st.shape == (tips, sites)
    elif generator == "xorshift64":        # SURVEY 8d to the letter (bench.py)
        st = section8d_states(tips, sites, states, mutate_pct)
    else:
        st = random_states(tips, sites, states, seed, mutate_pct)
    kw = {}
    if tips_as == "states":
        if states == 4:
            cmap, alphabet = map_nt(), NT_CHARS
        elif states == 20:
            cmap, alphabet = map_aa(), AA_CHARS
        else:
            cmap = map_generic(states)
            alphabet = bytes(range(48, 48 + states))
        seqs = states_to_sequences(st, alphabet)
        if ambiguity_pct:
            rng = np.random.Generator(np.random.PCG64(seed + 77))
            amb = ord("-")
            seqs2 = []
            for sq in seqs:
                a = np.frombuffer(sq, dtype=np.uint8).copy()
                a[rng.integers(0, 100, size=sites) < ambiguity_pct] = amb
                seqs2.append(a.tobytes())
            seqs = seqs2
        if partial_pct:
            # partially ambiguous characters of the alphabet (IUPAC for DNA / protein)
            pool = np.frombuffer({4: b"RYSWKMBDHV", 20: b"BZJ"}.get(states, b"!#$"), dtype=np.uint8)
            rng = np.random.Generator(np.random.PCG64(seed + 78))
            seqs2 = []
            for sq in seqs:
                a = np.frombuffer(sq, dtype=np.uint8).copy()
                hit = rng.integers(0, 100, size=sites) < partial_pct
                a[hit] = pool[rng.integers(0, len(pool), size=int(hit.sum()))]
                seqs2.append(a.tobytes())
            seqs = seqs2
        kw.update(charmap=cmap, sequences=seqs)
    else:
        kw.update(tip_clvs=onehot_clvs(st, states))
    if asc_type is not None:
        # ascertainment-bias partition (extra per-state entries); asc_type 0 keeps the correction off
        from . import api
        attributes |= api.AB_FLAG
        kw.update(asc_type=asc_type, asc_weights=None if asc_weights is None else np.asarray(asc_weights, dtype=np.uint32))
    return Case(name=name, states=states, rate_cats=rate_cats, tips=tips, sites=sites, pmatrix=pm,
                freqs=freqs[None, :], op_batches=[ops], edges=[edge], attributes=attributes,
                clv_buffers=tips - 2, scale_buffers=(tips - 2) if scalers else 0,
                prop_invar=np.array([pinv]), pattern_weights=pattern_weights,
                model=dict(exch=np.asarray(exch, dtype=np.float64), rates=rates), **kw)
