"""GPU: randomised life of ONE partition against the oracle - the caches of the device layer (launch plans, cherry
tables per pair of tip matrices, packed sub-tree look-ups, class maps) must follow whatever a caller does between two
traversals: new transition matrices for some branches, new tip sequences, partial traversals, class maps re-used or
recomputed. Every log-likelihood is compared with the oracle's for the partition's state at that moment."""
import ctypes as C
import os

import numpy as np
import pytest

from compare import RTOL
from oracle import oracle as O
from pllamd import api, driver, workload as W

pytestmark = pytest.mark.gpu


def _oracle_lnl(case):
    return O.run_case(case)["lnl"][0]


def _write_matrices(lib, s, case):
    r, st = case.rate_cats, case.states
    for i in range(case.prob_matrices):
        dst = api.as_np(s.part.pmatrix[i], r * st * s.sp, np.float64).reshape(r, st, s.sp)
        dst[:, :, :st] = case.pmatrix[i]
    lib.pll_gpu_invalidate(s.p, api.DIRTY_PMATRIX, -1)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PLL_FUZZ_SEEDS", "16"))))
def test_one_partition_through_random_changes(amd_lib, seed):
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    states = int(rng.choice([4, 4, 4, 20, 20, 20, 7, 61]))
    attrs = int(rng.choice([0, api.PATTERN_TIP, api.SITE_REPEATS, api.SITE_REPEATS, api.RATE_SCALERS, api.SITE_REPEATS | api.RATE_SCALERS]))
    tree = str(rng.choice(["balanced", "random", "caterpillar"]))
    tips = int(rng.choice([8, 16, 32, 64])) if tree == "balanced" else int(rng.integers(6, 40))
    kw = dict(states=states, tips=tips if states < 40 else min(tips, 16), sites=int(rng.integers(65, 2500 if states < 40 else 500)), tree=tree,
              mutate_pct=int(rng.choice([2, 10, 35])), seed=8000 + seed, attributes=attrs)
    if states == 20 and rng.random() < 0.3:
        kw["rate_cats"] = int(rng.choice([1, 2, 3]))
    case = W.make_case("fz", **kw)
    other = W.make_case("fz", **dict(kw, seed=9000 + seed))  # another alignment and other branch lengths, same tree
    repeats = bool(attrs & api.SITE_REPEATS)
    e = case.edges[0]
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        cmap = (C.c_ulonglong * 256)(*[int(x) for x in case.charmap])
        s.update_partials()
        v0 = s.edge_lnl(e, persite=False)[0]
        ref0 = _oracle_lnl(case)
        assert abs(v0 - ref0) <= RTOL * abs(ref0)
        orig = case.pmatrix.copy()
        for step in range(6):
            what = int(rng.integers(0, 4))
            if what == 0:      # new matrices for a random subset of branches
                pick = rng.random(case.prob_matrices) < 0.4
                case.pmatrix[pick] = other.pmatrix[pick]
                _write_matrices(amd_lib, s, case)
                s.update_partials(update_repeats=0 if repeats else None)
            elif what == 1:    # all matrices back
                case.pmatrix[:] = orig
                _write_matrices(amd_lib, s, case)
                s.update_partials(update_repeats=0 if repeats else None)
            elif what == 2:    # new sequences for some tips (class maps recomputed)
                seqs = list(case.sequences)
                for t in rng.choice(case.tips, size=max(1, case.tips // 3), replace=False):
                    seqs[int(t)] = other.sequences[int(t)]
                    assert amd_lib.pll_set_tip_states(s.p, int(t), cmap, seqs[int(t)])
                case.sequences = seqs
                s.update_partials(update_repeats=1 if repeats else None)
            else:              # the same traversal again (cached plans), then only its last third
                s.update_partials(update_repeats=0 if repeats else None)
                batch = case.op_batches[0]
                tail = batch[-max(1, len(batch) // 3):]
                arr = api.make_ops(tail)
                if repeats:
                    amd_lib.pll_update_partials_rep(s.p, arr, len(tail), 0)
                else:
                    amd_lib.pll_update_partials(s.p, arr, len(tail))
            v = s.edge_lnl(e, persite=False)[0]
            ref = _oracle_lnl(case)
            assert abs(v - ref) <= RTOL * abs(ref), (seed, step, what, v, ref)
