"""CPU, world_size 2 over gloo: the N>1 logic of bench.py - contiguous site shards, independent
partitions, one all-reduce of the log-likelihood. The per-shard likelihood is computed by the
oracle here (no GPU in this container); what is under test is the sharding and the collective."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as O
from pllamd import api, sharding, workload as W


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kw, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = W.make_case("full", **kw)
    sub = sharding.shard_case(case, rank, world)
    res = O.run_case(sub)
    total = sharding.allreduce_sum(res["lnl"][0], dist)
    out[rank] = (total, sub.sites, res["persite"][0].sum())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kw", [dict(states=4, tips=8, sites=1000, seed=3),
                                dict(states=4, tips=8, sites=777, attributes=api.PATTERN_TIP | api.RATE_SCALERS, seed=4),
                                dict(states=20, tips=8, sites=130, pinv=0.2, mutate_pct=5, seed=5)],
                         ids=["dna", "dna-tip-rs-ragged", "aa-pinv"])
def test_two_rank_sharded_lnl(kw):
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), kw, out), nprocs=world, join=True)
    full = O.run_case(W.make_case("full", **kw))["lnl"][0]
    totals = [out[r][0] for r in range(world)]
    assert totals[0] == totals[1], "ranks disagree after the all-reduce"
    assert abs(totals[0] - full) <= 1e-12 * abs(full)
    assert sum(out[r][1] for r in range(world)) == kw["sites"]


@pytest.mark.parametrize("sites,world", [(100000, 8), (1000000, 8), (777, 2), (100, 8), (7, 8), (64, 1)])
def test_shard_bounds_cover_and_align(sites, world):
    b = sharding.shard_bounds(sites, world)
    assert b[0][0] == 0 and b[-1][1] == sites and len(b) == world
    assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
    assert all(lo <= hi for lo, hi in b)
    if sites >= world * 64:
        assert all(lo % 64 == 0 for lo, _ in b)
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 128


def test_sharded_case_slices_everything():
    case = W.make_case("w", 4, 8, 300, pattern_weights=np.arange(1, 301, dtype=np.uint32), seed=9)
    a, b = sharding.shard_case(case, 0, 2), sharding.shard_case(case, 1, 2)
    assert a.sites + b.sites == 300
    assert (np.concatenate([a.pattern_weights, b.pattern_weights]) == case.pattern_weights).all()
    assert all(x + y == z for x, y, z in zip(a.sequences, b.sequences, case.sequences))
    ra, rb, rf = O.run_case(a), O.run_case(b), O.run_case(case)
    assert abs(ra["lnl"][0] + rb["lnl"][0] - rf["lnl"][0]) < 1e-9
    assert np.allclose(np.concatenate([ra["persite"][0], rb["persite"][0]]), rf["persite"][0], rtol=0, atol=0)


def _group_worker(rank, world, name, kw, out):
    """--reduce peer on the CPU: the shard's value from the oracle, the sum through the library's own
    fixed-order exchange (csrc/host/group.c) - no torch.distributed collective on the data path"""
    import ctypes as C
    os.environ["PLL_AMD_HOST_ONLY"] = "1"
    lib = api.PllLib()
    g = lib.pll_gpu_group_join(name.encode(), rank, world, 20000)
    assert g, lib.errmsg()
    case = W.make_case("full", **kw)
    sub = sharding.shard_case(case, rank, world, sharding.balanced_bounds(case, world, align=8))
    res = O.run_case(sub)
    v, tot = np.array([res["lnl"][0]]), np.zeros(1)
    assert lib.pll_gpu_group_sum(g, api.dptr(v), 1, api.dptr(tot)), lib.errmsg()
    lib.pll_gpu_group_leave(g)
    out[rank] = (float(tot[0]), float(v[0]), sub.sites)


@pytest.mark.parametrize("world", [2, 3])
def test_group_exchange_sums_sharded_lnl_in_rank_order(world):
    import uuid
    import torch.multiprocessing as mp
    kw = dict(states=4, tips=16, sites=2000, attributes=api.SITE_REPEATS, mutate_pct=5, seed=11)
    out = mp.Manager().dict()
    mp.spawn(_group_worker, args=(world, "/pllamd-test-" + uuid.uuid4().hex[:12], kw, out), nprocs=world, join=True)
    full = O.run_case(W.make_case("full", **kw))["lnl"][0]
    want = out[0][1]
    for r in range(1, world):
        want = want + out[r][1]  # rank order
    assert all(out[r][0] == want for r in range(world)), "every rank holds the same bits: the rank-order sum"
    assert abs(want - full) <= 1e-12 * abs(full)
    assert sum(out[r][2] for r in range(world)) == kw["sites"]


def test_balanced_bounds_equalise_the_cost():
    """cuts of equal COST (sites + class entries of the tip-only subtrees) on a sorted alignment whose class
    counts vary along its length; they cover the alignment, stay tile-aligned and agree on every rank"""
    case = W.make_case("b", 4, 32, 40000, attributes=api.SITE_REPEATS, mutate_pct=8, seed=21)
    seqs = np.stack([np.frombuffer(s, dtype=np.uint8) for s in case.sequences])
    order = np.lexsort(seqs[::-1])
    case.sequences = [seqs[t][order].tobytes() for t in range(case.tips)]
    world = 8
    b = sharding.balanced_bounds(case, world)
    assert b == sharding.balanced_bounds(case, world)
    assert b[0][0] == 0 and b[-1][1] == case.sites and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
    assert all(lo % 64 == 0 and hi > lo for lo, hi in b)
    keys = sharding._tip_subtree_keys(case)

    def cost(lo, hi):
        return (hi - lo) + sharding.ENTRY_COST_PER_SITE_COST * sum(len(np.unique(k[lo:hi])) for k in keys)

    bal = [cost(lo, hi) for lo, hi in b]
    eq = [cost(lo, hi) for lo, hi in sharding.shard_bounds(case.sites, world)]
    assert max(bal) / min(bal) < 1.04 < max(eq) / min(eq)  # cuts move in steps of 64 sites = 1.3 % of a shard here
    assert max(bal) < max(eq)
    # no sequences / one rank: the plain cuts
    assert sharding.balanced_bounds(case, 1) == [(0, case.sites)]


def test_bench_refuses_a_mismatching_world_and_starts_its_own_ranks():
    """`bench.py --gpus N` must run N ranks or fail - never silently the one-GPU configuration.
    Without a launcher it starts N children of itself (which fail here: no GPU - loudly, no JSON line)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PLL_AMD_HOST_ONLY"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--sites", "3000", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "rank 0 exited" in r.stderr or "rank 1 exited" in r.stderr
