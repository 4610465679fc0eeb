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
