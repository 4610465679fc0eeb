"""GPU: two PRODUCT partitions, one per process, on two streams of the one device, each over its shard of
one alignment, log-likelihood evaluated asynchronously on the device and summed by one all-reduce (gloo) -
the multi-GPU flow of bench.py end to end, against the un-sharded evaluation."""
import json
import os
import socket
import subprocess
import sys

import pytest

from oracle import oracle as O
from pllamd import api, driver, workload as W

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("kw", [dict(states=4, tips=32, sites=20000, seed=21),
                                dict(states=4, tips=32, sites=9999, seed=22, attributes=api.SITE_REPEATS, mutate_pct=6),
                                dict(states=20, tips=8, sites=3000, seed=23)], ids=["dna", "dna-repeats-ragged", "aa"])
def test_two_ranks_two_streams_one_allreduce(amd_lib, kw):
    world, port = 2, str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_product_worker.py"), str(r), str(world), port, json.dumps(kw)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    case = W.make_case("full", **kw)
    with driver.Session(amd_lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        whole, _ = s.edge_lnl(case.edges[0], persite=False)
    exp = O.run_case(case)["lnl"][0]
    assert abs(whole - exp) <= 1e-10 * abs(exp)
    assert sum(o["sites"] for o in outs) == kw["sites"]
    for o in outs:
        assert o["totals"][0] == outs[0]["totals"][0] and len(set(o["totals"])) == 1  # every rank, every step: the same sum
        assert abs(o["totals"][0] - whole) <= 1e-12 * abs(whole)
    assert abs(sum(o["own"] for o in outs) - outs[0]["totals"][0]) <= 1e-12 * abs(whole)


def _run_workers(world, kw):
    import uuid
    name = "/pllamd-test-" + uuid.uuid4().hex[:12]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "group_deriv_worker.py"), str(r), str(world), name, json.dumps(kw)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    return outs


@pytest.mark.parametrize("kw", [dict(states=4, tips=16, sites=30000, seed=31), dict(states=20, tips=8, sites=4000, seed=32)], ids=["dna", "aa"])
def test_sharded_newton_steps_stay_in_lockstep(kw):
    """pll_gpu_group_likelihood_derivatives (SURVEY rows e + f1): three ranks optimise one branch of one alignment, each
    over its shard; the derivatives are summed in rank order, so every rank sees the same bits at every iteration and
    ends at the same branch length - which is also the un-sharded optimisation's, to the path's tolerance"""
    whole = _run_workers(1, kw)[0]
    outs = _run_workers(3, kw)
    assert sum(o["sites"] for o in outs) == kw["sites"]
    for o in outs[1:]:
        assert o["trace"] == outs[0]["trace"] and o["t"] == outs[0]["t"]  # the same bits on every rank
    for (t, d1, d2), (tw, d1w, d2w) in zip(outs[0]["trace"], whole["trace"]):
        t, d1, d2, tw, d1w, d2w = (float.fromhex(x) for x in (t, d1, d2, tw, d1w, d2w))
        assert abs(t - tw) <= 1e-9 * max(abs(tw), 1e-3)
        assert abs(d1 - d1w) <= 1e-7 * max(abs(d1w), 1.0) and abs(d2 - d2w) <= 1e-7 * abs(d2w)  # (t itself moved by rounding)
    first, first_w = outs[0]["trace"][0], whole["trace"][0]
    for a, b in zip(first[1:], first_w[1:]):  # same branch length: the path's tolerance
        a, b = float.fromhex(a), float.fromhex(b)
        assert abs(a - b) <= 1e-10 * abs(b) + 4e-15 * kw["sites"]
    # the step in which the last rank's evaluation fails: PLL_FAILURE everywhere, the others are told why
    assert all(not o["failed"][0] for o in outs), [o["failed"] for o in outs]
    assert "another rank" in outs[0]["failed"][2] and "another rank" not in outs[2]["failed"][2]
