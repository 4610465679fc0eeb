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
