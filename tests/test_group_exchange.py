"""CPU: the fixed-order exchange of a site-sharded run (csrc/host/group.c, include/pll_amd.h pll_gpu_group_*;
SURVEY section 8 row e; the sum it stands for: src/core_likelihood.c:1489). Ranks are separate processes that
meet in a POSIX shared-memory segment; every rank must return the SAME bits, equal to the slots added in
rank order, whatever the arrival order."""
import json
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from group_worker import values  # noqa: E402


def run_ranks(size, steps=50, count=3, timeout_ms=20000, ranks=None, name=None):
    name = name or "/pllamd-test-" + uuid.uuid4().hex[:12]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "group_worker.py"), name, str(r), str(size), str(steps), str(count), str(timeout_ms)],
                              stdout=subprocess.PIPE, text=True) for r in (ranks if ranks is not None else range(size))]
    outs = [json.loads(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in procs]
    try:
        os.unlink("/dev/shm" + name)  # only a failed run leaves it behind
    except OSError:
        pass
    return outs


@pytest.mark.parametrize("size", [1, 2, 4, 8])
def test_group_sum_is_the_rank_order_sum_on_every_rank(size):
    steps, count = 40, 3
    outs = run_ranks(size, steps, count)
    assert all("error" not in o for o in outs), outs
    for step in range(1, steps + 1):
        acc = values(0, step, count).copy()
        for r in range(1, size):
            acc = acc + values(r, step, count)  # rank order
        want = [float.hex(float(x)) for x in acc]
        for o in outs:
            assert o["sums"][step - 1] == want, (size, step, o["rank"])
    print(f"group of {size}: exchange {max(o['exchange_us'] for o in outs):.2f} us per step (slowest rank)")


def test_group_sum_order_matters_for_these_values():
    """the check above would pass for any order if the values were tame: they are not"""
    differs = 0
    for step in range(1, 41):
        vs = [values(r, step, 3) for r in range(4)]
        fwd = ((vs[0] + vs[1]) + vs[2]) + vs[3]
        rev = ((vs[3] + vs[2]) + vs[1]) + vs[0]
        differs += int(np.any(fwd != rev))
    assert differs > 10


def test_a_missing_rank_times_out_loudly():
    outs = run_ranks(3, steps=1, timeout_ms=400, ranks=[0, 2])
    assert all("error" in o and "joined" in o["error"] for o in outs), outs
    assert all(o["errno"] == 901 for o in outs)


def test_a_name_in_use_with_another_size_is_refused():
    name = "/pllamd-test-" + uuid.uuid4().hex[:12]
    first = run_ranks(1, steps=1, name=name + "a")
    assert "error" not in first[0]
    # two ranks claim sizes 2 and 3 under one name: the second to arrive is refused, the first times out
    a = subprocess.Popen([sys.executable, os.path.join(HERE, "group_worker.py"), name, "0", "2", "1", "1", "1500"], stdout=subprocess.PIPE, text=True)
    import time
    time.sleep(0.7)
    b = subprocess.Popen([sys.executable, os.path.join(HERE, "group_worker.py"), name, "1", "3", "1", "1", "1500"], stdout=subprocess.PIPE, text=True)
    ob = json.loads(b.communicate(timeout=60)[0].strip().splitlines()[-1])
    oa = json.loads(a.communicate(timeout=60)[0].strip().splitlines()[-1])
    try:
        os.unlink("/dev/shm" + name)
    except OSError:
        pass
    assert "error" in ob and "exists with 2 ranks" in ob["error"], ob
    assert "error" in oa, oa


def test_a_finished_run_removes_its_name_and_a_stale_segment_is_refused():
    """the last rank to leave unlinks the segment; one that a crashed run of the SAME size left behind (its ranks
    counted as joined) is refused by the next run under that name instead of being silently reused"""
    name = "/pllamd-test-" + uuid.uuid4().hex[:12]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "group_worker.py"), name, str(r), "2", "3", "1", "20000"],
                              stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = [json.loads(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in procs]
    assert all("error" not in o for o in outs), outs
    assert not os.path.exists("/dev/shm" + name)
    # the header of a run of two ranks that never left: {magic, size = 2, joined = 2, left = 0} (csrc/host/group.c)
    stale = bytearray(64 + 2 * 2 * 64)
    stale[0:8] = (0x504c4c4752503031).to_bytes(8, "little")
    stale[8:12] = (2).to_bytes(4, "little")
    stale[12:16] = (2).to_bytes(4, "little")
    with open("/dev/shm" + name, "wb") as f:
        f.write(stale)
    try:
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "group_worker.py"), name, str(r), "2", "1", "1", "1500"],
                                  stdout=subprocess.PIPE, text=True) for r in range(2)]
        outs = [json.loads(p.communicate(timeout=60)[0].strip().splitlines()[-1]) for p in procs]
        assert all("error" in o and "stale" in o["error"] for o in outs), outs
    finally:
        try:
            os.unlink("/dev/shm" + name)
        except OSError:
            pass
