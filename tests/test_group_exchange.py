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


def _stale_segment(name, size, joined, left=0, seated=()):
    """the bytes a run of `size` ranks leaves under /dev/shm when it dies: header {magic, size, joined, left, poisoned},
    one owner word per seated rank, the slots (csrc/host/group.c)"""
    owners = (size * 4 + 63) // 64 * 64
    stale = bytearray(64 + owners + size * 2 * 64)
    stale[0:8] = (0x504c4c4752503032).to_bytes(8, "little")
    stale[8:12] = size.to_bytes(4, "little")
    stale[12:16] = joined.to_bytes(4, "little")
    stale[16:20] = left.to_bytes(4, "little")
    for r in seated:
        stale[64 + 4 * r:68 + 4 * r] = (0x80000000 | 4242).to_bytes(4, "little")
    with open("/dev/shm" + name, "wb") as f:
        f.write(stale)


def _run_two(name, steps=3, timeout_ms=20000):
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "group_worker.py"), name, str(r), "2", str(steps), "1", str(timeout_ms)],
                              stdout=subprocess.PIPE, text=True) for r in range(2)]
    return [json.loads(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in procs]


@pytest.mark.parametrize("joined,left,seated", [(2, 0, (0, 1)), (1, 0, (0,)), (1, 0, (1,)), (1, 1, (0,)), (2, 1, (0, 1))],
                         ids=["killed-after-the-barrier", "killed-in-join-rank0", "killed-in-join-rank1", "gave-up-in-join", "one-rank-left"])
def test_a_finished_run_removes_its_name_and_a_stale_segment_is_replaced(joined, left, seated):
    """the last rank to leave unlinks the segment; one that a killed or failed run of the SAME size left behind -
    ranks counted as joined that will never come (the first ranks of the next run walked through the
    barrier at once and the others spun until their time-out) - is recognised by the next run under that name,
    poisoned, removed and replaced: the new run completes, nobody waits for a time-out"""
    name = "/pllamd-test-" + uuid.uuid4().hex[:12]
    outs = _run_two(name)
    assert all("error" not in o for o in outs), outs
    assert not os.path.exists("/dev/shm" + name)
    _stale_segment(name, 2, joined, left, seated)
    try:
        import time
        t0 = time.time()
        outs = _run_two(name, timeout_ms=8000)
        assert all("error" not in o for o in outs), outs
        assert outs[0]["sums"] == outs[1]["sums"]
        assert time.time() - t0 < 7.0  # (two interpreter starts; far from the 8 s time-out)
        assert not os.path.exists("/dev/shm" + name)
    finally:
        try:
            os.unlink("/dev/shm" + name)
        except OSError:
            pass


def test_a_failed_join_leaves_nothing_behind():
    """a rank whose join times out removes the name before it goes: the next run under it starts from nothing"""
    name = "/pllamd-test-" + uuid.uuid4().hex[:12]
    outs = run_ranks(2, steps=1, timeout_ms=300, ranks=[0], name=name + "x")
    assert "error" in outs[0] and "joined" in outs[0]["error"]
    a = subprocess.Popen([sys.executable, os.path.join(HERE, "group_worker.py"), name, "0", "2", "1", "1", "300"], stdout=subprocess.PIPE, text=True)
    oa = json.loads(a.communicate(timeout=60)[0].strip().splitlines()[-1])
    assert "error" in oa
    assert not os.path.exists("/dev/shm" + name)
    outs = _run_two(name)
    assert all("error" not in o for o in outs), outs


def test_a_rank_seated_twice_is_an_error_not_a_hang():
    """two processes claim rank 0 of a group of two: the second finds the seat taken, gives the segment up (the first
    stops waiting at once) and both fail well before the time-out"""
    import time
    name = "/pllamd-test-" + uuid.uuid4().hex[:12]
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "group_worker.py"), name, "0", "2", "1", "1", "2500"],
                              stdout=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [json.loads(p.communicate(timeout=60)[0].strip().splitlines()[-1]) for p in procs]
    try:
        os.unlink("/dev/shm" + name)
    except OSError:
        pass
    assert all("error" in o for o in outs), outs
    assert time.time() - t0 < 20


def test_sharded_derivatives_with_failing_ranks_return_instead_of_waiting():
    """pll_gpu_group_likelihood_derivatives when EVERY rank's evaluation fails (host-only shells here): each rank still
    takes part in the exchange (its failure is counted in a third value), gets PLL_FAILURE with its own error and nobody waits for a time-out"""
    import time
    name = "/pllamd-test-" + uuid.uuid4().hex[:12]
    code = r'''
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(%r, "libpll-2_amd"))
os.environ["PLL_AMD_HOST_ONLY"] = "1"
import numpy as np
from pllamd import api
lib = api.PllLib()
rank = int(sys.argv[1])
g = lib.pll_gpu_group_join(%r.encode(), rank, 2, 20000)
p = lib.pll_partition_create(4, 2, 4, 32, 1, 5, 4, 2, api.ARCH_AVX2)
fi = np.zeros(4, dtype=np.uint32); st = np.zeros(32 * 16); d1, d2 = C.c_double(), C.c_double()
ok = lib.pll_gpu_group_likelihood_derivatives(p, g, -1, -1, 0.1, api.uptr(fi), api.dptr(st), C.byref(d1), C.byref(d2))
print(json.dumps(dict(ok=bool(ok), errno=lib.errno())))
lib.pll_gpu_group_leave(g)
''' % (os.path.dirname(HERE), name)
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for r in range(2)]
    outs = [json.loads(p.communicate(timeout=60)[0].strip().splitlines()[-1]) for p in procs]
    assert all(not o["ok"] and o["errno"] == 900 for o in outs), outs
    assert time.time() - t0 < 15
