"""CPU: bench.py's own launcher (`python bench.py --gpus N` without torchrun: spawn_ranks) - the part of the N > 1 flow
that runs before anything touches a GPU. ADVICE r3: rank 0's stdout was a pipe read only after rank 0 had exited, so a
rank 0 that wrote more than a pipe buffer blocked in write() until the launcher's time limit; and on that limit the
children were killed but not reaped and whatever rank 0 had printed was dropped."""
import argparse
import io
import os
import sys
import time
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

RANK = r'''
import os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and os.environ["LOCAL_RANK"] == str(rank)
mode = sys.argv[1]
if mode == "big":
    if rank == 0:
        sys.stdout.write("x" * 400000 + "\n")
        sys.stdout.write('{"n_gpus": %d}\n' % world)
    sys.exit(0)
if mode == "hang":
    if rank == 0:
        sys.stdout.write("partial evidence from rank 0\n")
        sys.stdout.flush()
    time.sleep(600)
if mode == "fail":
    if rank == 1:
        sys.exit(7)
    time.sleep(600)
'''


def run(tmp_path, mode, gpus=2, limit=None):
    script = tmp_path / "rank.py"
    script.write_text(RANK)
    if limit is not None:
        os.environ["PLL_BENCH_TIMEOUT_S"] = str(limit)
    try:
        buf = io.StringIO()
        t0 = time.time()
        with redirect_stdout(buf):
            rc = bench.spawn_ranks(argparse.Namespace(gpus=gpus), command=[sys.executable, str(script), mode])
        return rc, buf.getvalue(), time.time() - t0
    finally:
        os.environ.pop("PLL_BENCH_TIMEOUT_S", None)


def test_a_talkative_rank_zero_does_not_block_the_launcher(tmp_path):
    rc, out, dt = run(tmp_path, "big", gpus=3, limit=60)
    assert rc == 0 and dt < 30
    assert out.endswith('{"n_gpus": 3}\n') and len(out) > 400000


def test_the_time_limit_reaps_the_ranks_and_keeps_what_rank_zero_said(tmp_path):
    rc, out, dt = run(tmp_path, "hang", limit=2)
    assert rc == 124 and dt < 40
    assert "partial evidence from rank 0" in out


def test_a_failing_rank_stops_the_others_and_its_code_is_returned(tmp_path):
    rc, out, dt = run(tmp_path, "fail", gpus=3, limit=120)
    assert rc == 7 and dt < 60
