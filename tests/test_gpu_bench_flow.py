"""GPU: bench.py's own flows as the driver runs them (nothing ran main_strong; `--gpus N` did not
start N ranks). Rehearsed on ONE device: two ranks share GPU 0 (PLL_BENCH_SAME_DEVICE=1), gloo as the control
plane, the path's exchange through the library's shared-memory group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env=None, timeout=900, rc=0):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == rc, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("cut", ["balanced", "equal"])
def test_gpus_2_starts_two_ranks_and_runs_the_strong_scaling_flow(cut):
    out = run_bench("--gpus", "2", "--backend", "gloo", "--sites", "200000", "--steps", "3", "--blocks", "2", "--warmup", "2", "--cut", cut,
                    env={"PLL_BENCH_SAME_DEVICE": "1"})
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 3
    assert out["lnl_rel_err_vs_unsharded"] <= 1e-12
    assert out["exchange"]["reduce"] == "peer" and out["exchange"]["peer_exchange_alone_us"] < 100
    assert len(out["config"]["sites_per_gpu"]) == 2 and sum(out["config"]["sites_per_gpu"]) == out["config"]["patterns"]
    assert out["value"] > 0 and out["ms_per_step_min"] <= out["ms_per_step"] <= out["ms_per_step_max"]
    assert "roofline" in out and out["t1_ms"] > 0
    # which call is timed, and the step as the reference's pll_update_partials defines it beside it
    # three timed figures (round 6): maps re-used, the default call on the unchanged tree, every map forced to be computed again
    assert "update_repeats = 0" in out["config"]["step"] and "update_repeats = 1" in out["config"]["step_default_call"]
    assert "PLL_GPU_FORGET_REPEATS" in out["config"]["step_with_class_maps"]
    assert out["ms_per_step_with_class_maps"] > out["ms_per_step"] and out["t1_ms_with_class_maps"] > out["t1_ms"]
    assert out["ms_per_step_with_class_maps"] > out["ms_per_step_default_call"] > 0 and out["t1_ms_with_class_maps"] > out["t1_ms_default_call"] > 0
    assert out["speedup_with_class_maps"] > 0 and out["value_with_class_maps"] > 0 and out["speedup_default_call"] > 0
    assert abs(out["lnl_with_class_maps"] - out["lnl"]) <= 1e-12 * abs(out["lnl"])
    assert abs(out["lnl_default_call"] - out["lnl"]) <= 1e-12 * abs(out["lnl"])
    assert out["exchange_timed"] == "peer" and len(out["cpu_affinity"]) == 2


@pytest.mark.parametrize("reduce", ["rccl", "peer"])
def test_the_rccl_form_of_the_exchange_runs_through_the_librarys_entry_point(reduce):
    """`--reduce rccl` called torch's all_reduce, never pll_gpu_edge_loglikelihood_allreduce. Now bench.py
    makes a communicator of its ranks (ncclGetUniqueId over the control plane, ncclCommInitRank) and its steps call the
    library's C entry point from the C step loop. RCCL refuses two ranks on one device, so the one-GPU rehearsal is the
    sharded flow with a world of ONE rank and a real communicator (several ranks: tests/test_gpu_c_caller.py through
    the stream-ordered stand-in); with --reduce peer the RCCL form is the line's second, guarded leg."""
    out = run_bench("--gpus", "1", "--backend", "nccl", "--reduce", reduce, "--sites", "200000", "--steps", "3", "--blocks", "2", "--warmup", "2",
                    env={"PLL_BENCH_FORCE_DIST": "1", "PLL_BENCH_FORCE_STRONG": "1"})
    ex = out["exchange"]
    assert ex["reduce"] == reduce and "rccl_error" not in ex
    assert "pll_gpu_edge_loglikelihood_allreduce" in ex["rccl_via"], ex
    assert ex["rccl_ms_per_step"] > 0
    assert out["lnl_rel_err_vs_unsharded"] <= 1e-12
    if reduce == "peer":
        assert ex["rccl_lnl_rel_diff"] <= 1e-14


def test_a_stuck_rccl_leg_does_not_cost_the_line():
    """the RCCL form of the exchange is the line's LAST leg and runs under a watchdog: when it does not come back in
    time (here: a limit it cannot meet) rank 0 still prints the one line, complete but for that leg - and every rank
    exits NON-ZERO: a collective that never completes is a hang, not a result"""
    out = run_bench("--gpus", "1", "--backend", "nccl", "--reduce", "peer", "--sites", "200000", "--steps", "3", "--blocks", "2", "--warmup", "2",
                    env={"PLL_BENCH_FORCE_DIST": "1", "PLL_BENCH_FORCE_STRONG": "1", "PLL_BENCH_RCCL_LEG_TIMEOUT_S": "0.02"}, rc=3)
    ex = out["exchange"]
    assert "watchdog" in ex["rccl_error"] and "rccl_ms_per_step" not in ex
    assert out["scaling"] == "strong" and out["t1_ms"] > 0 and out["speedup"] > 0 and "roofline" in out
    assert out["lnl_rel_err_vs_unsharded"] <= 1e-12


def test_repeats_line_times_both_forms_of_the_step():
    """a SITE_REPEATS line says which call its `value` is timed with and carries the step with the class maps
    recomputed - pll_update_partials as the reference defines it - as a second timed figure"""
    out = run_bench("--config", "c4", "--sites", "100000", "--steps", "3", "--blocks", "2", "--warmup", "2", "--no-cpu")
    assert "update_repeats = 0" in out["config"]["step"] and "update_repeats = 1" in out["config"]["step_default_call"]
    assert "PLL_GPU_FORGET_REPEATS" in out["config"]["step_with_class_maps"]
    assert out["ms_per_step_with_class_maps"] > out["ms_per_step"] > 0
    assert out["ms_per_step_with_class_maps"] > out["ms_per_step_default_call"] > 0   # an unchanged tree computes no maps
    assert 0 < out["value_with_class_maps"] < out["value"] and out["value_default_call"] > out["value_with_class_maps"]
    assert abs(out["lnl_with_class_maps"] - out["lnl"]) <= 1e-12 * abs(out["lnl"])
    assert abs(out["lnl_default_call"] - out["lnl"]) <= 1e-12 * abs(out["lnl"])


def test_default_line_carries_the_contract_fields():
    out = run_bench("--steps", "5", "--blocks", "3", "--warmup", "2")
    assert out["n_gpus"] == 1 and out["scaling"] is None and out["blocks"] == 3 and out["dtype"] == "f64"
    assert out["lnl_rel_err_pinned"] <= 1e-10 and out["lnl_rel_err"] <= 1e-10
    rf, cb = out["roofline"], out["cpu_baseline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and rf["achieved"] > 0
    assert (rf["traffic"] is None) == (rf["traffic_source"] is None)
    # round 6: the HBM bytes of the dominant launch come from two rocprofv3 counter passes of THIS run (within 2 % of the
    # algorithmic bytes: no wasted re-reads), the committed summary only where the passes could not be taken - and then it says why
    src = rf["traffic_source"]
    assert src.get("measured") == "in this run" or "live_passes" in src, src
    if src.get("measured") == "in this run":
        assert abs(rf["traffic"] / rf["algorithmic_bytes_per_launch"] - 1.0) < 0.02, (rf["traffic"], rf["algorithmic_bytes_per_launch"])
    assert 0 < rf["step"]["frac"] < 1 and abs(rf["step"]["ms"] - out["ms_per_step"]) < 1e-3 and "library" in rf["frac_numerator"]
    assert "NOT a roofline fraction" in rf["unfused_equivalent"]["label"]
    assert cb["kind"] == "reference" and len(cb["samples"]) == 3 and cb["one_core"]["cores"] == 1 and cb["pattern_tip"]["value"] > 0


def test_python_and_c_drivers_give_the_same_bits():
    a = run_bench("--steps", "3", "--blocks", "1", "--no-cpu", "--driver", "c", "--sites", "20000")
    b = run_bench("--steps", "3", "--blocks", "1", "--no-cpu", "--driver", "python", "--sites", "20000")
    assert a["lnl"] == b["lnl"]
