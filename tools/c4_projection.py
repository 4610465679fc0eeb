#!/usr/bin/env python3
"""One GPU: what BASELINE configs[3] would do on N GPUs. The pattern-sorted 1M-site alignment is cut into
N shards exactly as bench.py --gpus N does; the whole alignment and every shard are timed one after another
on this device with bench.py's own step (pll_update_partials_rep + edge log-likelihood), once with the class
maps re-used (update_repeats = 0) and once with them recomputed by every step (the reference's pll_update_partials).
projected speedup = t(whole) / (max over shards t(shard) + exchange). The exchange is MEASURED on this box:
tools/group_latency.c runs --shards processes that meet in the library's fixed-order shared-memory exchange
(pll_gpu_group_sum, csrc/host/group.c) - its time per step is what a step adds behind a result that the device
has already written to host memory. --exchange-us overrides (e.g. 15 for the RCCL all-reduce as bench.py timed
it at world size 1 in round 2).

    python tools/c4_projection.py [--shards 8] [--steps 20] [--shard-only R]   -> one JSON line
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sites", type=int, default=0)
    ap.add_argument("--exchange-us", type=float, default=-1.0, help="< 0: measure the shared-memory exchange with tools/group_latency.c")
    ap.add_argument("--cut", default="equal", choices=["equal", "balanced"])
    ap.add_argument("--blocks", type=int, default=5)
    ap.add_argument("--driver", default="c", choices=["c", "python"])
    ap.add_argument("--shard-only", type=int, default=-1, help="time just this shard (profiling runs)")
    ap.add_argument("--unsorted", action="store_true", help="cut the alignment as generated (no pattern sort)")
    args = ap.parse_args()
    from pllamd import api, driver, sharding

    class H:  # no collectives here
        dist = None
        world = 1
        rank = 0
        on_device = False

        @staticmethod
        def max_over_ranks(v):
            return v

    cfg = bench.CONFIGS["c4"]
    lib = api.PllLib()
    full = bench.build_case(cfg, args.sites or cfg["sites"], api.SITE_REPEATS)
    total_sites = full.sites
    if not args.unsorted:
        full = sharding.sort_columns(lib, full)
    out = {"total_sites": total_sites, "patterns": full.sites, "shards": args.shards, "sorted": not args.unsorted}

    def time_case(case):
        r = bench.Runner(H, lib, api, driver, case, True, reduce=None, c_driver=args.driver == "c")
        blocks, lnl = r.timed(args.warmup, args.steps, args.blocks)
        # the same step as the reference's pll_update_partials defines it: class maps recomputed by every step
        bcm, _ = r.timed_with_class_maps(args.warmup, args.steps, args.blocks)
        lv = r.level_entries()
        extra, _ = r.repeats_update_ms(reps=3)
        r.close()
        return bench.block_stats(blocks, args.steps)[0], lnl, lv, extra, bench.block_stats(bcm, args.steps)[0]

    def measure_exchange():
        import subprocess
        exe = "/tmp/pll_group_latency"
        subprocess.check_call(["gcc", "-O2", os.path.join(ROOT, "tools", "group_latency.c"), "-o", exe, "-ldl"])
        line = subprocess.check_output([exe, os.path.join(ROOT, "libpll-2_amd", "csrc", "libpll_amd.so"), str(args.shards), "200000"], text=True)
        return json.loads(line.strip().splitlines()[-1])["us_per_exchange"]

    if args.shard_only < 0:
        t1, lnl1, lv1, rep1, t1cm = time_case(full)
        out.update(t1_ms=round(t1, 4), t1_ms_with_class_maps=round(t1cm, 4), lnl_unsharded=lnl1, entries_per_level_unsharded=lv1, repeats_update_ms_unsharded=round(rep1, 3))
    ts, lnls, lvs, reps, tcms = [], [], [], [], []
    bounds = sharding.balanced_bounds(full, args.shards) if args.cut == "balanced" else sharding.shard_bounds(full.sites, args.shards)
    out.update(cut=args.cut, shard_sites=[hi - lo for lo, hi in bounds], driver=args.driver, blocks=args.blocks)
    for r in (range(args.shards) if args.shard_only < 0 else [args.shard_only]):
        t, lnl, lv, rep, tcm = time_case(sharding.shard_case(full, r, args.shards, bounds))
        ts.append(round(t, 4))
        tcms.append(round(tcm, 4))
        lnls.append(lnl)
        lvs.append(lv)
        reps.append(round(rep, 3))
    out.update(shard_ms=ts, shard_ms_with_class_maps=tcms, shard_entries_per_level=lvs, shard_repeats_update_ms=reps)
    if args.shard_only < 0:
        ex = args.exchange_us if args.exchange_us >= 0 else measure_exchange()
        tn = max(ts) + ex * 1e-3
        out.update(exchange_us=round(ex, 3), exchange="measured: tools/group_latency.c, %d processes on this host" % args.shards if args.exchange_us < 0 else "given",
                   projected_tN_ms=round(tn, 4), projected_speedup=round(out["t1_ms"] / tn, 3),
                   projected_tN_ms_with_class_maps=round(max(tcms) + ex * 1e-3, 4),
                   projected_speedup_with_class_maps=round(out["t1_ms_with_class_maps"] / (max(tcms) + ex * 1e-3), 3),
                   lnl_sum_of_shards=float(sum(lnls)), lnl_rel_diff=abs(sum(lnls) - lnl1) / abs(lnl1),
                   entries_sum_over_shards_div_unsharded=round(sum(sum(x) for x in lvs) / sum(lv1), 4))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
