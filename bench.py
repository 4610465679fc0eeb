#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X partial-likelihood hot path.

One "step" = one pass of the hot path over one synthetic alignment that is already resident in
HBM: pll_update_partials over the full post-order operation list (62 ops for the 64-taxon
balanced tree) followed by pll_compute_edge_loglikelihood at the root edge, which synchronises and
returns the log-likelihood (and, for N > 1, one all-reduce of that double over RCCL).

    metric  M site-CLV-updates/s = sites * ops * steps / t / 1e6          (BASELINE.json)
    config  configs[1]: 4-state DNA, 4 Gamma rates, 64-taxon balanced tree, 100k synthetic sites
            (tips as ordinary 0/1 CLVs: every update is inner x inner, 384 B + scalers)

Multi-GPU: one process per GPU (torch.distributed / RCCL), sites sharded - every rank owns an
independent partition over its own 100k-site shard (weak scaling); the only exchange is the
all-reduce of the final log-likelihood.

Besides the contract fields the JSON line carries
    roofline      HIP-event timing of the CLV-update kernel launches of one traversal on the
                  partition's stream vs the 8 TB/s HBM3E peak; algorithmic bytes per DESIGN.md
    cpu_baseline  the reference's own AVX2 path (oracle/_ref/libpll_ref.so, built from the
                  reference sources) timed on this host's cores on the same workload, rank 0 only
    lnl_rel_err   |lnL_gpu - lnL_reference| / |lnL_reference| on that workload
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# fp64 matrix pipe: 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz (v_mfma_f64_4x4x4: 512 flop / 16 clk).
# The guide lists no fp64 row; tools/mfma_f64_probe.hip measures 70.6 TF on this part
# (profiles/r1_fp64_issue_rates.md) - the spec figure is the one priced against.
FP64_MFMA_PEAK_TF = 78.6

CONFIGS = {
    # name: (states, tips, sites, attributes-as-names, description)
    "c2": dict(states=4, tips=64, sites=100000, desc="4-state DNA GTR+G4, 64-taxon balanced tree, 100k sites"),
    "c3": dict(states=20, tips=64, sites=50000, desc="20-state protein (LG)+G4, 64 taxa, 50k sites"),
    "c5": dict(states=61, tips=32, sites=20000, desc="61-state codon stand-in +G4, 32 taxa, 20k sites"),
    # configs[3]: 1M sites over 8 GPUs = 125k sites per GPU, PLL_ATTRIB_SITE_REPEATS
    # not a BASELINE configuration: C3's shape with PLL_ATTRIB_SITE_REPEATS (what the any-state gather path costs)
    "c3r": dict(states=20, tips=64, sites=50000, repeats=True, desc="20-state protein (LG)+G4, 64 taxa, 50k sites, SITE_REPEATS"),
    "c4": dict(states=4, tips=128, sites=125000, repeats=True,
               desc="4-state DNA GTR+G4, 128 taxa, 125k-site shard of the 1M-site alignment, SITE_REPEATS"),
}


def build_case(cfg, sites, seed, attributes, tree="balanced"):
    import numpy as np
    from pllamd import workload as W
    kw = {}
    if cfg["states"] == 20:
        lg = np.load(os.path.join(ROOT, "tests", "golden", "model_lg.npz"))
        kw.update(exch=lg["rates"], freqs=lg["freqs"])
    return W.make_case("bench", cfg["states"], cfg["tips"], sites, attributes=attributes, seed=seed, tree=tree, **kw)


def tips_are_codes(case, api):
    """tips set from sequences reach the device as one-byte codes (PATTERN_TIP, or the library's
    compact form of an indicator CLV) unless PLL_AMD_NO_TIP_CODES=1 forces dense tip CLVs"""
    if case.attributes & api.PATTERN_TIP:
        return True
    return case.sequences is not None and os.environ.get("PLL_AMD_NO_TIP_CODES", "0") in ("", "0")


def op_bytes(case, api, ops, entries=None):
    """algorithmic HBM bytes of the given ops: per update 3 CLV entries for
    inner x inner, 2 + 1 B for tip x inner, 1 + 2 B for tip x tip, plus 4 B per scaler touched.
    entries: {parent clv: class count} under site repeats (only that many updates are computed;
    the 12 B of class-map gathers per update are not counted)"""
    s, r, n = case.states, case.rate_cats, case.sites
    entry = s * r * 8
    pattern_tip = tips_are_codes(case, api)
    per_rate = r if (case.attributes & api.RATE_SCALERS) else 1
    total = 0
    for (pc, psc, c1, m1, s1, c2, m2, s2) in ops:
        b = entry
        for c, sc in ((c1, s1), (c2, s2)):
            if pattern_tip and c < case.tips:
                b += 1
            else:
                b += entry
                if sc >= 0:
                    b += 4 * per_rate
        if psc >= 0:
            b += 4 * per_rate
        total += b * (entries[pc] if entries else n)
    return total


def cpu_baseline(case, api, driver, budget_s=12.0):
    """reference AVX2 path on the host cores: T threads, each its own partition over sites/T
    contiguous sites (how applications parallelise libpll), traversal repeated to fill ~budget_s
    of CPU work. Falls back to the scalar restatement (kind 'port', 1 core, small sample)."""
    import numpy as np
    from oracle import oracle as O
    ops = len(case.op_batches[0])
    if os.path.exists(O.REF_LIB):
        ref = api.PllLib(O.REF_LIB)
        cores = max(1, min(len(os.sched_getaffinity(0)), 16))
        bounds = np.linspace(0, case.sites, cores + 1).astype(int)
        shards = []
        for t in range(cores):
            lo, hi = int(bounds[t]), int(bounds[t + 1])
            sub = driver.Case(name=f"shard{t}", states=case.states, rate_cats=case.rate_cats, tips=case.tips,
                              sites=hi - lo, pmatrix=case.pmatrix, freqs=case.freqs, op_batches=case.op_batches,
                              edges=case.edges, charmap=case.charmap,
                              sequences=[sq[lo:hi] for sq in case.sequences], attributes=case.attributes,
                              clv_buffers=case.clv_buffers, scale_buffers=case.scale_buffers)
            shards.append(driver.Session(ref, sub, api.ARCH_AVX2))
        # one traversal to size the sample, then reps traversals timed
        t0 = time.perf_counter()
        shards[0].update_partials()
        one = (time.perf_counter() - t0) * cores  # core-seconds per full traversal (approx.)
        reps = int(max(2, min(200, budget_s / max(one, 1e-4))))
        lnls = [0.0] * cores

        rep_mode = bool(case.attributes & api.SITE_REPEATS)
        if rep_mode:  # class maps once, outside the timed region (same policy as the GPU leg)
            for sh in shards:
                sh.update_partials(update_repeats=1)

        def work(i):
            for _ in range(reps):
                shards[i].update_partials(update_repeats=0 if rep_mode else 1)
            lnls[i] = shards[i].edge_lnl(case.edges[0], persite=False)[0]

        th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        for sh in shards:
            sh.close()
        return dict(value=case.sites * ops * reps / dt / 1e6, unit="M site-CLV-updates/s", cores=cores, kind="reference",
                    sample=f"{reps} full traversals ({ops} ops) of the same {case.sites}-site alignment, reference "
                           f"PLL_ATTRIB_ARCH_AVX2 path, {cores} threads x sites/{cores} partitions, {dt:.1f} s wall"), float(sum(lnls))
    # no reference library on this host: time the scalar restatement on a small slice
    n = min(case.sites, 2000)
    sub = driver.Case(name="slice", states=case.states, rate_cats=case.rate_cats, tips=case.tips, sites=n,
                      pmatrix=case.pmatrix, freqs=case.freqs, op_batches=case.op_batches, edges=case.edges,
                      charmap=case.charmap, sequences=[sq[:n] for sq in case.sequences], attributes=case.attributes,
                      clv_buffers=case.clv_buffers, scale_buffers=case.scale_buffers)
    t0 = time.perf_counter()
    O.run_case(sub)
    dt = time.perf_counter() - t0
    return dict(value=n * ops / dt / 1e6, unit="M site-CLV-updates/s", cores=1, kind="port",
                sample=f"one traversal of the first {n} sites with the scalar C restatement (oracle/pll_oracle.c)"), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--sites", type=int, default=0, help="override sites per GPU")
    ap.add_argument("--pattern-tip", action="store_true", help="PLL_ATTRIB_PATTERN_TIP variant (tip codes instead of tip CLVs)")
    ap.add_argument("--tree", default="balanced", choices=["balanced", "random", "caterpillar"],
                    help="topology (BASELINE's configs are balanced; the others show what irregular level structures cost)")
    ap.add_argument("--taxa", type=int, default=0, help="override the number of taxa")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + PLL_BENCH_SAME_DEVICE=1 rehearses the N>1 flow on a one-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    same_device = os.environ.get("PLL_BENCH_SAME_DEVICE") == "1"
    os.environ.setdefault("PLL_AMD_DEVICE", "0" if same_device else str(local))
    dist = None
    torch = None
    dist_mode = world > 1 or os.environ.get("PLL_BENCH_FORCE_DIST") == "1"
    if dist_mode:  # FORCE_DIST: rehearse the collective path at world 1
        import torch
        import torch.distributed as dist
        # RCCL prints a version banner on stdout when it initialises: keep stdout for the one JSON line
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    tdev = f"cuda:{local}" if (dist and args.backend == "nccl") else "cpu"

    def tsync():
        if dist and args.backend == "nccl":
            torch.cuda.synchronize()
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import numpy as np
    from pllamd import api, driver

    cfg = dict(CONFIGS[args.config])
    if args.taxa:
        cfg["tips"] = args.taxa
        cfg["desc"] += f" [{args.taxa} taxa]"
    if args.tree != "balanced":
        cfg["desc"] = cfg["desc"].replace("balanced tree", "") + f" [{args.tree} tree]"
    sites = args.sites or cfg["sites"]
    attributes = api.PATTERN_TIP if args.pattern_tip else 0
    if cfg.get("repeats"):
        attributes |= api.SITE_REPEATS
    case = build_case(cfg, sites, seed=1000 + rank, attributes=attributes, tree=args.tree)
    nops = len(case.op_batches[0])
    lib = api.PllLib()
    sess = driver.Session(lib, case, api.ARCH_AVX2)  # uploads happen on first use (warm-up)
    edge = case.edges[0]
    red = torch.zeros(2, dtype=torch.float64, device=tdev) if dist else None
    on_device = bool(dist) and args.backend == "nccl"
    if on_device:
        # the partition works on torch's stream, the shard's lnL stays in HBM ({lnL, sequence} in `red`)
        # and RCCL reduces it there: no host round trip before the one exchange of the path
        tstream = torch.cuda.Stream()
        torch.cuda.set_stream(tstream)
        if not lib.pll_gpu_set_stream(sess.p, tstream.cuda_stream):
            raise SystemExit(f"pll_gpu_set_stream: [{lib.errno()}] {lib.errmsg()}")
        fi = np.ascontiguousarray(case.freqs_indices, dtype=np.uint32)
        # the reduced {lnL, sequence} pair comes back through pinned host memory that the host polls,
        # like the single-GPU path does (a stream synchronise costs more than the copy)
        pinned = torch.zeros(2, dtype=torch.float64).pin_memory()
        pview = pinned.numpy()
        expected = [None]

    # site repeats: class maps are computed once (host, integer) and re-used, as applications do
    # between topology changes: pll_update_partials_rep(..., update_repeats = 0)
    upd = [1]

    def step():
        sess.update_partials(update_repeats=upd[0])
        if cfg.get("repeats"):
            upd[0] = 0
        if on_device:
            if not lib.pll_gpu_edge_loglikelihood_async(sess.p, edge[0], edge[1], edge[2], edge[3], edge[4], api.uptr(fi),
                                                        red.data_ptr()):
                raise SystemExit(f"pll_gpu_edge_loglikelihood_async: [{lib.errno()}] {lib.errmsg()}")
            # the path's one exchange: sum of the shards' log-likelihoods (word 0); word 1 = every rank's
            # call sequence number, so its sum tells the host which evaluation the pair belongs to
            dist.all_reduce(red)
            pinned[0:1].copy_(red[0:1], non_blocking=True)
            pinned[1:2].copy_(red[1:2], non_blocking=True)  # stream-ordered behind the value
            if expected[0] is not None:
                expected[0] += world
                t_spin = time.perf_counter()
                while pview[1] != expected[0]:
                    if time.perf_counter() - t_spin > 0.02:
                        expected[0] = None
                        break
            if expected[0] is None:
                torch.cuda.current_stream().synchronize()
                expected[0] = float(pview[1])
            return float(pview[0])
        v, _ = sess.edge_lnl(edge, persite=False)
        if dist:
            red[0] = v
            dist.all_reduce(red[:1])
            v = float(red[0].item())
        return v

    def fence():
        lib.pll_gpu_synchronize(sess.p)
        if dist:
            tsync()
            dist.barrier()
            tsync()

    lnl = None
    for _ in range(args.warmup):
        lnl = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lnl = step()
    fence()
    dt = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if not np.isfinite(lnl):
        raise SystemExit(f"hot path failed: lnL = {lnl} [{lib.errno()}] {lib.errmsg()}")

    total_sites = sites * world
    value = total_sites * nops * args.steps / dt / 1e6

    # ---- full traversal alone (no lnL), HIP events on the partition's stream
    reps = 20
    lib.pll_gpu_synchronize(sess.p)
    lib.pll_gpu_timer_start(sess.p)
    for _ in range(reps):
        sess.update_partials(update_repeats=upd[0])
    ms_full = lib.pll_gpu_timer_stop(sess.p)
    launches_full = lib.pll_gpu_last_launch_count(sess.p)
    bytes_full = lib.pll_gpu_last_algorithmic_bytes(sess.p)
    # ---- roofline leg: the DOMINANT kernel = the inner x inner CLV update. Its launches are timed by
    # re-running the part of the traversal whose children are both inner CLVs (a valid partial
    # traversal: the tip-level parents it reads are already in HBM)
    all_ops = case.op_batches[0]
    codes = tips_are_codes(case, api)
    ii_ops = [op for op in all_ops if op[2] >= case.tips and op[5] >= case.tips] if codes else list(all_ops)
    if not ii_ops:
        ii_ops = list(all_ops)
    fused = cfg["states"] == 4 and not cfg.get("repeats") and not os.environ.get("PLL_AMD_NO_FUSE", "0").strip("0")
    cc = fused and codes and not os.environ.get("PLL_AMD_NO_FUSE_CC", "0").strip("0")
    if cc:
        # 4x4 with tips as codes: most of the step is ONE launch, the groups of seven ops over complete
        # 8-tip subtrees (k_partials_dna_cc<5,5>). The leg re-runs exactly those ops: everything within
        # three levels of the tips.
        depth = {t: 0 for t in range(case.tips)}
        for op in all_ops:
            depth[op[0]] = 1 + max(depth[op[2]], depth[op[5]])
        low = [op for op in all_ops if depth[op[0]] <= 3]
        if len(low) == 7 * (case.tips // 8):
            ii_ops = low
        else:
            cc = False
    if args.tree != "balanced":  # irregular levels: no single dominant launch shape - the leg is the whole traversal
        ii_ops, cc = list(all_ops), False
    ii_arr = api.make_ops(ii_ops)
    for _ in range(3):
        lib.pll_update_partials_rep(sess.p, ii_arr, len(ii_ops), 0)
    lib.pll_gpu_synchronize(sess.p)
    lib.pll_gpu_timer_start(sess.p)
    for _ in range(reps):
        lib.pll_update_partials_rep(sess.p, ii_arr, len(ii_ops), 0)
    ms = lib.pll_gpu_timer_stop(sess.p)
    launches = lib.pll_gpu_last_launch_count(sess.p)
    entries = {op[0]: sess.entries(op[0]) for op in all_ops} if cfg.get("repeats") else None
    # bytes the launches had to move AS GROUPED (fused producer/consumer groups do not read the
    # intermediate CLVs back): reported by the library; op_bytes() = the same ops launched one by one
    unfused_bytes = op_bytes(case, api, ii_ops, entries)
    trav_bytes = lib.pll_gpu_last_algorithmic_bytes(sess.p) or unfused_bytes
    per_launch_bytes = trav_bytes / launches
    per_launch_ms = ms / reps / launches
    achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
    if os.path.exists(tfile) and not args.pattern_tip and not args.sites and not args.taxa and args.tree == "balanced":  # PMC-derived HBM bytes per launch of the same command (profiles/README.md)
        traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
    mfma = cfg["states"] > 32 and not os.environ.get("PLL_AMD_NO_MFMA", "0").strip("0")
    kernel = {4: "k_partials_dna_cc<5,5>%.0s" if cc else "k_partials_dna_fused<4,4>%.0s" if fused else "k_partials_dna<false,false,%s>", 20: "k_partials_tiled<20,false,false,%s>",
              61: "k_partials_mfma<false,false,%s>" if mfma else "k_partials_tiled<32,false,false,%s>"
              }[cfg["states"]] % ("true" if cfg.get("repeats") else "false")
    if args.tree != "balanced":
        kernel = "all update launches of the traversal"
    if mfma:
        # 33..64 states sit past the fp64 ridge: the bounding line is the fp64 matrix pipe.
        # Algorithmic flop per update = the two 64-padded matrix-vector products the MFMA tiles
        # perform per (site, rate) would overstate it; count the reference's own arithmetic:
        # 2 children x S x S multiply-adds + S products per (site, rate)  (core_partials.c:739-757)
        S, R = cfg["states"], 4
        flop_per_update = R * (2 * 2 * S * S + S)
        per_launch_flop = flop_per_update * sum((entries[op[0]] if entries else sites) for op in ii_ops) / launches
        tf = per_launch_flop / (per_launch_ms * 1e-3) / 1e12
        roofline = dict(bound="mfma", achieved=round(tf, 2), peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s",
                        frac=round(tf / FP64_MFMA_PEAK_TF, 4), traffic=None, kernel=kernel,
                        launches=launches, ops_in_those_launches=len(ii_ops), avg_launch_ms=round(per_launch_ms, 5),
                        algorithmic_flop_per_launch=int(per_launch_flop), algorithmic_bytes_per_launch=int(per_launch_bytes),
                        hbm_GBps=round(achieved, 1))
    else:
        roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, kernel=kernel,
                        launches=launches, ops_in_those_launches=len(ii_ops), avg_launch_ms=round(per_launch_ms, 5),
                        algorithmic_bytes_per_launch=int(per_launch_bytes))
    roofline["unfused_equivalent"] = dict(
        note="the same ops priced at SURVEY 8d's per-update bytes (every op reads both children from HBM)",
        GBps=round(unfused_bytes / launches / (per_launch_ms * 1e-3) / 1e9, 1), bytes_per_launch=int(unfused_bytes / launches),
        frac_of_hbm_peak=round(unfused_bytes / launches / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    roofline.update(
                    full_traversal=dict(launches=launches_full, ms=round(ms_full / reps, 5),
                                        algorithmic_GBps=round((bytes_full or op_bytes(case, api, all_ops, entries)) / (ms_full / reps * 1e-3) / 1e9, 1),
                                        update_partials_only_M_per_s=round(sites * nops / (ms_full / reps * 1e-3) / 1e6, 1)))

    out = {
        "metric": "M site-CLV-updates/s", "value": round(value, 1), "unit": "M site-CLV-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": cfg["desc"] + (", PLL_ATTRIB_PATTERN_TIP" if args.pattern_tip else
                                              (", tips set with pll_set_tip_states (device reads 1-byte codes: tip x tip / tip x inner kernels at the leaves)"
                                               if codes else ", tips as dense 0/1 CLVs (every update inner x inner)")),
                   "sites_per_gpu": sites, "ops_per_traversal": nops, "states": cfg["states"], "rate_cats": 4,
                   "taxa": cfg["tips"], "step": "pll_update_partials(full traversal) + pll_compute_edge_loglikelihood"
                   + (" + all-reduce(lnL)" if world > 1 else ""), "parallelism": f"sites sharded x{world}"},
        "lnl": lnl, "roofline": roofline,
    }
    sess.close()
    if rank == 0 and world == 1 and not args.no_cpu:
        cb, ref_lnl = cpu_baseline(case, api, driver)
        out["cpu_baseline"] = cb
        if ref_lnl is not None and world == 1:
            out["lnl_rel_err"] = abs(lnl - ref_lnl) / abs(ref_lnl)
            out["lnl_reference"] = ref_lnl
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if dist_mode:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
