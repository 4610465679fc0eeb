#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X partial-likelihood hot path.

One "step" = one pass of the hot path over one synthetic alignment that is already resident in
HBM: pll_update_partials over the full post-order operation list followed by
pll_compute_edge_loglikelihood at the root edge, which synchronises and returns the log-likelihood.

    metric  M site-CLV-updates/s = sites * ops * steps / t / 1e6          (BASELINE.json)

    --gpus 1 (default)   BASELINE configs[1] "C2": 4-state DNA, 4 Gamma rates, 64-taxon balanced tree,
                         100k synthetic sites, 62 ops. Tips are set with pll_set_tip_states; the library
                         hands the device one byte per tip and site, so the 32 leaf-level updates are
                         tip x tip and the rest inner x inner.
    --gpus N > 1         BASELINE configs[3] "C4", STRONG scaling: ONE alignment - 4-state DNA, 128 taxa,
                         1M sites, PLL_ATTRIB_SITE_REPEATS, the same bytes on every rank - is pattern-sorted
                         (pll_compress_site_patterns, as applications do before they shard) and cut into N
                         contiguous site ranges; rank r owns an independent partition over range r and the
                         only exchange is ONE all-reduce of the shard log-likelihoods per step (RCCL on the
                         device). value = 1M sites * 126 ops * steps / max-over-ranks
                         time. Rank 0 afterwards times the UN-sharded alignment on its own GPU so that the
                         line carries t1_ms, tN_ms and speedup = t1 / tN.

Inputs are SURVEY.md section 8d to the letter (xorshift64 alignment, seed 88172645463325252, balanced
tree, branch lengths 0.05 + 0.01 (i mod 10), GTR / LG / 61-state stand-in, Gamma(0.5) x 4 mean rates).
The log-likelihood of every configuration is compared with the value the reference's AVX2 path gives
for the same inputs: tests/golden/section8d_lnl.json (made by tools/gen_section8d_lnl.py in the
authoring container) -> lnl_rel_err_pinned; at N = 1 the reference itself also runs in cpu_baseline
-> lnl_rel_err.

Besides the contract fields the JSON line carries
    roofline      HIP-event timing of the dominant CLV-update kernel's launches on the partition's
                  stream vs the 8 TB/s HBM3E peak; algorithmic bytes per DESIGN.md section 5
    cpu_baseline  the reference's own AVX2 path (oracle/_ref/libpll_ref.so, built from the reference
                  sources) timed on this host's cores on the same workload, rank 0 at N = 1 only
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
    "c2": dict(states=4, tips=64, sites=100000, desc="4-state DNA GTR+G4, 64-taxon balanced tree, 100k sites"),
    "c3": dict(states=20, tips=64, sites=50000, desc="20-state protein (LG)+G4, 64 taxa, 50k sites"),
    # the codon stand-in feeds its tips as one-hot CLVs through pll_set_tip_clv (every update
    # inner x inner); --tips states sets them with pll_set_tip_states instead (1-byte codes on the device)
    "c5": dict(states=61, tips=32, sites=20000, tips_as="clv", desc="61-state codon stand-in +G4, 32 taxa, 20k sites"),
    # not a BASELINE configuration: C3's shape with PLL_ATTRIB_SITE_REPEATS (what the any-state gather path costs)
    "c3r": dict(states=20, tips=64, sites=50000, repeats=True, desc="20-state protein (LG)+G4, 64 taxa, 50k sites, SITE_REPEATS"),
    # configs[3] on ONE GPU: the whole 1M-site alignment (the t1 leg of the strong-scaling run)
    "c4": dict(states=4, tips=128, sites=1000000, repeats=True, sort=True,
               desc="4-state DNA GTR+G4, 128 taxa, 1M sites, SITE_REPEATS"),
}
PINNED = os.path.join(ROOT, "tests", "golden", "section8d_lnl.json")


def build_case(cfg, sites, attributes, tree="balanced", tips_as=None):
    import numpy as np
    from pllamd import workload as W
    kw = {}
    if cfg["states"] == 20:
        lg = np.load(os.path.join(ROOT, "tests", "golden", "model_lg.npz"))
        kw.update(exch=lg["rates"], freqs=lg["freqs"])
    return W.make_case("bench", cfg["states"], cfg["tips"], sites, attributes=attributes, tree=tree,
                       generator="xorshift64", tips_as=tips_as or cfg.get("tips_as", "states"), **kw)


def pinned_lnl(key):
    if not os.path.exists(PINNED):
        return None
    return json.load(open(PINNED)).get(key)


def tips_are_codes(case, api):
    """tips set from sequences reach the device as one-byte codes (PATTERN_TIP, or the library's
    compact form of an indicator CLV) unless PLL_AMD_NO_TIP_CODES=1 forces dense tip CLVs"""
    if case.attributes & api.PATTERN_TIP:
        return True
    # (tip CLVs that are indicator vectors - what --tips clv feeds - are recognised as state masks by pll_set_tip_clv)
    return os.environ.get("PLL_AMD_NO_TIP_CODES", "0") in ("", "0") and not (case.attributes & api.SITE_REPEATS and case.sequences is None)


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


def sub_case(case, driver, lo, hi, name):
    import numpy as np
    kw = dict(name=name, states=case.states, rate_cats=case.rate_cats, tips=case.tips, sites=hi - lo,
              pmatrix=case.pmatrix, freqs=case.freqs, op_batches=case.op_batches, edges=case.edges,
              attributes=case.attributes, clv_buffers=case.clv_buffers, scale_buffers=case.scale_buffers,
              pattern_weights=np.asarray(case.pattern_weights)[lo:hi])
    if case.sequences is not None:
        kw.update(charmap=case.charmap, sequences=[sq[lo:hi] for sq in case.sequences])
    else:
        kw.update(tip_clvs=case.tip_clvs[:, lo:hi])
    return driver.Case(**kw)


def cpu_baseline(case, api, driver, budget_s=12.0):
    """reference AVX2 path on the host cores: T threads, each its own partition over sites/T
    contiguous sites (how applications parallelise libpll), traversal repeated to fill ~budget_s
    of CPU work. Falls back to the scalar restatement (kind 'port', 1 core, small sample)."""
    import numpy as np
    from oracle import oracle as O
    ops = len(case.op_batches[0])
    total_sites = int(np.asarray(case.pattern_weights, dtype=np.uint64).sum())
    if os.path.exists(O.REF_LIB):
        ref = api.PllLib(O.REF_LIB)
        cores = max(1, min(len(os.sched_getaffinity(0)), 16))
        bounds = np.linspace(0, case.sites, cores + 1).astype(int)
        shards = [driver.Session(ref, sub_case(case, driver, int(bounds[t]), int(bounds[t + 1]), f"shard{t}"), api.ARCH_AVX2)
                  for t in range(cores)]
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
        return dict(value=total_sites * ops * reps / dt / 1e6, unit="M site-CLV-updates/s", cores=cores, kind="reference",
                    sample=f"{reps} full traversals ({ops} ops) of the same {total_sites}-site alignment, reference "
                           f"PLL_ATTRIB_ARCH_AVX2 path, {cores} threads x sites/{cores} partitions, {dt:.1f} s wall"), float(sum(lnls))
    # no reference library on this host: time the scalar restatement on a small slice
    n = min(case.sites, 2000)
    sub = sub_case(case, driver, 0, n, "slice")
    t0 = time.perf_counter()
    O.run_case(sub)
    dt = time.perf_counter() - t0
    return dict(value=n * ops / dt / 1e6, unit="M site-CLV-updates/s", cores=1, kind="port",
                sample=f"one traversal of the first {n} sites with the scalar C restatement (oracle/pll_oracle.c)"), None


class Harness:
    """torch.distributed plumbing (only when WORLD_SIZE > 1 or PLL_BENCH_FORCE_DIST=1)"""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.same_device = os.environ.get("PLL_BENCH_SAME_DEVICE") == "1"
        os.environ.setdefault("PLL_AMD_DEVICE", "0" if self.same_device else str(self.local))
        self.dist = None
        self.torch = None
        self.backend = args.backend
        self.real_stdout = None
        if self.world > 1 or os.environ.get("PLL_BENCH_FORCE_DIST") == "1":
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            # RCCL prints a version banner on stdout when it initialises: keep stdout for the one JSON line
            sys.stdout.flush()
            self.real_stdout = os.dup(1)
            os.dup2(2, 1)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if self.backend == "nccl":
                torch.cuda.set_device(self.local)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local))
            else:
                dist.init_process_group("gloo")
        self.on_device = bool(self.dist) and self.backend == "nccl"
        self.tdev = f"cuda:{self.local}" if self.on_device else "cpu"
        self.tstream = None
        if self.on_device:
            self.tstream = self.torch.cuda.Stream()
            self.torch.cuda.set_stream(self.tstream)

    def tsync(self):
        if self.on_device:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.dist:
            self.tsync()
            self.dist.barrier()
            self.tsync()

    def max_over_ranks(self, v):
        if not self.dist:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_ints(self, vals):
        """[world][len(vals)] on every rank"""
        if not self.dist:
            return [list(vals)]
        t = self.torch.tensor(list(vals), dtype=self.torch.int64, device=self.tdev)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[int(x) for x in o.tolist()] for o in out]

    def finish(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()
        if self.real_stdout is not None:
            sys.stdout.flush()
            os.dup2(self.real_stdout, 1)


class Runner:
    """one partition + the step the benchmark times"""

    def __init__(self, h, lib, api, driver, case, repeats, collective=True):
        import numpy as np
        self.h, self.lib, self.api, self.case = h, lib, api, case
        self.sess = driver.Session(lib, case, api.ARCH_AVX2)  # uploads happen on first use (warm-up)
        self.edge = case.edges[0]
        self.repeats = repeats
        self.upd = 1  # site repeats: class maps are computed by the first step and re-used, as applications do
        #               between topology changes: pll_update_partials_rep(..., update_repeats = 0)
        self.collective = collective and bool(h.dist)
        self.device_path = self.collective and h.on_device
        if self.device_path:
            torch = h.torch
            # the partition works on torch's stream, the shard's lnL stays in HBM ({lnL, sequence} in `red`)
            # and RCCL reduces it there: no host round trip before the one exchange of the path
            if not lib.pll_gpu_set_stream(self.sess.p, h.tstream.cuda_stream):
                raise SystemExit(f"pll_gpu_set_stream: [{lib.errno()}] {lib.errmsg()}")
            self.red = torch.zeros(2, dtype=torch.float64, device=h.tdev)
            self.fi = np.ascontiguousarray(case.freqs_indices, dtype=np.uint32)
            # the reduced {lnL, sequence} pair comes back through pinned host memory that the host polls,
            # like the single-GPU path does (a stream synchronise costs more than the copy)
            self.pinned = torch.zeros(2, dtype=torch.float64).pin_memory()
            self.pview = self.pinned.numpy()
            self.expected = None
        elif self.collective:
            self.red = h.torch.zeros(1, dtype=h.torch.float64)

    def step(self):
        lib, sess, e, h = self.lib, self.sess, self.edge, self.h
        sess.update_partials(update_repeats=self.upd)
        if self.repeats:
            self.upd = 0
        if self.device_path:
            if not lib.pll_gpu_edge_loglikelihood_async(sess.p, e[0], e[1], e[2], e[3], e[4], self.api.uptr(self.fi),
                                                        self.red.data_ptr()):
                raise SystemExit(f"pll_gpu_edge_loglikelihood_async: [{lib.errno()}] {lib.errmsg()}")
            # the path's one exchange: sum of the shards' log-likelihoods (word 0); word 1 = every rank's
            # call sequence number, so its sum tells the host which evaluation the pair belongs to
            h.dist.all_reduce(self.red)
            # one 16-byte copy of {value, sequence} (two stream-ordered 8-byte copies cost ~5 us more per step); the host
            # reads the value only after it has seen the sequence word twice
            self.pinned.copy_(self.red, non_blocking=True)
            if self.expected is not None:
                self.expected += h.world
                t_spin = time.perf_counter()
                seen = 0
                while seen < 2:
                    seen = seen + 1 if self.pview[1] == self.expected else 0
                    if time.perf_counter() - t_spin > 0.02:
                        self.expected = None
                        break
            if self.expected is None:
                h.torch.cuda.current_stream().synchronize()
                self.expected = float(self.pview[1])
            return float(self.pview[0])
        v, _ = sess.edge_lnl(e, persite=False)
        if self.collective:
            self.red[0] = v
            h.dist.all_reduce(self.red)
            v = float(self.red[0].item())
        return v

    def fence(self):
        self.lib.pll_gpu_synchronize(self.sess.p)
        if self.collective:
            self.h.barrier()

    def timed(self, warmup, steps):
        """`warmup` untimed steps, then exactly `steps` steps between two fences; max over ranks"""
        lnl = None
        for _ in range(warmup):
            lnl = self.step()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            lnl = self.step()
        self.fence()
        dt = time.perf_counter() - t0
        if self.collective:
            dt = self.h.max_over_ranks(dt)
        return dt, lnl

    def repeats_update_ms(self, reps=5):
        """what the class maps cost when the topology changed: a full pll_update_partials_rep(.., 1) minus the
        same call with update_repeats = 0, both to completion (the cost the timed steps keep outside)"""
        lib, sess = self.lib, self.sess
        out = []
        for ur in (0, 1):
            sess.update_partials(update_repeats=ur)
            lib.pll_gpu_synchronize(sess.p)
            t0 = time.perf_counter()
            for _ in range(reps):
                sess.update_partials(update_repeats=ur)
            lib.pll_gpu_synchronize(sess.p)
            out.append((time.perf_counter() - t0) / reps * 1e3)
        return max(out[1] - out[0], 0.0), out[1]

    def level_entries(self):
        """site repeats: entries actually computed per tree level [(ops, total entries)]"""
        ops = self.case.op_batches[0]
        depth = {t: 0 for t in range(self.case.tips)}
        per = {}
        for op in ops:
            d = depth[op[0]] = 1 + max(depth[op[2]], depth[op[5]])
            per.setdefault(d, []).append(self.sess.entries(op[0]))
        return [sum(per[d]) for d in sorted(per)]

    def close(self):
        self.sess.close()


def roofline_leg(args, cfg, lib, api, runner, reps=20):
    """HIP-event timing of the dominant kernel's launches"""
    case, sess = runner.case, runner.sess
    sites = case.sites
    nops = len(case.op_batches[0])
    upd = runner.upd
    # ---- full traversal alone (no lnL), HIP events on the partition's stream
    lib.pll_gpu_synchronize(sess.p)
    lib.pll_gpu_timer_start(sess.p)
    for _ in range(reps):
        sess.update_partials(update_repeats=upd)
    ms_full = lib.pll_gpu_timer_stop(sess.p)
    launches_full = lib.pll_gpu_last_launch_count(sess.p)
    bytes_full = lib.pll_gpu_last_algorithmic_bytes(sess.p)
    # ---- the DOMINANT kernel. Its launches are timed by re-running the part of the traversal it evaluates
    # (a valid partial traversal: whatever it reads is already in HBM)
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
    # 17..32 states with tips as codes: the bottom two levels are ONE launch of (tip x tip, tip x tip -> inner x inner)
    # groups on the matrix pipe (k_partials_mfma_cc) - half of the step; the leg re-runs exactly those ops
    grouped = (17 <= cfg["states"] <= 32 and codes and not cfg.get("repeats") and not os.environ.get("PLL_AMD_NO_FUSE", "0").strip("0")
               and os.environ.get("PLL_AMD_FUSE_GENERIC", "2") not in ("0", "1") and not os.environ.get("PLL_AMD_MFMA_MIN_STATES"))
    if grouped:
        depth = {t: 0 for t in range(case.tips)}
        for op in all_ops:
            depth[op[0]] = 1 + max(depth[op[2]], depth[op[5]])
        low = [op for op in all_ops if depth[op[0]] <= 2]
        if len(low) == 3 * (case.tips // 4):
            ii_ops = low
        else:
            grouped = False
    if args.tree != "balanced":  # irregular levels: no single dominant launch shape - the leg is the whole traversal
        ii_ops, cc, grouped = list(all_ops), False, False
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
    if os.path.exists(tfile) and not args.pattern_tip and not args.sites and not args.taxa and args.tree == "balanced" and not args.tips:
        traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")  # PMC-derived HBM bytes per launch of the same command (profiles/README.md)
    mfma = cfg["states"] > 32 and not os.environ.get("PLL_AMD_NO_MFMA", "0").strip("0")
    gg = cfg["states"] == 4 and cfg.get("repeats") and not os.environ.get("PLL_AMD_NO_FUSE_GG", "0").strip("0")
    kernel = {4: "k_partials_dna_cc<5,5>%.0s" if cc else "k_partials_dna_fused<4,4>%.0s" if fused else
                 "k_partials_dna<false,false,true>%.0s (compressed levels) + k_partials_dna_gg (where compression ends)" if gg else "k_partials_dna<false,false,%s>",
              20: "k_partials_mfma_cc<5>%.0s" if grouped else
                  ("k_partials_lean<5,false,false,true>%.0s (gathering launches) + k_partials_tiled<20,false,false,false>"
                   if cfg.get("repeats") and not os.environ.get("PLL_AMD_NO_LEAN", "0").strip("0") else "k_partials_tiled<20,false,false,%s>"),
              61: "k_partials_mfma<false,false,%s>" if mfma else "k_partials_tiled<32,false,false,%s>"
              }[cfg["states"]] % ("true" if cfg.get("repeats") else "false")
    if args.tree != "balanced":
        kernel = "all update launches of the traversal"
    if mfma:
        # 33..64 states sit past the fp64 ridge: the bounding line is the fp64 matrix pipe.
        # Algorithmic flop per update = the reference's own arithmetic: 2 children x S x S multiply-adds
        # + S products per (site, rate)  (core_partials.c:739-757) - not the 64-padded MFMA tiles
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
    roofline.update(full_traversal=dict(
        launches=launches_full, ms=round(ms_full / reps, 5),
        algorithmic_GBps=round((bytes_full or op_bytes(case, api, all_ops, entries)) / (ms_full / reps * 1e-3) / 1e9, 1),
        update_partials_only_M_per_s=round(sites * nops / (ms_full / reps * 1e-3) / 1e6, 1)))
    return roofline, codes


def tip_note(case, api, args):
    if args.pattern_tip:
        return ", PLL_ATTRIB_PATTERN_TIP"
    if case.sequences is None and tips_are_codes(case, api):
        return ", tips as one-hot CLVs through pll_set_tip_clv (recognised as indicator vectors: the device reads 1-byte codes, tip x tip / tip x inner kernels at the leaves; PLL_AMD_NO_TIP_CODES=1 keeps them dense)"
    if case.sequences is None:
        return ", tips as one-hot CLVs through pll_set_tip_clv, kept dense (every update inner x inner)"
    if tips_are_codes(case, api):
        return ", tips set with pll_set_tip_states (device reads 1-byte codes: tip x tip / tip x inner kernels at the leaves)"
    return ", tips as dense 0/1 CLVs (every update inner x inner)"


def main_single(args, h):
    """N = 1: one configuration on one GPU (default C2)"""
    import numpy as np
    from pllamd import api, driver, sharding

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
    lib = api.PllLib()
    case = build_case(cfg, sites, attributes, tree=args.tree, tips_as=args.tips)
    total_sites = sites
    if cfg.get("sort"):  # the 1M-site alignment as applications hand it over: pattern-sorted, weights attached
        case = sharding.sort_columns(lib, case)
    nops = len(case.op_batches[0])
    runner = Runner(h, lib, api, driver, case, cfg.get("repeats"), collective=bool(h.dist))
    dt, lnl = runner.timed(args.warmup, args.steps)
    if not np.isfinite(lnl):
        raise SystemExit(f"hot path failed: lnL = {lnl} [{lib.errno()}] {lib.errmsg()}")
    value = total_sites * nops * args.steps / dt / 1e6
    roofline, codes = roofline_leg(args, cfg, lib, api, runner)
    out = {
        "metric": "M site-CLV-updates/s", "value": round(value, 1), "unit": "M site-CLV-updates/s",
        "n_gpus": h.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (SURVEY 8d: xorshift64 alignment, seed 88172645463325252)",
        "config": {"workload": cfg["desc"] + tip_note(case, api, args),
                   "sites_per_gpu": total_sites, "ops_per_traversal": nops, "states": cfg["states"], "rate_cats": 4,
                   "taxa": cfg["tips"], "step": "pll_update_partials(full traversal) + pll_compute_edge_loglikelihood",
                   "parallelism": "1 GPU"},
        "lnl": lnl, "roofline": roofline,
    }
    if cfg.get("repeats"):
        extra, full = runner.repeats_update_ms()
        out["repeats_update_ms"] = round(extra, 4)
        out["traversal_with_class_maps_ms"] = round(full, 4)
        out["entries_per_level"] = runner.level_entries()
        out["patterns"] = case.sites
    default_shape = not (args.sites or args.taxa or args.pattern_tip or args.tree != "balanced" or args.tips)
    pin = pinned_lnl(args.config) if default_shape else None
    if pin is not None:
        out["lnl_pinned_reference"] = pin
        out["lnl_rel_err_pinned"] = abs(lnl - pin) / abs(pin)
    runner.close()
    if h.rank == 0 and not args.no_cpu:
        cb, ref_lnl = cpu_baseline(case, api, driver)
        out["cpu_baseline"] = cb
        if ref_lnl is not None:
            out["lnl_rel_err"] = abs(lnl - ref_lnl) / abs(ref_lnl)
            out["lnl_reference"] = ref_lnl
    return out


def main_strong(args, h):
    """N > 1: configs[3], ONE 1M-site alignment sharded over the ranks (strong scaling)"""
    import numpy as np
    from pllamd import api, driver, sharding

    cfg = dict(CONFIGS["c4"])
    total_sites = args.sites or cfg["sites"]  # --sites: the WHOLE alignment here (rehearsals)
    lib = api.PllLib()
    # the same bytes on every rank: the generator is deterministic, nothing is broadcast
    full = build_case(cfg, total_sites, api.SITE_REPEATS)
    full = sharding.sort_columns(lib, full)  # unique columns in lexicographic order + weights (device radix sort)
    nops = len(full.op_batches[0])
    mine = sharding.shard_case(full, h.rank, h.world)
    runner = Runner(h, lib, api, driver, mine, True)
    dt, lnl = runner.timed(args.warmup, args.steps)
    if not np.isfinite(lnl):
        raise SystemExit(f"hot path failed: lnL = {lnl} [{lib.errno()}] {lib.errmsg()}")
    value = total_sites * nops * args.steps / dt / 1e6
    tN_ms = dt / args.steps * 1e3
    rep_extra, rep_full = runner.repeats_update_ms()
    rep_extra = h.max_over_ranks(rep_extra)
    shard_levels = h.gather_ints(runner.level_entries())
    shard_sites = h.gather_ints([mine.sites])
    # roofline of the dominant kernel on this rank's shard (rank 0 reports)
    args.config = "c4"
    roofline, _ = roofline_leg(args, cfg, lib, api, runner)
    runner.close()
    h.barrier()
    out = None
    if h.rank == 0:
        # t1: the un-sharded alignment on ONE GPU, same step, same K/W, nothing else running on the node
        solo = Runner(h, lib, api, driver, full, True, collective=False)
        dt1, lnl1 = solo.timed(args.warmup, args.steps)
        t1_ms = dt1 / args.steps * 1e3
        global_levels = solo.level_entries()
        rep1_extra, _ = solo.repeats_update_ms(reps=3)
        solo.close()
        pin = pinned_lnl("c4") if not args.sites else None
        out = {
            "metric": "M site-CLV-updates/s", "value": round(value, 1), "unit": "M site-CLV-updates/s",
            "n_gpus": h.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(tN_ms, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (SURVEY 8d: xorshift64 alignment, seed 88172645463325252; the same alignment on every rank)",
            "config": {"workload": cfg["desc"] + f", pattern-sorted (pll_compress_site_patterns) and cut into {h.world} contiguous site ranges"
                       + tip_note(full, api, args),
                       "total_sites": total_sites, "patterns": full.sites, "sites_per_gpu": [s[0] for s in shard_sites],
                       "ops_per_traversal": nops, "states": 4, "rate_cats": 4, "taxa": cfg["tips"],
                       "step": "pll_update_partials(full traversal) + edge log-likelihood on the device + all-reduce(lnL)",
                       "parallelism": f"sites sharded x{h.world}, one all-reduce of one double per step ({h.backend})"},
            "lnl": lnl, "t1_ms": round(t1_ms, 4), "tN_ms": round(tN_ms, 4), "speedup": round(t1_ms / tN_ms, 3),
            "t1_value": round(total_sites * nops / (t1_ms * 1e-3) / 1e6, 1), "lnl_unsharded": lnl1,
            "lnl_rel_err_vs_unsharded": abs(lnl - lnl1) / abs(lnl1),
            "repeats_update_ms": round(rep_extra, 4), "repeats_update_ms_unsharded": round(rep1_extra, 4),
            "entries_per_level": {"unsharded": global_levels, "shards": shard_levels,
                                  "sum_over_shards_div_unsharded": round(sum(sum(s) for s in shard_levels) / max(sum(global_levels), 1), 4)},
            "roofline": roofline,
        }
        if pin is not None:
            out["lnl_pinned_reference"] = pin
            out["lnl_rel_err_pinned"] = abs(lnl - pin) / abs(pin)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--sites", type=int, default=0, help="override the number of sites (N > 1: of the whole alignment)")
    ap.add_argument("--pattern-tip", action="store_true", help="PLL_ATTRIB_PATTERN_TIP variant")
    ap.add_argument("--tips", default=None, choices=["states", "clv"],
                    help="how tips are set: pll_set_tip_states or one-hot CLVs through pll_set_tip_clv (default: per config, SURVEY 8d)")
    ap.add_argument("--tree", default="balanced", choices=["balanced", "random", "caterpillar"],
                    help="topology (BASELINE's configs are balanced; the others show what irregular level structures cost)")
    ap.add_argument("--taxa", type=int, default=0, help="override the number of taxa")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + PLL_BENCH_SAME_DEVICE=1 rehearses the N>1 flow on a one-GPU box")
    args = ap.parse_args()
    h = Harness(args)
    assert h.world == args.gpus or h.world == 1, f"--gpus {args.gpus} but WORLD_SIZE={h.world}"
    out = main_strong(args, h) if h.world > 1 else main_single(args, h)
    h.finish()
    if h.rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
