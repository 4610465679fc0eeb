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
                         contiguous site ranges of equal site count (--cut balanced: of equal modelled cost,
                         pllamd/sharding.py - measured slower: the larger shards fall out of the Infinity
                         Cache); rank r owns an independent partition over range r and the
                         only exchange is ONE sum of the shard log-likelihoods per step (SURVEY.md section 8e;
                         the sum it stands for: src/core_likelihood.c:1489):
                           --reduce peer (default)  fixed rank order through shared host memory
                                                    (pll_gpu_group_edge_loglikelihood, csrc/host/group.c):
                                                    reproducible bits, ~0.6 us at 8 ranks
                           --reduce rccl            one RCCL all-reduce of the value where the kernel left it
                         With peer and the nccl backend the same steps are timed again with rccl and both
                         appear in the line (exchange.*). value = 1M sites * 126 ops * steps / max-over-ranks
                         time. Rank 0 afterwards times the UN-sharded alignment on its own GPU so that the
                         line carries t1_ms, tN_ms and speedup = t1 / tN.
                         Started WITHOUT a launcher (WORLD_SIZE unset) the process starts its own N ranks
                         (one child per GPU, 127.0.0.1 rendezvous) and relays rank 0's line; under
                         torch.distributed.run it is one of the ranks. --gpus N with another WORLD_SIZE is
                         an error.

The timed region is `blocks` (default 5) blocks of exactly `steps` steps, each bracketed by a barrier and a
device synchronise on both sides, max over ranks per block; ms_per_step / value come from the MEDIAN block,
ms_per_step_min / _max give the spread. The steps themselves are issued by a C loop over the library's two
C-ABI calls (csrc/workload/step_loop.c, --driver python for the ctypes loop): an application is C / C++.

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

RCCL_LEG_HUNG_RC = 3   # exit status of every rank when the RCCL leg's watchdog fired
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

# which calls a timed step is made of (config.step). Under PLL_ATTRIB_SITE_REPEATS `value` is timed with the class maps
# of the first step re-used - what applications do between topology changes; the step as the reference's
# pll_update_partials defines it (update_repeats = 1: class maps recomputed, src/partials.c:237-255) is timed beside it
# and carried as ms_per_step_with_class_maps / value_with_class_maps.
STEP_PLAIN = "pll_update_partials(full traversal) + pll_compute_edge_loglikelihood"
STEP_REUSED = ("pll_update_partials_rep(full traversal, update_repeats = 0: the class maps of the first step re-used) "
               "+ pll_compute_edge_loglikelihood")
STEP_MAPS = ("pll_gpu_invalidate(PLL_GPU_FORGET_REPEATS) + pll_update_partials(full traversal): EVERY class map of the traversal "
             "computed again by every step - what the reference does on every call (src/partials.c:237-255) and what a step after "
             "a change of every tip costs - + pll_compute_edge_loglikelihood")
STEP_DEFAULT = ("pll_update_partials(full traversal) = pll_update_partials_rep(.., update_repeats = 1), the reference's default call, on "
                "an UNCHANGED tree: the library finds every class map's inputs as they were (version stamps, csrc/host/repeats.c) and "
                "computes none + pll_compute_edge_loglikelihood")


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


def cpu_baseline(case, api, driver, budget_s=4.0):
    """reference AVX2 path on the host cores (SURVEY 8d "CPU baseline plan"): T threads, each its own partition
    over sites/T contiguous sites (how applications parallelise libpll). Three legs, each a bounded sample:
      value        T = all host cores (<= 16), PLL_ATTRIB_ARCH_AVX2          median of 3 samples
      pattern_tip  the same with | PLL_ATTRIB_PATTERN_TIP (the reference's faster tip kernels,
                   src/core_partials_avx.c:1310-1506; only where tips are sequences)   median of 3
      one_core     T = 1, plain AVX2                                          one sample
    Falls back to the scalar restatement (kind 'port', 1 core, small sample) without the reference library."""
    import numpy as np
    from oracle import oracle as O
    ops = len(case.op_batches[0])
    total_sites = int(np.asarray(case.pattern_weights, dtype=np.uint64).sum())
    if not os.path.exists(O.REF_LIB):
        # no reference library on this host: time the scalar restatement on a small slice
        n = min(case.sites, 2000)
        sub = sub_case(case, driver, 0, n, "slice")
        t0 = time.perf_counter()
        O.run_case(sub)
        dt = time.perf_counter() - t0
        return dict(value=n * ops / dt / 1e6, unit="M site-CLV-updates/s", cores=1, kind="port",
                    sample=f"one traversal of the first {n} sites with the scalar C restatement (oracle/pll_oracle.c)"), None
    ref = api.PllLib(O.REF_LIB)
    rep_mode = bool(case.attributes & api.SITE_REPEATS)

    def leg(cores, extra_attr, samples, budget, maps_every_step=False):
        c2 = case
        if extra_attr:
            kw = {f: getattr(case, f) for f in case.__dataclass_fields__}
            kw["attributes"] = case.attributes | extra_attr
            c2 = driver.Case(**kw)
        bounds = np.linspace(0, c2.sites, cores + 1).astype(int)
        shards = [driver.Session(ref, sub_case(c2, driver, int(bounds[t]), int(bounds[t + 1]), f"shard{t}"), api.ARCH_AVX2)
                  for t in range(cores)]
        # one traversal to size the sample, then reps traversals timed
        t0 = time.perf_counter()
        shards[0].update_partials()
        one = (time.perf_counter() - t0)  # wall seconds per traversal of one thread's share
        reps = int(max(2, min(200, budget / max(one, 1e-4))))
        lnls = [0.0] * cores
        if rep_mode:  # class maps once, outside the timed region (same policy as the GPU leg)
            for sh in shards:
                sh.update_partials(update_repeats=1)

        def work(i):
            for _ in range(reps):
                shards[i].update_partials(update_repeats=0 if rep_mode and not maps_every_step else 1)
            lnls[i] = shards[i].edge_lnl(c2.edges[0], persite=False)[0]

        vals, walls = [], []
        for _ in range(samples):
            th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            dt = time.perf_counter() - t0
            walls.append(dt)
            vals.append(total_sites * ops * reps / dt / 1e6)
        for sh in shards:
            sh.close()
        return vals, reps, sum(walls), float(sum(lnls))

    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    vals, reps, wall, lnl = leg(cores, 0, 3, budget_s)
    out = dict(value=round(float(np.median(vals)), 2), unit="M site-CLV-updates/s", cores=cores, kind="reference",
               samples=[round(v, 2) for v in vals],
               sample=f"median of 3 samples of {reps} full traversals ({ops} ops) of the same {total_sites}-site alignment, reference "
                      f"PLL_ATTRIB_ARCH_AVX2 path, {cores} threads x sites/{cores} partitions, {wall:.1f} s wall in all")
    if case.sequences is not None and not rep_mode:
        v2, r2, w2, _ = leg(cores, api.PATTERN_TIP, 3, budget_s * 0.75)
        out["pattern_tip"] = dict(value=round(float(np.median(v2)), 2), cores=cores, samples=[round(v, 2) for v in v2],
                                  sample=f"the same with | PLL_ATTRIB_PATTERN_TIP, {r2} traversals per sample, {w2:.1f} s wall")
    if rep_mode:
        # the same step as the reference's pll_update_partials defines it: class maps recomputed by every traversal
        v3, r3, w3, _ = leg(cores, 0, 3, budget_s * 0.75, maps_every_step=True)
        out["with_class_maps"] = dict(value=round(float(np.median(v3)), 2), cores=cores, samples=[round(v, 2) for v in v3],
                                      sample=f"the same with update_repeats = 1 on every traversal (pll_update_partials), {r3} traversals per sample, {w3:.1f} s wall")
    v1, r1, w1, _ = leg(1, 0, 1, budget_s)
    out["one_core"] = dict(value=round(v1[0], 3), cores=1, sample=f"{r1} traversals of the whole alignment on one thread, {w1:.1f} s wall")
    return out, lnl


def gpu_numa_node(index):
    """NUMA node of the index-th AMD GPU as sysfs lists them (render nodes in PCI order = HIP's order with no
    *_VISIBLE_DEVICES set), or None. No GPU call."""
    import glob
    cards = []
    for dev in glob.glob("/sys/class/drm/renderD*/device"):
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
            cards.append((os.path.basename(os.path.realpath(dev)), int(open(os.path.join(dev, "numa_node")).read())))
        except (OSError, ValueError):
            continue
    cards.sort()
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if vis:
        try:
            index = [int(v) for v in vis.split(",")][index]
        except (ValueError, IndexError):
            return None
    return cards[index][1] if index < len(cards) and cards[index][1] >= 0 else None


def pin_near_gpu(index):
    """sched_setaffinity of this process to the cores of its GPU's NUMA node (intersected with the cores it may
    use). Returns what the line reports: {"numa_node", "cpus"} or a reason."""
    node = gpu_numa_node(index)
    if node is None:
        return {"pinned": False, "why": "no NUMA node for this GPU in sysfs"}
    try:
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"pinned": False, "numa_node": node, "why": "none of the node's cores is available to this process"}
        os.sched_setaffinity(0, cpus)
        return {"pinned": True, "numa_node": node, "cpus": len(cpus)}
    except (OSError, ValueError) as exc:
        return {"pinned": False, "numa_node": node, "why": f"{type(exc).__name__}: {exc}"}


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(args, command=None):
    """--gpus N > 1 without a launcher: this process becomes the launcher. It starts N children of this very
    command line (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous), relays rank 0's stdout
    - the one JSON line - and returns the worst exit code. It never touches the GPU (nor imports torch).
    (`command`: what a rank runs instead of this file - tests/test_bench_launcher.py.)"""
    import subprocess
    command = command or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(command, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's stdout is drained while it runs: a child that writes more than a pipe buffer (a long line, library
    # diagnostics on stdout) would otherwise block in write() and the launcher sit out its whole time limit
    import threading
    chunks = []

    def drain():
        for piece in iter(lambda: procs[0].stdout.read(65536), b""):
            chunks.append(piece)

    reader = threading.Thread(target=drain, daemon=True)
    reader.start()

    def relay():
        reader.join(timeout=10)
        text = b"".join(chunks).decode(errors="replace")
        if text:
            sys.stdout.write(text)
            sys.stdout.flush()

    worst = 0
    pending = set(range(args.gpus))
    deadline = time.monotonic() + float(os.environ.get("PLL_BENCH_TIMEOUT_S", "1500"))
    while pending:
        if time.monotonic() > deadline:  # ranks that wait for each other forever must not keep the launcher forever
            print(f"bench.py: ranks {sorted(pending)} still running at the launcher's time limit; stopping them", file=sys.stderr)
            for q in pending:  # exactly the children started here, by pid
                procs[q].kill()
            for q in pending:  # ... reaped, so that none is left a zombie and rank 0's pipe reaches its end
                try:
                    procs[q].wait(timeout=30)
                except subprocess.TimeoutExpired:
                    print(f"bench.py: rank {q} (pid {procs[q].pid}) did not die within 30 s of SIGKILL", file=sys.stderr)
            relay()  # whatever rank 0 had printed is evidence, not something to drop
            return worst or 124
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                worst = worst or rc
                print(f"bench.py: rank {r} exited with {rc}; stopping the others", file=sys.stderr)
                for q in pending:  # exactly the children started here, by pid
                    procs[q].terminate()
        time.sleep(0.05)
    relay()
    return worst


class Harness:
    """torch.distributed plumbing (only when WORLD_SIZE > 1 or PLL_BENCH_FORCE_DIST=1): rendezvous, the barrier and
    the max over ranks around the timed blocks. The path's own exchange is csrc/host/group.c (--reduce peer) or one
    all-reduce on the device (--reduce rccl)."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.same_device = os.environ.get("PLL_BENCH_SAME_DEVICE") == "1"
        os.environ.setdefault("PLL_AMD_DEVICE", "0" if self.same_device else str(self.local))
        # every step ends in a host poll of mapped memory and, sharded, a shared-memory exchange: keep this rank's CPU
        # next to its GPU (before anything touches the GPU or imports torch)
        self.cpu_affinity = pin_near_gpu(0 if self.same_device else self.local) if self.world > 1 or os.environ.get("PLL_BENCH_PIN") == "1" else None
        self.dist = None
        self.torch = None
        self.backend = args.backend
        self.real_stdout = None
        self.group = None
        self.group_lib = None
        self.rccl = None  # (ncclComm_t, librccl) of make_rccl_comm, or (None, reason)
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

    def join_group(self, lib):
        """the shared-memory segment of --reduce peer: rank 0 picks a name that is unique to this run"""
        if not self.dist or self.group:
            return self.group
        import uuid
        name = ["/pllamd-%s" % uuid.uuid4().hex[:16] if self.rank == 0 else None]
        self.dist.broadcast_object_list(name, src=0)
        g = lib.pll_gpu_group_join(name[0].encode(), self.rank, self.world, 120000)
        if not g:
            raise SystemExit(f"pll_gpu_group_join: [{lib.errno()}] {lib.errmsg()}")
        self.group, self.group_lib = g, lib
        return g

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

    def gather_objects(self, obj):
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def finish(self):
        if self.group:
            self.group_lib.pll_gpu_group_leave(self.group)
            self.group = None
        if self.rccl and self.rccl[0] is not None:
            self.tsync()
            self.rccl[1].ncclCommDestroy(self.rccl[0])
            self.rccl = None
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()
        if self.real_stdout is not None:
            sys.stdout.flush()
            os.dup2(self.real_stdout, 1)


_STEP_LOOP = None


def step_loop_fn():
    """csrc/workload/step_loop.c: the caller side of the timed region in C (the library's entry points come in as
    function pointers)"""
    global _STEP_LOOP
    if _STEP_LOOP is None:
        import ctypes as C
        dll = C.CDLL(os.path.join(ROOT, "libpll-2_amd", "csrc", "libpll_workload.so"))
        fn = dll.pllwl_step_loop
        fn.restype = C.c_double
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint,
                       C.POINTER(C.c_int), C.POINTER(C.c_uint), C.c_uint, C.POINTER(C.c_double), C.c_void_p]
        _STEP_LOOP = fn
    return _STEP_LOOP


def make_rccl_comm(h):
    """A communicator of this run's ranks for the LIBRARY's all-reduce entry point (pll_gpu_edge_loglikelihood_allreduce
    takes a caller-made ncclComm_t; torch does not hand its own out): rank 0 draws the id, the control plane carries it,
    every rank calls ncclCommInitRank on its device. Every step that one rank can fail alone is followed by an agreement
    over the control plane, so that no rank walks into a collective the others have given up on.
    Returns (comm, dll) or (None, reason) - the same kind on every rank."""
    import ctypes as C

    def all_ok(mine):
        t = h.torch.tensor([1.0 if mine else 0.0], dtype=h.torch.float64, device=h.tdev)
        h.dist.all_reduce(t, op=h.dist.ReduceOp.MIN)
        return float(t.item()) == 1.0

    dll, why = None, ""
    try:
        dll = C.CDLL(os.environ.get("PLL_AMD_RCCL_LIB") or "librccl.so.1", mode=C.RTLD_GLOBAL)
    except OSError as exc:
        why = f"librccl not loadable: {exc}"
    if not all_ok(dll is not None):
        return None, why or "librccl not loadable on another rank"

    class UniqueId(C.Structure):  # ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
        _fields_ = [("internal", C.c_ubyte * 128)]

    uid = UniqueId()
    drawn = h.rank != 0 or dll.ncclGetUniqueId(C.byref(uid)) == 0
    blob = [bytes(uid.internal) if (h.rank == 0 and drawn) else None]
    h.dist.broadcast_object_list(blob, src=0)
    if blob[0] is None:
        return None, "ncclGetUniqueId failed on rank 0"
    C.memmove(C.byref(uid), blob[0], 128)
    dll.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    dll.ncclCommInitRank.restype = C.c_int
    comm = C.c_void_p()
    rc = dll.ncclCommInitRank(C.byref(comm), h.world, uid, h.rank)
    if not all_ok(rc == 0 and bool(comm.value)):
        return None, f"ncclCommInitRank failed on some rank (here: rc {rc})"
    return comm, dll


_ALLREDUCE_LOOP = None


def _allreduce_loop_fn():
    global _ALLREDUCE_LOOP
    if _ALLREDUCE_LOOP is None:
        import ctypes as C
        dll = C.CDLL(os.path.join(ROOT, "libpll-2_amd", "csrc", "libpll_workload.so"))
        fn = dll.pllwl_step_loop_allreduce
        fn.restype = C.c_double
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint,
                       C.POINTER(C.c_int), C.POINTER(C.c_uint), C.c_uint, C.POINTER(C.c_double), C.c_void_p]
        _ALLREDUCE_LOOP = fn
    return _ALLREDUCE_LOOP


class Runner:
    """one partition + the step the benchmark times. reduce: None (one partition is the whole job), 'peer' (fixed
    rank order through shared host memory) or 'rccl' (all-reduce on the device)."""

    def __init__(self, h, lib, api, driver, case, repeats, reduce=None, c_driver=True):
        import ctypes as C
        import numpy as np
        self.h, self.lib, self.api, self.case = h, lib, api, case
        self.sess = driver.Session(lib, case, api.ARCH_AVX2)  # uploads happen on first use (warm-up)
        self.edge = case.edges[0]
        self.repeats = repeats
        self.upd = 1  # site repeats: class maps are computed by the first step and re-used, as applications do
        #               between topology changes: pll_update_partials_rep(..., update_repeats = 0)
        # timed_with_class_maps: None = as above; "default" = update_repeats = 1 on EVERY step (the reference's
        # pll_update_partials) on the unchanged tree; "forced" = the same with PLL_GPU_FORGET_REPEATS ahead of every step
        self.maps_mode = None
        self.reduce = reduce if h.dist else None
        self.collective = self.reduce is not None  # this runner's steps are steps of ALL ranks (barriers, max over ranks)
        self.c_driver = c_driver
        self.fi = np.ascontiguousarray(case.freqs_indices, dtype=np.uint32)
        self.edge_c = (C.c_int * 5)(*[int(v) for v in self.edge])
        self.device_path = False
        self.rccl_comm, self.rccl_via = None, None
        if self.reduce == "peer":
            self.group = h.join_group(lib)
        elif self.reduce == "rccl":
            self.setup_rccl()

    def setup_rccl(self):
        h, lib = self.h, self.lib
        self.device_path = h.on_device
        self.rccl_comm, self.rccl_via = None, "torch.distributed.all_reduce on a host value"
        if self.device_path and os.environ.get("PLL_BENCH_TORCH_ALLREDUCE") != "1":
            # the product's own exchange: the library's C entry point on a communicator of this run's ranks
            if h.rccl is None:
                h.rccl = make_rccl_comm(h)
            comm, what = h.rccl
            if comm is not None:
                ok = h.torch.tensor([1.0 if lib.pll_gpu_allreduce_prepare(self.sess.p, comm) else 0.0], dtype=h.torch.float64, device=h.tdev)
                h.dist.all_reduce(ok, op=h.dist.ReduceOp.MIN)  # set-up failures are agreed on BEFORE the first collective
                if float(ok.item()) == 1.0:
                    self.rccl_comm = comm
                    self.rccl_via = "pll_gpu_edge_loglikelihood_allreduce (the library's C entry point, ncclAllReduce on the partition's stream)"
                    return
                what = f"pll_gpu_allreduce_prepare: [{lib.errno()}] {lib.errmsg()}"
            print(f"bench.py: rank {h.rank}: no communicator for the library's all-reduce ({what}); torch.distributed.all_reduce instead", file=sys.stderr)
        if self.device_path:
            self.rccl_via = "pll_gpu_edge_loglikelihood_async + torch.distributed.all_reduce on the device value"
            torch = h.torch
            # the partition works on torch's stream, the shard's lnL stays in HBM ({lnL, sequence} in `red`)
            # and RCCL reduces it there: no host round trip before the one exchange of the path
            if not lib.pll_gpu_set_stream(self.sess.p, h.tstream.cuda_stream):
                raise SystemExit(f"pll_gpu_set_stream: [{lib.errno()}] {lib.errmsg()}")
            self.red = torch.zeros(2, dtype=torch.float64, device=h.tdev)
            # the reduced {lnL, sequence} pair comes back through pinned host memory that the host polls: two
            # stream-ordered 8-byte copies, the value first - when the sequence word has arrived the value is there
            self.pinned = torch.zeros(2, dtype=torch.float64).pin_memory()
            self.pview = self.pinned.numpy()
            self.expected = None
        else:
            self.red = h.torch.zeros(1, dtype=h.torch.float64)

    def step(self):
        lib, sess, e, h = self.lib, self.sess, self.edge, self.h
        if self.maps_mode == "forced":
            lib.pll_gpu_invalidate(sess.p, self.api.FORGET_REPEATS, -1)
        sess.update_partials(update_repeats=1 if self.maps_mode else self.upd)
        if self.repeats:
            self.upd = 0
        if self.reduce == "peer":
            return lib.pll_gpu_group_edge_loglikelihood(sess.p, self.group, e[0], e[1], e[2], e[3], e[4], self.api.uptr(self.fi), None)
        if self.reduce == "rccl" and self.rccl_comm is not None:
            return lib.pll_gpu_edge_loglikelihood_allreduce(sess.p, self.rccl_comm, e[0], e[1], e[2], e[3], e[4], self.api.uptr(self.fi))
        if self.device_path:
            if not lib.pll_gpu_edge_loglikelihood_async(sess.p, e[0], e[1], e[2], e[3], e[4], self.api.uptr(self.fi),
                                                        self.red.data_ptr()):
                raise SystemExit(f"pll_gpu_edge_loglikelihood_async: [{lib.errno()}] {lib.errmsg()}")
            # the path's one exchange: sum of the shards' log-likelihoods (word 0); word 1 = every rank's
            # call sequence number, so its sum tells the host which evaluation the pair belongs to
            h.dist.all_reduce(self.red)
            self.pinned[0:1].copy_(self.red[0:1], non_blocking=True)
            self.pinned[1:2].copy_(self.red[1:2], non_blocking=True)
            if self.expected is not None:
                self.expected += h.world
                t_spin = time.perf_counter()
                while self.pview[1] != self.expected:
                    if time.perf_counter() - t_spin > 0.02:
                        self.expected = None
                        break
            if self.expected is None:
                h.torch.cuda.current_stream().synchronize()
                self.expected = float(self.pview[1])
            return float(self.pview[0])
        v, _ = sess.edge_lnl(e, persite=False)
        if self.reduce == "rccl":
            self.red[0] = v
            h.dist.all_reduce(self.red)
            v = float(self.red[0].item())
        return v

    def loop_mode(self):
        """first_update_repeats of csrc/workload/step_loop.c"""
        return {"forced": 3, "default": 2}.get(self.maps_mode, self.upd)

    def steps(self, n):
        """n steps; returns the last log-likelihood. The C loop serves every mode whose step is two C-ABI calls."""
        if n <= 0:
            return None
        if self.c_driver and self.reduce in (None, "peer") and len(self.case.op_batches) == 1:
            import ctypes as C
            lib, sess = self.lib, self.sess
            fp = lambda f: C.cast(f, C.c_void_p)
            lnl = C.c_double(0.0)
            grp = self.group if self.reduce == "peer" else None
            step_loop_fn()(fp(lib.pll_update_partials_rep), fp(lib.pll_compute_edge_loglikelihood),
                           fp(lib.pll_gpu_group_edge_loglikelihood), C.cast(sess.p, C.c_void_p), grp,
                           C.cast(sess._op_arrays[0], C.c_void_p), len(self.case.op_batches[0]), self.loop_mode(), self.edge_c,
                           self.api.uptr(self.fi), n, C.byref(lnl), fp(lib.pll_gpu_invalidate))
            if self.repeats:
                self.upd = 0
            return lnl.value
        if self.c_driver and self.reduce == "rccl" and self.rccl_comm is not None and len(self.case.op_batches) == 1:
            import ctypes as C
            lib, sess = self.lib, self.sess
            fp = lambda f: C.cast(f, C.c_void_p)
            lnl = C.c_double(0.0)
            fn = _allreduce_loop_fn()
            fn(fp(lib.pll_update_partials_rep), fp(lib.pll_gpu_edge_loglikelihood_allreduce), C.cast(sess.p, C.c_void_p), self.rccl_comm,
               C.cast(sess._op_arrays[0], C.c_void_p), len(self.case.op_batches[0]), self.loop_mode(), self.edge_c, self.api.uptr(self.fi), n, C.byref(lnl),
               fp(lib.pll_gpu_invalidate))
            if self.repeats:
                self.upd = 0
            return lnl.value
        v = None
        for _ in range(n):
            v = self.step()
        return v

    def fence(self):
        self.lib.pll_gpu_synchronize(self.sess.p)
        if self.collective:
            self.h.barrier()

    def timed(self, warmup, steps, blocks=1):
        """`warmup` untimed steps, then `blocks` blocks of exactly `steps` steps, each between two fences, max over
        ranks per block. Returns (block times, last lnL)."""
        lnl = self.steps(warmup)
        out = []
        for _ in range(blocks):
            self.fence()
            t0 = time.perf_counter()
            lnl = self.steps(steps)
            self.fence()
            dt = time.perf_counter() - t0
            out.append(self.h.max_over_ranks(dt) if self.collective else dt)
        return out, lnl

    def timed_with_class_maps(self, warmup, steps, blocks=1, mode="forced"):
        """the same timed blocks with pll_update_partials as the reference defines it on a SITE_REPEATS partition
        (src/partials.c:237-255: update_repeats = 1, pll_update_repeats inside the op loop) in EVERY step. mode "forced":
        every map is computed again by every step (PLL_GPU_FORGET_REPEATS first - the reference's own cost model, and the
        figure the scaling claim is made on); mode "default": the call as it is, on the unchanged tree - the library
        recognises that no map's inputs moved. Returns (block times, last lnL)."""
        self.maps_mode = mode
        try:
            return self.timed(warmup, steps, blocks)
        finally:
            self.maps_mode = None

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
                if ur:
                    lib.pll_gpu_invalidate(sess.p, self.api.FORGET_REPEATS, -1)
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


def block_stats(block_s, steps):
    """median block -> ms per step; min / max for the spread"""
    import numpy as np
    ms = sorted(b / steps * 1e3 for b in block_s)
    return float(np.median(ms)), ms[0], ms[-1]


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
    cc16 = False
    if cc:
        # 4x4 with tips as codes: most of the step is ONE launch, the groups of seven ops over complete
        # 8-tip subtrees (k_partials_dna_cc<5,5>). The leg re-runs exactly those ops: everything within
        # three levels of the tips.
        depth = {t: 0 for t in range(case.tips)}
        for op in all_ops:
            depth[op[0]] = 1 + max(depth[op[2]], depth[op[5]])
        low = [op for op in all_ops if depth[op[0]] <= 3]
        # round 4: complete 16-tip subtrees are groups of FIFTEEN ops (k_partials_dna_cc16): everything within four levels
        low16 = [op for op in all_ops if depth[op[0]] <= 4]
        # (used by size - csrc/hip/chain_plan.h: use_cc16 - unless PLL_AMD_FUSE_CC16 says 0 / 1)
        want16 = {"0": False, "1": True}.get(os.environ.get("PLL_AMD_FUSE_CC16", ""), sites * nops >= 9000000)
        if len(low16) == 15 * (case.tips // 16) and case.tips >= 32 and want16:
            ii_ops, cc16 = low16, True
        elif len(low) == 7 * (case.tips // 8):
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
        ii_ops, cc, cc16, grouped = list(all_ops), False, False, False
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
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
    if os.path.exists(tfile) and not args.pattern_tip and not args.sites and not args.taxa and args.tree == "balanced" and not args.tips:
        # PMC-derived HBM bytes per launch of the same command: NOT measured in this run (counters need rocprofv3
        # around the process) but read from the committed summary of the round's PMC passes - and only when that
        # summary is about the kernel this leg launches
        tj = json.load(open(tfile))
        traffic, traffic_source = tj.get("hbm_bytes_per_launch"), dict(
            file="profiles/" + os.path.basename(tfile), kernel=tj.get("kernel"), measured_at=tj.get("commit", "round 2 PMC passes"),
            note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/profile_round.sh), not this run")
    mfma = cfg["states"] > 32 and not os.environ.get("PLL_AMD_NO_MFMA", "0").strip("0")
    gg = cfg["states"] == 4 and cfg.get("repeats") and not os.environ.get("PLL_AMD_NO_FUSE_GG", "0").strip("0")
    kernel = {4: "k_partials_dna_cc16%.0s" if cc16 else "k_partials_dna_cc<5,5>%.0s" if cc else "k_partials_dna_fused<4,4>%.0s" if fused else
                 "k_partials_dna<false,false,true>%.0s (compressed levels) + k_partials_dna_gg (where compression ends)" if gg else "k_partials_dna<false,false,%s>",
              20: "k_partials_mfma_cc<5>%.0s" if grouped else
                  ("k_partials_lean<5,false,false,true>%.0s (gathering launches) + k_partials_tiled<20,false,false,false>"
                   if cfg.get("repeats") and not os.environ.get("PLL_AMD_NO_LEAN", "0").strip("0") else "k_partials_tiled<20,false,false,%s>"),
              61: ("k_partials_mfma<16,false,false,%s>" if cfg.get("repeats") or os.environ.get("PLL_AMD_MFMA_WIDE") == "0" else
                   "k_partials_mfma_wide<15,1,8>%.0s") if mfma else "k_partials_tiled<32,false,false,%s>"
              }[cfg["states"]] % ("true" if cfg.get("repeats") else "false")
    if args.tree != "balanced":
        kernel = "all update launches of the traversal"
    if traffic_source and (traffic_source["kernel"] or "").replace(" ", "") not in kernel.replace(" ", ""):
        traffic, traffic_source = None, None  # the summary belongs to another kernel (a stale file): say nothing
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
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_source, kernel=kernel,
                        launches=launches, ops_in_those_launches=len(ii_ops), avg_launch_ms=round(per_launch_ms, 5),
                        algorithmic_bytes_per_launch=int(per_launch_bytes))
    # what `frac` is made of: numerator = the launches' bytes AS GROUPED, reported by the library; checked against the PMC
    # passes' HBM bytes of the same launch where a committed summary exists (traffic / numerator)
    roofline["frac_numerator"] = ("algorithmic bytes of the timed launches as grouped - fused groups do not read their intermediate CLVs back - "
                                  "reported by the library (pll_gpu_last_algorithmic_bytes)"
                                  + (", checked against traffic = %.3f x" % (traffic / per_launch_bytes) if traffic else ", no PMC summary for this configuration"))
    roofline["unfused_equivalent"] = dict(
        label="equivalent bandwidth, NOT a roofline fraction (fusion removes the child reads this accounting charges; it may exceed the peak)",
        note="the same ops priced at SURVEY 8d's per-update bytes (every op reads both children from HBM)",
        GBps=round(unfused_bytes / launches / (per_launch_ms * 1e-3) / 1e9, 1), bytes_per_launch=int(unfused_bytes / launches),
        frac_of_hbm_peak=round(unfused_bytes / launches / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    roofline.update(full_traversal=dict(
        launches=launches_full, ms=round(ms_full / reps, 5), algorithmic_bytes=int(bytes_full or op_bytes(case, api, all_ops, entries)),
        algorithmic_GBps=round((bytes_full or op_bytes(case, api, all_ops, entries)) / (ms_full / reps * 1e-3) / 1e9, 1),
        update_partials_only_M_per_s=round(sites * nops / (ms_full / reps * 1e-3) / 1e6, 1)))
    return roofline, codes


def live_traffic(args, kernel):
    """roofline.traffic measured IN THIS RUN: two child passes of this very command under rocprofv3 (--pmc FETCH_SIZE and --pmc
    WRITE_SIZE cannot share a pass on gfx950; --kernel-trace only, as the guide prescribes), 3 steps each, the mean over the
    dispatches of `kernel`; counters are KiB, FETCH_SIZE doubled (gfx950 counts 64 B per 128-B read request: tools/pmc_calib.hip),
    WRITE_SIZE exact. Returns (bytes per launch, source dict) or (None, reason). Not under a profiler, not in a child of itself."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "running under a profiler"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "no rocprofv3 on this box"
    want = kernel.replace(" ", "")
    tmp = tempfile.mkdtemp(prefix="pll_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--config", args.config, "--steps", "3", "--warmup", "1", "--blocks", "1", "--no-cpu", "--no-traffic"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=60)
            except subprocess.TimeoutExpired:
                return None, f"the {counter} pass did not finish within 60 s"
            if r.returncode != 0:
                return None, f"the {counter} pass failed (exit {r.returncode}): " + r.stderr.decode(errors="replace")[-200:]
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter and want in row["Kernel_Name"].replace(" ", ""):
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"no {counter} rows for {kernel}"
            sums[counter] = (sum(vals) / len(vals), len(vals))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    per_launch = (2.0 * sums["FETCH_SIZE"][0] + sums["WRITE_SIZE"][0]) * 1024.0
    return int(round(per_launch)), dict(
        measured="in this run", kernel=kernel, dispatches=[sums["FETCH_SIZE"][1], sums["WRITE_SIZE"][1]],
        fetch_KiB_mean=round(sums["FETCH_SIZE"][0], 2), write_KiB_mean=round(sums["WRITE_SIZE"][0], 2),
        method="two child passes of this command under rocprofv3 (--pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only, "
               "--steps 3 --warmup 1 --no-cpu); mean over the kernel's dispatches; counters are KiB; FETCH_SIZE doubled (gfx950 counts 64 B per "
               "128-B read request, tools/pmc_calib.hip); WRITE_SIZE exact")


def step_roofline(roofline, ms_per_step):
    """roofline.step: the WHOLE timed step - every update launch of the traversal as grouped plus the evaluation, whose
    operands the last launch leaves behind - against the HBM peak: as-launched bytes / ms_per_step"""
    b = roofline["full_traversal"]["algorithmic_bytes"]
    gbps = b / (ms_per_step * 1e-3) / 1e9
    roofline["step"] = dict(algorithmic_bytes=b, ms=round(ms_per_step, 5), GBps=round(gbps, 1), frac=round(gbps / HBM_PEAK_GBS, 4),
                            note="as-launched bytes of all update launches of one step / ms_per_step (launch gaps, the evaluation's tail and the host hand-off included)")


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
    runner = Runner(h, lib, api, driver, case, cfg.get("repeats"), reduce="rccl" if h.dist else None, c_driver=args.driver == "c")
    blocks_s, lnl = runner.timed(args.warmup, args.steps, args.blocks)
    if not np.isfinite(lnl) and not os.environ.get("PLL_BENCH_MEASUREMENT_BUILD"):  # (tools/r3_wide_experiments.sh: builds whose results are wrong on purpose)
        raise SystemExit(f"hot path failed: lnL = {lnl} [{lib.errno()}] {lib.errmsg()}")
    ms, ms_min, ms_max = block_stats(blocks_s, args.steps)
    value = total_sites * nops / (ms * 1e-3) / 1e6
    roofline, codes = roofline_leg(args, cfg, lib, api, runner)
    step_roofline(roofline, ms)
    if roofline.get("bound") == "hbm" and roofline.get("traffic_source") and not args.no_cpu and not args.no_traffic and h.world == 1:
        # the HBM bytes of the dominant kernel measured by THIS run (the committed summary stays as the fall-back, and says so)
        kname = roofline["kernel"]
        if " " not in kname and "+" not in kname:
            got, src = live_traffic(args, kname.replace(",", ", "))
            if got:
                roofline["traffic"], roofline["traffic_source"] = got, src
                roofline["frac_numerator"] = roofline["frac_numerator"].split(", checked against")[0] + ", checked against traffic = %.3f x" % (
                    got / roofline["algorithmic_bytes_per_launch"])
            else:
                roofline["traffic_source"]["live_passes"] = "not taken: " + str(src)
    out = {
        "metric": "M site-CLV-updates/s", "value": round(value, 1), "unit": "M site-CLV-updates/s",
        "n_gpus": h.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
        "blocks": args.blocks, "ms_per_step_min": round(ms_min, 4), "ms_per_step_max": round(ms_max, 4),
        "higher_is_better": True, "scaling": None, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (SURVEY 8d: xorshift64 alignment, seed 88172645463325252)",
        "config": {"workload": cfg["desc"] + tip_note(case, api, args),
                   "sites_per_gpu": total_sites, "ops_per_traversal": nops, "states": cfg["states"], "rate_cats": 4,
                   "taxa": cfg["tips"], "step": STEP_REUSED if cfg.get("repeats") else STEP_PLAIN,
                   "timed": f"median of {args.blocks} blocks of {args.steps} steps, each block between two device synchronises",
                   "driver": "C loop over the two C-ABI calls (csrc/workload/step_loop.c)" if args.driver == "c" and not h.dist else "Python ctypes loop",
                   "parallelism": "1 GPU"},
        "lnl": lnl, "roofline": roofline,
    }
    if cfg.get("repeats"):
        # THREE timed figures of the same blocks x steps: maps re-used (value, above); the reference's default call on the
        # unchanged tree; every class map forced to be computed again by every step
        bdc, lnl_dc = runner.timed_with_class_maps(args.warmup, args.steps, args.blocks, mode="default")
        ms_dc = block_stats(bdc, args.steps)[0]
        out["config"]["step_default_call"] = STEP_DEFAULT
        out["ms_per_step_default_call"] = round(ms_dc, 4)
        out["value_default_call"] = round(total_sites * nops / (ms_dc * 1e-3) / 1e6, 1)
        out["lnl_default_call"] = lnl_dc
        bcm, lnl_cm = runner.timed_with_class_maps(args.warmup, args.steps, args.blocks, mode="forced")
        ms_cm = block_stats(bcm, args.steps)[0]
        out["config"]["step_with_class_maps"] = STEP_MAPS
        out["ms_per_step_with_class_maps"] = round(ms_cm, 4)
        out["value_with_class_maps"] = round(total_sites * nops / (ms_cm * 1e-3) / 1e6, 1)
        out["lnl_with_class_maps"] = lnl_cm
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
    # every rank computes the same cuts from the same alignment
    bounds = sharding.balanced_bounds(full, h.world) if args.cut == "balanced" else sharding.shard_bounds(full.sites, h.world)
    mine = sharding.shard_case(full, h.rank, h.world, bounds)
    reduce = args.reduce
    runner = Runner(h, lib, api, driver, mine, True, reduce=reduce, c_driver=args.driver == "c")
    blocks_s, lnl = runner.timed(args.warmup, args.steps, args.blocks)
    if not np.isfinite(lnl):
        raise SystemExit(f"hot path failed: lnL = {lnl} [{lib.errno()}] {lib.errmsg()}")
    tN_ms, tN_min, tN_max = block_stats(blocks_s, args.steps)
    value = total_sites * nops / (tN_ms * 1e-3) / 1e6
    exchange = {"reduce": reduce, reduce + "_ms_per_step": round(tN_ms, 4)}
    # the same steps with the other exchange, for the record (both in one line): RCCL needs the nccl backend
    if reduce == "rccl":
        exchange["rccl_via"] = runner.rccl_via
    # what the exchange adds behind a result that is already in host memory: the group sum alone, all ranks in step
    if reduce == "peer":
        v = np.array([1.0])
        o = np.zeros(1)
        h.barrier()
        t0 = time.perf_counter()
        for _ in range(2000):
            lib.pll_gpu_group_sum(runner.group, api.dptr(v), 1, api.dptr(o))
        exchange["peer_exchange_alone_us"] = round(h.max_over_ranks((time.perf_counter() - t0) / 2000 * 1e6), 3)
        exchange["peer_exchange_alone_note"] = "2000 group sums back to back through ctypes (~1-2 us of that is the Python call); tools/group_latency.c measures the C call"
    # the same steps with the reference's default call on the unchanged tree, and with every class map forced to be
    # computed again by every step (speedup_with_class_maps is defined on the FORCED figure)
    bdc, lnl_dc = runner.timed_with_class_maps(args.warmup, args.steps, args.blocks, mode="default")
    tN_dc_ms = block_stats(bdc, args.steps)[0]
    bcm, lnl_cm = runner.timed_with_class_maps(args.warmup, args.steps, args.blocks, mode="forced")
    tN_cm_ms = block_stats(bcm, args.steps)[0]
    rep_extra, rep_full = runner.repeats_update_ms()
    rep_extra = h.max_over_ranks(rep_extra)
    shard_levels = h.gather_ints(runner.level_entries())
    shard_sites = h.gather_ints([mine.sites])
    affinities = h.gather_objects(h.cpu_affinity)  # per rank: the cores it was pinned to (next to its GPU)
    # roofline of the dominant kernel on this rank's shard (rank 0 reports)
    args.config = "c4"
    roofline, _ = roofline_leg(args, cfg, lib, api, runner)
    step_roofline(roofline, tN_ms)
    h.barrier()
    out = None
    if h.rank == 0:
        # t1: the un-sharded alignment on ONE GPU, same step, same K/W, nothing else running on the node
        solo = Runner(h, lib, api, driver, full, True, reduce=None, c_driver=args.driver == "c")
        b1, lnl1 = solo.timed(args.warmup, args.steps, args.blocks)
        t1_ms = block_stats(b1, args.steps)[0]
        b1dc, _ = solo.timed_with_class_maps(args.warmup, args.steps, args.blocks, mode="default")
        t1_dc_ms = block_stats(b1dc, args.steps)[0]
        b1cm, _ = solo.timed_with_class_maps(args.warmup, args.steps, args.blocks, mode="forced")
        t1_cm_ms = block_stats(b1cm, args.steps)[0]
        global_levels = solo.level_entries()
        rep1_extra, _ = solo.repeats_update_ms(reps=3)
        solo.close()
        pin = pinned_lnl("c4") if not args.sites else None
        out = {
            "metric": "M site-CLV-updates/s", "value": round(value, 1), "unit": "M site-CLV-updates/s",
            "n_gpus": h.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(tN_ms, 4),
            "blocks": args.blocks, "ms_per_step_min": round(tN_min, 4), "ms_per_step_max": round(tN_max, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (SURVEY 8d: xorshift64 alignment, seed 88172645463325252; the same alignment on every rank)",
            "config": {"workload": cfg["desc"] + f", pattern-sorted (pll_compress_site_patterns) and cut into {h.world} contiguous site ranges"
                       + (" of equal cost (sites + 0.19 x class entries of the tip-only subtrees)" if args.cut == "balanced" else " of equal site count")
                       + tip_note(full, api, args),
                       "total_sites": total_sites, "patterns": full.sites, "sites_per_gpu": [s[0] for s in shard_sites],
                       "ops_per_traversal": nops, "states": 4, "rate_cats": 4, "taxa": cfg["tips"],
                       "step_with_class_maps": STEP_MAPS.replace("pll_compute_edge_loglikelihood", "edge log-likelihood + the same exchange"),
                       "step_default_call": STEP_DEFAULT.replace("pll_compute_edge_loglikelihood", "edge log-likelihood + the same exchange"),
                       "step": "pll_update_partials_rep(full traversal, update_repeats = 0: the class maps of the first step re-used) + edge log-likelihood + the one exchange: sum of the shard lnLs"
                               + (" in rank order through shared host memory (pll_gpu_group_edge_loglikelihood)" if reduce == "peer" else " by one RCCL all-reduce on the device"),
                       "timed": f"median of {args.blocks} blocks of {args.steps} steps, each block between two barriers + device synchronises, max over ranks per block",
                       "driver": "C loop over the two C-ABI calls (csrc/workload/step_loop.c)" if args.driver == "c" and reduce == "peer" else "Python ctypes loop",
                       "parallelism": f"sites sharded x{h.world}, one sum of one double per step ({reduce}; control plane {h.backend})"},
            "lnl": lnl, "t1_ms": round(t1_ms, 4), "tN_ms": round(tN_ms, 4), "speedup": round(t1_ms / tN_ms, 3),
            "t1_value": round(total_sites * nops / (t1_ms * 1e-3) / 1e6, 1), "lnl_unsharded": lnl1,
            "exchange_timed": reduce,  # which exchange `value` / tN_ms were timed with (never the other one's number)
            "cpu_affinity": affinities,
            "ms_per_step_with_class_maps": round(tN_cm_ms, 4), "value_with_class_maps": round(total_sites * nops / (tN_cm_ms * 1e-3) / 1e6, 1),
            "t1_ms_with_class_maps": round(t1_cm_ms, 4), "speedup_with_class_maps": round(t1_cm_ms / tN_cm_ms, 3),
            "lnl_with_class_maps": lnl_cm,
            "ms_per_step_default_call": round(tN_dc_ms, 4), "value_default_call": round(total_sites * nops / (tN_dc_ms * 1e-3) / 1e6, 1),
            "t1_ms_default_call": round(t1_dc_ms, 4), "speedup_default_call": round(t1_dc_ms / tN_dc_ms, 3), "lnl_default_call": lnl_dc,
            "lnl_rel_err_vs_unsharded": abs(lnl - lnl1) / abs(lnl1), "exchange": exchange,
            "repeats_update_ms": round(rep_extra, 4), "repeats_update_ms_unsharded": round(rep1_extra, 4),
            "entries_per_level": {"unsharded": global_levels, "shards": shard_levels,
                                  "sum_over_shards_div_unsharded": round(sum(sum(s) for s in shard_levels) / max(sum(global_levels), 1), 4)},
            "roofline": roofline,
        }
        if pin is not None:
            out["lnl_pinned_reference"] = pin
            out["lnl_rel_err_pinned"] = abs(lnl - pin) / abs(pin)
    # ---- LAST: the same steps with the other exchange, for the record (both in one line). The RCCL form goes through the
    # library's entry point on a communicator made here - the one part of this flow that only a multi-GPU node exercises -
    # so it runs when everything else of the line exists, under a watchdog: if it does not come back (a collective that
    # never completes), rank 0 prints the line without it and every rank leaves; if it raises, the line says so.
    if reduce == "peer" and h.on_device:
        h.barrier()
        limit = float(os.environ.get("PLL_BENCH_RCCL_LEG_TIMEOUT_S", "180"))

        def give_up():
            if h.rank == 0 and out is not None:
                out["exchange"]["rccl_error"] = f"the RCCL leg did not finish within {limit:.0f} s; line printed by the watchdog"
                line = (json.dumps(out) + "\n").encode()
                os.write(h.real_stdout if h.real_stdout is not None else 1, line)
            if h.group:  # the shared-memory segment of the peer exchange: leave it, so that its name is unlinked (host memory only)
                h.group_lib.pll_gpu_group_leave(h.group)
                h.group = None
            # a collective that never completes is a hang on a process that holds the GPU: the line is out, the exit
            # status says what happened (the launcher returns the worst rank's)
            os._exit(RCCL_LEG_HUNG_RC)

        import threading
        dog = threading.Timer(limit, give_up)
        dog.daemon = True
        dog.start()
        try:
            runner.reduce = "rccl"
            runner.setup_rccl()
            b2, lnl2 = runner.timed(args.warmup, args.steps, args.blocks)
            if out is not None:
                out["exchange"]["rccl_ms_per_step"] = round(block_stats(b2, args.steps)[0], 4)
                out["exchange"]["rccl_lnl_rel_diff"] = abs(lnl2 - lnl) / abs(lnl)
                out["exchange"]["rccl_via"] = runner.rccl_via
        except Exception as exc:  # noqa: BLE001
            if out is not None:
                out["exchange"]["rccl_error"] = f"{type(exc).__name__}: {exc}"
        dog.cancel()
        runner.reduce, runner.device_path = "peer", False
    runner.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of `steps` steps each; the line reports the median block")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--sites", type=int, default=0, help="override the number of sites (N > 1: of the whole alignment)")
    ap.add_argument("--pattern-tip", action="store_true", help="PLL_ATTRIB_PATTERN_TIP variant")
    ap.add_argument("--tips", default=None, choices=["states", "clv"],
                    help="how tips are set: pll_set_tip_states or one-hot CLVs through pll_set_tip_clv (default: per config, SURVEY 8d)")
    ap.add_argument("--tree", default="balanced", choices=["balanced", "random", "caterpillar"],
                    help="topology (BASELINE's configs are balanced; the others show what irregular level structures cost)")
    ap.add_argument("--taxa", type=int, default=0, help="override the number of taxa")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (and the live HBM-counter passes)")
    ap.add_argument("--no-traffic", action="store_true", help="roofline.traffic from the committed summary instead of two rocprofv3 counter passes of this command")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the control plane (barriers, max over ranks) and of --reduce rccl; "
                         "gloo + PLL_BENCH_SAME_DEVICE=1 rehearses the N>1 flow on a one-GPU box")
    ap.add_argument("--reduce", default="peer", choices=["peer", "rccl"],
                    help="N > 1: how the shard log-likelihoods are summed (see the docstring)")
    ap.add_argument("--cut", default="equal", choices=["equal", "balanced"],
                    help="N > 1: shards of equal site count (default) or of equal modelled cost with a size cap (pllamd/sharding.py: balanced_bounds - about 2 % better on one GPU, inside the spread: see there)")
    ap.add_argument("--driver", default="c", choices=["c", "python"], help="who issues the steps of the timed region")
    args = ap.parse_args()
    if args.gpus < 1 or args.blocks < 1 or args.steps < 1:
        ap.error("--gpus, --steps and --blocks must be positive")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher: become one (before anything touches the GPU or imports torch)
        sys.exit(spawn_ranks(args))
    if env_world is not None and int(env_world) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: refusing to run another configuration than the one asked for",
              file=sys.stderr)
        sys.exit(2)
    h = Harness(args)
    # (PLL_BENCH_FORCE_STRONG=1 + PLL_BENCH_FORCE_DIST=1: the sharded flow with a world of one - tests/test_gpu_bench_flow.py
    # rehearses the RCCL form of the exchange on the one GPU a test box has, where RCCL refuses two ranks on a device)
    strong = h.world > 1 or (os.environ.get("PLL_BENCH_FORCE_STRONG") == "1" and h.dist is not None)
    out = main_strong(args, h) if strong else main_single(args, h)
    h.finish()
    if h.rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
