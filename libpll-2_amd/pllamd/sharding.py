"""Site sharding for multi-GPU runs: contiguous site ranges per rank, every
rank owns an independent partition over its range; the only exchange is the sum of the per-shard
log-likelihoods. No data-path collective exists anywhere else."""
import os

import numpy as np

from .driver import Case


def shard_bounds(sites, world, align=64):
    """[lo, hi) per rank; interior cuts are multiples of `align` (wave64 tiles) when possible"""
    cuts = [0]
    for r in range(1, world):
        c = int(round(sites * r / world / align)) * align if sites >= world * align else sites * r // world
        cuts.append(min(max(c, cuts[-1]), sites))
    cuts.append(sites)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


# what a shard's step costs, relative: one site of the alignment against one class entry of the ops whose
# subtrees are tips only (the compressing levels of a site-repeats traversal). From profiles/r2_c4_projection.json:
# shards of equal site count took 0.114 .. 0.124 ms with 127k .. 234k such entries (0.09 us per 1000 entries) and
# about 0.48 us per 1000 sites above the fixed part of a step.
ENTRY_COST_PER_SITE_COST = 0.19


def _tip_subtree_keys(case: Case):
    """for every op whose subtree holds tips only and at most 8 of them: one 64-bit word per site that is
    equal for two sites iff all tips below the op agree there (its site-repeat class, src/repeats.c:299-382)"""
    tips_below = {t: [t] for t in range(case.tips)}
    keys = []
    seqs = [np.frombuffer(bytes(sq), dtype=np.uint8) for sq in case.sequences]
    for op in case.op_batches[0]:
        l, r = tips_below.get(op[2]), tips_below.get(op[5])
        if l is None or r is None or len(l) + len(r) > 8:
            continue
        tips_below[op[0]] = l + r
        k = np.zeros(case.sites, dtype=np.uint64)
        for i, t in enumerate(tips_below[op[0]]):
            k |= seqs[t].astype(np.uint64) << np.uint64(8 * i)
        keys.append(k)
    return keys


def balanced_bounds(case: Case, world, align=64, rounds=6, max_ratio=1.04):
    """Cuts that equalise the COST of the shards rather than their site counts. With site repeats a shard
    computes one entry per class below the level where compression ends, and a contiguous range of the
    pattern-sorted alignment holds very different class counts depending on where it lies (the first and last
    eighth of the sorted 1M-site alignment: 127k entries at the 8-tip level, the second: 234k). cost(range) =
    sites + ENTRY_COST_PER_SITE_COST x sum over tip-only subtrees of the distinct columns in the range; the cuts
    move until the costs agree. Every rank computes the same cuts from the same alignment: nothing is exchanged.

    MEASURED (round 3, tools/c4_projection.py --cut balanced against --cut equal, two boxes): NOT a gain for the 1M-site
    configuration on 8 ranks - the slowest shard went 0.121 -> 0.123-0.127 ms. The cuts give the entry-poor first and
    last eighth 134k sites instead of 125k, and at 134k sites a shard's uncompressed CLVs (14 nodes x 17 MB) no longer
    fit the 256 MB Infinity Cache beside the compressed ones: what the model gains in entries it loses to HBM.
    bench.py therefore cuts by site count (--cut equal) unless told otherwise; the function stays for alignments
    whose shards are far from that cliff.

    Round 4: with `max_ratio` (no shard larger than 1.04 x the mean: below the cliff) the same cost model no longer
    loses - tools/round4_calls/r4_cuts.sh, same box, alternating, slowest shard in ms: equal sizes 0.1123 / 0.1127,
    uncapped 0.1137 / 0.1139, capped at 1.03 0.1120 / 0.1076, at 1.04 0.1089 / 0.1122, at 1.05 0.1110 / 0.1123: about 2 %
    better than equal sizes on average, inside the per-shard spread of one call (+-3 %). --cut equal stays the default;
    --cut balanced now means the capped form (PLL_SHARD_MAX_RATIO=0 removes the cap)."""
    if world == 1 or case.sequences is None or not case.op_batches:
        return shard_bounds(case.sites, world, align)
    keys = _tip_subtree_keys(case)
    # Round 4: no shard more than `max_ratio` x the mean size (the cliff described above sits at ~1.055 x for the 1M-site
    # configuration on 8 ranks: 132k sites). None = uncapped (round 3's form).
    env = os.environ.get("PLL_SHARD_MAX_RATIO")
    if env is not None:
        max_ratio = float(env) if float(env) > 0 else None
    cap = None if max_ratio is None else int(np.ceil(max_ratio * case.sites / world / align)) * align

    def cost(lo, hi):
        return (hi - lo) + ENTRY_COST_PER_SITE_COST * sum(len(np.unique(k[lo:hi])) for k in keys)

    cuts = [b[0] for b in shard_bounds(case.sites, world, align)] + [case.sites]
    for _ in range(rounds):
        costs = [cost(cuts[r], cuts[r + 1]) for r in range(world)]
        target = sum(costs) / world
        # piecewise-linear cumulative cost over the current cuts; new cut r where it reaches r x target
        cum = np.concatenate([[0.0], np.cumsum(costs)])
        new = [0]
        for r in range(1, world):
            c = int(round(np.interp(r * target, cum, cuts) / align)) * align
            new.append(min(max(c, new[-1] + align), case.sites - (world - r) * align))
        new.append(case.sites)
        new = _cap_shards(new, cap, align)
        if new == cuts:
            break
        cuts = new
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _cap_shards(cuts, cap, align):
    """no shard larger than `cap` sites: the excess moves on to the next shard (forward pass), then back from the last
    (backward pass); cuts stay multiples of `align`"""
    if cap is None:
        return cuts
    cuts = list(cuts)
    world = len(cuts) - 1
    for r in range(1, world):          # forward: shard r - 1 gives its excess to shard r
        cuts[r] = min(cuts[r], cuts[r - 1] + cap)
    for r in range(world - 1, 0, -1):  # backward: shard r gives its excess to shard r - 1
        cuts[r] = max(cuts[r], cuts[r + 1] - cap)
    return cuts


def shard_case(case: Case, rank, world, bounds=None):
    lo, hi = (bounds or shard_bounds(case.sites, world))[rank]
    kw = dict(name=f"{case.name}[{lo}:{hi}]", states=case.states, rate_cats=case.rate_cats, tips=case.tips,
              sites=hi - lo, pmatrix=case.pmatrix, freqs=case.freqs, op_batches=case.op_batches, edges=case.edges,
              roots=case.roots, attributes=case.attributes, clv_buffers=case.clv_buffers,
              scale_buffers=case.scale_buffers, rate_weights=case.rate_weights,
              pattern_weights=np.asarray(case.pattern_weights)[lo:hi], prop_invar=case.prop_invar,
              freqs_indices=case.freqs_indices, dump_clvs=case.dump_clvs)
    if case.sequences is not None:
        kw.update(charmap=case.charmap, sequences=[s[lo:hi] for s in case.sequences])
    else:
        kw.update(tip_clvs=np.asarray(case.tip_clvs)[:, lo:hi])
    return Case(**kw)


def sort_columns(lib, case: Case):
    """The alignment the way an application hands it to the likelihood code: through
    pll_compress_site_patterns (src/compress.c:171-410; here the device radix sort of
    csrc/hip/compress.hip), i.e. unique columns in lexicographic order plus their multiplicities as
    pattern weights. The log-likelihood is a weighted sum over columns, so it does not change; sites
    that share a prefix of tips become neighbours, and a contiguous shard of the sorted alignment
    holds fewer site-repeat classes per node than the same number of columns taken at random."""
    import ctypes as C
    from . import api
    assert case.sequences is not None and np.all(np.asarray(case.pattern_weights) == 1)
    n, length = len(case.sequences), case.sites
    bufs = [C.create_string_buffer(bytes(s), length + 1) for s in case.sequences]
    arr = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    m = (C.c_ulonglong * 256)(*[int(v) for v in case.charmap])
    ln = C.c_int(length)
    w = lib.pll_compress_site_patterns(arr, m, n, C.byref(ln))
    if not w:
        raise RuntimeError(f"pll_compress_site_patterns: [{lib.errno()}] {lib.errmsg()}")
    weights = api.as_np(w, ln.value, np.uint32).copy()
    # the weights block came from the library's allocator (free() as in the reference); ctypes leaves it
    kw = dict(name=case.name + "[sorted]", states=case.states, rate_cats=case.rate_cats, tips=case.tips,
              sites=ln.value, pmatrix=case.pmatrix, freqs=case.freqs, op_batches=case.op_batches, edges=case.edges,
              roots=case.roots, attributes=case.attributes, clv_buffers=case.clv_buffers,
              scale_buffers=case.scale_buffers, rate_weights=case.rate_weights, pattern_weights=weights,
              prop_invar=case.prop_invar, freqs_indices=case.freqs_indices, dump_clvs=case.dump_clvs,
              charmap=case.charmap, sequences=[b.raw[:ln.value] for b in bufs], model=case.model)
    return Case(**kw)


def allreduce_sum(value, dist, device=None):
    """sum a Python float over all ranks with one all_reduce of one double (RCCL on GPUs, gloo on
    CPU); identity when torch.distributed is not initialised"""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t)
    return float(t.item())
