"""Site sharding for multi-GPU runs: contiguous site ranges per rank, every
rank owns an independent partition over its range; the only exchange is the sum of the per-shard
log-likelihoods. No data-path collective exists anywhere else."""
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


def shard_case(case: Case, rank, world):
    lo, hi = shard_bounds(case.sites, world)[rank]
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
