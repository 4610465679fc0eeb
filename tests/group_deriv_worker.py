"""one rank of tests/test_gpu_sharding_product.py::test_sharded_newton_steps_stay_in_lockstep: a shard of one alignment,
sumtable of the root edge, then Newton-Raphson on that branch with pll_gpu_group_likelihood_derivatives - every rank
must see the same derivative bits and end at the same branch length."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
sys.path.insert(0, ROOT)


def main():
    rank, world, name, kwjson = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["PLL_AMD_DEVICE"] = "0"
    import numpy as np
    from pllamd import api, driver, sharding, workload as W
    kw = json.loads(kwjson)
    case = W.make_case("full", **kw)
    sub = sharding.shard_case(case, rank, world) if world > 1 else case
    lib = api.PllLib()
    g = lib.pll_gpu_group_join(name.encode(), rank, world, 60000) if world > 1 else None
    assert world == 1 or g, lib.errmsg()
    trace = []
    with driver.Session(lib, sub, api.ARCH_AVX2) as s:
        s.set_model(case.model["exch"], case.freqs, case.model["rates"])  # (the model is the whole alignment's)
        nmat = sub.prob_matrices
        pi = np.zeros(sub.rate_cats, dtype=np.uint32)
        brl = np.ascontiguousarray(W.branch_lengths(nmat))
        assert lib.pll_update_prob_matrices(s.p, api.uptr(pi), api.uptr(np.arange(nmat, dtype=np.uint32)), api.dptr(brl), nmat)
        s.update_partials()
        e = sub.edges[0]
        st = s.new_sumtable()
        s.update_sumtable(e, st)
        t = 0.3
        for it in range(8):  # Newton-Raphson on the root branch (examples/newton/newton.c:55-80), sharded
            d1, d2 = C.c_double(), C.c_double()
            ok = lib.pll_gpu_group_likelihood_derivatives(s.p, g, e[1], e[3], t, api.uptr(pi), api.dptr(st), C.byref(d1), C.byref(d2))
            assert ok, lib.errmsg()
            trace.append((float.hex(t), float.hex(d1.value), float.hex(d2.value)))
            t = max(1e-6, t - d1.value / d2.value) if d2.value > 0 else t * 0.5
        # a failing evaluation on the last rank only: every rank is told, the failing one keeps its own error
        t_bad = -1.0 if rank == world - 1 and world > 1 else t  # a negative branch length is refused before anything is launched
        ok = lib.pll_gpu_group_likelihood_derivatives(s.p, g, e[1], e[3], t_bad, api.uptr(pi), api.dptr(st), C.byref(d1), C.byref(d2))
        failed = (bool(ok), lib.errno(), lib.errmsg() if not ok else "")
    if g:
        lib.pll_gpu_group_leave(g)
    print(json.dumps(dict(rank=rank, trace=trace, t=float.hex(t), sites=sub.sites, failed=failed)))


if __name__ == "__main__":
    main()
