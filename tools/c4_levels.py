import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "libpll-2_amd"))
from pllamd import api, driver, workload as W
lib = api.PllLib()
case = W.make_case("c4", 4, 128, 125000, attributes=api.SITE_REPEATS, seed=1000)
with driver.Session(lib, case, api.ARCH_AVX2) as s:
    s.update_partials(update_repeats=1)
    ops = case.op_batches[0]
    ent = {op[0]: s.entries(op[0]) for op in ops}
    depth = {t: 0 for t in range(case.tips)}
    rows = []
    for op in ops:
        depth[op[0]] = 1 + max(depth[op[2]], depth[op[5]])
    import collections
    by = collections.defaultdict(list)
    for op in ops:
        by[depth[op[0]]].append(ent[op[0]])
    for d in sorted(by):
        v = by[d]
        print("level", d, "ops", len(v), "entries min/max", min(v), max(v), "uncompressed", sum(1 for x in v if x == 125000))
