import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT,"libpll-2_amd")); sys.path.insert(0, ROOT)
import numpy as np
from pllamd import api, driver, workload as W
lib = api.PllLib()
for scalers in (True, False):
    case = W.make_case("c4", 4, 128, 125000, attributes=api.SITE_REPEATS, mutate_pct=4, seed=1000, scalers=scalers)
    ops = api.make_ops(case.op_batches[0]); n=len(case.op_batches[0])
    with driver.Session(lib, case, api.ARCH_AVX2) as s:
        lib.pll_update_partials(s.p, ops, n)
        lib.pll_gpu_synchronize(s.p)
        lib.pll_gpu_timer_start(s.p)
        for _ in range(20): lib.pll_update_partials_rep(s.p, ops, n, 0)
        ms = lib.pll_gpu_timer_stop(s.p)/20
        rep = s.part.repeats.contents
        ent = [rep.pernode_ids[i] or 125000 for i in range(case.tips, s.part.nodes)]
        print("scalers", scalers, "traversal ms", round(ms,4), "launches", lib.pll_gpu_last_launch_count(s.p), "entries total", sum(ent), "bytes MB", round(lib.pll_gpu_last_algorithmic_bytes(s.p)/1e6,1))
        print(sorted(ent)[-40:])
