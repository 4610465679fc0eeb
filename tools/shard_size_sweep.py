#!/usr/bin/env python3
"""How a C4 shard's step time depends on its size: the FIRST shard of the pattern-sorted 1M-site alignment (the
entry-poor end) cut at growing site counts, and the entry-rich second eighth cut at shrinking ones.
    python tools/shard_size_sweep.py  -> lines of (first site, sites, entries at the 8-tip level, ms per step)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
import bench  # noqa: E402
from pllamd import api, driver, sharding  # noqa: E402


class H:
    dist = None
    world = 1
    rank = 0
    on_device = False


def main():
    cfg = bench.CONFIGS["c4"]
    lib = api.PllLib()
    full = sharding.sort_columns(lib, bench.build_case(cfg, cfg["sites"], api.SITE_REPEATS))
    n = full.sites
    for lo, hi in [(0, 112000), (0, 118016), (0, 124992), (0, 128000), (0, 131008), (0, 134016), (0, 140032),
                   (124992, 124992 + 112000), (124992, 124992 + 118016), (124992, 249984), (124992, 124992 + 131008)]:
        case = sharding.shard_case(full, 0, 1, [(lo, hi)])
        r = bench.Runner(H, lib, api, driver, case, True, reduce=None)
        blocks, lnl = r.timed(5, 20, 5)
        lv = r.level_entries()
        r.close()
        print(json.dumps(dict(first=lo, sites=hi - lo, entries_level3=lv[2], ms=round(bench.block_stats(blocks, 20)[0], 4))), flush=True)


if __name__ == "__main__":
    main()
