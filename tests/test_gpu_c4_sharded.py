"""GPU: BASELINE configs[3] as stated - 4-state DNA, 128 taxa, 1M sites, PLL_ATTRIB_SITE_REPEATS, sites
sharded eight ways - on ONE device: the eight shards of the pattern-sorted alignment run one after another
and the sum of their log-likelihoods must equal the reference's value for the whole alignment
(src/core_likelihood.c:1489 is the only cross-site operation), as must the un-sharded run; where the
prebuilt reference is present its class counts per node are compared too (integers: exact)."""
import json
import os
import sys

import numpy as np
import pytest

from compare import RTOL
from oracle import oracle as O
from pllamd import api, driver, sharding

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def c4():
    import bench
    cfg = bench.CONFIGS["c4"]
    return bench.build_case(cfg, cfg["sites"], api.SITE_REPEATS)


def _lnl(lib, case, ids=False):
    with driver.Session(lib, case, api.ARCH_AVX2) as s:
        s.update_partials()
        v, _ = s.edge_lnl(case.edges[0], persite=False)
        counts = [s.entries(c) for c in range(case.tips, case.tips + case.clv_buffers)] if ids else None
    return v, counts


def test_eight_shards_sum_to_the_reference_value(amd_lib, c4):
    pin = json.load(open(os.path.join(ROOT, "tests", "golden", "section8d_lnl.json")))["c4"]
    ordered = sharding.sort_columns(amd_lib, c4)
    assert int(np.asarray(ordered.pattern_weights, dtype=np.uint64).sum()) == c4.sites
    parts, entries = [], 0
    for r in range(8):
        sub = sharding.shard_case(ordered, r, 8)
        v, counts = _lnl(amd_lib, sub, ids=True)
        assert np.isfinite(v)
        parts.append(v)
        entries += sum(counts)
    total = float(np.sum(parts))
    assert abs(total - pin) <= RTOL * abs(pin), (total, pin)
    whole, wcounts = _lnl(amd_lib, ordered, ids=True)
    assert abs(whole - pin) <= RTOL * abs(pin), (whole, pin)
    assert abs(total - whole) <= 1e-12 * abs(whole)
    # a shard cannot see repeats across its borders: the shards together compute at least as many entries as
    # the whole alignment, and (sorted columns) not many more
    assert sum(wcounts) <= entries <= 1.15 * sum(wcounts), (entries, sum(wcounts))


def test_whole_alignment_class_counts_against_reference(amd_lib, c4):
    if not os.path.exists(O.REF_LIB):
        pytest.skip("oracle/_ref/libpll_ref.so not shipped")
    ref = api.PllLib(O.REF_LIB)
    r_lnl, r_ids = _lnl(ref, c4, ids=True)
    g_lnl, g_ids = _lnl(amd_lib, c4, ids=True)
    assert g_ids == r_ids
    assert abs(g_lnl - r_lnl) <= RTOL * abs(r_lnl)
