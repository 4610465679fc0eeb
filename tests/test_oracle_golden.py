"""CPU: the restatement (oracle/pll_oracle.c) against every golden vector produced by the
reference and against the values pinned in the reference's own test outputs."""
import pytest

from conftest import GOLDEN, golden_ids
from compare import assert_kat, assert_results_match, scalers_equal
from oracle import oracle as O
from pllamd import fixtures


@pytest.mark.parametrize("path", GOLDEN, ids=golden_ids())
def test_oracle_reproduces_golden(path):
    case, exp, extra = fixtures.load(path)
    got = O.run_case(case)
    assert_results_match(got, exp, rtol=1e-12, what=case.name)
    assert scalers_equal(got, exp), "scaler vectors differ from the reference"
    assert_kat(got, extra, case.name)


@pytest.mark.parametrize("path", [g for g in GOLDEN if "s5_" in g or "s7_" in g or "s61_plain" in g], ids=lambda p: p[-14:-4])
def test_oracle_padded_stride(path):
    """same numbers when the arrays carry AVX-style padding (states_padded = 4*ceil(s/4))"""
    case, exp, _ = fixtures.load(path)
    got = O.run_case(case, states_padded=(case.states + 3) & ~3)
    assert_results_match(got, exp, rtol=1e-12, what=case.name)


def test_golden_inventory():
    names = golden_ids()
    assert len(names) >= 40
    for must in ("kat_00010_plain", "kat_00011_tip", "kat_00012_plain", "kat_00020_plain", "dna_deep_rate",
                 "aa_deep_rate_repeats", "s61_plain"):
        assert must in names
