"""CPU: the derivative restatement (orc_update_sumtable / orc_likelihood_derivatives) against the
golden vectors produced by the reference, including the values pinned in test/out/derivatives.out."""
import numpy as np
import pytest

from conftest import DERIV_GOLDEN
from deriv_common import assert_sumtable, close, load
from oracle import oracle_deriv as OD


@pytest.mark.parametrize("path", DERIV_GOLDEN, ids=lambda p: p.split("/")[-1][:-4])
def test_oracle_derivatives_reproduce_golden(path):
    case, eig, rates, edges, brlens, exp_d, exp_st, extra = load(path)
    for i, (edge, after) in enumerate(edges):
        sub = type(case)(**{**case.__dict__, "op_batches": case.op_batches[:after + 1]})
        got = OD.run_derivatives(sub, eig, rates, [edge], brlens)
        assert_sumtable(got["sumtable"][0], exp_st[i], case.name)
        for (g1, g2), (e1, e2) in zip(got["d"][0], exp_d[i]):
            assert close(g1, e1, sites=case.sites) and close(g2, e2, sites=case.sites), (case.name, g1, e1, g2, e2)
        if "kat" in extra:  # printed with %12.4e in the reference's own expected output
            for (g1, g2), (_, p1, p2) in zip(got["d"][0], extra["kat"][i]):
                assert abs(g1 - p1) <= 6e-5 * abs(p1) + 1e-13 and abs(g2 - p2) <= 6e-5 * abs(p2) + 1e-13


def test_derivative_inventory():
    names = [p.split("/")[-1] for p in DERIV_GOLDEN]
    assert len(names) >= 16 and any(n.startswith("kat_deriv") for n in names)
