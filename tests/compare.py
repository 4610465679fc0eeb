"""shared comparison helpers for the parity tests"""
import numpy as np

from pllamd import driver

# north_star tolerance: per-site CLVs and log-likelihoods within 1e-10 relative of the reference.
# A per-site log-likelihood is compared with |d| <= RTOL * max(|lnL|, 1): a relative error eps on
# the site LIKELIHOOD is an absolute error eps on its logarithm (all-gap sites have lnL == 0).
RTOL = 1e-10


def assert_results_match(got, exp, rtol=RTOL, what=""):
    assert set(exp["clv"]) <= set(got["clv"]), f"{what}: missing CLVs"
    for k, e in exp["clv"].items():
        g = got["clv"][k]
        assert g.shape == e.shape, f"{what}: clv {k} shape {g.shape} != {e.shape}"
        err = driver.rel_err_normalised(g, got["scaler"].get(k), e, exp["scaler"].get(k))
        assert err <= rtol, f"{what}: clv {k} rel err {err:.3e}"
    for i, (g, e) in enumerate(zip(got["lnl"], exp["lnl"])):
        assert np.isfinite(g), f"{what}: lnl {g}"
        assert abs(g - e) <= rtol * abs(e), f"{what}: edge lnl {g!r} vs {e!r}"
    for g, e in zip(got["persite"], exp["persite"]):
        assert np.all(np.abs(g - e) <= rtol * np.maximum(np.abs(e), 1.0)), \
            f"{what}: persite max abs diff {np.max(np.abs(g - e)):.3e}"
    for g, e in zip(got["root_lnl"], exp["root_lnl"]):
        assert abs(g - e) <= rtol * abs(e), f"{what}: root lnl {g!r} vs {e!r}"
    for g, e in zip(got["root_persite"], exp["root_persite"]):
        assert np.all(np.abs(g - e) <= rtol * np.maximum(np.abs(e), 1.0)), what
    assert len(got["lnl"]) == len(exp["lnl"]) and len(got["root_lnl"]) == len(exp["root_lnl"])


def scalers_equal(got, exp):
    return all((got["scaler"][k] == v).all() for k, v in exp["scaler"].items())


def assert_kat(got, extra, what=""):
    """values pinned by the reference's own test outputs (printed with %.6f / %.7f)"""
    if "kat_lnl" in extra:
        assert abs(got["lnl"][0] - extra["kat_lnl"]) < 5.1e-7, what
        assert np.allclose(got["persite"][0], extra["kat_persite"], atol=5.1e-8, rtol=0), what
    if "kat_root_lnl" in extra:
        assert abs(got["root_lnl"][0] - extra["kat_root_lnl"]) < 5.1e-7, what
        assert np.allclose(got["root_persite"][0], extra["kat_root_persite"], atol=5.1e-8, rtol=0), what
