"""Oracle (oracle/lba_oracle.c) against the committed golden vectors.

The goldens come from tests/golden/make_golden.py: an independent dense numpy LM that
shares no code with the oracle (full normal equations, matrix exponential).  The
reference itself pins nothing for this path: parity unpinned."""
import numpy as np
import pytest

from conftest import load_golden, quat_angle

CASES = ["lba_tiny", "lba_small", "lba_hard", "lba_norobust", "lba_stereo", "lba_cameras"]     # lba_cameras: a camera per keyframe


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_golden(oracle_mod, name):
    w, g = load_golden(name)
    r = oracle_mod.solve(w)
    assert r["status"] == 0
    assert r["iters_done"] == int(g["iters"])
    assert np.array_equal(r["trace"]["accept"], g["tr_accept"])
    np.testing.assert_allclose(r["trace"]["lam"], g["tr_lam"], rtol=1e-9)
    np.testing.assert_allclose(r["trace"]["f1"], g["tr_f1"], rtol=1e-9)
    assert quat_angle(r["poses"][:, :4], g["poses"][:, :4]).max() < 1e-9
    np.testing.assert_allclose(r["poses"][:, 4:], g["poses"][:, 4:], atol=1e-9)
    np.testing.assert_allclose(r["points"], g["points"], atol=1e-8)
    np.testing.assert_allclose(r["chi2"], g["chi2"], rtol=1e-7, atol=1e-8)
    assert np.array_equal(r["outlier"], g["outlier"])


def test_hard_case_has_rejected_trials():
    _, g = load_golden("lba_hard")
    assert (g["tr_accept"] == 0).sum() >= 3      # the fixture must exercise pop() / lambda growth


@pytest.mark.parametrize("name", ["lba_small", "lba_hard", "lba_stereo", "lba_cameras"])
def test_openmp_build_of_the_oracle_agrees_with_the_serial_one(oracle_mod, name):
    """bench.py times the OpenMP build for context only; its sums run in a different order, nothing else differs."""
    w, _ = load_golden(name)
    a, b = oracle_mod.solve(w), oracle_mod.solve(w, omp=True)
    assert np.array_equal(a["trace"]["accept"], b["trace"]["accept"]) and a["n_solves"] == b["n_solves"]
    np.testing.assert_allclose(b["poses"], a["poses"], atol=1e-10)
    np.testing.assert_allclose(b["points"], a["points"], atol=1e-9)
    assert np.array_equal(a["outlier"], b["outlier"])
