"""GPU parity: HIP particleSmoother / particleSmootherInformationForm vs the numpy oracle."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def run_both(rbpf, c, info_form):
    ref = cases.oracle_smoother(c, info_form)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    f = rbpf.particleSmootherInformationForm if info_form else rbpf.particleSmoother
    out = f(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
            c["N_P"], c["N_K"], c["dt"], rng=cases.device_rng(rbpf, c), extras=True)
    return ref, out


def check(ref, out, N_K):
    XNK, XLK, PK, ex = out
    tr = ref["trace"]
    np.testing.assert_array_equal(ex["ak"], tr["ak"])
    np.testing.assert_array_equal(ex["ai"][:, 1:], tr["ai"][:, 1:])          # bit-exact ancestors incl. slot N_P
    assert rel(ex["w"], tr["w"]) <= RTOL
    for k in range(1, N_K):                                                  # ancestor probabilities AI(:,t)
        a, b = ex["paNt"][k, 1:], tr["paNt"][k, 1:]
        assert np.max(np.abs(a - b)) <= RTOL * max(1.0, np.max(np.abs(b)))
    assert rel(XNK, ref["XNK"]) <= RTOL
    assert rel(XLK, ref["XLK"]) <= RTOL
    assert rel(PK, ref["PK"]) <= RTOL


@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 8, 6, 16), ("mag", 6, 5, 130), ("radio", 10, 8, 32),
                                            ("radio", 9, 7, 128)])
def test_covariance_form_smoother_matches_oracle(rbpf, kind, N_P, N_T, m):
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=7, N_K=3)
    ref, out = run_both(rbpf, c, info_form=False)
    check(ref, out, 3)


@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 8, 6, 16), ("mag", 6, 5, 130), ("radio", 10, 8, 32),
                                            ("radio", 9, 7, 128)])
def test_information_form_smoother_matches_oracle(rbpf, kind, N_P, N_T, m):
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=9, N_K=3)
    ref, out = run_both(rbpf, c, info_form=True)
    check(ref, out, 3)


def test_information_form_smoother_above_8192_particles(rbpf):
    """N_P > 8192: weights and ancestor probabilities are normalised by the multi-workgroup pipeline (parallel prefix,
    certified search, exact fallback) instead of the single-workgroup kernels -- same indices as the strict
    left-to-right cumsum of tools/sample.m:30."""
    c = cases.radio_case(8300, 3, 8, seed=13, N_K=2)
    ref, out = run_both(rbpf, c, info_form=True)
    check(ref, out, 2)
