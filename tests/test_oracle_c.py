"""The plain-C restatement (CPU baseline) against the numpy oracle (both test infrastructure)."""
import numpy as np
import pytest

import cases
import oracle_c


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 8, 6, 16), ("mag", 6, 5, 130), ("radio", 10, 9, 32)])
def test_c_restatement_matches_numpy_oracle(rbpf, kind, N_P, N_T, m):
    c = cases.mag_case(N_P, N_T, m, seed=11) if kind == "mag" else cases.radio_case(N_P, N_T, m, seed=11)
    ref = cases.oracle_filter(c)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    out, secs = oracle_c.particle_filter(rbpf, mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                                         c["N_P"], c["dt"], cases.device_rng(rbpf, c), n_threads=2)
    assert secs > 0
    tr = ref["trace"]
    np.testing.assert_array_equal(out["trace_ai"].T[1:], tr["ai"][1:])
    assert int(out["iw_max"][0]) == ref["iw_max"]
    assert np.max(np.abs(out["trace_logw"].T - tr["logw"])) < 1e-9 * max(1, np.max(np.abs(tr["logw"])))
    for k in ("traj_max", "traj_mean", "xl_max", "xl_mean", "P_max", "P_mean", "xn_traj"):
        assert rel(out[k], ref[k]) < 1e-10, k
    assert rel(out["traj_sample_iwmax"], ref["traj_sample_iwmax"]) < 1e-10
    assert rel(out["final_P"], tr["P"]) < 1e-10
    assert rel(out["final_xl"], tr["xl"]) < 1e-10


@pytest.mark.parametrize("info_form", [False, True])
@pytest.mark.parametrize("kind,N_P,N_T,m,drn", [("mag", 7, 6, 16, True), ("mag", 5, 5, 130, True), ("radio", 9, 8, 32, True),
                                                ("radio", 8, 6, 128, True)])
def test_c_smoothers_match_numpy_oracle(rbpf, kind, N_P, N_T, m, drn, info_form):
    """Two independently structured restatements of src/particleSmoother.m / src/particleSmootherInformationForm.m (numpy,
    vectorised over nothing; plain C with explicit *_pred banks) agree on every ancestor index and to 1e-9 on weights,
    ancestor probabilities and outputs."""
    c = (cases.mag_case if kind == "mag" else cases.radio_case)(N_P, N_T, m, seed=15, N_K=3)
    ref = cases.oracle_smoother(c, info_form, use_dynResNorm=drn)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    out, secs = oracle_c.particle_smoother(rbpf, mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, N_P, 3, c["dt"],
                                           cases.device_rng(rbpf, c), info_form, use_dyn_res_norm=drn, n_threads=2)
    tr = ref["trace"]
    np.testing.assert_array_equal(out["ak"], tr["ak"])
    np.testing.assert_array_equal(out["ai"][:, 1:], tr["ai"][:, 1:])
    assert rel(out["w"], tr["w"]) < 1e-9
    for k in range(1, 3):
        a, b = out["paNt"][k, 1:], tr["paNt"][k, 1:]
        assert np.max(np.abs(a - b)) <= 1e-9 * max(1.0, np.max(np.abs(b)))
    for key in ("XNK", "XLK", "PK"):
        assert rel(out[key], ref[key]) < 1e-9, key
