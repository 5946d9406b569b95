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


def test_arbiter_build_brackets_the_fp64_restatements(rbpf):
    """The extended-precision build of the C restatement (-DRBPF_ORACLE_LONG_DOUBLE: the arbiter of tests/test_gpu_r05_parity.py) runs the
    same statements: same indices as the fp64 build and the numpy oracle, results within fp64 rounding of both -- and at the r04 "near
    tie" (dense-radio m = 512, t = 3) it is the fp64 C build's arg-max it confirms, not numpy's (test infrastructure only, CPU)."""
    lib = oracle_c.build_arbiter()
    c = cases.mag_case(6, 8, 130, seed=21)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    args = (rbpf, mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"], cases.device_rng(rbpf, c))
    arb, _ = oracle_c.particle_filter(*args, n_threads=2, lib_path=lib)
    c64, _ = oracle_c.particle_filter(*args, n_threads=2)
    ref = cases.oracle_filter(c)
    np.testing.assert_array_equal(arb["trace_ai"], c64["trace_ai"])
    np.testing.assert_array_equal(arb["trace_ai"].T[1:], ref["trace"]["ai"][1:])
    for k in ("trace_w", "traj_mean", "xl_max", "P_max", "final_xl"):
        assert rel(c64[k], arb[k]) < 1e-11, k
    assert rel(ref["xl_max"], arb["xl_max"]) < 1e-11 and rel(ref["trace"]["w"], arb["trace_w"].T) < 1e-11
    c = cases.radio_case(6, 7, 512, seed=41, N_K=2)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    args = (rbpf, mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"], cases.device_rng(rbpf, c))
    arb, _ = oracle_c.particle_filter(*args, n_threads=2, lib_path=lib)
    c64, _ = oracle_c.particle_filter(*args, n_threads=2)
    tr = cases.oracle_filter(c)["trace"]
    np.testing.assert_array_equal(np.argmax(arb["trace_w"], axis=0), np.argmax(c64["trace_w"], axis=0))
    assert int(np.argmax(arb["trace_w"][:, 3])) == 3 and int(np.argmax(tr["w"][3])) == 0


def test_arbiter_fixtures_are_complete():
    """The committed arbiter fixtures carry what the GPU tests read: the measurements, the inputs' checksum, the arbiter's results and
    the fp64 restatement's distance from them."""
    import os
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    need = {"arbiter_filter_N64_T3000_m512.npz": ("y", "inputs_sha256", "trace_ai", "trace_w", "xl_max", "err_c64_nofma__trace_w", "err_c64_fma__trace_w"),
            "arbiter_filter_N64_T3000_m256.npz": ("y", "inputs_sha256", "trace_ai", "trace_w", "trace_w_stride", "err_c64_nofma__trace_w"),
            "arbiter_filter_N32_T1000_m1024.npz": ("y", "inputs_sha256", "trace_ai", "trace_w", "trace_w_stride", "err_c64_nofma__trace_w"),
            "arbiter_info_smoother_N64_T1000_m512.npz": ("y", "inputs_sha256", "ai", "ak", "w", "paNt", "XNK", "PK_rows", "err_c64_nofma__paNt"),
            "arbiter_radio_smoothers_N64_T200_m128.npz": ("y", "inputs_sha256", "info__ai", "cov__ai", "info__paNt", "cov__w", "info__err_c64_nofma__w")}
    for name, keys in need.items():
        fx = np.load(os.path.join(golden, name))
        for k in keys:
            assert k in fx.files, (name, k)
        assert np.all(np.isfinite(fx["y"]))
    fx = np.load(os.path.join(golden, "arbiter_filter_N64_T3000_m512.npz"))
    assert 1e-9 < float(fx["err_c64_nofma__trace_w"]) < 1e-7          # the fp64 restatement itself is NOT within 1e-9 over this horizon
