"""GPU parity at the exact schedules and sizes both halves of the metric run on ("Next round" item 1):

(i)   the filter schedule bench.py times -- N = 65 536, m = 512, block-lower covariance storage, lazy_depth 4, `inplace = 0`
      (two banks fit: ping-pong banks + shared flush, `launch_share_plan`) -- properties, determinism, and equality with the
      single-bank schedule (`inplace = 1`) on the same Philox streams (particleFilter.m:100-218);
(ii)  the metric's smoother configuration -- particleSmootherInformationForm, m = 512, block-lower `P`, packed `Imat`, lazy_depth 3,
      fresh factorisation and carried factors -- against the plain-C restatement at N_P = 512, T = 40, N_K = 2
      (particleSmootherInformationForm.m:98-362; ancestor weights :186-335);
(iii) the filter over the metric's full horizon T = 3000: moved to tests/test_gpu_r05_parity.py, where an extended-precision arbiter
      replaces the r04 comparison of two fp64 evaluations;
(iv)  the largest single-GPU smoother size, N_P = 32 768 at m = 512: properties over a short run.

Tolerances: ancestor / resampling indices bit-exact, fp64 quantities 1e-9 relative (north_star)."""
import importlib

import numpy as np
import pytest

import cases
import oracle_c
from test_gpu_configs import check_filter_properties, mag_inputs, rel, run_session

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def test_bench_schedule_at_configs2_size_two_banks_shared_flush(rbpf):
    """(i) N = 65 536, fp64sym, lazy_depth 4, inplace = 0 -> two 78 GB banks, shared flush: properties, two runs bit-identical,
    and the single-bank schedule on the same streams gives the same resampling indices and the same outputs to rounding."""
    N, steps = 65536, 10
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 512)
    want = ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "trace_ai", "xl_mean")
    a = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=0, storage="fp64sym")
    check_filter_properties(a, N, steps, P0)
    a2 = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=0, storage="fp64sym")
    for k in want:
        np.testing.assert_array_equal(a[k], a2[k], err_msg=k)
    del a2
    b = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=1, storage="fp64sym")
    np.testing.assert_array_equal(a["trace_ai"], b["trace_ai"])
    for k in want:
        if k != "trace_ai":
            sl = (slice(None), slice(0, steps)) if k in ("traj_max", "traj_mean", "trace_w") else Ellipsis
            assert rel(a[k][sl], b[k][sl]) <= 1e-11, k


def test_bench_schedule_really_uses_two_banks(rbpf):
    """The comparison above only means something if `inplace = 0` picks the two-bank schedule at this size on this card: the
    session reports its choice."""
    d, mdl, x0, P0, R = mag_inputs(rbpf, 12, 512)
    with rbpf.FilterSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, 65536, 0.01, rng=rbpf.PhiloxRNG(5),
                            lazy_depth=4, inplace=0, storage="fp64sym") as s:
        assert s.schedule() == (2, True)                                         # two banks, shared flush


@pytest.fixture(scope="module")
def c_smoother_m512(rbpf, tmp_path_factory):
    import bench
    c = cases.mag_case(512, 40, 512, seed=83, N_K=2)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    lib = oracle_c.build(native_dir=str(tmp_path_factory.mktemp("oracle_native_sm512")))
    ref, _ = oracle_c.particle_smoother(rbpf, mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], 2, c["dt"],
                                        cases.device_rng(rbpf, c), True, n_threads=bench.usable_cores(), lib_path=lib)
    return c, ref


@pytest.mark.parametrize("chol_refresh", [1, 32])
def test_metric_smoother_configuration_against_the_c_restatement(rbpf, c_smoother_m512, chol_refresh):
    """(ii) N_P = 512, T = 40, N_K = 2 at nLin = 515: block-lower P, packed Imat, lazy_depth 3 (the quad mapping and the shared-flush
    readers of step_sym_kernel<3, 3, ., 1, 8>), chol_solve64 every step (chol_refresh 1) and the carried factors with refreshes at
    t = 1 and t = 33 (chol_refresh 32): every ancestor, every trajectory draw, weights / ancestor probabilities / outputs 1e-9
    (2e-9 absolute on paNt with the carried factors: their stated tolerance)."""
    c, ref = c_smoother_m512
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    XNK, XLK, PK, ex = rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"],
                                                            c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], 2, c["dt"],
                                                            rng=cases.device_rng(rbpf, c), extras=True, storage="fp64sym",
                                                            lazy_depth=3, chol_refresh=chol_refresh)
    np.testing.assert_array_equal(ex["ak"], ref["ak"])
    np.testing.assert_array_equal(ex["ai"][:, 1:], ref["ai"][:, 1:])
    assert rel(ex["w"], ref["w"]) <= RTOL
    a, b = ex["paNt"][1, 1:], ref["paNt"][1, 1:]
    assert np.max(np.abs(a - b)) <= (2e-9 if chol_refresh > 1 else RTOL)
    assert rel(XNK, ref["XNK"]) <= RTOL and rel(XLK, ref["XLK"]) <= RTOL and rel(PK, ref["PK"]) <= RTOL


def test_smoother_at_the_largest_single_gpu_size(rbpf):
    """(iv) particleSmootherInformationForm at N_P = 32 768, m = 512 (6.9 MB of state per particle: block-lower P, packed Imat),
    lazy_depth 3, fresh factorisation, T = 6, N_K = 2: finite outputs, normalised weights and ancestor probabilities
    (particleSmootherInformationForm.m:243-245), ancestors in range, the second iteration conditioned on the first's draw."""
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    N, T, N_K = 32768, 6, 2
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = rbpf.dense_mag_prior(512, d["LL"], cases.THETA_MAG)
    XNK, XLK, PK, ex = rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],
                                                            x0, P0, cases.Q_MAG, R, N, N_K, 0.01, rng=rbpf.PhiloxRNG(9), extras=True,
                                                            storage="fp64sym", lazy_depth=3)
    assert np.all(np.isfinite(XNK)) and np.all(np.isfinite(XLK)) and np.all(np.isfinite(PK))
    np.testing.assert_allclose(ex["w"].sum(axis=2), 1.0, rtol=0, atol=1e-12)
    assert ex["ai"][:, 1:].min() >= 0 and ex["ai"][:, 1:].max() < N
    pa = ex["paNt"][1, 1:]
    np.testing.assert_allclose(pa.sum(axis=1), 1.0, rtol=0, atol=1e-12)
    assert np.all(ex["ai"][1, 1:, N - 1] >= 0)                                   # slot N: the reference trajectory's sampled ancestors
    for k in range(N_K):
        P = PK[:, :, k]
        assert rel(P, P.T) < 1e-12 and np.linalg.eigvalsh(0.5 * (P + P.T)).min() > 0
        assert np.all(np.diag(P) <= np.diag(P0) * (1 + 1e-12))
