"""The C-ABI library loads on a CPU-only machine, exports every symbol include/rbpf.h declares, and
the product path fails loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rbpf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rbpf_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_match_export_list(rbpf):
    assert declared_functions() == sorted(rbpf.EXPORTS)


def test_library_exports_every_declared_symbol(rbpf):
    lib = rbpf.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), name
    import ctypes as C
    ffi = __import__("importlib").import_module(rbpf.__name__ + "._ffi")
    src = open(os.path.join(ROOT, "include", "rbpf.h")).read()
    assert lib.rbpf_abi_version() == ffi.ABI_VERSION == int(re.search(r"#define RBPF_ABI_VERSION (\d+)", src).group(1))
    for which, mirror in enumerate(ffi.ABI_STRUCTS):                       # the ctypes mirror against the library's own sizeof
        assert lib.rbpf_abi_sizeof(which) == C.sizeof(mirror), mirror.__name__
    assert lib.rbpf_abi_sizeof(99) == -1
    assert lib.rbpf_status_string(0) == b"ok"
    assert b"positive definite" in lib.rbpf_status_string(rbpf.RBPF_ERR_CHOL_FAILED)


def test_options_of_another_layout_are_refused(rbpf):
    """rbpf_options.struct_size (ABI 9): an entry point handed options of another size refuses them BEFORE reading a field -- a
    binding compiled against an older header cannot make the library read past its struct (no device needed: the check comes first)."""
    import ctypes as C
    ffi = __import__("importlib").import_module(rbpf.__name__ + "._ffi")
    host = __import__("importlib").import_module(rbpf.__name__ + ".host")
    lib = rbpf.load_library()
    c = cases.radio_case(4, 3, 8, seed=1)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    prob = host._Problem(mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"])
    blk, _keep = host._rng_block(cases.device_rng(rbpf, c), prob.N_P, prob.N_T, mdl.nw, 1)
    mdesc = mdl.descriptor()
    good = ffi.rbpf_options(keep_history=1)
    assert good.struct_size == C.sizeof(ffi.rbpf_options) == lib.rbpf_abi_sizeof(3)
    nbytes = C.c_size_t(0)
    assert lib.rbpf_filter_workspace_bytes(C.byref(mdesc), C.byref(prob.c), C.byref(good), C.byref(nbytes)) == rbpf.RBPF_OK
    stale = ffi.rbpf_options(keep_history=1)
    stale.struct_size = C.sizeof(ffi.rbpf_options) - 8
    ctx = C.c_void_p()
    out = ffi.rbpf_filter_out()
    sm = ffi.rbpf_smoother_out()
    for call in (lambda: lib.rbpf_filter_workspace_bytes(C.byref(mdesc), C.byref(prob.c), C.byref(stale), C.byref(nbytes)),
                 lambda: lib.rbpf_filter_create(C.byref(mdesc), C.byref(prob.c), C.byref(blk), C.byref(stale), C.byref(ctx)),
                 lambda: lib.rbpf_particle_filter(C.byref(mdesc), C.byref(prob.c), C.byref(blk), C.byref(stale), C.byref(out)),
                 lambda: lib.rbpf_particle_smoother(C.byref(mdesc), C.byref(prob.c), C.byref(blk), C.byref(stale), 1, 1, C.byref(sm))):
        assert call() == rbpf.RBPF_ERR_INVALID_ARG
        assert b"struct_size" in lib.rbpf_last_error()


def test_workspace_bytes_follow_the_storage_option(rbpf):
    """rbpf_filter_workspace_bytes (no device needed) at BASELINE.json configs[4]'s basis size: the lower block triangle in fp32
    (storage 3) needs less than the fp32 full square (1), which needs less than the fp64 lower block triangle (2) ... than the fp64
    full square (0); the block-lower figures include the sixteen-tile-row kernel's column-strip workspace (118 KB per particle)."""
    import ctypes as C
    ffi = __import__("importlib").import_module(rbpf.__name__ + "._ffi")
    host = __import__("importlib").import_module(rbpf.__name__ + ".host")
    lib = rbpf.load_library()
    c = cases.mag_case(64, 4, 1024, seed=1)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    prob = host._Problem(mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"])
    mdesc = mdl.descriptor()
    need = {}
    for storage in (0, 1, 2, 3):
        nbytes = C.c_size_t(0)
        opt = ffi.rbpf_options(keep_history=1, storage=storage, lazy_depth=2)
        assert lib.rbpf_filter_workspace_bytes(C.byref(mdesc), C.byref(prob.c), C.byref(opt), C.byref(nbytes)) == rbpf.RBPF_OK
        need[storage] = nbytes.value
    assert need[3] < need[1] < need[2] < need[0]
    n, N = 1027, 64
    assert need[0] - need[2] > 0.8 * N * n * n * 8                           # two banks x (n^2 - 0.53 n^2 - strips)
    assert abs((need[2] - need[3]) - 2 * N * (136 * 4096 + 3 * 1028) * 4) <= 0.02 * need[2]   # fp32 tiles: half of two block-lower banks


def test_chol_refresh_resolution(rbpf):
    """rbpf_options.chol_refresh = 0 is automatic: carried factors (K = 32) for the recognised dense families from nLin = 128 on,
    the from-scratch factorisation elsewhere; explicit values are kept (rbpf_chol_refresh_resolve, no device needed)."""
    lib = rbpf.load_library()
    MAG, RADIO, SPARSE, GENERIC = 1, 2, 3, 4
    assert lib.rbpf_chol_refresh_resolve(MAG, 515, 3, 0) == 32 and lib.rbpf_chol_refresh_resolve(MAG, 259, 3, 0) == 32
    assert lib.rbpf_chol_refresh_resolve(RADIO, 128, 1, 0) == 32
    assert lib.rbpf_chol_refresh_resolve(MAG, 19, 3, 0) == 1 and lib.rbpf_chol_refresh_resolve(RADIO, 127, 1, 0) == 1
    assert lib.rbpf_chol_refresh_resolve(MAG, 1027, 3, 0) == 1                    # the sweep holds nLin <= 575
    assert lib.rbpf_chol_refresh_resolve(GENERIC, 515, 3, 0) == 1 and lib.rbpf_chol_refresh_resolve(SPARSE, 200, 3, 0) == 1
    assert lib.rbpf_chol_refresh_resolve(MAG, 515, 3, 1) == 1 and lib.rbpf_chol_refresh_resolve(MAG, 515, 3, 16) == 16
    assert lib.rbpf_chol_refresh_resolve(MAG, 19, 3, 4) == 4


def test_no_cpu_fallback_without_a_device(rbpf):
    if rbpf.device_count() > 0:
        pytest.skip("a GPU is visible")
    c = cases.radio_case(4, 3, 8, seed=1)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    with pytest.raises(rbpf.RBPFError) as ei:
        rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                            c["N_P"], c["dt"], rng=cases.device_rng(rbpf, c))
    assert ei.value.status == rbpf.RBPF_ERR_NO_DEVICE
    with pytest.raises(rbpf.RBPFError):
        rbpf.sample(np.ones(4) / 4, [0.3])


def test_arbitrary_callables_are_not_emulated_on_the_cpu(rbpf):
    """Unrecognised handles take the generic family (host evaluates the handles, device does the rest of the step):
    without a device that is an error like every other entry point, never a CPU emulation."""
    if rbpf.device_count() > 0:
        pytest.skip("a GPU is visible: covered by test_generic_host_callback_path_matches_oracle")
    c = cases.radio_case(4, 3, 8, seed=1)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    with pytest.raises(rbpf.RBPFError) as ei:
        rbpf.particleFilter(lambda xn, dx, dt, Q: xn, lambda xn: np.zeros((xn.shape[1], 8)), c["odometry"], c["y"],
                            c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"])
    assert ei.value.status == rbpf.RBPF_ERR_NO_DEVICE


def test_host_side_basis_selection_matches_oracle(rbpf, oracle):
    for m, d, LL in [(16, 3, [[-11.8, -7.4, -2.4], [11.8, 7.4, 2.4]]), (256, 3, [[-12.02, -12.02, -2.4], [12.02, 12.02, 2.4]]),
                     (128, 2, [[-1.0, -2.5], [1.0, 2.5]]), (40, 2, [[-3.0, -3.0], [3.0, 3.0]])]:
        L1, NN1 = rbpf.domain_cartesian_dx(m, d, LL)
        L2, NN2 = oracle.domain_cartesian_dx(m, d, np.array(LL))
        np.testing.assert_array_equal(NN1, NN2.astype(np.int32))
        np.testing.assert_allclose(L1, L2, rtol=0)
        np.testing.assert_allclose(rbpf.eigenval(NN1, L1), oracle.eigenval(NN2, L2), rtol=1e-15)


def test_priors_match_oracle(rbpf, oracle):
    LL = np.array([[-12.0, -11.0, -2.4], [12.0, 11.0, 2.4]])
    _, x0a, P0a, Ra = rbpf.dense_mag_prior(64, LL, cases.THETA_MAG)
    _, x0b, P0b, Rb = oracle.dense_mag_prior(64, LL, cases.THETA_MAG)
    np.testing.assert_allclose(P0a, P0b, rtol=1e-14)
    np.testing.assert_array_equal(Ra, Rb)
    LL2 = np.array([[-1.0, -2.5], [1.0, 2.5]])
    _, _, P0a, Ra = rbpf.dense_radio_prior(32, LL2, cases.THETA_RADIO)
    _, _, P0b, Rb = oracle.dense_radio_prior(32, LL2, cases.THETA_RADIO)
    np.testing.assert_allclose(P0a, P0b, rtol=1e-14)
    np.testing.assert_array_equal(Ra, Rb)


def test_product_datagen_matches_oracle_generator(rbpf, oracle):
    import importlib
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    a = dg.bean_6D(30, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=5, m_sim=80)
    b = oracle.generate_bean_6D(30, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=5, m_sim=80)
    for k in ("dx", "initState", "y", "LL"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("traj,N_T", [("line_3D", 32), ("square_3D", 48)])
def test_product_radio_datagen_matches_oracle_generator(rbpf, oracle, traj, N_T):
    import importlib
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    Q = dg.radio_Q(N_T, traj)
    if traj == "line_3D":
        np.testing.assert_array_equal(Q, cases.radio_case(2, N_T, 8)["Q"])
    a = dg.planar_heading(N_T, Q, cases.THETA_RADIO, 1.0, seed=7, m_sim=90, nLL=4, traj=traj)
    b = oracle.generate_line_3D(N_T, Q, cases.THETA_RADIO, 1.0, seed=7, m_sim=90, nLL=4, traj=traj)
    for k in ("dx", "initState", "y", "LL"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-11, atol=1e-12)


def test_replay_rng_shape_validation(rbpf):
    c = cases.radio_case(4, 3, 8, seed=1)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    bad = rbpf.ReplayRNG(np.zeros((1, 2, 3)), np.zeros((1, 2, 3, 1)))
    with pytest.raises(ValueError):
        rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                            c["N_P"], c["dt"], rng=bad)


def test_procrustes_recovers_a_similarity_transform(rbpf):
    """metrics.procrustes (MATLAB `procrustes`, used by calc_rmses.m:37) recovers scale, rotation and translation."""
    import importlib
    mt = importlib.import_module(rbpf.__name__ + ".metrics")
    rs = np.random.RandomState(0)
    Y = rs.standard_normal((20, 2))
    th = 0.7
    T = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    X = 1.7 * Y @ T + np.array([3.0, -2.0])
    d, Z, tr = mt.procrustes(X, Y)
    assert d < 1e-12 and abs(tr["b"] - 1.7) < 1e-12
    np.testing.assert_allclose(Z, X, atol=1e-12)
    np.testing.assert_allclose(tr["T"], T, atol=1e-12)
    rp, rm = mt.calc_rmses(X, Y, X[:5], Y[:5])
    assert rp < 1e-12 and rm < 1e-12


def test_acceptance_metrics_match_the_oracle(rbpf, oracle):
    """the product's Procrustes-aligned RMSEs (examples/slam-sparse-visual/calc_rmses.m:35-55,
    run_dense3D_magfield.m:155-181) against the oracle's restatement (a different route to the same definition), and a
    known answer: a similarity transform of the truth is undone exactly."""
    import importlib
    M = importlib.import_module(rbpf.__name__ + ".metrics")
    rs = np.random.RandomState(8)
    mp = rs.standard_normal((20, 2)) * 3
    th = 0.7
    Rm = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    traj = np.column_stack((np.cumsum(rs.standard_normal((50, 2)) * 0.2, axis=0), rs.standard_normal(50)))
    map_est = 1.7 * mp @ Rm + np.array([0.3, -2.0]) + 0.05 * rs.standard_normal(mp.shape)
    traj_est = np.column_stack((1.7 * traj[:, 0:2] @ Rm + np.array([0.3, -2.0]) + 0.05 * rs.standard_normal((50, 2)), traj[:, 2]))
    got = M.calc_rmses(mp, map_est, traj, traj_est)
    want = oracle.calc_rmses_oracle(mp, map_est, mp, map_est, traj, traj_est)
    np.testing.assert_allclose(got, want, rtol=1e-10)
    d, Z, tr = M.procrustes(mp, 1.7 * mp @ Rm + np.array([0.3, -2.0]))                  # exact similarity: undone exactly
    assert d < 1e-14 and np.max(np.abs(Z - mp)) < 1e-12 and abs(tr["b"] - 1 / 1.7) < 1e-12
    d2, Z2, tr2 = oracle.procrustes_oracle(mp, map_est)
    d1, Z1, tr1 = M.procrustes(mp, map_est)
    assert abs(d1 - d2) < 1e-12 and np.max(np.abs(Z1 - Z2)) < 1e-10 and np.max(np.abs(tr1["T"] - tr2["T"])) < 1e-10
    # dense-mag acceptance numbers on a synthetic estimate
    c = cases.mag_case(4, 30, 16, seed=12)
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    dd = dg.bean_6D(30, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=5, m_sim=100)
    dd["pos"][2] = 0.4 * np.sin(np.linspace(0, 5, 30))         # a non-planar path: the alignment is unique (full-rank X0'Y0)
    est = np.vstack((dd["pos"] + 0.1 * rs.standard_normal(dd["pos"].shape), dd["quat"].T))
    qn = rs.standard_normal(4) * 0.02 + np.array([1.0, 0, 0, 0])
    est[3:7] = np.stack([oracle.qLeft(qn / np.linalg.norm(qn)) @ est[3:7, i] for i in range(30)], axis=1)
    gp, go = M.rmse_dense_mag(dd["pos"], dd["quat"], est)
    wp, wo = oracle.rmse_dense_mag_oracle(dd["pos"], dd["quat"], est)
    np.testing.assert_allclose(gp, wp, rtol=1e-9)
    np.testing.assert_allclose(go, wo, rtol=1e-9)
    assert np.all(gp < 0.3) and np.all(go < 10.0)
    del c
