"""GPU parity of the stand-alone pieces of SURVEY 8(a): sample (a2), dynModel (a3/a4), quaternion helpers through
them (a5), basis / measModel (a7/a8), dynResNorm (a17), JacobianPhi3D (a19) -- HIP vs the numpy oracle."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def test_sample_is_bit_exact_including_bin_edges(rbpf, oracle):
    rs = np.random.RandomState(3)
    for N in (1, 7, 64, 1000, 5000):
        w = rs.random_sample(N) ** 3
        w[rs.random_sample(N) < 0.2] = 0.0                       # plateaus in the running sum
        w = w / w.sum() if w.sum() > 0 else np.ones(N) / N
        wc = np.cumsum(w)
        u = np.concatenate((rs.random_sample(200), wc[:50], np.nextafter(wc[:50], 2.0), np.nextafter(wc[:50], -1.0),
                            [0.0, 1e-300, wc[-1]]))
        u = u[u <= wc[-1]]                                       # beyond wc(end) MATLAB errors; device clamps
        want = np.array([oracle.sample(w, ui) for ui in u])
        got = rbpf.sample(w, u)
        np.testing.assert_array_equal(got, np.minimum(want, N - 1))
    got = rbpf.sample(np.array([0.25, 0.25, 0.25, 0.25]), [0.9999999, 1.5])
    np.testing.assert_array_equal(got, [3, 3])                   # u > wc(end): N+1 in MATLAB -> clamped to N


def test_dense_mag_dynmodel(rbpf, oracle):
    rs = np.random.RandomState(4)
    c = cases.mag_case(4, 4, 16, seed=1)
    mdl, *_ = cases.device_model(rbpf, c)
    for _ in range(5):
        q = rs.standard_normal(4)
        xn = np.concatenate((rs.standard_normal(3), q / np.linalg.norm(q)))
        dq = rs.standard_normal(4)
        dx = np.concatenate((rs.standard_normal(3) * 0.1, dq / np.linalg.norm(dq)))
        z = rs.standard_normal(6)
        A = rs.standard_normal((6, 6))
        Q = np.zeros((6, 6))
        Q[:3, :3] = (A @ A.T)[:3, :3] * 1e-2
        Q[3:, 3:] = (A @ A.T)[3:, 3:] * 1e-3                      # non-diagonal blocks exercise the 3x3 chol
        want, _ = c["model"].dynModel(xn, dx, 0.02, Q, z)
        got = mdl.dynModel(xn, dx, 0.02, Q, z)
        assert rel(got, want) < 1e-13
    # a rotation increment large enough to flip the sign in expq (expq.m:30)
    z = np.array([0, 0, 0, 40.0, 0, 0])
    Qb = np.diag([1e-2] * 3 + [1.0] * 3)
    want, _ = c["model"].dynModel(xn, dx, 0.01, Qb, z)
    assert rel(mdl.dynModel(xn, dx, 0.01, Qb, z), want) < 1e-13


def test_dense_radio_dynmodel(rbpf, oracle):
    rs = np.random.RandomState(5)
    c = cases.radio_case(4, 4, 16, seed=1)
    mdl, *_ = cases.device_model(rbpf, c)
    xn = rs.standard_normal((3, 6))
    z = rs.standard_normal((1, 6))
    dx = np.array([0.1, -0.2, 0.05])
    got = mdl.dynModel(xn, dx, 1.0, np.array([[0.09]]), z)
    for i in range(6):
        want, _ = c["model"].dynModel(xn[:, i], dx, 1.0, np.array([[0.09]]), z[:, i])
        assert rel(got[:, i], want) < 1e-14


@pytest.mark.parametrize("m", [16, 256, 512])
def test_dense_mag_measmodel(rbpf, m):
    c = cases.mag_case(4, 4, m, seed=2)
    mdl, *_ = cases.device_model(rbpf, c)
    rs = np.random.RandomState(6)
    xn = np.zeros((7, 5))
    xn[0:3] = rs.uniform(-1, 1, (3, 5)) * c["model"].L[:, None] * 0.9
    q = rs.standard_normal((4, 5))
    xn[3:7] = q / np.linalg.norm(q, axis=0)
    want = c["model"].measModel(xn)                               # [Npred x 3 x n]
    got = mdl.measModel(xn)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))


def test_dense_radio_measmodel(rbpf):
    c = cases.radio_case(4, 4, 128, seed=2)
    mdl, *_ = cases.device_model(rbpf, c)
    rs = np.random.RandomState(7)
    xn = rs.uniform(-1, 1, (3, 9))
    want = c["model"].measModel(xn)                               # [Npred x m] (2-D, ny = 1)
    got = mdl.measModel(xn)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))


def test_dynresnorm_both_families(rbpf):
    rs = np.random.RandomState(8)
    c = cases.mag_case(4, 4, 16, seed=1)
    mdl, *_ = cases.device_model(rbpf, c)
    for _ in range(5):
        def rq():
            q = rs.standard_normal(4)
            return q / np.linalg.norm(q)
        xk = np.concatenate((rs.standard_normal(3), rq()))
        xi = np.concatenate((rs.standard_normal(3), rq()))
        dx = np.concatenate((rs.standard_normal(3) * 0.1, rq()))
        A = rs.standard_normal((6, 6))
        Q = A @ A.T * 1e-2 + np.eye(6) * 1e-3                     # full (non-diagonal) Q: r'/L != L\r
        want = c["model"].dynResNorm(xk, xi, dx, 0.02, Q)
        got = mdl.dynResNorm(xk, xi, dx, 0.02, Q)
        assert rel(got, want) < 1e-12
    c2 = cases.radio_case(4, 4, 16, seed=1)
    mdl2, *_ = cases.device_model(rbpf, c2)
    want = c2["model"].dynResNorm(np.array([0.1, 0.2, 0.7]), np.array([0.0, 0.1, 0.3]), np.array([0.1, 0.0, 0.2]), 1.0,
                                  np.array([[0.04]]))
    got = mdl2.dynResNorm(np.array([0.1, 0.2, 0.7]), np.array([0.0, 0.1, 0.3]), np.array([0.1, 0.0, 0.2]), 1.0,
                          np.array([[0.04]]))
    assert rel(got, want) < 1e-14


def test_jacobianphi3d(rbpf, oracle):
    c = cases.mag_case(4, 4, 48, seed=3)
    mdl, *_ = cases.device_model(rbpf, c)
    LL = c["LL"]
    rs = np.random.RandomState(9)
    x = rs.uniform(-0.8, 0.8, (3, 6)) * c["model"].L[:, None]
    want = oracle.JacobianPhi3D(x, 48, LL[0, 0], LL[1, 0], LL[0, 1], LL[1, 1], LL[0, 2], LL[1, 2], c["model"].NN)
    got = mdl.JacobianPhi3D(x, LL[0], LL[1])
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))


@pytest.mark.gpu
def test_ekf_baseline_matches_oracle(rbpf):
    """examples/slam-dense-mag/ekf_dense.m with measModel_ekf / dynModel_ekf (run_dense3D_magfield.m:281-299,310-316):
    host recursion over the device helper kernels (rotated basis gradient, basis Hessian) against the oracle."""
    import importlib
    import rbpf_oracle as O
    ekf = importlib.import_module(rbpf.__name__ + ".ekf")
    c = cases.mag_case(4, 14, 40, seed=8)
    mdl, x0l, P0l, R = cases.device_model(rbpf, c)
    n = mdl.nLin
    x0 = np.concatenate((c["x0_nonLin"][0:3], np.zeros(3), x0l))                # run_dense3D_magfield.m:248-250
    P0 = np.zeros((6 + n, 6 + n))
    P0[6:, 6:] = P0l
    q0 = c["x0_nonLin"][3:7]
    ref = O.ekf_dense(c["model"], c["LL"], c["odometry"], c["y"], x0, q0, P0, c["Q"], c["R"], c["dt"])
    got = ekf.ekf_dense(mdl, c["LL"], c["odometry"], c["y"], x0, q0, P0, c["Q"], R, c["dt"])
    for g, r in zip(got, ref):
        assert np.max(np.abs(g - r)) <= 1e-9 * max(1.0, np.max(np.abs(r)))


@pytest.mark.gpu
def test_quaternion_helpers_match_oracle(rbpf, oracle):
    """expq / logq (scalar and batched branches, quirk Q7), qLeft, qRight, qInv, quat2rmat, mcross on the
    device against the restatement of tools/*.m, bit for bit where the arithmetic is exact, 1e-15 otherwise."""
    rs = np.random.RandomState(4)
    n = 257
    phi = rs.standard_normal((n, 3)) * rs.choice([1e-9, 0.3, 2.0, 4.0], (n, 1))
    phi[0] = 0.0                                                      # mag_phi == 0 guard (expq.m:24)
    phi[1] = [np.pi / 2, 0.0, 0.0]                                    # cos = 6e-17 > 0: no flip
    phi[2] = [0.0, 3 * np.pi / 2, 0.0]                                # cos < 0 by rounding: flip in both branches
    q = rs.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    q[0] = [1.0, 0.0, 0.0, 0.0]                                       # na == 0 guard (logq.m:30)
    q[1] = [0.0, 0.6, 0.0, 0.8]                                       # q0 == 0: scalar branch keeps, batched flips
    q[2] = [1.0 + 2e-16, 1e-9, 0.0, 0.0]                              # q0 > 1 by rounding: clamped (quirk Q7)
    q[3] = [-0.5, 0.5, -0.5, 0.5]
    got = rbpf.quat_helper("expq", phi)
    want = np.stack([oracle.expq(p) for p in phi])
    assert np.max(np.abs(got - want)) <= 2e-15                        # a few ulp: the device's sincos vs libm's
    gb = rbpf.quat_helper("expq_batched", phi)
    assert np.max(np.abs(gb - oracle.expq_batched(phi))) <= 2e-15
    pz = np.array([[np.pi / 2 * (1 + 1e-16), 0.0, 0.0]])             # a q0 that is exactly +0 or tiny: both stay valid rotations
    assert np.allclose(np.abs(rbpf.quat_helper("expq", pz)), np.abs(rbpf.quat_helper("expq_batched", pz)), atol=1e-15)
    gl = rbpf.quat_helper("logq", q)
    wl = np.stack([oracle.logq(x) for x in q])
    assert np.max(np.abs(gl - wl)) <= 4e-15 * max(1.0, np.max(np.abs(wl)))
    glb = rbpf.quat_helper("logq_batched", q)
    wlb = oracle.logq_batched(q)
    assert np.max(np.abs(glb - wlb)) <= 4e-15 * max(1.0, np.max(np.abs(wlb)))
    assert np.array_equal(glb[1], -gl[1]) and np.any(gl[1] != 0)      # the q0 == 0 row: opposite signs (quirk Q7)
    np.testing.assert_array_equal(rbpf.quat_helper("qLeft", q), np.stack([oracle.qLeft(x) for x in q]))
    np.testing.assert_array_equal(rbpf.quat_helper("qRight", q), np.stack([oracle.qRight(x) for x in q]))
    np.testing.assert_array_equal(rbpf.quat_helper("qLeft", q), oracle.qLeft_batched(q))
    np.testing.assert_array_equal(rbpf.quat_helper("qRight", q), oracle.qRight_batched(q))
    np.testing.assert_array_equal(rbpf.quat_helper("qInv", q), oracle.qInv(q))
    np.testing.assert_array_equal(rbpf.quat_helper("mcross", phi), oracle.mcross_batched(phi))
    np.testing.assert_array_equal(rbpf.quat_helper("mcross", phi), np.stack([oracle.mcross(p) for p in phi]))
    gr = rbpf.quat_helper("quat2rmat", q)
    wr = oracle.quat2rmat_batched(q)
    assert np.max(np.abs(gr - wr)) <= 1e-15                           # fused multiply-adds on the device
    # identities the reference's algebra rests on (SURVEY 8c anchors), on the device outputs themselves
    a, b = q[5:40], q[40:75]
    QL, QR = rbpf.quat_helper("qLeft", a), rbpf.quat_helper("qRight", b)
    assert np.max(np.abs(np.einsum("nij,nj->ni", QL, b) - np.einsum("nij,nj->ni", QR, a))) <= 1e-15   # qLeft(a)*b == qRight(b)*a
    nrm = np.linalg.norm(phi, axis=1)
    small = phi[nrm < np.pi / 2]
    back = rbpf.quat_helper("logq", rbpf.quat_helper("expq", small))
    # logq(expq(phi)) = phi for |phi| < pi/2 -- to sqrt(eps) only: tools/logq.m:29 takes acos(q0), which cannot resolve
    # angles below ~1.5e-8 (q0 rounds to 1); the reference's arithmetic, reproduced as is
    assert np.max(np.abs(back - small)) <= 3e-8
    mid = phi[(nrm > 0.05) & (nrm < np.pi / 2)]
    assert np.max(np.abs(rbpf.quat_helper("logq", rbpf.quat_helper("expq", mid)) - mid)) <= 1e-13
    assert np.max(np.abs(np.einsum("nij,nkj->nik", gr, gr) - np.eye(3))) <= 1e-14                     # orthonormal


@pytest.mark.gpu
def test_synthetic_data_generators_on_the_device(rbpf, oracle):
    """the field draws of generateData_dense.m:216-257 (tools/gp_rnd_scalar_potential_fast.m:42-102,
    gp_rnd_SE1D_fast.m:44-85) with the 2000-function simulation basis evaluated by the HIP measurement-model kernel,
    against the host tables and against the oracle's generator (same seeded stream)."""
    import importlib
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    a = dg.bean_6D(60, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=7, m_sim=2000)
    b = dg.bean_6D(60, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=7, m_sim=2000, device=True)
    for k in ("dx", "initState", "LL", "pos", "quat"):
        np.testing.assert_array_equal(a[k], b[k])
    assert np.max(np.abs(a["y"] - b["y"])) <= 1e-10 * np.max(np.abs(a["y"]))
    o = oracle.generate_bean_6D(60, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=7, m_sim=2000)
    assert np.max(np.abs(o["y"] - b["y"])) <= 1e-9 * np.max(np.abs(o["y"]))
    Qr = dg.radio_Q(48, "square_3D")
    ra = dg.planar_heading(48, Qr, cases.THETA_RADIO, 1.0, seed=3, nLL=4, traj="square_3D", m_sim=2000)
    rb = dg.planar_heading(48, Qr, cases.THETA_RADIO, 1.0, seed=3, nLL=4, traj="square_3D", m_sim=2000, device=True)
    np.testing.assert_array_equal(ra["dx"], rb["dx"])
    assert np.max(np.abs(ra["y"] - rb["y"])) <= 1e-10 * np.max(np.abs(ra["y"]))
