"""Independent anchors for the oracle: the reference ships no tests or golden vectors, so
the numpy restatement is pinned by identities and cross-implementation checks.  CPU only."""
import numpy as np
import pytest

import cases
import rbpf_oracle as O

RS = np.random.RandomState(42)


def rand_quat(rs):
    q = rs.standard_normal(4)
    return q / np.linalg.norm(q)


def test_quat2rmat_is_a_rotation():
    for _ in range(20):
        R = O.quat2rmat(rand_quat(RS))
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-14)
        assert abs(np.linalg.det(R) - 1.0) < 1e-13


def test_qleft_qright_commute():
    for _ in range(20):
        a, b = rand_quat(RS), rand_quat(RS)
        np.testing.assert_allclose(O.qLeft(a) @ b, O.qRight(b) @ a, atol=1e-15)      # tools/qLeft.m vs qRight.m


def test_quaternion_product_composes_rotations():
    for _ in range(10):
        a, b = rand_quat(RS), rand_quat(RS)
        np.testing.assert_allclose(O.quat2rmat(O.qLeft(a) @ b), O.quat2rmat(a) @ O.quat2rmat(b), atol=1e-14)


def test_logq_inverts_expq_and_qinv():
    for _ in range(20):
        phi = RS.uniform(-0.5, 0.5, 3)
        np.testing.assert_allclose(O.logq(O.expq(phi)), phi, atol=1e-14)
        q = rand_quat(RS)
        np.testing.assert_allclose(O.qLeft(O.qInv(q)) @ q, [1, 0, 0, 0], atol=1e-15)
    np.testing.assert_array_equal(O.expq(np.zeros(3)), [1, 0, 0, 0])                 # mag==0 guard, expq.m:25


def test_expq_sign_flip_quirk_q7():
    phi = np.array([2.0, 0.0, 0.0])           # cos(2) < 0 -> flipped in both branches
    assert O.expq(phi)[0] > 0 and O.expq_batched(phi[None])[0, 0] > 0
    phi = np.array([np.pi / 2, 0.0, 0.0])     # cos == 6e-17 > 0: neither branch flips
    assert O.expq(phi)[1] > 0
    b = O.expq_batched(np.array([[0.3, -0.2, 0.5], [1.9, 0.1, 0.0]]))
    for row, ph in zip(b, [[0.3, -0.2, 0.5], [1.9, 0.1, 0.0]]):
        np.testing.assert_allclose(row, O.expq(ph), atol=1e-15)
    lq = O.logq_batched(np.stack([O.expq([0.1, 0.2, -0.3]), -O.expq([0.2, 0.0, 0.1])]))
    np.testing.assert_allclose(lq, [[0.1, 0.2, -0.3], [0.2, 0.0, 0.1]], atol=1e-14)


def test_rmat2quat_planar_round_trip():
    for th in np.linspace(-3.0, 3.0, 13):
        q = O.rmat2quat_planar(th)
        R = np.array([[np.cos(th), np.sin(th), 0], [-np.sin(th), np.cos(th), 0], [0, 0, 1]])   # generateData_dense.m:196
        np.testing.assert_allclose(O.quat2rmat(q), R, atol=1e-14)


def test_domain_cartesian_dx_selection_is_sorted_and_stable():
    L, NN = O.domain_cartesian_dx(64, 3, np.array([[-12.0, -12.0, -2.4], [12.0, 12.0, 2.4]]))
    lam = O.eigenval(NN, L)
    assert np.all(np.diff(lam) >= 0)
    assert NN.shape == (64, 3) and NN.min() == 1
    # ties (x/y symmetric box) keep enumeration order: first axis slowest -> (1,2,.) before (2,1,.)
    i12 = np.where((NN[:, 0] == 1) & (NN[:, 1] == 2) & (NN[:, 2] == 1))[0][0]
    i21 = np.where((NN[:, 0] == 2) & (NN[:, 1] == 1) & (NN[:, 2] == 1))[0][0]
    assert i12 < i21


def test_eigenfun_dx_matches_finite_differences():
    L, NN = O.domain_cartesian_dx(40, 3, np.array([[-3.0, -2.0, -1.0], [3.0, 2.0, 1.0]]))
    x = RS.uniform(-0.8, 0.8, (5, 3))
    h = 1e-6
    for di in range(3):
        e = np.zeros(3)
        e[di] = h
        fd = (O.eigenfun(NN, x + e, L) - O.eigenfun(NN, x - e, L)) / (2 * h)
        np.testing.assert_allclose(O.eigenfun_dx(NN, x, di, L), fd, rtol=1e-6, atol=1e-8)


def test_eigenfun_is_orthonormal_on_the_box():
    L, NN = O.domain_cartesian_dx(12, 2, np.array([[-1.5, -1.0], [1.5, 1.0]]))
    g = [np.linspace(-L[a], L[a], 401) for a in range(2)]
    X, Y = np.meshgrid(*g, indexing="ij")
    Phi = O.eigenfun(NN, np.column_stack((X.ravel(), Y.ravel())), L)
    w = np.outer(np.gradient(g[0]), np.gradient(g[1])).ravel()           # trapezoid-like weights
    G = Phi.T @ (Phi * w[:, None])
    np.testing.assert_allclose(G, np.eye(12), atol=2e-3)


def test_jacobianphi3d_is_the_derivative_of_eigenfun_dx():
    lo, up = np.array([-3.0, -2.0, -1.0]), np.array([3.0, 2.0, 1.0])
    L, NN = O.domain_cartesian_dx(30, 3, np.vstack((lo, up)))
    x = RS.uniform(-0.7, 0.7, (3, 4))
    J = O.JacobianPhi3D(x, 30, lo[0], up[0], lo[1], up[1], lo[2], up[2], NN)
    h = 1e-6
    for a in range(3):
        for b in range(3):
            e = np.zeros(3)
            e[b] = h
            fd = (O.eigenfun_dx(NN, (x.T + e), a, L) - O.eigenfun_dx(NN, (x.T - e), a, L)) / (2 * h)   # [Np x m]
            np.testing.assert_allclose(J[a, b].T, fd, rtol=1e-5, atol=1e-7)


def test_sample_frequencies_follow_the_weights():
    """tools/sample.m:36-64 (the commented-out self-test): empirical frequencies ~ w."""
    rs = np.random.RandomState(0)
    w = rs.random_sample(8)
    w /= w.sum()
    u = rs.random_sample(100000)
    idx = np.array([O.sample(w, ui) for ui in u[:20000]])
    freq = np.bincount(idx, minlength=8) / idx.size
    assert np.max(np.abs(freq - w)) < 0.012
    wc = np.cumsum(w)
    assert O.sample(w, wc[3]) == 3 and O.sample(w, np.nextafter(wc[3], 2)) == 4      # strict '<' at a bin edge
    assert O.sample(w, 0.0) == 0
    assert O.sample(w, 2.0) == 8                                                     # N+1 in MATLAB -> error


def test_sequential_kalman_updates_equal_batch_gp_posterior():
    """Along a FIXED trajectory the per-step Kalman updates of particleFilter.m:184-198 must reproduce the
    batch reduced-rank GP posterior mean L'\\(L\\(Phi'*y)) with L = chol(Phi'Phi + diag(sigma2./k))
    (tools/gp_scalar_potential_fast.m:190-192)."""
    c = cases.mag_case(N_P=1, N_T=25, m=24, seed=4)
    model, P0, R, y = c["model"], c["P0_lin"], c["R"], c["y"]
    sigma2 = R[0, 0]
    k = np.diag(P0)
    rs = np.random.RandomState(1)
    xn = np.zeros((7, 25))
    xn[0:3] = rs.uniform(-2, 2, (3, 25))
    xn[3] = 1.0                                                   # identity attitude -> H = [dPhix;dPhiy;dPhiz]
    dy = model.measModel(xn)                                      # [T x 3 x n]
    xl, P = np.zeros(model.nLin), P0.copy()
    for t in range(25):
        xl, P, _, _ = O._kalman_update(y[t], dy[t], xl, P, R, 1e-3)
    Phi = dy.reshape(-1, model.nLin)                              # stacked [3T x n]
    Lc = np.linalg.cholesky(Phi.T @ Phi + np.diag(sigma2 / k))
    foo = np.linalg.solve(Lc.T, np.linalg.solve(Lc, Phi.T @ y.reshape(-1)))
    np.testing.assert_allclose(xl, foo, rtol=1e-8, atol=1e-9)
    # posterior covariance: sigma2 * inv(Phi'Phi + diag(sigma2./k))
    np.testing.assert_allclose(P, sigma2 * np.linalg.inv(Lc @ Lc.T), rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("kind", ["mag", "radio"])
def test_covariance_and_information_form_smoothers_agree(kind):
    """particleSmootherInformationForm.m:35-37: 'identical ... but computes the weights in information form'.
    After normalisation w and the ancestor probabilities must agree (quirk Q6: only after normalisation)."""
    c = cases.mag_case(8, 7, 16, seed=21, N_K=3) if kind == "mag" else cases.radio_case(9, 8, 24, seed=21, N_K=3)
    s1 = cases.oracle_smoother(c, info_form=False)
    s2 = cases.oracle_smoother(c, info_form=True)
    np.testing.assert_array_equal(s1["trace"]["ai"], s2["trace"]["ai"])
    np.testing.assert_allclose(s1["trace"]["w"], s2["trace"]["w"], rtol=0, atol=1e-10)
    a, b = s1["trace"]["paNt"][1:, 1:], s2["trace"]["paNt"][1:, 1:]
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)
    np.testing.assert_allclose(s1["XNK"], s2["XNK"], atol=1e-12)
    np.testing.assert_allclose(s1["PK"], s2["PK"], rtol=1e-9, atol=1e-12)


def test_filter_quirk_q3_p_mean_is_last_particle_term_only():
    c = cases.radio_case(6, 5, 16, seed=2)
    r = cases.oracle_filter(c)
    tr = r["trace"]
    i = c["N_P"] - 1
    d = r["xl_mean"] - tr["xl"][:, i]
    np.testing.assert_allclose(r["P_mean"], tr["w"][-1, i] * (tr["P"][:, :, i] + np.outer(d, d)), rtol=1e-14)


def test_jitter_retry_path_q2():
    """chol fails on S -> retried on S + jitter*I (particleFilter.m:145-148); a second failure raises."""
    S = np.array([[-1e-4, 0.0], [0.0, 5e-4]])
    L = O._chol_lower_with_jitter(S, 1e-3)
    np.testing.assert_allclose(L @ L.T, S + 1e-3 * np.eye(2), atol=1e-18)
    with pytest.raises(O.CholeskyFailure):
        O._chol_lower_with_jitter(-np.eye(2), 1e-3)


def test_smoother_reference_slot_consumes_one_uniform_q9():
    """For k>1 slot N_P takes no randn and exactly one rand (particleSmoother.m:132-137,241)."""
    c = cases.radio_case(6, 6, 16, seed=3, N_K=2)
    base = cases.oracle_smoother(c, info_form=False)
    rng2 = O.ReplayRNG(c["rng"].U.copy(), c["rng"].Z.copy(), c["rng"].Ufin.copy())
    rng2.Z[1, :, -1, :] = 123.0                                 # normals of slot N_P in iteration 2: never read
    c2 = dict(c, rng=rng2)
    again = cases.oracle_smoother(c2, info_form=False)
    np.testing.assert_array_equal(base["XNK"], again["XNK"])


def test_ekf_measurement_jacobian_matches_finite_differences(oracle):
    """measModel_ekf (run_dense3D_magfield.m:281-299): the position block uses JacobianPhi3D with the domain's lower /
    upper bounds, the prediction uses eigenfun_dx with half-widths -- central differences of yhat tie the two together;
    the orientation block Rnb'*[dPhi*xl x] is the derivative of R(q (x) expq(eta/2))' g at eta = 0 up to the sign
    convention of the error state, checked through the map block (exact, linear)."""
    import cases
    c = cases.mag_case(3, 4, 24, seed=4)
    m, n = c["model"], c["model"].nLin
    rs = np.random.RandomState(1)
    x = np.concatenate((0.3 * rs.standard_normal(3), np.zeros(3), rs.standard_normal(n)))
    q = oracle.expq(0.2 * rs.standard_normal(3))
    yh, dy = oracle.measModel_ekf(m, c["LL"], x, q)
    J = np.zeros((3, 3))
    for j in range(3):
        d = np.zeros_like(x)
        d[j] = 1e-6
        J[:, j] = (oracle.measModel_ekf(m, c["LL"], x + d, q)[0] - oracle.measModel_ekf(m, c["LL"], x - d, q)[0]) / 2e-6
    np.testing.assert_allclose(dy[:, 0:3], J, rtol=1e-6, atol=1e-8 * np.abs(J).max())
    np.testing.assert_allclose(dy[:, 6:] @ x[6:], yh, rtol=1e-12)              # linear in the map states


@pytest.mark.parametrize("info_form", [False, True])
def test_oracle_cpf_as_reproduces_rts_smoother_moments(info_form):
    """A known answer that pins the restatement of CPF-AS itself (particleSmoother.m:88-366, ...InformationForm.m:98-362), not
    just its agreement with a second restatement: on the jointly linear-Gaussian toy model of tests/kat_rts.py the oracle's
    trajectory draws and map posteriors reproduce the Rauch-Tung-Striebel moments (300 iterations, 60 discarded, fixed seeds;
    means within 4 standard errors at an effective sample size of a quarter of the draws, variances within [0.7, 1.45])."""
    import kat_rts
    p = kat_rts.problem()

    class Toy:
        nNonLin, ny, nw, nLin = 1, 1, 1, 2

        def dynModel(self, xn, dx, dt, Qt, z):
            return np.asarray(xn).ravel() + np.asarray(dx).ravel() + np.sqrt(dt * Qt[0, 0]) * np.asarray(z).ravel(), None

        def measModel(self, xn):
            return kat_rts.measModel(xn)

    N_P, N_K = 32, 300
    f = O.particleSmootherInformationForm if info_form else O.particleSmoother
    out = f(Toy(), p["odometry"], p["y"], p["x0_nonLin"], p["x0_lin"], p["P0_lin"], p["Q"], p["R"], N_P, N_K, p["dt"],
            O.ReplayRNG.draw(7, N_K, p["T"], N_P, 1), trace=False, use_dynResNorm=False)
    kat_rts.check_moments(p, out["XNK"], out["XLK"], out["PK"], burn=60, n_se=4.0, var_lo=0.7, var_hi=1.45)
