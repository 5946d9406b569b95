"""GPU parity: HIP particleSmoother / particleSmootherInformationForm vs the numpy oracle."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def run_both(rbpf, c, info_form, chol_variant=0, **kw):
    ref = cases.oracle_smoother(c, info_form)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    f = rbpf.particleSmootherInformationForm if info_form else rbpf.particleSmoother
    out = f(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
            c["N_P"], c["N_K"], c["dt"], rng=cases.device_rng(rbpf, c), extras=True, chol_variant=chol_variant, **kw)
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


@pytest.mark.parametrize("chol_refresh", [1, 0])
@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 8, 6, 16), ("mag", 6, 5, 130), ("radio", 10, 8, 32),
                                            ("radio", 9, 7, 128)])
def test_information_form_smoother_matches_oracle(rbpf, kind, N_P, N_T, m, chol_refresh):
    """chol_refresh = 1: the reference's arithmetic (chol(Imat_i + ImatAddt) at every step, :228); 0: the library default (carried
    factors from nLin = 128 on, the same as 1 below)."""
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=9, N_K=3)
    ref, out = run_both(rbpf, c, info_form=True, chol_refresh=chol_refresh)
    check(ref, out, 3)


def test_information_form_smoother_above_8192_particles(rbpf):
    """N_P > 8192: weights and ancestor probabilities are normalised by the multi-workgroup pipeline (parallel prefix,
    certified search, exact fallback) instead of the single-workgroup kernels -- same indices as the strict
    left-to-right cumsum of tools/sample.m:30."""
    c = cases.radio_case(8300, 3, 8, seed=13, N_K=2)
    ref, out = run_both(rbpf, c, info_form=True)
    check(ref, out, 2)


def _closures(c, N_P, N_T):
    """Plain Python closures over the oracle's model objects (unknown to the library): the generic family.  dynModel
    replays the oracle's normals in the reference's call order -- per step slots 0..N-1 for k = 0 and 0..N-2 afterwards
    (particleSmoother.m:132-137,149-152)."""
    mdl, Z = c["model"], c["rng"].Z
    state = {"k": 0, "t": 0, "i": 0, "calls": 0}

    def dynModel(xn, dx, dt, Q):
        k, t, i = state["k"], state["t"], state["i"]
        out = mdl.dynModel(xn, dx, dt, Q, Z[k, t, i])
        state["calls"] += 1
        i += 1
        if i == (N_P if k == 0 else N_P - 1):
            i, t = 0, t + 1
            if t == N_T - 1:
                t, k = 0, k + 1
        state.update(k=k, t=t, i=i)
        return out[0] if isinstance(out, tuple) else out

    return dynModel, (lambda xn: mdl.measModel(xn)), (lambda xnk, xni, dx, dt, Q: mdl.dynResNorm(xnk, xni, dx, dt, Q)), state


@pytest.mark.parametrize("info_form", [False, True])
@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 7, 6, 130), ("radio", 9, 7, 128), ("radio", 8, 6, 24)])
def test_smoothers_accept_arbitrary_handles(rbpf, kind, N_P, N_T, m, info_form):
    """particleSmoother.m:134-136,175-180,264 take any handles: unrecognised callables run through the generic family
    (rbpf_callbacks: dynModel / measModel / dynResNorm evaluated on the host, everything else on the device) and give
    the oracle's answers."""
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=27, N_K=3)
    ref = cases.oracle_smoother(c, info_form)
    dyn, meas, drn, state = _closures(c, N_P, N_T)
    f = rbpf.particleSmootherInformationForm if info_form else rbpf.particleSmoother
    out = f(dyn, meas, drn, c["odometry"], c["y"], c["x0_nonLin"], c["x0_lin"], c["P0_lin"], c["Q"], c["R"], N_P, 3, c["dt"],
            rng=cases.device_rng(rbpf, c), extras=True)
    assert state["calls"] == (N_T - 1) * (N_P + 2 * (N_P - 1))
    check(ref, out, 3)


def test_generic_smoother_with_an_empty_dynResNorm(rbpf):
    """isempty(dynResNorm): the additive default eDyn = (x'_t - x_i - odometry')' / chol(dt*Q,'lower')
    (particleSmoother.m:175-177) needs size(Q,1) == nNonLin; a 3-state additive toy model with a scalar field."""
    N_P, N_T, m, N_K = 8, 6, 20, 3
    c = cases.radio_case(N_P, N_T, m, seed=33, N_K=N_K)
    import rbpf_oracle as O
    rs = np.random.RandomState(5)
    Q = np.diag([0.02, 0.03, 0.01])
    Zk = rs.standard_normal((N_K, N_T - 1, N_P, 3))

    class Additive:                                              # oracle-side model object with the same closures
        nNonLin, ny, nw = 3, 1, 3
        NN, L = c["model"].NN, c["model"].L
        nLin = m

        def dynModel(self, xn, dx, dt, Qt, z):
            return np.asarray(xn).ravel() + np.asarray(dx).ravel() + np.linalg.cholesky(dt * Qt) @ np.asarray(z).ravel(), None

        def measModel(self, xn):
            return c["model"].measModel(xn)

        def dynResNorm(self, *a):
            raise AssertionError("not used: use_dynResNorm=False")

    mdl = Additive()
    rng = O.ReplayRNG(c["rng"].U, Zk, c["rng"].Ufin)
    ref = O.particleSmootherInformationForm(mdl, c["odometry"], c["y"], c["x0_nonLin"], c["x0_lin"], c["P0_lin"], Q, c["R"], N_P,
                                            N_K, 1.0, rng, trace=True, use_dynResNorm=False)
    st = {"k": 0, "t": 0, "i": 0}

    def dyn(xn, dx, dt, Qt):
        k, t, i = st["k"], st["t"], st["i"]
        out = mdl.dynModel(xn, dx, dt, Qt, Zk[k, t, i])[0]
        i += 1
        if i == (N_P if k == 0 else N_P - 1):
            i, t = 0, t + 1
            if t == N_T - 1:
                t, k = 0, k + 1
        st.update(k=k, t=t, i=i)
        return out

    out = rbpf.particleSmootherInformationForm(dyn, mdl.measModel, [], c["odometry"], c["y"], c["x0_nonLin"], c["x0_lin"], c["P0_lin"],
                                               Q, c["R"], N_P, N_K, 1.0, rng=rbpf.ReplayRNG(c["rng"].U, Zk, c["rng"].Ufin), extras=True)
    check(ref, out, N_K)


def test_smoother_makePlots_runs_after_every_iteration(rbpf):
    """particleSmoother.m:360-362: makePlots(xnk, xlk, k, XNK, XLK, PK) after iteration k, through the library's on_step
    hook: pages <= k are final, later pages still NaN."""
    c = cases.radio_case(8, 6, 24, seed=3, N_K=3)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    seen = []

    def makePlots(xnk, xlk, k, XNK, XLK, PK):
        seen.append((k, xnk.copy(), xlk.copy(), bool(np.all(np.isfinite(XNK[:, :, :k + 1]))), bool(np.all(np.isnan(XNK[:, :, k + 1:]))),
                     bool(np.all(np.isnan(PK[:, :, k + 1:])))))

    XNK, XLK, PK = rbpf.particleSmoother(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0,
                                         c["Q"], R, c["N_P"], 3, c["dt"], False, makePlots, rng=cases.device_rng(rbpf, c))
    assert [s[0] for s in seen] == [0, 1, 2]
    for k, xnk, xlk, fin, nan_later, nan_pk in seen:
        assert fin and nan_later and nan_pk
        np.testing.assert_array_equal(xnk, XNK[:, :, k])
        np.testing.assert_array_equal(xlk, XLK[:, k])

    def bad(*a):
        raise KeyError("boom")
    with pytest.raises(KeyError):
        rbpf.particleSmoother(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                              c["N_P"], 2, c["dt"], False, bad, rng=cases.device_rng(rbpf, c))


@pytest.mark.parametrize("lazy_depth", [2, 3])
@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 7, 8, 130), ("mag", 6, 7, 256), ("mag", 6, 7, 512), ("radio", 9, 9, 128),
                                            ("radio", 8, 8, 140)])
def test_information_form_smoother_with_the_lazy_covariance_update(rbpf, kind, N_P, N_T, m, lazy_depth):
    """rbpf_options.lazy_depth in particleSmootherInformationForm: the stored covariances are rewritten every C-th step only,
    the pending rank-ny downdates of :331 applied on the fly (at nLin = 515 with the pending factors passing through the
    blocked LDS stage).  Same algebra -> the oracle's ancestors bit for bit and its numbers to 1e-9."""
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=53, N_K=3)
    ref = cases.oracle_smoother(c, True)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    out = rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0,
                                               c["Q"], R, N_P, 3, c["dt"], rng=cases.device_rng(rbpf, c), extras=True,
                                               lazy_depth=lazy_depth)
    check(ref, out, 3)


@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 8, 7, 16), ("mag", 12, 9, 130), ("radio", 9, 8, 24), ("radio", 14, 10, 128)])
def test_device_covariance_and_information_form_agree(rbpf, kind, N_P, N_T, m):
    """Known answer that needs no restatement of the reference: particleSmootherInformationForm.m:35-37 says the two smoothers
    are identical up to the form the weights are computed in.  On the DEVICE, with the same random numbers, the covariance form
    (stacked future innovations: MFMA GEMMs + a (n_y (T - t)) x (n_y (T - t)) Cholesky) and the information form (suffix sums of
    H' R^-1 H + an n x n Cholesky, other kernels altogether) must give the same normalised weights, ancestor probabilities,
    ancestor indices and outputs (quirk Q6: the unnormalised weights differ by a constant)."""
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=83, N_K=3)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    outs = []
    for f in (rbpf.particleSmoother, rbpf.particleSmootherInformationForm):
        outs.append(f(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R,
                      c["N_P"], c["N_K"], c["dt"], rng=cases.device_rng(rbpf, c), extras=True))
    (X1, L1, P1, e1), (X2, L2, P2, e2) = outs
    np.testing.assert_array_equal(e1["ai"][:, 1:], e2["ai"][:, 1:])
    np.testing.assert_array_equal(e1["ak"], e2["ak"])
    np.testing.assert_allclose(e1["w"], e2["w"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(e1["paNt"][1:, 1:], e2["paNt"][1:, 1:], rtol=0, atol=1e-9)
    np.testing.assert_allclose(X1, X2, rtol=0, atol=1e-12)
    assert rel(L1, L2) <= 1e-9 and rel(P1, P2) <= 1e-9
