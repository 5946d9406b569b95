"""Known-answer problem for CPF-AS (particleSmoother.m:88-366, particleSmootherInformationForm.m:98-362) that needs no
restatement of the reference: a conditionally linear model which is jointly linear-Gaussian, so the exact smoothing moments
come from a Kalman filter + Rauch-Tung-Striebel smoother.

  non-linear state  x_t (scalar):  x_1 = x0,  x_{t+1} = x_t + u_t + w_t,  w_t ~ N(0, dt*q)          (additive dynModel)
  linear state      z = [z1, z2]:  z1 ~ N(1, 1e-14) (pinned), z2 = b ~ N(0, sb2) (an unknown measurement bias)
  measurement       y_t = [x_t 1] z + e_t = x_t + b + e_t,  e_t ~ N(0, r)                               (dense, ny = 1)

The smoothers' outputs after burn-in: XNK(1,t,k) ~ p(x_t | y_{1:T}); XLK(:,k), PK(:,:,k) = the map posterior given draw k, so
E[b | y] = mean_k XLK(2,k) and Var[b | y] = mean_k PK(2,2,k) + var_k XLK(2,k) (law of total variance).
Used on the oracle (CPU, tests/test_oracle_identities.py) and on the HIP smoothers (tests/test_gpu_headline_parity.py)."""
import numpy as np


def problem(T=12, q=0.09, r=0.04, sb2=0.25, dt=1.0, x1=0.3, b_true=0.4, seed=12345):
    rs = np.random.RandomState(seed)
    u = 0.2 * rs.standard_normal(T - 1)
    xs = x1 + np.concatenate(([0.0], np.cumsum(u + np.sqrt(q) * rs.standard_normal(T - 1))))
    y = xs + b_true + np.sqrt(r) * rs.standard_normal(T)
    return dict(T=T, q=q, r=r, sb2=sb2, dt=dt, x1=x1, u=u, yv=y, odometry=u.reshape(-1, 1), y=y.reshape(-1, 1),
                x0_nonLin=np.array([x1]), x0_lin=np.array([1.0, 0.0]), P0_lin=np.diag([1e-14, sb2]), Q=np.array([[q]]),
                R=np.array([[r]]))


def measModel(xn):
    """dy [N x nLin] for ny = 1 (particleFilter.m:124,139): H_i = [x_i, 1]."""
    X = np.asarray(xn).reshape(1, -1)
    return np.stack((X[0], np.ones(X.shape[1])), axis=1)


def rts(p):
    """Kalman filter + RTS smoother of s_t = [x_t, b].  Returns smoothed means [T, 2] and covariances [T, 2, 2]."""
    y, u, T = p["yv"], p["u"], p["T"]
    Hm = np.array([[1.0, 1.0]])
    Qm = np.diag([p["dt"] * p["q"], 0.0])
    mp, Pp, mf, Pf = np.zeros((T, 2)), np.zeros((T, 2, 2)), np.zeros((T, 2)), np.zeros((T, 2, 2))
    m, P = np.array([p["x1"], 0.0]), np.diag([0.0, p["sb2"]])
    for t in range(T):
        if t > 0:
            m = m + np.array([u[t - 1], 0.0])
            P = P + Qm
        mp[t], Pp[t] = m, P
        S = (Hm @ P @ Hm.T)[0, 0] + p["r"]
        K = (P @ Hm.T)[:, 0] / S
        m = m + K * (y[t] - (Hm @ m)[0])
        P = P - np.outer(K, K) * S
        mf[t], Pf[t] = m, P
    ms, Ps = mf.copy(), Pf.copy()
    for t in range(T - 2, -1, -1):
        G = Pf[t] @ np.linalg.pinv(Pp[t + 1])
        ms[t] = mf[t] + G @ (ms[t + 1] - mp[t + 1])
        Ps[t] = Pf[t] + G @ (Ps[t + 1] - Pp[t + 1]) @ G.T
    return ms, Ps


def check_moments(p, XNK, XLK, PK, burn, n_se, var_lo, var_hi):
    """Means within n_se standard errors (effective sample size (N_K - burn) / 4), variances within [var_lo, var_hi] of the
    RTS values; the pinned coefficient stays pinned; the conditional bias mean is anti-correlated with x_T (y sees the sum)."""
    ms, Ps = rts(p)
    X = XNK[0, :, burn:]
    ess = X.shape[1] / 4.0
    assert np.all(X[0] == p["x1"])                                       # every trajectory starts at x0_nonLin
    for t in range(1, p["T"]):
        se = np.sqrt(Ps[t, 0, 0] / ess)
        assert abs(X[t].mean() - ms[t, 0]) <= n_se * se, (t, X[t].mean(), ms[t, 0], se)
        assert var_lo <= X[t].var() / Ps[t, 0, 0] <= var_hi, (t, X[t].var(), Ps[t, 0, 0])
    bk, vk = XLK[1, burn:], PK[1, 1, burn:]
    assert abs(bk.mean() - ms[-1, 1]) <= n_se * np.sqrt(Ps[-1, 1, 1] / ess), (bk.mean(), ms[-1, 1])
    assert var_lo <= (vk.mean() + bk.var()) / Ps[-1, 1, 1] <= var_hi, (vk.mean() + bk.var(), Ps[-1, 1, 1])
    assert np.max(np.abs(XLK[0, burn:] - 1.0)) < 1e-6
    assert np.corrcoef(X[-1], bk)[0, 1] < 0 and Ps[-1, 0, 1] < 0
