"""Synthetic SLAM data sets for the dense model families (host-side, once per run).

Product counterpart of examples/slam-dense-radio/generateData_dense.m (:181-213 bean_6D trajectory,
:216-257 field draw in a 2000-function reduced-rank basis, :294-325 odometry noise by running
dynModel forward) and tools/gp_rnd_scalar_potential_fast.m:42-102, scaled to an arbitrary number of
time steps N_T by spreading the three laps over N_T points (the box, hence LL and NN, stays the
reference's).  Implemented with per-axis sin/cos tables instead of the reference's m x d loops.
MATLAB's randn is replaced by a seeded numpy stream; the test grid the reference also evaluates for
plotting (:233-236) is not generated.
"""
from __future__ import annotations

import math

import numpy as np

from .host import domain_cartesian_dx, eigenval


def _qmul(q, p):
    """qLeft(q) * p (tools/qLeft.m:30-35)."""
    return np.array([q[0] * p[0] - q[1] * p[1] - q[2] * p[2] - q[3] * p[3],
                     q[1] * p[0] + q[0] * p[1] - q[3] * p[2] + q[2] * p[3],
                     q[2] * p[0] + q[3] * p[1] + q[0] * p[2] - q[1] * p[3],
                     q[3] * p[0] - q[2] * p[1] + q[1] * p[2] + q[0] * p[3]])


def _expq(phi):
    """tools/expq.m:22-31."""
    mag = math.sqrt(float(phi[0] ** 2 + phi[1] ** 2 + phi[2] ** 2))
    den = mag + (1.0 if mag == 0.0 else 0.0)
    eq = np.array([math.cos(mag), phi[0] / den * math.sin(mag), phi[1] / den * math.sin(mag),
                   phi[2] / den * math.sin(mag)])
    return -eq if eq[0] < 0 else eq


def _quat2rmat(q):
    """tools/quat2rmat.m:27-33."""
    q0, q1, q2, q3 = q
    return np.array([[q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3, 2 * q1 * q2 - 2 * q0 * q3, 2 * q1 * q3 + 2 * q0 * q2],
                     [2 * q1 * q2 + 2 * q0 * q3, q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3, 2 * q2 * q3 - 2 * q0 * q1],
                     [2 * q1 * q3 - 2 * q0 * q2, 2 * q2 * q3 + 2 * q0 * q1, q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3]])


def _axis_tables(NN, L, x):
    """S[a][pt, k], C[a][pt, k]: sin / cos of pi k (x_a + L_a) / (2 L_a), k = 0..kmax_a."""
    S, Cc = [], []
    for a in range(NN.shape[1]):
        k = np.arange(0, int(NN[:, a].max()) + 1, dtype=np.float64)
        arg = np.pi * k[None, :] * (x[:, a:a + 1] + L[a]) / (2.0 * L[a])
        S.append(np.sin(arg))
        Cc.append(np.cos(arg))
    return S, Cc


def _device_field(kind, NN, L, x, foo, chunk=512):
    """The reduced-rank field at the points x from the DEVICE basis kernels: the simulation basis (2000
    functions in the reference) is just another measModel of the family -- H(x) @ foo with the identity attitude -- so the
    a7 / a8 kernels evaluate it (rbpf_meas_model); the host only draws the coefficients and adds the noise."""
    from . import host
    if kind == "mag":
        mdl = host.DenseMagModel(NN, L)
        xn = np.zeros((7, x.shape[0]))
        xn[0:3], xn[3] = x.T, 1.0                               # q = [1 0 0 0]: Rnb = I
    else:
        mdl = host.DenseRadioModel(NN, L)
        xn = np.zeros((3, x.shape[0]))
        xn[0:2] = x.T
    out = []
    for i0 in range(0, x.shape[0], chunk):                        # [npts x ny x nLin] in chunks: 48 KB per point at m = 2000
        dy = mdl.measModel(xn[:, i0:i0 + chunk])
        out.append(dy @ foo)
    return np.concatenate(out, axis=0)


def curl_free_field_draw(x, m, LL, theta, rs, device=False):
    """Gradient-field draw of tools/gp_rnd_scalar_potential_fast.m:42-102 at points x [npts x 3]:
    returns (df, y) = (true field, field + sqrt(sigma2) * noise).  device=True evaluates the m basis functions with the HIP
    measurement-model kernel instead of the numpy tables (same numbers to ~1e-12)."""
    LL = np.asarray(LL, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64) - LL.mean(axis=0)
    L, NN = domain_cartesian_dx(m, 3, (LL.max(axis=0) - LL.min(axis=0))[None, :] / 2.0)
    lam = eigenval(NN, L)
    linSigma2, lengthScale, magnSigma2, sigma2 = (float(t) for t in np.asarray(theta).ravel())
    Sse = magnSigma2 * math.sqrt(2 * math.pi) ** 3 * lengthScale ** 3 * np.exp(-lam * lengthScale ** 2 / 2)
    foo = np.sqrt(np.concatenate(([linSigma2] * 3, Sse))) * rs.standard_normal(m + 3)
    if device:
        df = _device_field("mag", NN, L, x, foo)
        return df, df + math.sqrt(sigma2) * rs.standard_normal(df.shape)
    S, Cc = _axis_tables(NN, L, x)
    amp = 1.0 / np.sqrt(L)
    dfac = np.pi * NN / (2.0 * L * np.sqrt(L))                      # [m x 3]
    s = [S[a][:, NN[:, a]] * amp[a] for a in range(3)]              # [npts x m]
    c = [Cc[a][:, NN[:, a]] * dfac[:, a][None, :] for a in range(3)]
    w = foo[3:]
    df = np.column_stack(((c[0] * s[1] * s[2]) @ w + foo[0],
                          (s[0] * c[1] * s[2]) @ w + foo[1],
                          (s[0] * s[1] * c[2]) @ w + foo[2]))
    y = df + math.sqrt(sigma2) * rs.standard_normal(df.shape)
    return df, y


def bean_6D(N_T, Q, theta, dt, seed=1, m_sim=2000, nLL=2, laps=3, a=15.0, device=False):
    """generateData_dense.m 'bean_6D' scaled to N_T samples -> dict(dx, initState, y, LL, pos, quat).  device=True: the field
    draw's m_sim basis functions are evaluated by the HIP kernels."""
    rs = np.random.RandomState(seed)
    psi = np.linspace(0.0, laps * np.pi, N_T)
    r = a * np.sin(psi) ** 3 + a * np.cos(psi) ** 3
    u, v = r * np.cos(psi) - 0.3, r * np.sin(psi) - 0.3
    th = np.arctan2(np.diff(v), np.diff(u))
    th = np.concatenate((th, th[-1:]))
    pos = np.vstack((u, v, np.zeros_like(u)))
    # rmat2quat of the planar rotation [c s 0; -s c 0; 0 0 1]: expq([0;0;-th]/2)  (tools/rmat2quat.m:35-36)
    quat = np.stack([_expq(np.array([0.0, 0.0, -t / 2.0])) for t in th], axis=0)
    pos = pos - 0.5 * (pos.min(axis=1) + pos.max(axis=1))[:, None]
    initState = np.concatenate((pos[:, 0], quat[0]))
    dPos = np.diff(pos.T, axis=0)
    qc = quat[:-1] * np.array([1.0, -1.0, -1.0, -1.0])
    dQuat = np.stack([_qmul(qc[i], quat[i + 1]) for i in range(N_T - 1)], axis=0)
    ls = float(np.asarray(theta).ravel()[1])
    LL = np.array([[pos[0].min() - nLL * ls, pos[1].min() - nLL * ls, -nLL * ls],
                   [pos[0].max() + nLL * ls, pos[1].max() + nLL * ls, nLL * ls]])
    _, yn = curl_free_field_draw(pos.T, m_sim, LL, theta, rs, device=device)
    y = np.stack([_quat2rmat(quat[i]).T @ yn[i] for i in range(N_T)], axis=0)
    # noisy odometry: push the truth through dynModel (run_dense3D_magfield.m:301-308)
    Q = np.asarray(Q, dtype=np.float64)
    Lp = np.linalg.cholesky(dt * Q[0:3, 0:3])
    La = np.linalg.cholesky(dt * Q[3:6, 3:6])
    x = np.zeros((N_T, 7))
    dQn = np.zeros((N_T - 1, 4))
    x[0] = initState
    zo = rs.standard_normal((N_T - 1, 6))
    for i in range(1, N_T):
        x[i, 0:3] = x[i - 1, 0:3] + dPos[i - 1] + Lp @ zo[i - 1, 0:3]
        dQn[i - 1] = _qmul(dQuat[i - 1], _expq(La @ zo[i - 1, 3:6]))
        x[i, 3:7] = _qmul(x[i - 1, 3:7], dQn[i - 1])
    dx = np.hstack((np.diff(x[:, 0:3], axis=0), dQn))
    return dict(dx=dx, initState=initState, y=y, LL=LL, pos=pos, quat=quat)


def scalar_field_draw(x, m, LL, theta, rs, device=False):
    """Scalar-field draw of tools/gp_rnd_SE1D_fast.m:44-85 at points x [npts x 2]: returns (f, y).  device=True: basis on the
    HIP kernels."""
    LL = np.asarray(LL, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64) - LL.mean(axis=0)
    L, NN = domain_cartesian_dx(m, 2, (LL.max(axis=0) - LL.min(axis=0))[None, :] / 2.0)
    lam = eigenval(NN, L)
    lengthScale, magnSigma2, sigma2 = (float(t) for t in np.asarray(theta).ravel())
    k = magnSigma2 * math.sqrt(2 * math.pi) ** 2 * lengthScale ** 2 * np.exp(-lam * lengthScale ** 2 / 2)
    foo = np.sqrt(k) * rs.standard_normal(m)
    noise = rs.standard_normal(x.shape[0])
    if device:
        f = _device_field("radio", NN, L, x, foo)
        return f, f + math.sqrt(sigma2) * noise
    S, _ = _axis_tables(NN, L, x)
    amp = 1.0 / np.sqrt(L)
    f = ((S[0][:, NN[:, 0]] * amp[0]) * (S[1][:, NN[:, 1]] * amp[1])) @ foo
    return f, f + math.sqrt(sigma2) * noise


def radio_Q(N_T, traj="line_3D"):
    """Time-varying heading noise of run_dense2D_withHeading.m:69-73 (line) / :83-86 (square), [1 x 1 x N_T]."""
    Q = 1e-6 * np.ones(N_T)
    if traj == "line_3D":
        Q[N_T // 2 - 1] = 0.3 ** 2
    else:
        for j in range(3):
            Q[N_T // 4 + (N_T // 4) * j - 1] = 0.1 ** 2
    return Q.reshape(1, 1, N_T)


def planar_heading(N_T, Q, theta, dt, seed=1, m_sim=2000, nLL=4, traj="line_3D", device=False):
    """generateData_dense.m 'line_3D' (:113-132) / 'square_3D' (:101-112) with N = N_T points, scalar field
    (:258-290), odometry noise on the heading by running dynModel forward (:310-323;
    run_dense2D_withHeading.m:75-76) -> dict(dx, initState, y [N_T x 1], LL, pos)."""
    rs = np.random.RandomState(seed)
    N = int(N_T)
    if traj == "line_3D":
        pos = np.vstack((np.zeros(N), np.concatenate((np.linspace(0, 3, N // 2), np.linspace(3, 0, N - N // 2)))))
    elif traj == "square_3D":
        q = N // 4
        pos = np.vstack((np.concatenate((np.zeros(q), np.linspace(0, 2, q), 2 * np.ones(q), np.linspace(2, 0, N - 3 * q))),
                         np.concatenate((np.linspace(0, 2, q), 2 * np.ones(q), np.linspace(2, 0, q), np.zeros(N - 3 * q)))))
    else:
        raise ValueError("traj must be 'line_3D' or 'square_3D'")
    pos = pos - pos.mean(axis=1, keepdims=True)
    initState = np.concatenate((pos[:, 0], [0.0]))
    dx = np.hstack((np.diff(pos.T, axis=0), np.zeros((N - 1, 1))))
    ls = float(np.asarray(theta).ravel()[0])
    LL = np.array([[pos[0].min() - nLL * ls, pos[1].min() - nLL * ls],
                   [pos[0].max() + nLL * ls, pos[1].max() + nLL * ls]])
    _, y = scalar_field_draw(pos.T, m_sim, LL, theta, rs, device=device)
    Q = np.asarray(Q, dtype=np.float64).reshape(1, 1, -1)
    dtv = np.broadcast_to(np.asarray(dt, dtype=np.float64).ravel(), (1,)) if np.ndim(dt) == 0 else np.asarray(dt)
    zo = rs.standard_normal(N - 1)
    th = np.zeros(N)
    for i in range(1, N):
        q = Q[0, 0, (i - 1) if Q.shape[2] > 1 else 0]
        h = dtv[(i - 1) if dtv.size > 1 else 0]
        th[i] = th[i - 1] + dx[i - 1, 2] + math.sqrt(h * q) * zo[i - 1]
    dxn = np.hstack((dx[:, 0:2], np.diff(th)[:, None]))                    # :319
    return dict(dx=dxn, initState=initState, y=y.reshape(-1, 1), LL=LL, pos=pos)


def sparse_visual_load(mat_path, seed=42, posVar=0.04 ** 2, posBias=0.01, angleVar=(0.001 ** 2) ** 2, obsStd=0.01,
                       guessMapVar=1.0, initMapVar=4.0 ** 2, noiseVar=0.1 ** 2, N_P=100, N_T=None):
    """examples/slam-sparse-visual/load_data.m:55-89 + the problem set-up of pfslam.m:84-94 on the reference's data file
    `curve-x2.mat` (the only data fixture the reference ships): noisy odometry u from the true path, noisy observations
    Y = Yclean + 0.01*randn (NaN = landmark not in view), per-particle initial maps, priors and noise covariances.
    MATLAB's randn stream is replaced by a seeded numpy stream.  N_T: use only the first N_T steps.
    -> dict(model args for particleFilter / particleSmoother, plus ground truth)."""
    import scipy.io as sio
    d = sio.loadmat(mat_path)
    rs = np.random.RandomState(seed)
    p, th, mp = d["p"], d["th"].ravel(), d["map"]
    Yclean = d["Yclean"]
    T = p.shape[1] if N_T is None else int(N_T)
    p, th, Yclean = p[:, :T], th[:T], Yclean[:, :T]
    dPos = np.diff(p, axis=1)                                                 # :71
    dTheta = np.diff(np.unwrap(th))                                           # :72
    u = np.vstack((dPos, dTheta[None, :])).T                                  # :75
    u[:, 0:2] = u[:, 0:2] + math.sqrt(posVar) * rs.standard_normal((T - 1, 2)) + posBias    # :78
    u[:, 2] = u[:, 2] + math.sqrt(angleVar) * rs.standard_normal(T - 1)       # :79
    Y = Yclean + obsStd * rs.standard_normal(Yclean.shape)                    # :82
    nLand = mp.shape[1]
    x0_lin = mp.T.reshape(-1)[:, None] + math.sqrt(guessMapVar) * rs.standard_normal((2 * nLand, N_P))   # pfslam.m:91
    return dict(odometry=u, y=Y.T, x0_nonLin=np.concatenate((p[:, 0], th[:1])), x0_lin=x0_lin,
                P0_lin=initMapVar * np.eye(2 * nLand), Q=np.diag([0.1 ** 2, 0.1 ** 2, 0.001 ** 2]),
                R=noiseVar * np.eye(nLand), dt=1.0, N_P=N_P, nLand=nLand, map=mp, groundTruth=np.vstack((p, th[None, :])),
                cam=(1.5, 0.0, 1.0))
