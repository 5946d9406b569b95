"""EKF comparison baseline of examples/slam-dense-mag (ekf_dense.m:41-102 with the closures measModel_ekf / dynModel_ekf
of run_dense3D_magfield.m:281-299,310-316) -- SURVEY 8 (f3).

One Gaussian state [position(3); orientation deviation(3); map(m+3)], serial in time, so the recursion stays on the
host (numpy); the two pieces that touch the reduced-rank basis run through the device helper kernels of the C ABI:
the rotated basis gradient `Rnb' * dPhi` (rbpf_meas_model) and the basis Hessian (rbpf_jacobian_phi3d)."""
from __future__ import annotations

import numpy as np

from ._ffi import RBPFError, RBPF_ERR_CHOL_FAILED


def _qleft(q):
    q0, q1, q2, q3 = q
    return np.array([[q0, -q1, -q2, -q3], [q1, q0, -q3, q2], [q2, q3, q0, -q1], [q3, -q2, q1, q0]])     # tools/qLeft.m:30-35


def _expq(phi):
    mag = float(np.sqrt(phi @ phi))                                          # tools/expq.m:22-31
    den = mag + (1.0 if mag == 0.0 else 0.0)
    eq = np.concatenate(([np.cos(mag)], phi / den * np.sin(mag)))
    return -eq if eq[0] < 0 else eq


def _quat2rmat(q):
    q0, q1, q2, q3 = q                                                       # tools/quat2rmat.m:27-33
    return np.array([[q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3, 2 * q1 * q2 - 2 * q0 * q3, 2 * q1 * q3 + 2 * q0 * q2],
                     [2 * q1 * q2 + 2 * q0 * q3, q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3, 2 * q2 * q3 - 2 * q0 * q1],
                     [2 * q1 * q3 - 2 * q0 * q2, 2 * q2 * q3 + 2 * q0 * q1, q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3]])


def _mcross(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])                         # tools/mcross.m:33-37


def _chol_jitter(SS, jitter):
    try:
        return np.linalg.cholesky(SS)
    except np.linalg.LinAlgError:
        try:
            return np.linalg.cholesky(SS + jitter * np.eye(SS.shape[0]))                                 # ekf_dense.m:84-86
        except np.linalg.LinAlgError as exc:
            raise RBPFError(RBPF_ERR_CHOL_FAILED, "matrix must be positive definite") from exc


def measModel_ekf(model, LL, x, q):
    """run_dense3D_magfield.m:281-299 -> (yhat [3], dy [3 x (6 + nLin)])."""
    LL = np.asarray(LL, dtype=np.float64)
    xn = np.concatenate((x[0:3], q))
    RtdPhi = model.measModel(xn)[0]                                          # Rnb' * dPhi  [3 x nLin]  (device)
    Rnb = _quat2rmat(q)
    yhat = RtdPhi @ x[6:]                                                    # :290
    J = model.JacobianPhi3D(x[0:3], LL[0], LL[1])[:, :, :, 0]                # [3 x 3 x m]  (device), :292-294
    J3 = np.tensordot(J, x[9:], axes=([2], [0]))
    dy = np.zeros((3, RtdPhi.shape[1] + 6))
    dy[:, 0:3] = Rnb.T @ J3                                                  # :296
    dy[:, 3:6] = Rnb.T @ _mcross(Rnb @ yhat)                                 # :297  (dPhi*x(7:end) = Rnb * yhat)
    dy[:, 6:] = RtdPhi                                                       # :298
    return yhat, dy


def ekf_dense(model, LL, odometry, y, x0, q0, P0, Q, R, dt):
    """ekf_dense.m:41-102 for a DenseMagModel family object -> (xf_traj, qnb_traj, Pf_traj)."""
    y = np.atleast_2d(np.asarray(y, dtype=np.float64))
    odometry = np.atleast_2d(np.asarray(odometry, dtype=np.float64))
    xf = np.asarray(x0, dtype=np.float64).ravel().copy()
    Pf = np.asarray(P0, dtype=np.float64).copy()
    q_nb = np.asarray(q0, dtype=np.float64).ravel().copy()
    R = np.atleast_2d(np.asarray(R, dtype=np.float64))
    nS, N_T = xf.size, y.shape[0]
    Q = np.asarray(Q, dtype=np.float64)
    Qp = Q if Q.ndim == 3 else np.repeat(Q[:, :, None], max(N_T - 1, 1), axis=2)                         # :47-49
    dtv = np.atleast_1d(np.asarray(dt, dtype=np.float64)).ravel()
    if dtv.size == 1:
        dtv = dtv[0] * np.ones(max(N_T - 1, 1))                                                          # :52-54
    xf_traj, Pf_traj, qnb_traj = np.full((nS, N_T), np.nan), np.full((nS, nS, N_T), np.nan), np.full((4, N_T), np.nan)
    xp, Pp = xf, Pf
    for t in range(N_T):
        if t != 0:                                                           # :69-74 with dynModel_ekf :310-316
            dx = odometry[t - 1, :]
            xp = xf.copy()
            xp[0:3] = xf[0:3] + dx[0:3]
            q_nb = _qleft(q_nb) @ dx[3:7]
            G = np.zeros((nS, 6))
            G[0:3, 0:3] = np.eye(3)
            G[3:6, 3:6] = _quat2rmat(q_nb)
            Pp = Pf + G @ (dtv[t - 1] * Qp[:, :, t - 1]) @ G.T               # F = I
        yhat, dy = measModel_ekf(model, LL, xp, q_nb)                        # :78
        e = y[t, :] - yhat
        SS = dy @ Pp @ dy.T + R
        cS = _chol_jitter(SS, 1e-3)
        Mx = np.linalg.solve(cS, dy).T
        Mx = np.linalg.solve(cS.T, Mx.T).T
        K = Pp @ Mx                                                          # :87
        xf = xp + K @ e
        Pf = Pp - K @ SS @ K.T
        Pf = 0.5 * (Pf + Pf.T)                                               # :92
        q_nb = _qleft(_expq(xf[3:6] / 2.0)) @ q_nb                           # :95
        xf[3:6] = 0.0
        xf_traj[:, t], Pf_traj[:, :, t], qnb_traj[:, t] = xf, Pf, q_nb
    return xf_traj, qnb_traj, Pf_traj
