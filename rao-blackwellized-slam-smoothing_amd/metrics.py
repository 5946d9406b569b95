"""Acceptance metrics of the reference's example scripts (host-side numpy).

`calc_rmses` restates examples/slam-sparse-visual/calc_rmses.m:35-55: the estimated map is aligned with the true map
by a Procrustes transformation (MATLAB's `procrustes`, scaling and reflection allowed) and the same transformation
is applied to the estimated trajectory before the position RMSE is taken."""
from __future__ import annotations

import numpy as np


def procrustes(X, Y):
    """MATLAB `[d, Z, tr] = procrustes(X, Y)` (default options): min || X - (b * Y * T + c) ||_F.
    X, Y: [n x p].  Returns (d, Z, dict(b=, T=, c=))."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    muX, muY = X.mean(axis=0), Y.mean(axis=0)
    X0, Y0 = X - muX, Y - muY
    normX, normY = np.sqrt((X0 ** 2).sum()), np.sqrt((Y0 ** 2).sum())
    X0, Y0 = X0 / normX, Y0 / normY
    L, D, Mt = np.linalg.svd(X0.T @ Y0)
    T = Mt.T @ L.T
    traceTA = D.sum()
    b = traceTA * normX / normY
    d = 1.0 - traceTA ** 2
    Z = normX * traceTA * (Y0 @ T) + muX
    c = muX - b * (muY @ T)
    return d, Z, dict(b=b, T=T, c=c)


def calc_rmses(map_true, map_est, traj_true, traj_est):
    """calc_rmses.m:35-55.  map_* [nLand x 2], traj_* [T x >=2] -> (rmse_path, rmse_map)."""
    _, _, tr = procrustes(map_true, map_est)
    Z = tr["b"] * (np.asarray(traj_est)[:, 0:2] @ tr["T"]) + tr["c"]
    Zmap = tr["b"] * (np.asarray(map_est) @ tr["T"]) + tr["c"]
    rmse_path = float(np.sqrt(np.mean(np.sum((np.asarray(traj_true)[:, 0:2] - Z) ** 2, axis=1))))
    rmse_map = float(np.sqrt(np.mean(np.sum((np.asarray(map_true) - Zmap) ** 2, axis=1))))
    return rmse_path, rmse_map


def quat2euler(q):
    """tools/quat2euler.m:26-34 (degrees): q [n x 4] or [4] -> [n x 3] / [3]."""
    q = np.asarray(q, dtype=np.float64)
    single = q.ndim == 1
    q = q.reshape(-1, 4)
    q0, q1, q2, q3 = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    e = 180.0 / np.pi * np.column_stack((np.arctan2(2 * q2 * q3 - 2 * q0 * q1, 2 * q0 ** 2 + 2 * q3 ** 2 - 1),
                                         -np.arcsin(np.clip(2 * q1 * q3 + 2 * q0 * q2, -1.0, 1.0)),
                                         np.arctan2(2 * q1 * q2 - 2 * q0 * q3, 2 * q0 ** 2 + 2 * q1 ** 2 - 1)))
    return e[0] if single else e


def rmse_dense_mag(pos_true, quat_true, traj):
    """The acceptance numbers of examples/slam-dense-mag/run_dense3D_magfield.m:155-183 (filter) / :216-237 (smoother) for one
    estimated trajectory traj [7 x T] (position rows 0..2, quaternion rows 3..6): per-axis RMS position error after a
    Procrustes alignment of the estimated path with the true one, and per-axis RMS orientation error
    quat2euler(qLeft(q_est) * qInv(q_true)) in degrees.  pos_true [3 x T], quat_true [T x 4] -> (rmse_pos [3], rmse_ori [3])."""
    pos_true = np.asarray(pos_true, dtype=np.float64)
    traj = np.asarray(traj, dtype=np.float64)
    _, Z, _ = procrustes(pos_true.T, traj[0:3, :].T)                       # :160-161
    rmse_pos = np.sqrt(np.mean((pos_true.T - Z) ** 2, axis=0))            # rms(...) column-wise, :164-165
    qt = np.asarray(quat_true, dtype=np.float64).reshape(-1, 4)
    err = np.empty((traj.shape[1], 3))
    for ii in range(traj.shape[1]):                                        # :170-179
        q, p = traj[3:7, ii], qt[ii] * np.array([1.0, -1.0, -1.0, -1.0])   # qInv
        prod = np.array([q[0] * p[0] - q[1] * p[1] - q[2] * p[2] - q[3] * p[3],
                         q[1] * p[0] + q[0] * p[1] - q[3] * p[2] + q[2] * p[3],
                         q[2] * p[0] + q[3] * p[1] + q[0] * p[2] - q[1] * p[3],
                         q[3] * p[0] - q[2] * p[1] + q[1] * p[2] + q[0] * p[3]])
        err[ii] = quat2euler(prod)
    return rmse_pos, np.sqrt(np.mean(err ** 2, axis=0))
