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
