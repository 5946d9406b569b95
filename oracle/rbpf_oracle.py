"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

A numpy fp64, line-by-line CPU restatement of the MATLAB reference
(manonkok/Rao-Blackwellized-SLAM-smoothing) for the Rao-Blackwellized particle
filter / conditional particle smoother hot path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

PARITY STATUS: **parity unpinned by the reference itself.**  The reference ships
no tests / golden vectors / result checkpoints and neither MATLAB nor Octave is
available in the build container or on the GPU box, so this file pins the build
to *its reading of* the cited .m files.  It is anchored by independent checks in
tests/test_oracle_*.py: covariance-form vs information-form smoother weight
equality (particleSmootherInformationForm.m:35-37), sequential Kalman vs batch
reduced-rank GP posterior (tools/gp_scalar_potential_fast.m:190-192),
quaternion / basis-derivative identities, and the frequency check sketched in
tools/sample.m:36-64.

Conventions
-----------
* All `file:line` citations are relative to /root/reference/.
* Arrays follow the MATLAB shapes of the reference (e.g. xn is [nNonLin x N_P]);
  particle / time indices are 0-based here, 1-based in MATLAB.  Ancestor indices
  returned in traces are 0-based.
* Randomness is *injected*, never drawn inside: every stochastic call site of the
  reference (rand in tools/sample.m:31; randn inside the dynModel closures) reads
  from replay buffers (see `ReplayRNG`).  This is what makes seed-exact parity with
  the HIP path testable without re-implementing MATLAB's ziggurat randn.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np

LOG2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------
# tools/sample.m
# --------------------------------------------------------------------------------------
def sample(w: np.ndarray, u: float) -> int:
    """tools/sample.m:30-32 -- `wc = cumsum(w); u = rand; ind = sum(wc < u) + 1`.

    Returns the 0-based index (MATLAB's ind-1).  np.cumsum is a strict left-to-right
    fp64 running sum, like MATLAB's cumsum on a vector.  If u > wc[-1] the reference
    returns N+1 (an out-of-range index -> MATLAB indexing error); here that shows up
    as the value N, and callers raise IndexError when they index with it.
    """
    wc = np.cumsum(np.asarray(w, dtype=np.float64).ravel())
    return int(np.sum(wc < u))


# --------------------------------------------------------------------------------------
# tools/ quaternion algebra
# --------------------------------------------------------------------------------------
def mcross(v):
    """tools/mcross.m:33-37 (single vector branch)."""
    v = np.asarray(v, dtype=np.float64).ravel()
    return np.array([[0.0, -v[2], v[1]],
                     [v[2], 0.0, -v[0]],
                     [-v[1], v[0], 0.0]])


def mcross_batched(v):
    """tools/mcross.m:38-42 (N x 3 branch): reshape([0 v3 -v2 -v3 0 v1 v2 -v1 0]', 3, 3, N) -- the matrix is filled
    column by column via the unvec operation.  Returned here as [N,3,3]."""
    v = np.asarray(v, dtype=np.float64)
    N = v.shape[0]
    z = np.zeros(N)
    rows = np.column_stack((z, v[:, 2], -v[:, 1], -v[:, 2], z, v[:, 0], v[:, 1], -v[:, 0], z))   # N x 9
    M = rows.T.reshape((3, 3, N), order="F")                                                      # reshape(rows', 3, 3, N)
    return np.transpose(M, (2, 0, 1))


def qLeft_batched(q):
    """tools/qLeft.m:36-40 (N x 4 branch): [q0, -qv'; qv, q0*I + mcross(qv)] per page.  Returned as [N,4,4]."""
    q = np.asarray(q, dtype=np.float64)
    N = q.shape[0]
    out = np.empty((N, 4, 4))
    out[:, 0, 0] = q[:, 0]
    out[:, 0, 1:] = -q[:, 1:4]
    out[:, 1:, 0] = q[:, 1:4]
    out[:, 1:, 1:] = q[:, 0][:, None, None] * np.eye(3)[None] + mcross_batched(q[:, 1:4])
    return out


def qRight_batched(q):
    """tools/qRight.m:35-39 (N x 4 branch): [q0, -qv'; qv, multiprod(q0, I) - mcross(qv)] per page.  [N,4,4]."""
    q = np.asarray(q, dtype=np.float64)
    N = q.shape[0]
    out = np.empty((N, 4, 4))
    out[:, 0, 0] = q[:, 0]
    out[:, 0, 1:] = -q[:, 1:4]
    out[:, 1:, 0] = q[:, 1:4]
    out[:, 1:, 1:] = q[:, 0][:, None, None] * np.eye(3)[None] - mcross_batched(q[:, 1:4])
    return out


def expq(phi):
    """tools/expq.m:22-31 (vector branch, any(size(phi)==1)): sign flip on eq(1) < 0."""
    phi = np.asarray(phi, dtype=np.float64).ravel()
    mag = math.sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2])
    nphi = phi / (mag + (1.0 if mag == 0.0 else 0.0))
    eq = np.concatenate(([math.cos(mag)], nphi * math.sin(mag)))
    if eq[0] < 0:
        eq = -eq
    return eq


def expq_batched(phi):
    """tools/expq.m:33-37 (N x 3 branch): sign flip on eq(:,1) <= 0 (quirk Q7)."""
    phi = np.asarray(phi, dtype=np.float64)
    mag = np.sqrt(phi[:, 0] ** 2 + phi[:, 1] ** 2 + phi[:, 2] ** 2)
    nphi = phi / (mag + (mag == 0))[:, None]
    eq = np.column_stack((np.cos(mag), nphi * np.sin(mag)[:, None]))
    flip = eq[:, 0] <= 0
    eq[flip] = -eq[flip]
    return eq


def logq(q):
    """tools/logq.m:25-31 (vector branch).  q0 > 1 by rounding makes acos complex in
    MATLAB (quirk Q7); we clamp to 1 (documented deviation, rounding-level)."""
    q = np.asarray(q, dtype=np.float64).ravel().copy()
    if q[0] < 0:
        q = -q
    na = math.acos(min(q[0], 1.0))
    return na * q[1:4] / (math.sin(na) + (1.0 if na == 0.0 else 0.0))


def logq_batched(q):
    """tools/logq.m:32-35 (N x 4 branch): flip on <= 0."""
    q = np.asarray(q, dtype=np.float64).copy()
    flip = q[:, 0] <= 0
    q[flip] = -q[flip]
    na = np.arccos(np.minimum(q[:, 0], 1.0))
    return na[:, None] * q[:, 1:4] / (np.sin(na) + (na == 0))[:, None]


def qLeft(q):
    """tools/qLeft.m:30-35: [q0 -qv'; qv q0*I + [qv x]]."""
    q = np.asarray(q, dtype=np.float64).ravel()
    pL = np.empty((4, 4))
    pL[0, 0] = q[0]
    pL[0, 1:] = -q[1:4]
    pL[1:, 0] = q[1:4]
    pL[1:, 1:] = q[0] * np.eye(3) + mcross(q[1:4])
    return pL


def qRight(q):
    """tools/qRight.m:29-34: [q0 -qv'; qv q0*I - [qv x]]."""
    q = np.asarray(q, dtype=np.float64).ravel()
    qR = np.empty((4, 4))
    qR[0, 0] = q[0]
    qR[0, 1:] = -q[1:4]
    qR[1:, 0] = q[1:4]
    qR[1:, 1:] = q[0] * np.eye(3) - mcross(q[1:4])
    return qR


def qInv(q):
    """tools/qInv.m:27-31."""
    q = np.asarray(q, dtype=np.float64).copy()
    if q.ndim == 1:
        q[1:4] = -q[1:4]
        return q
    return np.column_stack((q[:, 0], -q[:, 1], -q[:, 2], -q[:, 3]))


def quat2rmat(q):
    """tools/quat2rmat.m:27-33 (single quaternion)."""
    q0, q1, q2, q3 = (float(x) for x in np.asarray(q, dtype=np.float64).ravel())
    return np.array([
        [q0 ** 2 + q1 ** 2 - q2 ** 2 - q3 ** 2, 2 * q1 * q2 - 2 * q0 * q3, 2 * q1 * q3 + 2 * q0 * q2],
        [2 * q1 * q2 + 2 * q0 * q3, q0 ** 2 - q1 ** 2 + q2 ** 2 - q3 ** 2, 2 * q2 * q3 - 2 * q0 * q1],
        [2 * q1 * q3 - 2 * q0 * q2, 2 * q2 * q3 + 2 * q0 * q1, q0 ** 2 - q1 ** 2 - q2 ** 2 + q3 ** 2]])


def quat2rmat_batched(q):
    """tools/quat2rmat.m:34-39 (N x 4 -> 3 x 3 x N); returned here as [N,3,3]."""
    q = np.asarray(q, dtype=np.float64)
    return np.stack([quat2rmat(qi) for qi in q], axis=0)


def rmat2quat_planar(th):
    """tools/rmat2quat.m:29-37 specialised to the planar rotations that
    examples/slam-dense-radio/generateData_dense.m:196-200 builds,
    R = [cos th, sin th, 0; -sin th, cos th, 0; 0 0 1]: logR(R) = [0;0;-th] (tools/logR.m:28-29,
    principal logm for |th| < pi) and q = expq(phi/2)."""
    return expq(np.array([0.0, 0.0, -th / 2.0]))


# --------------------------------------------------------------------------------------
# tools/domain_cartesian_dx.m
# --------------------------------------------------------------------------------------
def ndgridm(N):
    """tools/domain_cartesian_dx.m:174-218 -- index tuples, first axis slowest."""
    N = [int(x) for x in N]
    grids = np.meshgrid(*[np.arange(1, n + 1) for n in N], indexing="ij")
    return np.stack([g.ravel() for g in grids], axis=1).astype(np.float64)


def domain_cartesian_dx(m, d, LL):
    """tools/domain_cartesian_dx.m:26-51.

    Returns (L, NN): half-widths L [d] and the m x d index table NN (floats holding ints).
    LL may be 2 x d (lower; upper bounds) or already a length-d half-width vector.
    """
    LL = np.atleast_2d(np.asarray(LL, dtype=np.float64))
    if LL.shape[0] > 1:                                  # :27-29
        L = (LL.max(axis=0) - LL.min(axis=0)) / 2.0
    else:
        L = LL.ravel()
    N = np.ceil(m ** (1.0 / d) * L / L.min())            # :33
    NN = ndgridm(N)                                      # :36
    lam = eigenval(NN, L)                                # :40
    ind = np.argsort(lam, kind="stable")                 # :43 (MATLAB sort is stable)
    NN = NN[ind[:m], :]
    return L, NN


def eigenval(NN, L):
    """tools/domain_cartesian_dx.m:40 -- sum((pi*n./(2L)).^2,2)."""
    return np.sum((np.pi * NN / (2.0 * L)) ** 2, axis=1)


def eigenfun(NN, x, L):
    """tools/domain_cartesian_dx.m:84-93 (laplace_eig_cart_dirichlet). x: [Npts x d]."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    v = np.ones((x.shape[0], NN.shape[0]))
    for j in range(NN.shape[1]):
        for i in range(NN.shape[0]):
            v[:, i] = v[:, i] * 1.0 / math.sqrt(L[j]) * \
                np.sin(np.pi * NN[i, j] * (x[:, j] + L[j]) / (2.0 * L[j]))
    return v


def eigenfun_dx(NN, x, di, L):
    """tools/domain_cartesian_dx.m:142-170 (laplace_eig_cart_dirichlet_dx). di is 0-based."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    v = np.ones((x.shape[0], NN.shape[0]))
    for j in range(NN.shape[1]):
        if j == di:
            for i in range(NN.shape[0]):
                v[:, i] = v[:, i] * np.pi * NN[i, j] / (2.0 * L[j] * math.sqrt(L[j])) * \
                    np.cos(np.pi * NN[i, j] * (x[:, j] + L[j]) / (2.0 * L[j]))
        else:
            for i in range(NN.shape[0]):
                v[:, i] = v[:, i] * 1.0 / math.sqrt(L[j]) * \
                    np.sin(np.pi * NN[i, j] * (x[:, j] + L[j]) / (2.0 * L[j]))
    return v


def JacobianPhi3D(x, N_m, xl, xu, yl, yu, zl, zu, Indices):
    """tools/JacobianPhi3D.m:29-64.  x: [3 x Np] -> J [3 x 3 x N_m x Np]."""
    x = np.asarray(x, dtype=np.float64).reshape(3, -1)
    j = np.asarray(Indices, dtype=np.float64)
    Np = x.shape[1]
    J = np.zeros((3, 3, N_m, Np))
    a = np.array([xl, yl, zl], dtype=np.float64)
    b = np.array([xu, yu, zu], dtype=np.float64)
    f = np.zeros((N_m, 3))
    for d in range(3):
        f[:, d] = (np.pi * j[:, d]) / (b[d] - a[d])
    for i in range(Np):
        s = np.zeros((N_m, 3))
        c = np.zeros((N_m, 3))
        for d in range(3):
            core = np.pi * j[:, d] * (x[d, i] - a[d]) / (b[d] - a[d])
            mult = 1.0 / math.sqrt(0.5 * (b[d] - a[d]))
            s[:, d] = np.sin(core) * mult
            c[:, d] = np.cos(core) * mult
        J[0, 0, :, i] = -f[:, 0] ** 2 * s[:, 0] * s[:, 1] * s[:, 2]
        J[0, 1, :, i] = f[:, 0] * f[:, 1] * c[:, 0] * c[:, 1] * s[:, 2]
        J[0, 2, :, i] = f[:, 0] * f[:, 2] * c[:, 0] * s[:, 1] * c[:, 2]
        J[1, 0, :, i] = f[:, 1] * f[:, 0] * c[:, 0] * c[:, 1] * s[:, 2]
        J[1, 1, :, i] = -f[:, 1] ** 2 * s[:, 0] * s[:, 1] * s[:, 2]
        J[1, 2, :, i] = f[:, 1] * f[:, 2] * s[:, 0] * c[:, 1] * c[:, 2]
        J[2, 0, :, i] = f[:, 2] * f[:, 0] * c[:, 0] * s[:, 1] * c[:, 2]
        J[2, 1, :, i] = f[:, 2] * f[:, 1] * s[:, 0] * c[:, 1] * c[:, 2]
        J[2, 2, :, i] = -f[:, 2] ** 2 * s[:, 0] * s[:, 1] * s[:, 2]
    return J


# --------------------------------------------------------------------------------------
# Injected randomness
# --------------------------------------------------------------------------------------
class ReplayRNG:
    """Replay buffers standing in for MATLAB's global stream.

    U[k, t-1, i]      the `rand` consumed by sample() when drawing the ancestor of slot i at
                      step t (particleFilter.m:106, particleSmoother.m:134,149,241).  For smoother
                      iterations k>1 the slot N_P-1 entry is the single `rand` of
                      particleSmoother.m:241 (quirk Q9).
    Z[k, t-1, i, :]   the randn values consumed by dynModel for slot i at step t, in call order
                      (dense-mag: randn(3,1) for position then randn(3,1) for orientation,
                      run_dense3D_magfield.m:304-305; dense-radio: one randn,
                      run_dense2D_withHeading.m:76).  Unused for slot N_P-1 when k>1.
    Ufin[k]           the `rand` of `ak = sample(w)` (particleSmoother.m:346).
    The filter uses k = 0 only.
    """

    def __init__(self, U, Z, Ufin=None):
        self.U = np.asarray(U, dtype=np.float64)
        self.Z = np.asarray(Z, dtype=np.float64)
        if self.U.ndim == 2:
            self.U = self.U[None]
        if self.Z.ndim == 3:
            self.Z = self.Z[None]
        self.Ufin = None if Ufin is None else np.asarray(Ufin, dtype=np.float64).ravel()

    @staticmethod
    def draw(seed, N_K, N_T, N_P, nw):
        """Seeded numpy streams (NOT MATLAB's twister/ziggurat interleaving)."""
        rs = np.random.RandomState(seed)
        U = rs.random_sample((N_K, max(N_T - 1, 0), N_P))
        Z = rs.standard_normal((N_K, max(N_T - 1, 0), N_P, nw))
        Ufin = rs.random_sample(N_K)
        return ReplayRNG(U, Z, Ufin)


# --------------------------------------------------------------------------------------
# Model families (the closures of the example runners)
# --------------------------------------------------------------------------------------
@dataclass
class DenseMagModel:
    """examples/slam-dense-mag/run_dense3D_magfield.m closures: dynModel (:301-308),
    measModel (:265-279), dynResNorm (:202-203), GP prior (:83-107,122-131)."""
    NN: np.ndarray            # [m x 3]
    L: np.ndarray             # [3]
    nNonLin: int = 7
    ny: int = 3
    nw: int = 6
    sparse: bool = False

    @property
    def nLin(self):
        return self.NN.shape[0] + 3

    def dynModel(self, xn, dx, dt, Q, z):
        """run_dense3D_magfield.m:301-308; z = the six randn values in call order."""
        xn = np.asarray(xn, dtype=np.float64).ravel()
        dx = np.asarray(dx, dtype=np.float64).ravel()
        Lp = np.linalg.cholesky(dt * Q[0:3, 0:3])
        La = np.linalg.cholesky(dt * Q[3:6, 3:6])
        xpred_pos = xn[0:3] + dx[0:3] + Lp @ z[0:3]                     # :304
        dQuat = qLeft(dx[3:7]) @ expq(La @ z[3:6])                       # :305
        xpred_quat = qLeft(xn[3:7]) @ dQuat                              # :306
        return np.concatenate((xpred_pos, xpred_quat)), dQuat

    def measModel(self, xn):
        """run_dense3D_magfield.m:265-279.  xn: [7 x Npred] -> dy [Npred x 3 x (m+3)]."""
        xn = np.asarray(xn, dtype=np.float64).reshape(7, -1)
        Np = xn.shape[1]
        pos = xn[0:3, :].T
        dPhix = eigenfun_dx(self.NN, pos, 0, self.L)
        dPhiy = eigenfun_dx(self.NN, pos, 1, self.L)
        dPhiz = eigenfun_dx(self.NN, pos, 2, self.L)
        one, zero = np.ones((Np, 1)), np.zeros((Np, 1))
        dPhix = np.hstack((one, zero, zero, dPhix))                       # :270
        dPhiy = np.hstack((zero, one, zero, dPhiy))                       # :271
        dPhiz = np.hstack((zero, zero, one, dPhiz))                       # :272
        Rnb = quat2rmat_batched(xn[3:7, :].T)                             # :273
        dy = np.zeros((Np, 3, dPhix.shape[1]))
        for i in range(Np):
            dy[i] = Rnb[i].T @ np.vstack((dPhix[i], dPhiy[i], dPhiz[i]))  # :276-277
        return dy

    def dynResNorm(self, xnk, xni, dx, dt, Q):
        """run_dense3D_magfield.m:202-203 -- row vector r' / chol(dt*Q,'lower')."""
        xnk = np.asarray(xnk, dtype=np.float64).ravel()
        xni = np.asarray(xni, dtype=np.float64).ravel()
        dx = np.asarray(dx, dtype=np.float64).ravel()
        qres = qLeft(qLeft(qInv(dx[3:7])) @ qInv(xni[3:7])) @ xnk[3:7]
        r = np.concatenate((xnk[0:3] - xni[0:3] - dx[0:3], logq(qres)))
        Lq = np.linalg.cholesky(dt * Q)
        return mrdivide_row(r, Lq)


@dataclass
class DenseRadioModel:
    """examples/slam-dense-radio/run_dense2D_withHeading.m closures: dynModel (:75-76 / :89-90),
    dynResNorm (:77 / :91), measModel (:168)."""
    NN: np.ndarray            # [m x 2]
    L: np.ndarray             # [2]
    nNonLin: int = 3
    ny: int = 1
    nw: int = 1
    sparse: bool = False

    @property
    def nLin(self):
        return self.NN.shape[0]

    def dynModel(self, xn, dx, dt, Q, z):
        xn = np.asarray(xn, dtype=np.float64).ravel()
        dx = np.asarray(dx, dtype=np.float64).ravel()
        c, s = math.cos(xn[2]), math.sin(xn[2])
        Rot = np.array([[c, -s], [s, c]])
        Lq = np.linalg.cholesky(np.atleast_2d(dt * Q))
        xy = xn[0:2] + Rot.T @ dx[0:2]
        th = xn[2] + dx[2] + (Lq @ np.atleast_1d(z)[0:1])[0]
        return np.array([xy[0], xy[1], th]), None

    def measModel(self, xn):
        """:168 -- eigenfun(NN, xn(iPos,:)') -> [Npred x m] (2-D; ny = 1)."""
        xn = np.asarray(xn, dtype=np.float64).reshape(3, -1)
        return eigenfun(self.NN, xn[0:2, :].T, self.L)

    def dynResNorm(self, xnk, xni, dx, dt, Q):
        """:77 -- heading residual only (position mismatch ignored)."""
        r = np.array([xnk[2] - xni[2] - dx[2]])
        Lq = np.linalg.cholesky(np.atleast_2d(dt * Q))
        return mrdivide_row(r, Lq)


@dataclass
class SparseVisualModel:
    """examples/slam-sparse-visual closures (pfslam.m:81-82, psslam.m:91-92): 2-D pose (x, y, heading), nLand point
    landmarks seen by a 1-D pinhole camera (measurement.m:32-84); sparseFeatures = true, dynResNorm = []."""
    f: float = 1.5            # load_data.m:58-60
    fp: float = 0.0
    fw: float = 1.0
    nLand: int = 20
    nNonLin: int = 3
    nw: int = 3
    sparse: bool = True
    dynResNorm = None

    @property
    def ny(self):
        return self.nLand

    @property
    def nLin(self):
        return 2 * self.nLand

    def dynModel(self, xn, dx, dt, Q, z):
        """pfslam.m:81: xn + dx' + sqrt(dt*Q)*randn(3,1) -- sqrt is ELEMENT-wise."""
        xn = np.asarray(xn, dtype=np.float64).ravel()
        dx = np.asarray(dx, dtype=np.float64).ravel()
        return xn + dx + np.sqrt(dt * np.atleast_2d(Q)) @ np.asarray(z, dtype=np.float64).ravel(), None

    def measModel(self, xn, xl):
        """measurement([xn(1:3); xl], f, fp, fw, true) (measurement.m:32-84) -> (yhat [nLand], dy [nLand x 2 nLand])."""
        xn = np.asarray(xn, dtype=np.float64).ravel()
        mp = np.asarray(xl, dtype=np.float64).reshape(-1, 2).T           # reshape(x(4:end),2,[])  (:43)
        p, th = xn[0:2], xn[2]
        Rm = np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])   # :35
        K = np.array([[self.f, self.fp], [0.0, 1.0]])                    # :46
        u = K @ np.hstack((Rm.T, (-Rm.T @ p)[:, None])) @ np.vstack((mp, np.ones((1, mp.shape[1]))))   # :49
        y = u[0] / u[1]                                                  # :52
        div = (mp[1] * math.cos(th) - p[1] * math.cos(th) - mp[0] * math.sin(th) + p[0] * math.sin(th)) ** 2   # :61
        dym1 = (self.f * (mp[1] - p[1])) / div                           # :74
        dym2 = -(self.f * (mp[0] - p[0])) / div                          # :77
        dy = np.zeros((mp.shape[1], 2 * mp.shape[1]))
        idx = np.arange(mp.shape[1])
        dy[idx, 2 * idx] = dym1                                          # :80
        dy[idx, 2 * idx + 1] = dym2                                      # :81
        return y, dy                                                     # onlyLin (:84-86)


def mrdivide_row(r, Lq):
    """MATLAB `r' / Lq` for a row vector: solve x*Lq = r'  <=>  Lq' x' = r."""
    return np.linalg.solve(Lq.T, np.asarray(r, dtype=np.float64).ravel())


def dense_mag_prior(m, LL, theta):
    """run_dense3D_magfield.m:83-107,122-131: basis, spectral density, x0_lin, P0_lin, R."""
    L, NN = domain_cartesian_dx(m, 3, LL)
    lam = eigenval(NN, L)
    linSigma2, lengthScale, magnSigma2, sigma2 = (float(t) for t in theta)
    d = 3
    w = np.sqrt(lam)
    Sse = magnSigma2 * math.sqrt(2 * math.pi) ** d * lengthScale ** d * np.exp(-w ** 2 * lengthScale ** 2 / 2)
    k = np.concatenate(([linSigma2] * 3, Sse))
    x0_lin = np.zeros(m + 3)
    P0_lin = np.diag(k)
    R = sigma2 * np.eye(3)
    return DenseMagModel(NN=NN, L=L), x0_lin, P0_lin, R


def dense_radio_prior(m, LL, theta):
    """run_dense2D_withHeading.m:107-128,137-146."""
    L, NN = domain_cartesian_dx(m, 2, LL)
    lam = eigenval(NN, L)
    lengthScale, magnSigma2, sigma2 = (float(t) for t in theta)
    d = 2
    w = np.sqrt(lam)
    k = magnSigma2 * math.sqrt(2 * math.pi) ** d * lengthScale ** d * np.exp(-w ** 2 * lengthScale ** 2 / 2)
    return DenseRadioModel(NN=NN, L=L), np.zeros(m), np.diag(k), sigma2 * np.eye(1)


# --------------------------------------------------------------------------------------
# Shared pieces of the three estimators
# --------------------------------------------------------------------------------------
class CholeskyFailure(RuntimeError):
    """Second chol failure: MATLAB throws (particleFilter.m:147)."""


def _chol_lower_with_jitter(SS, jitter):
    """[cS,flag]=chol(SS,'lower'); if flag>0, cS=chol(SS+jitter*I,'lower') -- particleFilter.m:145-148."""
    try:
        return np.linalg.cholesky(SS)
    except np.linalg.LinAlgError:
        try:
            return np.linalg.cholesky(SS + jitter * np.eye(SS.shape[0]))
        except np.linalg.LinAlgError as exc:
            raise CholeskyFailure("matrix must be positive definite") from exc


def _expand_Q_dt(Q, dt, N_T):
    """particleFilter.m:74-82."""
    Q = np.asarray(Q, dtype=np.float64)
    if Q.ndim == 0:
        Q = Q.reshape(1, 1)
    if Q.ndim == 2:
        Q = np.repeat(Q[:, :, None], max(N_T - 1, 1), axis=2)
    dt = np.atleast_1d(np.asarray(dt, dtype=np.float64)).ravel()
    if dt.size == 1:
        dt = dt[0] * np.ones(max(N_T - 1, 1))
    return Q, dt


def _dy_of(dy, i):
    """squeeze(dy(i,:,:)) -- particleFilter.m:139.  [Npred x ny x n] -> [ny x n]; for a 2-D dy
    ([Npred x n], ny = 1) MATLAB's squeeze gives a 1 x n row."""
    if dy.ndim == 3:
        return dy[i]
    return dy[i][None, :]


def _importance_logw(yt, dyt, xl_i, P_i, R, jitter):
    """particleFilter.m:139-150 (dense branch)."""
    e = yt - dyt @ xl_i
    SS = dyt @ P_i @ dyt.T + R
    cS = _chol_lower_with_jitter(SS, jitter)
    v = np.linalg.solve(cS, e)
    return -np.sum(np.log(np.diag(cS))) - 0.5 * (v @ v) - 0.5 * e.size * LOG2PI


def _kalman_update(yt, dyt, xl_i, P_i, R, jitter):
    """particleFilter.m:184-198 (dense branch): K = P*((dyt'/cS')/cS); xl += K*e; P -= K*SS*K'."""
    yhat = dyt @ xl_i
    e = yt - yhat
    SS = dyt @ P_i @ dyt.T + R
    cS = _chol_lower_with_jitter(SS, jitter)
    M = np.linalg.solve(cS, dyt).T          # dyt'/cS'  ==  (cS \ dyt)'
    M = np.linalg.solve(cS.T, M.T).T        # (.)/cS    ==  (cS' \ (.)')'
    K = P_i @ M
    return xl_i + K @ e, P_i - K @ SS @ K.T, yhat, cS


def _sparse_innovation(model, yt, xn_i, xl_i, P_i, R):
    """particleFilter.m:129-137: EKF linearisation, innovation and its covariance restricted to the observed outputs."""
    yhat, dy = model.measModel(xn_i, xl_i)
    e = yt - yhat
    SS = dy @ P_i @ dy.T + R
    ind = ~np.isnan(yt)
    return e[ind], SS[np.ix_(ind, ind)], dy, ind, yhat


def _sparse_logw(model, yt, xn_i, xl_i, P_i, R, jitter):
    """particleFilter.m:129-150 (sparse branch)."""
    e, SS, _, _, _ = _sparse_innovation(model, yt, xn_i, xl_i, P_i, R)
    if e.size == 0:
        return 0.0
    cS = _chol_lower_with_jitter(SS, jitter)
    v = np.linalg.solve(cS, e)
    return -np.sum(np.log(np.diag(cS))) - 0.5 * (v @ v) - 0.5 * e.size * LOG2PI


def _sparse_update(model, yt, xn_i, xl_i, P_i, R, jitter):
    """particleFilter.m:165-181,197-198 (sparse branch)."""
    e, SS, dy, ind, yhat = _sparse_innovation(model, yt, xn_i, xl_i, P_i, R)
    if e.size == 0:
        return xl_i.copy(), P_i.copy(), yhat
    cS = _chol_lower_with_jitter(SS, jitter)
    dyo = dy[ind, :]
    M = np.linalg.solve(cS, dyo).T                                   # dy(ind,:)'/cS'
    M = np.linalg.solve(cS.T, M.T).T                                 # (.)/cS
    K = P_i @ M
    return xl_i + K @ e, P_i - K @ SS @ K.T, yhat


def _normalise(logw):
    """particleFilter.m:154-156."""
    c = np.max(logw)
    lse = c + math.log(np.sum(np.exp(logw - c)))
    return np.exp(logw - lse)


# --------------------------------------------------------------------------------------
# src/particleFilter.m
# --------------------------------------------------------------------------------------
def particleFilter(model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, dt, rng: ReplayRNG,
                   sparseFeatures=False, makePlots: Optional[Callable] = None, trace=False, fix_p_mean=False):
    """src/particleFilter.m:1-234 (dense branch, and the sparseFeatures EKF branch :127-137,165-181 for models
    whose measModel(xn_i, xl_i) returns (yhat, dy)).

    `model` supplies dynModel / measModel (the reference passes them as handles).
    Returns a dict with the 8 reference outputs (same shapes) and, if trace, per-step
    `ai` [T x N] (0-based, row 0 unused), `logw`, `w` [T x N] and final `xl`, `P`.
    """
    y = np.atleast_2d(np.asarray(y, dtype=np.float64))
    odometry = np.atleast_2d(np.asarray(odometry, dtype=np.float64))
    x0_nonLin = np.asarray(x0_nonLin, dtype=np.float64).ravel()
    x0_lin = np.asarray(x0_lin, dtype=np.float64)
    P0_lin = np.asarray(P0_lin, dtype=np.float64)
    R = np.atleast_2d(np.asarray(R, dtype=np.float64))

    w = 1.0 / N_P * np.ones(N_P)                                   # :55
    logw = np.log(w)                                               # :56
    xn = np.repeat(x0_nonLin[:, None], N_P, axis=1)                # :59
    if x0_lin.ndim == 2 and x0_lin.shape[1] > 1:                   # :60-64
        xl = x0_lin.copy()
    else:
        xl = np.repeat(x0_lin.reshape(-1, 1), N_P, axis=1)
    P = np.repeat(P0_lin[:, :, None], N_P, axis=2)                 # :67
    nNonLin = x0_nonLin.shape[0]
    N_T = y.shape[0]
    Q, dt = _expand_Q_dt(Q, dt, N_T)
    jitter = 1e-3                                                  # :89 (quirk Q2)

    traj_max = np.full((nNonLin, N_T), np.nan)
    traj_mean = np.full((nNonLin, N_T), np.nan)
    yhattraj = np.full((y.shape[1], N_T), np.nan)
    xn_traj = np.zeros((nNonLin, N_P, N_T))
    xn_traj[:, :, 0] = xn
    ai = np.zeros(N_P, dtype=np.int64)
    tr = {"ai": np.zeros((N_T, N_P), dtype=np.int64), "logw": np.zeros((N_T, N_P)),
          "w": np.zeros((N_T, N_P))} if trace else None
    iw_max = 0

    for t in range(N_T):                                           # :100
        xn_ = xn.copy()                                            # :102
        if t != 0:
            for i in range(N_P):                                   # :104-109
                ai[i] = sample(w, rng.U[0, t - 1, i])
                xn[:, i], _ = model.dynModel(xn_[:, ai[i]], odometry[t - 1, :], dt[t - 1],
                                             Q[:, :, t - 1], rng.Z[0, t - 1, i, :])
            xl = xl[:, ai]                                         # :112
            P = P[:, :, ai]                                        # :113
            xn_traj[:, :, t] = xn                                  # :117
            xn_traj[:, :, :t] = xn_traj[:, ai, :t]                 # :118
        yt = y[t, :]                                               # :122
        if not sparseFeatures:
            dy = model.measModel(xn)                               # :124
        for i in range(N_P):                                       # :126-151
            if sparseFeatures:
                logw[i] = _sparse_logw(model, yt, xn[:, i], xl[:, i], P[:, :, i], R, jitter)
            else:
                logw[i] = _importance_logw(yt, _dy_of(dy, i), xl[:, i], P[:, :, i], R, jitter)
        w = _normalise(logw)                                       # :154-156
        iw_max = int(np.argmax(w))                                 # :159 (first maximum)
        traj_max[:, t] = xn[:, iw_max]                             # :160
        traj_mean[:, t] = np.sum(xn * w, axis=1)                   # :161 (quirk Q8)
        if trace:
            tr["ai"][t] = ai
            tr["logw"][t] = logw
            tr["w"][t] = w
        xl = xl.copy()
        P = P.copy()
        for i in range(N_P):                                       # :164-204
            if sparseFeatures:
                xl[:, i], P[:, :, i], yhat = _sparse_update(model, yt, xn[:, i], xl[:, i], P[:, :, i], R, jitter)
            else:
                xl[:, i], P[:, :, i], yhat, _ = _kalman_update(yt, _dy_of(dy, i), xl[:, i], P[:, :, i], R, jitter)
            if i == iw_max:
                yhattraj[:, t] = yhat                              # :201-203
        if makePlots is not None:                                  # :215-217
            makePlots(xn, xl[:, iw_max], P[:, :, iw_max], traj_max, yhattraj, xn_traj, traj_mean, xl, P)

    xl_max = xl[:, iw_max].copy()                                  # :222
    P_max = P[:, :, iw_max].copy()                                 # :223
    xl_mean = np.sum(xl * w, axis=1)                               # :226
    P_mean = np.zeros((xl_mean.size, xl_mean.size))
    for i in range(N_P):                                           # :228-230 -- quirk Q3: '=' not '+='
        dlt = xl_mean - xl[:, i]
        term = w[i] * (P[:, :, i] + np.outer(dlt, dlt))
        P_mean = P_mean + term if fix_p_mean else term          # fix_p_mean: the evident intent ("+="), not the reference
    traj_sample_iwmax = xn_traj[:, iw_max, :].copy()               # :233
    out = dict(traj_max=traj_max, traj_mean=traj_mean, xl_max=xl_max, xl_mean=xl_mean,
               P_max=P_max, P_mean=P_mean, traj_sample_iwmax=traj_sample_iwmax, xn_traj=xn_traj,
               yhattraj=yhattraj, iw_max=iw_max)
    if trace:
        tr["xl"] = xl
        tr["P"] = P
        tr["xn"] = xn
        out["trace"] = tr
    return out


# --------------------------------------------------------------------------------------
# src/particleSmoother.m  (covariance-form ancestor weights)
# --------------------------------------------------------------------------------------
def _default_dyn_res_norm(xnkt, xni, odo, dt, Q):
    """particleSmoother.m:176-177 -- additive default."""
    Lq = np.linalg.cholesky(np.atleast_2d(dt * Q))
    return mrdivide_row(xnkt - xni - odo, Lq)


def particleSmoother(model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, N_K, dt, rng: ReplayRNG,
                     sparseFeatures=False, makePlots: Optional[Callable] = None,
                     use_dynResNorm=True, trace=False):
    """src/particleSmoother.m:1-367 (dense branch, and the sparseFeatures branch :194-217,267-277,306-321)."""
    y = np.atleast_2d(np.asarray(y, dtype=np.float64))
    odometry = np.atleast_2d(np.asarray(odometry, dtype=np.float64))
    x0_nonLin = np.asarray(x0_nonLin, dtype=np.float64).ravel()
    x0_lin = np.asarray(x0_lin, dtype=np.float64)
    P0_lin = np.asarray(P0_lin, dtype=np.float64)
    R = np.atleast_2d(np.asarray(R, dtype=np.float64))
    nNonLin = x0_nonLin.shape[0]
    nLin = x0_lin.shape[0]
    N_T, ny = y.shape
    Q, dt = _expand_Q_dt(Q, dt, N_T)
    jitter = 1e-2                                                  # :70 (quirk Q2)
    xn_traj = np.zeros((nNonLin, N_P, N_T))                        # :73 -- persists across k
    ai = np.zeros(N_P, dtype=np.int64)
    XNK = np.full((nNonLin, N_T, N_K), np.nan)
    XLK = np.full((nLin, N_K), np.nan)
    PK = np.full((nLin, nLin, N_K), np.nan)
    tr = {"ai": np.zeros((N_K, N_T, N_P), dtype=np.int64), "logw": np.zeros((N_K, N_T, N_P)),
          "w": np.zeros((N_K, N_T, N_P)), "paNt": np.full((N_K, N_T, N_P), np.nan),
          "ak": np.zeros(N_K, dtype=np.int64)} if trace else None
    dyn_res = model.dynResNorm if use_dynResNorm else None
    xnk = None

    for k in range(N_K):                                           # :88
        xn = np.repeat(x0_nonLin[:, None], N_P, axis=1)            # :91
        if k != 0:
            xn[:, N_P - 1] = xnk[:, 0]                             # :95
        if x0_lin.ndim == 2 and x0_lin.shape[1] > 1:               # :99-103
            xl = x0_lin.copy()
        else:
            xl = np.repeat(x0_lin.reshape(-1, 1), N_P, axis=1)
        P = np.repeat(P0_lin[:, :, None], N_P, axis=2)             # :104
        w = 1.0 / N_P * np.ones(N_P)                               # :107
        logw = np.log(w)
        if k != 0:
            xn_traj[:, N_P - 1, :] = xnk                           # :112
        xn_traj[:, :, 0] = xn                                      # :116
        if k != 0 and not sparseFeatures:
            dy_xnk = model.measModel(xnk)                          # :120  [T x ny x n] or [T x n]

        for t in range(N_T):                                       # :124
            if t != 0:
                xn_pred = np.zeros_like(xn)
                xl_pred = np.zeros_like(xl)
                P_pred = np.zeros_like(P)
                for i in range(N_P - 1):                           # :132-137
                    ai[i] = sample(w, rng.U[k, t - 1, i])
                    xn_pred[:, i], _ = model.dynModel(xn[:, ai[i]], odometry[t - 1, :], dt[t - 1],
                                                      Q[:, :, t - 1], rng.Z[k, t - 1, i, :])
                xl_pred[:, :-1] = xl[:, ai[:-1]]                   # :140
                P_pred[:, :, :-1] = P[:, :, ai[:-1]]               # :141
                if k == 0:                                         # :145-155
                    i = N_P - 1
                    ai[i] = sample(w, rng.U[k, t - 1, i])
                    xn_pred[:, i], _ = model.dynModel(xn[:, ai[i]], odometry[t - 1, :], dt[t - 1],
                                                      Q[:, :, t - 1], rng.Z[k, t - 1, i, :])
                    xl_pred[:, i] = xl[:, ai[i]]
                    P_pred[:, :, i] = P[:, :, ai[i]]
                else:
                    paNtLog = np.zeros(N_P)                        # :159
                    xnkt = xnk[:, t]                               # :170
                    if not sparseFeatures:
                        dyf = dy_xnk[t:N_T]                        # :163
                        if dyf.ndim == 3:                          # :164-166 -> [(ny*(T-t)) x n], time-major
                            dyf = dyf.reshape(ny * (N_T - t), nLin)
                        ytf = y[t:, :].reshape(ny * (N_T - t))     # :192
                        Rbig = np.kron(np.eye(N_T - t), R)         # :191
                    for i in range(N_P):                           # :171-233
                        if dyn_res is None:
                            eDyn = _default_dyn_res_norm(xnkt, xn[:, i], odometry[t - 1, :], dt[t - 1], Q[:, :, t - 1])
                        else:
                            eDyn = dyn_res(xnkt, xn[:, i], odometry[t - 1, :], dt[t - 1], Q[:, :, t - 1])
                        logwDyn = -0.5 * float(eDyn @ eDyn)         # :182
                        if not sparseFeatures:
                            SS = dyf @ P[:, :, i] @ dyf.T + Rbig   # :191
                            e = ytf - dyf @ xl[:, i]               # :193
                        else:                                      # :194-217: future observations, linearised at xl_i
                            es, dys, inds = [], [], []
                            for ti in range(t, N_T):
                                yti = y[ti, :]
                                ind = ~np.isnan(yti)
                                yhat_i, dyi = model.measModel(xnk[:, ti], xl[:, i])
                                es.append((yti - yhat_i)[ind])
                                dys.append(dyi[ind, :])
                                inds.append(np.nonzero(ind)[0])
                            e = np.concatenate(es)
                            dyf = np.vstack(dys)
                            RS = np.zeros((e.size, e.size))
                            o = 0
                            for idx in inds:                       # blkdiag(RS, R(ind,ind))  (:211)
                                RS[o:o + idx.size, o:o + idx.size] = R[np.ix_(idx, idx)]
                                o += idx.size
                            SS = dyf @ P[:, :, i] @ dyf.T + RS     # :215
                        if e.size:
                            cS = _chol_lower_with_jitter(SS, jitter)   # :221-224
                            v = np.linalg.solve(cS, e)
                            logwMeas = -np.sum(np.log(np.diag(cS))) - 0.5 * (v @ v) - e.size / 2.0 * LOG2PI  # :229
                        else:
                            logwMeas = 0.0
                        paNtLog[i] = math.log(w[i]) + logwDyn + logwMeas if w[i] > 0 else -np.inf   # :232
                    paNt = _normalise(paNtLog)                     # :236-238
                    if trace:
                        tr["paNt"][k, t] = paNt
                    i = N_P - 1
                    ai[i] = sample(paNt, rng.U[k, t - 1, i])       # :241 (quirk Q9)
                    xn_pred[:, i] = xnkt                           # :242
                    xl_pred[:, i] = xl[:, ai[i]]                   # :243
                    P_pred[:, :, i] = P[:, :, ai[i]]               # :244
                xn, xl, P = xn_pred, xl_pred, P_pred               # :251-253
                xn_traj[:, :, t] = xn                              # :256
                xn_traj[:, :, :t] = xn_traj[:, ai, :t]             # :257

            yt = y[t, :]
            if not sparseFeatures:
                dy = model.measModel(xn)                           # :264
            for i in range(N_P):                                   # :266-294
                if sparseFeatures:
                    logw[i] = _sparse_logw(model, yt, xn[:, i], xl[:, i], P[:, :, i], R, jitter)
                else:
                    logw[i] = _importance_logw(yt, _dy_of(dy, i), xl[:, i], P[:, :, i], R, jitter)
            w = _normalise(logw)                                   # :300-302
            if trace:
                tr["ai"][k, t] = ai
                tr["logw"][k, t] = logw
                tr["w"][k, t] = w
            for i in range(N_P):                                   # :305-340
                if sparseFeatures:
                    xl[:, i], P[:, :, i], _ = _sparse_update(model, yt, xn[:, i], xl[:, i], P[:, :, i], R, jitter)
                else:
                    xl[:, i], P[:, :, i], _, _ = _kalman_update(yt, _dy_of(dy, i), xl[:, i], P[:, :, i], R, jitter)

        ak = sample(w, rng.Ufin[k])                                # :346
        xnk = xn_traj[:, ak, :].copy()                             # :347
        XNK[:, :, k] = xnk                                         # :352
        XLK[:, k] = xl[:, ak]
        PK[:, :, k] = P[:, :, ak]
        if trace:
            tr["ak"][k] = ak
        if makePlots is not None:
            makePlots(xnk, xl[:, ak], k, XNK, XLK, PK)             # :360-362
    out = dict(XNK=XNK, XLK=XLK, PK=PK)
    if trace:
        out["trace"] = tr
    return out


# --------------------------------------------------------------------------------------
# src/particleSmootherInformationForm.m
# --------------------------------------------------------------------------------------
def particleSmootherInformationForm(model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, N_K, dt,
                                    rng: ReplayRNG, sparseFeatures=False,
                                    makePlots: Optional[Callable] = None, use_dynResNorm=True, trace=False):
    """src/particleSmootherInformationForm.m:1-362."""
    if sparseFeatures:                                             # :77-80 (quirk Q5: returns unassigned)
        raise NotImplementedError("This code has only been implemented for dense features")
    y = np.atleast_2d(np.asarray(y, dtype=np.float64))
    odometry = np.atleast_2d(np.asarray(odometry, dtype=np.float64))
    x0_nonLin = np.asarray(x0_nonLin, dtype=np.float64).ravel()
    x0_lin = np.asarray(x0_lin, dtype=np.float64)
    P0_lin = np.asarray(P0_lin, dtype=np.float64)
    R = np.atleast_2d(np.asarray(R, dtype=np.float64))
    nNonLin = x0_nonLin.shape[0]
    nLin = x0_lin.shape[0]
    N_T, ny = y.shape
    Q, dt = _expand_Q_dt(Q, dt, N_T)
    jitter = 1e-2                                                  # :75
    xn_traj = np.zeros((nNonLin, N_P, N_T))
    ai = np.zeros(N_P, dtype=np.int64)
    XNK = np.full((nNonLin, N_T, N_K), np.nan)
    XLK = np.full((nLin, N_K), np.nan)
    PK = np.full((nLin, nLin, N_K), np.nan)
    tr = {"ai": np.zeros((N_K, N_T, N_P), dtype=np.int64), "logw": np.zeros((N_K, N_T, N_P)),
          "w": np.zeros((N_K, N_T, N_P)), "paNt": np.full((N_K, N_T, N_P), np.nan),
          "ak": np.zeros(N_K, dtype=np.int64)} if trace else None
    dyn_res = model.dynResNorm if use_dynResNorm else None
    Rinv = np.linalg.inv(R)
    halfLogDetR = 0.5 * math.log(np.linalg.det(R))
    x0l = x0_lin.reshape(nLin, -1)[:, 0]                           # quirk Q5: repmat(x0_lin,1,N_P)
    xnk = None

    for k in range(N_K):                                           # :98
        xn = np.repeat(x0_nonLin[:, None], N_P, axis=1)
        if k != 0:
            xn[:, N_P - 1] = xnk[:, 0]                             # :105
        xl = np.repeat(x0l[:, None], N_P, axis=1)                  # :109
        ivec0 = np.diag(1.0 / np.diag(P0_lin)) @ x0l               # :110
        ivec = np.repeat(ivec0[:, None], N_P, axis=1)              # :111
        P = np.repeat(P0_lin[:, :, None], N_P, axis=2)             # :112
        Imat = np.repeat(np.diag(1.0 / np.diag(P0_lin))[:, :, None], N_P, axis=2)   # :113
        halfLogDetP = np.sum(np.log(np.sqrt(np.diag(P0_lin)))) * np.ones(N_P)      # :115
        w = 1.0 / N_P * np.ones(N_P)
        logw = np.log(w)
        if k != 0:
            xn_traj[:, N_P - 1, :] = xnk                           # :123
        xn_traj[:, :, 0] = xn                                      # :127
        if k != 0:                                                 # :132-146
            dy_xnk = model.measModel(xnk)
            ImatAddt = np.zeros((nLin, nLin))
            ivecAddt = np.zeros(nLin)
            for jj in range(N_T):
                Hj = _dy_of(dy_xnk, jj)
                ivecAddt = ivecAddt + Hj.T @ Rinv @ y[jj, :]
                ImatAddt = ImatAddt + Hj.T @ Rinv @ Hj

        for t in range(N_T):                                       # :149
            if t != 0:
                xn_pred = np.zeros_like(xn)
                xl_pred = np.zeros_like(xl)
                P_pred = np.zeros_like(P)
                ivec_pred = np.zeros_like(ivec)
                Imat_pred = np.zeros_like(Imat)
                for i in range(N_P - 1):                           # :159-164
                    ai[i] = sample(w, rng.U[k, t - 1, i])
                    xn_pred[:, i], _ = model.dynModel(xn[:, ai[i]], odometry[t - 1, :], dt[t - 1],
                                                      Q[:, :, t - 1], rng.Z[k, t - 1, i, :])
                xl_pred[:, :-1] = xl[:, ai[:-1]]                   # :167-170
                P_pred[:, :, :-1] = P[:, :, ai[:-1]]
                ivec_pred[:, :-1] = ivec[:, ai[:-1]]
                Imat_pred[:, :, :-1] = Imat[:, :, ai[:-1]]
                i = N_P - 1
                if k == 0:                                         # :174-186
                    ai[i] = sample(w, rng.U[k, t - 1, i])
                    xn_pred[:, i], _ = model.dynModel(xn[:, ai[i]], odometry[t - 1, :], dt[t - 1],
                                                      Q[:, :, t - 1], rng.Z[k, t - 1, i, :])
                else:
                    paNtLog = np.zeros(N_P)
                    Hm = _dy_of(dy_xnk, t - 1)                     # :194-201
                    ivecAddt = ivecAddt - Hm.T @ Rinv @ y[t - 1, :]
                    ImatAddt = ImatAddt - Hm.T @ Rinv @ Hm
                    xnkt = xnk[:, t]
                    for j in range(N_P):                           # :205-240
                        if dyn_res is None:
                            eDyn = _default_dyn_res_norm(xnkt, xn[:, j], odometry[t - 1, :], dt[t - 1], Q[:, :, t - 1])
                        else:
                            eDyn = dyn_res(xnkt, xn[:, j], odometry[t - 1, :], dt[t - 1], Q[:, :, t - 1])
                        logwDyn = -0.5 * float(eDyn @ eDyn)
                        ivecEnd = ivec[:, j] + ivecAddt            # :224
                        ImatEnd = Imat[:, :, j] + ImatAddt         # :225
                        try:
                            cIend = np.linalg.cholesky(ImatEnd)    # :228
                        except np.linalg.LinAlgError as exc:
                            # quirk Q4: the reference re-factorises the failed partial factor
                            # (:229-231), which cannot produce a usable result; treat as an error.
                            raise CholeskyFailure("information matrix not positive definite") from exc
                        vIend = np.linalg.solve(cIend, ivecEnd)    # :233
                        logwMeas = (-0.5 * (ivec[:, j] @ P[:, :, j] @ ivec[:, j]) - halfLogDetP[j]
                                    - np.sum(np.log(np.diag(cIend))) + 0.5 * (vIend @ vIend))   # :234-236
                        paNtLog[j] = (math.log(w[j]) if w[j] > 0 else -np.inf) + logwDyn + logwMeas
                    paNt = _normalise(paNtLog)                     # :243-245
                    if trace:
                        tr["paNt"][k, t] = paNt
                    ai[i] = sample(paNt, rng.U[k, t - 1, i])       # :248
                    xn_pred[:, i] = xnkt
                xl_pred[:, i] = xl[:, ai[i]]
                P_pred[:, :, i] = P[:, :, ai[i]]
                ivec_pred[:, i] = ivec[:, ai[i]]
                Imat_pred[:, :, i] = Imat[:, :, ai[i]]
                xn, xl, P, ivec, Imat = xn_pred, xl_pred, P_pred, ivec_pred, Imat_pred   # :260-264
                halfLogDetP = halfLogDetP[ai]                      # :267
                xn_traj[:, :, t] = xn
                xn_traj[:, :, :t] = xn_traj[:, ai, :t]             # :270-271

            yt = y[t, :]
            dy = model.measModel(xn)                               # :276
            halfLogDetPplus = np.zeros_like(halfLogDetP)
            for i in range(N_P):                                   # :279-305
                dyi = _dy_of(dy, i)
                SS = dyi @ P[:, :, i] @ dyi.T + R
                cS = _chol_lower_with_jitter(SS, jitter)
                ivecPlus = ivec[:, i] + dyi.T @ Rinv @ yt          # :292
                M = np.linalg.solve(cS, dyi).T
                M = np.linalg.solve(cS.T, M.T).T
                K = P[:, :, i] @ M                                 # :293
                Pplus = P[:, :, i] - K @ SS @ K.T                  # :294
                halfLogDetPplus[i] = -np.sum(np.log(np.diag(cS))) + halfLogDetR + halfLogDetP[i]   # :298
                logw[i] = (-0.5 * (ivec[:, i] @ P[:, :, i] @ ivec[:, i]) - halfLogDetP[i] + halfLogDetPplus[i]
                           + 0.5 * (ivecPlus @ Pplus @ ivecPlus) - 0.5 * (yt @ Rinv @ yt)
                           - 0.5 * math.log((2 * math.pi) ** yt.size * np.linalg.det(R)))          # :301-304
            halfLogDetP = halfLogDetPplus                          # :308
            w = _normalise(logw)                                   # :311-313
            if trace:
                tr["ai"][k, t] = ai
                tr["logw"][k, t] = logw
                tr["w"][k, t] = w
            for i in range(N_P):                                   # :316-335
                dyi = _dy_of(dy, i)
                xl[:, i], P[:, :, i], _, _ = _kalman_update(yt, dyi, xl[:, i], P[:, :, i], R, jitter)
                ivec[:, i] = ivec[:, i] + dyi.T @ Rinv @ yt        # :333
                Imat[:, :, i] = Imat[:, :, i] + dyi.T @ Rinv @ dyi  # :334

        ak = sample(w, rng.Ufin[k])                                # :341
        xnk = xn_traj[:, ak, :].copy()
        XNK[:, :, k] = xnk
        XLK[:, k] = xl[:, ak]
        PK[:, :, k] = P[:, :, ak]
        if trace:
            tr["ak"][k] = ak
        if makePlots is not None:
            makePlots(xnk, xl[:, ak], k, XNK, XLK, PK)
    out = dict(XNK=XNK, XLK=XLK, PK=PK)
    if trace:
        out["trace"] = tr
    return out


# --------------------------------------------------------------------------------------
# Synthetic data: examples/slam-dense-radio/generateData_dense.m
# --------------------------------------------------------------------------------------
def gp_rnd_scalar_potential_fast(x, m, LL, theta, zf, zy):
    """tools/gp_rnd_scalar_potential_fast.m:42-102.  zf: randn(m+3), zy: randn(size(df))."""
    LL = np.asarray(LL, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64) - LL.mean(axis=0)
    Lh = (LL.max(axis=0) - LL.min(axis=0)) / 2.0
    L, NN = domain_cartesian_dx(m, 3, Lh[None, :])
    lam = eigenval(NN, L)
    npts = x.shape[0]
    one, zero = np.ones((npts, 1)), np.zeros((npts, 1))
    Phi = np.hstack((x, eigenfun(NN, x, L)))
    dPhix = np.hstack((one, zero, zero, eigenfun_dx(NN, x, 0, L)))
    dPhiy = np.hstack((zero, one, zero, eigenfun_dx(NN, x, 1, L)))
    dPhiz = np.hstack((zero, zero, one, eigenfun_dx(NN, x, 2, L)))
    linSigma2, lengthScale, magnSigma2, sigma2 = (float(t) for t in theta)
    d = 3
    w = np.sqrt(lam)
    Sse = magnSigma2 * math.sqrt(2 * math.pi) ** d * lengthScale ** d * np.exp(-w ** 2 * lengthScale ** 2 / 2)
    k = np.concatenate(([linSigma2] * 3, Sse))
    foo = np.sqrt(k) * zf
    f = Phi @ foo
    df = np.column_stack((dPhix @ foo, dPhiy @ foo, dPhiz @ foo))
    yv = df + math.sqrt(sigma2) * zy
    return f, df, yv


def gp_rnd_SE1D_fast(x, m, LL, theta, zf, zy):
    """tools/gp_rnd_SE1D_fast.m:44-85."""
    LL = np.asarray(LL, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64) - LL.mean(axis=0)
    Lh = (LL.max(axis=0) - LL.min(axis=0)) / 2.0
    d = x.shape[1]
    L, NN = domain_cartesian_dx(m, d, Lh[None, :])
    lam = eigenval(NN, L)
    Phi = eigenfun(NN, x, L)
    lengthScale, magnSigma2, sigma2 = (float(t) for t in theta)
    w = np.sqrt(lam)
    k = magnSigma2 * math.sqrt(2 * math.pi) ** d * lengthScale ** d * np.exp(-w ** 2 * lengthScale ** 2 / 2)
    foo = np.sqrt(k) * zf
    f = Phi @ foo
    return f, f + math.sqrt(sigma2) * zy


def generate_bean_6D(N_T, Q, theta, dt, seed, m_sim=2000, nLL=2, laps=3, a=15.0):
    """generateData_dense.m:181-213 (bean_6D), :216-257 (field), :294-325 (odometry), with
    nLaps*nDataPointsPerLap = N_T so the box (hence LL, NN) matches the reference's.
    Seeded numpy normals replace MATLAB randn.  Returns dict(dx, initState, y, LL, pos, quat)."""
    rs = np.random.RandomState(seed)
    psi = np.linspace(0.0, laps * np.pi, N_T)
    r = a * np.sin(psi) ** 3 + a * np.cos(psi) ** 3
    u = r * np.cos(psi) - 0.3
    v = r * np.sin(psi) - 0.3
    th = np.arctan2(np.diff(v), np.diff(u))
    th = np.concatenate((th, th[-1:]))
    pos = np.vstack((u, v, np.zeros_like(u)))
    quat = np.stack([rmat2quat_planar(t) for t in th], axis=0)          # [N x 4]
    pos = pos - np.mean(np.column_stack((pos.min(axis=1), pos.max(axis=1))), axis=1)[:, None]
    initState = np.concatenate((pos[:, 0], quat[0]))
    dPos = np.diff(pos.T, axis=0)
    dQuat = np.stack([qLeft(qInv(quat[i])) @ quat[i + 1] for i in range(N_T - 1)], axis=0)
    dx = np.hstack((dPos, dQuat))
    lengthScale = float(theta[1])
    LL = np.array([[pos[0].min() - nLL * lengthScale, pos[1].min() - nLL * lengthScale, -nLL * lengthScale],
                   [pos[0].max() + nLL * lengthScale, pos[1].max() + nLL * lengthScale, nLL * lengthScale]])
    zf = rs.standard_normal(m_sim + 3)
    zy = rs.standard_normal((N_T, 3))
    _, _, yn = gp_rnd_scalar_potential_fast(pos.T, m_sim, LL, theta, zf, zy)
    y = np.stack([quat2rmat(quat[i]).T @ yn[i] for i in range(N_T)], axis=0)   # :253-257
    # odometry noise: run dynModel forward (:302-309)
    Qe, dte = _expand_Q_dt(Q, dt, N_T)
    mdl = DenseMagModel(NN=np.zeros((1, 3)), L=np.ones(3))
    x = np.zeros((N_T, 7))
    dQn = np.zeros((N_T - 1, 4))
    x[0] = initState
    zo = rs.standard_normal((N_T - 1, 6))
    for i in range(1, N_T):
        x[i], dQn[i - 1] = mdl.dynModel(x[i - 1], dx[i - 1], dte[i - 1], Qe[:, :, i - 1], zo[i - 1])
    dxn = np.hstack((np.diff(x[:, 0:3], axis=0), dQn))
    return dict(dx=dxn, initState=initState, y=y, LL=LL, pos=pos, quat=quat)


def generate_line_3D(N_T, Q, theta, dt, seed, m_sim=2000, nLL=2, traj="line_3D"):
    """generateData_dense.m:101-132 (square_3D / line_3D), :258-290, :310-323."""
    rs = np.random.RandomState(seed)
    N = N_T
    if traj == "line_3D":
        pos = np.vstack((np.zeros(N), np.concatenate((np.linspace(0, 3, N // 2), np.linspace(3, 0, N - N // 2)))))
    else:
        q = N // 4
        pos = np.vstack((np.concatenate((np.zeros(q), np.linspace(0, 2, q), 2 * np.ones(q), np.linspace(2, 0, N - 3 * q))),
                         np.concatenate((np.linspace(0, 2, q), 2 * np.ones(q), np.linspace(2, 0, q), np.zeros(N - 3 * q)))))
    pos = pos - pos.mean(axis=1, keepdims=True)
    initState = np.concatenate((pos[:, 0], [0.0]))
    dx = np.hstack((np.diff(pos.T, axis=0), np.zeros((N - 1, 1))))
    lengthScale = float(theta[0])
    LL = np.array([[pos[0].min() - nLL * lengthScale, pos[1].min() - nLL * lengthScale],
                   [pos[0].max() + nLL * lengthScale, pos[1].max() + nLL * lengthScale]])
    zf = rs.standard_normal(m_sim)
    zy = rs.standard_normal(N)
    _, y = gp_rnd_SE1D_fast(pos.T, m_sim, LL, theta, zf, zy)
    Qe, dte = _expand_Q_dt(Q, dt, N_T)
    mdl = DenseRadioModel(NN=np.zeros((1, 2)), L=np.ones(2))
    x = np.zeros((N, 3))
    x[0] = initState
    zo = rs.standard_normal((N - 1, 1))
    for i in range(1, N):
        x[i], _ = mdl.dynModel(x[i - 1], dx[i - 1], dte[i - 1], Qe[:, :, i - 1], zo[i - 1])
    dxn = np.hstack((dx[:, 0:2], np.diff(x[:, 2])[:, None]))               # :319
    return dict(dx=dxn, initState=initState, y=y.reshape(-1, 1), LL=LL, pos=pos)


# --------------------------------------------------------------------------------------
# examples/slam-dense-mag/ekf_dense.m -- the EKF comparison baseline
# --------------------------------------------------------------------------------------
def measModel_ekf(model, LL, x, q):
    """run_dense3D_magfield.m:281-299: [yhat, dy] with the state x = [pos(3); orientation deviation(3); map(m+3)]."""
    x = np.asarray(x, dtype=np.float64).ravel()
    LL = np.asarray(LL, dtype=np.float64)
    pos = x[0:3][None, :]
    m = model.NN.shape[0]
    dPhi = np.vstack((np.concatenate(([1.0, 0.0, 0.0], eigenfun_dx(model.NN, pos, 0, model.L)[0])),      # :282-287
                      np.concatenate(([0.0, 1.0, 0.0], eigenfun_dx(model.NN, pos, 1, model.L)[0])),
                      np.concatenate(([0.0, 0.0, 1.0], eigenfun_dx(model.NN, pos, 2, model.L)[0]))))
    Rnb = quat2rmat(q)                                                                                  # :288
    yhat = Rnb.T @ dPhi @ x[6:]                                                                         # :290
    dy = np.zeros((3, dPhi.shape[1] + 6))
    J = JacobianPhi3D(x[0:3], m, LL[0, 0], LL[1, 0], LL[0, 1], LL[1, 1], LL[0, 2], LL[1, 2], model.NN)  # :292-294
    J3 = np.tensordot(J[:, :, :, 0], x[9:], axes=([2], [0]))                                            # reshape(9,m)*x(10:end)
    dy[:, 0:3] = Rnb.T @ J3                                                                             # :296
    dy[:, 3:6] = Rnb.T @ mcross(dPhi @ x[6:])                                                           # :297
    dy[:, 6:] = Rnb.T @ dPhi                                                                            # :298
    return yhat, dy


def dynModel_ekf(x, q, dx):
    """run_dense3D_magfield.m:310-316."""
    x = np.asarray(x, dtype=np.float64).ravel()
    dx = np.asarray(dx, dtype=np.float64).ravel()
    xpred = x.copy()
    xpred[0:3] = x[0:3] + dx[0:3]
    qpred = qLeft(q) @ dx[3:7]
    F = np.eye(x.size)
    G = np.zeros((x.size, 6))
    G[0:3, 0:3] = np.eye(3)
    G[3:6, 3:6] = quat2rmat(qpred)
    return xpred, qpred, F, G


def ekf_dense(model, LL, odometry, y, x0, q0, P0, Q, R, dt):
    """examples/slam-dense-mag/ekf_dense.m:41-102 -> (xf_traj [nStates x T], qnb_traj [4 x T], Pf_traj [n x n x T])."""
    y = np.atleast_2d(np.asarray(y, dtype=np.float64))
    odometry = np.atleast_2d(np.asarray(odometry, dtype=np.float64))
    xp = np.asarray(x0, dtype=np.float64).ravel().copy()
    Pp = np.asarray(P0, dtype=np.float64).copy()
    q_nb = np.asarray(q0, dtype=np.float64).ravel().copy()
    nStates, N_T = xp.size, y.shape[0]
    Q, dt = _expand_Q_dt(Q, dt, N_T)
    jitter = 1e-3                                                   # :58
    xf_traj = np.full((nStates, N_T), np.nan)
    Pf_traj = np.full((nStates, nStates, N_T), np.nan)
    qnb_traj = np.full((4, N_T), np.nan)
    xf, Pf = xp, Pp
    for t in range(N_T):                                            # :67
        if t != 0:
            xp, q_nb, F, G = dynModel_ekf(xf, q_nb, odometry[t - 1, :])          # :71
            Qt = dt[t - 1] * Q[:, :, t - 1]
            Pp = F @ Pf @ F.T + G @ Qt @ G.T                        # :73
        yhat, dy = measModel_ekf(model, LL, xp, q_nb)               # :78
        e = y[t, :] - yhat
        SS = dy @ Pp @ dy.T + R
        cS = _chol_lower_with_jitter(SS, jitter)                    # :83-86
        Mx = np.linalg.solve(cS, dy).T
        Mx = np.linalg.solve(cS.T, Mx.T).T
        K = Pp @ Mx                                                 # :87
        xf = xp + K @ e                                             # :90
        Pf = Pp - K @ SS @ K.T
        Pf = 0.5 * (Pf + Pf.T)                                      # :92
        q_nb = qLeft(expq(xf[3:6] / 2.0)) @ q_nb                    # :95
        xf[3:6] = 0.0
        xf_traj[:, t], Pf_traj[:, :, t], qnb_traj[:, t] = xf, Pf, q_nb
    return xf_traj, qnb_traj, Pf_traj


# --------------------------------------------------------------------------------------
# acceptance metrics of the example scripts
# --------------------------------------------------------------------------------------
def procrustes_oracle(X, Y):
    """MATLAB's procrustes(X, Y) with its default options (scaling and reflection allowed), the function
    examples/slam-sparse-visual/calc_rmses.m:38 and run_dense3D_magfield.m:160-161 call.  It is a Statistics-Toolbox
    function, absent from the reference tree; this restates its published definition by a route of its own -- the orthogonal
    factor of the polar decomposition via the symmetric eigenproblem of A'A instead of an SVD: with X0, Y0 centred and scaled
    to unit Frobenius norm and A = X0'Y0, T = (A'A)^(-1/2) A' maximises trace(T A), b = trace((A'A)^(1/2)) |X| / |Y|,
    d = 1 - trace((A'A)^(1/2))^2, Z = |X| trace(.) Y0 T + mean(X), c = mean(X) - b mean(Y) T.  Needs X0'Y0 of full rank
    (a planar truth against a 3-D estimate leaves the rotation about the plane's normal undetermined in any formulation)."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    muX, muY = X.mean(axis=0), Y.mean(axis=0)
    X0, Y0 = X - muX, Y - muY
    nX, nY = math.sqrt(float(np.sum(X0 * X0))), math.sqrt(float(np.sum(Y0 * Y0)))
    X0, Y0 = X0 / nX, Y0 / nY
    A = X0.T @ Y0
    lam, V = np.linalg.eigh(A.T @ A)
    root = np.sqrt(np.maximum(lam, 0.0))
    T = V @ np.diag(1.0 / root) @ V.T @ A.T                   # (A'A)^(-1/2) A'
    tr = float(np.sum(root))
    b = tr * nX / nY
    d = 1.0 - tr * tr
    Z = nX * tr * (Y0 @ T) + muX
    c = muX - b * (muY @ T)
    return d, Z, dict(b=b, T=T, c=c)


def calc_rmses_oracle(truth, estimate, map_true, map_est, traj, traj_est):
    """examples/slam-sparse-visual/calc_rmses.m:35-55 -> (rmse_path, rmse_map); the angle RMSE is NaN by construction (:44)."""
    _, _, tr = procrustes_oracle(truth, estimate)                                       # :38
    Z = tr["b"] * np.asarray(traj_est)[:, 0:2] @ tr["T"] + tr["c"]                       # :41
    Zm = tr["b"] * np.asarray(map_est) @ tr["T"] + tr["c"]                               # :47
    d = np.sqrt(np.sum((np.asarray(traj)[:, 0:2] - Z) ** 2, axis=1))                     # :50
    dm = np.sqrt(np.sum((np.asarray(map_true) - Zm) ** 2, axis=1))                       # :57
    return math.sqrt(float(np.mean(d ** 2))), math.sqrt(float(np.mean(dm ** 2)))


def quat2euler_oracle(q):
    """tools/quat2euler.m:26-34, one quaternion, degrees."""
    q0, q1, q2, q3 = (float(v) for v in np.asarray(q, dtype=np.float64).ravel())
    return 180.0 / math.pi * np.array([math.atan2(2 * q2 * q3 - 2 * q0 * q1, 2 * q0 ** 2 + 2 * q3 ** 2 - 1),
                                       -math.asin(max(-1.0, min(1.0, 2 * q1 * q3 + 2 * q0 * q2))),
                                       math.atan2(2 * q1 * q2 - 2 * q0 * q3, 2 * q0 ** 2 + 2 * q1 ** 2 - 1)])


def rmse_dense_mag_oracle(pos_true, quat_true, traj):
    """run_dense3D_magfield.m:160-181 for one trajectory estimate [7 x T]."""
    pos_true = np.asarray(pos_true, dtype=np.float64)
    _, Z, _ = procrustes_oracle(pos_true.T, np.asarray(traj)[0:3, :].T)
    rmse_pos = np.sqrt(np.mean((pos_true.T - Z) ** 2, axis=0))
    err = np.stack([quat2euler_oracle(qLeft(np.asarray(traj)[3:7, ii]) @ qInv(np.asarray(quat_true)[ii])) for ii in range(np.asarray(traj).shape[1])])
    return rmse_pos, np.sqrt(np.mean(err ** 2, axis=0))
