"""Seeded problem instances shared by the golden-vector generator and the parity tests.

Everything is built from the oracle's synthetic-data generators (oracle/rbpf_oracle.py, which restate
examples/slam-dense-radio/generateData_dense.m) with fixed numpy seeds; sizes are small enough for
the numpy oracle to finish in seconds.
"""
import numpy as np

import rbpf_oracle as O

Q_MAG = np.diag(np.concatenate((10 ** 2 * np.array([0.05 ** 2, 0.05 ** 2, 0.01 ** 2]),
                                (np.array([0.01, 0.01, 0.3]) * np.pi / 180) ** 2)))   # slam-dense-mag/main.m:22
THETA_MAG = np.array([650.0, 1.2, 200.0, 10.0])                                       # slam-dense-mag/main.m:23
THETA_RADIO = np.array([0.25, 2.0, 0.01])                                             # slam-dense-radio/main.m:24


def mag_case(N_P, N_T, m, seed=1, N_K=1, dt=0.01, m_sim=300):
    d = O.generate_bean_6D(N_T, Q_MAG, THETA_MAG, dt, seed=seed, m_sim=m_sim)
    model, x0_lin, P0, R = O.dense_mag_prior(m, d["LL"], THETA_MAG)
    rng = O.ReplayRNG.draw(seed + 100, N_K, N_T, N_P, 6)
    return dict(kind="mag", model=model, odometry=d["dx"], y=d["y"], x0_nonLin=d["initState"], x0_lin=x0_lin,
                P0_lin=P0, Q=Q_MAG, R=R, N_P=N_P, dt=dt, rng=rng, N_K=N_K, LL=d["LL"], theta=THETA_MAG, m=m)


def radio_case(N_P, N_T, m, seed=1, N_K=1, traj="line_3D"):
    Qs = 1e-6 * np.ones(N_T)
    Qs[N_T // 2 - 1] = 0.3 ** 2                                  # run_dense2D_withHeading.m:71-72
    Q = Qs.reshape(1, 1, N_T)
    d = O.generate_line_3D(N_T, Q, THETA_RADIO, 1.0, seed=seed, m_sim=300, nLL=4, traj=traj)
    model, x0_lin, P0, R = O.dense_radio_prior(m, d["LL"], THETA_RADIO)
    rng = O.ReplayRNG.draw(seed + 200, N_K, N_T, N_P, 1)
    return dict(kind="radio", model=model, odometry=d["dx"], y=d["y"], x0_nonLin=d["initState"], x0_lin=x0_lin,
                P0_lin=P0, Q=Q, R=R, N_P=N_P, dt=1.0, rng=rng, N_K=N_K, LL=d["LL"], theta=THETA_RADIO, m=m)


def oracle_filter(c, trace=True):
    return O.particleFilter(c["model"], c["odometry"], c["y"], c["x0_nonLin"], c["x0_lin"], c["P0_lin"], c["Q"],
                            c["R"], c["N_P"], c["dt"], c["rng"], trace=trace)


def oracle_smoother(c, info_form, trace=True, use_dynResNorm=True):
    f = O.particleSmootherInformationForm if info_form else O.particleSmoother
    return f(c["model"], c["odometry"], c["y"], c["x0_nonLin"], c["x0_lin"], c["P0_lin"], c["Q"], c["R"], c["N_P"],
             c["N_K"], c["dt"], c["rng"], trace=trace, use_dynResNorm=use_dynResNorm)


def device_model(rbpf, c):
    """The product-side model-family object for an oracle case (same NN / L)."""
    if c["kind"] == "mag":
        mdl, x0, P0, R = rbpf.dense_mag_prior(c["m"], c["LL"], c["theta"])
    else:
        mdl, x0, P0, R = rbpf.dense_radio_prior(c["m"], c["LL"], c["theta"])
    return mdl, x0, P0, R


def device_rng(rbpf, c):
    r = c["rng"]
    return rbpf.ReplayRNG(r.U, r.Z, r.Ufin)
