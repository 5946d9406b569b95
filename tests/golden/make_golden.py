#!/usr/bin/env python3
"""Generates the committed golden vectors in tests/golden/*.npz from the numpy oracle
(oracle/rbpf_oracle.py) on the seeded cases of tests/cases.py.

The reference ships no golden vectors and cannot be run here (MATLAB), so these pin the build -- and
regressions of the oracle itself -- to this repository's reading of the cited .m files.  Each file
holds the full inputs needed to re-run the case through the C ABI plus the expected outputs
(traces of ancestor indices / weights, trajectories, maps; covariances as diagonal + one matrix).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import cases  # noqa: E402


def inputs(c):
    return dict(kind=c["kind"], m=c["m"], LL=c["LL"], theta=c["theta"], odometry=c["odometry"], y=c["y"],
                x0_nonLin=c["x0_nonLin"], Q=c["Q"], N_P=c["N_P"], dt=c["dt"], N_K=c["N_K"], U=c["rng"].U,
                Z=c["rng"].Z, Ufin=c["rng"].Ufin, NN=c["model"].NN, L=c["model"].L)


def save_filter(name, c):
    r = cases.oracle_filter(c)
    tr = r["trace"]
    np.savez_compressed(os.path.join(HERE, name), **inputs(c), ai=tr["ai"], logw=tr["logw"], w=tr["w"],
                        traj_max=r["traj_max"], traj_mean=r["traj_mean"], xl_max=r["xl_max"], xl_mean=r["xl_mean"],
                        P_max=r["P_max"], P_mean_diag=np.diag(r["P_mean"]).copy(),
                        traj_sample_iwmax=r["traj_sample_iwmax"], iw_max=r["iw_max"],
                        final_xl=tr["xl"], final_P_diag=np.stack([np.diag(tr["P"][:, :, i]) for i in range(c["N_P"])]))


def save_smoother(name, c, info_form):
    r = cases.oracle_smoother(c, info_form)
    tr = r["trace"]
    np.savez_compressed(os.path.join(HERE, name), **inputs(c), info_form=info_form, ai=tr["ai"], w=tr["w"],
                        paNt=tr["paNt"], ak=tr["ak"], XNK=r["XNK"], XLK=r["XLK"], PK=r["PK"])


def save_sparse(name, c):
    """slam-sparse-visual on the reference's data file curve-x2.mat (sparseFeatures branch): filter and smoother."""
    r = cases.oracle_filter(c)
    sm = cases.oracle_smoother(c, False)
    np.savez_compressed(os.path.join(HERE, name), N_P=c["N_P"], N_T=c["y"].shape[0], N_K=c["N_K"], y=c["y"],
                        odometry=c["odometry"], filter_ai=r["trace"]["ai"], filter_w=r["trace"]["w"],
                        filter_traj_mean=r["traj_mean"], filter_xl_mean=r["xl_mean"], filter_P_max=r["P_max"],
                        smoother_ak=sm["trace"]["ak"], smoother_ai=sm["trace"]["ai"], smoother_XNK=sm["XNK"],
                        smoother_XLK=sm["XLK"])


if __name__ == "__main__":
    save_sparse("sparse_curve_n40.npz", cases.sparse_curve_case(None, 10, 30, N_K=3))
    save_filter("filter_mag_n19.npz", cases.mag_case(8, 8, 16, seed=31))
    save_filter("filter_mag_n133.npz", cases.mag_case(6, 6, 130, seed=32))
    save_filter("filter_radio_n32.npz", cases.radio_case(12, 10, 32, seed=33))
    save_smoother("smoother_cov_mag_n19.npz", cases.mag_case(8, 6, 16, seed=34, N_K=3), False)
    save_smoother("smoother_info_mag_n19.npz", cases.mag_case(8, 6, 16, seed=34, N_K=3), True)
    save_smoother("smoother_cov_radio_n24.npz", cases.radio_case(10, 8, 24, seed=35, N_K=3), False)
    save_smoother("smoother_info_radio_n24.npz", cases.radio_case(10, 8, 24, seed=35, N_K=3), True)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
