"""GPU: read-only steps of the filter through the family products (rbpf_options.family_products = 1, csrc/rbpf_family.hip):
P_base * [H_1' ... H_f'] per family of particles that share a stored covariance on the fp64 matrix cores, then the per-particle rest of
the step (step_sym_kernel<.., PX>) -- particleFilter.m:100-204 on block-lower storage with lazy_depth >= 2.

Parity bar as everywhere: resampling indices bit-exact, fp64 quantities within 1e-9 relative of the numpy oracle / the plain-C
restatement, and of the default path (one workgroup per particle) on the same random numbers."""
import importlib

import numpy as np
import pytest

import cases
import oracle_c
from test_gpu_filter import check_filter, rel

pytestmark = pytest.mark.gpu
RTOL = 1e-9


@pytest.mark.parametrize("lazy_depth,inplace", [(2, -1), (3, 1), (4, -1), (4, 1), (6, 1), (8, -1)])
def test_family_products_filter_matches_oracle(rbpf, lazy_depth, inplace):
    """slam-dense-mag m = 512 (nLin = 515), N = 8: read-only steps with 1 .. 7 pending sets through the family products (families of
    one to a few members, every stored tile with its transposed contribution, border rows and columns from block B), flushes as
    before, against the numpy oracle."""
    c = cases.mag_case(8, 19 if lazy_depth > 4 else 13, 512, seed=61)
    ref = cases.oracle_filter(c)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                              rng=cases.device_rng(rbpf, c), extras=True, lazy_depth=lazy_depth, inplace=inplace, storage="fp64sym",
                              family_products=1)
    check_filter(ref, out)


def test_family_products_against_the_c_restatement(rbpf, tmp_path_factory):
    """N = 512, T = 60, m = 512 on replayed random numbers (families of up to a dozen members: several passes of five over one matrix):
    all indices, weights, final maps and covariances of all particles against the plain-C restatement."""
    import bench
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    N, T, m = 512, 60, 512
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], cases.THETA_MAG)
    rs = np.random.RandomState(91)
    rng = rbpf.ReplayRNG(rs.random_sample((1, T - 1, N)), rs.standard_normal((1, T - 1, N, 6)))
    lib = oracle_c.build(native_dir=str(tmp_path_factory.mktemp("oracle_native_fam")))
    ref, _ = oracle_c.particle_filter(rbpf, mdl, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng,
                                      n_threads=bench.usable_cores(), want_full=True, lib_path=lib)
    for lazy_depth, inplace in ((4, -1), (7, 1)):
        out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng=rng,
                                  extras=True, lazy_depth=lazy_depth, inplace=inplace, storage="fp64sym", family_products=1)
        ex = out[8]
        np.testing.assert_array_equal(ex["ai"][1:], ref["trace_ai"].T[1:])
        assert rel(ex["w"], ref["trace_w"].T) <= RTOL
        assert rel(out[0], ref["traj_max"]) <= RTOL and rel(out[1], ref["traj_mean"]) <= RTOL
        assert rel(out[2], ref["xl_max"]) <= RTOL and rel(out[3], ref["xl_mean"]) <= RTOL
        assert rel(out[4], ref["P_max"]) <= RTOL and rel(out[5], ref["P_mean"]) <= RTOL
        assert rel(ex["xl"], ref["final_xl"]) <= RTOL and rel(ex["P"], ref["final_P"]) <= RTOL


def test_family_products_equal_the_default_path_on_philox_streams(rbpf):
    """N = 4096, m = 512, 21 steps, lazy_depth 4, two banks with the shared flush: same resampling indices as the default path, outputs to
    1e-11 (same algebra, sums in another order), two runs bit-identical (the product of a member does not depend on who else is in
    its family: fixed summation order per particle)."""
    from test_gpu_configs import check_filter_properties, mag_inputs, run_session
    N, steps = 4096, 21
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 512)
    want = ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "trace_ai", "xl_mean")
    base = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=-1, storage="fp64sym")
    a = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=-1, storage="fp64sym", family_products=1)
    a2 = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=-1, storage="fp64sym", family_products=1)
    check_filter_properties(a, N, steps, P0)
    np.testing.assert_array_equal(a["trace_ai"], base["trace_ai"])
    for k in want:
        sl = (slice(None), slice(0, steps)) if k in ("traj_max", "traj_mean", "trace_w") else Ellipsis
        if k != "trace_ai":
            assert rel(a[k][sl], base[k][sl]) <= 1e-11, k
        np.testing.assert_array_equal(a[k], a2[k], err_msg=k)
    assert np.any(a["trace_w"][:, :steps] != base["trace_w"][:, :steps])         # (it IS another path: not bit-identical to the default)


def test_family_products_at_configs2_size(rbpf):
    """BASELINE.json configs[2], filter: N = 65 536, lazy_depth 4, two banks: properties hold and the run agrees with the default path
    (indices equal, outputs 1e-10) over ten steps."""
    from test_gpu_configs import check_filter_properties, mag_inputs, run_session
    N, steps = 65536, 10
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 512)
    want = ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "trace_ai", "xl_mean")
    a = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=0, storage="fp64sym", family_products=1)
    check_filter_properties(a, N, steps, P0)
    b = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=0, storage="fp64sym")
    np.testing.assert_array_equal(a["trace_ai"], b["trace_ai"])
    for k in want:
        if k != "trace_ai":
            sl = (slice(None), slice(0, steps)) if k in ("traj_max", "traj_mean", "trace_w") else Ellipsis
            assert rel(a[k][sl], b[k][sl]) <= 1e-10, k
