"""GPU: symmetric covariance storage (rbpf_options.storage = 2, rbpf_step_sym.hip) -- the lower block triangle of every P_i only
(particleFilter.m:198 keeps P symmetric up to rounding), 0.5625 n^2 stored elements at nLin = 515.

Parity bar as everywhere: resampling indices bit-exact, fp64 quantities within 1e-9 relative of the numpy oracle / the plain-C
restatement (particleFilter.m:100-218), and of the full-square storage on the same Philox streams."""
import ctypes as C
import importlib

import numpy as np
import pytest

import cases
import oracle_c
from test_gpu_filter import check_filter, rel

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def test_wave_reduction_primitive(rbpf):
    """v_permlane32_swap / v_permlane16_swap folds + DPP row rotations: four values per lane -> four lane sums."""
    lib = rbpf.load_library()
    lib.rbpf_probe_wave_reduce.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rs = np.random.RandomState(3)
    for trial in range(4):
        x = np.ascontiguousarray(rs.standard_normal((4, 64)) * 10.0 ** rs.randint(-3, 4, size=(4, 1)))
        if trial == 0:
            x = np.ascontiguousarray(np.arange(256, dtype=np.float64).reshape(4, 64))      # exact in any order
        out = np.zeros(4)
        st = lib.rbpf_probe_wave_reduce(x.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
        assert st == 0
        want = x.sum(axis=1)
        if trial == 0:
            np.testing.assert_array_equal(out, want)
        np.testing.assert_allclose(out, want, rtol=0, atol=1e-13 * np.abs(x).sum(axis=1).max())


def run_sym(rbpf, c, lazy_depth, inplace, storage="fp64sym"):
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    return rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                               rng=cases.device_rng(rbpf, c), extras=True, lazy_depth=lazy_depth, inplace=inplace, storage=storage)


@pytest.mark.parametrize("lazy_depth,inplace", [(0, -1), (2, -1), (3, -1), (3, 1), (4, -1), (4, 1), (5, -1), (6, 1), (8, -1), (8, 1)])
def test_symmetric_storage_filter_matches_oracle(rbpf, lazy_depth, inplace):
    """slam-dense-mag m = 512 (nLin = 515), 13 steps: every variant of step_sym_kernel -- t = 0 (no pending set), rewrite every
    step (lazy 0), read-only steps with 1..3 sets (epilogue correction P H' - KS (K' H')) and flushes with 2..4 sets, ping-pong
    banks and the single bank rewritten in place -- against the numpy oracle.  lazy_depth 5 .. 8 (symmetric storage only): read-only
    steps with up to 7 pending sets, flushes with up to 8 (the two tile rows of a wave one after the other: split flush)."""
    c = cases.mag_case(8, 19 if lazy_depth > 4 else 13, 512, seed=61)
    ref = cases.oracle_filter(c)
    out = run_sym(rbpf, c, lazy_depth, inplace)
    check_filter(ref, out)


def test_symmetric_storage_against_the_c_restatement(rbpf, tmp_path_factory):
    """N = 512, T = 60, m = 512 on replayed random numbers: symmetric storage, lazy_depth 4, in place (the bench configuration
    at small N) against the plain-C restatement -- all indices, weights, final maps and covariances of all particles."""
    import bench
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    N, T, m = 512, 60, 512
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], cases.THETA_MAG)
    rs = np.random.RandomState(91)
    rng = rbpf.ReplayRNG(rs.random_sample((1, T - 1, N)), rs.standard_normal((1, T - 1, N, 6)))
    lib = oracle_c.build(native_dir=str(tmp_path_factory.mktemp("oracle_native_sym")))
    ref, _ = oracle_c.particle_filter(rbpf, mdl, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng,
                                      n_threads=bench.usable_cores(), want_full=True, lib_path=lib)
    for lazy_depth, inplace in ((4, 1), (4, -1), (8, 1), (6, -1)):
        out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng=rng,
                                  extras=True, lazy_depth=lazy_depth, inplace=inplace, storage="fp64sym")
        ex = out[8]
        np.testing.assert_array_equal(ex["ai"][1:], ref["trace_ai"].T[1:])
        assert rel(ex["w"], ref["trace_w"].T) <= RTOL
        assert rel(out[0], ref["traj_max"]) <= RTOL and rel(out[1], ref["traj_mean"]) <= RTOL
        assert rel(out[2], ref["xl_max"]) <= RTOL and rel(out[3], ref["xl_mean"]) <= RTOL
        assert rel(out[4], ref["P_max"]) <= RTOL and rel(out[5], ref["P_mean"]) <= RTOL
        assert rel(ex["xl"], ref["final_xl"]) <= RTOL and rel(ex["P"], ref["final_P"]) <= RTOL
        assert rel(out[4], out[4].T) < 1e-13                                    # one stored value per (r, c) / (c, r) pair (+ pending sets)


@pytest.mark.parametrize("lazy_depth", [3, 4])
def test_symmetric_storage_equals_full_storage_on_philox_streams(rbpf, lazy_depth):
    """N = 4096, m = 512, 21 steps on the device generator: same resampling indices as the full-square storage, outputs to
    1e-9; the in-place schedule gives the same indices and the same outputs to rounding as the ping-pong schedule (with two banks
    the children of one parent share ONE flushed matrix and the siblings of its writer run the read-only arithmetic at a flush
    step: shared flush); two runs are bit-identical."""
    from test_gpu_configs import check_filter_properties, mag_inputs, run_session
    N, steps = 4096, 21
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 512)
    want = ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "trace_ai", "xl_mean")
    full = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=lazy_depth, inplace=-1)
    a = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=lazy_depth, inplace=-1, storage="fp64sym")
    b = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=lazy_depth, inplace=1, storage="fp64sym")
    a2 = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=lazy_depth, inplace=-1, storage="fp64sym")
    check_filter_properties(a, N, steps, P0)
    np.testing.assert_array_equal(a["trace_ai"], full["trace_ai"])
    for k in ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "xl_mean"):
        sl = (slice(None), slice(0, steps)) if k in ("traj_max", "traj_mean", "trace_w") else Ellipsis   # NaN beyond the steps run
        assert rel(a[k][sl], full[k][sl]) <= RTOL, k
    np.testing.assert_array_equal(a["trace_ai"], b["trace_ai"])
    for k in want:
        sl = (slice(None), slice(0, steps)) if k in ("traj_max", "traj_mean", "trace_w") else Ellipsis
        if k != "trace_ai":
            assert rel(a[k][sl], b[k][sl]) <= 1e-11, k
        np.testing.assert_array_equal(a[k], a2[k], err_msg=k)


def test_symmetric_storage_at_configs2_size(rbpf):
    """BASELINE.json configs[2], filter: N = 65 536, m = 512, symmetric storage (78 GB per bank), lazy_depth 4: properties hold
    and two runs are bit-identical."""
    from test_gpu_configs import check_filter_properties, mag_inputs, run_session
    N, steps = 65536, 10
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 512)
    want = ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "trace_ai", "xl_mean")
    a = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=1, storage="fp64sym")
    check_filter_properties(a, N, steps, P0)
    b = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=1, storage="fp64sym")
    for k in want:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("lazy_depth,inplace", [(0, -1), (2, -1), (3, 1), (4, -1), (4, 1)])
def test_symmetric_storage_filter_at_m256_matches_oracle(rbpf, lazy_depth, inplace):
    """nLin = 259 (BASELINE.json configs[1]): four tile rows -- two waves share a row pair and split its column pairs by parity, their
    row sums meet in LDS -- against the numpy oracle."""
    c = cases.mag_case(9, 13, 256, seed=67)
    ref = cases.oracle_filter(c)
    out = run_sym(rbpf, c, lazy_depth, inplace)
    check_filter(ref, out)


def test_symmetric_storage_reduced_c2_against_the_c_restatement(rbpf, tmp_path_factory):
    """BASELINE.md "reduced C2" (N = 1024, T = 300, m = 256) on symmetric storage, lazy_depth 3, against the C restatement."""
    import bench
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    N, T, m = 1024, 300, 256
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], cases.THETA_MAG)
    rs = np.random.RandomState(77)
    rng = rbpf.ReplayRNG(rs.random_sample((1, T - 1, N)), rs.standard_normal((1, T - 1, N, 6)))
    lib = oracle_c.build(native_dir=str(tmp_path_factory.mktemp("oracle_native_sym256")))
    ref, _ = oracle_c.particle_filter(rbpf, mdl, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng,
                                      n_threads=bench.usable_cores(), want_full=False, lib_path=lib)
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng=rng,
                              want_xn_traj=False, lazy_depth=3, storage="fp64sym")
    assert rel(out[0], ref["traj_max"]) <= RTOL and rel(out[1], ref["traj_mean"]) <= RTOL
    assert rel(out[2], ref["xl_max"]) <= RTOL and rel(out[3], ref["xl_mean"]) <= RTOL


def test_information_form_smoother_on_symmetric_storage_at_m256(rbpf):
    import test_gpu_smoother as ts
    c = cases.mag_case(6, 8, 256, seed=71, N_K=3)
    ref = cases.oracle_smoother(c, True)
    out = _smooth(rbpf, c, True, storage="fp64sym", lazy_depth=3)
    ts.check(ref, out, 3)


def test_symmetric_storage_is_rejected_where_it_is_not_implemented(rbpf):
    c = cases.mag_case(6, 5, 125, seed=1)                                       # nLin = 128: two tile rows
    with pytest.raises(rbpf.RBPFError):
        run_sym(rbpf, c, 3, -1)
    c = cases.radio_case(6, 5, 256, seed=1, N_K=2)                              # dense-radio: nLin = 128 only (r05)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    with pytest.raises(rbpf.RBPFError):
        rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, 6, 1.0,
                            rng=cases.device_rng(rbpf, c), storage="fp64sym")


# ---- smoothers on symmetric storage (step_sym_kernel<.., E = 1>: P * ivec streamed, P * ivecPlus formed from the columns) ----------
def _smooth(rbpf, c, info_form, **kw):
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    f = rbpf.particleSmootherInformationForm if info_form else rbpf.particleSmoother
    return f(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["N_K"],
             c["dt"], rng=cases.device_rng(rbpf, c), extras=True, **kw)


@pytest.mark.parametrize("lazy_depth", [0, 2, 3])
def test_information_form_smoother_on_symmetric_storage_matches_oracle(rbpf, lazy_depth):
    """particleSmootherInformationForm.m:98-362 at nLin = 515 with the covariances in symmetric storage, N_K = 3, 8 steps (two lazy
    cycles of depth 3): ancestors and trajectory draws bit-exact, weights / ancestor probabilities / XNK / XLK / PK to 1e-9."""
    import test_gpu_smoother as ts
    c = cases.mag_case(6, 8, 512, seed=37, N_K=3)
    ref = cases.oracle_smoother(c, True)
    out = _smooth(rbpf, c, True, storage="fp64sym", lazy_depth=lazy_depth)
    ts.check(ref, out, 3)


def test_covariance_form_smoother_on_symmetric_storage_matches_oracle(rbpf):
    import test_gpu_smoother as ts
    c = cases.mag_case(6, 5, 512, seed=39, N_K=2)
    ref = cases.oracle_smoother(c, False)
    out = _smooth(rbpf, c, False, storage="fp64sym")
    ts.check(ref, out, 2)


def test_carried_factors_on_symmetric_storage(rbpf):
    """The bench's second smoother configuration (lazy_depth 3, chol_refresh) on symmetric storage at N_P = 2048, m = 512 against
    the reference's arithmetic (chol_refresh = 1) on full storage, same Philox streams: same ancestors and draws, paNt within 1e-9, outputs 1e-9."""
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    N, T = 2048, 10
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = rbpf.dense_mag_prior(512, d["LL"], cases.THETA_MAG)
    run = lambda **kw: rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],   # noqa: E731
                                                            x0, P0, cases.Q_MAG, R, N, 2, 0.01, rng=rbpf.PhiloxRNG(9), extras=True, **kw)
    a = run(chol_refresh=1)                               # from scratch at every step, full storage
    b = run(lazy_depth=3, chol_refresh=4, storage="fp64sym")
    np.testing.assert_array_equal(a[3]["ai"][:, 1:], b[3]["ai"][:, 1:])
    np.testing.assert_array_equal(a[3]["ak"], b[3]["ak"])
    assert np.max(np.abs(a[3]["paNt"][1, 1:] - b[3]["paNt"][1, 1:])) <= 1e-9
    assert rel(b[3]["w"], a[3]["w"]) <= 1e-9
    assert rel(b[0], a[0]) <= RTOL and rel(b[1], a[1]) <= RTOL and rel(b[2], a[2]) <= RTOL


def test_scheduled_bytes_count_distinct_matrices_and_shared_flush_writers(rbpf):
    """rbpf_timing.scheduled_bytes_per_launch on symmetric storage with two banks: every DISTINCT stored covariance a step reads
    (families of siblings / cousins share theirs) and, at a flush, one written matrix per parent with children (shared flush).
    Recomputed here from the ancestor trace of the same run: base slots propagate down the lineages, a flush step rebases every
    child on the smallest child of its parent."""
    from test_gpu_configs import mag_inputs
    N, C, W, K = 1024, 4, 5, 8
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 512)
    with rbpf.FilterSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng=rbpf.PhiloxRNG(5),
                            keep_history=True, trace=True, lazy_depth=C, inplace=-1, storage="fp64sym") as s:
        s.advance(W)
        s.sync()
        s.timing(enable=True)
        s.advance(K)
        s.sync()
        tm = s.timing(reset=True)
        ai = s.finish(want=("trace_ai",))["trace_ai"]
    n = mdl.nLin
    # stored doubles of one covariance: lower block triangle of 64 x 64 tiles over the 512 core rows + 3 border rows of ldb
    # (the library's own figure is not exposed: bound it from both sides instead)
    stored_lo, stored_hi = 0.5 * n * n * 8, 0.62 * n * n * 8
    base = np.arange(N)                                  # after step 0 (a flush) every particle sits on its own entry
    reads = writes = 0
    for t in range(1, W + K):
        anc = ai[:, t]
        ell = (t - 1) % C + 1
        src = base[anc]                                  # the matrix each child reads
        if t >= W:
            reads += np.unique(src).size
        if ell == C:                                     # flush: one entry per parent with children, the smallest child's
            lead = np.full(N, N, dtype=np.int64)
            np.minimum.at(lead, anc, np.arange(N))
            if t >= W:
                writes += np.unique(anc).size
            base = lead[anc]
        else:
            base = src
    fixed = 8.0 * N * K * (2.0 * n * 3 + 2.0 * n + 2.0 * 7)          # at least: one factor set out, means, states
    total = tm["scheduled_bytes_per_launch"] * tm["launches"]
    assert tm["launches"] == K
    assert (reads + writes) * stored_lo + fixed <= total <= (reads + writes) * stored_hi + 6.0 * fixed
    assert reads < 0.8 * N * K and writes < 0.8 * N * (K // C + 1)   # the sharing is real on this workload


# ---- sixteen tile rows (nLin = 1027, BASELINE.json configs[4]'s basis size) and fp32 tiles (rbpf_options.storage = 3), r05 ------------
@pytest.fixture(scope="module")
def oracle_m1024():
    c = cases.mag_case(6, 9, 1024, seed=29)
    return c, cases.oracle_filter(c)


@pytest.fixture(scope="module")
def oracle_m512_short():
    c = cases.mag_case(6, 9, 512, seed=29)
    return c, cases.oracle_filter(c)


@pytest.mark.parametrize("lazy_depth,inplace", [(0, -1), (2, -1), (3, 1), (4, -1), (4, 1)])
def test_symmetric_storage_filter_at_m1024_matches_oracle(rbpf, oracle_m1024, lazy_depth, inplace):
    """nLin = 1027: sixteen tile rows -- eight waves, rows {w, 15 - w}, the column sums of one block column staged in LDS and kept in
    the global strip workspace (rbpf_step_sym.hip) -- every variant of a lazy cycle, both bank schedules (shared flush with two banks),
    against the numpy oracle (particleFilter.m:100-218): indices bit-exact, fp64 quantities 1e-9."""
    c, ref = oracle_m1024
    check_filter(ref, run_sym(rbpf, c, lazy_depth, inplace))


@pytest.mark.parametrize("lazy_depth,inplace", [(0, -1), (2, -1), (3, 1), (4, -1), (4, 1)])
@pytest.mark.parametrize("m", [512, 1024])
def test_fp32_tiles_match_oracle_to_storage_precision(rbpf, oracle_m512_short, oracle_m1024, m, lazy_depth, inplace):
    """storage = "fp32sym": the lower block triangle in fp32, arithmetic fp64 -- the rounding of storage = "fp32" (6e-8 per stored
    element and rewrite) on the layout of "fp64sym".  Against the numpy oracle: tolerance 2e-5 as for "fp32"
    (tests/test_gpu_filter.py), the same resampling indices for this seed, and it really is another storage precision."""
    c, ref = oracle_m512_short if m == 512 else oracle_m1024
    out = run_sym(rbpf, c, lazy_depth, inplace, storage="fp32sym")
    ex, tr = out[8], ref["trace"]
    np.testing.assert_array_equal(ex["ai"][1:], tr["ai"][1:])
    TOL = 2e-5
    assert rel(ex["w"], tr["w"]) <= TOL
    assert rel(out[1], ref["traj_mean"]) <= TOL and rel(out[2], ref["xl_max"]) <= TOL and rel(out[4], ref["P_max"]) <= TOL
    assert rel(ex["xl"], tr["xl"]) <= TOL and rel(ex["P"], tr["P"]) <= TOL
    assert rel(ex["P"], tr["P"]) > 1e-12
    assert rel(out[4], out[4].T) < 1e-13                                        # one stored value per (r, c) / (c, r) pair


@pytest.mark.parametrize("m", [1021, 1148])
def test_sixteen_tile_rows_at_the_ends_of_the_supported_range(rbpf, m):
    """nLin = 1024 (no border row) and nLin = 1151 (127 border rows): the ends of the range the sixteen-tile-row kernel takes, fp64 tiles
    against the numpy oracle at 1e-9 and fp32 tiles at storage precision; nLin = 1152 is refused (tested with the other refusals)."""
    c = cases.mag_case(6, 6, m, seed=31)
    ref = cases.oracle_filter(c)
    for lazy_depth, inplace in ((0, -1), (3, -1), (4, 1)):
        check_filter(ref, run_sym(rbpf, c, lazy_depth, inplace))
    out = run_sym(rbpf, c, 3, -1, storage="fp32sym")
    np.testing.assert_array_equal(out[8]["ai"][1:], ref["trace"]["ai"][1:])
    assert rel(out[8]["P"], ref["trace"]["P"]) <= 2e-5 and rel(out[8]["w"], ref["trace"]["w"]) <= 2e-5


def test_sixteen_tile_rows_against_the_c_restatement(rbpf, tmp_path_factory):
    """N = 256, T = 40, m = 1024 on replayed random numbers: block-lower storage at sixteen tile rows, lazy_depth 4 with two banks
    (shared flush) and in place, lazy_depth 2, against the plain-C restatement -- every resampling index, weights, final maps and
    covariances of all particles (particleFilter.m:100-218)."""
    import bench
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    N, T, m = 256, 40, 1024
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], cases.THETA_MAG)
    rs = np.random.RandomState(93)
    rng = rbpf.ReplayRNG(rs.random_sample((1, T - 1, N)), rs.standard_normal((1, T - 1, N, 6)))
    lib = oracle_c.build(native_dir=str(tmp_path_factory.mktemp("oracle_native_sym16")))
    ref, _ = oracle_c.particle_filter(rbpf, mdl, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng,
                                      n_threads=bench.usable_cores(), want_full=True, lib_path=lib)
    for lazy_depth, inplace in ((4, -1), (4, 1), (2, -1)):
        out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng=rng,
                                  extras=True, lazy_depth=lazy_depth, inplace=inplace, storage="fp64sym")
        ex = out[8]
        np.testing.assert_array_equal(ex["ai"][1:], ref["trace_ai"].T[1:])
        assert rel(ex["w"], ref["trace_w"].T) <= RTOL
        assert rel(out[0], ref["traj_max"]) <= RTOL and rel(out[1], ref["traj_mean"]) <= RTOL
        assert rel(out[2], ref["xl_max"]) <= RTOL and rel(out[3], ref["xl_mean"]) <= RTOL
        assert rel(out[4], ref["P_max"]) <= RTOL and rel(out[5], ref["P_mean"]) <= RTOL
        assert rel(ex["xl"], ref["final_xl"]) <= RTOL and rel(ex["P"], ref["final_P"]) <= RTOL
        assert rel(out[4], out[4].T) < 1e-13


@pytest.mark.parametrize("storage,tol_banks", [("fp64sym", 1e-11), ("fp32sym", 2e-6)])
def test_sixteen_tile_rows_on_philox_streams(rbpf, storage, tol_banks):
    """N = 2048, m = 1024, 13 steps on the device generator, lazy_depth 4: the shared flush (two banks) and the single bank rewritten
    in place give the same indices and the same outputs to rounding (fp32 tiles: a last-bit difference of an fp64 value can move its
    fp32 rounding -- isolated elements differ by one fp32 ulp); two runs are bit-identical; "fp64sym" equals the full square to 1e-9
    with identical indices."""
    from test_gpu_configs import check_filter_properties, mag_inputs, run_session
    N, steps = 2048, 13
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 1024)
    want = ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "trace_ai", "xl_mean")
    a = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=-1, storage=storage)
    b = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=1, storage=storage)
    a2 = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=-1, storage=storage)
    check_filter_properties(a, N, steps, P0)
    np.testing.assert_array_equal(a["trace_ai"], b["trace_ai"])
    for k in want:
        sl = (slice(None), slice(0, steps)) if k in ("traj_max", "traj_mean", "trace_w") else Ellipsis
        if k != "trace_ai":
            assert rel(a[k][sl], b[k][sl]) <= tol_banks, k
        np.testing.assert_array_equal(a[k], a2[k], err_msg=k)
    full = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=-1, storage="fp64" if storage == "fp64sym" else "fp32")
    if storage == "fp64sym":
        np.testing.assert_array_equal(a["trace_ai"], full["trace_ai"])
    if np.array_equal(a["trace_ai"], full["trace_ai"]):
        for k in ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "xl_mean"):
            sl = (slice(None), slice(0, steps)) if k in ("traj_max", "traj_mean", "trace_w") else Ellipsis
            assert rel(a[k][sl], full[k][sl]) <= (RTOL if storage == "fp64sym" else 2e-5), k


def test_configs4_share_on_fp32_tiles(rbpf):
    """Per-GPU share of BASELINE.json configs[4] on the block-lower layout: N = 32 768, m = 1024, fp32 tiles (73 GB per bank, two banks,
    shared flush), lazy_depth 4: properties hold, two runs are bit-identical, and the session reports the schedule it chose."""
    from test_gpu_configs import check_filter_properties, mag_inputs, run_session
    N, steps = 32768, 7
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 1024)
    want = ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "trace_ai")
    a = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=0, storage="fp32sym")
    check_filter_properties(a, N, steps, P0)
    b = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=0, storage="fp32sym")
    for k in want:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    with rbpf.FilterSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng=rbpf.PhiloxRNG(5),
                            lazy_depth=4, inplace=0, storage="fp32sym") as s:
        assert s.schedule() == (2, True)                                         # two banks, shared flush


def test_block_lower_storage_is_refused_where_it_is_not_built(rbpf):
    """fp32 tiles and sixteen tile rows serve the filter; the smoothers, four tile rows in fp32 and dense-radio are refused by name."""
    c = cases.mag_case(6, 5, 1024, seed=29, N_K=2)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    for storage in ("fp64sym", "fp32sym"):
        with pytest.raises(rbpf.RBPFError) as ei:
            rbpf.particleSmoother(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, 6, 2, c["dt"],
                                  rng=cases.device_rng(rbpf, c), storage=storage)
        assert ei.value.status == rbpf.RBPF_ERR_UNSUPPORTED
    for m, storage in ((256, "fp32sym"), (1149, "fp64sym")):                    # fp32 tiles at four tile rows; nLin = 1152: eighteen tile rows
        c = cases.mag_case(6, 5, m, seed=29)
        mdl, x0, P0, R = cases.device_model(rbpf, c)
        with pytest.raises(rbpf.RBPFError) as ei:
            rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, 6, c["dt"],
                                rng=cases.device_rng(rbpf, c), storage=storage)
        assert ei.value.status == rbpf.RBPF_ERR_UNSUPPORTED
    c = cases.mag_case(6, 5, 512, seed=29, N_K=2)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    with pytest.raises(rbpf.RBPFError) as ei:
        rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, 6, 2,
                                             c["dt"], rng=cases.device_rng(rbpf, c), storage="fp32sym")
    assert ei.value.status == rbpf.RBPF_ERR_UNSUPPORTED


# ---- dense-radio (n_y = 1, nLin = m = 128: two tile rows, four waves share the one row pair), r05 ---------------------------------------
@pytest.mark.parametrize("lazy_depth,inplace", [(0, -1), (2, -1), (3, -1), (3, 1), (4, -1), (4, 1)])
def test_symmetric_storage_radio_filter_matches_oracle(rbpf, lazy_depth, inplace):
    """slam-dense-radio (run_dense2D_withHeading.m:75-76,168) on block-lower storage: three of the four 64 x 64 tiles, the four waves
    split the column pairs of the one row pair (four column phases), their row sums meet in LDS in a fixed order.  lazy_depth 0 is the
    one-set flush whose wave-private LDS stage needed compiler barriers (written as doubles, read as 16-byte pairs)."""
    c = cases.radio_case(8, 11, 128, seed=5)
    ref = cases.oracle_filter(c)
    check_filter(ref, run_sym(rbpf, c, lazy_depth, inplace))


@pytest.mark.parametrize("info_form,kw", [(False, {}), (True, {}), (True, dict(lazy_depth=3)), (True, dict(lazy_depth=3, chol_refresh=1))])
def test_symmetric_storage_radio_smoothers_match_oracle(rbpf, info_form, kw):
    """Both smoothers of dense-radio on block-lower storage, N_K = 3, against the numpy oracle (particleSmoother.m:124-341,
    particleSmootherInformationForm.m:98-362): carried factors (default) and the from-scratch factorisation."""
    import test_gpu_smoother as ts
    c = cases.radio_case(8, 9, 128, seed=7, N_K=3)
    ref, out = ts.run_both(rbpf, c, info_form=info_form, storage="fp64sym", **kw)
    ts.check(ref, out, 3)


def test_symmetric_storage_radio_at_configs3_size(rbpf):
    """N = 65 536 dense-radio particles on block-lower storage, lazy_depth 3, 12 steps on the device generator: the same resampling
    indices as the full square, outputs to 1e-9, two runs bit-identical."""
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    T, N, steps = 24, 65536, 12
    Qr = dg.radio_Q(T, "square_3D")
    th = [0.25, 2.0, 0.01]
    d = dg.planar_heading(T, Qr, th, 1.0, seed=1, nLL=4, traj="square_3D")
    mdl, x0, P0, R = rbpf.dense_radio_prior(128, d["LL"], th)
    want = ("traj_max", "traj_mean", "xl_max", "P_max", "trace_w", "trace_ai", "xl_mean")

    def go(storage):
        with rbpf.FilterSession(mdl, d["dx"], d["y"], d["initState"], x0, P0, Qr, R, N, 1.0, rng=rbpf.PhiloxRNG(5), keep_history=True, trace=True,
                                lazy_depth=3, storage=storage) as s:
            s.advance(steps)
            s.sync()
            return s.finish(want=want)
    a, a2, full = go("fp64sym"), go("fp64sym"), go("fp64")
    np.testing.assert_array_equal(a["trace_ai"], full["trace_ai"])
    for k in want:
        np.testing.assert_array_equal(a[k], a2[k], err_msg=k)
        if k != "trace_ai":
            sl = (slice(None), slice(0, steps)) if k in ("traj_max", "traj_mean", "trace_w") else Ellipsis
            assert rel(a[k][sl], full[k][sl]) <= RTOL, k


def test_symmetric_storage_radio_is_refused_at_other_sizes(rbpf):
    c = cases.radio_case(6, 5, 16, seed=1)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    with pytest.raises(rbpf.RBPFError) as ei:
        rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, 6, c["dt"],
                            rng=cases.device_rng(rbpf, c), storage="fp64sym")
    assert ei.value.status == rbpf.RBPF_ERR_UNSUPPORTED


# ---- one covariance bank with the shared flush (launch_share_inplace_plan), r05 ---------------------------------------------------------
@pytest.mark.parametrize("pattern", ["one_parent", "two_parents", "identity_like"])
@pytest.mark.parametrize("m,lazy_depth", [(512, 4), (512, 2), (1024, 3)])
def test_single_bank_shared_flush_on_extreme_ancestries(rbpf, pattern, m, lazy_depth):
    """The plan of the single-bank shared flush at its corners, against the numpy oracle on replayed random numbers: every particle
    drawing the SAME ancestor at every step (one writer overwrites the one live matrix in place, everybody else reads), two ancestors
    (two writers: one in place, one into a dead entry), and draws spread over all particles (about as many writers as particles).
    particleFilter.m:105-113 (the gather this replaces), tools/sample.m:30-32."""
    c = cases.mag_case(8, 10, m, seed=71)
    U = c["rng"].U
    if pattern == "one_parent":
        U[...] = 0.37
    elif pattern == "two_parents":
        U[..., :4] = 0.21
        U[..., 4:] = 0.83
    else:
        U[...] = (np.arange(8) + 0.5) / 8.0
    ref = cases.oracle_filter(c)
    out = run_sym(rbpf, c, lazy_depth, 1)
    check_filter(ref, out)
    ai = ref["trace"]["ai"][1:]
    if pattern == "one_parent":
        assert all(len(set(row)) == 1 for row in ai)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    with rbpf.FilterSession(mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, 8, c["dt"], rng=cases.device_rng(rbpf, c),
                            lazy_depth=lazy_depth, inplace=1, storage="fp64sym") as s:
        assert s.schedule() == (1, True)                                         # one bank, shared flush
