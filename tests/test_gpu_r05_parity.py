"""GPU parity, round 5 (VERDICT r04 "Next round" item 1): the four holes that were still open under the metric.

(i)   Near-ties of the importance weights.  `iw_max` / `traj_max` are index work (particleFilter.m:159 takes the FIRST maximum), and
      in dense-radio siblings tie STRUCTURALLY (measModel sees the position only, run_dense2D_withHeading.m:168, which propagates
      without noise, :75-76).  Two statements are tested: siblings' log-weights are bit-identical on the device whatever slot computed
      them, so exact ties resolve as in the reference; and where two DIFFERENT particles' weights differ by less than fp64 evaluation
      noise (profiles/r04_size_scan.txt's "FAIL": radio m = 512, t = 3) the extended-precision arbiter decides -- it sides with the
      device and the C restatement, against the numpy oracle's rounding.
(ii)  The filter over the metric's horizon T = 3000 against the ARBITER (oracle/rbpf_oracle_c.c built with -DRBPF_ORACLE_LONG_DOUBLE;
      fixture tests/golden/arbiter_filter_N64_T3000_m512.npz, generated on the CPU by tests/golden/make_arbiter_fixture.py): every
      index identical, 1e-9 up to t = 500, and over the whole horizon err(HIP, arbiter) <= 1.25 x err(C fp64, arbiter) per quantity --
      the product is no further from the exact result of the reference's formulas than a correctly rounded fp64 restatement is.
(iii) particleSmootherInformationForm over a long horizon (T = 1000, m = 512, N_P = 64, N_K = 2, block-lower P, lazy_depth 3) with the
      from-scratch factorisation (chol_refresh 1) and with carried factors (32, the default): every ancestor, every draw, weights /
      ancestor probabilities / outputs by the same rule against the arbiter (particleSmootherInformationForm.m:186-335).
(iv)  N = 65 536, 10 steps, the same Philox streams: the bench's schedule (block-lower storage, lazy_depth 4, two banks, shared flush)
      against the reference's literal schedule (full-square storage rewritten every step): identical resampling indices, outputs 1e-10.

Tolerances: indices bit-exact; fp64 quantities 1e-9 relative where the problem's conditioning allows, otherwise the arbiter rule."""
import importlib
import os
import sys

import numpy as np
import pytest

import cases
import oracle_c
from test_gpu_configs import check_filter_properties, mag_inputs, rel, run_session

pytestmark = pytest.mark.gpu
RTOL = 1e-9
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
SLACK = 1.25                                # err(HIP, arbiter) <= SLACK x err(C fp64, arbiter)


def test_radio_siblings_tie_exactly_and_near_ties_follow_the_arbiter(rbpf):
    """(i) dense-radio, m = 512, N_P = 6, T = 7, seed 41 -- the case profiles/r04_size_scan.txt recorded as FAIL."""
    c = cases.radio_case(6, 7, 512, seed=41, N_K=2)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    arb, _ = oracle_c.particle_filter(rbpf, mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                                      cases.device_rng(rbpf, c), lib_path=oracle_c.build_arbiter())
    c64, _ = oracle_c.particle_filter(rbpf, mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                                      cases.device_rng(rbpf, c))
    for kw in ({}, dict(lazy_depth=3)):
        out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, c["N_P"], c["dt"],
                                  rng=cases.device_rng(rbpf, c), extras=True, **kw)
        ex = out[8]
        ai, logw, w = ex["ai"], ex["logw"], ex["w"]
        np.testing.assert_array_equal(ai[1:], arb["trace_ai"].T[1:])
        # siblings: same ancestor => same position (noise-free) => same H, same prior => the same weight, BIT FOR BIT
        n_sib = 0
        for t in range(1, ai.shape[0]):
            for a in np.unique(ai[t]):
                sib = np.flatnonzero(ai[t] == a)
                n_sib += len(sib) > 1
                assert len(set(float(v).hex() for v in logw[t, sib])) == 1, (t, a, [float(v).hex() for v in logw[t, sib]])
        assert n_sib >= 5
        # the arg-max of every step is the arbiter's (first maximum among exact ties); the device agrees with the C restatement too
        np.testing.assert_array_equal(np.argmax(w, axis=1), np.argmax(arb["trace_w"], axis=0))
        np.testing.assert_array_equal(np.argmax(w, axis=1), np.argmax(c64["trace_w"], axis=0))
        assert int(ex["iw_max"]) == int(arb["iw_max"][0])
        assert rel(out[0], arb["traj_max"]) <= RTOL and rel(out[1], arb["traj_mean"]) <= RTOL
        assert rel(w, arb["trace_w"].T) <= RTOL
    # what the r04 scan tripped over: at t = 3 particles 0 and 3 -- different lineages -- differ by less than the evaluations' noise
    tr = cases.oracle_filter(c)["trace"]
    gap_true = abs(arb["trace_logw"][0, 3] - arb["trace_logw"][3, 3])
    noise_numpy = np.max(np.abs(tr["logw"][3] - arb["trace_logw"][:, 3]))
    assert gap_true < noise_numpy                                    # the numpy oracle's order of the two is decided by its rounding
    assert int(np.argmax(arb["trace_w"][:, 3])) == 3 and int(np.argmax(tr["w"][3])) == 0


def load_fixture(name, d, *more):
    """The fixture and its measurements: y is taken from the fixture (the generator's BLAS product rounds differently on another host
    CPU, and one ulp on an input is 1e-9 on an output over these horizons); every other input is regenerated and checked."""
    import make_arbiter_fixture as maf
    fx = np.load(os.path.join(GOLDEN, name))
    assert np.max(np.abs(d["y"] - fx["y"])) <= 1e-12 * np.max(np.abs(fx["y"]))       # the same problem, up to the generator's rounding
    d["y"] = fx["y"]
    assert str(fx["inputs_sha256"]) == maf.checksum(d["dx"], d["y"], d["initState"], *more), "the generators no longer produce the fixture's inputs: regenerate it"
    return fx


def test_filter_over_the_full_horizon_against_the_arbiter(rbpf):
    """(ii) m = 512, N = 64, T = 3000 (750 lazy cycles, the bean trajectory three times round): block-lower storage with lazy_depth 4
    in both bank schedules, and full-square storage rewritten every step (the reference's own schedule)."""
    import make_arbiter_fixture as maf
    N, T = 64, 3000
    d, mdl, x0, P0, R, rng = maf.filter_inputs()
    fx = load_fixture("arbiter_filter_N64_T3000_m512.npz", d, x0, P0, R, rng.U, rng.Z)
    rows = fx["P_rows"]
    ai_ref = fx["trace_ai"].astype(np.int32).T
    keys = ("trace_w", "traj_max", "traj_mean", "xl_max", "xl_mean", "final_xl", "traj_sample_iwmax", "P_max_rows", "P_mean_rows", "P_max_diag")
    report = {}
    for tag, kw in (("sym_lazy4_two_banks", dict(lazy_depth=4, inplace=-1, storage="fp64sym")),
                    ("sym_lazy4_in_place", dict(lazy_depth=4, inplace=1, storage="fp64sym")),
                    ("full_square_every_step", dict())):
        out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng=rng,
                                  extras=True, **kw)
        ex = out[8]
        np.testing.assert_array_equal(ex["ai"][1:], ai_ref[1:])                  # 2999 x 64 resampling indices
        assert int(ex["iw_max"]) == int(fx["iw_max"][0])
        got = dict(trace_w=ex["w"].T, traj_max=out[0], traj_mean=out[1], xl_max=out[2], xl_mean=out[3], final_xl=ex["xl"],
                   traj_sample_iwmax=out[6], P_max_rows=out[4][rows, :], P_mean_rows=out[5][rows, :], P_max_diag=np.diag(out[4]))
        assert rel(got["trace_w"][:, :500], fx["trace_w"][:, :500]) <= RTOL      # the well-conditioned stretch: north_star's 1e-9
        assert rel(out[0][:, :500], fx["traj_max"][:, :500]) <= RTOL and rel(out[1][:, :500], fx["traj_mean"][:, :500]) <= RTOL
        for k in keys:
            e_hip, e_c = rel(got[k], fx[k]), float(fx["err_c64_nofma__" + k])
            report[(tag, k)] = (e_hip, e_c)
            assert e_hip <= max(RTOL, SLACK * e_c), (tag, k, e_hip, e_c)
            # block-lower storage keeps P exactly symmetric -- the plain form's drift is the growth of its asymmetric part -- and holds
            # north_star's 1e-9 against the arbiter over the WHOLE horizon (measured 1e-11 on the weights, where fp64 C is at 7.6e-9)
            if tag.startswith("sym_"):
                assert e_hip <= RTOL, (tag, k, e_hip)
    print({f"{t}:{k}": f"{a:.2e} (C {b:.2e})" for (t, k), (a, b) in report.items()})


@pytest.mark.parametrize("chol_refresh,inplace", [(1, 0), (32, 0), (10 ** 6, 1)])
def test_information_form_smoother_over_a_long_horizon_against_the_arbiter(rbpf, chol_refresh, inplace):
    """(iii) T = 1000, m = 512, N_P = 64, N_K = 2, block-lower P, lazy_depth 3; chol_refresh 1 = chol(Imat_i + ImatAddt) at every step
    (the reference's arithmetic), 32 = carried factors with 31 refreshes along the run (the default), 10^6 = refresh-free (999 sweeps
    on end, no information matrix stored) with one covariance bank in place: the configuration that runs N_P = 65 536 on one GPU."""
    import make_arbiter_fixture as maf
    N, T, N_K = 64, 1000, 2
    d, mdl, x0, P0, R, rng = maf.smoother_inputs()
    fx = load_fixture("arbiter_info_smoother_N64_T1000_m512.npz", d, x0, P0, R, rng.U, rng.Z, rng.Ufin)
    assert rbpf.chol_refresh_in_use(mdl, 0) == 32
    XNK, XLK, PK, ex = rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],
                                                            x0, P0, cases.Q_MAG, R, N, N_K, 0.01, rng=rng, extras=True, storage="fp64sym",
                                                            lazy_depth=3, chol_refresh=chol_refresh, inplace=inplace)
    np.testing.assert_array_equal(ex["ak"], fx["ak"])                            # the trajectory draws
    np.testing.assert_array_equal(ex["ai"][:, 1:], fx["ai"].astype(np.int32)[:, 1:])   # every ancestor incl. slot N_P's (:241)
    rows = fx["P_rows"]
    got = dict(w=ex["w"], paNt=ex["paNt"][1, 1:], XNK=XNK, XLK=XLK, PK_rows=PK[rows, :, :],
               PK_diag=np.stack([np.diag(PK[:, :, k]) for k in range(N_K)], axis=1))
    report = {}
    for k, v in got.items():
        ref = fx[k]
        if k == "paNt" and ref.shape[0] == T:                                    # stored with the (undefined) row of t = 0
            ref = ref[1:]
        e_hip = float(np.max(np.abs(v - ref))) if k == "paNt" else rel(v, ref)
        e_c = float(fx["err_c64_nofma__" + k])
        report[k] = (e_hip, e_c)
        # carried factors: the stated tolerance of the option on the ancestor probabilities (2e-9 absolute) next to the arbiter rule
        tol = max(RTOL, SLACK * e_c)
        if chol_refresh > 1 and k == "paNt":
            tol = max(tol, 2e-9)
        assert e_hip <= tol, (k, e_hip, e_c)
    print({k: f"{a:.2e} (C {b:.2e})" for k, (a, b) in report.items()})


def test_bench_schedule_equals_the_literal_schedule_at_configs2_size(rbpf):
    """(iv) N = 65 536, m = 512, 10 steps on the same Philox streams: block-lower storage / lazy_depth 4 / two banks / shared flush (what
    bench.py times; 2.96 x fewer bytes than SURVEY 8(d)'s formula) against full-square storage rewritten at every step
    (particleFilter.m:112-113,198 as written: 139 GB per bank).  Two full-square banks are 282 GB -- where they do not fit the single
    bank rewritten at every second step (lazy_depth 2, in place) stands in."""
    N, steps = 65536, 10
    d, mdl, x0, P0, R = mag_inputs(rbpf, 40, 512)
    want = ("traj_max", "traj_mean", "xl_max", "xl_mean", "P_max", "trace_w", "trace_ai")
    a = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=4, inplace=0, storage="fp64sym")
    check_filter_properties(a, N, steps, P0)
    literal = "full-square, every step, two banks"
    try:
        b = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=0, inplace=-1, storage="fp64")
    except rbpf.RBPFError as e:
        if e.status != rbpf.RBPF_ERR_OUT_OF_MEMORY:
            raise
        literal = "full-square, every second step, one bank in place"
        b = run_session(rbpf, d, mdl, x0, P0, R, N, steps, want, lazy_depth=2, inplace=1, storage="fp64")
    print("literal schedule:", literal)
    np.testing.assert_array_equal(a["trace_ai"], b["trace_ai"])
    for k in want:
        if k != "trace_ai":
            sl = (slice(None), slice(0, steps)) if k in ("traj_max", "traj_mean", "trace_w") else Ellipsis
            assert rel(a[k][sl], b[k][sl]) <= 1e-10, (k, rel(a[k][sl], b[k][sl]))


@pytest.mark.parametrize("form,opts", [("info", dict(chol_refresh=1)), ("info", dict()), ("info", dict(lazy_depth=3, chol_refresh=16)), ("cov", dict())])
def test_radio_smoothers_against_the_arbiter(rbpf, form, opts):
    """dense-radio (BASELINE.json configs[3]'s family and basis size, m = 128), N_P = 64, T = 200, N_K = 3: particleSmootherInformationForm
    -- from scratch, library default (carried factors), and the bench's configs[3] options -- and particleSmoother against the
    arbiter: every ancestor and draw identical, weights / ancestor probabilities / outputs by the arbiter rule
    (run_dense2D_withHeading.m:75-77,168; particleSmoother.m:159-241; particleSmootherInformationForm.m:186-335)."""
    import make_arbiter_fixture as maf
    N, T, N_K = 64, 200, 3
    d, mdl, x0, P0, R, Q, rng = maf.radio_inputs()
    fx = load_fixture("arbiter_radio_smoothers_N64_T200_m128.npz", d, x0, P0, R, Q, rng.U, rng.Z, rng.Ufin)
    f = rbpf.particleSmootherInformationForm if form == "info" else rbpf.particleSmoother
    XNK, XLK, PK, ex = f(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, Q, R, N, N_K, 1.0, rng=rng,
                         extras=True, **opts)
    np.testing.assert_array_equal(ex["ak"], fx[form + "__ak"])
    np.testing.assert_array_equal(ex["ai"][:, 1:], fx[form + "__ai"].astype(np.int32)[:, 1:])
    got = dict(w=ex["w"], paNt=ex["paNt"][1:, 1:], XNK=XNK, XLK=XLK, PK=PK)
    carried = form == "info" and rbpf.chol_refresh_in_use(mdl, opts.get("chol_refresh", 0)) > 1
    report = {}
    for k, v in got.items():
        ref = fx[f"{form}__{k}"]
        e_hip = float(np.max(np.abs(v - ref))) if k == "paNt" else rel(v, ref)
        e_c = float(fx[f"{form}__err_c64_nofma__{k}"])
        report[k] = (e_hip, e_c)
        # The information form's ancestor weight is -qf/2 + v'v/2 + ... (:234-236): two quadratic forms of size ~1e5 whose difference is
        # O(1), one from the covariance (carried through the update algebraically on the device, a fresh P*ivec product in the
        # restatement), one from the factorisation.  Every fp64 evaluation of it sits at ~1e-9 absolute here (C: 8.6e-10, device
        # 1.3e-9): "no further from the exact result than twice a correctly rounded restatement" is the rule for this quantity.
        tol = max(RTOL, (2.0 if (form == "info" and k == "paNt") else SLACK) * e_c)
        if carried and k == "paNt":
            tol = max(tol, 2e-9)
        assert e_hip <= tol, (k, e_hip, e_c)
    print(form, opts, {k: f"{a:.2e} (C {b:.2e})" for k, (a, b) in report.items()})


@pytest.mark.parametrize("opts", [dict(), dict(lazy_depth=3), dict(lazy_depth=4, storage="fp64sym")])
def test_filter_at_configs1_basis_size_over_the_full_horizon_against_the_arbiter(rbpf, opts):
    """m = 256 (nLin = 259: BASELINE.json configs[1]'s basis size), N = 64, T = 3000: full-square storage rewritten every step, configs[1]'s
    own options (lazy_depth 3) and block-lower storage with four tile rows, against the arbiter: all indices, 1e-9 up to t = 500, the
    arbiter rule over the whole horizon."""
    import make_arbiter_fixture as maf
    N, T = 64, 3000
    d, mdl, x0, P0, R, rng = maf.filter_m256_inputs()
    fx = load_fixture("arbiter_filter_N64_T3000_m256.npz", d, x0, P0, R, rng.U, rng.Z)
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng=rng,
                              extras=True, **opts)
    ex = out[8]
    np.testing.assert_array_equal(ex["ai"][1:], fx["trace_ai"].astype(np.int32).T[1:])
    assert int(ex["iw_max"]) == int(fx["iw_max"][0])
    st = int(fx["trace_w_stride"])
    got = dict(trace_w=ex["w"].T[:, ::st], traj_mean=out[1], xl_max=out[2], xl_mean=out[3], final_xl=ex["xl"], P_max_diag=np.diag(out[4]),
               traj_max=out[0])
    assert rel(got["trace_w"][:, :500 // st], fx["trace_w"][:, :500 // st]) <= RTOL
    report = {}
    for k, v in got.items():
        e_hip, e_c = rel(v, fx[k]), float(fx["err_c64_nofma__" + k])
        report[k] = (e_hip, e_c)
        assert e_hip <= max(RTOL, SLACK * e_c), (k, e_hip, e_c)
    print(opts, {k: f"{a:.2e} (C {b:.2e})" for k, (a, b) in report.items()})


@pytest.mark.parametrize("opts", [dict(lazy_depth=4, inplace=-1, storage="fp64sym"), dict(lazy_depth=4, inplace=1, storage="fp64sym"),
                                  dict(lazy_depth=2, storage="fp64sym"), dict()])
def test_filter_at_sixteen_tile_rows_over_a_long_horizon_against_the_arbiter(rbpf, opts):
    """m = 1024 (nLin = 1027: BASELINE.json configs[4]'s basis size, sixteen tile rows of the block-lower storage, r05), N = 32, T = 1000:
    block-lower storage in both bank schedules and at lazy_depth 2, and the full square rewritten every step (the reference's own
    schedule), against the extended-precision arbiter: every index, 1e-9 up to t = 500, the arbiter rule over the whole horizon."""
    import make_arbiter_fixture as maf
    N, T = 32, 1000
    d, mdl, x0, P0, R, rng = maf.filter_m1024_inputs()
    fx = load_fixture("arbiter_filter_N32_T1000_m1024.npz", d, x0, P0, R, rng.U, rng.Z)
    out = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01, rng=rng,
                              extras=True, **opts)
    ex = out[8]
    np.testing.assert_array_equal(ex["ai"][1:], fx["trace_ai"].astype(np.int32).T[1:])
    assert int(ex["iw_max"]) == int(fx["iw_max"][0])
    st = int(fx["trace_w_stride"])
    got = dict(trace_w=ex["w"].T[:, ::st], traj_mean=out[1], xl_max=out[2], xl_mean=out[3], final_xl=ex["xl"], P_max_diag=np.diag(out[4]),
               traj_max=out[0])
    assert rel(got["trace_w"][:, :500 // st], fx["trace_w"][:, :500 // st]) <= RTOL
    report = {}
    for k, v in got.items():
        e_hip, e_c = rel(v, fx[k]), float(fx["err_c64_nofma__" + k])
        report[k] = (e_hip, e_c)
        assert e_hip <= max(RTOL, SLACK * e_c), (k, e_hip, e_c)
    print(opts, {k: f"{a:.2e} (C {b:.2e})" for k, (a, b) in report.items()})


def test_smoother_at_the_metrics_full_particle_count_on_one_gpu(rbpf):
    """particleSmootherInformationForm at the metric's N_P = 65 536, m = 512 on ONE GPU: factors carried and never refactorised (no
    information matrix stored: chol_refresh >= N_T), one block-lower covariance bank rewritten in place, lazy_depth 3 -- 3.7 MB of state
    per particle.  T = 8, N_K = 2: finite outputs, normalised weights and ancestor probabilities (particleSmootherInformationForm.m:243-245),
    ancestors in range, symmetric positive-definite PK; and the same run at N_P = 4096 equals the library default there (two banks,
    refreshes) in every index and to 1e-9 -- the full-size configuration is the small one's arithmetic."""
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    T, N_K = 8, 2
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = rbpf.dense_mag_prior(512, d["LL"], cases.THETA_MAG)
    go = lambda N, **kw: rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"],   # noqa: E731
                                                              x0, P0, cases.Q_MAG, R, N, N_K, 0.01, rng=rbpf.PhiloxRNG(9), extras=True,
                                                              storage="fp64sym", lazy_depth=3, **kw)
    a = go(4096)
    b = go(4096, chol_refresh=T, inplace=1)
    np.testing.assert_array_equal(a[3]["ai"][:, 1:], b[3]["ai"][:, 1:])
    np.testing.assert_array_equal(a[3]["ak"], b[3]["ak"])
    assert np.max(np.abs(a[3]["paNt"][1, 1:] - b[3]["paNt"][1, 1:])) <= 1e-9
    assert rel(b[0], a[0]) <= RTOL and rel(b[1], a[1]) <= RTOL and rel(b[2], a[2]) <= RTOL
    del a, b
    N = 65536
    XNK, XLK, PK, ex = go(N, chol_refresh=T, inplace=1)
    assert np.all(np.isfinite(XNK)) and np.all(np.isfinite(XLK)) and np.all(np.isfinite(PK))
    np.testing.assert_allclose(ex["w"].sum(axis=2), 1.0, rtol=0, atol=1e-12)
    assert ex["ai"][:, 1:].min() >= 0 and ex["ai"][:, 1:].max() < N
    np.testing.assert_allclose(ex["paNt"][1, 1:].sum(axis=1), 1.0, rtol=0, atol=1e-12)
    for k in range(N_K):
        P = PK[:, :, k]
        assert rel(P, P.T) < 1e-12 and np.linalg.eigvalsh(0.5 * (P + P.T)).min() > 0
        assert np.all(np.diag(P) <= np.diag(P0) * (1 + 1e-12))


def test_out_of_memory_names_the_configuration_that_fits(rbpf):
    """N_P = 65 536 at m = 512 with the default layout (two covariance banks, information matrices stored at the refreshes: 7.5 MB per
    particle = 490 GB) does not fit one GPU: RBPF_ERR_OUT_OF_MEMORY, and the message names the options that do
    (test_smoother_at_the_metrics_full_particle_count_on_one_gpu runs them).  A call made after the refusal works."""
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    d = dg.bean_6D(6, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=1)
    mdl, x0, P0, R = rbpf.dense_mag_prior(512, d["LL"], cases.THETA_MAG)
    with pytest.raises(rbpf.RBPFError) as ei:
        rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R,
                                             65536, 2, 0.01, rng=rbpf.PhiloxRNG(9), storage="fp64sym", lazy_depth=3)
    assert ei.value.status == rbpf.RBPF_ERR_OUT_OF_MEMORY
    assert "inplace = 1" in str(ei.value) and "chol_refresh >= N_T" in str(ei.value)
    # ... and the refusal leaves nothing behind: neither memory nor the runtime's sticky error (the next call's launches succeed)
    out = run_session(rbpf, d, mdl, x0, P0, R, 256, 5, ("traj_max", "trace_w"), lazy_depth=3, storage="fp64sym")
    assert np.all(np.isfinite(out["trace_w"][:, :5]))
