"""GPU: the in-library multi-device driver (rbpf_options.n_devices, csrc/rbpf_multi.hip) -- ONE host process shards the particles
over several GPUs: one C++ thread per device runs the sharded step loop and the library issues the collectives itself (RCCL, or
a host-staged transport when ranks share a GPU).  A gpurun box has one GPU, so the tests run

  * a world of one over RCCL (device_ids = [0]: the real ncclAllGather / grouped send-recv calls on the context's stream), and
  * a world of two sharing GPU 0 (device_ids = [0, 0]: the two-rank loop in two threads over the host-staged transport; particle
    records do cross ranks),

against the ordinary single-GPU entry points with the same N_P on the same Philox streams (keyed by logical slot, so sharding
does not change them): particleFilter.m:100-233 / particleSmootherInformationForm.m:98-362 outputs bit for bit (weighted means
over two partial sums and the lazy update: 1e-9)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _mag(rbpf, T, m, seed=3):
    import importlib
    dg = importlib.import_module(rbpf.__name__ + ".datagen")
    d = dg.bean_6D(T, cases.Q_MAG, cases.THETA_MAG, 0.01, seed=seed, m_sim=200)
    mdl, x0, P0, R = rbpf.dense_mag_prior(m, d["LL"], cases.THETA_MAG)
    return d, mdl, x0, P0, R


def _filter(rbpf, d, mdl, x0, P0, R, N, **kw):
    return rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01,
                               rng=rbpf.PhiloxRNG(11), want_xn_traj=False, **kw)


@pytest.mark.parametrize("ids", [[0], [0, 0]])
@pytest.mark.parametrize("m,N", [(130, 48), (256, 32)])
def test_multi_device_filter_equals_single_gpu(rbpf, m, N, ids):
    d, mdl, x0, P0, R = _mag(rbpf, 9, m)
    ref = _filter(rbpf, d, mdl, x0, P0, R, N)
    out = _filter(rbpf, d, mdl, x0, P0, R, N, n_devices=len(ids), device_ids=ids)
    for k in (0, 1, 2, 4, 6):                         # traj_max, traj_mean, xl_max, P_max, traj_sample_iwmax
        np.testing.assert_array_equal(out[k], ref[k], err_msg=str(k))
    np.testing.assert_allclose(out[3], ref[3], rtol=1e-9, atol=1e-12)      # xl_mean: sum of the ranks' shares
    np.testing.assert_allclose(out[5], ref[5], rtol=1e-9, atol=1e-12)      # P_mean (quirk Q3: the last particle's term)
    assert out[7] is None                                                    # want_xn_traj=False
    full = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01,
                               rng=rbpf.PhiloxRNG(11))
    multi = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01,
                                rng=rbpf.PhiloxRNG(11), n_devices=len(ids), device_ids=ids)
    np.testing.assert_array_equal(multi[7], full[7])                         # xn_traj from the replicated state history


def test_xn_traj_is_traced_in_chunks(rbpf):
    """rbpf_shard_xn_traj traces the ancestral paths through a scratch buffer of at most 256 MB: N = 8192, T = 600, nN = 7 is 275 MB of
    output, i.e. two chunks of paths, each copied into its columns of the host array; equal to the single-GPU xn_traj
    (particleFilter.m:117-118)."""
    d, mdl, x0, P0, R = _mag(rbpf, 600, 16)
    N = 8192
    full = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01,
                               rng=rbpf.PhiloxRNG(11))
    multi = rbpf.particleFilter(mdl.dynModel, mdl.measModel, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, 0.01,
                                rng=rbpf.PhiloxRNG(11), n_devices=1, device_ids=[0])
    assert multi[7].shape == (7, N, 600) and multi[7].nbytes > 256 << 20
    np.testing.assert_array_equal(multi[7], full[7])


@pytest.mark.parametrize("storage,lazy_depth,m", [("fp64", 3, 130), ("fp64sym", 4, 512), ("fp64sym", 0, 512)])
def test_multi_device_filter_with_lazy_update_and_symmetric_storage(rbpf, storage, lazy_depth, m):
    d, mdl, x0, P0, R = _mag(rbpf, 11, m)
    N = 48
    ref = _filter(rbpf, d, mdl, x0, P0, R, N)
    out = _filter(rbpf, d, mdl, x0, P0, R, N, n_devices=2, device_ids=[0, 0], lazy_depth=lazy_depth, storage=storage)
    for k in (0, 1, 2, 3, 4, 6):
        np.testing.assert_allclose(out[k], ref[k], rtol=1e-9, atol=1e-11, err_msg=str(k))


@pytest.mark.parametrize("ids", [[0], [0, 0], [0, 0, 0]])
def test_multi_device_information_form_smoother_equals_single_gpu(rbpf, ids):
    """particleSmootherInformationForm with the particles of every CPF-AS iteration sharded inside the library (records with the
    information state cross ranks; ancestor weights where each particle lives): XNK, XLK, PK bit for bit."""
    d, mdl, x0, P0, R = _mag(rbpf, 8, 130)
    N, N_K = 36, 3
    args = (mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, N_K, 0.01)
    ref = rbpf.particleSmootherInformationForm(*args, rng=rbpf.PhiloxRNG(7))
    out = rbpf.particleSmootherInformationForm(*args, rng=rbpf.PhiloxRNG(7), n_devices=len(ids), device_ids=ids)
    for a, b in zip(out, ref):
        np.testing.assert_array_equal(a, b)


def test_multi_device_radio_smoother_with_lazy_update(rbpf):
    c = cases.radio_case(40, 9, 128, seed=5, N_K=2)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    args = (mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, 40, 2, c["dt"])
    ref = rbpf.particleSmootherInformationForm(*args, rng=rbpf.PhiloxRNG(9))
    out = rbpf.particleSmootherInformationForm(*args, rng=rbpf.PhiloxRNG(9), n_devices=2, device_ids=[0, 0], lazy_depth=3)
    for a, b in zip(out, ref):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11)


def test_multi_device_argument_errors(rbpf):
    d, mdl, x0, P0, R = _mag(rbpf, 5, 20)
    with pytest.raises(rbpf.RBPFError):                                   # N_P not a multiple of n_devices
        _filter(rbpf, d, mdl, x0, P0, R, 9, n_devices=2, device_ids=[0, 0])
    with pytest.raises(rbpf.RBPFError):                                   # a device that does not exist
        _filter(rbpf, d, mdl, x0, P0, R, 8, n_devices=2, device_ids=[0, 63])
    with pytest.raises(rbpf.RBPFError):                                   # the covariance form is not sharded
        rbpf.host._smoother(False, mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, 8, 2,
                            0.01, False, None, rbpf.PhiloxRNG(1), False, n_devices=2, device_ids=[0, 0])


@pytest.mark.parametrize("lazy_depth", [0, 3])
def test_multi_device_smoother_on_symmetric_storage(rbpf, lazy_depth):
    """The sharded information-form smoother with symmetric covariance storage (records carry the lower block triangle; the packer
    applies pending sets over it) at nLin = 515, two ranks sharing the GPU, against the single-GPU smoother on full storage."""
    d, mdl, x0, P0, R = _mag(rbpf, 8, 512)
    N, N_K = 24, 2
    args = (mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, N, N_K, 0.01)
    ref = rbpf.particleSmootherInformationForm(*args, rng=rbpf.PhiloxRNG(7))
    out = rbpf.particleSmootherInformationForm(*args, rng=rbpf.PhiloxRNG(7), n_devices=2, device_ids=[0, 0], storage="fp64sym",
                                               lazy_depth=lazy_depth)
    for a, b in zip(out, ref):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("kind,ids,cap", [("radio", [0, 0], 0), ("mag", [0, 0, 0], 0), ("mag", [0], 0), ("radio", [0, 0], -1), ("mag", [0, 0, 0], -1)])
def test_multi_device_smoother_with_carried_factors(rbpf, kind, ids, cap):
    """chol_refresh = K in the in-library driver: the factors migrate inside the particle records, the refreshes fetch base matrices
    across ranks by the plan every rank derives from the replicated tables (rbpf_multi.hip plan_refresh == multigpu.plan_refresh).
    Against the single-GPU smoother with the same options: same trajectory draws, outputs to 1e-9."""
    if kind == "radio":
        c = cases.radio_case(40, 14, 128, seed=5, N_K=3)
        mdl, x0, P0, R = cases.device_model(rbpf, c)
        args = (mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, 40 // len(ids) * len(ids), 3, c["dt"])
    else:
        d, mdl, x0, P0, R = _mag(rbpf, 14, 130)
        args = (mdl.dynModel, mdl.measModel, mdl.dynResNorm, d["dx"], d["y"], d["initState"], x0, P0, cases.Q_MAG, R, 36, 3, 0.01)
    for kw in (dict(lazy_depth=3, chol_refresh=4),
               dict(lazy_depth=3, chol_refresh=10 ** 6)):           # never refreshed after t = 1: no information matrix stored or exchanged
        ref = rbpf.particleSmootherInformationForm(*args, rng=rbpf.PhiloxRNG(9), **kw)
        # exchange_capacity = -1: record AND refresh buffers start at one / two entries and grow on demand (every rank by the same rule)
        out = rbpf.particleSmootherInformationForm(*args, rng=rbpf.PhiloxRNG(9), n_devices=len(ids), device_ids=ids, exchange_capacity=cap, **kw)
        for a, b in zip(out, ref):
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11)
