"""GPU: the particle-sharded information-form smoother equals the single-GPU smoother with N = world * N_local
particles bit for bit (and the oracle to 1e-9).  Two ranks share the box's one GPU over the host/gloo transport; a
world-size-1 RCCL group exercises the device transport."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case(kind, N, T, m, N_K):
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    return mk(N, T, m, seed=31, N_K=N_K)


def _worker(rank, world, port, backend, transport, kind, n_local, T, m, N_K, q, lazy_depth=0, chol_refresh=0):
    import importlib
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        rbpf = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
        mg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.multigpu")
        c = _case(kind, world * n_local, T, m, N_K)
        mdl, x0, P0, R = cases.device_model(rbpf, c)
        s = mg.ShardedSmootherSession(mdl, c["odometry"], c["y"], c["x0_nonLin"], x0, P0, c["Q"], R, n_local, N_K, c["dt"],
                                      rng=cases.device_rng(rbpf, c), rank=rank, world=world, transport=transport,
                                      lazy_depth=lazy_depth, chol_refresh=chol_refresh, force_collectives=(world == 1))
        XNK, XLK, PK = s.run()
        stats = dict(s.stats)
        aks = list(s.aks)
        s.close()
        q.put((rank, XNK, XLK, PK, aks, stats))
    finally:
        dist.destroy_process_group()


def _run(world, backend, transport, kind, n_local, T, m, N_K, lazy_depth=0, chol_refresh=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, transport, kind, n_local, T, m, N_K, q, lazy_depth, chol_refresh)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def _single(rbpf, kind, N, T, m, N_K, **opts):
    c = _case(kind, N, T, m, N_K)
    mdl, x0, P0, R = cases.device_model(rbpf, c)
    out = rbpf.particleSmootherInformationForm(mdl.dynModel, mdl.measModel, mdl.dynResNorm, c["odometry"], c["y"], c["x0_nonLin"],
                                               x0, P0, c["Q"], R, N, N_K, c["dt"], rng=cases.device_rng(rbpf, c), extras=True, **opts)
    return c, out


# m = 130 / 128: the register-resident factorisation kernel; m = 16: the 16-column kernel; m = 200 (nLin = 203, 13 row
# tiles): the 64-column kernel, whose element loaders then read received records
@pytest.mark.parametrize("kind,n_local,T,m,N_K", [("mag", 12, 8, 130, 3), ("mag", 40, 7, 16, 3), ("radio", 24, 10, 128, 3),
                                                  ("mag", 10, 6, 200, 2)])
def test_two_ranks_equal_single_gpu_smoother(rbpf, kind, n_local, T, m, N_K):
    # chol_refresh = 1: the from-scratch factorisation of every step (the factorisation kernels read received records); the carried
    # factors -- the default since r05 -- have their own sharded tests below
    res = _run(2, "gloo", "host", kind, n_local, T, m, N_K, chol_refresh=1)
    c, ref = _single(rbpf, kind, 2 * n_local, T, m, N_K, chol_refresh=1)
    for rank, XNK, XLK, PK, aks, stats in res:                 # every rank returns the full outputs
        np.testing.assert_array_equal(np.asarray(aks), ref[3]["ak"])
        np.testing.assert_array_equal(XNK, ref[0])             # bit for bit
        np.testing.assert_array_equal(XLK, ref[1])
        np.testing.assert_array_equal(PK, ref[2])
    assert res[0][5]["migrated"] > 0                           # particle records (incl. Imat) did cross ranks
    # and the oracle (same replayed random numbers)
    orc = cases.oracle_smoother(c, info_form=True)
    np.testing.assert_allclose(res[0][1], orc["XNK"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(res[0][2], orc["XLK"], rtol=1e-9, atol=1e-9 * np.max(np.abs(orc["XLK"])))
    np.testing.assert_allclose(res[0][3], orc["PK"], rtol=1e-9, atol=1e-9 * np.max(np.abs(orc["PK"])))


@pytest.mark.parametrize("kind,n_local,T,m,N_K,lazy_depth", [("mag", 24, 11, 130, 3, 3), ("mag", 16, 9, 256, 2, 2), ("radio", 40, 12, 128, 3, 3)])
def test_two_ranks_with_lazy_update_match_single_gpu_smoother(rbpf, kind, n_local, T, m, N_K, lazy_depth):
    """Sharded information-form smoother + multi-step lazy covariance update: a migrating particle's record carries its
    covariance with the pending downdates applied (another rounding point than the single-GPU flush), so the two agree to
    1e-9 instead of bit for bit; the oracle bounds both."""
    res = _run(2, "gloo", "host", kind, n_local, T, m, N_K, lazy_depth, chol_refresh=1)
    c, ref = _single(rbpf, kind, 2 * n_local, T, m, N_K, chol_refresh=1)
    for rank, XNK, XLK, PK, aks, stats in res:
        np.testing.assert_array_equal(np.asarray(aks), ref[3]["ak"])
        np.testing.assert_allclose(XNK, ref[0], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(XLK, ref[1], rtol=1e-9, atol=1e-9 * np.max(np.abs(ref[1])))
        np.testing.assert_allclose(PK, ref[2], rtol=1e-9, atol=1e-9 * np.max(np.abs(ref[2])))
    np.testing.assert_array_equal(res[0][1], res[1][1])        # the ranks agree with each other exactly
    np.testing.assert_array_equal(res[0][3], res[1][3])
    assert res[0][5]["migrated"] > 0
    orc = cases.oracle_smoother(c, info_form=True)
    np.testing.assert_allclose(res[0][1], orc["XNK"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(res[0][3], orc["PK"], rtol=1e-9, atol=1e-9 * np.max(np.abs(orc["PK"])))


@pytest.mark.parametrize("kind,n_local,T,m,N_K,lazy_depth,K", [("mag", 24, 14, 130, 3, 0, 4), ("mag", 24, 14, 130, 3, 3, 5),
                                                                ("radio", 40, 16, 128, 3, 3, 4), ("mag", 12, 9, 256, 2, 2, 3),
                                                                ("mag", 14, 12, 256, 2, 3, 10 ** 6), ("radio", 30, 14, 128, 3, 2, 10 ** 6)])   # never refreshed: no Imat stored / exchanged
def test_two_ranks_with_carried_factors(rbpf, kind, n_local, T, m, N_K, lazy_depth, K):
    """chol_refresh = K in the sharded smoother: the factors migrate inside the particle records, the refreshes fetch base
    matrices from the other rank.  Same ancestor draws and trajectory as the single-GPU smoother with the same option (which
    tests/test_gpu_chol_carry.py holds to the fresh factorisation), outputs to 1e-9, and the oracle to 1e-9."""
    res = _run(2, "gloo", "host", kind, n_local, T, m, N_K, lazy_depth, K)
    c, ref = _single(rbpf, kind, 2 * n_local, T, m, N_K, lazy_depth=lazy_depth, chol_refresh=K)
    for rank, XNK, XLK, PK, aks, stats in res:
        np.testing.assert_array_equal(np.asarray(aks), ref[3]["ak"])
        np.testing.assert_allclose(XNK, ref[0], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(XLK, ref[1], rtol=1e-9, atol=1e-9 * np.max(np.abs(ref[1])))
        np.testing.assert_allclose(PK, ref[2], rtol=1e-9, atol=1e-9 * np.max(np.abs(ref[2])))
        assert stats["refreshes"] == (N_K - 1) * (1 + (T - 2) // K)
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][5]["migrated"] > 0
    if K < T - 1:
        assert res[0][5]["refresh_fetched"] + res[1][5]["refresh_fetched"] > 0     # base matrices did cross ranks
    else:
        assert res[0][5]["refresh_fetched"] + res[1][5]["refresh_fetched"] == 0    # never refreshed: nothing to fetch, nothing stored
    orc = cases.oracle_smoother(c, info_form=True)
    np.testing.assert_allclose(res[0][1], orc["XNK"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(res[0][3], orc["PK"], rtol=1e-9, atol=1e-9 * np.max(np.abs(orc["PK"])))


def test_world_size_one_carried_factors_equal_single_gpu(rbpf):
    """World 1, RCCL device transport: the sharded code path with carried factors against the single-GPU entry point."""
    kind, n_local, T, m, N_K, K = "mag", 32, 12, 130, 2, 4
    res = _run(1, "nccl", "device", kind, n_local, T, m, N_K, 3, K)
    _, ref = _single(rbpf, kind, n_local, T, m, N_K, lazy_depth=3, chol_refresh=K)
    np.testing.assert_array_equal(np.asarray(res[0][4]), ref[3]["ak"])
    np.testing.assert_allclose(res[0][1], ref[0], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(res[0][3], ref[2], rtol=1e-9, atol=1e-9 * np.max(np.abs(ref[2])))


def test_two_ranks_above_8192_particles_equal_single_gpu(rbpf):
    """N_global > 8192: the replicated normalisation runs the multi-workgroup pipeline on every rank."""
    kind, n_local, T, m, N_K = "radio", 4200, 4, 8, 2
    res = _run(2, "gloo", "host", kind, n_local, T, m, N_K)
    _, ref = _single(rbpf, kind, 2 * n_local, T, m, N_K)
    for rank, XNK, XLK, PK, aks, stats in res:
        np.testing.assert_array_equal(np.asarray(aks), ref[3]["ak"])
        np.testing.assert_array_equal(XNK, ref[0])
        np.testing.assert_array_equal(XLK, ref[1])
        np.testing.assert_array_equal(PK, ref[2])


def test_world_size_one_rccl_smoother(rbpf):
    kind, n_local, T, m, N_K = "mag", 20, 6, 130, 2
    res = _run(1, "nccl", "device", kind, n_local, T, m, N_K)
    _, ref = _single(rbpf, kind, n_local, T, m, N_K)
    np.testing.assert_array_equal(res[0][1], ref[0])
    np.testing.assert_array_equal(res[0][2], ref[1])
    np.testing.assert_array_equal(res[0][3], ref[2])
