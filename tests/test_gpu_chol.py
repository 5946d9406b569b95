"""GPU parity of the batched ancestor-weight factorisation (particleSmoother.m:221-229,
particleSmootherInformationForm.m:224-236) on its own: both kernels (16-column, 64-column) against numpy's Cholesky,
the jitter retry, the failure flag, and the smoothers with the 64-column kernel forced at small sizes."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
LOG2PI = 1.8378770664093453


def spd_batch(B, M, seed, scale_spread=0.0):
    rs = np.random.RandomState(seed)
    A = rs.standard_normal((B, M, M + 8))
    S = A @ np.transpose(A, (0, 2, 1)) / (M + 8) + 0.5 * np.eye(M)
    if scale_spread:                                     # badly scaled diagonal, like diag(1./diag(P0)) of the information form
        s = 10.0 ** rs.uniform(-scale_spread, scale_spread, (B, M))
        S = S * s[:, :, None] * s[:, None, :]
    e = rs.standard_normal((B, M))
    return S, e


def numpy_logw(S, e, jitter=None):
    out = np.empty(S.shape[0])
    for b in range(S.shape[0]):
        try:
            L = np.linalg.cholesky(S[b])
        except np.linalg.LinAlgError:
            if jitter is None:
                out[b] = np.nan
                continue
            try:
                L = np.linalg.cholesky(S[b] + jitter * np.eye(S.shape[1]))     # particleSmoother.m:223
            except np.linalg.LinAlgError:
                out[b] = np.nan
                continue
        import scipy.linalg as sl
        v = sl.solve_triangular(L, e[b], lower=True)
        out[b] = -np.sum(np.log(np.diag(L))) - 0.5 * v @ v - 0.5 * S.shape[1] * LOG2PI
    return out


@pytest.mark.parametrize("variant", [16, 648, 644])
@pytest.mark.parametrize("M", [5, 16, 63, 64, 65, 130, 259, 515, 516, 576, 640])
def test_batched_factorisation_matches_numpy(rbpf, M, variant):
    B = 9 if M > 300 else 21
    S, e = spd_batch(B, M, seed=M)
    got, status, _ = rbpf.chol_weights(S, e, jitter=1e-2, variant=variant)
    want = numpy_logw(S, e)
    assert status == 0
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("variant", [16, 64])
def test_factorisation_of_badly_scaled_matrices(rbpf, variant):
    """Row/column scales spread over 8 decades (the information form starts from diag(1./diag(P0)))."""
    S, e = spd_batch(7, 259, seed=5, scale_spread=4.0)
    got, status, _ = rbpf.chol_weights(S, e, variant=variant)
    want = numpy_logw(S, e)
    assert status == 0
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-8)


@pytest.mark.parametrize("variant", [16, 64])
@pytest.mark.parametrize("M", [40, 300])
def test_jitter_retry_and_failure_flag(rbpf, M, variant):
    """particleSmoother.m:221-224: a failed chol is retried once with S + jitter*I; a second failure is an error."""
    S, e = spd_batch(6, M, seed=3)
    w, V = np.linalg.eigh(S[1])
    w[0] = -1e-3                                                    # slightly indefinite: passes with jitter 1e-2
    S[1] = (V * w) @ V.T
    w, V = np.linalg.eigh(S[4])
    w[0] = -5.0                                                     # fails twice
    S[4] = (V * w) @ V.T
    got, status, _ = rbpf.chol_weights(S, e, jitter=1e-2, variant=variant)
    want = numpy_logw(S, e, jitter=1e-2)
    assert status & 2
    assert np.isnan(got[4]) and np.isnan(want[4])
    ok = [0, 1, 2, 3, 5]
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-9, atol=1e-8)


def test_largest_supported_size(rbpf):
    S, e = spd_batch(3, 1023, seed=11)
    for variant in (16, 648, 644):
        got, status, _ = rbpf.chol_weights(S, e, variant=variant)
        assert status == 0
        np.testing.assert_allclose(got, numpy_logw(S, e), rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("info_form", [False, True])
@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 6, 5, 130), ("radio", 9, 7, 128)])
def test_smoothers_with_the_64_column_kernel_forced(rbpf, kind, N_P, N_T, m, info_form):
    """rbpf_options.chol_variant = 64 sends every ancestor-weight factorisation of the run through the 64-column kernel."""
    import test_gpu_smoother as ts
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=21, N_K=3)
    ref, out = ts.run_both(rbpf, c, info_form=info_form, chol_variant=64)
    ts.check(ref, out, 3)


def test_smoother_with_the_16_column_kernel_forced(rbpf):
    """chol_variant = 16 at a size the automatic choice gives to the register-resident kernel (n = 128)."""
    import test_gpu_smoother as ts
    c = cases.radio_case(9, 7, 128, seed=22, N_K=2)
    ref, out = ts.run_both(rbpf, c, info_form=True, chol_variant=16)
    ts.check(ref, out, 2)
    with pytest.raises(rbpf.RBPFError):
        ts.run_both(rbpf, c, info_form=True, chol_variant=7)


@pytest.mark.parametrize("kind,m,variant", [("mag", 256, 16), ("mag", 200, 16), ("mag", 256, 644), ("mag", 256, 648), ("mag", 173, 16)])
def test_packed_information_matrices_through_every_loader(rbpf, kind, m, variant):
    """nLin >= 176: the banks, Imat0, ImatAddt hold the information matrices in packed block-lower storage (imat_packed_index).
    The 16-column kernel reads them through the general loader (chol_aug_elems), the 64-column kernel through its call-free strip /
    diagonal-block / last-row-tile loaders as well -- every path against the oracle (nLin = 259: no partial last tile row;
    203: 11 valid rows in it; 176 = the first packed size, a right-hand-side row alone in its tile)."""
    import test_gpu_smoother as ts
    c = cases.mag_case(7, 6, m, seed=23, N_K=3)
    ref, out = ts.run_both(rbpf, c, info_form=True, chol_variant=variant)
    ts.check(ref, out, 3)


@pytest.mark.parametrize("m", [256, 300])
def test_information_form_smoother_at_the_benchmark_basis_sizes(rbpf, m):
    """m = 256 (nLin = 259, 17 row tiles) and m = 300 (nLin = 303, 19 row tiles): the size class of the 4-wave shape of the
    64-column kernel (two workgroups per CU)."""
    import test_gpu_smoother as ts
    c = cases.mag_case(6, 5, m, seed=23, N_K=2)
    ref, out = ts.run_both(rbpf, c, info_form=True, chol_refresh=1)          # the factorisation at every step (0 = automatic carries the factors)
    ts.check(ref, out, 2)


@pytest.mark.parametrize("kind,N_P,N_T,m", [("mag", 4, 100, 16), ("radio", 5, 300, 24)])
def test_covariance_form_smoother_with_long_horizons(rbpf, kind, N_P, N_T, m):
    """particleSmoother.m:159-241 with d*(T-t) up to 297 / 299 stacked future measurements: the covariance form's
    S = dy*P*dy' + kron(I, R) goes through the 64-column kernel (more than 16 row tiles) for the early time steps and
    through the 16-column kernel for the late ones."""
    import test_gpu_smoother as ts
    mk = cases.mag_case if kind == "mag" else cases.radio_case
    c = mk(N_P, N_T, m, seed=31, N_K=2)
    ref, out = ts.run_both(rbpf, c, info_form=False)
    ts.check(ref, out, 2)


def numpy_logw_info(S, e):
    import scipy.linalg as sl
    out = np.empty(S.shape[0])
    for b in range(S.shape[0]):
        L = np.linalg.cholesky(S[b])
        v = sl.solve_triangular(L, e[b], lower=True)
        out[b] = -np.sum(np.log(np.diag(L))) + 0.5 * v @ v                   # particleSmootherInformationForm.m:234-236
    return out


@pytest.mark.parametrize("variant", [1, 10, 11, 12, 14, 16, 644])
@pytest.mark.parametrize("M", [64, 65, 79, 80, 100, 127, 128, 129, 143])
def test_information_form_factorisation_of_small_matrices(rbpf, M, variant):
    """5..9 row tiles: the register-resident kernel (variant 1; 11 / 12 / 14 = one / two / four waves per matrix) against numpy
    and the two other kernels."""
    S, e = spd_batch(33, M, seed=100 + M, scale_spread=1.0)
    got, status, _ = rbpf.chol_weights(S, e, variant=variant, info_form=True)
    assert status == 0
    np.testing.assert_allclose(got, numpy_logw_info(S, e), rtol=1e-11, atol=1e-9)


def test_information_form_failure_is_flagged(rbpf):
    """particleSmootherInformationForm.m:224-236 has no usable retry (quirk Q4): a failed factorisation is an error."""
    S, e = spd_batch(5, 128, seed=9)
    w, V = np.linalg.eigh(S[2])
    w[0] = -1.0
    S[2] = (V * w) @ V.T
    got, status, _ = rbpf.chol_weights(S, e, variant=1, info_form=True)
    assert status & 2 and np.isnan(got[2])
    ok = [0, 1, 3, 4]
    np.testing.assert_allclose(got[ok], numpy_logw_info(S[ok], e[ok]), rtol=1e-11, atol=1e-9)


def test_every_kernel_over_a_sweep_of_sizes(rbpf):
    """All residues of M modulo 16 and 64 around the tile / block-column / dispatch boundaries, every kernel that accepts
    the size, both forms (the information form without retry)."""
    sizes = sorted(set(list(range(1, 20)) + list(range(60, 70)) + list(range(124, 150)) + [175, 176, 177, 191, 192, 193,
                   207, 208, 255, 256, 257, 271, 272, 287, 288, 289, 319, 320, 321, 383, 384, 385, 431, 432, 433, 447, 448, 449]))
    for M in sizes:
        S, e = spd_batch(3, M, seed=1000 + M)
        want0, want1 = numpy_logw(S, e), numpy_logw_info(S, e)
        rt = (M + 16) // 16
        for variant in (0, 16, 644, 648):
            got, status, _ = rbpf.chol_weights(S, e, jitter=1e-2, variant=variant)
            assert status == 0, (M, variant)
            np.testing.assert_allclose(got, want0, rtol=1e-11, atol=1e-9, err_msg=f"M={M} variant={variant}")
        for variant in (0, 16, 644) + ((1,) if 5 <= rt <= 9 else ()):
            got, status, _ = rbpf.chol_weights(S, e, variant=variant, info_form=True)
            assert status == 0, (M, variant)
            np.testing.assert_allclose(got, want1, rtol=1e-11, atol=1e-9, err_msg=f"M={M} variant={variant} info")


@pytest.mark.parametrize("n,d", [(24, 1), (63, 3), (64, 3), (128, 1), (130, 3), (259, 3), (515, 3), (575, 3)])
def test_sweep_kernel_equals_a_fresh_factorisation(rbpf, n, d):
    """The carried-factor sweep (rbpf_chol_sweep.hpp) at kernel level against numpy: d rank-1 updates and d rank-1 downdates of
    an augmented Cholesky factor [L 0; z' *] in one pass == chol(A + U'U - V'V) with the forward solve of the updated right-hand
    side carried in the extra row, and the log-weight expression of particleSmootherInformationForm.m:234-236 out of the same
    pass.  Sizes cover one slot, the slot boundaries, the compact tail (n = 128, 515), both vector storages, and the maximum."""
    rs = np.random.RandomState(100 + n)
    G = rs.randn(n, n + 40)
    A = G @ G.T / (n + 40) + np.eye(n)                       # well-conditioned SPD
    b = rs.randn(n)
    V = 0.3 * rs.randn(d, n) / np.sqrt(n)                    # small enough that A - V'V stays positive definite
    U = rs.randn(d, n) / np.sqrt(n)
    eta = rs.randn(d)
    L = np.linalg.cholesky(A)
    Laug = np.zeros((n + 1, n + 1))
    Laug[:n, :n] = L
    Laug[n, :n] = np.linalg.solve(L, b)
    Lo, logw, status, ms = rbpf.chol_sweep_probe(Laug, U, V, eta, batch=3)
    assert status == 0
    A2 = A + U.T @ U - V.T @ V
    b2 = b + U.T @ eta - V.T @ eta
    L2 = np.linalg.cholesky(A2)
    z2 = np.linalg.solve(L2, b2)
    scale = np.max(np.abs(L2))
    np.testing.assert_allclose(Lo[:n, :n], L2, rtol=0, atol=1e-12 * scale * n)
    np.testing.assert_allclose(Lo[n, :n], z2, rtol=0, atol=1e-11 * max(1.0, np.max(np.abs(z2))) * n)
    want = -np.sum(np.log(np.diag(L2))) + 0.5 * z2 @ z2
    assert abs(logw - want) <= 1e-10 * max(1.0, abs(want))
    assert np.all(np.triu(Lo[:n, :n], 1) == 0.0)


def test_sweep_kernel_reports_a_lost_downdate(rbpf):
    """A downdate that makes the matrix indefinite: status bit 1, NaN weight (as the fresh factorisation reports a failed chol)."""
    n, d = 40, 1
    L = np.eye(n + 1)
    L[n, :n] = 0.1
    V = np.zeros((d, n)); V[0, 3] = 2.0                      # A - v v' has a negative pivot
    Lo, logw, status, ms = rbpf.chol_sweep_probe(L, np.zeros((d, n)), V, np.zeros(d))
    assert status & 2 and np.isnan(logw)
