"""GPU: the family product (csrc/rbpf_family.hip) on its own -- P_base * [H_1' ... H_f'] for particles that share a stored covariance,
on the fp64 matrix cores, against numpy.  particleFilter.m:139-141,185-198 (P_i * H_i'); block-lower storage of
rbpf_options.storage = 2.  The filter's read-only steps through it: test_gpu_family_filter.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def pack_sym(P, CH):
    """Core part of a symmetric matrix -> block T of Layout::sym: lower block triangle in 64 x 64 tiles, tile (I, J) at
    (I (I + 1) / 2 + J) * 4096, element (r, c) at ((c % 64) / 2) * 128 + (r % 64) * 2 + c % 2."""
    nt = CH * (CH + 1) // 2
    T = np.zeros(nt * 4096)
    for I in range(CH):
        for J in range(I + 1):
            blk = P[64 * I:64 * I + 64, 64 * J:64 * J + 64]            # [r][c]
            t = blk.T.reshape(32, 2, 64).transpose(0, 2, 1)            # [pair][row][parity]
            T[(I * (I + 1) // 2 + J) * 4096:(I * (I + 1) // 2 + J + 1) * 4096] = t.reshape(-1)
    return T


def family_pht(rbpf, CH, mats, H, fam_start, fam_base, reps=1, replicate=1):
    """H [N][mc][3] (columns of H_i'); returns PHt [N][3][mc]."""
    lib = rbpf.load_library()
    mc = 64 * CH
    T = np.ascontiguousarray(np.stack([pack_sym(P, CH) for P in mats]))
    N, F = H.shape[0], len(fam_base)
    out = np.zeros((N, 3, mc))
    ms = C.c_double(0.0)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))                    # noqa: E731
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))                     # noqa: E731
    fs, fb = np.ascontiguousarray(fam_start, dtype=np.int32), np.ascontiguousarray(fam_base, dtype=np.int32)
    Hc = np.ascontiguousarray(np.transpose(H, (0, 2, 1)))                     # the entry takes [N][3][mc]
    lib.rbpf_probe_family_pht.argtypes = [C.c_int32] * 4 + [C.POINTER(C.c_double)] * 2 + [C.POINTER(C.c_int32)] * 2 + [C.c_int32] * 2 + \
        [C.POINTER(C.c_double)] * 2
    st = lib.rbpf_probe_family_pht(CH, len(mats), N, F, dp(T), dp(Hc), ip(fs), ip(fb), reps, replicate, dp(out), C.byref(ms))
    assert st == 0, rbpf.load_library().rbpf_last_error()
    return out, ms.value


@pytest.mark.parametrize("CH", [8, 4])
def test_family_product_matches_numpy(rbpf, CH):
    """Families of 1 .. 12 members (one, two and three passes of five), several stored matrices, every tile of the block triangle incl.
    the transposed contributions: P * H' to 1e-12 of the row sums' scale."""
    rs = np.random.RandomState(5 + CH)
    mc = 64 * CH
    mats = []
    for _ in range(3):
        A = rs.standard_normal((mc, mc))
        mats.append(A + A.T + np.diag(rs.uniform(1, 5, mc)))
    sizes = [1, 2, 5, 6, 12, 3, 10, 4]
    fam_start = np.concatenate(([0], np.cumsum(sizes)))
    fam_base = np.array([0, 1, 2, 0, 1, 2, 2, 0])
    N = int(fam_start[-1])
    H = rs.standard_normal((N, mc, 3))
    got, _ = family_pht(rbpf, CH, mats, H, fam_start, fam_base)
    for f, b in enumerate(fam_base):
        for i in range(fam_start[f], fam_start[f + 1]):
            want = (mats[b] @ H[i]).T                                           # [3][mc]
            scale = np.abs(mats[b]) @ np.abs(H[i])
            assert np.max(np.abs(got[i] - want) / scale.T) < 1e-13, (f, i)
