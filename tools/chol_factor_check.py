#!/usr/bin/env python3
"""Diagnostic (tuning build): factorises one batch with a given kernel variant, dumps the factor workspace of matrix 0
(RBPF_CHOL_DUMP, fragment order) and compares it tile by tile with numpy's Cholesky factor of the augmented matrix.
   RBPF_LIB_PATH=.../librbpf_hip_tuning.so python tools/chol_factor_check.py --M 432 --variant 128"""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=432)
    ap.add_argument("--variant", type=int, default=128)
    args = ap.parse_args()
    M = args.M
    dump = f"/tmp/chol_dump_{M}_{args.variant}.bin"
    os.environ["RBPF_CHOL_DUMP"] = dump
    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    rs = np.random.RandomState(3)
    A = rs.standard_normal((2, M, M + 8))
    S = A @ np.transpose(A, (0, 2, 1)) / (M + 8) + 0.5 * np.eye(M)
    e = rs.standard_normal((2, M))
    logw, status, _ = pkg.chol_weights(S, e, variant=args.variant, info_form=True)
    L = np.linalg.cholesky(S[0])
    import scipy.linalg as sl
    v = sl.solve_triangular(L, e[0], lower=True)
    RT = (M + 16) // 16
    want = np.zeros((16 * RT, 16 * RT))
    want[:M, :M] = L
    want[M, :M] = v
    raw = np.fromfile(dump)
    KGS = 4 * RT
    got = np.zeros_like(want)
    fr = raw.reshape(RT, KGS, 4, 16)                       # [rt][kg][kk][r]
    for rt in range(RT):
        for kg in range(KGS):
            got[16 * rt:16 * rt + 16, 4 * kg:4 * kg + 4] = fr[rt, kg].T
    print("logw got", logw[0], "want", -np.sum(np.log(np.diag(L))) + 0.5 * v @ v, "status", status)
    bad = []
    for rt in range(RT):
        row = ""
        for ct in range(rt + 1):
            r1 = min(16 * rt + 16, M + 1)
            c1 = min(16 * ct + 16, M)
            g, w = got[16 * rt:r1, 16 * ct:c1], want[16 * rt:r1, 16 * ct:c1]
            if rt == ct:
                g, w = np.tril(g), np.tril(w)
            err = np.max(np.abs(g - w)) if g.size else 0.0
            row += "." if err < 1e-9 else ("x" if np.isfinite(err) else "N")
            if not err < 1e-9:
                bad.append((rt, ct, float(err)))
        print(f"{rt:3d} {row}")
    print("first bad tiles:", bad[:12])
    if bad and M >= 272:
        # reverse-engineer what the first wrong strip's panel product was: V_got = X_got Ld', D = A - V_got against P = L(r,:K) L(C,:K)'
        rt, ct, _ = bad[0]
        J2 = ct // 8
        C = slice(128 * J2, min(128 * J2 + 128, M))
        K = 128 * J2
        r = slice(16 * rt, min(16 * rt + 16, M + 1))
        Aaug = np.zeros_like(want); Aaug[:M, :M] = S[0]; Aaug[M, :M] = e[0]
        Ld = L[C, C]
        Xg = got[r, C]
        Vg = Xg @ Ld.T
        D = Aaug[r, C] - Vg
        P = want[r, :K] @ L[C, :K].T
        print("strip", rt, "super-block", J2, "|D-P|/|P|", np.linalg.norm(D - P) / np.linalg.norm(P), "|D|/|P|", np.linalg.norm(D) / np.linalg.norm(P))
        for c in range(D.shape[1] // 16):
            cs = slice(16 * c, 16 * c + 16)
            print("  sub-col", c, "|D-P|/|P|", np.linalg.norm(D[:, cs] - P[:, cs]) / np.linalg.norm(P[:, cs]), "|D|/|P|", np.linalg.norm(D[:, cs]) / np.linalg.norm(P[:, cs]))
        for rr in range(8 * J2, RT):
            Pr = want[16 * rr:16 * rr + 16, :K] @ L[C, :K].T
            if Pr.shape == D.shape:
                print("  B from row tile", rr, np.linalg.norm(D - Pr) / np.linalg.norm(P))
        for k0 in range(0, K, 16):
            Pk = want[r, k0:k0 + 16] @ L[C, k0:k0 + 16].T
            print("  K cols", k0, "projection coeff", float(np.sum(D * Pk) / np.sum(Pk * Pk)))


if __name__ == "__main__":
    main()
