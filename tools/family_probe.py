#!/usr/bin/env python3
"""Times the family product probe (csrc/rbpf_family.hip) at the headline's size: N = 65 536 particles, nLin = 515 (512 core rows),
families drawn like the bench's read-only steps (a given fraction of distinct stored matrices).
   python tools/family_probe.py [--distinct 0.33] [--N 65536]
Prints one JSON line: launch ms, bytes of distinct matrices streamed, TB/s, and the fp64 MFMA time the launch would need alone."""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=65536)
    ap.add_argument("--distinct", type=float, default=0.33)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import test_gpu_family_probe as tp
    rbpf = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    rs = np.random.RandomState(1)
    CH, mc, N = 8, 512, args.N
    F = int(args.distinct * N)
    # multinomial resampling: family sizes = counts of N draws over F equally likely parents, empty ones dropped
    cnt = np.bincount(rs.randint(0, F, N), minlength=F)
    cnt = cnt[cnt > 0]
    F = cnt.size
    fam_start = np.concatenate(([0], np.cumsum(cnt))).astype(np.int32)
    n_host = 64                                                     # host matrices, tiled on the device: every family reads its own
    mats = []
    for _ in range(n_host):
        A = rs.standard_normal((mc, mc))
        mats.append(A + A.T)
    rep = (F + n_host - 1) // n_host
    fam_base = np.arange(F, dtype=np.int32)
    H = rs.standard_normal((N, mc, 3))
    out, ms = tp.family_pht(rbpf, CH, mats, H, fam_start, fam_base, reps=args.reps, replicate=rep)
    by = F * 36 * 4096 * 8.0
    passes = int(np.sum((cnt + 4) // 5))
    print(json.dumps({"kernel": "family_pht_kernel<8>", "N": N, "families": int(F), "distinct_fraction": F / N, "largest_family": int(cnt.max()),
                      "matrix_passes": passes, "ms": ms, "GB_streamed": by / 1e9, "TBps": by / (ms * 1e-3) / 1e12,
                      "mfma_only_ms": passes * 36 * 2 * 64 * 64 / 4.0 / 256 / 2.4e9 * 1e3, "finite": bool(np.all(np.isfinite(out)))}))


if __name__ == "__main__":
    main()
