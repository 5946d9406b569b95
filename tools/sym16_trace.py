"""One filter run at BASELINE.json configs[4]'s per-GPU share for `rocprofv3 --kernel-trace --stats`:  python tools/sym16_trace.py [storage] [lazy] [N] [steps]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    storage = sys.argv[1] if len(sys.argv) > 1 else "fp32sym"
    lazy = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
    dg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
    r, *_ = bench.filter_leg(pkg, dg, N, 1024, 3000, K, 4, 1, lazy, 0, storage)
    print(storage, lazy, N, r["value"] / 1e6, "M particle-steps/s", r["ms_per_step"], "ms per step")
