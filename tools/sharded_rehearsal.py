"""Rehearsal of the N-GPU smoother on ONE GPU: `world` ranks share the card over the host/gloo transport and run the sharded
information-form smoother at a realistic per-rank size, so that the exchange volumes that size the buffers (migrating records per
step, base matrices fetched per refresh of the carried factors) are measured rather than guessed.  Times are NOT representative
(records cross through host memory, the ranks time-share the GPU).

    python tools/sharded_rehearsal.py world=2 n_local=4096 m=512 T=100 lazy_depth=3 chol_refresh=32
"""
import importlib
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, o, q):
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
        mg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.multigpu")
        dg = importlib.import_module("rao-blackwellized-slam-smoothing_amd.datagen")
        bench = importlib.import_module("bench")
        Q = bench.q_mag()
        d = dg.bean_6D(o["T"], Q, bench.THETA_MAG, 0.01, seed=1)
        model, x0, P0, R = pkg.dense_mag_prior(o["m"], d["LL"], bench.THETA_MAG)
        s = mg.ShardedSmootherSession(model, d["dx"], d["y"], d["initState"], x0, P0, Q, R, o["n_local"], o["N_K"], 0.01,
                                      rng=pkg.PhiloxRNG(1), rank=rank, world=world, transport="host", lazy_depth=o["lazy_depth"],
                                      chol_refresh=o["chol_refresh"], exchange_capacity=o["exchange_capacity"])
        t0 = time.perf_counter()
        XNK, XLK, PK = s.run()
        dt = time.perf_counter() - t0
        st = {k: v for k, v in s.stats.items() if k != "phase_s"}
        st.update(rank=rank, seconds=dt, finite=bool(np.all(np.isfinite(XNK)) and np.all(np.isfinite(PK))), aks=list(s.aks),
                  refresh_capacity=s.refresh_capacity, send_capacity=int(s.v.send_capacity), recv_capacity=int(s.v.recv_capacity))
        s.close()
        st["XNK"] = XNK                                       # for tests; main() drops it before printing
        q.put(st)
    except Exception as exc:                                   # report, so that the parent does not wait for the timeout
        q.put({"rank": rank, "error": f"{type(exc).__name__}: {exc}"})
    finally:
        dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    o = dict(world=2, n_local=4096, m=512, T=100, N_K=2, lazy_depth=3, chol_refresh=32, exchange_capacity=0)
    for a in sys.argv[1:]:
        k, v = a.split("=")
        o[k] = int(v)
    sys.path.insert(0, ROOT)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, o["world"], port, o, q)) for r in range(o["world"])]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in procs), key=lambda r: r["rank"])
    for p in procs:
        p.join(60)
    for r in res:
        r.pop("XNK", None)
    steps = max(1, res[0].get("steps", 1))
    out = {"options": o, "ranks": res}
    if "error" not in res[0]:
        out["migrated_children_per_step"] = res[0]["migrated"] / steps
        out["records_sent_per_step_per_rank"] = [r["sent_records"] / steps for r in res]
        out["base_matrices_fetched_per_refresh_per_rank"] = [r["refresh_fetched"] / max(1, r["refreshes"]) for r in res]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
