"""The N>1 path on CPU: the exchange plan (pure host logic) and the all_to_all of whole particles,
exercised with world_size 2 and 3 over gloo.  Each rank owns a fake 'bank' whose rows encode their global
particle id; after the planned exchange every slot must find its ancestor's row at anc_bank in
[local bank | recv region] -- exactly what the HIP step kernel dereferences."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _plan_module():
    sys.path.insert(0, ROOT)
    import importlib
    return importlib.import_module("rao-blackwellized-slam-smoothing_amd.multigpu")


def test_plan_is_consistent_across_ranks_without_communication():
    mg = _plan_module()
    rs = np.random.RandomState(0)
    for world, nl in [(2, 16), (3, 10), (8, 64), (1, 9)]:
        for peaked in (False, True):
            N = world * nl
            w = rs.random_sample(N) ** (8 if peaked else 1)
            ai = rs.choice(N, size=N, p=w / w.sum())
            plans = [mg.build_plan(ai, r, world, nl) for r in range(world)]
            for g in range(world):
                pg = plans[g]
                assert pg.recv_counts[g] == 0 and pg.send_counts[g] == 0
                assert pg.anc_bank.min() >= 0 and pg.anc_bank.max() < nl + pg.recv_counts.sum()
                for r in range(world):
                    assert pg.recv_counts[r] == plans[r].send_counts[g]          # both sides agree
                # unique: nobody receives a particle twice
                mine = ai[g * nl:(g + 1) * nl]
                assert pg.recv_counts.sum() == np.unique(mine[mine // nl != g]).size
            # emulate the exchange
            for g in range(world):
                recv_rows = []
                for r in range(world):
                    pr = plans[r]
                    off = int(pr.send_counts[:g].sum())
                    recv_rows.extend(r * nl + pr.send_idx[off:off + int(pr.send_counts[g])])
                space = np.concatenate((np.arange(g * nl, (g + 1) * nl), np.array(recv_rows, dtype=np.int64)))
                np.testing.assert_array_equal(space[plans[g].anc_bank], ai[g * nl:(g + 1) * nl])


def _worker(rank, world, port, nl, width, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mg = _plan_module()
        rs = np.random.RandomState(seed)                      # same stream on every rank -> same ai
        ok = True
        for step in range(4):
            N = world * nl
            w = rs.random_sample(N) ** (1 + 3 * step)
            ai = rs.choice(N, size=N, p=w / w.sum())
            plan = mg.build_plan(ai, rank, world, nl)
            gid = torch.arange(rank * nl, (rank + 1) * nl, dtype=torch.float64)
            bank = gid[:, None] * 1000.0 + torch.arange(width, dtype=torch.float64)[None, :] + step
            send = bank[torch.from_numpy(plan.send_idx.astype(np.int64))] if plan.send_idx.size else torch.empty((0, width), dtype=torch.float64)
            recv = torch.full((max(int(plan.recv_counts.sum()), 1), width), -1.0, dtype=torch.float64)
            send_buf = torch.empty((max(send.shape[0], 1), width), dtype=torch.float64)
            send_buf[:send.shape[0]] = send
            mg.exchange_rows(send_buf, recv, plan.send_counts, plan.recv_counts, dist)
            space = torch.cat((bank, recv[:int(plan.recv_counts.sum())]))
            got = space[torch.from_numpy(plan.anc_bank.astype(np.int64))]
            want_gid = torch.from_numpy(ai[rank * nl:(rank + 1) * nl].astype(np.float64))
            want = want_gid[:, None] * 1000.0 + torch.arange(width, dtype=torch.float64)[None, :] + step
            ok = ok and bool(torch.equal(got, want))
        flag = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            q.put(float(flag.item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 12, 5, 123, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 1.0
