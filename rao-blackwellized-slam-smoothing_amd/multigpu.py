"""Particle-sharded forward filter over torch.distributed (RCCL on MI355X, gloo for the CPU tests).

One process per GPU.  Logical slot i of the global filter (N = world * N_local) lives on rank
i // N_local.  Per time step:

  1. all_gather of the small forward bank: log-weights [N_local] and non-linear states [nN x N_local]
  2. every rank normalises the GLOBAL weights and draws the GLOBAL ancestor vector with the same
     kernels as the single-GPU path (identical on every rank, so no index exchange is needed and the
     result equals the single-GPU run with N particles bit for bit)
  3. all_to_all of the map state (covariance blocks, pending factors, mean) of the UNIQUE remote
     ancestors only -- this is the one heavy message of the algorithm (reference: the gather
     `xl = xl(:,ai); P = P(:,:,ai)`, src/particleFilter.m:112-113, when ai(i) lives on another GPU)
  4. the fused step kernel; remote ancestors are read from the receive region behind the local bank.

The exchange plan is pure numpy on the (replicated) ancestor vector, so it is unit-tested on CPU with
gloo (tests/test_multigpu_plan.py).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi
from ._ffi import check, load_library
from .host import PhiloxRNG, _Problem, _dp, _ip, _rng_block


# ------------------------------------------------------------------------------------------------
# exchange plan (host logic, identical on every rank)
# ------------------------------------------------------------------------------------------------
@dataclass
class ExchangePlan:
    send_idx: np.ndarray      # int32 local indices to pack, ordered by destination rank then index
    send_counts: np.ndarray   # int64 [world] particles sent to each rank
    recv_counts: np.ndarray   # int64 [world] particles received from each rank
    anc_bank: np.ndarray      # int32 [N_local] ancestor index in [local bank | recv region]


def build_plan(ai_global: np.ndarray, rank: int, world: int, n_local: int) -> ExchangePlan:
    """ai_global: ancestor (0-based global slot id) of every global slot, identical on all ranks.

    Receive region layout on rank g: for source ranks r = 0..world-1 (r != g) in order, the unique
    ancestors living on r that g's slots need, ascending.  The sender derives the same lists from the
    same vector, so both sides agree without any handshake."""
    ai_global = np.asarray(ai_global, dtype=np.int64)
    if ai_global.size != world * n_local:
        raise ValueError("ai_global must have world * n_local entries")
    owner = ai_global // n_local
    dest = np.arange(ai_global.size, dtype=np.int64) // n_local
    # --- what I receive: unique remote ancestors of my slots (np.unique sorts by owner, then index)
    mine = ai_global[rank * n_local:(rank + 1) * n_local]
    remote = np.unique(mine[mine // n_local != rank])
    recv_counts = np.bincount(remote // n_local, minlength=world).astype(np.int64)
    anc_bank = np.empty(n_local, dtype=np.int32)
    local_mask = (mine // n_local) == rank
    anc_bank[local_mask] = (mine[local_mask] - rank * n_local).astype(np.int32)
    anc_bank[~local_mask] = (n_local + np.searchsorted(remote, mine[~local_mask])).astype(np.int32)
    # --- what I send: for every other rank q, the unique ancestors of q's slots that live on me
    m = (owner == rank) & (dest != rank)
    key = np.unique(dest[m] * (world * n_local) + ai_global[m])        # sorted by destination, then index
    send_dest = key // (world * n_local)
    send_idx = (key % (world * n_local) - rank * n_local).astype(np.int32)
    send_counts = np.bincount(send_dest, minlength=world).astype(np.int64)
    return ExchangePlan(send_idx, send_counts, recv_counts, anc_bank)


def exchange_rows(send, recv, send_counts, recv_counts, dist):
    """all_to_all of whole rows (dim 0) between two torch tensors of shape [rows, width]."""
    width = send.shape[1] if send.dim() > 1 else 1
    ns, nr = int(np.sum(send_counts)), int(np.sum(recv_counts))
    dist.all_to_all_single(recv[:nr].reshape(-1), send[:ns].reshape(-1),
                           output_split_sizes=[int(c) * width for c in recv_counts],
                           input_split_sizes=[int(c) * width for c in send_counts])


# ------------------------------------------------------------------------------------------------
# device-pointer views for torch
# ------------------------------------------------------------------------------------------------
class _DevArray:
    """Zero-copy view of library-owned device memory for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr, shape, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def _view(torch, ptr, shape, device):
    if ptr is None or int(np.prod(shape)) == 0:
        return torch.empty(tuple(int(s) for s in shape), dtype=torch.float64, device=device)
    return torch.as_tensor(_DevArray(C.cast(ptr, C.c_void_p).value, shape), device=device)


class rbpf_shard_views(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("N_local", C.c_int32), ("N_global", C.c_int32),
                ("szT", C.c_size_t), ("szB", C.c_size_t), ("szF", C.c_size_t), ("szX", C.c_size_t),
                ("recv_capacity", C.c_size_t), ("send_capacity", C.c_size_t),
                ("logw_local", _ffi.c_double_p), ("xn_local", _ffi.c_double_p), ("logw_gather", _ffi.c_double_p),
                ("xn_gather", _ffi.c_double_p), ("send_Pt", _ffi.c_double_p), ("send_Pb", _ffi.c_double_p),
                ("send_F", _ffi.c_double_p), ("send_xl", _ffi.c_double_p), ("recv_Pt", _ffi.c_double_p),
                ("recv_Pb", _ffi.c_double_p), ("recv_F", _ffi.c_double_p), ("recv_xl", _ffi.c_double_p)]


class ShardedFilterSession:
    """Same surface as host.FilterSession (advance / sync / timing / finish / close), one rank's share.

    transport="device": collectives on the device buffers (RCCL; the production path).
    transport="host"  : device -> pinned host -> gloo -> device (lets two ranks share ONE GPU in tests)."""

    def __init__(self, model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_local, dt, rng=None, rank=0, world=1,
                 transport="device"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.lib = load_library()
        for name, argt in (("rbpf_shard_create", [C.POINTER(_ffi.rbpf_model), C.POINTER(_ffi.rbpf_problem),
                                                  C.POINTER(_ffi.rbpf_rng), C.POINTER(_ffi.rbpf_options), C.c_int32,
                                                  C.c_int32, C.POINTER(C.c_void_p)]),
                           ("rbpf_shard_views_get", [C.c_void_p, C.POINTER(rbpf_shard_views)]),
                           ("rbpf_shard_normalise_search", [C.c_void_p, _ffi.c_int32_p]),
                           ("rbpf_shard_pack", [C.c_void_p, _ffi.c_int32_p, C.c_int32]),
                           ("rbpf_shard_step", [C.c_void_p, _ffi.c_int32_p]),
                           ("rbpf_shard_trajectories", [C.c_void_p, _ffi.c_double_p, _ffi.c_double_p])):
            getattr(self.lib, name).argtypes = argt
        self.model, self.rank, self.world, self.transport = model, int(rank), int(world), transport
        self.prob = _Problem(model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_local, dt)
        # replay buffers (tests) hold all world*N_local slots; Philox streams are keyed by global slot id
        self.blk, self._rng = _rng_block(rng if rng is not None else PhiloxRNG(1), self.prob.N_P * self.world,
                                         self.prob.N_T, model.nw, 1)
        self.opt = _ffi.rbpf_options(keep_history=0, trace=0, fix_p_mean=0, reserved=0, jitter=0.0)
        self.mdesc = model.descriptor()
        self.ctx = C.c_void_p()
        check(self.lib.rbpf_shard_create(C.byref(self.mdesc), C.byref(self.prob.c), C.byref(self.blk),
                                         C.byref(self.opt), self.rank, self.world, C.byref(self.ctx)))
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.N_local, self.N_global = self.prob.N_P, self.prob.N_P * self.world
        self.t = 0
        self.t_norm = 0
        self.ai = np.empty(self.N_global, dtype=np.int32)
        self.stats = dict(sent_particles=0, recv_particles=0, steps=0)

    # -- helpers ---------------------------------------------------------------------------------
    def _views(self):
        v = rbpf_shard_views()
        check(self.lib.rbpf_shard_views_get(self.ctx, C.byref(v)))
        return v

    def _gather(self, v):
        torch, dist = self.torch, self.dist
        nN, Nl, W = self.model.nNonLin, self.N_local, self.world
        pairs = ((v.logw_local, v.logw_gather, Nl), (v.xn_local, v.xn_gather, nN * Nl))
        for src_p, dst_p, cnt in pairs:
            src = _view(torch, src_p, (cnt,), self.device)
            dst = _view(torch, dst_p, (W * cnt,), self.device)
            if W == 1:
                dst.copy_(src)
            elif self.transport == "device":
                dist.all_gather_into_tensor(dst, src)
            else:
                h = torch.empty(W * cnt, dtype=torch.float64)
                dist.all_gather_into_tensor(h, src.cpu())
                dst.copy_(h)
        torch.cuda.synchronize()

    def _exchange(self, v, plan):
        torch, dist = self.torch, self.dist
        ns, nr = int(plan.send_counts.sum()), int(plan.recv_counts.sum())
        if nr > v.recv_capacity or ns > v.send_capacity:
            raise _ffi.RBPFError(_ffi.RBPF_ERR_OUT_OF_MEMORY, f"exchange of {ns}/{nr} particles exceeds the staging "
                                 f"capacity {v.send_capacity}/{v.recv_capacity}")
        idx = np.ascontiguousarray(plan.send_idx)
        check(self.lib.rbpf_shard_pack(self.ctx, _ip(idx), ns))
        for sp, rp, width in ((v.send_Pt, v.recv_Pt, v.szT), (v.send_Pb, v.recv_Pb, v.szB),
                              (v.send_F, v.recv_F, v.szF), (v.send_xl, v.recv_xl, v.szX)):
            if width == 0:
                continue
            send = _view(torch, sp, (max(ns, 1), width), self.device)
            recv = _view(torch, rp, (max(nr, 1), width), self.device)
            if self.transport == "device":
                exchange_rows(send, recv, plan.send_counts, plan.recv_counts, dist)
            else:
                hs, hr = send[:ns].cpu(), torch.empty((nr, width), dtype=torch.float64)
                exchange_rows(hs, hr, plan.send_counts, plan.recv_counts, dist)
                if nr:
                    recv[:nr].copy_(hr)
        torch.cuda.synchronize()
        self.stats["sent_particles"] += ns
        self.stats["recv_particles"] += nr

    # -- FilterSession surface -------------------------------------------------------------------
    def advance(self, n_steps):
        for _ in range(int(n_steps)):
            if self.t == 0:
                check(self.lib.rbpf_shard_step(self.ctx, None))
            else:
                v = self._views()
                self._gather(v)
                check(self.lib.rbpf_shard_normalise_search(self.ctx, _ip(self.ai)))
                self.t_norm += 1
                plan = build_plan(self.ai, self.rank, self.world, self.N_local)
                if self.world > 1:
                    self._exchange(v, plan)
                anc = np.ascontiguousarray(plan.anc_bank)
                check(self.lib.rbpf_shard_step(self.ctx, _ip(anc)))
            self.t += 1
            self.stats["steps"] += 1

    def sync(self):
        self.torch.cuda.synchronize()

    def reset(self):
        raise NotImplementedError("create a new ShardedFilterSession")

    def timing(self, enable=None, reset=False):
        if enable is not None:
            check(self.lib.rbpf_timing_enable(self.ctx, 1 if enable else 0))
            return None
        tm = _ffi.rbpf_timing()
        check(self.lib.rbpf_timing_read(self.ctx, C.byref(tm), 1 if reset else 0))
        return dict(ms=tm.stream_kernel_ms, launches=tm.stream_kernel_launches,
                    bytes_per_launch=tm.algorithmic_bytes_per_launch)

    def finish(self, want=("traj_max", "traj_mean")):
        """Normalises the last finished step and returns the global trajectory summaries."""
        if self.t_norm < self.t:
            v = self._views()
            self._gather(v)
            check(self.lib.rbpf_shard_normalise_search(self.ctx, None))
            self.t_norm += 1
        nN, T = self.model.nNonLin, self.prob.N_T
        out = dict(traj_max=np.full((nN, T), np.nan, order="F"), traj_mean=np.full((nN, T), np.nan, order="F"))
        check(self.lib.rbpf_shard_trajectories(self.ctx, _dp(out["traj_max"]), _dp(out["traj_mean"])))
        return out

    def close(self):
        if self.ctx:
            self.lib.rbpf_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
