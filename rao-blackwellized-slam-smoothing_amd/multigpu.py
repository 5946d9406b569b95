"""Particle-sharded forward filter over torch.distributed (RCCL on MI355X, gloo for the CPU tests).

One process per GPU.  The global filter has N = world * N_local *logical* slots; a logical slot keeps its
identity (RNG stream, position in the outputs) but its particle may live on any rank.  Per time step:

  1. ONE all_gather of the small forward bank (log-weights + non-linear states, (nN+1) x N_local doubles)
  2. every rank normalises the GLOBAL weights and draws the GLOBAL ancestor vector with the same kernels
     as the single-GPU path -- identical on every rank, so no index exchange is needed and the result
     equals the single-GPU run with N particles bit for bit
  3. placement of the new generation ("owner computes"): a child is computed on the rank that already
     holds its ancestor's map state; only the load-imbalance excess migrates, and every rank ends up with
     exactly N_local particles.  This is the distributed form of the reference's gather
     `xl = xl(:,ai); P = P(:,:,ai)` (src/particleFilter.m:112-113): with static slot ownership ~(W-1)/W of
     the covariances would cross xGMI every step; with owner-computes it is the binomial imbalance
     (~0.4 % of the particles at W = 8 under near-uniform weights)
  4. ONE all_to_all of whole particle records for the migrating ancestors
  5. the fused step kernel on the N_local physical slots (placed in ancestor order, so sibling reads hit
     the Infinity Cache exactly as in the single-GPU schedule).

The placement / exchange plan is pure numpy on the replicated ancestor vector and is unit-tested on CPU
with gloo (tests/test_multigpu_plan.py).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi
from ._ffi import check, load_library
from .host import PhiloxRNG, _Problem, _dp, _ip, _rng_block, _storage_code


# ------------------------------------------------------------------------------------------------
# placement + exchange plan (host logic, identical on every rank)
# ------------------------------------------------------------------------------------------------
@dataclass
class GenerationPlan:
    new_rank: np.ndarray      # int64 [N] rank that computes / holds logical slot i in the new generation
    new_idx: np.ndarray       # int64 [N] its physical slot there
    pair_dest: np.ndarray     # int64 [P] exchange pairs (destination rank, ...
    pair_src: np.ndarray      # int64 [P]  ... global physical id of the ancestor), unique, sorted (dest, src)
    migrated: int             # children computed away from their ancestor's rank


def _stable_argsort_small(keys, n_values):
    """Stable argsort of non-negative integer keys < n_values.  numpy's stable sort is an O(N) radix sort for
    16-bit keys; wider keys are split into two 16-bit passes (LSD radix)."""
    if n_values <= 65536:
        return np.argsort(keys.astype(np.uint16), kind="stable")
    lo = np.argsort((keys & 0xFFFF).astype(np.uint16), kind="stable")
    hi = np.argsort((keys[lo] >> 16).astype(np.uint16), kind="stable")
    return lo[hi]


def plan_generation(ai, cur_rank, cur_idx, world: int, n_local: int) -> GenerationPlan:
    """ai[i]: ancestor (logical id) of logical slot i; cur_rank/cur_idx: where every logical slot's current
    particle lives.  Deterministic, so every rank derives the same plan from the same vector."""
    ai = np.asarray(ai, dtype=np.int64)
    N = world * n_local
    if ai.size != N:
        raise ValueError("ai must have world * n_local entries")
    anc_gid = (np.asarray(cur_rank, dtype=np.int64) * n_local + np.asarray(cur_idx, dtype=np.int64))[ai]
    anc_rank = anc_gid // n_local
    load = np.bincount(anc_rank, minlength=world)
    excess = np.maximum(load - n_local, 0)
    deficit = np.maximum(n_local - load, 0)
    stay_cnt = load - excess
    # children grouped by their ancestor's rank, inside a group by ancestor slot (siblings adjacent), ties by
    # logical id: a stable sort on the ancestor's global physical id
    order = _stable_argsort_small(anc_gid, N)
    grp = anc_rank[order]
    start = np.concatenate(([0], np.cumsum(load)))[:-1]
    pos = np.arange(N) - start[grp]
    keep = pos < stay_cnt[grp]
    new_rank = anc_rank.copy()
    new_idx = np.empty(N, dtype=np.int64)
    new_idx[order[keep]] = pos[keep]                     # stayers: position among the kept children of their rank
    moved = order[~keep]                                 # the last `excess` children of every overloaded rank
    receivers = np.repeat(np.arange(world), deficit)     # ranks with room, in rank order
    new_rank[moved] = receivers
    imp_start = np.concatenate(([0], np.cumsum(deficit)))[:-1]
    new_idx[moved] = stay_cnt[receivers] + (np.arange(moved.size) - imp_start[receivers])
    # one record per (destination, ancestor) pair; the moved list is ordered by source id and the receivers are
    # non-decreasing, so equal pairs are adjacent
    key = receivers * N + anc_gid[moved]
    if key.size:
        key = key[np.concatenate(([True], key[1:] != key[:-1]))]
        key = np.sort(key)
    return GenerationPlan(new_rank, new_idx, key // N, key % N, int(moved.size))


@dataclass
class RankPlan:
    slot_ids: np.ndarray      # int32 [n_local] logical id of each physical slot of the new generation
    anc_bank: np.ndarray      # int32 [n_local] ancestor location: < n_local local slot, else n_local + record index
    send_idx: np.ndarray      # int32 local physical indices to pack, ordered by destination rank then index
    send_counts: np.ndarray   # int64 [world]
    recv_counts: np.ndarray   # int64 [world]


def rank_view(plan: GenerationPlan, ai, cur_rank, cur_idx, rank: int, world: int, n_local: int) -> RankPlan:
    ai = np.asarray(ai, dtype=np.int64)
    N = world * n_local
    mine = np.nonzero(plan.new_rank == rank)[0]
    slot_ids = np.empty(n_local, dtype=np.int64)
    slot_ids[plan.new_idx[mine]] = mine
    a = ai[slot_ids]
    a_rank = np.asarray(cur_rank, dtype=np.int64)[a]
    a_idx = np.asarray(cur_idx, dtype=np.int64)[a]
    anc_bank = a_idx.copy()
    # what I receive: pairs with dest == rank, already sorted by (source rank, source slot)
    rsel = plan.pair_dest == rank
    recv_src = plan.pair_src[rsel]
    recv_counts = np.bincount(recv_src // n_local, minlength=world).astype(np.int64)
    rem = a_rank != rank
    if rem.any():
        anc_bank[rem] = n_local + np.searchsorted(recv_src, a_rank[rem] * n_local + a_idx[rem])
    # what I send: pairs whose ancestor lives on me, ordered by (destination, slot)
    ssel = (plan.pair_src // n_local) == rank
    sd, ss = plan.pair_dest[ssel], plan.pair_src[ssel] - rank * n_local
    o = np.lexsort((ss, sd))
    send_counts = np.bincount(sd, minlength=world).astype(np.int64)
    return RankPlan(slot_ids.astype(np.int32), anc_bank.astype(np.int32), ss[o].astype(np.int32), send_counts,
                    recv_counts)


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
    addr = C.cast(ptr, C.c_void_p).value
    if not addr or int(np.prod(shape)) == 0:
        return torch.empty(tuple(int(s) for s in shape), dtype=torch.float64, device=device)
    return torch.as_tensor(_DevArray(addr, shape), device=device)


class rbpf_shard_views(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("N_local", C.c_int32), ("N_global", C.c_int32),
                ("n_nonlin", C.c_int32), ("record_doubles", C.c_size_t), ("recv_capacity", C.c_size_t),
                ("send_capacity", C.c_size_t), ("fwd_local", _ffi.c_double_p), ("fwd_gather", _ffi.c_double_p),
                ("send_rec", _ffi.c_double_p), ("recv_rec", _ffi.c_double_p), ("fwd_rows", C.c_int32)]


class ShardedFilterSession:
    """Same surface as host.FilterSession (advance / sync / timing / finish / close), one rank's share.

    transport="device": collectives on the device buffers (RCCL; the production path).
    transport="host"  : device -> host -> gloo -> device (lets two ranks share ONE GPU in tests)."""

    def __init__(self, model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_local, dt, rng=None, rank=0, world=1,
                 transport="device", planner="device", lazy_depth=0, storage="fp64", keep_history=False, exchange_capacity=0,
                 sync_phases=False, force_collectives=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.sync_phases = bool(sync_phases)   # diagnostic: synchronise after every phase so that stats["phase_s"] is GPU time
        # world 1 normally short-cuts the collectives (a device copy, no exchange); force_collectives issues the real calls --
        # all_gather_into_tensor / all_to_all_single on the library's buffers and stream -- so that a one-GPU box exercises them
        self.force_collectives = bool(force_collectives)
        self.lib = load_library()
        for name, argt in (("rbpf_shard_create", [C.POINTER(_ffi.rbpf_model), C.POINTER(_ffi.rbpf_problem),
                                                  C.POINTER(_ffi.rbpf_rng), C.POINTER(_ffi.rbpf_options), C.c_int32,
                                                  C.c_int32, C.POINTER(C.c_void_p)]),
                           ("rbpf_shard_views_get", [C.c_void_p, C.POINTER(rbpf_shard_views)]),
                           ("rbpf_shard_normalise_search", [C.c_void_p, _ffi.c_int32_p, _ffi.c_int32_p]),
                           ("rbpf_shard_pack", [C.c_void_p, _ffi.c_int32_p, C.c_int32]),
                           ("rbpf_shard_step", [C.c_void_p, _ffi.c_int32_p, _ffi.c_int32_p]),
                           ("rbpf_shard_plan", [C.c_void_p, C.POINTER(C.c_int64)]),
                           ("rbpf_shard_normalise_plan", [C.c_void_p, C.POINTER(C.c_int64)]),
                           ("rbpf_shard_plan_read", [C.c_void_p, _ffi.c_int32_p, _ffi.c_int32_p, _ffi.c_int32_p, C.c_int32,
                                                     _ffi.c_int32_p]),
                           ("rbpf_shard_trajectories", [C.c_void_p, _ffi.c_double_p, _ffi.c_double_p]),
                           ("rbpf_stream_get", [C.c_void_p, C.POINTER(C.c_void_p)]),
                           ("rbpf_shard_set_async", [C.c_void_p, C.c_int32]),
                           ("rbpf_shard_finish", [C.c_void_p, C.c_int32, _ffi.c_double_p, _ffi.c_double_p, _ffi.c_double_p,
                                                  _ffi.c_double_p, _ffi.c_double_p, _ffi.c_int32_p]),
                           ("rbpf_shard_set_ancestors", [C.c_void_p, _ffi.c_int32_p])):
            getattr(self.lib, name).argtypes = argt
        self.model, self.rank, self.world, self.transport = model, int(rank), int(world), transport
        self.planner = planner          # "device": rbpf_shard_plan (production); "host": the numpy specification
        self.prob = _Problem(model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_local, dt)
        # replay buffers (tests) hold all world*N_local slots; Philox streams are keyed by logical slot id
        self.blk, self._rng = _rng_block(rng if rng is not None else PhiloxRNG(1), self.prob.N_P * self.world,
                                         self.prob.N_T, model.nw, self._n_iter())
        if lazy_depth >= 2 and planner != "device":
            raise ValueError("lazy_depth >= 2 needs planner='device'")
        self.opt = _ffi.rbpf_options(keep_history=1 if keep_history else 0, trace=0, fix_p_mean=0, lazy_depth=int(lazy_depth),
                                     jitter=0.0, storage=_storage_code(storage), exchange_capacity=int(exchange_capacity),
                                     chol_refresh=int(getattr(self, "chol_refresh_requested", 0)))
        self.mdesc = model.descriptor()
        self.ctx = C.c_void_p()
        self._create()
        self.device = torch.device("cuda", torch.cuda.current_device())
        # Collectives are issued on the library's own stream: kernels and collectives are then ordered on the device and a
        # step needs ONE host synchronisation (the split sizes of the all_to_all) instead of one per phase.
        sp = C.c_void_p()
        check(self.lib.rbpf_stream_get(self.ctx, C.byref(sp)))
        self.stream = torch.cuda.ExternalStream(sp.value, device=self.device)
        check(self.lib.rbpf_shard_set_async(self.ctx, 1 if transport == "device" else 0))
        self.N_local, self.N_global = self.prob.N_P, self.prob.N_P * self.world
        self.t = 0
        self.t_norm = 0
        self.ai = np.empty(self.N_global, dtype=np.int32)
        gid = np.arange(self.N_global, dtype=np.int64)
        self.cur_rank, self.cur_idx = gid // self.N_local, gid % self.N_local
        self.identity = True
        self.stats = dict(migrated=0, sent_records=0, recv_records=0, steps=0)
        v = rbpf_shard_views()
        check(self.lib.rbpf_shard_views_get(self.ctx, C.byref(v)))
        self.v = v
        fwd = int(v.fwd_rows) * self.N_local          # states, log-weights (+ the smoother's ancestor-weight row)
        self.t_fwd_local = _view(torch, v.fwd_local, (fwd,), self.device)
        self.t_fwd_gather = _view(torch, v.fwd_gather, (self.world * fwd,), self.device)
        coll = self.world > 1 or self.force_collectives
        self.t_send = _view(torch, v.send_rec if v.send_capacity else None, (max(int(v.send_capacity), 1), int(v.record_doubles)),
                            self.device) if coll else None
        self.t_recv = _view(torch, v.recv_rec if v.recv_capacity else None, (max(int(v.recv_capacity), 1), int(v.record_doubles)),
                            self.device) if coll else None

    def _n_iter(self):
        return 1

    def _refresh_views(self):
        """rbpf_shard_plan may have grown the record buffers (rbpf_options.exchange_capacity <= 0): new pointers / capacities."""
        v = rbpf_shard_views()
        check(self.lib.rbpf_shard_views_get(self.ctx, C.byref(v)))
        same = (C.cast(v.send_rec, C.c_void_p).value == C.cast(self.v.send_rec, C.c_void_p).value and
                C.cast(v.recv_rec, C.c_void_p).value == C.cast(self.v.recv_rec, C.c_void_p).value and
                int(v.recv_capacity) == int(self.v.recv_capacity))
        if same:
            return
        self.v = v
        self.stats["regrown"] = self.stats.get("regrown", 0) + 1
        if self.world > 1 or self.force_collectives:
            torch = self.torch
            self.t_send = _view(torch, v.send_rec if v.send_capacity else None, (max(int(v.send_capacity), 1), int(v.record_doubles)), self.device)
            self.t_recv = _view(torch, v.recv_rec if v.recv_capacity else None, (max(int(v.recv_capacity), 1), int(v.record_doubles)), self.device)

    def _create(self):
        check(self.lib.rbpf_shard_create(C.byref(self.mdesc), C.byref(self.prob.c), C.byref(self.blk),
                                         C.byref(self.opt), self.rank, self.world, C.byref(self.ctx)))

    # -- collectives -----------------------------------------------------------------------------
    def _phase_sync(self):
        if self.sync_phases or self.transport != "device":
            self.stream.synchronize()

    def _gather(self):
        torch, dist = self.torch, self.dist
        with torch.cuda.stream(self.stream):
            if self.world == 1 and not self.force_collectives:
                self.t_fwd_gather.copy_(self.t_fwd_local)
            elif self.transport == "device":
                dist.all_gather_into_tensor(self.t_fwd_gather, self.t_fwd_local)
            else:
                h = torch.empty(self.t_fwd_gather.shape, dtype=torch.float64)
                dist.all_gather_into_tensor(h, self.t_fwd_local.cpu())
                self.t_fwd_gather.copy_(h)
        self._phase_sync()

    def _exchange(self, rp, recv_off=0):
        """rp: RankPlan (host planner) or (send_counts, recv_counts) of the device plan.  recv_off: first record of
        the receive buffer this exchange writes (records persist during a lazy cycle).  Whether the exchange fits the
        buffers was decided -- identically on every rank -- by rbpf_shard_plan; the host planner repeats that check here
        for every rank from the replicated plan."""
        torch, dist = self.torch, self.dist
        send_counts, recv_counts = (rp.send_counts, rp.recv_counts) if isinstance(rp, RankPlan) else rp
        if not isinstance(rp, RankPlan):
            self._refresh_views()
        ns, nr = int(send_counts.sum()), int(recv_counts.sum())
        if recv_off + nr > self.v.recv_capacity or ns > self.v.send_capacity:
            raise _ffi.RBPFError(_ffi.RBPF_ERR_OUT_OF_MEMORY, f"exchange of {ns}/{nr} records exceeds the buffer "
                                 f"capacity {self.v.send_capacity}/{self.v.recv_capacity}")
        if isinstance(rp, RankPlan):
            idx = np.ascontiguousarray(rp.send_idx)
            check(self.lib.rbpf_shard_pack(self.ctx, _ip(idx), ns))
        else:
            check(self.lib.rbpf_shard_pack(self.ctx, None, ns))
        rp = RankPlan(None, None, None, send_counts, recv_counts)
        t_recv = self.t_recv[recv_off:]
        with torch.cuda.stream(self.stream):
            if self.transport == "device":
                exchange_rows(self.t_send, t_recv, rp.send_counts, rp.recv_counts, dist)
            else:
                width = int(self.v.record_doubles)
                hs, hr = self.t_send[:ns].cpu(), torch.empty((nr, width), dtype=torch.float64)
                exchange_rows(hs, hr, rp.send_counts, rp.recv_counts, dist)
                if nr:
                    t_recv[:nr].copy_(hr)
        self._phase_sync()
        self.stats["sent_records"] += ns
        self.stats["recv_records"] += nr

    def _normalise(self, want_ancestors):
        perm = None
        if self.planner == "device":
            # the library keeps the placement on the device; ancestors stay there too (self.ai is only a D2H
            # target the C side requires to trigger the draw)
            check(self.lib.rbpf_shard_normalise_search(self.ctx, None, _ip(self.ai) if want_ancestors else None))
            self.t_norm += 1
            return
        if not self.identity:
            perm = np.ascontiguousarray((self.cur_rank * self.N_local + self.cur_idx).astype(np.int32))
        check(self.lib.rbpf_shard_normalise_search(self.ctx, None if perm is None else _ip(perm),
                                                   _ip(self.ai) if want_ancestors else None))
        self.t_norm += 1

    # -- FilterSession surface -------------------------------------------------------------------
    def advance(self, n_steps):
        import time
        tm = self.stats.setdefault("phase_s", dict(gather=0.0, normalise=0.0, plan=0.0, exchange=0.0, step=0.0))
        for _ in range(int(n_steps)):
            if self.t == 0:
                check(self.lib.rbpf_shard_step(self.ctx, None, None))
            else:
                t0 = time.perf_counter()
                self._gather()
                t1 = time.perf_counter()
                if self.planner == "device":
                    # normalise + draw + plan in one library call: ancestors stay on the device, one stream sync
                    cnt = np.zeros(2 * self.world + 2, dtype=np.int64)
                    check(self.lib.rbpf_shard_normalise_plan(self.ctx, cnt.ctypes.data_as(C.POINTER(C.c_int64))))
                    self.t_norm += 1
                    t2 = t3 = time.perf_counter()
                    if self.world > 1 or self.force_collectives:
                        self._exchange((cnt[:self.world], cnt[self.world:2 * self.world]), int(cnt[2 * self.world + 1]))
                    t4 = time.perf_counter()
                    check(self.lib.rbpf_shard_step(self.ctx, None, None))
                    if self.sync_phases:
                        self.stream.synchronize()
                    t5 = time.perf_counter()
                    self.stats["migrated"] += int(cnt[2 * self.world])
                    tm["gather"] += t1 - t0; tm["normalise"] += t2 - t1; tm["plan"] += t3 - t2
                    tm["exchange"] += t4 - t3; tm["step"] += t5 - t4
                    self.t += 1
                    self.stats["steps"] += 1
                    continue
                self._normalise(True)
                t2 = time.perf_counter()
                plan = plan_generation(self.ai, self.cur_rank, self.cur_idx, self.world, self.N_local)
                rp = rank_view(plan, self.ai, self.cur_rank, self.cur_idx, self.rank, self.world, self.N_local)
                if self.world > 1 and plan.pair_dest.size:        # every rank checks every rank (the plan is replicated)
                    rcv = np.bincount(plan.pair_dest, minlength=self.world)
                    snd = np.bincount(plan.pair_src // self.N_local, minlength=self.world)
                    if rcv.max() > self.v.recv_capacity or snd.max() > self.v.send_capacity:
                        raise _ffi.RBPFError(_ffi.RBPF_ERR_OUT_OF_MEMORY, "exchange does not fit the record buffers of rank "
                                             f"{int(np.argmax(np.maximum(rcv, snd)))}: raise exchange_capacity")
                t3 = time.perf_counter()
                if self.world > 1:
                    self._exchange(rp)
                t4 = time.perf_counter()
                check(self.lib.rbpf_shard_step(self.ctx, _ip(rp.anc_bank), _ip(rp.slot_ids)))
                if self.sync_phases:
                    self.stream.synchronize()
                t5 = time.perf_counter()
                tm["gather"] += t1 - t0; tm["normalise"] += t2 - t1; tm["plan"] += t3 - t2
                tm["exchange"] += t4 - t3; tm["step"] += t5 - t4
                self.cur_rank, self.cur_idx = plan.new_rank, plan.new_idx
                self.identity = False
                self.stats["migrated"] += plan.migrated
            self.t += 1
            self.stats["steps"] += 1

    def sync(self):
        self.stream.synchronize()
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
                    bytes_per_launch=tm.algorithmic_bytes_per_launch,
                    scheduled_bytes_per_launch=tm.scheduled_bytes_per_launch)

    def finish(self, want=("traj_max", "traj_mean")):
        """Normalises the last finished step and returns the requested outputs of src/particleFilter.m:220-233 for the GLOBAL
        filter, identical on every rank: traj_max, traj_mean, xl_max, P_max, xl_mean, P_mean (quirk Q3, as the reference),
        traj_sample_iwmax (needs keep_history=True), iw_max.  xl_max / P_max / P_mean come from the rank that holds the particle
        (one small all_reduce); xl_mean is the sum of the ranks' shares."""
        if self.t_norm < self.t:
            self._gather()
            self._normalise(False)
        self.sync()
        nN, n, T = self.model.nNonLin, self.model.nLin, self.prob.N_T
        out = dict(traj_max=np.full((nN, T), np.nan, order="F"), traj_mean=np.full((nN, T), np.nan, order="F"))
        check(self.lib.rbpf_shard_trajectories(self.ctx, _dp(out["traj_max"]), _dp(out["traj_mean"])))
        extra = [k for k in want if k not in ("traj_max", "traj_mean")]
        if extra:
            need_mean = "xl_mean" in want or "P_mean" in want
            xl_max, P_max = np.zeros(n), np.zeros((n, n), order="F")
            xl_mean = np.zeros(n)
            tsi = np.full((nN, self.t), np.nan, order="F") if "traj_sample_iwmax" in want else None
            iw = C.c_int32(0)
            check(self.lib.rbpf_shard_finish(self.ctx, 0, _dp(xl_max) if "xl_max" in want else None,
                                             _dp(P_max) if "P_max" in want else None, _dp(xl_mean) if need_mean else None, None,
                                             _dp(tsi) if tsi is not None else None, C.byref(iw)))
            buf = self._all_reduce_host(np.concatenate((xl_max, P_max.ravel(order="F"), xl_mean)))
            xl_max, P_max, xl_mean = buf[:n], buf[n:n + n * n].reshape((n, n), order="F"), buf[n + n * n:]
            out.update(xl_max=xl_max, P_max=P_max, xl_mean=xl_mean, iw_max=int(iw.value))
            if tsi is not None:
                out["traj_sample_iwmax"] = tsi
            if "P_mean" in want:
                P_mean = np.zeros((n, n), order="F")
                xm = np.ascontiguousarray(xl_mean)
                check(self.lib.rbpf_shard_finish(self.ctx, 1, None, None, _dp(xm), _dp(P_mean), None, None))
                out["P_mean"] = self._all_reduce_host(P_mean.ravel(order="F")).reshape((n, n), order="F")
        return {k: v for k, v in out.items() if k in want or k == "iw_max"}

    def _all_reduce_host(self, vec):
        """Sum over ranks of a host vector (final extraction only: a few n^2 doubles, once per run)."""
        if self.world == 1:
            return np.array(vec, dtype=np.float64)
        t = self.torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64))
        if self.transport == "device":
            t = t.to(self.device)
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def close(self):
        if self.ctx:
            self.sync()
            self.t_fwd_local = self.t_fwd_gather = self.t_send = self.t_recv = None
            self.lib.rbpf_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ------------------------------------------------------------------------------------------------
# particle-sharded information-form smoother
# ------------------------------------------------------------------------------------------------
class rbpf_shard_smoother_views(C.Structure):
    _fields_ = [("anc_local", _ffi.c_double_p), ("anc_gather", _ffi.c_double_p),
                ("refresh_send", _ffi.c_double_p), ("refresh_recv", _ffi.c_double_p),
                ("refresh_capacity", C.c_int64), ("matrix_doubles", C.c_int64)]


class RefreshPlan:
    """Fetch plan of one refresh of the carried factors, for one rank (see plan_refresh)."""
    __slots__ = ("send_slots", "send_counts", "recv_counts", "base_index", "send_totals", "recv_totals")

    def __init__(self, send_slots, send_counts, recv_counts, base_index, send_totals, recv_totals):
        self.send_slots, self.send_counts, self.recv_counts = send_slots, send_counts, recv_counts
        self.base_index, self.send_totals, self.recv_totals = base_index, send_totals, recv_totals


def plan_refresh_lib(lib, owner_now, base_loc, n_local, world, rank):
    """The fetch plan of one refresh from the library's planner (rbpf_plan_refresh, csrc/rbpf_multi.hip) -- the implementation
    both drivers run (this one under torchrun, the in-library multi-device driver behind rbpf_options.n_devices).  The numpy
    function below is its specification: tests/test_multigpu_plan.py holds the two equal."""
    owner_now = np.ascontiguousarray(owner_now, dtype=np.int32)
    base_loc = np.ascontiguousarray(base_loc, dtype=np.int32)
    N = owner_now.size
    send_slots = np.zeros(max(N, 1), dtype=np.int32)
    n_send = C.c_int32(0)
    sc, rc, st, rt = (np.zeros(world, dtype=np.int64) for _ in range(4))
    base_index = np.zeros(n_local, dtype=np.int32)
    lp = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))                   # noqa: E731
    lib.rbpf_plan_refresh.argtypes = [_ffi.c_int32_p, _ffi.c_int32_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _ffi.c_int32_p, C.c_int32,
                                      _ffi.c_int32_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64), _ffi.c_int32_p]
    check(lib.rbpf_plan_refresh(_ip(owner_now), _ip(base_loc), int(N), int(n_local), int(world), int(rank), _ip(send_slots), int(send_slots.size),
                                C.byref(n_send), lp(sc), lp(rc), lp(st), lp(rt), _ip(base_index)))
    return RefreshPlan(send_slots[:n_send.value].copy(), sc, rc, base_index, st, rt)


def plan_refresh(owner_now, base_loc, n_local, world, rank):
    """Which base matrices cross ranks at a refresh of the carried factors (rbpf_shard_smoother_refresh_*).  SPECIFICATION of
    rbpf_plan_refresh (the sessions call the library's planner, plan_refresh_lib).

    owner_now[j] = r * n_local + p: rank and physical slot of logical slot j's particle; base_loc[j] = q * n_local + s: rank
    and bank slot of the matrix its information matrix is rebuilt from.  Both tables are replicated, so every rank derives the
    same plan without communicating: rank q sends rank r each matrix some particle on r needs, once (siblings share it), in
    ascending slot order; a receive buffer therefore holds the matrices of rank 0, 1, ... in that order.  Returns the
    RefreshPlan of `rank`: send_slots (concatenated per destination), send_counts / recv_counts [world], base_index [n_local]
    (a slot of the own bank, or n_local + position in the receive buffer), and every rank's totals (capacity check)."""
    owner_now = np.asarray(owner_now, dtype=np.int64)
    base_loc = np.asarray(base_loc, dtype=np.int64)
    r, p = owner_now // n_local, owner_now % n_local
    q, s = base_loc // n_local, base_loc % n_local
    remote = r != q
    key = (r * world + q) * n_local + s                         # sorts by destination, source, slot
    uniq = np.unique(key[remote])
    ur, uq, us = uniq // (world * n_local), (uniq // n_local) % world, uniq % n_local
    mine_out = uq == rank                                       # what I send, ordered by destination then slot
    send_slots = us[mine_out].astype(np.int32)
    send_counts = np.bincount(ur[mine_out], minlength=world).astype(np.int64)
    mine_in = ur == rank                                        # what I receive, ordered by source then slot
    recv_keys = uniq[mine_in]
    recv_counts = np.bincount(uq[mine_in], minlength=world).astype(np.int64)
    base_index = np.zeros(n_local, dtype=np.int32)
    here = r == rank
    loc = here & ~remote
    base_index[p[loc]] = s[loc]
    imp = here & remote
    base_index[p[imp]] = n_local + np.searchsorted(recv_keys, key[imp])
    return RefreshPlan(send_slots, send_counts, recv_counts, base_index,
                       np.bincount(uq, minlength=world).astype(np.int64), np.bincount(ur, minlength=world).astype(np.int64))


class ShardedSmootherSession(ShardedFilterSession):
    """src/particleSmootherInformationForm.m with the N = world * N_local particles of every CPF-AS iteration sharded
    over the ranks (one process per GPU).  On top of the filter's collectives every step of an iteration k > 1 adds
    one all_gather of N ancestor log-weights (:205-240 are evaluated where each particle lives); the information
    state (ivec, Imat, halfLogDetP) migrates inside the particle records.  run() returns the reference's outputs
    (XNK [nN x T x N_K], XLK [n x N_K], PK [n x n x N_K]) on every rank, equal to the single-GPU smoother with N
    particles bit for bit."""

    def __init__(self, model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_local, N_K, dt, rng=None, rank=0, world=1,
                 transport="device", exchange_capacity=0, sync_phases=False, lazy_depth=0, chol_refresh=0, force_collectives=False,
                 storage="fp64"):
        self.N_K = int(N_K)
        self.chol_refresh_requested = int(chol_refresh)
        lib = load_library()
        # the K in use (0 = automatic): the step loop below issues the refresh exchange at the steps the library refreshes at
        self.chol_refresh = int(lib.rbpf_chol_refresh_resolve(int(model.descriptor().kind), int(model.nLin), int(model.ny), int(chol_refresh)))
        for name, argt in (("rbpf_shard_smoother_create", [C.POINTER(_ffi.rbpf_model), C.POINTER(_ffi.rbpf_problem),
                                                           C.POINTER(_ffi.rbpf_rng), C.POINTER(_ffi.rbpf_options), C.c_int32,
                                                           C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
                           ("rbpf_shard_smoother_views_get", [C.c_void_p, C.POINTER(rbpf_shard_smoother_views)]),
                           ("rbpf_shard_smoother_begin", [C.c_void_p, C.c_int32]),
                           ("rbpf_shard_smoother_normalise", [C.c_void_p, C.c_int32]),
                           ("rbpf_shard_smoother_anc_weights", [C.c_void_p]),
                           ("rbpf_shard_smoother_anc_sample", [C.c_void_p, C.c_int32]),
                           ("rbpf_shard_smoother_refresh_begin", [C.c_void_p, _ffi.c_int32_p, _ffi.c_int32_p]),
                           ("rbpf_shard_smoother_refresh_pack", [C.c_void_p, _ffi.c_int32_p, C.c_int32]),
                           ("rbpf_shard_smoother_refresh_end", [C.c_void_p, _ffi.c_int32_p, C.c_int32]),
                           ("rbpf_shard_smoother_step", [C.c_void_p]),
                           ("rbpf_shard_smoother_end", [C.c_void_p, _ffi.c_double_p, _ffi.c_double_p, _ffi.c_double_p,
                                                        _ffi.c_int32_p, _ffi.c_int32_p])):
            getattr(lib, name).argtypes = argt
        super().__init__(model, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_local, dt, rng=rng, rank=rank, world=world,
                         transport=transport, planner="device", lazy_depth=lazy_depth, exchange_capacity=exchange_capacity,
                         sync_phases=sync_phases, force_collectives=force_collectives, storage=storage)
        self.lib.rbpf_shard_smoother_refresh_reserve.argtypes = [C.c_void_p, C.c_int64]
        self._smoother_views()
        self.stats["refreshes"] = 0
        self.stats["refresh_fetched"] = 0

    def _smoother_views(self):
        """(Re-)read the smoother's device buffers: the refresh buffers move when they grow (rbpf_shard_smoother_refresh_reserve)."""
        sv = rbpf_shard_smoother_views()
        check(self.lib.rbpf_shard_smoother_views_get(self.ctx, C.byref(sv)))
        self.t_anc_local = _view(self.torch, sv.anc_local, (self.N_local,), self.device)
        self.t_anc_gather = _view(self.torch, sv.anc_gather, (self.N_global,), self.device)
        self.refresh_capacity = int(sv.refresh_capacity)
        self.t_rf_send = self.t_rf_recv = None
        if self.chol_refresh > 1:
            shape = (self.refresh_capacity, int(sv.matrix_doubles))
            self.t_rf_send = _view(self.torch, sv.refresh_send, shape, self.device)
            self.t_rf_recv = _view(self.torch, sv.refresh_recv, shape, self.device)

    def _n_iter(self):
        return self.N_K

    def _create(self):
        check(self.lib.rbpf_shard_smoother_create(C.byref(self.mdesc), C.byref(self.prob.c), C.byref(self.blk),
                                                  C.byref(self.opt), self.N_K, self.rank, self.world, C.byref(self.ctx)))

    def _gather_anc(self):
        torch, dist = self.torch, self.dist
        with torch.cuda.stream(self.stream):
            if self.world == 1 and not self.force_collectives:
                self.t_anc_gather.copy_(self.t_anc_local)
            elif self.transport == "device":
                dist.all_gather_into_tensor(self.t_anc_gather, self.t_anc_local)
            else:
                h = torch.empty(self.t_anc_gather.shape, dtype=torch.float64)
                dist.all_gather_into_tensor(h, self.t_anc_local.cpu())
                self.t_anc_gather.copy_(h)
        self._phase_sync()

    def advance(self, n_steps):
        raise NotImplementedError("use run()")

    def _refresh(self):
        """Refresh of the carried factors in place of rbpf_shard_smoother_anc_weights: the information matrices of my
        particles are rebuilt from the last materialised generation, whose matrices may sit on other ranks (one all_to_all of
        the unique ones; include/rbpf.h)."""
        torch, dist, lib, W = self.torch, self.dist, self.lib, self.world
        own = np.empty(self.N_global, dtype=np.int32)
        bl = np.empty(self.N_global, dtype=np.int32)
        check(lib.rbpf_shard_smoother_refresh_begin(self.ctx, _ip(own), _ip(bl)))
        self.stats["refreshes"] += 1
        if bl[0] < 0:                                    # first refresh of an iteration: the common initial matrix
            check(lib.rbpf_shard_smoother_refresh_end(self.ctx, None, 0))
            return
        rp = plan_refresh_lib(lib, own, bl, self.N_local, W, self.rank)
        worst = int(max(rp.send_totals.max(), rp.recv_totals.max()))
        if worst > self.refresh_capacity:                # replicated plan: every rank reaches this verdict and grows alike
            check(lib.rbpf_shard_smoother_refresh_reserve(self.ctx, worst))      # (exchange_capacity > 0: a hard limit, fails on every rank)
            self._smoother_views()
        ns, nr = int(rp.send_counts.sum()), int(rp.recv_counts.sum())
        slots = np.ascontiguousarray(rp.send_slots)
        check(lib.rbpf_shard_smoother_refresh_pack(self.ctx, _ip(slots) if ns else None, ns))
        if W > 1 or self.force_collectives:
            plan = RankPlan(None, None, None, rp.send_counts, rp.recv_counts)
            with torch.cuda.stream(self.stream):
                if self.transport == "device":
                    exchange_rows(self.t_rf_send, self.t_rf_recv, plan.send_counts, plan.recv_counts, dist)
                else:
                    hs = self.t_rf_send[:ns].cpu()
                    hr = torch.empty((nr, self.t_rf_recv.shape[1]), dtype=torch.float64)
                    exchange_rows(hs, hr, plan.send_counts, plan.recv_counts, dist)
                    if nr:
                        self.t_rf_recv[:nr].copy_(hr)
            self._phase_sync()
        self.stats["refresh_fetched"] += nr
        bi = np.ascontiguousarray(rp.base_index)
        check(lib.rbpf_shard_smoother_refresh_end(self.ctx, _ip(bi), nr))

    def run(self, progress=None):
        import time
        lib, T, W = self.lib, self.prob.N_T, self.world
        nN, n = self.model.nNonLin, self.model.nLin
        XNK = np.zeros((nN, T, self.N_K), order="F")
        XLK = np.zeros((n, self.N_K), order="F")
        PK = np.zeros((n, n, self.N_K), order="F")
        self.aks = []
        tm = self.stats.setdefault("phase_s", dict(gather=0.0, normalise=0.0, anc=0.0, plan=0.0, exchange=0.0, step=0.0))
        self.stats["iter_s"] = []
        for k in range(self.N_K):
            t_iter = time.perf_counter()
            check(lib.rbpf_shard_smoother_begin(self.ctx, k))
            for t in range(T):
                if t == 0:
                    check(lib.rbpf_shard_smoother_step(self.ctx))
                    continue
                diag = self.stream.synchronize if self.sync_phases else (lambda: None)     # device time per phase (diagnostic)
                K = self.chol_refresh
                refresh = k > 0 and K > 1 and (t == 1 or (t - 1) % K == 0)
                ta = time.perf_counter()
                if k > 0 and not refresh:
                    # measurement part of the ancestor weights (one factorisation or sweep per particle): local data only, so it
                    # runs BEFORE the gather, whose extra row carries it along -- one collective per step instead of two
                    check(lib.rbpf_shard_smoother_anc_weights(self.ctx))
                    diag()
                t0 = time.perf_counter()
                self._gather()
                t1 = time.perf_counter()
                check(lib.rbpf_shard_smoother_normalise(self.ctx, 1))
                diag()
                t2 = time.perf_counter()
                if k > 0:
                    if refresh:                      # the refresh walks the state history: after gather + normalise, own all_gather
                        self._refresh()
                        self._gather_anc()
                    check(lib.rbpf_shard_smoother_anc_sample(self.ctx, 1 if refresh else 0))
                    diag()
                t3 = time.perf_counter()
                cnt = np.zeros(2 * W + 2, dtype=np.int64)
                check(lib.rbpf_shard_plan(self.ctx, cnt.ctypes.data_as(C.POINTER(C.c_int64))))
                t4 = time.perf_counter()
                if W > 1 or self.force_collectives:
                    self._exchange((cnt[:W], cnt[W:2 * W]), int(cnt[2 * W + 1]))
                t5 = time.perf_counter()
                check(lib.rbpf_shard_smoother_step(self.ctx))
                diag()
                t6 = time.perf_counter()
                tm["gather"] += t1 - t0; tm["normalise"] += t2 - t1; tm["anc"] += (t3 - t2) + (t0 - ta); tm["plan"] += t4 - t3
                tm["exchange"] += t5 - t4; tm["step"] += t6 - t5
                self.stats["migrated"] += int(cnt[2 * W])
                self.stats["steps"] += 1
            self._gather()
            check(lib.rbpf_shard_smoother_normalise(self.ctx, 0))
            xnk = np.zeros((nN, T), order="F")
            xlk = np.zeros(n)
            pk = np.zeros((n, n), order="F")
            ak, owner = C.c_int32(0), C.c_int32(0)
            check(lib.rbpf_shard_smoother_end(self.ctx, _dp(xnk), _dp(xlk), _dp(pk), C.byref(ak), C.byref(owner)))
            if W > 1:                                      # the owner's rows reach every rank (zeros elsewhere)
                buf = self._all_reduce_host(np.concatenate((xlk, pk.ravel(order="F"))))
                xlk, pk = buf[:n], buf[n:].reshape((n, n), order="F")
            XNK[:, :, k], XLK[:, k], PK[:, :, k] = xnk, xlk, pk
            self.aks.append(int(ak.value))
            self.stats["iter_s"].append(time.perf_counter() - t_iter)     # rbpf_shard_smoother_end synchronised
            if progress:
                progress(k)
        return XNK, XLK, PK

    def close(self):
        if self.ctx:
            self.sync()
        self.t_anc_local = self.t_anc_gather = self.t_rf_send = self.t_rf_recv = None
        super().close()
