"""Device-resident prioritized replay: ring buffer of transitions + 64-ary integer sum-tree.

Host-side mirror of PrioritizedReplayBuffer (/root/reference/src/mctsq_config_dqn_per.py:238-321):
push (max-priority rule), sample (proportional, with replacement, IS weights), update_priorities
(sequential semantics: last duplicate wins).  Logical index 0 is the oldest entry, as in
deque(maxlen).  All arithmetic is in libmctsq_hip.so; see DESIGN.md for the fixed-point mass
formulation that makes sampled indices bit-identical to the oracle.
"""
from __future__ import annotations

import numpy as np
import torch

from ._lib import lib, ptr, stream_ptr, MctsqError

ROW = 96          # floats per stored state record (91 used): price 25 | process 10x6 | gas 3x2 | pad


class DevicePER:
    def __init__(self, capacity: int, alpha: float = 0.6, device="cuda:0", mass_capacity: int | None = None):
        if not torch.cuda.is_available():
            raise MctsqError("DevicePER needs a GPU: the HIP path has no CPU fallback")
        self.capacity, self.alpha = int(capacity), float(alpha)
        # capacity the fixed-point mass scale is derived from: the sum over all shards when this
        # ring is one shard of a data-parallel buffer (mctsq_hip.h, "sharded replay")
        self.mass_capacity = int(mass_capacity) if mass_capacity is not None else self.capacity
        self.device = torch.device(device)
        d = self.device
        self.head, self.size = 0, 0
        words = lib.check(lib.mctsq_per_tree_words(self.capacity))
        self.tree = torch.zeros(words, dtype=torch.int64, device=d)       # uint64 bit patterns
        self.prio = torch.zeros(self.capacity, dtype=torch.float64, device=d)
        self.maxprio = torch.ones(1, dtype=torch.float64, device=d)
        self.states = torch.zeros(self.capacity, ROW, dtype=torch.float32, device=d)
        self.next_states = torch.zeros(self.capacity, ROW, dtype=torch.float32, device=d)
        self.actions = torch.zeros(self.capacity, dtype=torch.int32, device=d)
        self.rewards = torch.zeros(self.capacity, dtype=torch.float32, device=d)
        self.dones = torch.zeros(self.capacity, dtype=torch.float32, device=d)
        self.stamp = torch.full((self.capacity,), -1, dtype=torch.int32, device=d)
        self._max_dirty = False
        self._ws = torch.zeros(1 << 16, dtype=torch.uint8, device=d)      # control words must start out zero
        self.last_local_wmax = torch.ones(1, dtype=torch.float64, device=d)   # max IS weight before normalisation

    def __len__(self):
        return self.size

    def _st(self):
        return stream_ptr(self.device)

    def _ws_for(self, B):
        need = lib.check(lib.mctsq_per_workspace_bytes(self.capacity, B))
        if self._ws.numel() < need:
            self._ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---- push -------------------------------------------------------------------------
    def push_rows(self, rows, actions, rewards, next_rows, dones):
        """Append K transitions.  rows/next_rows: [K,91] (or [K,96]) fp32; actions int; rewards,
        dones float.  Every new entry gets the current max priority (dqn_per.py:265-267)."""
        K = int(rows.shape[0])
        if K == 0:
            return
        if K > self.capacity:
            raise ValueError("push batch larger than capacity")
        d = self.device
        if self._max_dirty:
            ws = self._ws_for(1)
            lib.check(lib.mctsq_per_refresh_max(ptr(self.prio), self.capacity, self.head, self.size,
                                                ptr(self.maxprio), ptr(ws), ws.numel(), self._st()))
            self._max_dirty = False
        # The K new slots are one contiguous run of the ring (two when it wraps): slice copies, no index kernels.
        # Host inputs: the two row blocks as they are, the three small vectors as one [3, K] fp32 array (actions are
        # small integers, exact in fp32).
        start = (self.head + self.size) % self.capacity
        n1 = min(K, self.capacity - start)
        w = int(rows.shape[1])
        host = not any(torch.is_tensor(t) for t in (rows, actions, rewards, next_rows, dones))
        if host:
            small = np.empty((3, K), dtype=np.float32)
            small[0] = actions; small[1] = rewards; small[2] = dones
            sm = torch.from_numpy(small).to(d)
            same = next_rows is rows or (isinstance(rows, np.ndarray) and isinstance(next_rows, np.ndarray) and
                                         rows.shape == next_rows.shape and rows.dtype == next_rows.dtype and
                                         rows.__array_interface__["data"][0] == next_rows.__array_interface__["data"][0] and
                                         rows.strides == next_rows.strides)
            rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32)).to(d)
            # the expansion transitions of the tree search carry the same row as state and next state (mcts_per.py:337-340)
            next_rows = rows if same else torch.from_numpy(np.ascontiguousarray(next_rows, dtype=np.float32)).to(d)
            actions, rewards, dones = sm[0].to(torch.int32), sm[1], sm[2]
        else:
            rows = torch.as_tensor(rows, dtype=torch.float32, device=d)
            next_rows = torch.as_tensor(next_rows, dtype=torch.float32, device=d)
            actions = torch.as_tensor(actions, device=d).to(torch.int32)
            rewards = torch.as_tensor(rewards, device=d).to(torch.float32)
            dones = torch.as_tensor(dones, device=d).to(torch.float32)
        for dst, src in ((self.states[:, :w], rows), (self.next_states[:, :w], next_rows), (self.actions, actions),
                         (self.rewards, rewards), (self.dones, dones)):
            dst[start:start + n1] = src[:n1]
            if n1 < K:
                dst[:K - n1] = src[n1:]
        lib.check(lib.mctsq_per_push(ptr(self.tree), ptr(self.prio), ptr(self.maxprio), self.capacity,
                                     self.head, self.size, K, self.alpha, self.mass_capacity, self._st()))
        over = max(0, self.size + K - self.capacity)
        self.size = min(self.size + K, self.capacity)
        self.head = (self.head + over) % self.capacity

    # ---- sample -----------------------------------------------------------------------
    def sample_indices(self, u, beta: float = 0.4):
        """u: fp64 uniforms in [0,1) (host array or device tensor).  Returns device tensors
        (logical idx int64 [B], physical slot int64 [B], IS weights fp32 [B])."""
        if self.size == 0:
            raise ValueError("cannot sample from an empty buffer")
        u = torch.as_tensor(u, dtype=torch.float64).to(self.device).contiguous()
        B = u.numel()
        idx = torch.empty(B, dtype=torch.int64, device=self.device)
        phys = torch.empty(B, dtype=torch.int64, device=self.device)
        w = torch.empty(B, dtype=torch.float32, device=self.device)
        ws = self._ws_for(B)
        lib.check(lib.mctsq_per_sample(ptr(self.tree), ptr(self.prio), self.capacity, self.head, self.size,
                                       ptr(u), B, self.alpha, beta, self.mass_capacity, ptr(idx), ptr(phys), ptr(w),
                                       ptr(self.last_local_wmax), ptr(ws), ws.numel(), self._st()))
        return idx, phys, w

    def _batch_tensors(self, B):
        d = self.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=d)
        return dict(sprice=f(B, 25, 1), sprocess=f(B, 10, 6), sgas=f(B, 3, 2), nprice=f(B, 25, 1),
                    nprocess=f(B, 10, 6), ngas=f(B, 3, 2), act=torch.empty(B, dtype=torch.int32, device=d),
                    rew=f(B), done=f(B))

    def _reused(self, key, make):
        """Output buffers of the shard primitives when the caller asked for reuse (parallel.ShardedSampler in a
        steady-state loop: the results of one step are dead before the next one samples): allocated once per shape."""
        c = self.__dict__.setdefault("_out_cache", {})
        if key not in c:
            c[key] = make()
        return c[key]

    def _ring_ptrs(self):
        rp = self.__dict__.get("_ring_ptr_cache")
        if rp is None or rp[0] is not self.states:           # the ring tensors live as long as the buffer
            rp = (self.states, (ptr(self.states), ptr(self.next_states), ptr(self.actions), ptr(self.rewards),
                                ptr(self.dones)))
            self._ring_ptr_cache = rp
        return rp[1]

    @staticmethod
    def _batch_ptrs(out):
        return tuple(ptr(out[k]) for k in ("sprice", "sprocess", "sgas", "nprice", "nprocess", "ngas", "act", "rew",
                                           "done"))

    def sample_batch(self, u, beta: float = 0.4):
        """sample_indices + gather in ONE launch (dqn_per.py:283-305, 415-452): returns (batch dict of device
        tensors, IS weights fp32 [B], logical idx int64 [B], physical slots int64 [B])."""
        if self.size == 0:
            raise ValueError("cannot sample from an empty buffer")
        u = torch.as_tensor(u, dtype=torch.float64).to(self.device).contiguous()
        B = u.numel()
        idx = torch.empty(B, dtype=torch.int64, device=self.device)
        phys = torch.empty(B, dtype=torch.int64, device=self.device)
        w = torch.empty(B, dtype=torch.float32, device=self.device)
        out = self._batch_tensors(B)
        ws = self._ws_for(B)
        lib.check(lib.mctsq_per_sample_batch(ptr(self.tree), ptr(self.prio), self.capacity, self.head, self.size,
                                             ptr(u), B, self.alpha, beta, self.mass_capacity, ptr(idx), ptr(phys),
                                             ptr(w), ptr(self.last_local_wmax), *self._ring_ptrs(),
                                             *self._batch_ptrs(out), ptr(ws), ws.numel(), self._st()))
        return out, w, idx, phys

    # ---- shard primitives (one ring = one shard of a buffer sampled as a whole; parallel.py) ------
    def mass_and_size(self, reuse: bool = False):
        """int64 device tensor [2] = (root of the sum-tree as a uint64 bit pattern < 2^62, live entries): the pair
        the ranks all-gather; made on the device so no host value is copied in."""
        mk = lambda: torch.empty(2, dtype=torch.int64, device=self.device)
        out = self._reused("mass", mk) if reuse else mk()
        lib.check(lib.mctsq_per_total_mass(ptr(self.tree), self.capacity, self.size, ptr(out), self._st()))
        return out

    def total_mass(self):
        return self.mass_and_size()[:1]

    def shard_targets(self, u, shard_mass, rank: int):
        """u fp64 [B] (the SAME draws on every rank), shard_mass int64 [W] -> (local integer targets
        int64 [B], mine int32 [B])."""
        u = torch.as_tensor(u, dtype=torch.float64).to(self.device).contiguous()
        shard_mass = shard_mass.to(self.device, torch.int64).contiguous()
        B = u.numel()
        targets = torch.empty(B, dtype=torch.int64, device=self.device)
        mine = torch.empty(B, dtype=torch.int32, device=self.device)
        lib.check(lib.mctsq_per_shard_targets(ptr(u), B, ptr(shard_mass), shard_mass.numel(), int(rank),
                                              ptr(targets), ptr(mine), self._st()))
        return targets, mine

    def shard_select(self, u, shard_mass, rank: int, bcap: int, reuse: bool = False):
        """This rank's share of the global draws `u`, compacted on the device: (targets int64 [bcap], draw
        positions int32 [bcap] (-1 = padding), count int32 [2] = (kept, dropped))."""
        u = torch.as_tensor(u, dtype=torch.float64).to(self.device).contiguous()
        mk = lambda: (torch.empty(bcap, dtype=torch.int64, device=self.device),
                      torch.empty(bcap, dtype=torch.int32, device=self.device),
                      torch.empty(2, dtype=torch.int32, device=self.device))
        targets, draw, count = self._reused(("select", bcap), mk) if reuse else mk()
        lib.check(lib.mctsq_per_shard_select(ptr(u), u.numel(), ptr(shard_mass), shard_mass.numel(), int(rank),
                                             int(bcap), ptr(targets), ptr(draw), ptr(count), self._st()))
        return targets, draw, count

    def sample_shard(self, targets, count, shard_mass, shard_size, beta: float = 0.4, reuse: bool = False):
        """bcap rows at the compacted targets, totals from the all-gathered device arrays, transitions gathered
        in the same launch: (batch, logical idx (-1 = padding), raw fp64 weights, local max fp64 [1])."""
        bcap = targets.numel()
        mk = lambda: (torch.empty(bcap, dtype=torch.int64, device=self.device),
                      torch.empty(bcap, dtype=torch.int64, device=self.device),
                      torch.empty(bcap, dtype=torch.float64, device=self.device),
                      torch.empty(1, dtype=torch.float64, device=self.device), self._batch_tensors(bcap))
        idx, phys, w64, wmax, out = self._reused(("sample", bcap), mk) if reuse else mk()
        ws = self._ws_for(1)
        lib.check(lib.mctsq_per_sample_shard(ptr(self.tree), ptr(self.prio), self.capacity, self.head, self.size,
                                             ptr(targets), ptr(count), bcap, self.alpha, beta, self.mass_capacity,
                                             ptr(shard_mass), ptr(shard_size), shard_mass.numel(), ptr(idx),
                                             ptr(phys), ptr(w64), ptr(wmax), *self._ring_ptrs(),
                                             *self._batch_ptrs(out), ptr(ws), ws.numel(), self._st()))
        return out, idx, w64, wmax

    def sample_at(self, targets, n_total: float, mass_total: float, beta: float = 0.4):
        """Local draws at explicit integer targets -> (logical idx, physical slot, raw fp64 weights,
        their max as fp64 [1])."""
        targets = targets.to(self.device, torch.int64).contiguous()
        B = targets.numel()
        idx = torch.empty(B, dtype=torch.int64, device=self.device)
        phys = torch.empty(B, dtype=torch.int64, device=self.device)
        w64 = torch.empty(B, dtype=torch.float64, device=self.device)
        wmax = torch.zeros(1, dtype=torch.float64, device=self.device)
        if B:
            lib.check(lib.mctsq_per_sample_at(ptr(self.tree), ptr(self.prio), self.capacity, self.head, self.size,
                                              ptr(targets), B, self.alpha, beta, self.mass_capacity, float(n_total),
                                              float(mass_total), ptr(idx), ptr(phys), ptr(w64), ptr(wmax),
                                              self._st()))
        return idx, phys, w64, wmax

    def normalize_weights(self, w64, wmax, reuse: bool = False):
        mk = lambda: torch.empty(w64.numel(), dtype=torch.float32, device=self.device)
        w = self._reused(("norm", w64.numel()), mk) if reuse else mk()
        lib.check(lib.mctsq_per_normalize_weights(ptr(w64), w64.numel(), ptr(wmax), ptr(w), self._st()))
        return w

    def gather(self, phys):
        """Batch assembly (dqn_per.py:415-452) on the device: six window tensors + a, r, done."""
        B = phys.numel()
        out = self._batch_tensors(B)
        lib.check(lib.mctsq_per_gather(*self._ring_ptrs(), ptr(phys), B, *self._batch_ptrs(out), self._st()))
        return out

    # ---- priority write-back ----------------------------------------------------------
    def update_priorities(self, idx, prio):
        """prio[idx[i]] = p[i] in order (last duplicate wins), idx are LOGICAL indices."""
        if not torch.is_tensor(idx) or not idx.is_cuda:      # host indices: range-checked like deque indexing
            hidx = np.asarray(idx.cpu() if torch.is_tensor(idx) else idx, dtype=np.int64)
            if hidx.size and (hidx.min() < -self.size or hidx.max() >= self.size):
                raise IndexError("deque index out of range")
            idx = np.where(hidx < 0, hidx + self.size, hidx)
        idx = torch.as_tensor(idx, dtype=torch.int64).to(self.device).contiguous()
        prio = torch.as_tensor(prio, dtype=torch.float32).to(self.device).contiguous()
        if idx.numel() != prio.numel():
            raise ValueError("indices / priorities length mismatch")
        lib.check(lib.mctsq_per_update(ptr(self.tree), ptr(self.prio), ptr(self.stamp), self.capacity, self.head, self.size,
                                       ptr(idx), ptr(prio), idx.numel(), self.alpha, self.mass_capacity, self._st()))
        self._max_dirty = True

    def set_priorities_exact(self, prio):
        """Install priorities for all live entries in logical order (checkpoint import).  The
        reference's priorities are 1.0 or fp32 |td| values, i.e. exactly representable in fp32."""
        prio = np.asarray(prio, dtype=np.float64)
        if len(prio) != self.size:
            raise ValueError("priority list length differs from the number of stored transitions")
        p32 = prio.astype(np.float32)
        if not np.array_equal(p32.astype(np.float64), prio):
            raise MctsqError("checkpoint priorities are not fp32-representable (not written by PER-DQN)")
        self.update_priorities(np.arange(self.size), p32)

    # ---- export (checkpoint interchange, dqn_per.py:487-492) ----------------------------
    def logical_priorities(self):
        p = self.prio.cpu().numpy()
        order = (self.head + np.arange(self.size)) % self.capacity
        return p[order]
