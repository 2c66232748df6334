"""Data-parallel orchestration: one process per GPU, torch.distributed over RCCL.

* Leaf evaluation: every rank evaluates the leaves of its own trees; nothing is exchanged.
* Replay: every rank owns one shard (a DevicePER ring fed by its own trees).  The shards are sampled as
  ONE buffer whose logical order is the concatenation of the shards in rank order, with the same
  exact-integer inverse CDF as a single buffer: every rank draws the same B uniforms (shared seed),
  the W root masses are all-gathered, and each rank serves the draws that land in its interval of
  the concatenated mass line.  IS weights use the global N, the global mass and the global maximum.
* Update: each rank back-propagates its own draws with the loss scaled by B_local / B_global, ONE
  all-reduce(SUM) over the flat gradient buffer (+ the loss behind it) yields the exact global-batch
  gradient, then every rank applies the identical clip / Adam / Polyak step, so replicas stay bit-identical.

Nothing here reads device memory on the host inside a step.  The number of draws a rank serves varies from
step to step, so every rank provisions a fixed number of rows (`draw_capacity`) and the kernels pad the
unused ones with weight 0 / index -1: such rows contribute exactly nothing to loss, gradient and priorities.
The (kept, dropped) counters of a step are copied to pinned host memory asynchronously and looked at one
step later; a dropped draw (the rank's share outgrew its provision) raises.

The functions take a `shard` object (DevicePER, or any stand-in with the same methods) and a process group,
so the host logic is exercised on CPU with the gloo backend (tests/test_parallel_cpu.py).
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist


def _world(group):
    return (dist.get_world_size(group), dist.get_rank(group)) if group is not None else (1, 0)


def all_reduce(t: torch.Tensor, op=None, group=None):
    """dist.all_reduce; under the gloo backend (CPU tests, several ranks rehearsed on one GPU) device
    tensors are staged through the host, which gloo on ROCm needs for anything but tiny tensors."""
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and dist.get_backend(group) == "gloo":
        c = t.cpu()
        dist.all_reduce(c, op=op, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def all_gather_rows(local: torch.Tensor, group=None):
    """[k] on every rank -> [W, k] on the device (host-staged under gloo like all_reduce)."""
    world, _ = _world(group)
    if world == 1:
        return local.unsqueeze(0)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        c = local.cpu()
        parts = [torch.empty_like(c) for _ in range(world)]
        dist.all_gather(parts, c, group=group)
        return torch.stack(parts).to(local.device)
    out = torch.empty(world, local.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), local.contiguous(), group=group)
    return out


def shared_uniforms(rng, batch_size: int):
    """The B global draws.  `rng` is a numpy RandomState / Generator seeded identically on every rank."""
    return rng.random_sample(int(batch_size))


def draw_capacity(batch_global: int, world: int, share: float | None = None, sigmas: float = 10.0) -> int:
    """Rows a rank provisions for its share of B global draws: mean + `sigmas` standard deviations of
    Binomial(B, share) + 16, rounded up to the kernels' 16-row tile, never more than B rounded up."""
    B = int(batch_global)
    if world <= 1:
        return B
    p = 1.0 / world if share is None else min(1.0, max(0.0, float(share)))
    n = B * p + sigmas * math.sqrt(max(B * p * (1.0 - p), 1.0)) + 16
    return int(min(-(-B // 16) * 16, -(-int(math.ceil(n)) // 16) * 16))


class ShardedSampler:
    """Per-rank state of the sync-free sharded sampling: the row provision and the lagged look at the
    previous step's counters."""

    def __init__(self, shard, group=None, beta: float = 0.4, bcap: int | None = None, reuse: bool = True):
        """reuse: the tensors of one sample() are overwritten by the next one (a training loop consumes them within
        the step; the step is host-bound and ~20 allocations per step are a tenth of it).  Shards that do not take
        the `reuse` keyword (test stand-ins) allocate as before."""
        self.shard, self.group, self.beta, self.bcap = shard, group, float(beta), bcap
        self._kw = dict(reuse=True) if reuse and getattr(shard, "_reused", None) is not None else {}
        self.world, self.rank = _world(group)
        self._pending = None            # (pinned host copy of [kept, dropped], event) of the previous step
        self._slots, self._turn = None, 0

    def _check_previous(self):
        if self._pending is None:
            return
        host, ev = self._pending
        if ev is not None:
            ev.synchronize()            # recorded a whole step ago: already complete, no stall
        if int(host[1]) != 0:
            raise RuntimeError(f"sharded replay: {int(host[1])} draws of the previous step exceeded this rank's "
                               f"provision of {self.bcap} rows; raise bcap (draw_capacity)")
        self._pending = None

    def sample(self, u):
        """u: fp64 [B] global uniforms, identical on every rank.  Returns dict(batch, idx (-1 = padding), w fp32
        (0 = padding), draw int32 positions in u (-1 = padding), count int32 [2] device, batch_global)."""
        shard, group = self.shard, self.group
        dev = shard.device
        self._check_previous()
        u = torch.as_tensor(u, dtype=torch.float64).to(dev)
        B = int(u.numel())
        if self.bcap is None:
            self.bcap = draw_capacity(B, self.world)
        kw = self._kw
        allv = all_gather_rows(shard.mass_and_size(**kw), group)      # [W, 2] on the device
        masses, sizes = allv[:, 0].contiguous(), allv[:, 1].contiguous()
        targets, draw, count = shard.shard_select(u, masses, self.rank, self.bcap, **kw)
        batch, idx, w64, wmax = shard.sample_shard(targets, count, masses, sizes, beta=self.beta, **kw)
        if self.world > 1:
            all_reduce(wmax, dist.ReduceOp.MAX, group)
        w = shard.normalize_weights(w64, wmax, **kw)
        if count.is_cuda:
            if self._slots is None:                 # two (pinned words, event) pairs, used in turn: no allocation per step
                self._slots = [(torch.empty(2, dtype=torch.int32, pin_memory=True), torch.cuda.Event()) for _ in range(2)]
            self._turn ^= 1
            host, ev = self._slots[self._turn]
            host.copy_(count, non_blocking=True)
            ev.record()
            self._pending = (host, ev)
        else:
            self._pending = (count.clone(), None)
        return dict(batch=batch, idx=idx, w=w, draw=draw, count=count, batch_global=B)


def sharded_sample(shard, u, group=None, beta: float = 0.4, bcap: int | None = None):
    """One-shot form of ShardedSampler.sample (tests, single calls)."""
    return ShardedSampler(shard, group, beta, bcap, reuse=False).sample(u)


def allreduce_gradient(flat_grad: torch.Tensor, group=None):
    """The one collective of the update: SUM over ranks of gradients already scaled by B_local/B_global."""
    if group is not None and dist.get_world_size(group) > 1:
        all_reduce(flat_grad, dist.ReduceOp.SUM, group)
    return flat_grad


def tree_shard(n_trees: int, group=None):
    """Contiguous block of tree ids owned by this rank (leaf evaluation shards by tree, no exchange)."""
    world, rank = _world(group)
    per, rem = divmod(int(n_trees), world)
    lo = rank * per + min(rank, rem)
    return range(lo, lo + per + (1 if rank < rem else 0))
