"""Data-parallel orchestration: one process per GPU, torch.distributed over RCCL.

* Leaf evaluation: every rank evaluates the leaves of its own trees; nothing is exchanged.
* Replay: every rank owns one shard (a DevicePER ring fed by its own trees).  The shards are sampled as
  ONE buffer whose logical order is the concatenation of the shards in rank order, with the same
  exact-integer inverse CDF as a single buffer: every rank draws the same B uniforms (shared seed),
  the W root masses are all-gathered, and each rank serves the draws that land in its interval of
  the concatenated mass line.  IS weights use the global N, the global mass and the global maximum.
* Update: each rank back-propagates its own draws with the loss scaled by B_local / B_global, ONE
  all-reduce(SUM) over the flat gradient buffer yields the exact global-batch gradient, then every
  rank applies the identical clip / Adam / Polyak step, so replicas stay bit-identical.

The functions take a `shard` object (DevicePER, or any stand-in with the same five methods) and a
process group, so the host logic is exercised on CPU with the gloo backend (tests/test_parallel_cpu.py).
"""
from __future__ import annotations

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


def shared_uniforms(rng, batch_size: int):
    """The B global draws.  `rng` is a numpy RandomState / Generator seeded identically on every rank."""
    return rng.random_sample(int(batch_size))


def sharded_sample(shard, u, group=None, beta: float = 0.4):
    """Sample the concatenation of all ranks' shards at the global uniforms `u` (fp64 [B], identical
    on every rank).  Returns dict(idx, phys: local indices of THIS rank's draws, w: their fp32 IS
    weights, draw: positions of those draws in `u`, n_global, batch_global)."""
    world, rank = _world(group)
    dev = shard.device
    local = torch.stack([shard.total_mass().reshape(()), torch.tensor(len(shard), dtype=torch.int64, device=dev)])
    if world > 1:
        allv = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(allv, local, group=group)
        allv = torch.stack(allv)                               # [W, 2] on the device, no host sync
    else:
        allv = local.unsqueeze(0)
    masses, sizes = allv[:, 0].contiguous(), allv[:, 1]
    targets, mine = shard.shard_targets(u, masses, rank)
    draw = torch.nonzero(mine, as_tuple=False).squeeze(1)      # the one host sync: B_local is a launch size
    n_total = float(sizes.sum().item())
    mass_total = float(sum(int(m) for m in masses.tolist()))
    idx, phys, w64, wmax = shard.sample_at(targets[draw], n_total, mass_total, beta=beta)
    if world > 1:
        all_reduce(wmax, dist.ReduceOp.MAX, group)
    w = shard.normalize_weights(w64, wmax)
    return dict(idx=idx, phys=phys, w=w, draw=draw, n_global=int(n_total), batch_global=int(u.shape[0]))


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
