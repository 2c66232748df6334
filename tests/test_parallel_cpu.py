"""CPU, world_size 2 over gloo: the host logic of the data-parallel path (mcts_q_amd/parallel.py).

The kernels cannot run here, so each rank's shard is an oracle-backed stand-in with the DevicePER
shard interface (mass_and_size / shard_select / sample_shard / normalize_weights); what is under test is
the orchestration: mass all-gather, ownership of the shared draws, global N / mass / max in the IS
weights, loss scaling + one SUM all-reduce reproducing the single-process global-batch gradient."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import per_numpy, torch_ref  # noqa: E402


class OracleShard:
    """numpy stand-in for one DevicePER shard (logical order == storage order)."""

    def __init__(self, prio, mass_capacity, alpha=0.6):
        self.prio = np.asarray(prio, dtype=np.float64)
        self.alpha, self.device = alpha, torch.device("cpu")
        self.shift = per_numpy.mass_shift(mass_capacity)
        self.m, self.p = per_numpy.masses(self.prio, alpha, self.shift)
        self.cum = np.cumsum(self.m, dtype=np.uint64)

    def __len__(self):
        return len(self.prio)

    def total_mass(self):
        return torch.tensor([int(self.cum[-1])], dtype=torch.int64)

    def shard_targets(self, u, shard_mass, rank):
        masses = [int(x) for x in shard_mass.tolist()]
        M, prefix = sum(masses), sum(masses[:rank])
        k = np.rint(np.asarray(u, dtype=np.float64) * 2.0 ** 53).astype(np.int64)
        t = [(int(ki) * M) >> 53 for ki in k]
        mine = [prefix <= ti < prefix + masses[rank] for ti in t]
        loc = [ti - prefix if mi else 0 for ti, mi in zip(t, mine)]
        return torch.tensor(loc, dtype=torch.int64), torch.tensor(mine, dtype=torch.int32)

    def sample_at(self, targets, n_total, mass_total, beta=0.4):
        tg = targets.numpy().astype(np.uint64)
        idx = self.cum.searchsorted(tg, side="right").astype(np.int64)
        ratio = (np.float64(n_total) * np.minimum(self.p[idx], 256.0)) * np.float64(2.0 ** self.shift) / np.float64(mass_total)
        w64 = per_numpy.pow_det(ratio, -beta) if len(idx) else np.zeros(0)
        wmax = torch.tensor([w64.max() if len(idx) else 0.0], dtype=torch.float64)
        return torch.from_numpy(idx), torch.from_numpy(idx.copy()), torch.from_numpy(np.asarray(w64)), wmax

    # ---- the sync-free protocol (DevicePER.mass_and_size / shard_select / sample_shard) --------------
    def mass_and_size(self):
        return torch.tensor([int(self.cum[-1]), len(self.prio)], dtype=torch.int64)

    def shard_select(self, u, shard_mass, rank, bcap):
        loc, mine = self.shard_targets(np.asarray(u), shard_mass, rank)
        pos = np.nonzero(mine.numpy())[0]
        kept = min(len(pos), bcap)
        targets, draw = torch.zeros(bcap, dtype=torch.int64), torch.full((bcap,), -1, dtype=torch.int32)
        targets[:kept] = loc[pos[:kept]]
        draw[:kept] = torch.from_numpy(pos[:kept].astype(np.int32))
        return targets, draw, torch.tensor([kept, len(pos) - kept], dtype=torch.int32)

    def sample_shard(self, targets, count, shard_mass, shard_size, beta=0.4):
        n, bcap = int(count[0]), targets.numel()
        idx, _, w64, wmax = self.sample_at(targets[:n], float(shard_size.sum()), float(sum(int(m) for m in shard_mass.tolist())), beta)
        idx_p, w_p = torch.full((bcap,), -1, dtype=torch.int64), torch.zeros(bcap, dtype=torch.float64)
        idx_p[:n], w_p[:n] = idx, w64
        return dict(rows=idx_p.clone()), idx_p, w_p, wmax

    def normalize_weights(self, w64, wmax):
        return (w64 / wmax).float()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from mcts_q_amd import parallel
    try:
        cap, B = 600, 64
        rng = np.random.RandomState(42)
        prio_all = np.abs(rng.standard_normal(world * cap)).astype(np.float32).astype(np.float64)
        prio_all[::11] = 1.0
        prio_all[5::97] = 0.0
        sizes = [cap, cap - 150]                       # second shard not full
        lo = sum(sizes[:rank])
        shard = OracleShard(prio_all[lo:lo + sizes[rank]], mass_capacity=world * cap)
        u = parallel.shared_uniforms(np.random.RandomState(7), B)           # same seed on every rank
        assert parallel.draw_capacity(B, world) == 64 and parallel.draw_capacity(1024, 8) == 256
        sampler = parallel.ShardedSampler(shard, dist.group.WORLD)
        out = sampler.sample(u)
        # ---- single-buffer oracle over the concatenation ---------------------------------------
        concat = np.concatenate([prio_all[:sizes[0]], prio_all[sizes[0]:sizes[0] + sizes[1]]])
        idx_or, w_or, _ = per_numpy.fixed_point_sample(concat, u, capacity=world * cap)
        mine = (idx_or >= lo) & (idx_or < lo + sizes[rank])
        n = int(mine.sum())
        assert out["count"].tolist() == [n, 0] and out["batch_global"] == B
        assert out["idx"].numel() == sampler.bcap == 64                        # fixed provision, padded
        assert np.array_equal(out["draw"].numpy()[:n], np.nonzero(mine)[0]) and (out["draw"].numpy()[n:] == -1).all()
        assert np.array_equal(out["idx"].numpy()[:n] + lo, idx_or[mine]) and (out["idx"].numpy()[n:] == -1).all()
        assert np.array_equal(out["w"].numpy()[:n], w_or[mine])             # global N, mass and max
        assert (out["w"].numpy()[n:] == 0).all()                            # padding rows carry no weight
        counts = out["count"][:1].clone().long(); dist.all_reduce(counts)
        assert int(counts) == B                                              # every draw served exactly once
        sampler.sample(u)                                                    # the lagged counter check passes
        # a provision that is too small is reported one step later
        tight = parallel.ShardedSampler(shard, dist.group.WORLD, bcap=16)
        tight.sample(u)
        with pytest.raises(RuntimeError, match="provision"):
            tight.sample(u)

        # ---- gradient: local draws, loss scaled by B_local/B, one SUM all-reduce ------------------
        cfg = dict(price="gru", process="conv", gas="mlp", act="relu", flavor="per")
        torch.manual_seed(3)
        net, tgt = torch_ref.RefQNet(cfg), torch_ref.RefQNet(cfg)
        g = torch.Generator().manual_seed(11)
        rows, nrows = torch.rand(B, 91, generator=g), torch.rand(B, 91, generator=g)
        a, r = torch.randint(0, 5, (B,), generator=g), torch.rand(B, generator=g)
        w = torch.from_numpy(w_or)

        def grads_of(sel, scale):
            s, s2 = torch_ref.split_state_rows(rows[sel].numpy()), torch_ref.split_state_rows(nrows[sel].numpy())
            with torch.no_grad():
                y = r[sel] + 0.99 * tgt(*s2).max(1)[0]
            _, _, gr = torch_ref.td_loss_and_grads(net, s, a[sel], y, w[sel])
            return torch.cat([v.flatten() for v in gr.values()]) * scale

        sel = torch.from_numpy(np.nonzero(mine)[0])
        flat = grads_of(sel, len(sel) / B)
        parallel.allreduce_gradient(flat, dist.group.WORLD)
        ref = grads_of(torch.arange(B), 1.0)
        assert float((flat - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
        # replicas hold bit-identical reduced gradients
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1])
        assert list(parallel.tree_shard(9, dist.group.WORLD)) == ([0, 1, 2, 3, 4] if rank == 0 else [5, 6, 7, 8])
        open(os.path.join(outdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_sampling_and_gradient_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_single_rank_path_equals_plain_sampling():
    from mcts_q_amd import parallel
    rng = np.random.RandomState(1)
    prio = np.abs(rng.standard_normal(500)).astype(np.float32).astype(np.float64)
    u = rng.random_sample(32)
    out = parallel.sharded_sample(OracleShard(prio, 500), u, group=None)
    idx_or, w_or, _ = per_numpy.fixed_point_sample(prio, u, capacity=500)
    assert np.array_equal(out["idx"].numpy(), idx_or) and np.array_equal(out["w"].numpy(), w_or)
    assert np.array_equal(out["draw"].numpy(), np.arange(32)) and out["count"].tolist() == [32, 0]
