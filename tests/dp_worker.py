"""One rank of the data-parallel PER-DQN rehearsal (tests/test_gpu_dataparallel.py starts two of these through
torch.distributed.run, gloo backend, both on cuda:0).  Each rank owns one replay shard and one replica of the learner,
takes `steps` steps through DeviceLearner.step_sharded and writes what it ended with to <outdir>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def shard_content(rank, n):
    """Deterministic transitions and priorities of shard `rank` (the parent process rebuilds the same arrays)."""
    rng = np.random.RandomState(1000 + rank)
    rows, nrows = rng.random_sample((n, 91)).astype(np.float32), rng.random_sample((n, 91)).astype(np.float32)
    act, rew = rng.randint(0, 5, n).astype(np.int32), rng.random_sample(n).astype(np.float32)
    done = (rng.random_sample(n) < 0.05).astype(np.float32)
    prio = np.abs(rng.standard_normal(n)).astype(np.float32)
    prio[:: 17 + rank] = 0.0
    prio[: n // 50] = 1.0
    return rows, act, rew, nrows, done, prio


def build_shard(rank, n, cap, mass_cap, dev):
    from mcts_q_amd.per import DevicePER
    rows, act, rew, nrows, done, prio = shard_content(rank, n)
    buf = DevicePER(cap, device=dev, mass_capacity=mass_cap)
    buf.push_rows(rows, act, rew, nrows, done)
    buf.update_priorities(np.arange(n), prio)
    return buf


CFG = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
SIZES, CAP, B, STEPS, TAU = (3000, 1700), 3000, 256, 3, 0.005


def seeded_weights():
    from oracle import torch_ref
    torch.manual_seed(77)
    return {k: v.detach().numpy() for k, v in torch_ref.RefQNet(CFG).state_dict().items()}


def main():
    outdir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from mcts_q_amd import NetSpec
    from mcts_q_amd.learner import DeviceLearner
    spec = NetSpec.from_cfg(CFG)
    L = DeviceLearner(spec, dev, process_group=dist.group.WORLD).load_state_dict(seeded_weights())
    buf = build_shard(rank, SIZES[rank], CAP, CAP * world, dev)
    u_all = np.random.RandomState(5).random_sample((STEPS, B))
    rec = {}
    for i in range(STEPS):
        L.step_sharded(buf, u_all[i], tau=TAU)
        torch.cuda.synchronize()
        d = L._dp
        rec[f"masses{i}"] = d["allv"].cpu().numpy()           # [masses W | sizes W] the step sampled with
        rec[f"count{i}"] = d["count"].cpu().numpy()
        rec[f"draw{i}"] = d["draw"].cpu().numpy()
        rec[f"exchange{i}"] = L._exchange.cpu().numpy().copy()
        rec[f"loss{i}"] = L.loss.cpu().numpy().copy()
        rec[f"gnorm{i}"] = L.grad_norm.cpu().numpy().copy()
        rec[f"prio{i}"] = buf.logical_priorities().copy()
        rec[f"grad{i}"] = L.grad.cpu().numpy().copy()
    rec.update(params=L.policy.flat.cpu().numpy(), target=L.target.flat.cpu().numpy(), m=L.exp_avg.cpu().numpy(),
               v=L.exp_avg_sq.cpu().numpy(), bcap=np.array(L._dp["bcap"]))
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **rec)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
