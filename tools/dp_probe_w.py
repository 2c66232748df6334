"""Probe: raw fp64 importance weights of the sharded sampling path against the single-buffer path (same draws)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import dp_worker as W
from mcts_q_amd.per import DevicePER
dev = torch.device("cuda", 0)
shards = [W.build_shard(r, n, W.CAP, W.CAP * 2, dev) for r, n in enumerate(W.SIZES)]
single = DevicePER(W.CAP * 2, device=dev)
prios = []
for r, n in enumerate(W.SIZES):
    rows, act, rew, nrows, done, prio = W.shard_content(r, n)
    single.push_rows(rows, act, rew, nrows, done); prios.append(prio)
single.update_priorities(np.arange(sum(W.SIZES)), np.concatenate(prios))
u = np.random.RandomState(5).random_sample((W.STEPS, W.B))[0]
idx, phys, w = single.sample_indices(u)
torch.cuda.synchronize()
raw_single = single._ws[64:64 + 8 * W.B].view(torch.float64).cpu().numpy()
print("single wmax", repr(float(single.last_local_wmax.item())), raw_single.max())
ms = torch.stack([s.mass_and_size() for s in shards])
masses, sizes = ms[:, 0].contiguous(), ms[:, 1].contiguous()
print("masses", masses.tolist(), "single root", int(single.total_mass().item()), sum(masses.tolist()))
for r, s in enumerate(shards):
    t, draw, count = s.shard_select(u, masses, r, 224)
    batch, li, w64, wmax = s.sample_shard(t, count, masses, sizes)
    k = int(count[0]); d = draw[:k].cpu().numpy()
    a, b = w64[:k].cpu().numpy(), raw_single[d]
    print(r, k, "max rel diff raw", np.abs(a - b).max() / b.max(), "wmax", repr(float(wmax.item())))
