#!/usr/bin/env python3
"""Runs K PER-DQN train steps (sample -> gather -> grad -> clip/Adam/Polyak -> priority write-back) for
profiling: `rocprofv3 --kernel-trace --stats -- python3 tools/train_loop.py [combo] [B] [K]`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mcts_q_amd import NetSpec
from mcts_q_amd.learner import DeviceLearner
from mcts_q_amd.per import DevicePER
from bench import seeded_net
combo = sys.argv[1] if len(sys.argv) > 1 else "gru/gru/gru"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
N = int(os.environ.get("MCTSQ_REPLAY", "1000000"))
pe, pr, ga = combo.split("/")
cfg = dict(price=pe, process=pr, gas=ga, act="relu", flavor="per")
spec = NetSpec.from_cfg(cfg)
_, sd = seeded_net(cfg)
L = DeviceLearner(spec).load_state_dict(sd)
buf = DevicePER(N)
buf.push_rows(torch.rand(N, 91, device="cuda"), torch.randint(0, 5, (N,), device="cuda"), torch.rand(N, device="cuda"),
              torch.rand(N, 91, device="cuda"), torch.zeros(N, device="cuda"))
buf.update_priorities(torch.arange(N, device="cuda"), torch.rand(N, device="cuda") + 0.01)
u = torch.from_numpy(np.random.RandomState(0).random_sample((K + 10, B))).cuda()
def step(i):
    if os.environ.get("MCTSQ_MODULAR"):
        batch, w, idx, _ = buf.sample_batch(u[i])
        abs_td = L.compute_grad(batch, w)
        L.apply_grad(tau=0.005)
        buf.update_priorities(idx, abs_td)
    else:
        L.step(buf, u[i], tau=0.005)
for i in range(10):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    step(10 + i)
t_host = time.perf_counter() - t0            # the host has enqueued everything; the device may still be busy
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"host enqueue {t_host / K * 1e3:.3f} ms/step")
print(f"{combo} B={B}: {dt / K * 1e3:.3f} ms/step, {K / dt:.0f} steps/s, loss {L.loss.item():.5f}")
