#!/usr/bin/env python3
"""Where the HOST time of one learner step goes: Python (argument struct, bookkeeping) vs the library call
(kernel launches, events).  python3 tools/host_split.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mcts_q_amd import NetSpec
from mcts_q_amd._lib import lib
from mcts_q_amd.learner import DeviceLearner
from mcts_q_amd.per import DevicePER
from bench import seeded_net
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N, B = 1000000, 1024
cfg = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
spec = NetSpec.from_cfg(cfg)
_, sd = seeded_net(cfg)
L = DeviceLearner(spec).load_state_dict(sd)
buf = DevicePER(N)
buf.push_rows(torch.rand(N, 91, device="cuda"), torch.randint(0, 5, (N,), device="cuda"), torch.rand(N, device="cuda"),
              torch.rand(N, 91, device="cuda"), torch.zeros(N, device="cuda"))
buf.update_priorities(torch.arange(N, device="cuda"), torch.rand(N, device="cuda") + 0.01)
u = torch.from_numpy(np.random.RandomState(0).random_sample((K + 10, B))).cuda()
inner = [0.0]
dll = lib._load()
real = dll.mctsq_per_dqn_step
def timed(*a):
    t = time.perf_counter()
    r = real(*a)
    inner[0] += time.perf_counter() - t
    return r
for i in range(10):
    L.step(buf, u[i], tau=0.005)
torch.cuda.synchronize()
lib.__dict__["mctsq_per_dqn_step"] = timed
setattr(lib, "mctsq_per_dqn_step", timed)
t0 = time.perf_counter()
for i in range(K):
    L.step(buf, u[10 + i], tau=0.005)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host per step {th / K * 1e6:.1f} us, inside mctsq_per_dqn_step {inner[0] / K * 1e6:.1f} us")
