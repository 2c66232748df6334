#!/usr/bin/env python3
"""Feasibility probe: the learner step captured as ONE HIP graph (torch.cuda.CUDAGraph around DeviceLearner.step with
frozen arguments - the Adam bias correction does not advance, so this is a TIMING experiment only) against the plain
enqueue.  python3 tools/graph_probe.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mcts_q_amd import NetSpec
from mcts_q_amd.learner import DeviceLearner
from mcts_q_amd.per import DevicePER
from bench import seeded_net
K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
N, B = 1000000, 1024
cfg = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
spec = NetSpec.from_cfg(cfg)
_, sd = seeded_net(cfg)
L = DeviceLearner(spec).load_state_dict(sd)
buf = DevicePER(N)
buf.push_rows(torch.rand(N, 91, device="cuda"), torch.randint(0, 5, (N,), device="cuda"), torch.rand(N, device="cuda"),
              torch.rand(N, 91, device="cuda"), torch.zeros(N, device="cuda"))
buf.update_priorities(torch.arange(N, device="cuda"), torch.rand(N, device="cuda") + 0.01)
u = torch.from_numpy(np.random.RandomState(0).random_sample((B,))).cuda()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for i in range(10):
        L.step(buf, u, tau=0.005)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        L.step(buf, u, tau=0.005)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"plain: host {th / K * 1e6:.1f} us/step, total {(time.perf_counter() - t0) / K * 1e6:.1f} us/step")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    L.step(buf, u, tau=0.005)
torch.cuda.synchronize()
for i in range(10):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    g.replay()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"graph: host {th / K * 1e6:.1f} us/step, total {(time.perf_counter() - t0) / K * 1e6:.1f} us/step, loss {L.loss.item():.5f}")
