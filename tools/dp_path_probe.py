#!/usr/bin/env python3
"""The data-parallel code path of the learner step on ONE rank (collectives are no-ops at world 1): what the step costs
before any collective is added.  python3 tools/dp_path_probe.py [K] [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mcts_q_amd import NetSpec, parallel
from mcts_q_amd.learner import DeviceLearner
from mcts_q_amd.per import DevicePER
from bench import seeded_net
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = 1000000
cfg = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
spec = NetSpec.from_cfg(cfg)
_, sd = seeded_net(cfg)
L = DeviceLearner(spec).load_state_dict(sd)
buf = DevicePER(N)
buf.push_rows(torch.rand(N, 91, device="cuda"), torch.randint(0, 5, (N,), device="cuda"), torch.rand(N, device="cuda"),
              torch.rand(N, 91, device="cuda"), torch.zeros(N, device="cuda"))
buf.update_priorities(torch.arange(N, device="cuda"), torch.rand(N, device="cuda") + 0.01)
u = torch.from_numpy(np.random.RandomState(0).random_sample((K + 10, B))).cuda()
sampler = parallel.ShardedSampler(buf, None)
def step(i):
    smp = sampler.sample(u[i])
    abs_td = L.compute_grad(smp["batch"], smp["w"], global_batch=B)
    L.apply_grad(tau=0.005)
    buf.update_priorities(smp["idx"], abs_td)
for i in range(10):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    step(10 + i)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"DP path, world 1, B={B}: host {th / K * 1e3:.3f} ms/step, total {(time.perf_counter() - t0) / K * 1e3:.3f} ms/step")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(K):
        step(10 + i)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
