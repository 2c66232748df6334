#!/usr/bin/env python3
"""The lock-step search with the HIP network: wall-clock split of the C driver per search of all trees, for a number
of tree blocks.  python3 tools/search_split.py [trees] [searches] [iterations] [blocks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mcts_q_amd import NetSpec
from mcts_q_amd.qnet import DeviceQNet
from mcts_q_amd.mcts import DeviceLeafEvaluator
from mcts_q_amd.ptgenv import PtGBatchEnv
from mcts_q_amd.synthetic_env import synthetic_ptg_kwargs
from mcts_q_amd.treesearch import LockstepMCTS
from bench import seeded_net

K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
halves = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cfg = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
_, sd = seeded_net(cfg)
net = DeviceQNet(NetSpec.from_cfg(cfg)).load_state_dict(sd)
env = PtGBatchEnv(synthetic_ptg_kwargs(0, scale=1.0, train_days=46, env_values=dict(eps_len_d=40)), seed=0)
drv = LockstepMCTS(K, DeviceLeafEvaluator(net, -(-K // halves), halves, zero_copy=True), iters, 42, 19652, 1.25,
                   batch_env=env, seed=0)
st = env.reset(np.arange(K))
a = drv.predict(st, train=False)
drv.prof = dict(wait=0.0, backup=0.0, expand=0.0, launch=0.0)
t0 = time.perf_counter()
for _ in range(S):
    st = env.step_batch(st, a)[0]
    a = drv.predict(st, train=False)
dt = (time.perf_counter() - t0) / S
print(f"{K} trees x {iters} iterations, {halves} blocks: {dt * 1e3:.2f} ms per search = {K * (iters + 1) / dt / 1e6:.2f} M leaf evals/s; " +
      ", ".join(f"{k} {v / S * 1e3:.2f}" for k, v in drv.prof.items()))
