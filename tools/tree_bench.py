#!/usr/bin/env python3
"""Lock-step tree search in situ (BASELINE config 5 per-GPU shape): K trees x `iterations` search iterations
per environment step on the synthetic `_per` environment, one leaf batch per iteration, one PER-DQN update
per step.  Prints one JSON line with the time split between host tree logic / environment and the device.
    python3 tools/tree_bench.py [K=512] [steps=3] [iterations=50] [per|batch|ptg|ptg0]
`per` steps K Python environments one by one (the reference's protocol); `batch` uses the batched-environment
protocol (one vectorised step per iteration) on the hash environment; `ptg` runs the real PtG environment arithmetic
(mcts_q_amd/ptgenv.py, synthetic tables of the real sizes) through the C lock-step driver with the two-block
host / device overlap; `ptg0` the same with the kernel reading the pinned staging area directly (no H2D copy)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mcts_q_amd.dqn import DQNModel
from mcts_q_amd.mcts import LockstepAgent
from mcts_q_amd.synthetic_env import FakeBatchEnv, FakePerEnv

K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
mode = sys.argv[4] if len(sys.argv) > 4 else "per"
dqn = DQNModel(1, 6, 2, 5, 64, 2, 128, buffer_capacity=1_000_000, batch_size=1024, gamma=0.99, lr=1e-3,
               price_encoder_type="gru", process_encoder_type="gru", gas_eua_encoder_type="gru",
               learning_starts=1024, seed=3654)
if mode in ("ptg", "ptg0"):
    from mcts_q_amd.ptgenv import PtGBatchEnv
    from mcts_q_amd.synthetic_env import synthetic_ptg_kwargs
    benv = PtGBatchEnv(synthetic_ptg_kwargs(0, scale=1.0, train_days=46, env_values=dict(eps_len_d=40)), seed=1)
    agent = LockstepAgent(None, dqn, iterations=iters, maximum_depth=42, c_base=19652, c_init=1.25,
                          batch_env=benv, n_trees=K, zero_copy=(mode == "ptg0"),
                          blocks=int(os.environ.get("MCTSQ_BLOCKS", "2")))
elif mode == "batch":
    agent = LockstepAgent(None, dqn, iterations=iters, maximum_depth=42, c_base=19652, c_init=1.25,
                          batch_env=FakeBatchEnv(), n_trees=K)
else:
    agent = LockstepAgent([FakePerEnv(i) for i in range(K)], dqn, iterations=iters, maximum_depth=42, c_base=19652,
                          c_init=1.25)
t_eval = [0.0]
inner = agent.search.evaluate
def timed(rows):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = inner(rows)
    torch.cuda.synchronize(); t_eval[0] += time.perf_counter() - t0
    return r
if mode not in ("ptg", "ptg0"):
    agent.search.evaluate = timed
t_env = [0.0]
class Timed:
    def __init__(self, e): self.e = e
    def step(self, a):
        t0 = time.perf_counter(); r = self.e.step(a); t_env[0] += time.perf_counter() - t0; return r
    def step_batch(self, s, a):
        t0 = time.perf_counter(); r = self.e.step_batch(s, a); t_env[0] += time.perf_counter() - t0; return r
    def __getattr__(self, k): return getattr(self.e, k)
if mode in ("ptg", "ptg0"):
    pass                                               # the C driver calls the environment itself
elif mode == "batch":
    agent.batch_env = agent.search.batch_env = Timed(agent.batch_env)
else:
    agent.envs[:] = [Timed(e) for e in agent.envs]
agent.step()                                           # warm-up
t_eval[0] = t_env[0] = 0.0; agent.leaf_evals = 0
phases = {}
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); phases[tag] = phases.get(tag, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
if mode in ("ptg", "ptg0"):
    agent.search.prof = dict(wait=0.0, backup=0.0, expand=0.0, launch=0.0)
    wrap(agent.search, "_retain", "retain"); wrap(agent.search, "_evaluate_nodes", "reevaluate_roots_and_kept")
    wrap(agent.search, "_search_fast", "iterations"); wrap(dqn.replay_buffer.dev, "push_rows", "replay_push")
    wrap(dqn, "update", "learner_step"); wrap(agent.batch_env, "step_batch", "root_env_step")
    wrap(dqn, "update_target_network", "polyak")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    agent.step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"mode": mode, "trees": K, "iterations": iters, "env_steps": steps, "s_per_env_step": dt / steps,
                  "leaf_evals_per_s_in_situ": agent.leaf_evals / dt, "tree_env_steps_per_s": K * steps / dt,
                  "frac_device_eval_incl_copies": t_eval[0] / dt, "frac_synthetic_env": t_env[0] / dt,
                  "frac_host_tree_logic_and_learner": 1 - (t_eval[0] + t_env[0]) / dt,
                  "replay_entries": len(dqn.replay_buffer),
                  "c_driver_split_s_per_env_step": ({k: v / steps for k, v in agent.search.prof.items()}
                                                    if agent.search.prof else None),
                  "host_phases_s_per_env_step": {k: v / steps for k, v in phases.items()}}))
