"""Lock-step tree driver with the HIP network behind it (mcts_q_amd/treesearch.py, mcts.py) against the
per-node reference search of oracle/mcts_ref.py running the torch-CPU oracle network with the same weights.
Q differs between the two by fp32 rounding (~1e-7), which can flip a PUCT near-tie, so the bar is: root
visit distributions identical on (nearly) every tree, never a different most-visited action where the
reference's top two visit counts differ by more than one."""
import os
import random

import numpy as np
import pytest
import torch

from mcts_q_amd import NetSpec
from mcts_q_amd.synthetic_env import FakeGymEnv, FakePerEnv
from oracle import mcts_ref, torch_ref

pytestmark = pytest.mark.gpu
CFG = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
MCFG = dict(iterations=30, c_init=1.25, c_base=19652, maximum_depth=42, learning_rate=1e-3, discount_factor=0.99,
            buffer_size=5000, batch_size=32, hidden_layers=2, hidden_units=128, activation='relu', tau=0.005,
            seq_length=10, embed_dim=64, price_encoder_type='gru', process_encoder_type='gru',
            gas_eua_encoder_type='gru', learning_starts=40, tree_remain_interval=5, seq_step=30)
ENVCFG = dict(price_ahead=13, price_past=12, time_step_op=2)


def test_lockstep_hip_search_matches_reference_search():
    from mcts_q_amd.leafbatch import split_rows
    from mcts_q_amd.qnet import DeviceQNet
    from mcts_q_amd.treesearch import LockstepMCTS
    torch.manual_seed(0)
    ref_net = torch_ref.RefQNet(CFG).eval()
    sd = {k: v.detach().numpy() for k, v in ref_net.state_dict().items()}
    net = DeviceQNet(NetSpec.from_cfg(CFG)).load_state_dict(sd)
    launches = [0]

    def evaluate(rows):
        launches[0] += 1
        out = net(*split_rows(torch.from_numpy(np.ascontiguousarray(rows)).cuda()))
        return out["vmax"].cpu().numpy(), out["prior"].cpu().numpy()

    def q_fn(p, q, g):
        with torch.no_grad():
            return ref_net(p, q, g)

    K, iters = 48, 30
    envs = [FakePerEnv(seed=i) for i in range(K)]
    drv = LockstepMCTS(K, evaluate, iters, 42, 19652, 1.25, rngs=[random.Random(7 + i) for i in range(K)])
    scs = [e.reset()[1]["state_c"] for e in envs]
    acts = drv.predict(scs, lambda i, a, sc: envs[i].step([a, sc]), train=False)
    assert launches[0] == iters + 1                      # one leaf batch per iteration (+ the roots)
    same, bad = 0, 0
    for i in range(K):
        ref = mcts_ref.RefSearch(q_fn, envs[i], iters, 42, 19652, 1.25, random.Random(7 + i))
        a_ref = ref.predict(scs[i], "eval")
        root = drv.root[i]
        got = [(int(drv.action[c]), int(drv.visits[c])) for c in drv.child[root][:drv.nchild[root]]]
        exp = [(c.action, c.visits) for c in ref.root.children]
        same += got == exp
        v = sorted((c.visits for c in ref.root.children), reverse=True)
        if acts[i] != a_ref and (len(v) < 2 or v[0] - v[1] > 1):
            bad += 1
    print(f"identical root visit distributions: {same}/{K}")
    assert same >= int(0.9 * K) and bad == 0
    # With the SAME network on both sides - the per-node search calling the HIP forward one state at a time, as the
    # reference calls its module (mcts_per.py:291-296, 367-370) - nothing is left to rounding: the forward is row-wise,
    # so batch composition cannot change a value, and every tree must come out identical, node for node.
    def q_hip(p, q, g):
        return net(p.cuda(), q.cuda(), g.cuda())["q"].cpu()
    for i in range(K):
        ref = mcts_ref.RefSearch(q_hip, envs[i], iters, 42, 19652, 1.25, random.Random(7 + i))
        assert ref.predict(scs[i], "eval") == acts[i]
        stack = [(ref.root, int(drv.root[i]))]
        while stack:
            rn, dn = stack.pop()
            assert (rn.visits, rn.total_value) == (int(drv.visits[dn]), float(drv.total[dn]))
            kids = [int(c) for c in drv.child[dn][:drv.nchild[dn]]]
            assert [c.action for c in rn.children] == [int(drv.action[c]) for c in kids]
            stack.extend(zip(rn.children, kids))


def test_mctsq_learn_predict_save_load(tmp_path):
    from mcts_q_amd.mcts import MCTSQ, MCTSQConfiguration

    class Cb:
        env, val_steps = FakeGymEnv(99, eps_sim_steps=3), 25

    agent = MCTSQ(FakeGymEnv(1, horizon=20), seed=3, config=MCTSQConfiguration(values=dict(MCFG)), env_config=ENVCFG)
    agent.learn(40, Cb())
    n = len(agent.dqn.replay_buffer)
    assert n >= 40                                      # 1 + #expansions pushes per step (mcts_per.py:148, 340)
    loss = agent.dqn.update()
    assert loss is not None and np.isfinite(float(loss))
    path = os.path.join(tmp_path, "ckpt.pt")
    agent.save(path)
    other = MCTSQ(FakeGymEnv(1), seed=4, config=MCTSQConfiguration(values=dict(MCFG)), env_config=ENVCFG)
    other.load(path)
    sc = FakeGymEnv(5).reset()[1]["state_c"]
    random.seed(11); a1 = agent.predict(sc, "eval")
    random.seed(11); a2 = other.predict(sc, "eval")
    assert a1 == a2 and 0 <= a1 < 5


def test_non_per_agent_searches_a_deep_copied_environment(tmp_path):
    """src/mctsq_config_mcts.py flavour (/root/reference/src/mctsq_config_mcts.py:103-185): the search runs on deep
    copies (the live environment is untouched by predict), the tree equals the per-node restatement driven through
    the same copies with the same network, learn() pushes ONE transition per step, copies the target network every
    `target_update` steps and stops at `terminated`."""
    from mcts_q_amd.synthetic_env import FakeStdGymEnv
    from src.mctsq_config_mcts import MCTSQ, MCTSQConfiguration
    cfg = dict(MCFG, target_update=7, learning_starts=32, iterations=20)
    env = FakeStdGymEnv(3, horizon=60)
    agent = MCTSQ(env, seed=5, config=MCTSQConfiguration(values=cfg))
    assert agent.dqn.policy_net.spec.flavor == "std" and agent.dqn.el_input_dim == 3
    before = dict(env._sc)
    random.seed(2)
    a = agent.predict(env)
    assert env._sc == before and 0 <= a < 5 and agent._search.n == 0
    # the same search, node by node, over state records that carry their environment copy
    import copy
    from mcts_q_amd import mcts_std

    class CopyEnv:                                            # `_per` protocol on top of the deep copies
        def step(self, arg):
            return mcts_std._step_copy(0, arg[0], arg[1])

    def q_fn(p, q, g):
        return agent.dqn.policy_net(p, q, g)
    random.seed(2)
    ref = mcts_ref.RefSearch(q_fn, CopyEnv(), 20, 42, 19652, 1.25, random)
    assert ref.predict(mcts_std._record(copy.deepcopy(env)), "eval") == a

    class Cb:
        env, val_steps = FakeStdGymEnv(99, eps_sim_steps=2), 30
    tgt0 = agent.dqn.target_net.state_dict()["fc_layers.0.weight"].cpu().numpy().copy()
    agent.learn(100, Cb())
    assert agent.step == 59 and len(agent.dqn.replay_buffer) == 60          # broke at `terminated`, one push per step
    pol = agent.dqn.policy_net.state_dict()["fc_layers.0.weight"].cpu().numpy().copy()
    tgt = agent.dqn.target_net.state_dict()["fc_layers.0.weight"].cpu().numpy().copy()
    assert not np.array_equal(tgt, tgt0)                                  # hard copies happened (steps 7, 14, ...)
    assert not np.array_equal(tgt, pol)                                   # ... and updates after the last one
    path = os.path.join(tmp_path, "std.pt")
    agent.save(path)
    other = MCTSQ(FakeStdGymEnv(1), seed=9, config=MCTSQConfiguration(values=cfg))
    other.load(path)
    e2 = FakeStdGymEnv(8)
    random.seed(4); a1 = agent.predict(e2)
    random.seed(4); a2 = other.predict(e2)
    assert a1 == a2


def test_lockstep_agent_many_trees_one_learner():
    from mcts_q_amd.dqn import DQNModel
    from mcts_q_amd.mcts import LockstepAgent
    dqn = DQNModel(1, 6, 2, 5, 64, 2, 128, buffer_capacity=20000, batch_size=64, gamma=0.99, lr=1e-3,
                   price_encoder_type="gru", process_encoder_type="gru", gas_eua_encoder_type="gru",
                   learning_starts=64, seed=1)
    K = 32
    agent = LockstepAgent([FakePerEnv(i, horizon=6) for i in range(K)], dqn, iterations=10, maximum_depth=42,
                          c_base=19652, c_init=1.25, tree_remain_interval=3)
    losses = []
    for _ in range(8):
        acts, rew, loss = agent.step()
        assert acts.shape == (K,) and ((0 <= acts) & (acts < 5)).all()
        if loss is not None:
            losses.append(float(loss))
    # root transition + at most one expansion per iteration and tree (terminal leaves are not expanded)
    assert 8 * K < len(dqn.replay_buffer) <= 8 * K * (1 + 10)
    assert losses and np.isfinite(losses).all()


def test_lockstep_agent_batched_environment():
    from mcts_q_amd.dqn import DQNModel
    from mcts_q_amd.mcts import LockstepAgent
    from mcts_q_amd.synthetic_env import FakeBatchEnv
    dqn = DQNModel(1, 6, 2, 5, 64, 2, 128, buffer_capacity=50000, batch_size=64, gamma=0.99, lr=1e-3,
                   price_encoder_type="gru", process_encoder_type="gru", gas_eua_encoder_type="gru",
                   learning_starts=64, seed=1)
    K = 64
    agent = LockstepAgent(None, dqn, iterations=10, maximum_depth=42, c_base=19652, c_init=1.25,
                          tree_remain_interval=3, batch_env=FakeBatchEnv(horizon=5), n_trees=K)
    losses = []
    for _ in range(7):
        acts, rew, loss = agent.step()
        assert acts.shape == (K,) and ((0 <= acts) & (acts < 5)).all()
        if loss is not None:
            losses.append(float(loss))
    assert 7 * K < len(dqn.replay_buffer) <= 7 * K * 11
    assert losses and np.isfinite(losses).all()


@pytest.mark.parametrize("zero_copy,events", [(False, False), (True, False), (True, True)])
def test_c_driver_with_the_device_evaluator_equals_the_numpy_driver(zero_copy, events, monkeypatch):
    """SURVEY 8f-2 on the device: the C lock-step driver over the vectorised PtG environment, leaves staged in pinned
    memory and evaluated by one forward launch per block of trees (outputs written to pinned host memory by the kernel),
    the two blocks alternating between host and device - against the numpy driver calling the plain device evaluator
    with the same counter-based draws.  The forward is row-wise, so trees, visit counts, values and replay pushes are
    identical bit for bit; the agent-level loop (search, environment step, replay, learner step) runs on top of it."""
    from mcts_q_amd.dqn import DQNModel
    from mcts_q_amd.mcts import DeviceLeafEvaluator, LockstepAgent
    from mcts_q_amd.ptgenv import PtGBatchEnv
    from mcts_q_amd.synthetic_env import synthetic_ptg_kwargs
    from mcts_q_amd.treesearch import LockstepMCTS
    # completion of a leaf batch: a pinned word the stream writes (default) or an event per batch (MCTSQ_SEARCH_EVENTS)
    if events:
        monkeypatch.setenv("MCTSQ_SEARCH_EVENTS", "1")
    else:
        monkeypatch.delenv("MCTSQ_SEARCH_EVENTS", raising=False)
    kw = synthetic_ptg_kwargs(5, scale=0.2)
    dqn = DQNModel(1, 6, 2, 5, 64, 2, 128, buffer_capacity=20000, batch_size=64, gamma=0.99, lr=1e-3,
                   price_encoder_type="gru", process_encoder_type="gru", gas_eua_encoder_type="gru", learning_starts=64,
                   seed=11)
    K, iters = 96, 12
    runs = []
    for fast in (True, False):
        env = PtGBatchEnv(kw, seed=1)
        ev = DeviceLeafEvaluator(dqn.policy_net, K // 2, 2, zero_copy=zero_copy)
        drv = LockstepMCTS(K, ev, iters, 8, 19652, 1.25, batch_env=env, seed=3)
        assert drv.fast
        drv.fast = fast
        states = env.reset(np.arange(K))
        warm = np.random.RandomState(0)
        for _ in range(4):
            states = env.step_batch(states, warm.randint(0, 5, K), noise=np.zeros(K))[0]
        rec = []
        for step in range(3):
            pushed = []
            a = drv.predict(states, train=True, reuse=np.ones(K, dtype=bool),
                            push_rows=lambda *x: pushed.append([np.array(v, dtype=np.float64) for v in x]))
            n = drv.n
            rec.append([a.copy(), np.array(n), drv.visits[:n].copy(), drv.total[:n].copy(), drv.parent[:n].copy(),
                        drv.vmax[:n].copy(), drv.prior_a[:n].copy()] +
                       [np.concatenate([p[q] for p in pushed]) for q in range(5)])
            states = env.step_batch(states, a, noise=np.zeros(K))[0]
            drv.advance(a, keep=np.ones(K, dtype=bool))
        runs.append(rec)
    for x, y in zip(*runs):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)
    env = PtGBatchEnv(kw, seed=1)
    agent = LockstepAgent(None, dqn, iterations=iters, maximum_depth=8, c_base=19652, c_init=1.25, batch_env=env,
                          n_trees=K, zero_copy=zero_copy)
    loss = None
    for _ in range(3):
        actions, rewards, loss = agent.step()
    assert len(dqn.replay_buffer) >= 3 * K * (iters + 1) - 3 * K and loss is not None and np.isfinite(float(loss))
    assert agent.leaf_evals >= 3 * K * iters
