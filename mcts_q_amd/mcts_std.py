"""Non-`_per` flavour of the MCTS_Q agent: host-side mirror of the reference's `MCTSQ` / `MCTSQConfiguration` in
/root/reference/src/mctsq_config_mcts.py:32-360, 438-532 (the module `mctsq_main.py:27` imports).

What differs from the `_per` agent (mcts.py), per the reference's text:
  * the environment is an opaque object that is deep-copied for the root and for every expansion (:168, :243) and
    stepped with a bare action; nodes read `env.get_obs()` and `env.meth_state` (:284, :412);
  * the network is the time-feature variant (`DQNModelStd`: inputs F + 2, mctsq_config_mcts.py:74-82);
  * `learn` pushes only the real transition (no pushes while expanding), stops at `terminated` (:126-127), copies
    the target network every `target_update` steps (:129-130), keeps no sub-tree between steps;
  * `predict(env, deterministic=False)`; the validation roll-out reports the reference's `cum_reward_call`,
    which that file never increments (:136, :150).

The search itself is the lock-step driver with ONE tree (treesearch.LockstepMCTS): the deep-copied environment
rides in the node's state record, so every node is still evaluated exactly once, in one launch per iteration
(the reference evaluates each child of a visited node again, one B = 1 call per child, :203-210 - with fixed
weights inside predict() those are the same numbers).  Not reproduced: the reference leaves this flavour's
policy network in train mode, so a transformer encoder would apply dropout inside the search; the HIP path
evaluates without dropout."""
from __future__ import annotations

import copy

import numpy as np
import yaml

from .dqn import DQNModelStd
from .mcts import MCTSQConfiguration as _PerConfiguration
from .mcts import _device_evaluator
from .treesearch import LockstepMCTS


class MCTSQConfiguration(_PerConfiguration):
    """config/config_mctsq.yaml -> attributes and the run-id string; this flavour's identifier has no
    `Softupdate factor` entry (mctsq_config_mcts.py:460-476)."""
    HYPER = tuple(h for h in _PerConfiguration.HYPER if h[2] != 'tau')


def _record(env):
    """The node record the tree driver reads (`Meth_State` for the legal-action mask, `obs_norm` for the leaf row)
    with the environment copy it belongs to."""
    return {'Meth_State': getattr(env, 'meth_state', None), 'obs_norm': env.get_obs(), 'env': env}


def _step_copy(_tree, action, record):
    """mctsq_config_mcts.py:243-248: expand = deep copy of the node's environment, stepped by the action."""
    env = copy.deepcopy(record['env'])
    obs, reward, terminated, truncated, info = env.step(action)
    out = dict(info) if isinstance(info, dict) else {}
    out['state_c'] = _record(env)
    return obs, reward, terminated, truncated, out


class MCTSQ:
    """Drop-in for mctsq_config_mcts.MCTSQ: `MCTSQ(env, seed, config, tb_log)`."""

    def __init__(self, env, seed: int, config: MCTSQConfiguration | None = None, tb_log: str | None = None,
                 device="cuda:0") -> None:
        if config is not None:
            self.__dict__.update(config.__dict__)
        else:
            with open("config/config_mctsq.yaml", "r", encoding="utf-8") as f:
                self.__dict__.update(yaml.safe_load(f))
        self.env, self.step = env, None
        self.tb_log, self.writer = tb_log, None
        if tb_log is not None:
            try:                                        # tensorboard is optional here (:61-62)
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(log_dir=tb_log)
            except Exception as e:                      # noqa: BLE001
                print(f"    > TensorBoard logging disabled: {e}")
        self.eval_processes = []
        temporal_encodings = 2                          # sin / cos columns (:74)
        self.dqn = DQNModelStd(
            el_input_dim=1 + temporal_encodings, process_input_dim=6 + temporal_encodings,
            gas_eua_input_dim=2 + temporal_encodings, action_dim=env.action_space.n, embed_dim=self.embed_dim,
            hidden_layers=self.hidden_layers, hidden_units=self.hidden_units, buffer_capacity=self.buffer_size,
            batch_size=self.batch_size, gamma=self.discount_factor, lr=self.learning_rate,
            price_encoder_type=self.price_encoder_type, process_encoder_type=self.process_encoder_type,
            gas_eua_encoder_type=self.gas_eua_encoder_type, activation=self.activation,
            learning_starts=self.learning_starts, seed=seed, device=device,
            seq_lens=self._window_lengths(env))
        self.action_type, self.deterministic = "discrete", False
        self._search = LockstepMCTS(1, _device_evaluator(self.dqn.policy_net), self.iterations, self.maximum_depth,
                                    self.c_base, self.c_init, rngs=None)

    def _window_lengths(self, env):
        """The three window lengths, read off the environment's observation (the reference's modules are
        shape-agnostic; the kernels are built per length)."""
        obs = env.get_obs() if hasattr(env, "get_obs") else None
        if obs is None:
            return (25, self.seq_length, 3)
        return (len(obs["Elec_Price"]), len(obs["T_CAT"]), len(obs["Gas_Price"]))

    def predict(self, env, deterministic: bool = False) -> int:
        """mctsq_config_mcts.py:162-185: a fresh tree over a deep copy of `env`."""
        self.deterministic = deterministic
        self._search.reset()
        a = self._search.predict([_record(copy.deepcopy(env))], _step_copy, train=False)
        action = int(a[0])
        self._search.reset()                            # `del root_node; gc.collect()` (:182-183): drop the env copies
        return action

    def learn(self, total_timesteps: int, callback) -> None:
        """mctsq_config_mcts.py:103-160."""
        state, _ = self.env.reset()
        for self.step in range(total_timesteps):
            action = self.predict(self.env)
            next_state, reward, terminated, _, _ = self.env.step(action)
            self.dqn.replay_buffer.push(state, action, reward, next_state, terminated)
            loss = self.dqn.update()
            if self.writer is not None and loss is not None:
                self.writer.add_scalar("Training/Loss", float(loss), global_step=self.step)
            state = next_state
            if terminated:
                break
            if self.step % self.target_update == 0 and self.step != 0:
                self.dqn.update_target_network()
            if callback is not None and self.step % callback.val_steps == 0 and self.step != 0:
                callback.env.reset()
                cum_reward_call = 0
                for _ in range(callback.env.eps_sim_steps):
                    action_call = self.predict(callback.env)
                    _, _, terminated_call, _, _ = callback.env.step(action_call)
                    if terminated_call:
                        break
                print(f"   >>Step: {self.step} - Cumulative Reward: {cum_reward_call}")
                if self.writer is not None:
                    self.writer.add_scalar("Validation/CumulativeReward", cum_reward_call, global_step=self.step)
        if self.writer is not None:
            self.writer.close()

    def save(self, filepath: str) -> None:
        self.dqn.save(filepath)

    def load(self, filepath: str) -> None:
        self.dqn.load(filepath)

    def test(self, env_config, eps_sim_steps_test: int):
        """mctsq_config_mcts.py:318-359: roll-out on the test environment into the statistics dict."""
        self.env.reset()
        stats = np.zeros((eps_sim_steps_test, len(env_config.stats_names)))
        codes = {'standby': 0, 'cooldown': 1, 'startup': 2, 'partial_load': 3}
        info = {}
        for i in range(eps_sim_steps_test):
            action = self.predict(self.env)
            _, _, terminated, _, info = self.env.step(action)
            if terminated:
                break
            for j, key in enumerate(info):
                if j < 24:
                    stats[i, j] = codes.get(info[key], 4) if key == 'Meth_Action' else info[key]
        print(f"   >>Cumulative Reward {info['cum_reward']}")
        return {name: stats[:eps_sim_steps_test, m] for m, name in enumerate(env_config.stats_names)}
