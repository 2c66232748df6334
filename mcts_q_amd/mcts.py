"""MCTS_Q agent: host-side mirror of the reference's `MCTSQ` and `MCTSQConfiguration`
(/root/reference/src/mctsq_config_mcts_per.py:33-466, 555-651) on top of the lock-step tree driver
(treesearch.py) and the HIP learner (dqn.py).  Same constructor arguments, attributes and methods
(`learn`, `predict`, `save`, `load`, `test`), same cadence of replay pushes / updates / Polyak target
syncs / sub-tree reuse in `learn`; the network calls go through `DQNModel.policy_net.evaluate_leaves`
(one launch per search iteration) instead of a scripted torch module.

`LockstepAgent` is the many-environment form the north star asks for: K environments / trees per process
advance one search iteration together, so each iteration is ONE leaf batch of K states (BASELINE config
5: 512 trees per GPU).  It has no counterpart in the reference (one tree, B = 1 calls)."""
from __future__ import annotations

import random

import numpy as np
import torch
import yaml

from .dqn import DQNModel
from .leafbatch import split_rows
from .treesearch import LockstepMCTS


class MCTSQConfiguration:
    """config/config_mctsq.yaml -> attributes, and the run-id string (mcts_per.py:555-651)."""

    HYPER = (('Iterations', "_it", 'iterations'), ('PUCT Initial Exploration', "_pu", 'c_init'),
             ('PUCT Exploration Increase', "_pi", 'c_base'), ('Max Tree depth', "_md", 'maximum_depth'),
             ('Learning rate', "_al", 'learning_rate'), ('Discount factor', "_ga", 'discount_factor'),
             ('Replay buffer size', "_rb", 'buffer_size'), ('Batch size', "_bs", 'batch_size'),
             ('Hidden layers', "_hl", 'hidden_layers'), ('Hidden units', "_hu", 'hidden_units'),
             ('Activation function', "_af", 'activation'), ('Softupdate factor', "_ta", 'tau'),
             ('Process data sequence', "_sq", 'seq_length'), ('Embedding dimensions', "_ed", 'embed_dim'),
             ('Price encoder type', "_ee", 'price_encoder_type'), ('Process encoder type', "_pe", 'process_encoder_type'),
             ('Gas EUA encoder type', "_ge", 'gas_eua_encoder_type'))

    def __init__(self, path: str = "config/config_mctsq.yaml", values: dict | None = None) -> None:
        if values is None:
            with open(path, "r", encoding="utf-8") as f:
                values = yaml.safe_load(f)
        self.__dict__.update(values)
        self.rl_alg_hyp = values
        self.str_alg = ""
        self.hyper = {name: {'abb': abb, 'var': var} for name, abb, var in self.HYPER}

    def get_hyper(self) -> str:
        print("    > MCTS_Q algorithm : >>> <<<")
        for name, _, _ in self.HYPER:
            self.hyp_print(name)
        print(' ')
        return self.str_alg

    def hyp_print(self, hyp_name: str) -> None:
        if hyp_name not in self.hyper:
            raise ValueError(f"Specified hyperparameter ({hyp_name}) is not part of the implemented settings!")
        abb, val = self.hyper[hyp_name]['abb'], self.rl_alg_hyp[self.hyper[hyp_name]['var']]
        n = len(hyp_name)
        tabs = "" if n > 32 else "\t" if n > 22 else "\t\t" if n > 15 else "\t\t\t"
        print(f"         {hyp_name} ({abb}):{tabs} {val}")
        self.str_alg += abb + str(val)


def _device_evaluator(policy_net):
    """rows fp32 [n, 91] (host) -> (vmax [n], prior [n, A]) through ONE launch of the HIP forward."""
    dev = policy_net.device
    fwd = policy_net.evaluate_leaves if hasattr(policy_net, "evaluate_leaves") else policy_net.forward   # DeviceQNet

    def evaluate(rows):
        t = torch.from_numpy(np.ascontiguousarray(rows)).to(dev, non_blocking=True)
        out = fwd(*split_rows(t))
        return out["vmax"].cpu().numpy(), out["prior"].cpu().numpy()
    return evaluate


class DeviceLeafEvaluator:
    """Leaf evaluation for the C lock-step driver (treesearch.LockstepMCTS._search_fast): `halves` staging areas of
    `cap` leaves each in pinned host memory, laid out as the forward kernel's three inputs (price [cap,25] | process
    [cap,60] | gas [cap,6]), which the driver fills in place.  launch(h, n) enqueues ONE copy of the area to the device
    and ONE forward launch whose epilogue stores max_a Q and softmax(Q) straight into pinned host memory (no D2H copy
    of its own), then records an event; wait(h) blocks on that event.  With zero_copy the kernel reads the pinned
    staging area itself and the H2D copy goes away too.  Called as a function it is the plain evaluator
    (rows -> vmax, prior) the generic driver uses."""

    def __init__(self, policy_net, cap: int, halves: int = 2, zero_copy: bool = False):
        import ctypes as C
        from ._lib import lib, ptr
        self._C, self._lib, self._ptr = C, lib, ptr
        self.net = policy_net.qnet if hasattr(policy_net, "qnet") else policy_net    # DeviceQNet
        self.plain = _device_evaluator(policy_net)
        self.cap, self.halves, self.zero_copy = int(cap), int(halves), bool(zero_copy)
        net, dev = self.net, self.net.device
        A = net.spec.action_dim
        self.n_act = A
        cap = self.cap
        self._hin = [torch.zeros(cap * 91, dtype=torch.float32, pin_memory=True) for _ in range(halves)]
        self._din = [torch.zeros(cap * 91, dtype=torch.float32, device=dev) for _ in range(halves)]
        self._hout = [torch.zeros(cap * (1 + A), dtype=torch.float32, pin_memory=True) for _ in range(halves)]
        self._q = [torch.empty(cap, A, dtype=torch.float32, device=dev) for _ in range(halves)]
        self._arg = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(halves)]
        self._ev = [torch.cuda.Event() for _ in range(halves)]
        # one stream per block: a block's leaf batch covers a few CUs and takes the kernel's full latency whatever its
        # size, so the blocks' launches are worth overlapping on the device, not only with the host
        self._streams = [torch.cuda.Stream(device=dev) for _ in range(halves)]
        self._ready = torch.cuda.Event()
        nbytes = lib.check(lib.mctsq_q_forward_workspace_bytes(C.byref(net.desc), cap))
        self._ws = [torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev) for _ in range(halves)]
        self._args = None

    def __call__(self, rows):
        return self.plain(rows)

    def begin(self):
        """Bind the pointers of one search: the weights do not change inside predict()."""
        net, C, ptr, cap = self.net, self._C, self._ptr, self.cap
        self._stream = torch.cuda.current_stream(net.device)
        prep = ptr(net._prepared())
        self._ready.record(self._stream)              # weights (and their prepared fragments) as of the caller's stream
        self._args = []
        for h in range(self.halves):
            self._streams[h].wait_event(self._ready)
            st = self._streams[h].cuda_stream
            src = self._hin[h] if self.zero_copy else self._din[h]
            base = src.data_ptr()
            self._args.append((C.byref(net.desc), ptr(net.flat), prep, base, base + 4 * cap * 25, base + 4 * cap * 85,
                               ptr(self._q[h]), self._hout[h].data_ptr(), ptr(self._arg[h]),
                               self._hout[h].data_ptr() + 4 * cap, ptr(self._ws[h]), self._ws[h].numel(), st, None))

    def c_struct(self):
        """mctsq_leaf_eval for mctsq_lockstep_search: the forward entry point and the three HIP runtime calls as
        plain addresses (libmctsq_tree.so links no GPU runtime), buffers and events of both blocks."""
        from .hipevents import HipEvent, _rt
        from .treesearch import LeafEval
        C = self._C
        if not hasattr(self, "_hipev"):
            import os
            self._hipev = [HipEvent(timing=bool(os.environ.get("MCTSQ_EVT_TIMING"))) for _ in range(self.halves)]
        hip, dll = _rt(), self._lib._load()
        le = LeafEval()
        addr = lambda fn: C.cast(fn, C.c_void_p).value
        le.forward, le.event_record = addr(dll.mctsq_q_forward_prepared), addr(hip.hipEventRecord)
        le.event_sync, le.memcpy_async = addr(hip.hipEventSynchronize), addr(hip.hipMemcpyAsync)
        a = self._args[0]
        le.net, le.params, le.prep = C.addressof(self.net.desc), a[1], a[2]
        le.halves, le.cap, le.ws_bytes, le.n_act = self.halves, self.cap, self._ws[0].numel(), self.n_act
        for h in range(self.halves):
            le.stream[h] = self._streams[h].cuda_stream
            le.stage[h] = self._hin[h].data_ptr()
            le.dev_in[h] = None if self.zero_copy else self._din[h].data_ptr()
            le.q[h], le.argmax[h], le.ws[h] = self._q[h].data_ptr(), self._arg[h].data_ptr(), self._ws[h].data_ptr()
            le.vmax[h] = self._hout[h].data_ptr()
            le.prior[h] = self._hout[h].data_ptr() + 4 * self.cap
            le.event[h] = self._hipev[h].ptr
        # completion through a pinned word the stream writes behind each leaf batch (hipStreamWriteValue32) and the
        # search loop reads, instead of an event record + hipEventSynchronize per batch (MCTSQ_SEARCH_EVENTS=1: events)
        import os
        wr = getattr(hip, "hipStreamWriteValue32", None)
        if wr is not None and not os.environ.get("MCTSQ_SEARCH_EVENTS"):
            if not hasattr(self, "_flags"):
                self._flags = torch.zeros(16 * self.halves, dtype=torch.int32, pin_memory=True)   # one cache line each
            le.stream_write32 = addr(wr)
            for h in range(self.halves):
                le.done_flag[h] = self._flags.data_ptr() + 64 * h
        self._le = le                                                   # keeps the desc byref alive with it
        return le

    def abort(self):
        """After a failed search: wait for whatever the blocks' streams still hold, then reset the completion words
        (a late write of an abandoned batch would desynchronise the next search's sequence numbers)."""
        for st in self._streams:
            st.synchronize()
        if hasattr(self, "_flags"):
            self._flags.zero_()

    def stage_ptr(self, h):
        return self._hin[h].data_ptr()

    def vmax_ptr(self, h):
        return self._hout[h].data_ptr()

    def prior_ptr(self, h):
        return self._hout[h].data_ptr() + 4 * self.cap

    def launch(self, h, n):
        a = self._args[h]
        with torch.cuda.stream(self._streams[h]):
            if not self.zero_copy:
                self._din[h].copy_(self._hin[h], non_blocking=True)
            self._lib.check(self._lib.mctsq_q_forward_prepared(a[0], a[1], a[2], a[3], a[4], a[5], n, a[6], a[7], a[8],
                                                               a[9], a[10], a[11], a[12], a[13]))
            self._ev[h].record(self._streams[h])

    def wait(self, h):
        self._ev[h].synchronize()


class MCTSQ:
    """Drop-in for the reference's MCTSQ (mcts_per.py:33-466): one environment, one tree."""

    def __init__(self, env, seed: int, config: MCTSQConfiguration | None = None, log_path: str | None = None,
                 tb_log: str | None = None, env_config: dict | None = None, device="cuda:0") -> None:
        if config is not None:
            self.__dict__.update(config.__dict__)
        else:
            with open("config/config_mctsq.yaml", "r", encoding="utf-8") as f:
                self.__dict__.update(yaml.safe_load(f))
        self.env, self.step, self.seed = env, None, seed
        self.tb_log, self.log_path, self.writer = tb_log, log_path, None
        if tb_log is not None:
            try:                                        # tensorboard is optional here (mcts_per.py:66-67)
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(log_dir=tb_log)
            except Exception as e:                      # noqa: BLE001
                print(f"    > TensorBoard logging disabled: {e}")
        if env_config is None:
            with open("config/config_env.yaml", "r", encoding="utf-8") as f:
                env_config = yaml.safe_load(f)
        seq_len_price = env_config['price_ahead'] + env_config['price_past']          # :84
        self.dqn = DQNModel(
            el_input_dim=1, process_input_dim=6, gas_eua_input_dim=2, action_dim=env.action_space.n,
            embed_dim=self.embed_dim, hidden_layers=self.hidden_layers, hidden_units=self.hidden_units,
            buffer_capacity=self.buffer_size, batch_size=self.batch_size, gamma=self.discount_factor,
            lr=self.learning_rate, price_encoder_type=self.price_encoder_type,
            process_encoder_type=self.process_encoder_type, gas_eua_encoder_type=self.gas_eua_encoder_type,
            activation=self.activation, learning_starts=self.learning_starts, seed=seed, device=device,
            seq_lens=(seq_len_price, self.seq_length, 3))
        self.dqn.policy_net.eval()                      # :114 (scripting at :115 has no meaning for the HIP path)
        self.action_type, self.deterministic = "discrete", False
        self.callback, self.callback_run = None, False
        self.tree_remain, self.train_eval, self.treedepth = False, "train", 0
        self._search = LockstepMCTS(1, _device_evaluator(self.dqn.policy_net), self.iterations, self.maximum_depth,
                                    self.c_base, self.c_init, rngs=None)

    @property
    def root_node(self):
        r = self._search.root[0]
        return None if r < 0 else int(r)

    def predict(self, state_c, train_eval: str = "train", deterministic: bool = False) -> int:
        """mcts_per.py:220-258."""
        self.deterministic, self.train_eval = deterministic, train_eval
        env = self.callback.env if self.callback_run else self.env        # :330-335
        train = train_eval == "train" and not self.callback_run
        a = self._search.predict([state_c], lambda i, act, sc: env.step([act, sc]), train=train,
                                 push=self.dqn.replay_buffer.push if train else None,
                                 reuse=[self.tree_remain and train_eval == "train"])
        self.tree_remain = True                                           # :247
        self.treedepth = int(self._search.treedepth[0])
        return int(a[0])

    def learn(self, total_timesteps: int, callback) -> None:
        """mcts_per.py:127-218: predict -> env step -> push -> update -> sub-tree reuse -> Polyak sync."""
        state, info = self.env.reset()
        state_c = info['state_c']
        self.callback = callback
        for self.step in range(total_timesteps):
            action = self.predict(state_c, train_eval="train")
            next_state, reward, terminated, _, info = self.env.step([action, state_c])
            state_c = info['state_c']
            self.dqn.replay_buffer.push(state, action, reward, next_state, terminated)
            loss = self.dqn.update()
            state = next_state
            self._search.advance([action])                                # :153-167
            if self.writer is not None and loss is not None:
                self.writer.add_scalar("Training/Loss", float(loss), global_step=self.step)
                self.writer.add_scalar("Training/TreeDepth", self.treedepth, global_step=self.step)
            if terminated:
                state, info = self.env.reset()
                state_c = info['state_c']
                self.tree_remain = False
            if self.step % self.tree_remain_interval == 0 and self.step != 0:
                self.tree_remain = False
            self.dqn.update_target_network(self.tau)
            if self.callback is not None and self.step % self.callback.val_steps == 0 and self.step != 0:
                self.callback_run = True
                _, info = self.callback.env.reset()
                sc = info['state_c']
                for _ in range(self.callback.env.eps_sim_steps):
                    a = self.predict(sc, train_eval="eval")
                    _, _, term, _, info = self.callback.env.step([a, sc])
                    sc = info['state_c']
                    if term:
                        break
                cum = info['cum_reward']
                print(f"   >>Step: {self.step} - Cumulative Reward: {cum}")
                if self.writer is not None:
                    self.writer.add_scalar("Validation/CumulativeReward", cum, global_step=self.step)
                self.callback_run = False
                self.tree_remain = False
                self._search.reset()                    # the validation search replaced the root (:245-247)
        if self.writer is not None:
            self.writer.close()

    def save(self, filepath: str) -> None:
        self.dqn.save(filepath)

    def load(self, filepath: str) -> None:
        self.dqn.load(filepath)

    def test(self, env_config, eps_sim_steps_test: int):
        """mcts_per.py:424-466: greedy roll-out on the test environment into the statistics dict."""
        _, info = self.env.reset()
        sc = info['state_c']
        stats = np.zeros((eps_sim_steps_test, len(env_config.stats_names)))
        codes = {'standby': 0, 'cooldown': 1, 'startup': 2, 'partial_load': 3}
        for i in range(eps_sim_steps_test):
            action = self.predict(sc, train_eval="eval")
            _, _, terminated, _, info = self.env.step([action, sc])
            sc = info['state_c']
            if terminated:
                break
            for j, key in enumerate(info):
                if j < 24:
                    stats[i, j] = codes.get(info[key], 4) if key == 'Meth_Action' else info[key]
        print(f"   >>Cumulative Reward {info['cum_reward']}")
        return {name: stats[:eps_sim_steps_test, m] for m, name in enumerate(env_config.stats_names)}


class LockstepAgent:
    """K environments and K trees per process sharing ONE learner: every search iteration evaluates the K
    new leaves in one launch; after each environment step the K root transitions and the transitions met
    while expanding go to the replay buffer and the learner takes `updates_per_step` PER-DQN steps
    (data-parallel over ranks when the DQNModel's learner has a process group)."""

    def __init__(self, envs, dqn: DQNModel, iterations: int, maximum_depth: int, c_base: float, c_init: float,
                 tau: float = 0.005, tree_remain_interval: int = 10, seed: int = 0, updates_per_step: int = 1,
                 batch_env=None, n_trees: int | None = None, zero_copy: bool = False, blocks: int = 2):
        """envs: K environments with the `_per` protocol, OR batch_env (+ n_trees): one vectorised environment
        with the batched protocol of treesearch.LockstepMCTS (reset(seeds), observe, step_batch)."""
        self.dqn, self.batch_env = dqn, batch_env
        self.tau, self.tree_remain_interval, self.updates_per_step = tau, tree_remain_interval, updates_per_step
        self.step_count, self.leaf_evals = 0, 0
        if batch_env is not None:
            self.K = int(n_trees)
            if getattr(batch_env, "c_env", None):
                halves = max(1, min(int(blocks), 4)) if self.K >= 64 else 1
                ev = DeviceLeafEvaluator(dqn.policy_net, -(-self.K // halves), halves, zero_copy=zero_copy)
            else:
                ev = _device_evaluator(dqn.policy_net)
            self.search = LockstepMCTS(self.K, ev, iterations, maximum_depth, c_base, c_init, batch_env=batch_env, seed=seed)
            self._episode = np.arange(self.K, dtype=np.int64)
            self.states = batch_env.reset(self._episode + seed * 1000003)
            self.rows, _ = batch_env.observe(self.states)
            self.keep = np.zeros(self.K, dtype=bool)
            return
        self.envs, self.K = list(envs), len(envs)
        self.search = LockstepMCTS(self.K, _device_evaluator(dqn.policy_net), iterations, maximum_depth, c_base,
                                   c_init, rngs=[random.Random(seed * 1000003 + i) for i in range(self.K)])
        res = [e.reset() for e in self.envs]
        self.obs = [r[0] for r in res]
        self.state_c = [r[1]['state_c'] for r in res]
        self.keep = np.zeros(self.K, dtype=bool)

    def _step_batch(self, train: bool):
        K, benv, buf = self.K, self.batch_env, self.dqn.replay_buffer
        n0 = self.search.n

        def push_rows(rows, acts, rew, nrows, done):
            buf.flush()
            buf.dev.push_rows(rows, np.asarray(acts, dtype=np.int32), np.asarray(rew, dtype=np.float32), nrows,
                              np.asarray(done, dtype=np.float32))
        actions = self.search.predict(self.states, train=train, reuse=self.keep, push_rows=push_rows if train else None)
        self.leaf_evals += K * self.search.iterations + (n0 if train else 0) + K
        nst, nrows, rew, term, _, _ = benv.step_batch(self.states, actions)
        if train:
            push_rows(self.rows, actions, rew, nrows, term)
        if term.any():                                    # episode ends: fresh episodes for those trees
            idx = np.nonzero(term)[0]
            self._episode[idx] += K
            nst[idx] = benv.reset(self._episode[idx])
            nrows[idx] = benv.observe(nst[idx])[0]
        self.states, self.rows = nst, nrows
        return actions, rew, term

    def step(self, train: bool = True):
        """One environment step of every tree; returns (actions, rewards, loss of the last update)."""
        K = self.K
        if self.batch_env is not None:
            actions, rewards, term = self._step_batch(train)
        else:
            envs = self.envs
            push = self.dqn.replay_buffer.push if train else None
            n0 = self.search.n
            actions = self.search.predict(self.state_c, lambda i, a, sc: envs[i].step([a, sc]), train=train, push=push,
                                          reuse=self.keep)
            self.leaf_evals += K * self.search.iterations + (n0 if train else 0) + K
            rewards = np.zeros(K)
            term = np.zeros(K, dtype=bool)
            for i in range(K):
                nobs, rewards[i], term[i], _, info = envs[i].step([int(actions[i]), self.state_c[i]])
                if train:
                    self.dqn.replay_buffer.push(self.obs[i], int(actions[i]), rewards[i], nobs, bool(term[i]))
                self.obs[i], self.state_c[i] = nobs, info['state_c']
                if term[i]:
                    self.obs[i], info = envs[i].reset()
                    self.state_c[i] = info['state_c']
        loss = None
        if train:
            for _ in range(self.updates_per_step):
                loss = self.dqn.update()
                self.dqn.update_target_network(self.tau)
        self.step_count += 1
        self.keep[:] = True
        self.keep[term] = False
        if self.step_count % self.tree_remain_interval == 0:
            self.keep[:] = False
        self.search.advance(actions, keep=self.keep)
        return actions, rewards, loss
