"""Host-side mirror of the reference's learner classes on top of libmctsq_hip.so.

Same class names, constructor arguments, attributes and method semantics as
/root/reference/src/mctsq_config_dqn_per.py (`flavor="per"`) and
/root/reference/src/mctsq_config_dqn.py (`flavor="std"`: sin/cos time features appended in-kernel,
hard target copy):

    TripleEncoderDQN          dqn_per.py:146-236     forward -> Q[B, action_dim]
    PrioritizedReplayBuffer   dqn_per.py:238-321     push / sample / update_priorities / __len__
    DQNModel                  dqn_per.py:323-516     update / update_target_network / save / load

Everything numeric runs in HIP kernels on `device`; torch is used for device memory, the seeded
default initialisation of the parameter tensors (the reference's own torch.nn containers, so a
given seed yields the reference's initial weights), and checkpoint (de)serialisation.
There is no CPU fallback: constructing these classes without a GPU raises MctsqError.
"""
from __future__ import annotations

import random

import numpy as np
import torch
import torch.nn as nn

from ._lib import MctsqError
from .learner import DeviceLearner
from .leafbatch import PROCESS_KEYS, obs_to_row, row_to_obs
from .netdesc import NetSpec
from .per import DevicePER
from .qnet import DeviceQNet

_PRICE_KINDS = ("conv", "gru", "transformer")


def get_activation(activation_name: str) -> nn.Module:
    """dqn_per.py:132-143 (kept for API parity; the HIP head applies the activation itself)."""
    if activation_name.lower() == "relu":
        return nn.ReLU()
    if activation_name.lower() == "tanh":
        return nn.Tanh()
    raise ValueError(f"Unsupported activation function: {activation_name}")


# ---- seeded default initialisation through the reference's parameter containers ---------------
def _init_encoder(kind: str, in_dim: int, embed: int) -> nn.Module:
    m = nn.Module()
    if kind == "conv":                                   # dqn_per.py:25-37
        m.conv1 = nn.Conv1d(in_dim, embed, kernel_size=3, padding=1)
        m.conv2 = nn.Conv1d(embed, embed, kernel_size=3, padding=1)
        m.attention_weights = nn.Conv1d(embed, 1, kernel_size=1)
        m.norm = nn.LayerNorm(embed)
    elif kind == "gru":                                  # dqn_per.py:62-64
        m.gru = nn.GRU(in_dim, embed, batch_first=True)
        m.attention_weights = nn.Linear(embed, 1)
        m.norm = nn.LayerNorm(embed)
    elif kind == "transformer":                          # dqn_per.py:84-88
        m.embedding = nn.Linear(in_dim, embed)
        m.positional_encoding = nn.Parameter(torch.zeros(1, 500, embed))
        m.transformer = nn.TransformerEncoder(nn.TransformerEncoderLayer(embed, 4), 2)
        m.norm = nn.LayerNorm(embed)
    else:
        raise ValueError(kind)
    return m


def _default_init_state_dict(el_dim, process_dim, gas_dim, embed, hidden_layers, hidden_units, action_dim,
                             price_kind, process_kind, gas_kind) -> dict:
    """Parameter tensors created in the reference's construction order (dqn_per.py:180-214) so the
    global torch RNG is consumed identically."""
    root = nn.Module()
    root.price_encoder = _init_encoder(price_kind, el_dim, embed)
    root.process_encoder = _init_encoder(process_kind, process_dim, embed)
    gas = nn.Module()
    if gas_kind == "mlp":                                # dqn_per.py:109-114
        gas.encoder = nn.Sequential(nn.Linear(gas_dim * 3, embed), nn.ReLU(), nn.Linear(embed, embed), nn.ReLU())
    else:
        gas.encoder = _init_encoder("gru", gas_dim, embed)
    root.gas_eua_encoder = gas
    layers, d = [], embed * 3
    for _ in range(hidden_layers):
        layers += [nn.Linear(int(d), int(hidden_units)), nn.ReLU()]
        d = hidden_units
    layers.append(nn.Linear(int(hidden_units), int(action_dim)))
    root.fc_layers = nn.Sequential(*layers)
    return {k: v.detach().clone() for k, v in root.state_dict().items()}


class _ParamView:
    """Stands in for an nn.Parameter: `.data` is a view into the flat device buffer."""

    def __init__(self, data, grad):
        self.data, self.grad = data, grad


class TripleEncoderDQN:
    """Q network resident on the GPU (dqn_per.py:146-236).  Callable like the reference module:
    `net(price, process, gas_eua)` takes fp32 tensors [B,S,F] (CPU tensors are staged to the device,
    the result comes back on the inputs' device) and returns Q[B, action_dim]."""

    flavor = "per"

    def __init__(self, el_input_dim: int, process_input_dim: int, gas_eua_input_dim: int, embed_dim: int,
                 hidden_layers: int, hidden_units: int, action_dim: int, price_encoder_type: str = "conv",
                 process_encoder_type: str = "gru", gas_eua_encoder_type: str = "mlp", activation: str = "relu",
                 device="cuda:0", seq_lens=(25, 10, 3), _qnet: DeviceQNet | None = None):
        if price_encoder_type not in _PRICE_KINDS:
            raise ValueError("Invalid price encoder type.")
        if process_encoder_type not in _PRICE_KINDS:
            raise ValueError("Invalid process encoder type.")
        if gas_eua_encoder_type not in ("mlp", "gru"):
            raise ValueError("Invalid gas/EUA encoder type.")
        get_activation(activation)
        if int(embed_dim) != 64:
            raise MctsqError("the HIP kernels are built for embed_dim = 64 (config_mctsq.yaml:37)")
        extra = 2 if self.flavor == "std" else 0
        self.spec = NetSpec(price=price_encoder_type, process=process_encoder_type, gas=gas_eua_encoder_type,
                            act=activation, flavor=self.flavor, hidden_layers=int(hidden_layers),
                            hidden_units=int(hidden_units), action_dim=int(action_dim), seq=tuple(seq_lens),
                            data_dim=(el_input_dim - extra, process_input_dim - extra, gas_eua_input_dim - extra))
        self._dims = (el_input_dim, process_input_dim, gas_eua_input_dim, int(embed_dim))
        self._check_replay_geometry()
        self.qnet = _qnet if _qnet is not None else DeviceQNet(self.spec, device)
        self.device = self.qnet.device
        if _qnet is None:
            self.load_state_dict(_default_init_state_dict(
                el_input_dim, process_input_dim, gas_eua_input_dim, int(embed_dim), int(hidden_layers),
                int(hidden_units), int(action_dim), price_encoder_type, process_encoder_type, gas_eua_encoder_type))
        self.training = True

    def _check_replay_geometry(self, training: bool = False):
        """The replay ring stores 91-scalar records: price 25 | process 10 x 6 | gas/EUA 3 x 2
        (include/mctsq_hip.h: MCTSQ_STATE_ROW, mctsq_per_gather).  The network itself accepts other window
        lengths for leaf evaluation; DQNModel (which trains from the ring) does not."""
        if training and (tuple(self.spec.seq) != (25, 10, 3) or tuple(self.spec.data_dim) != (1, 6, 2)):
            raise MctsqError(f"the device replay stores windows of 25x1 / 10x6 / 3x2 (config_env.yaml:30-33, "
                             f"config_mctsq.yaml:32); got seq_lens {tuple(self.spec.seq)}, data dims "
                             f"{tuple(self.spec.data_dim)}")

    # nn.Module look-alikes used by the reference's callers
    def eval(self):
        self.training = False          # dropout is never applied by the HIP path (eval semantics)
        return self

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def state_dict(self):
        return self.qnet.state_dict()

    def load_state_dict(self, sd):
        self.qnet.load_state_dict(sd)
        return self

    def named_parameters(self, grad_flat=None):
        views = self.qnet.views()
        gviews = self.qnet.views(grad_flat) if grad_flat is not None else {}
        return [(k, _ParamView(v, gviews.get(k))) for k, v in views.items()]

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def evaluate_leaves(self, price, process, gas_eua, want_prior: bool = True):
        """Batched leaf evaluation on DEVICE tensors: dict(q, vmax, argmax, prior) from one launch
        (MCTSQ._evaluate / _select post-processing, mcts_per.py:291-296, 367-370)."""
        return self.qnet.forward(price, process, gas_eua, want_prior=want_prior)

    def forward(self, price_data, process_data, gas_eua_data):
        src = price_data.device
        dev = [torch.as_tensor(t, dtype=torch.float32).to(self.device, non_blocking=True)
               for t in (price_data, process_data, gas_eua_data)]
        q = self.qnet.forward(*dev, want_prior=False)["q"]
        return q if src == self.device else q.to(src)

    __call__ = forward


class PrioritizedReplayBuffer:
    """Prioritized replay on the device (dqn_per.py:238-321).  push() stages transitions on the host
    and flushes them to the device ring in one copy before the next sample / update / len-sensitive
    call; every staged transition receives the current maximum priority, exactly as consecutive
    reference pushes do (a pushed entry carries the maximum forward, so it cannot change between
    pushes that are not separated by update_priorities)."""

    def __init__(self, capacity: int, alpha: float = 0.6, device="cuda:0", flush_every: int = 4096,
                 mass_capacity: int | None = None):
        """mass_capacity: the summed capacity of all shards when this buffer is one rank's shard of a data-parallel
        replay (DQNModel(process_group=...)); None for a single buffer."""
        self.capacity, self.alpha, self._mass_capacity = int(capacity), alpha, mass_capacity
        self.dev = DevicePER(self.capacity, alpha=float(alpha), device=device, mass_capacity=mass_capacity)
        self._pend_rows, self._pend_next, self._pend_a, self._pend_r, self._pend_d = [], [], [], [], []
        self._flush_every = int(flush_every)

    # ---- reference API ---------------------------------------------------------------------
    def push(self, state, action, reward, next_state, done) -> None:
        row, nrow = obs_to_row(state), obs_to_row(next_state)
        if row.size != 91 or nrow.size != 91:          # ring records are price 25 | process 10 x 6 | gas 3 x 2
            raise MctsqError(f"observation has {row.size} scalars; the device replay stores 91-scalar states "
                             "(25 price + 10 x 6 process + 3 x 2 gas/EUA steps)")
        self._pend_rows.append(row)
        self._pend_next.append(nrow)
        self._pend_a.append(int(action)); self._pend_r.append(float(reward)); self._pend_d.append(float(done))
        if len(self._pend_rows) >= min(self._flush_every, self.capacity):
            self.flush()

    def flush(self) -> None:
        if not self._pend_rows:
            return
        self.dev.push_rows(np.stack(self._pend_rows), np.asarray(self._pend_a, dtype=np.int32),
                           np.asarray(self._pend_r, dtype=np.float32), np.stack(self._pend_next),
                           np.asarray(self._pend_d, dtype=np.float32))
        self._pend_rows, self._pend_next, self._pend_a, self._pend_r, self._pend_d = [], [], [], [], []

    def __len__(self) -> int:
        return min(self.dev.size + len(self._pend_rows), self.capacity)

    def sample_device(self, batch_size: int, beta: float = 0.4):
        """Device-side sample: (batch dict of device tensors, IS weights fp32 [B], logical indices).
        Uniforms come from the global numpy RNG, the stream np.random.choice draws from (:292)."""
        self.flush()
        u = np.random.random_sample(int(batch_size))
        batch, w, idx, _ = self.dev.sample_batch(u, beta=beta)     # sample + IS weights + gather: one launch
        return batch, w, idx

    def sample(self, batch_size: int, beta: float = 0.4):
        """Reference-shaped result (:303-305): (states, actions, rewards, next_states, dones, weights,
        indices) with states / next_states as arrays of observation dicts."""
        batch, w, idx = self.sample_device(batch_size, beta)
        rows = torch.cat([batch["sprice"].flatten(1), batch["sprocess"].flatten(1), batch["sgas"].flatten(1)], 1)
        nrows = torch.cat([batch["nprice"].flatten(1), batch["nprocess"].flatten(1), batch["ngas"].flatten(1)], 1)
        states = np.array([row_to_obs(r) for r in rows.cpu().numpy()], dtype=object)
        nstates = np.array([row_to_obs(r) for r in nrows.cpu().numpy()], dtype=object)
        return (states, batch["act"].cpu().numpy().astype(np.int64), batch["rew"].cpu().numpy(), nstates,
                batch["done"].cpu().numpy().astype(bool), w.cpu(), idx.cpu().numpy())

    def update_priorities(self, indices, priorities) -> None:
        self.flush()
        self.dev.update_priorities(indices, priorities)

    # ---- checkpoint interchange (:487-492, :504-508) ------------------------------------------
    # The reference's attributes are deques (save / load touch `.buffer.maxlen`, `list(.buffer)`, `list(.priorities)`).
    # Here they are read-only VIEWS of the device ring in logical order: len() and maxlen cost nothing, indexing
    # copies the one record, iteration streams the ring in chunks - nothing rebuilds a million-entry deque of
    # observation dicts because a caller asked for `len(rb.buffer)`.
    @property
    def priorities(self):
        self.flush()
        return _RingView(self, "priorities")

    @property
    def buffer(self):
        self.flush()
        return _RingView(self, "transitions")

    def load_lists(self, buffer, priorities) -> None:
        """Rebuild the device ring from the reference's list form."""
        self.dev = DevicePER(self.capacity, alpha=float(self.alpha), device=self.dev.device,
                             mass_capacity=self._mass_capacity)
        items = list(buffer)[-self.capacity:]
        prio = np.asarray(list(priorities)[-self.capacity:], dtype=np.float64)
        if not items:
            return
        self.dev.push_rows(np.stack([obs_to_row(t[0]) for t in items]),
                           np.asarray([t[1] for t in items], dtype=np.int32),
                           np.asarray([t[2] for t in items], dtype=np.float32),
                           np.stack([obs_to_row(t[3]) for t in items]),
                           np.asarray([float(t[4]) for t in items], dtype=np.float32))
        self.dev.set_priorities_exact(prio)


class _RingView:
    """Sequence view of the device ring in logical order (index 0 = oldest entry), standing in for the reference's
    `deque(maxlen=capacity)` attributes: `len`, `maxlen`, indexing (negative too), iteration, `list(...)`."""
    CHUNK = 65536

    def __init__(self, rb: "PrioritizedReplayBuffer", what: str):
        self._rb, self._what, self.maxlen = rb, what, rb.capacity

    def __len__(self):
        return self._rb.dev.size

    def _fetch(self, lo: int, hi: int):
        d = self._rb.dev
        phys = torch.as_tensor((d.head + np.arange(lo, hi)) % d.capacity, device=d.device)
        if self._what == "priorities":
            return d.prio[phys].cpu().numpy().tolist()
        s, n = d.states[phys].cpu().numpy(), d.next_states[phys].cpu().numpy()
        a, r, dn = d.actions[phys].cpu().numpy(), d.rewards[phys].cpu().numpy(), d.dones[phys].cpu().numpy()
        return [(row_to_obs(s[i]), int(a[i]), float(r[i]), row_to_obs(n[i]), bool(dn[i])) for i in range(hi - lo)]

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            raise TypeError("sequence index must be integer, not 'slice'")          # as collections.deque
        i = int(i)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("deque index out of range")
        return self._fetch(i, i + 1)[0]

    def __iter__(self):
        n = len(self)
        for lo in range(0, n, self.CHUNK):
            yield from self._fetch(lo, min(lo + self.CHUNK, n))

    def __array__(self, dtype=None, copy=None):                 # np.array(rb.priorities): one pass, not one fetch per index
        if self._what == "priorities":
            out = np.asarray(self._rb.dev.logical_priorities())
            return out.astype(dtype) if dtype is not None else out
        out = np.empty(len(self), dtype=object)
        for i, t in enumerate(self):
            out[i] = t
        return out


class _AdamFacade:
    """`DQNModel.optimizer` stand-in exposing torch.optim.Adam's state_dict layout (dqn_per.py:486, 503)."""

    def __init__(self, learner: DeviceLearner, lr: float):
        self.learner, self.lr = learner, lr

    def state_dict(self):
        L = self.learner
        m, v = L.policy.views(L.exp_avg), L.policy.views(L.exp_avg_sq)
        state = {}
        if L.step_count > 0:
            for i, k in enumerate(m):
                state[i] = {"step": torch.tensor(float(L.step_count)), "exp_avg": m[k].detach().cpu().clone(),
                            "exp_avg_sq": v[k].detach().cpu().clone()}
        group = {"lr": self.lr, "betas": tuple(L.betas), "eps": L.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(m)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        L = self.learner
        m, v = L.policy.views(L.exp_avg), L.policy.views(L.exp_avg_sq)
        L.exp_avg.zero_(); L.exp_avg_sq.zero_()
        steps = [int(float(s["step"])) for s in sd["state"].values()]
        L.step_count = max(steps) if steps else 0
        for i, k in enumerate(m):
            if i in sd["state"]:
                m[k].copy_(sd["state"][i]["exp_avg"]); v[k].copy_(sd["state"][i]["exp_avg_sq"])
        g = sd["param_groups"][0]
        self.lr = L.lr = float(g["lr"]); L.betas = tuple(g["betas"]); L.eps = float(g["eps"])

    def zero_grad(self):
        self.learner.grad.zero_()


class DQNModel:
    """dqn_per.py:323-516 on the device."""

    flavor = "per"
    _net_cls = TripleEncoderDQN

    def __init__(self, el_input_dim: int, process_input_dim: int, gas_eua_input_dim: int, action_dim: int,
                 embed_dim: int, hidden_layers: int, hidden_units: int, buffer_capacity: int, batch_size: int,
                 gamma: float, lr: float, price_encoder_type: str = "conv", process_encoder_type: str = "gru",
                 gas_eua_encoder_type: str = "mlp", activation: str = "relu", learning_starts: int = 10000,
                 seed: int | None = None, device="cuda:0", seq_lens=(25, 10, 3), process_group=None):
        if seed is not None:                              # dqn_per.py:368-373
            torch.manual_seed(seed)
            np.random.seed(seed)
            random.seed(seed)
            if torch.cuda.is_available():
                torch.cuda.manual_seed_all(seed)
        self.el_input_dim, self.gas_eua_input_dim = el_input_dim, gas_eua_input_dim
        self.process_input_dim, self.action_dim = process_input_dim, action_dim
        self.gamma, self.batch_size, self.learning_starts = gamma, batch_size, learning_starts
        args = (el_input_dim, process_input_dim, gas_eua_input_dim, embed_dim, hidden_layers, hidden_units,
                action_dim, price_encoder_type, process_encoder_type, gas_eua_encoder_type, activation)
        self.policy_net = self._net_cls(*args, device=device, seq_lens=seq_lens)
        self.policy_net._check_replay_geometry(training=True)
        self.learner = DeviceLearner(self.policy_net.spec, device, gamma=gamma, lr=lr, process_group=process_group,
                                     policy=self.policy_net.qnet)
        # the reference constructs (and default-initialises) the target net before overwriting it
        # with the policy weights (:389-396): consume the same RNG stream
        _default_init_state_dict(el_input_dim, process_input_dim, gas_eua_input_dim, int(embed_dim),
                                 int(hidden_layers), int(hidden_units), int(action_dim), price_encoder_type,
                                 process_encoder_type, gas_eua_encoder_type)
        self.learner.target.flat.copy_(self.learner.policy.flat)
        self.target_net = self._net_cls(*args, device=device, seq_lens=seq_lens, _qnet=self.learner.target).eval()
        self.optimizer = _AdamFacade(self.learner, lr)
        world = self.learner.world
        self.replay_buffer = PrioritizedReplayBuffer(buffer_capacity, device=device,
                                                     mass_capacity=buffer_capacity * world if world > 1 else None)

    def update(self):
        """One PER-DQN step (dqn_per.py:401-476): None until the buffer holds max(batch_size,
        learning_starts) transitions, else the scalar loss (0-dim device tensor)."""
        if len(self.replay_buffer) < self.batch_size or len(self.replay_buffer) < self.learning_starts:
            return None
        rb = self.replay_buffer
        rb.flush()
        u = np.random.random_sample(int(self.batch_size))     # the stream np.random.choice draws from (:292)
        if self.learner.world == 1:                           # the whole step is one library call
            self.learner.step(rb.dev, u)
        else:                                                 # replay shards sampled as one buffer (parallel.py):
            self.learner.step_sharded(rb.dev, u)              # every rank drew the same u (seeded alike, :368-373)
        return self.learner.loss[0].clone()

    def update_target_network(self, tau: float = 0.005) -> None:
        """Polyak update (dqn_per.py:510-516)."""
        self.learner.update_target(tau)

    def save(self, filepath: str) -> None:
        rb = self.replay_buffer
        torch.save({"policy_net_state_dict": self.policy_net.state_dict(),
                    "target_net_state_dict": self.target_net.state_dict(),
                    "optimizer_state_dict": self.optimizer.state_dict(),
                    "replay_buffer_data": {"buffer": list(rb.buffer), "priorities": list(rb.priorities),
                                           "alpha": rb.alpha, "capacity": rb.capacity}}, filepath)

    def load(self, filepath: str, allow_pickle: bool = False) -> None:
        """The reference loads with weights_only=False (:500).  Checkpoints hold only tensors, dicts,
        lists and numpy arrays, so by default this loader unpickles nothing but numpy's array
        reconstructors; pass allow_pickle=True for files you trust that need more."""
        if allow_pickle:
            ckpt = torch.load(filepath, map_location="cpu", weights_only=False)
        else:
            import numpy._core.multiarray as _ma
            allowed = [_ma._reconstruct, _ma.scalar, np.ndarray, np.dtype] + \
                      [type(np.dtype(t)) for t in (np.float32, np.float64, np.int64, np.int32, np.bool_)]
            with torch.serialization.safe_globals(allowed):
                ckpt = torch.load(filepath, map_location="cpu", weights_only=True)
        self.policy_net.load_state_dict(ckpt["policy_net_state_dict"])
        self.target_net.load_state_dict(ckpt["target_net_state_dict"])
        self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        if "replay_buffer_data" in ckpt:
            data = ckpt["replay_buffer_data"]
            world = self.learner.world
            self.replay_buffer = PrioritizedReplayBuffer(data["capacity"], alpha=data["alpha"], device=self.learner.device,
                                                         mass_capacity=data["capacity"] * world if world > 1 else None)
            self.replay_buffer.load_lists(data["buffer"], data["priorities"])


# ---- non-_per flavour: /root/reference/src/mctsq_config_dqn.py ------------------------------------
class TripleEncoderDQNStd(TripleEncoderDQN):
    """Input dims include the two appended sin/cos columns (mctsq_config_mcts.py:74-82); callers pass
    the raw windows and the kernels append the features (mctsq_config_dqn.py:242-273)."""

    flavor = "std"

    def forward(self, price_data, process_data, gas_eua_data, price_step_minutes=15, process_step_minutes=15,
                price_start_minute=0, process_start_minute=0, gas_eua_hours_offsets=[-12, 0, 12]):
        if (price_step_minutes, process_step_minutes, price_start_minute, process_start_minute) != (15, 15, 0, 0) \
                or list(gas_eua_hours_offsets) != [-12, 0, 12]:
            raise MctsqError("the HIP path builds in the time features every reference caller uses "
                             "(15-minute steps from minute 0, hours [-12, 0, 12])")
        return super().forward(price_data, process_data, gas_eua_data)

    __call__ = forward


class DQNModelStd(DQNModel):
    flavor = "std"
    _net_cls = TripleEncoderDQNStd

    def update_target_network(self) -> None:          # hard copy (mctsq_config_dqn.py:543-547)
        self.learner.update_target(1.0)
