"""ORACLE (test infrastructure, not product code).

Torch-CPU restatement of the reference learner arithmetic: the Q-network evaluated with stock
torch CPU ops (the same third-party kernels the reference runs: nn.GRU, F.conv1d,
nn.TransformerEncoderLayer, F.layer_norm) and the PER-DQN update written out step by step
(TD target, IS-weighted MSE, autograd, global-norm clip, Adam, priority write-back, Polyak / hard
target sync).  Used by tests/ as the checker for gradients and optimizer state, and by bench.py's
cpu_baseline leg as the timed host path.  Never imported by the product package.

Parity status: PINNED by tests/golden/qnet_*.npz (Q, per-parameter gradients) and
tests/golden/update_*.npz (one full DQNModel.update() of the reference), see
tests/test_oracle_golden.py.

Reference lines restated: /root/reference/src/mctsq_config_dqn_per.py:21-236 (network),
:401-476 (update), :510-516 (Polyak); /root/reference/src/mctsq_config_dqn.py:242-273, 543-587
(time features, hard target copy).  torch.optim.Adam / clip_grad_norm_ semantics are PyTorch's
(requirements.txt pins torch; lr 1e-3, betas (0.9, 0.999), eps 1e-8, no weight decay, max_norm 10).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import qnet_numpy

EMBED = 64


class _Enc(nn.Module):
    """One time-series encoder holding stock torch containers under the reference's key names."""

    def __init__(self, kind: str, in_dim: int, seq3: bool = False):
        super().__init__()
        self.kind = kind
        if kind == "conv":
            self.conv1 = nn.Conv1d(in_dim, EMBED, 3, padding=1)
            self.conv2 = nn.Conv1d(EMBED, EMBED, 3, padding=1)
            self.attention_weights = nn.Conv1d(EMBED, 1, 1)
            self.norm = nn.LayerNorm(EMBED)
        elif kind == "gru":
            self.gru = nn.GRU(in_dim, EMBED, batch_first=True)
            self.attention_weights = nn.Linear(EMBED, 1)
            self.norm = nn.LayerNorm(EMBED)
        elif kind == "transformer":
            self.embedding = nn.Linear(in_dim, EMBED)
            self.positional_encoding = nn.Parameter(torch.zeros(1, 500, EMBED))
            self.transformer = nn.TransformerEncoder(nn.TransformerEncoderLayer(EMBED, 4), 2,
                                                     enable_nested_tensor=False)
            self.norm = nn.LayerNorm(EMBED)
        else:
            raise ValueError(kind)

    def forward(self, x):
        if self.kind == "conv":
            h = F.relu(self.conv2(F.relu(self.conv1(x.transpose(1, 2)))))
            a = F.softmax(self.attention_weights(h), dim=-1)
            return self.norm((h * a).sum(-1))
        if self.kind == "gru":
            hs, _ = self.gru(x)
            a = F.softmax(self.attention_weights(hs).squeeze(-1), dim=-1)
            return self.norm(torch.bmm(a.unsqueeze(1), hs).squeeze(1))
        h = self.embedding(x) + self.positional_encoding[:, :x.size(1)]
        return self.norm(self.transformer(h.transpose(0, 1)).mean(0))


class _GasEnc(nn.Module):
    def __init__(self, kind: str, in_dim: int):
        super().__init__()
        self.kind = kind
        if kind == "mlp":
            self.encoder = nn.Sequential(nn.Linear(in_dim * 3, EMBED), nn.ReLU(),
                                         nn.Linear(EMBED, EMBED), nn.ReLU())
        elif kind == "gru":
            self.encoder = _Enc("gru", in_dim)
        else:
            raise ValueError(kind)

    def forward(self, x):
        return self.encoder(x.flatten(1) if self.kind == "mlp" else x)


class RefQNet(nn.Module):
    """Reference-equivalent Q network on torch CPU ops; load with .load_state_dict(reference sd)."""

    def __init__(self, cfg: dict):
        super().__init__()
        self.cfg = dict(cfg)
        extra = 2 if cfg.get("flavor", "per") == "std" else 0
        self.price_encoder = _Enc(cfg["price"], 1 + extra)
        self.process_encoder = _Enc(cfg["process"], 6 + extra)
        self.gas_eua_encoder = _GasEnc(cfg["gas"], 2 + extra)
        hu, hl, ad = cfg.get("hidden_units", 128), cfg.get("hidden_layers", 2), cfg.get("action_dim", 5)
        act = nn.ReLU() if cfg.get("act", "relu") == "relu" else nn.Tanh()
        layers, d = [], 3 * EMBED
        for _ in range(hl):
            layers += [nn.Linear(d, hu), act]
            d = hu
        layers.append(nn.Linear(hu, ad))
        self.fc_layers = nn.Sequential(*layers)
        self.eval()                                  # dropout off (mcts_per.py:114)

    @classmethod
    def from_state_dict(cls, cfg, sd):
        net = cls(cfg)
        net.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
        return net

    def forward(self, price, process, gas):
        if self.cfg.get("flavor", "per") == "std":   # fp64 features, concat, cast (dqn.py:248-273)
            p, q, g = qnet_numpy.append_time_features(price.detach().numpy(), process.detach().numpy(),
                                                      gas.detach().numpy())
            price, process, gas = torch.from_numpy(p), torch.from_numpy(q), torch.from_numpy(g)
        feats = torch.cat([self.price_encoder(price), self.process_encoder(process),
                           self.gas_eua_encoder(gas)], dim=-1)
        return self.fc_layers(feats)


def split_state_rows(rows):
    """Canonical 91-scalar rows -> (price [B,25,1], process [B,10,6], gas [B,3,2]) fp32 tensors."""
    r = torch.as_tensor(np.asarray(rows), dtype=torch.float32)
    return r[:, :25].reshape(-1, 25, 1), r[:, 25:85].reshape(-1, 10, 6), r[:, 85:91].reshape(-1, 3, 2)


def td_loss_and_grads(net, s, a, y, w):
    """loss = mean(w (Q(s)[a] - y)^2) and per-parameter gradients (dqn_per.py:455-470)."""
    net.zero_grad()
    q = net(*s).gather(1, a.long().unsqueeze(1)).squeeze(1)
    td = q - y
    loss = (w * td.pow(2)).mean()
    loss.backward()
    return loss.detach(), td.detach(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


def clip_coef(grads, max_norm=10.0):
    """clip_grad_norm_: coef = min(1, max_norm / (||g||_2 + 1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    return float(torch.clamp(max_norm / (total + 1e-6), max=1.0)), float(total)


def adam_step(params, grads, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor math, in place on dicts of tensors; `step` is 1-based."""
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    for k in params:
        g = grads[k]
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-lr / bc1)


def dqn_update(net, tgt, opt_m, opt_v, step, batch, weights, gamma=0.99, lr=1e-3, max_norm=10.0):
    """One PER-DQN update on an already-sampled batch (dqn_per.py:444-476).
    batch = (rows, actions, rewards, next_rows, dones).  Mutates net/opt state in place.
    Returns dict(loss, td, new_prio, grad_norm, coef)."""
    rows, acts, rews, nrows, dones = batch
    s, s2 = split_state_rows(rows), split_state_rows(nrows)
    a = torch.as_tensor(np.asarray(acts), dtype=torch.int64)
    r = torch.as_tensor(np.asarray(rews), dtype=torch.float32)
    d = torch.as_tensor(np.asarray(dones), dtype=torch.float32)
    w = torch.as_tensor(np.asarray(weights), dtype=torch.float32)
    with torch.no_grad():
        y = r + gamma * tgt(*s2).max(1)[0] * (1 - d)
    loss, td, grads = td_loss_and_grads(net, s, a, y, w)
    coef, gnorm = clip_coef(grads, max_norm)
    grads = {k: g * coef for k, g in grads.items()}
    params = dict(net.named_parameters())
    with torch.no_grad():
        adam_step({k: p.data for k, p in params.items()}, grads, opt_m, opt_v, step, lr)
    return dict(loss=loss, td=td, new_prio=td.abs().numpy(), grad_norm=gnorm, coef=coef,
                grads=grads, y=y)


def polyak(net, tgt, tau=0.005):
    """target <- tau * policy + (1 - tau) * target over parameters() (dqn_per.py:515-516)."""
    with torch.no_grad():
        for tp, p in zip(tgt.parameters(), net.parameters()):
            tp.copy_(tau * p + (1.0 - tau) * tp)


def hard_sync(net, tgt):
    """target <- policy (mctsq_config_dqn.py:547)."""
    tgt.load_state_dict(net.state_dict())
