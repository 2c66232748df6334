"""ORACLE (test infrastructure, not product code).

Plain-numpy restatement of the reference Q-network forward pass, written out as explicit
arithmetic from a state_dict (dict name -> ndarray).  It is the checker for the HIP kernels:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Parity status: PINNED by tests/golden/qnet_*.npz, which were produced by importing the
reference's own modules (tests/golden/gen_golden.py); see tests/test_oracle_golden.py.

Reference lines restated (all in /root/reference/src/mctsq_config_dqn_per.py unless noted):
  conv_attention_enc   :21-55     gru_attention_enc :58-77
  transformer_enc      :80-101    gas_eua_mlp       :104-130
  head / q_forward     :205-236   time features      mctsq_config_dqn.py:550-587, 242-273
  leaf value / prior   src/mctsq_config_mcts_per.py:291-296, 367-370

`dtype=np.float64` gives a high-precision "truth" used to measure how far both the torch
CPU path and the HIP path are from exact arithmetic.
"""
from __future__ import annotations

import numpy as np

LN_EPS = 1e-5


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def _layer_norm(x, w, b):
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)     # biased, as torch.layer_norm
    return (x - mu) / np.sqrt(var + LN_EPS) * w + b


def _sub(sd, prefix, dtype):
    n = len(prefix)
    return {k[n:]: np.asarray(v, dtype=dtype) for k, v in sd.items() if k.startswith(prefix)}


# ------------------------------------------------------------------------------- encoders
def conv_attention_enc(p, x):
    """x [B,S,F] -> [B,64].  Conv1d(k3,p1)-ReLU-Conv1d(k3,p1)-ReLU, 1x1 score conv, softmax over
    time, weighted sum, LayerNorm (dqn_per.py:45-55)."""
    def conv3(inp, w, b):                      # inp [B,S,Cin], w [Cout,Cin,3]
        B, S, _ = inp.shape
        pad = np.zeros((B, S + 2, inp.shape[2]), dtype=inp.dtype)
        pad[:, 1:S + 1] = inp
        out = np.zeros((B, S, w.shape[0]), dtype=inp.dtype)
        for tap in range(3):                   # out[t] += W[:,:,tap] . in[t + tap - 1]
            out += pad[:, tap:tap + S] @ w[:, :, tap].T
        return out + b
    h = np.maximum(conv3(x, p["conv1.weight"], p["conv1.bias"]), 0)
    h = np.maximum(conv3(h, p["conv2.weight"], p["conv2.bias"]), 0)
    score = h @ p["attention_weights.weight"][0, :, 0] + p["attention_weights.bias"][0]   # [B,S]
    a = _softmax(score, -1)
    pooled = (h * a[..., None]).sum(axis=1)
    return _layer_norm(pooled, p["norm.weight"], p["norm.bias"])


def gru_attention_enc(p, x):
    """x [B,S,F] -> [B,64].  Single-layer GRU (gate order r,z,n; h0 = 0), linear score per step,
    softmax over time, weighted sum of hidden states, LayerNorm (dqn_per.py:72-77)."""
    wih, whh = p["gru.weight_ih_l0"], p["gru.weight_hh_l0"]
    bih, bhh = p["gru.bias_ih_l0"], p["gru.bias_hh_l0"]
    B, S, _ = x.shape
    E = whh.shape[1]
    h = np.zeros((B, E), dtype=x.dtype)
    hs = []
    for t in range(S):
        gi = x[:, t] @ wih.T + bih
        gh = h @ whh.T + bhh
        r = _sigmoid(gi[:, :E] + gh[:, :E])
        z = _sigmoid(gi[:, E:2 * E] + gh[:, E:2 * E])
        n = np.tanh(gi[:, 2 * E:] + r * gh[:, 2 * E:])
        h = (1.0 - z) * n + z * h
        hs.append(h)
    H = np.stack(hs, axis=1)                                                 # [B,S,E]
    score = H @ p["attention_weights.weight"][0] + p["attention_weights.bias"][0]
    a = _softmax(score, -1)
    pooled = (H * a[..., None]).sum(axis=1)
    return _layer_norm(pooled, p["norm.weight"], p["norm.bias"])


def transformer_enc(p, x, num_heads=4, num_layers=2):
    """x [B,S,F] -> [B,64].  Linear embed + learned positional table, 2 post-LN encoder layers
    (4 heads, ff 2048, ReLU, dropout off), mean over time, LayerNorm (dqn_per.py:96-101)."""
    B, S, _ = x.shape
    h = x @ p["embedding.weight"].T + p["embedding.bias"]
    h = h + p["positional_encoding"][0, :S]
    E = h.shape[-1]
    hd = E // num_heads
    for l in range(num_layers):
        pre = f"transformer.layers.{l}."
        qkv = h @ p[pre + "self_attn.in_proj_weight"].T + p[pre + "self_attn.in_proj_bias"]
        q, k, v = (qkv[..., i * E:(i + 1) * E].reshape(B, S, num_heads, hd) for i in range(3))
        sc = np.einsum("bqhd,bkhd->bhqk", q, k) / np.sqrt(np.asarray(hd, dtype=x.dtype))
        att = _softmax(sc, -1)
        ctx = np.einsum("bhqk,bkhd->bqhd", att, v).reshape(B, S, E)
        sa = ctx @ p[pre + "self_attn.out_proj.weight"].T + p[pre + "self_attn.out_proj.bias"]
        h = _layer_norm(h + sa, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
        ff = np.maximum(h @ p[pre + "linear1.weight"].T + p[pre + "linear1.bias"], 0)
        ff = ff @ p[pre + "linear2.weight"].T + p[pre + "linear2.bias"]
        h = _layer_norm(h + ff, p[pre + "norm2.weight"], p[pre + "norm2.bias"])
    return _layer_norm(h.mean(axis=1), p["norm.weight"], p["norm.bias"])


def gas_eua_mlp(p, x):
    """x [B,3,F] -> flatten -> Linear-ReLU-Linear-ReLU, no LayerNorm (dqn_per.py:108-114,128-130)."""
    f = x.reshape(x.shape[0], -1)
    f = np.maximum(f @ p["0.weight"].T + p["0.bias"], 0)
    return np.maximum(f @ p["2.weight"].T + p["2.bias"], 0)


_ENC = {"conv": conv_attention_enc, "gru": gru_attention_enc, "transformer": transformer_enc}


# ------------------------------------------------------------------------- time features
def time_features(seq_len, step_minutes=15, start_minute=0):
    """[S,2] sin/cos minute-of-hour features (mctsq_config_dqn.py:562-567), fp64."""
    minutes = (np.arange(seq_len) * step_minutes + start_minute) % 60
    rad = 2 * np.pi * minutes / 60
    return np.stack([np.sin(rad), np.cos(rad)], axis=-1)


def gas_time_features(hours=(-12, 0, 12)):
    """[3,2] sin/cos hour-of-day features (mctsq_config_dqn.py:582-585), fp64."""
    rad = 2 * np.pi * (np.array(hours) % 24) / 24
    return np.stack([np.sin(rad), np.cos(rad)], axis=-1)


def append_time_features(price, process, gas):
    """Non-per flavour: features are computed in fp64, concatenated, then cast to fp32
    (mctsq_config_dqn.py:248-273)."""
    def cat(x, tf):
        tf = np.broadcast_to(tf, (x.shape[0],) + tf.shape)
        return np.concatenate([x.astype(np.float64), tf], axis=-1).astype(np.float32)
    return (cat(price, time_features(price.shape[1])), cat(process, time_features(process.shape[1])),
            cat(gas, gas_time_features()))


# ---------------------------------------------------------------------------------- model
def q_forward(sd, cfg, price, process, gas, dtype=np.float32, return_feats=False):
    """Q[B,A] for a TripleEncoderDQN state_dict.  cfg keys: price, process, gas (encoder types),
    act ('relu'|'tanh'), flavor ('per'|'std').  dqn_per.py:232-236."""
    if cfg.get("flavor", "per") == "std":
        price, process, gas = append_time_features(price, process, gas)
    price, process, gas = (np.asarray(a, dtype=dtype) for a in (price, process, gas))
    f1 = _ENC[cfg["price"]](_sub(sd, "price_encoder.", dtype), price)
    f2 = _ENC[cfg["process"]](_sub(sd, "process_encoder.", dtype), process)
    if cfg["gas"] == "mlp":
        f3 = gas_eua_mlp(_sub(sd, "gas_eua_encoder.encoder.", dtype), gas)
    else:
        f3 = gru_attention_enc(_sub(sd, "gas_eua_encoder.encoder.", dtype), gas)
    feats = np.concatenate([f1, f2, f3], axis=-1)
    act = (lambda v: np.maximum(v, 0)) if cfg.get("act", "relu") == "relu" else np.tanh
    fc = _sub(sd, "fc_layers.", dtype)
    n_lin = len([k for k in fc if k.endswith(".weight")])
    h = feats
    for i in range(n_lin):
        h = h @ fc[f"{2 * i}.weight"].T + fc[f"{2 * i}.bias"]
        if i < n_lin - 1:
            h = act(h)
    return (h, feats) if return_feats else h


def leaf_outputs(q):
    """value = max_a Q (mcts_per.py:370), greedy action, prior = softmax(Q) (mcts_per.py:293)."""
    return q.max(axis=1), q.argmax(axis=1), _softmax(q, -1)
