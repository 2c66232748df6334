"""Network shape descriptor shared by the kernels and the Python API.

Mirrors the constructor arguments of the reference TripleEncoderDQN
(/root/reference/src/mctsq_config_dqn_per.py:150-214) and produces (a) the C `mctsq_net_desc`,
(b) the flat-buffer layout (offset/size per parameter tensor) and (c) the state_dict key list in
torch named_parameters() order, so a reference state_dict can be scattered into / gathered from
the single device buffer the kernels, Adam and the all-reduce operate on.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from ._lib import EncoderDesc, NetDesc, lib

KIND = {"conv": 1, "gru": 2, "transformer": 3, "mlp": 4}
ACT = {"relu": 0, "tanh": 1}
E = 64


def _enc_keys(prefix, kind, F, S):
    """(key, shape) pairs in named_parameters() order for one encoder."""
    if kind == "gru":
        return [(prefix + "gru.weight_ih_l0", (192, F)), (prefix + "gru.weight_hh_l0", (192, E)),
                (prefix + "gru.bias_ih_l0", (192,)), (prefix + "gru.bias_hh_l0", (192,)),
                (prefix + "attention_weights.weight", (1, E)), (prefix + "attention_weights.bias", (1,)),
                (prefix + "norm.weight", (E,)), (prefix + "norm.bias", (E,))]
    if kind == "conv":
        return [(prefix + "conv1.weight", (E, F, 3)), (prefix + "conv1.bias", (E,)),
                (prefix + "conv2.weight", (E, E, 3)), (prefix + "conv2.bias", (E,)),
                (prefix + "attention_weights.weight", (1, E, 1)), (prefix + "attention_weights.bias", (1,)),
                (prefix + "norm.weight", (E,)), (prefix + "norm.bias", (E,))]
    if kind == "transformer":
        out = [(prefix + "positional_encoding", (1, 500, E)), (prefix + "embedding.weight", (E, F)),
               (prefix + "embedding.bias", (E,))]
        for l in range(2):
            p = f"{prefix}transformer.layers.{l}."
            out += [(p + "self_attn.in_proj_weight", (192, E)), (p + "self_attn.in_proj_bias", (192,)),
                    (p + "self_attn.out_proj.weight", (E, E)), (p + "self_attn.out_proj.bias", (E,)),
                    (p + "linear1.weight", (2048, E)), (p + "linear1.bias", (2048,)),
                    (p + "linear2.weight", (E, 2048)), (p + "linear2.bias", (E,)),
                    (p + "norm1.weight", (E,)), (p + "norm1.bias", (E,)),
                    (p + "norm2.weight", (E,)), (p + "norm2.bias", (E,))]
        return out + [(prefix + "norm.weight", (E,)), (prefix + "norm.bias", (E,))]
    if kind == "mlp":
        return [(prefix + "0.weight", (E, 3 * F)), (prefix + "0.bias", (E,)),
                (prefix + "2.weight", (E, E)), (prefix + "2.bias", (E,))]
    raise ValueError(f"unknown encoder kind {kind!r}")


@dataclass
class NetSpec:
    """Shape of one TripleEncoderDQN.  `flavor` 'per' = mctsq_config_dqn_per.py (no time features),
    'std' = mctsq_config_dqn.py (sin/cos columns appended inside the kernels)."""
    price: str = "gru"
    process: str = "gru"
    gas: str = "gru"
    act: str = "relu"
    flavor: str = "per"
    hidden_layers: int = 2
    hidden_units: int = 128
    action_dim: int = 5
    seq: tuple = (25, 10, 3)
    data_dim: tuple = (1, 6, 2)
    _cache: dict = field(default_factory=dict, repr=False, compare=False)

    def __post_init__(self):
        for k, allowed in ((self.price, ("conv", "gru", "transformer")),
                           (self.process, ("conv", "gru", "transformer"))):
            if k not in allowed:
                raise ValueError(f"Invalid encoder type {k!r}")
        if self.gas not in ("mlp", "gru"):
            raise ValueError("Invalid gas/EUA encoder type.")
        if self.act.lower() not in ACT:
            raise ValueError(f"Unsupported activation function: {self.act}")

    @classmethod
    def from_cfg(cls, cfg: dict) -> "NetSpec":
        return cls(price=cfg["price"], process=cfg["process"], gas=cfg["gas"], act=cfg.get("act", "relu"),
                   flavor=cfg.get("flavor", "per"), hidden_layers=cfg.get("hidden_layers", 2),
                   hidden_units=cfg.get("hidden_units", 128), action_dim=cfg.get("action_dim", 5))

    @property
    def feat_dims(self):
        extra = 2 if self.flavor == "std" else 0
        return tuple(d + extra for d in self.data_dim)

    def c_desc(self) -> NetDesc:
        d = NetDesc()
        kinds = (self.price, self.process, self.gas)
        for i in range(3):
            tf = 0 if self.flavor == "per" else (2 if i == 2 else 1)
            d.enc[i] = EncoderDesc(KIND[kinds[i]], self.seq[i], self.data_dim[i], tf)
        d.embed_dim, d.hidden_layers, d.hidden_units = E, self.hidden_layers, self.hidden_units
        d.action_dim, d.activation, d.reserved = self.action_dim, ACT[self.act.lower()], 0
        return d

    def param_keys(self):
        """[(state_dict key, shape)] in torch named_parameters() order."""
        F = self.feat_dims
        keys = _enc_keys("price_encoder.", self.price, F[0], self.seq[0])
        keys += _enc_keys("process_encoder.", self.process, F[1], self.seq[1])
        keys += _enc_keys("gas_eua_encoder.encoder.", self.gas, F[2], self.seq[2])
        d = 3 * E
        for i in range(self.hidden_layers):
            keys += [(f"fc_layers.{2 * i}.weight", (self.hidden_units, d)),
                     (f"fc_layers.{2 * i}.bias", (self.hidden_units,))]
            d = self.hidden_units
        n = 2 * self.hidden_layers
        keys += [(f"fc_layers.{n}.weight", (self.action_dim, d)), (f"fc_layers.{n}.bias", (self.action_dim,))]
        return keys

    def layout(self):
        """(offsets, sizes, total) in floats, from the library (single source of truth)."""
        if "layout" not in self._cache:
            desc = self.c_desc()
            offs = np.zeros(160, dtype=np.int64)
            sizes = np.zeros(160, dtype=np.int64)
            total = C.c_int64(0)
            n = lib.check(lib.mctsq_param_layout(C.byref(desc), offs.ctypes.data, sizes.ctypes.data, 160,
                                                 C.addressof(total)))
            keys = self.param_keys()
            if n != len(keys) or any(int(np.prod(s)) != int(sizes[i]) for i, (_, s) in enumerate(keys)):
                raise RuntimeError("parameter layout of libmctsq_hip.so disagrees with the key list")
            self._cache["layout"] = (offs[:n].copy(), sizes[:n].copy(), int(total.value))
        return self._cache["layout"]

    @property
    def n_flat(self) -> int:
        return self.layout()[2]

    def pack(self, sd, out=None):
        """state_dict (name -> array-like) -> flat fp32 numpy buffer (pads zero)."""
        offs, sizes, total = self.layout()
        flat = np.zeros(total, dtype=np.float32) if out is None else out
        for (k, shape), o, n in zip(self.param_keys(), offs, sizes):
            v = sd[k]
            v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            if tuple(v.shape) != tuple(shape):
                raise ValueError(f"{k}: expected shape {shape}, got {tuple(v.shape)}")
            flat[o:o + n] = v.astype(np.float32).ravel()
        return flat

    def unpack(self, flat):
        """flat buffer -> dict name -> numpy view of the right shape."""
        offs, sizes, _ = self.layout()
        return {k: flat[o:o + n].reshape(shape) for (k, shape), o, n in zip(self.param_keys(), offs, sizes)}
