"""The reference's module-level encoder classes as device-backed callables.

    ConvAttentionEnc(input_dim, embed_dim)                      dqn_per.py:21-55
    GRUAttentionEnc(input_dim, embed_dim)                       dqn_per.py:58-77
    TransformerEnc(input_dim, embed_dim, num_heads, num_layers) dqn_per.py:80-101
    GasEUAEncoder(input_dim, embed_dim, encoder_type="mlp")     dqn_per.py:104-130

Each owns one flat fp32 parameter buffer on the GPU (the encoder's named_parameters() order, the same
seeded default initialisation as the reference's torch.nn containers) and maps `[B, S, F] -> [B, 64]`
through `mctsq_encoder_forward` (include/mctsq_hip.h), i.e. the per-encoder HIP kernels the Q network
uses for mixed encoder combinations.  Also here: the two host-side time-feature helpers of the non-_per
module (mctsq_config_dqn.py:550-587), which the HIP network applies in-kernel.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import EncoderDesc, MctsqError, lib, ptr
from .netdesc import E, KIND, _enc_keys


class _DeviceEncoder:
    kind = None

    def __init__(self, input_dim: int, embed_dim: int, device="cuda:0", _prefix=""):
        if int(embed_dim) != E:
            raise MctsqError("the HIP kernels are built for embed_dim = 64 (config_mctsq.yaml:37)")
        if not torch.cuda.is_available():
            raise MctsqError("the encoders need a GPU: the HIP path has no CPU fallback")
        from .dqn import _init_encoder          # the reference's parameter containers, seeded default init
        self.input_dim, self.device, self._prefix = int(input_dim), torch.device(device), _prefix
        self._keys = _enc_keys(_prefix, self.kind, self.input_dim, 0)
        d = self._desc(3)
        offs, sizes = np.zeros(32, dtype=np.int64), np.zeros(32, dtype=np.int64)
        total = C.c_int64(0)
        n = lib.check(lib.mctsq_encoder_param_layout(C.byref(d), offs.ctypes.data, sizes.ctypes.data, 32,
                                                     C.addressof(total)))
        if n != len(self._keys) or any(int(np.prod(s)) != int(sizes[i]) for i, (_, s) in enumerate(self._keys)):
            raise RuntimeError("encoder parameter layout of libmctsq_hip.so disagrees with the key list")
        self._offs, self._sizes = offs[:n].copy(), sizes[:n].copy()
        self.flat = torch.zeros(int(total.value), dtype=torch.float32, device=self.device)
        self._ws = None
        self.load_state_dict(self._default_init(_init_encoder))

    def _default_init(self, init_encoder):
        return init_encoder(self.kind, self.input_dim, E).state_dict()

    def _desc(self, S):
        return EncoderDesc(KIND[self.kind], int(S), self.input_dim, 0)

    # ---- nn.Module look-alikes ------------------------------------------------------------
    def state_dict(self):
        host = self.flat.detach().cpu().numpy()
        return {k: torch.from_numpy(host[o:o + n].reshape(shape).copy())
                for (k, shape), o, n in zip(self._keys, self._offs, self._sizes)}

    def load_state_dict(self, sd):
        host = np.zeros(self.flat.numel(), dtype=np.float32)
        for (k, shape), o, n in zip(self._keys, self._offs, self._sizes):
            v = sd[k]
            v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            if tuple(v.shape) != tuple(shape):
                raise ValueError(f"{k}: expected shape {shape}, got {tuple(v.shape)}")
            host[o:o + n] = v.astype(np.float32).ravel()
        self.flat.copy_(torch.from_numpy(host))
        return self

    def eval(self):
        return self

    def train(self, mode: bool = True):
        return self

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, S, input_dim] fp32 (CPU tensors are staged to the device; the result comes back on the
        input's device) -> [B, 64]."""
        src = x.device
        xd = torch.as_tensor(x, dtype=torch.float32).to(self.device).contiguous()
        if xd.dim() != 3 or xd.shape[2] != self.input_dim:
            raise ValueError(f"expected [B, S, {self.input_dim}], got {tuple(xd.shape)}")
        B, S = int(xd.shape[0]), int(xd.shape[1])
        d = self._desc(S)
        need = lib.check(lib.mctsq_encoder_workspace_bytes(C.byref(d), B))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty(B, E, dtype=torch.float32, device=self.device)
        lib.check(lib.mctsq_encoder_forward(C.byref(d), ptr(self.flat), ptr(xd), B, ptr(out), ptr(self._ws),
                                            self._ws.numel(), torch.cuda.current_stream(self.device).cuda_stream))
        return out if src == self.device else out.to(src)

    __call__ = forward


class ConvAttentionEnc(_DeviceEncoder):
    kind = "conv"


class GRUAttentionEnc(_DeviceEncoder):
    kind = "gru"


class TransformerEnc(_DeviceEncoder):
    kind = "transformer"

    def __init__(self, input_dim: int, embed_dim: int, num_heads: int = 4, num_layers: int = 2, device="cuda:0"):
        if (int(num_heads), int(num_layers)) != (4, 2):
            raise MctsqError("the HIP transformer encoder is built for 4 heads x 2 layers (dqn_per.py:191, 202)")
        super().__init__(input_dim, embed_dim, device)


class GasEUAEncoder:
    """dqn_per.py:104-130: "mlp" = Linear(3 F, 64)-ReLU-Linear(64, 64)-ReLU on the flattened window (no
    LayerNorm; the window must be 3 steps long), "gru" = a GRUAttentionEnc."""

    def __init__(self, input_dim: int, embed_dim: int, encoder_type: str = "mlp", device="cuda:0"):
        if encoder_type == "mlp":
            self.encoder, self.flatten = _GasMlp(input_dim, embed_dim, device), True
        elif encoder_type == "gru":
            self.encoder, self.flatten = GRUAttentionEnc(input_dim, embed_dim, device), False
        else:
            raise ValueError("Invalid gas/EUA encoder type.")

    def state_dict(self):
        return {"encoder." + k: v for k, v in self.encoder.state_dict().items()}

    def load_state_dict(self, sd):
        self.encoder.load_state_dict({k[len("encoder."):]: v for k, v in sd.items()})
        return self

    def eval(self):
        return self

    def train(self, mode: bool = True):
        return self

    def forward(self, x):
        if self.flatten and x.shape[1] != 3:
            raise ValueError("the gas/EUA mlp encoder is sized input_dim * 3: seq_len must be 3")
        return self.encoder(x)

    __call__ = forward


class _GasMlp(_DeviceEncoder):
    kind = "mlp"

    def _default_init(self, init_encoder):
        import torch.nn as nn
        seq = nn.Sequential(nn.Linear(self.input_dim * 3, E), nn.ReLU(), nn.Linear(E, E), nn.ReLU())
        return seq.state_dict()


# ---- time features of the non-_per module (host NumPy, as the reference's callers use them) ---------
def add_time_features(data: np.ndarray, step_minutes: int, start_minute: int = 0) -> np.ndarray:
    """mctsq_config_dqn.py:550-569: append [sin, cos] of the minute-of-hour angle of every time step.
    data [batch, seq_len, F] -> [batch, seq_len, F + 2] (float64 columns, as numpy promotes)."""
    batch, seq_len, _ = data.shape
    minute = (np.arange(seq_len) * step_minutes + start_minute) % 60
    angle = 2 * np.pi * minute / 60
    cols = np.broadcast_to(np.stack([np.sin(angle), np.cos(angle)], axis=-1), (batch, seq_len, 2))
    return np.concatenate([data, cols], axis=-1)


def add_time_features_to_gas_eua(data: np.ndarray, hours_offsets=(-12, 0, 12)) -> np.ndarray:
    """mctsq_config_dqn.py:571-587: append [sin, cos] of the hour-of-day angle of the three price points."""
    angle = 2 * np.pi * (np.asarray(hours_offsets) % 24) / 24
    cols = np.broadcast_to(np.stack([np.sin(angle), np.cos(angle)], axis=-1), (data.shape[0], 3, 2))
    return np.concatenate([data, cols], axis=-1)
