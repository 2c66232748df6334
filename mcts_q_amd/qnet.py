"""Device-resident Q network: flat parameter buffer + fused HIP forward.

Host-side mirror of TripleEncoderDQN.forward (/root/reference/src/mctsq_config_dqn_per.py:216-236)
plus the leaf epilogue of MCTSQ._evaluate/_select (mctsq_config_mcts_per.py:291-296, 367-370).
PyTorch is used for device memory and streams only; all arithmetic runs in libmctsq_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import lib, ptr, stream_ptr, MctsqError
from .netdesc import NetSpec


def _stream_ptr(device):
    return stream_ptr(device)


class DeviceQNet:
    """One flat fp32 parameter buffer on `device` + leaf evaluation."""

    def __init__(self, spec: NetSpec, device="cuda:0", flat: torch.Tensor | None = None):
        if not torch.cuda.is_available():
            raise MctsqError("DeviceQNet needs a GPU: the HIP path has no CPU fallback")
        self.spec = spec
        self.device = torch.device(device)
        self.desc = spec.c_desc()
        self.n_flat = spec.n_flat
        self.flat = flat if flat is not None else torch.zeros(self.n_flat, dtype=torch.float32,
                                                               device=self.device)
        assert self.flat.numel() == self.n_flat and self.flat.is_contiguous()
        self._ws = None
        # weights as the matrix kernels read them (mctsq_q_prepare: split fragments; the GRU slots leave the tail of their
        # region unwritten, hence zeros, so two prepared buffers of equal parameters compare equal); rebuilt whenever the flat buffer
        # changed: torch's version counter sees in-place torch ops (also through views()), the library's own
        # kernels (Adam, Polyak) do not, so their callers use mark_dirty()
        nprep = lib.check(lib.mctsq_q_prepared_bytes(C.byref(self.desc)))
        self._prep = torch.zeros(max(nprep, 16), dtype=torch.uint8, device=self.device)
        self._prep_version = None

    def mark_dirty(self):
        self._prep_version = None

    def _prepared(self):
        if self._prep_version != self.flat._version:
            lib.check(lib.mctsq_q_prepare(C.byref(self.desc), ptr(self.flat), ptr(self._prep), self._prep.numel(),
                                          _stream_ptr(self.device)))
            self._prep_version = self.flat._version
        return self._prep

    # ---- parameters -------------------------------------------------------------------
    def load_state_dict(self, sd):
        self.flat.copy_(torch.from_numpy(self.spec.pack(sd)))
        return self

    def state_dict(self):
        host = self.flat.detach().cpu().numpy()
        return {k: torch.from_numpy(v.copy()) for k, v in self.spec.unpack(host).items()}

    def views(self, flat=None):
        """dict name -> device view into the flat buffer (no copy)."""
        flat = self.flat if flat is None else flat
        offs, sizes, _ = self.spec.layout()
        return {k: flat[o:o + n].view(shape) for (k, shape), o, n in zip(self.spec.param_keys(), offs, sizes)}

    # ---- forward ----------------------------------------------------------------------
    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=self.device)
        return self._ws

    def _check_in(self, t, i, B):
        S, F = self.spec.seq[i], self.spec.data_dim[i]
        if t.dtype != torch.float32 or t.device != self.device:
            raise ValueError("inputs must be fp32 tensors on the network's device")
        if tuple(t.shape) != (B, S, F):
            raise ValueError(f"input {i}: expected shape {(B, S, F)}, got {tuple(t.shape)}")
        return t.contiguous()

    def forward(self, price, process, gas, want_prior=True, out=None, mid_event=None):
        """Device tensors in -> dict(q [B,A], vmax [B], argmax [B] int32, prior [B,A])."""
        B = price.shape[0]
        price, process, gas = (self._check_in(t, i, B) for i, t in enumerate((price, process, gas)))
        A = self.spec.action_dim
        if out is None:
            out = dict(q=torch.empty(B, A, dtype=torch.float32, device=self.device),
                       vmax=torch.empty(B, dtype=torch.float32, device=self.device),
                       argmax=torch.empty(B, dtype=torch.int32, device=self.device),
                       prior=torch.empty(B, A, dtype=torch.float32, device=self.device) if want_prior else None)
        nbytes = lib.check(lib.mctsq_q_forward_workspace_bytes(C.byref(self.desc), B))
        ws = self._workspace(nbytes)
        lib.check(lib.mctsq_q_forward_prepared(C.byref(self.desc), ptr(self.flat), ptr(self._prepared()), ptr(price),
                                               ptr(process), ptr(gas), B, ptr(out["q"]), ptr(out["vmax"]), ptr(out["argmax"]),
                                               ptr(out.get("prior")), ptr(ws), ws.numel(),
                                               _stream_ptr(self.device), mid_event))
        return out

    __call__ = forward

    def forward_host(self, price, process, gas):
        """Reference calling convention: CPU fp32 tensors in, CPU Q tensor out (H2D/D2H inside)."""
        dev = [torch.as_tensor(t, dtype=torch.float32).to(self.device, non_blocking=True)
               for t in (price, process, gas)]
        return self.forward(*dev)["q"].cpu()
