"""Device-resident PER-DQN learner state: policy / target parameters, gradient and Adam moments
as flat fp32 buffers, and the update step of DQNModel.update
(/root/reference/src/mctsq_config_dqn_per.py:401-476, 510-516) expressed as C-ABI calls.

PyTorch provides device memory, streams and (for data-parallel runs) torch.distributed; all
arithmetic is in libmctsq_hip.so.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import lib, ptr, MctsqError
from .netdesc import NetSpec
from .parallel import all_reduce
from .qnet import DeviceQNet


class DeviceLearner:
    def __init__(self, spec: NetSpec, device="cuda:0", gamma=0.99, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 max_grad_norm=10.0, process_group=None, policy: DeviceQNet | None = None):
        if not torch.cuda.is_available():
            raise MctsqError("DeviceLearner needs a GPU: the HIP path has no CPU fallback")
        self.spec, self.device = spec, torch.device(device)
        self.desc = spec.c_desc()
        n = spec.n_flat
        z = lambda: torch.zeros(n, dtype=torch.float32, device=self.device)
        self.policy = policy if policy is not None else DeviceQNet(spec, self.device)
        self.target = DeviceQNet(spec, self.device)
        self.grad, self.exp_avg, self.exp_avg_sq = z(), z(), z()
        self.gamma, self.lr, self.betas, self.eps = float(gamma), float(lr), betas, float(eps)
        self.max_grad_norm = float(max_grad_norm)
        self.step_count = 0
        self.group = process_group
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        self._ws = None
        self._small = torch.zeros(4096, dtype=torch.uint8, device=self.device)     # norm partials
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=self.device)

    def _st(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def load_state_dict(self, sd, target_sd=None):
        self.policy.load_state_dict(sd)
        self.target.load_state_dict(sd if target_sd is None else target_sd)
        return self

    # ---- gradient of the IS-weighted TD loss (dqn_per.py:455-470) ------------------------------
    def compute_grad(self, batch: dict, isw: torch.Tensor, global_batch: int | None = None):
        """batch: dict from DevicePER.gather (sprice.. ngas, act, rew, done) on this device; isw fp32 [B].
        Fills self.grad (unclipped; after the all-reduce when data-parallel) and self.loss (local mean);
        returns |td| fp32 [B] (the new priorities).  `global_batch` = draws over all ranks (default
        B * world); B may be 0 on a rank that received no draw."""
        B = int(isw.numel())
        if global_batch is None:
            global_batch = B * self.world
        if B == 0:
            self.grad.zero_(); self.loss.zero_()
            if self.world > 1:
                all_reduce(self.grad, group=self.group)
                all_reduce(self.loss, group=self.group)
            return torch.empty(0, dtype=torch.float32, device=self.device)
        need = lib.check(lib.mctsq_dqn_grad_workspace_bytes(C.byref(self.desc), B))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        abs_td = torch.empty(B, dtype=torch.float32, device=self.device)
        lib.check(lib.mctsq_dqn_grad(
            C.byref(self.desc), ptr(self.policy.flat), ptr(self.target.flat),
            ptr(batch["sprice"]), ptr(batch["sprocess"]), ptr(batch["sgas"]),
            ptr(batch["nprice"]), ptr(batch["nprocess"]), ptr(batch["ngas"]),
            ptr(batch["act"]), ptr(batch["rew"]), ptr(batch["done"]), ptr(isw),
            self.gamma, B / float(global_batch), B, ptr(self.grad), ptr(self.loss), ptr(abs_td), None, None,
            ptr(self._ws), self._ws.numel(), self._st()))
        if self.world > 1:                      # data-parallel: one flat-gradient all-reduce (RCCL)
            all_reduce(self.grad, group=self.group)
            self.loss.mul_(B / float(global_batch))           # global mean = sum_r (B_r / B) * local mean
            all_reduce(self.loss, group=self.group)
        return abs_td

    # ---- clip_grad_norm_(10) + Adam.step (+ Polyak) (dqn_per.py:471-472, 510-516) ---------------
    def apply_grad(self, tau: float | None = None):
        """tau None: no target update in this call; 0 < tau <= 1: Polyak (1.0 = hard copy)."""
        self.step_count += 1
        n = self.spec.n_flat
        lib.check(lib.mctsq_clip_adam_polyak(
            ptr(self.policy.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
            ptr(self.target.flat), n, self.step_count, self.lr, self.betas[0], self.betas[1], self.eps,
            self.max_grad_norm, -1.0 if tau is None else float(tau), ptr(self.grad_norm),
            ptr(self._small), self._small.numel(), self._st()))
        self.policy.mark_dirty()
        self.target.mark_dirty()

    def update_target(self, tau: float = 1.0):
        lib.check(lib.mctsq_polyak(ptr(self.policy.flat), ptr(self.target.flat), self.spec.n_flat, float(tau),
                                   self._st()))
        self.target.mark_dirty()
