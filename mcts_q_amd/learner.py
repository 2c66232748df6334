"""Device-resident PER-DQN learner state: policy / target parameters, gradient and Adam moments
as flat fp32 buffers, and the update step of DQNModel.update
(/root/reference/src/mctsq_config_dqn_per.py:401-476, 510-516) expressed as C-ABI calls.

PyTorch provides device memory, streams and (for data-parallel runs) torch.distributed; all
arithmetic is in libmctsq_hip.so.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import lib, ptr, stream_ptr, MctsqError, StepArgs
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
        self.group = process_group
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        self.rank = 0 if process_group is None else torch.distributed.get_rank(process_group)
        # flat gradient with the loss in the word behind it and, behind the loss block, the exchange slots of the
        # data-parallel step (two floats per rank: its largest raw importance weight): data-parallel runs all-reduce
        # the whole block in ONE call
        nx = -(-2 * self.world // 16) * 16
        self._gradbuf = torch.zeros(n + 16 + nx, dtype=torch.float32, device=self.device)
        self.grad, self.loss = self._gradbuf[:n], self._gradbuf[n:n + 1]
        self._exchange = self._gradbuf[n + 16:n + 16 + 2 * self.world]
        self.exp_avg, self.exp_avg_sq = z(), z()
        self.gamma, self.lr, self.betas, self.eps = float(gamma), float(lr), betas, float(eps)
        self.max_grad_norm = float(max_grad_norm)
        self.step_count = 0
        self._ws = None
        self._small = torch.zeros(int(lib.mctsq_optim_workspace_bytes()), dtype=torch.uint8,
                                  device=self.device)                             # grad-norm partials
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=self.device)
        # weights as the kernels multiply by them (split 16-bit fragments), refreshed once per parameter change by ONE
        # launch: the policy's / target's forward fragments live in their DeviceQNet (leaf evaluation between steps
        # uses the policy's), the backward fragments here
        nb = lib.check(lib.mctsq_dqn_prepared_bwd_bytes(C.byref(self.desc)))
        self._prep_bwd = torch.empty(max(nb, 16), dtype=torch.uint8, device=self.device)
        self._prep_versions = None
        nparts = lib.check(lib.mctsq_dqn_grad_sumsq_parts(C.byref(self.desc)))
        self._sumsq_parts = torch.zeros(nparts, dtype=torch.float64, device=self.device)
        self._n_sumsq = C.c_int64(0)

    def _st(self):
        return stream_ptr(self.device)

    def _prepare(self):
        lib.check(lib.mctsq_dqn_prepare(C.byref(self.desc), ptr(self.policy.flat), ptr(self.target.flat),
                                        ptr(self.policy._prep), ptr(self.target._prep), ptr(self._prep_bwd), self._st()))
        self._prep_versions = (self.policy.flat._version, self.target.flat._version)
        self.policy._prep_version, self.target._prep_version = self._prep_versions

    def _ensure_prepared(self):
        """Torch ops on the flat buffers (load_state_dict, copy_) move their version counters; the library's own
        kernels (Adam, Polyak) do not, so their callers here re-prepare right behind them."""
        if (self._prep_versions != (self.policy.flat._version, self.target.flat._version)
                or self.policy._prep_version is None or self.target._prep_version is None):
            self._prepare()

    def load_state_dict(self, sd, target_sd=None):
        self.policy.load_state_dict(sd)
        self.target.load_state_dict(sd if target_sd is None else target_sd)
        return self

    # ---- gradient of the IS-weighted TD loss (dqn_per.py:455-470) ------------------------------
    def _check_batch(self, batch, isw):
        B = int(isw.numel())
        for key, i in (("sprice", 0), ("sprocess", 1), ("sgas", 2), ("nprice", 0), ("nprocess", 1), ("ngas", 2)):
            want = (B, self.spec.seq[i], self.spec.data_dim[i])
            t = batch[key]
            if tuple(t.shape) != want or t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"batch[{key!r}]: expected contiguous fp32 {want} on {self.device}, got "
                                 f"{tuple(t.shape)} {t.dtype} on {t.device}")
        for key, dt in (("act", torch.int32), ("rew", torch.float32), ("done", torch.float32)):
            t = batch[key]
            if tuple(t.shape) != (B,) or t.dtype != dt or t.device != self.device:
                raise ValueError(f"batch[{key!r}]: expected {dt} [{B}] on {self.device}")
        if isw.dtype != torch.float32 or isw.device != self.device:
            raise ValueError("importance weights must be fp32 on the learner's device")
        return B

    def compute_grad(self, batch: dict, isw: torch.Tensor, global_batch: int | None = None):
        """batch: dict from DevicePER.gather / sample_batch (sprice.. ngas, act, rew, done) on this device; isw
        fp32 [B] (rows with weight 0 are padding and contribute nothing).  Fills self.grad (unclipped) and self.loss
        with the GLOBAL-batch mean (after one all-reduce over both when data-parallel); returns |td| fp32 [B] (the
        new priorities).  `global_batch` = draws over all ranks (default B * world); B may be 0 on a rank that
        received no draw."""
        B = self._check_batch(batch, isw)
        if global_batch is None:
            global_batch = B * self.world
        if B == 0:
            self._gradbuf.zero_()
            if self.world > 1:
                all_reduce(self._gradbuf[:self.spec.n_flat + 16], group=self.group)
            return torch.empty(0, dtype=torch.float32, device=self.device)
        need = lib.check(lib.mctsq_dqn_grad_workspace_bytes(C.byref(self.desc), B))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        abs_td = torch.empty(B, dtype=torch.float32, device=self.device)
        # loss_scale = B_local / B_global: gradient and loss of this rank's rows as their share of the global mean
        self._ensure_prepared()
        single = self.world == 1                # the reduction's norm pieces are those of the FINAL gradient
        lib.check(lib.mctsq_dqn_grad_prepared(
            C.byref(self.desc), ptr(self.policy.flat), ptr(self.target.flat),
            ptr(self.policy._prep), ptr(self.target._prep), ptr(self._prep_bwd),
            ptr(batch["sprice"]), ptr(batch["sprocess"]), ptr(batch["sgas"]),
            ptr(batch["nprice"]), ptr(batch["nprocess"]), ptr(batch["ngas"]),
            ptr(batch["act"]), ptr(batch["rew"]), ptr(batch["done"]), ptr(isw),
            self.gamma, B / float(global_batch), B, ptr(self.grad), ptr(self.loss), ptr(abs_td), None, None,
            ptr(self._sumsq_parts) if single else None, C.addressof(self._n_sumsq),
            ptr(self._ws), self._ws.numel(), self._st()))
        self._have_sumsq = single
        if self.world > 1:                      # data-parallel: ONE all-reduce (RCCL) over [gradient | loss]
            all_reduce(self._gradbuf[:self.spec.n_flat + 16], group=self.group)
        return abs_td

    # ---- clip_grad_norm_(10) + Adam.step (+ Polyak) (dqn_per.py:471-472, 510-516) ---------------
    def apply_grad(self, tau: float | None = None):
        """tau None: no target update in this call; 0 < tau <= 1: Polyak (1.0 = hard copy)."""
        self.step_count += 1
        n = self.spec.n_flat
        parts = getattr(self, "_have_sumsq", False)
        lib.check(lib.mctsq_clip_adam_polyak_ex(
            ptr(self.policy.flat), ptr(self.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
            ptr(self.target.flat), n, self.step_count, self.lr, self.betas[0], self.betas[1], self.eps,
            self.max_grad_norm, -1.0 if tau is None else float(tau), ptr(self.grad_norm),
            ptr(self._sumsq_parts) if parts else None, self._n_sumsq.value if parts else 0,
            ptr(self._small), self._small.numel(), self._st()))
        self._have_sumsq = False                # the pieces describe one gradient, used once
        self._prepare()                         # policy (and target) changed: one launch refreshes every fragment

    # ---- DQNModel.update() as ONE library call (dqn_per.py:401-476) ---------------------------------
    def step(self, replay, u, tau: float | None = None, beta: float = 0.4):
        """sample (uniforms `u`, fp64 [B], device tensor or host array) -> batch -> gradient -> clip / Adam
        (-> Polyak when tau is given) -> priority write-back on `replay` (a DevicePER on this device), enqueued by
        one call of mctsq_per_dqn_step.  Single-GPU path; data-parallel runs use compute_grad / apply_grad around
        their all-reduce.  Leaves the loss in self.loss."""
        if self.world != 1:
            raise MctsqError("DeviceLearner.step is the single-GPU path; use compute_grad/apply_grad when data-parallel")
        if replay.size == 0:
            raise ValueError("cannot sample from an empty buffer")
        u = torch.as_tensor(u, dtype=torch.float64).to(self.device).contiguous()
        B = int(u.numel())
        # Deferred re-split: the call re-splits the weight fragments at its START, on a helper lane under the sampling
        # launch, and leaves them stale at its end (one launch and one boundary less on the step's serial chain).
        # Whoever needs the fragments next re-splits them itself: the next step(), compute_grad() through
        # _ensure_prepared(), a leaf evaluation through DeviceQNet._prepared().
        a = getattr(self, "_step_args", None)
        if a is None or self._step_B != B or self._step_replay is not replay:
            need = lib.check(lib.mctsq_per_dqn_step_workspace_bytes(C.byref(self.desc), B))
            self._step_ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            off = (-self._step_ws.data_ptr()) % 256
            a = StepArgs()
            a.d_ws, a.ws_bytes = self._step_ws.data_ptr() + off, need
            a.d_params, a.d_target_params, a.d_grad = ptr(self.policy.flat), ptr(self.target.flat), ptr(self.grad)
            a.d_m, a.d_v = ptr(self.exp_avg), ptr(self.exp_avg_sq)
            a.d_prep_policy, a.d_prep_target, a.d_prep_bwd = ptr(self.policy._prep), ptr(self.target._prep), ptr(self._prep_bwd)
            a.d_loss, a.d_gnorm = ptr(self.loss), ptr(self.grad_norm)
            a.B = B
            self._step_args, self._step_B, self._step_replay = a, B, replay
        r = replay
        ws = r._ws_for(B)
        a.d_tree, a.d_prio, a.d_stamp = ptr(r.tree), ptr(r.prio), ptr(r.stamp)
        a.d_states, a.d_next_states, a.d_actions = ptr(r.states), ptr(r.next_states), ptr(r.actions)
        a.d_rewards, a.d_dones = ptr(r.rewards), ptr(r.dones)
        a.capacity, a.head, a.size, a.mass_capacity = r.capacity, r.head, r.size, r.mass_capacity
        a.alpha, a.beta = r.alpha, float(beta)
        a.d_per_ws, a.per_ws_bytes, a.d_wmax = ptr(ws), ws.numel(), ptr(r.last_local_wmax)
        a.gamma, a.lr, a.beta1, a.beta2, a.eps = self.gamma, self.lr, self.betas[0], self.betas[1], self.eps
        a.max_norm, a.tau = self.max_grad_norm, (-1.0 if tau is None else float(tau))
        self.step_count += 1
        a.adam_step = self.step_count
        a.d_u = ptr(u)
        a.resplit_first = 1
        lib.check(lib.mctsq_per_dqn_step(C.byref(self.desc), C.byref(a), self._st()))
        self._prep_versions = None                          # parameters moved behind the fragments
        self.policy._prep_version = self.target._prep_version = None
        r._max_dirty = True

    # ---- the same step over W data-parallel ranks: two library calls around the one gradient collective -----------
    def step_sharded(self, replay, u, tau: float | None = None, beta: float = 0.4, bcap: int | None = None):
        """One PER-DQN step over the replay shards of all ranks sampled as ONE buffer (parallel.py): `u` = the B global
        uniforms, identical on every rank; `replay` = this rank's DevicePER shard (mass_capacity = the sum of the
        shard capacities).  Collectives per step: the all-gather of the (root mass, size) pairs and ONE all-reduce
        over [gradient | loss | weight-maximum slots]; nothing is read back to the host (the (kept, dropped) counters
        of a step are looked at one step later).  Works at world == 1 too (same results as step() up to the fp32
        rounding of the weight division, include/mctsq_hip.h)."""
        from .parallel import all_gather_rows, draw_capacity
        u = torch.as_tensor(u, dtype=torch.float64).to(self.device).contiguous()
        B = int(u.numel())
        if bcap is None:
            bcap = draw_capacity(B, self.world)
        bcap = -(-int(bcap) // 16) * 16
        d = getattr(self, "_dp", None)
        if d is None or d["B"] != B or d["bcap"] != bcap or d["replay"] is not replay:
            need = lib.check(lib.mctsq_per_dqn_step_workspace_bytes(C.byref(self.desc), bcap))
            ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            a = StepArgs()
            a.d_ws, a.ws_bytes = ws.data_ptr() + (-ws.data_ptr()) % 256, need
            a.d_params, a.d_target_params, a.d_grad = ptr(self.policy.flat), ptr(self.target.flat), ptr(self.grad)
            a.d_m, a.d_v = ptr(self.exp_avg), ptr(self.exp_avg_sq)
            a.d_prep_policy, a.d_prep_target, a.d_prep_bwd = ptr(self.policy._prep), ptr(self.target._prep), ptr(self._prep_bwd)
            a.d_loss, a.d_gnorm, a.d_exchange = ptr(self.loss), ptr(self.grad_norm), ptr(self._exchange)
            a.B, a.bcap, a.world, a.rank = B, bcap, self.world, self.rank
            count = torch.zeros(2, dtype=torch.int32, device=self.device)
            draw = torch.empty(bcap, dtype=torch.int32, device=self.device)
            a.d_count, a.d_draw = ptr(count), ptr(draw)
            slots = [(torch.zeros(2, dtype=torch.int32, pin_memory=True), torch.cuda.Event()) for _ in range(2)]
            d = self._dp = dict(B=B, bcap=bcap, replay=replay, ws=ws, args=a, count=count, draw=draw, slots=slots,
                                turn=0, pending=None)
        if d["pending"] is not None:                # counters of the previous step: recorded a whole step ago, no stall
            host, ev = d["pending"]
            ev.synchronize()
            if int(host[1]) != 0:
                raise MctsqError(f"sharded replay: {int(host[1])} draws of the previous step exceeded this rank's "
                                 f"provision of {bcap} rows; raise bcap (parallel.draw_capacity)")
        a, r = d["args"], replay
        allv = all_gather_rows(r.mass_and_size(reuse=True), self.group)          # collective 1: [W, 2] device words
        d["allv"] = allv = allv.t().contiguous()                                # masses [W] | sizes [W]
        a.d_shard_mass, a.d_shard_size = allv.data_ptr(), allv.data_ptr() + 8 * self.world
        ws = r._ws_for(bcap)
        a.d_tree, a.d_prio, a.d_stamp = ptr(r.tree), ptr(r.prio), ptr(r.stamp)
        a.d_states, a.d_next_states, a.d_actions = ptr(r.states), ptr(r.next_states), ptr(r.actions)
        a.d_rewards, a.d_dones = ptr(r.rewards), ptr(r.dones)
        a.capacity, a.head, a.size, a.mass_capacity = r.capacity, r.head, r.size, r.mass_capacity
        a.alpha, a.beta = r.alpha, float(beta)
        a.d_per_ws, a.per_ws_bytes, a.d_wmax = ptr(ws), ws.numel(), ptr(r.last_local_wmax)
        a.gamma, a.lr, a.beta1, a.beta2, a.eps = self.gamma, self.lr, self.betas[0], self.betas[1], self.eps
        a.max_norm, a.tau = self.max_grad_norm, (-1.0 if tau is None else float(tau))
        self.step_count += 1
        a.adam_step = self.step_count
        a.d_u = ptr(u)
        a.resplit_first = 1
        st = self._st()
        lib.check(lib.mctsq_per_dqn_step_begin(C.byref(self.desc), C.byref(a), st))
        d["turn"] ^= 1
        host, ev = d["slots"][d["turn"]]
        host.copy_(d["count"], non_blocking=True)
        ev.record()
        d["pending"] = (host, ev)
        if self.world > 1:                                                       # collective 2 (RCCL): the whole block
            all_reduce(self._gradbuf, group=self.group)
        lib.check(lib.mctsq_per_dqn_step_finish(C.byref(self.desc), C.byref(a), st))
        self._prep_versions = None                          # parameters moved behind the fragments
        self.policy._prep_version = self.target._prep_version = None
        self._have_sumsq = False
        r._max_dirty = True

    def update_target(self, tau: float = 1.0):
        lib.check(lib.mctsq_polyak(ptr(self.policy.flat), ptr(self.target.flat), self.spec.n_flat, float(tau),
                                   self._st()))
        # the target's fragments are stale now; whoever needs them next re-splits (the next step() does it as a
        # passenger of its sampling launch, compute_grad() through _ensure_prepared())
        self._prep_versions = None
        self.target._prep_version = None
