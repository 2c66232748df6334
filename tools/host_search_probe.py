#!/usr/bin/env python3
"""Host side of the lock-step search alone (no GPU): mctsq_lockstep_search with a leaf evaluation that only writes
constants, on the bench's environment.  Prints the C driver's time split per search of all trees.
python3 tools/host_search_probe.py [trees] [searches] [iterations]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcts_q_amd.ptgenv import PtGBatchEnv
from mcts_q_amd.synthetic_env import synthetic_ptg_kwargs
from mcts_q_amd.treesearch import LockstepMCTS, HostStagedEvaluator, LeafEval

K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
halves = 2
cap = -(-K // halves)


class Ev:
    def __init__(self):
        self.h = HostStagedEvaluator(lambda rows: None, cap, halves=halves)
        self.cap, self.halves = cap, halves
        FWD = ctypes.CFUNCTYPE(ctypes.c_int, *([ctypes.c_void_p] * 6 + [ctypes.c_int64] + [ctypes.c_void_p] * 5 +
                                               [ctypes.c_int64] + [ctypes.c_void_p] * 2))
        EV2 = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
        EV1 = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)
        vm = [np.ctypeslib.as_array(ctypes.cast(self.h.vmax_ptr(h), ctypes.POINTER(ctypes.c_float)), (cap,)) for h in range(halves)]
        pr = [np.ctypeslib.as_array(ctypes.cast(self.h.prior_ptr(h), ctypes.POINTER(ctypes.c_float)), (cap * 5,)) for h in range(halves)]
        for h in range(halves):
            vm[h][:] = 0.1
            pr[h][:] = 0.2
        self._keep = (FWD(lambda *a: 0), EV2(lambda e, s: 0), EV1(lambda e: 0))
        le = LeafEval()
        le.forward, le.event_record, le.event_sync = (ctypes.cast(f, ctypes.c_void_p).value for f in self._keep)
        le.halves, le.cap, le.n_act = halves, cap, 5
        for h in range(halves):
            le.stage[h], le.vmax[h], le.prior[h] = self.h.stage_ptr(h), self.h.vmax_ptr(h), self.h.prior_ptr(h)
        self.le = le

    def __call__(self, rows):
        n = len(rows)
        return np.full(n, 0.1, np.float32), np.full((n, 5), 0.2, np.float32)

    def begin(self):
        pass

    def c_struct(self):
        return self.le


env = PtGBatchEnv(synthetic_ptg_kwargs(0, scale=1.0, train_days=46, env_values=dict(eps_len_d=40)), seed=0)
drv = LockstepMCTS(K, Ev(), iters, 42, 19652, 1.25, batch_env=env, seed=0)
st = env.reset(np.arange(K))
a = drv.predict(st, train=False)
drv.prof = dict(wait=0.0, backup=0.0, expand=0.0, launch=0.0)
t0 = time.perf_counter()
for _ in range(S):
    st = env.step_batch(st, a)[0]
    a = drv.predict(st, train=False)
dt = (time.perf_counter() - t0) / S
print(f"{K} trees x {iters} iterations: {dt * 1e3:.2f} ms per search; " +
      ", ".join(f"{k} {v / S * 1e3:.2f}" for k, v in drv.prof.items()) + f"  ({os.cpu_count()} CPUs)")
