"""Bulk replay operations at BASELINE config 3 size (N = 1M): push of 1M / 25.6k entries (tree part only: the library
call, without the row copies), priority update of 1M entries.  HIP events on the launch stream; JSON line on stdout."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcts_q_amd._lib import lib, ptr  # noqa: E402
from mcts_q_amd.hipevents import HipEvent  # noqa: E402
from mcts_q_amd.per import DevicePER  # noqa: E402

N = 1_000_000
dev = torch.device("cuda", 0)
buf = DevicePER(N, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = HipEvent(), HipEvent()
        a.record(st); fn(); b.record(st)
        torch.cuda.synchronize()
        ts.append(a.elapsed_ms(b))
    return float(np.median(ts)) * 1e3


def push(K, head, size):
    lib.check(lib.mctsq_per_push(ptr(buf.tree), ptr(buf.prio), ptr(buf.maxprio), N, head, size, K, 0.6, N, st))


res = {}
res["push_1M_us"] = timed(lambda: push(N, 0, 0))
res["push_25600_us"] = timed(lambda: push(25600, 0, 300000))
res["push_25600_wrapped_us"] = timed(lambda: push(25600, 17, N))
res["push_512_us"] = timed(lambda: push(512, 17, N))
buf.size = N
idx = torch.arange(N, device=dev)
pr = torch.from_numpy(np.abs(np.random.RandomState(0).standard_normal(N)).astype(np.float32)).to(dev)
res["update_1M_sequential_us"] = timed(lambda: buf.update_priorities(idx, pr), reps=10)
perm = torch.randperm(N, device=dev)
res["update_1M_random_us"] = timed(lambda: buf.update_priorities(perm, pr), reps=10)
sub = perm[:25600].contiguous()
res["update_25600_random_us"] = timed(lambda: buf.update_priorities(sub, pr[:25600]), reps=10)
print(json.dumps(res))
