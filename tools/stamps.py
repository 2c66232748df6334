"""Diagnostic: in-kernel s_memtime stamps of the fused kernel (needs a -DMCTSQ_STAMPS build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mcts_q_amd import NetSpec, lib
from mcts_q_amd.qnet import DeviceQNet
from oracle import torch_ref
cfg = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
torch.manual_seed(0)
sd = {k: v.detach().numpy() for k, v in torch_ref.RefQNet(cfg).state_dict().items()}
net = DeviceQNet(NetSpec.from_cfg(cfg)).load_state_dict(sd)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = [torch.rand(B, 25, 1).cuda(), torch.rand(B, 10, 6).cuda(), torch.rand(B, 3, 2).cuda()]
for _ in range(5):
    net(*x)
torch.cuda.synchronize()
out = np.zeros(256, dtype=np.int64)
lib._load().mctsq_debug_stamps.argtypes = [C.c_void_p]
lib._load().mctsq_debug_stamps(out.ctypes.data)
st = out[:64].reshape(4, 16)[:, :8]
print("ticks per wave: first window (init+stage+loop+LN), second window (group 1), head")
d = np.diff(st, axis=0)
for w in range(8):
    print(w, d[:, w], "total", st[3, w] - st[0, w])
it = out[64:64 + 32]
for name, base in (("waves0-3 first", 0), ("waves4-7 first", 8), ("waves0-3 second", 16), ("waves4-7 second", 24)):
    v = it[base:base + 5]
    print(name, "LN detail: prefetch issue, fin0, sync, fin2:", it[base + 5] - it[base + 3], it[base + 6] - it[base + 5], it[base + 7] - it[base + 6], it[base + 4] - it[base + 7])
    print(name, "stamps (rel. kernel start; 0 = not taken):", np.where(v > 0, v - st[0].min(), 0), "init+stage, sync, loop, LN:", np.diff(v))
print("kernel span (all waves):", st[3].max() - st[0].min())
