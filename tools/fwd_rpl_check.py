#!/usr/bin/env python3
"""Forward + gradient of the gru/gru/gru network with 16-sample tiles against 8-sample half tiles (MCTSQ_FWD_RPL=4 / 2,
read once per process): child processes dump their outputs, the parent compares them bit for bit.
python3 tools/fwd_rpl_check.py [B ...]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from mcts_q_amd import NetSpec
from mcts_q_amd.qnet import DeviceQNet
from mcts_q_amd.learner import DeviceLearner
from bench import seeded_net
B = int(sys.argv[1]); out = sys.argv[2]
cfg = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
spec = NetSpec.from_cfg(cfg)
_, sd = seeded_net(cfg)
g = torch.Generator().manual_seed(B)
rows, nrows = torch.rand(B, 91, generator=g), torch.rand(B, 91, generator=g)
split = lambda r: (r[:, :25].reshape(-1, 25, 1).contiguous(), r[:, 25:85].reshape(-1, 10, 6).contiguous(), r[:, 85:91].reshape(-1, 3, 2).contiguous())
s, s2 = split(rows), split(nrows)
net = DeviceQNet(spec).load_state_dict(sd)
o = net.forward(s[0].cuda(), s[1].cuda(), s[2].cuda(), want_prior=True)
q, vmax, am, prior = o["q"], o["vmax"], o["argmax"], o["prior"]
L = DeviceLearner(spec).load_state_dict(sd)
a = torch.randint(0, 5, (B,), generator=g); r = torch.rand(B, generator=g); d = (torch.rand(B, generator=g) < 0.2).float()
w = torch.rand(B, generator=g) * 0.9 + 0.1
batch = dict(sprice=s[0].cuda(), sprocess=s[1].cuda(), sgas=s[2].cuda(), nprice=s2[0].cuda(), nprocess=s2[1].cuda(),
             ngas=s2[2].cuda(), act=a.int().cuda(), rew=r.cuda(), done=d.cuda())
abs_td = L.compute_grad(batch, w.cuda())
torch.cuda.synchronize()
np.savez(out, q=q.cpu().numpy(), vmax=vmax.cpu().numpy(), am=am.cpu().numpy(), prior=prior.cpu().numpy(),
         grad=L.grad.cpu().numpy(), td=abs_td.cpu().numpy(), loss=L.loss.cpu().numpy())
''' % ROOT
ok = True
for B in [int(x) for x in sys.argv[1:]] or [5, 100, 1000]:
    res = {}
    for mode in ("4", "2"):
        f = tempfile.mktemp(suffix=".npz")
        env = dict(os.environ, MCTSQ_FWD_RPL=mode)
        subprocess.run([sys.executable, "-c", CHILD, str(B), f], check=True, env=env, timeout=300)
        res[mode] = dict(np.load(f))
        os.remove(f)
    for k in res["4"]:
        same = np.array_equal(res["4"][k], res["2"][k])
        ok &= same
        if not same:
            d = np.abs(res["4"][k].astype(np.float64) - res["2"][k].astype(np.float64))
            print(f"B={B} {k}: DIFFERENT max|d|={d.max():.3e} at {np.unravel_index(d.argmax(), d.shape)}")
    print(f"B={B}: {'identical' if ok else 'MISMATCH'}")
sys.exit(0 if ok else 1)
