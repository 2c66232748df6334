import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_ref
from mcts_q_amd import NetSpec
from mcts_q_amd.learner import DeviceLearner
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
z = np.load(os.path.join(G, "qnet_transformer_transformer_gru_per_relu.npz"))
cfg = json.loads(str(z["cfg"]))
sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
if not sd:
    torch.manual_seed(cfg["seed"])
    sd = {k: v.detach().numpy() for k, v in torch_ref.RefQNet(cfg).state_dict().items()}
spec = NetSpec.from_cfg(cfg)
for seed in range(1, 9):
    L = DeviceLearner(spec).load_state_dict(sd)
    g = torch.Generator().manual_seed(seed)
    B = 8
    s = (torch.rand(B, 25, 1, generator=g), torch.rand(B, 10, 6, generator=g), torch.rand(B, 3, 2, generator=g))
    a, y, w = torch.from_numpy(z["a"]), torch.from_numpy(z["y"]), torch.from_numpy(z["w"])
    batch = dict(sprice=s[0].cuda(), sprocess=s[1].cuda(), sgas=s[2].cuda(), nprice=s[0].cuda(), nprocess=s[1].cuda(),
                 ngas=s[2].cuda(), act=a.int().cuda(), rew=y.cuda(), done=torch.ones(B, device="cuda"))
    L.compute_grad(batch, w.cuda())
    got = spec.unpack(L.grad.cpu().numpy())
    net64 = torch_ref.RefQNet.from_state_dict(cfg, sd).double()
    _, _, g64 = torch_ref.td_loss_and_grads(net64, tuple(t.double() for t in s), a, y.double(), w.double())
    net32 = torch_ref.RefQNet.from_state_dict(cfg, sd)
    _, _, g32 = torch_ref.td_loss_and_grads(net32, s, a, y, w)
    gmax = max(float(v.abs().max()) for v in g64.values())
    worst = ("", 0, 0)
    for k, ref in g64.items():
        scale = max(float(ref.abs().max()), 1e-3 * gmax)
        r = np.abs(got[k] - ref.numpy()).max() / scale
        r32 = np.abs(g32[k].numpy() - ref.numpy()).max() / scale
        if r > worst[1]: worst = (k, r, r32)
    k = worst[0]
    if worst[1] > 2e-5 and got[k].ndim == 2:
        err = np.abs(got[k] - g64[k].numpy())
        rows = np.unique(np.argwhere(err > 0.1 * err.max())[:, 0])
        print(f"   error of {k} sits in rows {rows.tolist()} (a ReLU unit whose pre-activation changes sign between fp32 and fp64 evaluations)")
    print(f"seed {seed}: worst {worst[0]} HIP {worst[1]:.2e} torch-fp32 {worst[2]:.2e}")
