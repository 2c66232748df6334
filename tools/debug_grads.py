"""Diagnostic: per-tensor gradient error of the HIP learner vs an fp64 autograd evaluation."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import torch_ref
from mcts_q_amd import NetSpec
from mcts_q_amd.learner import DeviceLearner
name = sys.argv[1] if len(sys.argv) > 1 else "transformer_transformer_gru_per_relu"
z = np.load(f"tests/golden/qnet_{name}.npz")
cfg = json.loads(str(z["cfg"]))
torch.manual_seed(cfg["seed"])
sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")} or \
     {k: v.detach().numpy() for k, v in torch_ref.RefQNet(cfg).state_dict().items()}
spec = NetSpec.from_cfg(cfg)
L = DeviceLearner(spec).load_state_dict(sd)
z = dict(z)
if len(sys.argv) > 2:                       # random inputs instead of the golden ones
    g = torch.Generator().manual_seed(int(sys.argv[2]))
    Bn = 8
    z["price"], z["process"], z["gas"] = (torch.rand(Bn, 25, 1, generator=g).numpy(), torch.rand(Bn, 10, 6, generator=g).numpy(),
                                          torch.rand(Bn, 3, 2, generator=g).numpy())
dev = {k: torch.from_numpy(z[k]).cuda() for k in ("price", "process", "gas")}
B = z["price"].shape[0]
batch = dict(sprice=dev["price"], sprocess=dev["process"], sgas=dev["gas"], nprice=dev["price"],
             nprocess=dev["process"], ngas=dev["gas"], act=torch.from_numpy(z["a"].astype(np.int32)).cuda(),
             rew=torch.from_numpy(z["y"]).cuda(), done=torch.ones(B, device="cuda"))
L.compute_grad(batch, torch.from_numpy(z["w"]).cuda())
got = spec.unpack(L.grad.cpu().numpy())
net64 = torch_ref.RefQNet.from_state_dict(cfg, sd).double()
s64 = tuple(torch.from_numpy(z[n]).double() for n in ("price", "process", "gas"))
_, _, g64 = torch_ref.td_loss_and_grads(net64, s64, torch.from_numpy(z["a"]), torch.from_numpy(z["y"]).double(),
                                        torch.from_numpy(z["w"]).double())
worst = 0
for k, g in got.items():
    t = g64[k].numpy()
    e = np.abs(g - t).max()
    if np.abs(t).max() > 1e-12: worst = max(worst, e / np.abs(t).max())
    if len(sys.argv) > 2 and e / max(np.abs(t).max(), 1e-30) < 2e-6: continue
    print(f"{k:60s} max|g|={np.abs(t).max():.3e} err={e:.3e} rel={e / max(np.abs(t).max(), 1e-30):.2e}")
print("worst rel", worst)
