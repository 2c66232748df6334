#!/usr/bin/env python3
"""GPU box: writes tests/golden-style fixture gpurun_out/ckpt_ours/ckpt_ours_conv_conv_mlp.{pt,npz}: a checkpoint
saved by THIS repository's DQNModel.save (40 pushes, one update, target sync), plus the arrays a reader must find
in it.  Copied into tests/golden/ afterwards; tests/test_boundary_cpu.py has the reference's DQNModel.load read it
(dev container only, where /root/reference exists)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from src.mctsq_config_dqn_per import DQNModel
from tests.test_gpu_dqn_api import make_obs

out = os.path.join("gpurun_out", "ckpt_ours"); os.makedirs(out, exist_ok=True)
m = DQNModel(el_input_dim=1, process_input_dim=6, gas_eua_input_dim=2, action_dim=5, embed_dim=64, hidden_layers=2,
             hidden_units=128, buffer_capacity=64, batch_size=8, gamma=0.99, lr=1e-3, price_encoder_type="conv",
             process_encoder_type="conv", gas_eua_encoder_type="mlp", activation="relu", learning_starts=10, seed=3654)
rng = np.random.RandomState(79)
obs = [make_obs(rng) for _ in range(41)]
for i in range(40):
    m.replay_buffer.push(obs[i], int(rng.randint(5)), float(rng.random_sample()), obs[i + 1], bool(rng.random_sample() < 0.1))
loss = m.update()
m.update_target_network(0.005)
m.save(os.path.join(out, "ckpt_ours_conv_conv_mlp.pt"))
z = {"sd/" + k: v.numpy() for k, v in m.policy_net.state_dict().items()}
z.update({"tgt/" + k: v.numpy() for k, v in m.target_net.state_dict().items()})
z["prio"] = np.array(m.replay_buffer.priorities)
z["price0"] = np.stack([t[0]["Elec_Price"] for t in m.replay_buffer.buffer])
z["actions"] = np.array([t[1] for t in m.replay_buffer.buffer])
z["loss"] = np.float32(loss.item())
osd = m.optimizer.state_dict()
z["exp_avg0"] = osd["state"][0]["exp_avg"].numpy()
np.savez_compressed(os.path.join(out, "ckpt_ours_conv_conv_mlp.npz"), **z)
print("wrote", out, "loss", float(loss.item()))
