#!/usr/bin/env python3
"""Round-3 golden vectors: Q forward on TRAINED weights (dev container only; imports the reference like
gen_golden.py does).

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_r3.py

The split-precision matrix products of the HIP path scale their operands from weight / activation ranges, and every
earlier Q fixture uses seeded default-init weights (or those after ONE update).  Here the reference's own training
loop body (DQNModel.update + update_target_network, dqn_per.py:401-476, 510-516) runs for N_UPDATES steps on seeded
synthetic transitions whose reward depends on observation and action, with the learning rate raised so that the
weights move well away from their initial ranges; then

  trained_<price>_<process>_<gas>.npz   sd/<key>            policy-network state_dict after training
                                        price, process, gas  inputs, B = 64, U[0, 1)
                                        Q, vmax, argmax, prior   the reference's fp32 forward (eval mode)
                                        Q64                  the same network evaluated in float64 (module.double())
                                        drift/<key>          max |w_trained - w_init| / max |w_init| per tensor

Only data leaves this script."""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import gen_golden as g1  # noqa: E402

ref_per = g1.ref_per
N_UPDATES, N_PUSH, BATCH, LR = 300, 1024, 32, 4e-3


def gen_trained(enc):
    model = ref_per.DQNModel(
        el_input_dim=1, process_input_dim=6, gas_eua_input_dim=2, action_dim=5, embed_dim=64, hidden_layers=2,
        hidden_units=128, buffer_capacity=2048, batch_size=BATCH, gamma=0.99, lr=LR, price_encoder_type=enc[0],
        process_encoder_type=enc[1], gas_eua_encoder_type=enc[2], activation="relu", learning_starts=10, seed=g1.SEED)
    model.policy_net.eval()                          # as MCTSQ.__init__ does (mcts_per.py:114): no dropout
    init = g1.sd_to_np(model.policy_net.state_dict())
    rng = np.random.RandomState(2024)
    obs = [g1.make_obs(rng) for _ in range(N_PUSH + 1)]
    for i in range(N_PUSH):
        a = int(rng.randint(5))
        o = obs[i]
        r = float(3.0 * o["Elec_Price"][12] - 2.0 * o["T_CAT"][-1] * (a >= 3) + o["Gas_Price"][1] * (a == 2)
                  + 0.1 * rng.standard_normal())
        model.replay_buffer.push(o, a, r, obs[i + 1], bool(rng.random_sample() < 0.05))
    losses = []
    for _ in range(N_UPDATES):
        loss = model.update()
        model.update_target_network(0.005)
        losses.append(float(loss))
    net = model.policy_net
    sd = g1.sd_to_np(net.state_dict())
    g = torch.Generator().manual_seed(4321)
    B = 64
    price, process, gas = torch.rand(B, 25, 1, generator=g), torch.rand(B, 10, 6, generator=g), torch.rand(B, 3, 2, generator=g)
    with torch.no_grad():
        q = net(price, process, gas)
        q64 = copy.deepcopy(net).double()(price.double(), process.double(), gas.double())
    out = {"sd/" + k: v for k, v in sd.items()}
    out.update(price=price.numpy(), process=process.numpy(), gas=gas.numpy(), Q=q.numpy(), Q64=q64.numpy(),
               vmax=q.max(1)[0].numpy(), argmax=q.argmax(1).numpy(), prior=torch.softmax(q, -1).numpy(),
               losses=np.array(losses))
    drift = {k: float(np.abs(sd[k] - init[k]).max() / max(np.abs(init[k]).max(), 1e-12)) for k in sd}
    cfg = dict(flavor="per", price=enc[0], process=enc[1], gas=enc[2], act="relu", updates=N_UPDATES, lr=LR,
               batch=BATCH, seed=g1.SEED, torch=torch.__version__, drift=drift)
    out["cfg"] = np.array(json.dumps(cfg))
    name = "trained_" + "_".join(enc) + ".npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    big = sorted(drift.items(), key=lambda kv: -kv[1])[:3]
    print("wrote", name, "loss %.4f -> %.4f" % (losses[0], losses[-1]), "max|Q| %.3f" % float(q.abs().max()),
          "ref fp32 vs fp64 %.2e" % float((q.double() - q64).abs().max() / q64.abs().max()), "largest drift", big)


def gen_trained_std(enc, act):
    """The non-`_per` flavour (src/mctsq_config_dqn.py: time features appended inside forward, hard target copy) with
    the `tanh` head: trained_std_<enc>_<act>.npz, same keys except Q64 - that module's forward casts to float32 in its
    NumPy round trip (:563-587), so the reference cannot be evaluated in float64; the tests use the fp64 oracle there."""
    model = g1.build_model(g1.ref_std, "std", enc, act, batch_size=BATCH, capacity=2048, learning_starts=10)
    for grp in model.optimizer.param_groups:
        grp["lr"] = LR
    model.policy_net.eval()
    init = g1.sd_to_np(model.policy_net.state_dict())
    rng = np.random.RandomState(2025)
    obs = [g1.make_obs(rng) for _ in range(N_PUSH + 1)]
    for i in range(N_PUSH):
        a = int(rng.randint(5))
        o = obs[i]
        r = float(3.0 * o["Elec_Price"][12] - 2.0 * o["T_CAT"][-1] * (a >= 3) + o["Gas_Price"][1] * (a == 2)
                  + 0.1 * rng.standard_normal())
        model.replay_buffer.push(o, a, r, obs[i + 1], bool(rng.random_sample() < 0.05))
    losses = []
    for i in range(N_UPDATES):
        losses.append(float(model.update()))
        if i % 50 == 49:
            model.update_target_network()
    net = model.policy_net
    sd = g1.sd_to_np(net.state_dict())
    g = torch.Generator().manual_seed(4322)
    B = 64
    price, process, gas = torch.rand(B, 25, 1, generator=g), torch.rand(B, 10, 6, generator=g), torch.rand(B, 3, 2, generator=g)
    with torch.no_grad():
        q = net(price, process, gas)
    out = {"sd/" + k: v for k, v in sd.items()}
    out.update(price=price.numpy(), process=process.numpy(), gas=gas.numpy(), Q=q.numpy(),
               vmax=q.max(1)[0].numpy(), argmax=q.argmax(1).numpy(), prior=torch.softmax(q, -1).numpy(),
               losses=np.array(losses))
    drift = {k: float(np.abs(sd[k] - init[k]).max() / max(np.abs(init[k]).max(), 1e-12)) for k in sd}
    cfg = dict(flavor="std", price=enc[0], process=enc[1], gas=enc[2], act=act, updates=N_UPDATES, lr=LR,
               batch=BATCH, seed=g1.SEED, torch=torch.__version__, drift=drift)
    out["cfg"] = np.array(json.dumps(cfg))
    name = "trained_std_" + "_".join(enc) + "_" + act + ".npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    big = sorted(drift.items(), key=lambda kv: -kv[1])[:3]
    print("wrote", name, "loss %.4f -> %.4f" % (losses[0], losses[-1]), "max|Q| %.3f" % float(q.abs().max()),
          "largest drift", big)


if __name__ == "__main__":
    torch.set_num_threads(4)
    os.chdir(g1.REF)
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    if only in ("", "per"):
        for enc in (("gru", "gru", "gru"), ("conv", "conv", "mlp"), ("transformer", "transformer", "gru")):
            gen_trained(enc)
    if only in ("", "std"):
        gen_trained_std(("gru", "gru", "gru"), "tanh")
