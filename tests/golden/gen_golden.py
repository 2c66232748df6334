#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the dev container, never on the GPU box).

Imports the reference's importable hot-path modules
(/root/reference/src/mctsq_config_dqn_per.py and .../mctsq_config_dqn.py),
drives them with seeded synthetic inputs of the configured shapes and stores
inputs + expected outputs as small .npz fixtures next to this script.

Only DATA leaves this script (arrays in, arrays out); no reference source text
is stored.  Re-run with:

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Fixture families
  qnet_<price>_<process>_<gas>_<flavor>_<act>.npz   forward + loss-gradient vectors
  update_<price>_<process>_<gas>_<flavor>.npz      one full DQNModel.update()
  per_cases.npz                                    PrioritizedReplayBuffer push/sample/update
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import src.mctsq_config_dqn_per as ref_per  # noqa: E402
import src.mctsq_config_dqn as ref_std  # noqa: E402

PROC_KEYS = ["T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow",
             "H2_res_MolarFlow", "H2O_DE_MassFlow", "Elec_Heating"]
SEED = 3654  # config/config_train.yaml seed default


def tensor_digest(t: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()[:16]


def make_obs(rng: np.random.RandomState) -> dict:
    """Synthetic normalised observation dict, same keys/lengths as PTGEnv's obs_norm."""
    obs = {"Elec_Price": rng.random_sample(25)}
    for k in PROC_KEYS:
        obs[k] = rng.random_sample(10)
    obs["Gas_Price"] = rng.random_sample(3)
    obs["EUA_Price"] = rng.random_sample(3)
    return obs


def obs_to_row(obs: dict) -> np.ndarray:
    """Canonical 91-scalar row: price[25] | process[10][6] (time-major) | gas_eua[3][2]."""
    proc = np.stack([obs[k] for k in PROC_KEYS], axis=-1)          # (10, 6)
    ge = np.stack([obs["Gas_Price"], obs["EUA_Price"]], axis=-1)    # (3, 2)
    return np.concatenate([obs["Elec_Price"], proc.ravel(), ge.ravel()])


def build_model(mod, flavor, enc, act, batch_size=8, capacity=64, learning_starts=10):
    extra = 0 if flavor == "per" else 2
    return mod.DQNModel(
        el_input_dim=1 + extra, process_input_dim=6 + extra, gas_eua_input_dim=2 + extra,
        action_dim=5, embed_dim=64, hidden_layers=2, hidden_units=128,
        buffer_capacity=capacity, batch_size=batch_size, gamma=0.99, lr=1e-3,
        price_encoder_type=enc[0], process_encoder_type=enc[1], gas_eua_encoder_type=enc[2],
        activation=act, learning_starts=learning_starts, seed=SEED)


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def gen_qnet(flavor, enc, act, store_weights=True):
    mod = ref_per if flavor == "per" else ref_std
    model = build_model(mod, flavor, enc, act)
    net = model.policy_net
    net.eval()  # mcts_per.py:114 puts the policy net in eval mode (dropout off)
    g = torch.Generator().manual_seed(1234)
    B = 8
    price = torch.rand(B, 25, 1, generator=g)
    process = torch.rand(B, 10, 6, generator=g)
    gas = torch.rand(B, 3, 2, generator=g)
    out = {}
    cfg = dict(flavor=flavor, price=enc[0], process=enc[1], gas=enc[2], act=act,
               embed_dim=64, hidden_layers=2, hidden_units=128, action_dim=5, seed=SEED,
               torch=torch.__version__)
    sd = sd_to_np(net.state_dict())
    if store_weights:
        for k, v in sd.items():
            out["sd/" + k] = v
    else:
        # transformer state dicts are ~5 MB: store digests + moments; tests rebuild the
        # weights from the seed through the same torch.nn containers and check these.
        cfg["sd_digest"] = {k: tensor_digest(v) for k, v in sd.items()}
        cfg["sd_sum"] = {k: float(v.astype(np.float64).sum()) for k, v in sd.items()}
    with torch.no_grad():
        q = net(price, process, gas)
    out.update(price=price.numpy(), process=process.numpy(), gas=gas.numpy(),
               Q=q.numpy(), vmax=q.max(1)[0].numpy(), argmax=q.argmax(1).numpy(),
               prior=torch.softmax(q, -1).numpy())
    # loss-gradient vectors: loss = mean(w * (Q[a] - y)^2)   (dqn_per.py:455-467)
    a = torch.randint(0, 5, (B,), generator=g)
    y = torch.rand(B, generator=g)
    w = torch.rand(B, generator=g) * 0.9 + 0.1
    net.zero_grad()
    qsa = net(price, process, gas).gather(1, a.unsqueeze(1)).squeeze(1)
    td = qsa - y
    loss = (w * td.pow(2)).mean()
    loss.backward()
    out.update(a=a.numpy(), y=y.numpy(), w=w.numpy(), loss=loss.detach().numpy(),
               td=td.detach().numpy())
    grads = {k: p.grad.detach().numpy().copy() for k, p in net.named_parameters()}
    if store_weights:
        for k, v in grads.items():
            out["grad/" + k] = v
    else:
        cfg["grad_norm"] = {k: float(np.sqrt((v.astype(np.float64) ** 2).sum()))
                            for k, v in grads.items()}
        cfg["grad_sum"] = {k: float(v.astype(np.float64).sum()) for k, v in grads.items()}
        for k, v in grads.items():      # keep the small tensors whole, slices of the big ones
            out["grad/" + k] = v if v.size <= 4096 else v.ravel()[:4096].copy()
    out["cfg"] = np.array(json.dumps(cfg))
    name = f"qnet_{enc[0]}_{enc[1]}_{enc[2]}_{flavor}_{act}.npz"
    np.savez_compressed(os.path.join(OUT, name), **out)
    print("wrote", name, "Q[0] =", q[0].numpy())


def gen_update(flavor, enc, n_push=40, B=8):
    """One full DQNModel.update() incl. PER sample, clip, Adam, priority write-back, target sync."""
    mod = ref_per if flavor == "per" else ref_std
    model = build_model(mod, flavor, enc, "relu", batch_size=B, capacity=64)
    if flavor == "per":
        model.policy_net.eval()         # as MCTSQ.__init__ does (mcts_per.py:114)
    rng = np.random.RandomState(77)
    obs = [make_obs(rng) for _ in range(n_push + 1)]
    rows, nrows, acts, rews, dones = [], [], [], [], []
    for i in range(n_push):
        a = int(rng.randint(5)); r = float(rng.random_sample()); d = bool(rng.random_sample() < 0.1)
        model.replay_buffer.push(obs[i], a, r, obs[i + 1], d)
        rows.append(obs_to_row(obs[i])); nrows.append(obs_to_row(obs[i + 1]))
        acts.append(a); rews.append(r); dones.append(d)
    # non-trivial priorities (fp32 |td| values as the reference writes them back)
    pr = np.abs(rng.standard_normal(n_push)).astype(np.float32)
    model.replay_buffer.update_priorities(np.arange(n_push - 8), pr[:n_push - 8])
    prio_before = np.array(model.replay_buffer.priorities, dtype=np.float64)
    pre = sd_to_np(model.policy_net.state_dict())
    tgt_pre = sd_to_np(model.target_net.state_dict())
    st = np.random.get_state()
    u = np.random.random_sample(B)       # what np.random.choice will draw (dqn_per.py:292)
    np.random.set_state(st)
    loss = model.update()
    out = {"sd_pre/" + k: v for k, v in pre.items()}
    out.update({"tgt_pre/" + k: v for k, v in tgt_pre.items()})
    out.update({"sd_post/" + k: v for k, v in sd_to_np(model.policy_net.state_dict()).items()})
    out.update({"grad_clipped/" + k: p.grad.detach().numpy().copy()
                for k, p in model.policy_net.named_parameters()})
    opt = model.optimizer.state_dict()["state"]
    names = [k for k, _ in model.policy_net.named_parameters()]
    for i, k in enumerate(names):
        out["exp_avg/" + k] = opt[i]["exp_avg"].numpy().copy()
        out["exp_avg_sq/" + k] = opt[i]["exp_avg_sq"].numpy().copy()
    if flavor == "per":
        model.update_target_network(0.005)
    else:
        model.update_target_network()
    out.update({"tgt_post/" + k: v for k, v in sd_to_np(model.target_net.state_dict()).items()})
    out.update(states=np.array(rows), next_states=np.array(nrows), actions=np.array(acts),
               rewards=np.array(rews), dones=np.array(dones), prio_before=prio_before, u=u,
               prio_after=np.array(model.replay_buffer.priorities, dtype=np.float64),
               loss=loss.detach().numpy())
    cfg = dict(flavor=flavor, price=enc[0], process=enc[1], gas=enc[2], act="relu", B=B,
               gamma=0.99, lr=1e-3, tau=0.005, alpha=0.6, beta=0.4, seed=SEED,
               param_order=names)
    out["cfg"] = np.array(json.dumps(cfg))
    name = f"update_{enc[0]}_{enc[1]}_{enc[2]}_{flavor}.npz"
    np.savez_compressed(os.path.join(OUT, name), **out)
    print("wrote", name, "loss =", float(loss))


def gen_per():
    """PrioritizedReplayBuffer cases (dqn_per.py:238-321)."""
    out = {}
    PRB = ref_per.PrioritizedReplayBuffer

    def run_case(tag, capacity, n_push, seed, B):
        rng = np.random.RandomState(seed)
        buf = PRB(capacity)
        for i in range(n_push):
            buf.push(i, i % 5, float(i), i + 1, False)   # payload = push ordinal (tracks eviction)
        n = len(buf)
        # mixed priorities: python float 1.0 (never sampled) and np.float32 |td| incl. zeros, dups
        k = n // 2
        idx_u = rng.randint(0, n, size=k)
        pr = np.abs(rng.standard_normal(k)).astype(np.float32)
        pr[::17] = 0.0
        idx_u[1::13] = idx_u[0]                          # duplicates: last one wins (:313-314)
        buf.update_priorities(idx_u, pr)
        prio = np.array(buf.priorities)                  # fp64 (mixed float / float32)
        np.random.seed(seed + 1)
        st = np.random.get_state()
        u = np.random.random_sample(B)
        np.random.set_state(st)
        s, a, r, s2, d, w, idx = buf.sample(B)
        out.update({f"{tag}/capacity": capacity, f"{tag}/n_push": n_push,
                    f"{tag}/upd_idx": idx_u, f"{tag}/upd_prio": pr, f"{tag}/prio": prio,
                    f"{tag}/u": u, f"{tag}/idx": idx, f"{tag}/w": w.numpy(),
                    f"{tag}/payload": np.array(s), f"{tag}/oldest": buf.buffer[0][0]})
        # max-priority rule of push (:265): lower the current max entry, then push again
        m = int(np.argmax(prio))
        buf.update_priorities(np.array([m]), np.array([np.float32(0.25)]))
        before = np.array(buf.priorities)
        buf.push(-1, 0, 0.0, -2, False)
        after = np.array(buf.priorities)
        out.update({f"{tag}/max_before": before.max(), f"{tag}/pushed_prio": after[-1],
                    f"{tag}/len_after": len(buf)})
        print(tag, "n", n, "idx[:6]", idx[:6], "w[:3]", w[:3].numpy())

    run_case("small", 1000, 1000, 11, 64)
    run_case("wrap", 1000, 1500, 12, 64)        # deque overflow: logical 0 == oldest survivor
    run_case("big", 65536, 65536, 13, 256)
    run_case("partial", 4096, 700, 14, 32)      # buffer not full
    np.savez_compressed(os.path.join(OUT, "per_cases.npz"), **out)
    print("wrote per_cases.npz")


def gen_time_features():
    """Non-per flavour's sin/cos features (mctsq_config_dqn.py:550-587)."""
    x = np.zeros((2, 25, 1)); y = np.zeros((2, 10, 6)); z = np.zeros((2, 3, 2))
    out = dict(price=ref_std.add_time_features(x, 15, 0)[0, :, -2:],
               process=ref_std.add_time_features(y, 15, 0)[0, :, -2:],
               gas=ref_std.add_time_features_to_gas_eua(z, [-12, 0, 12])[0, :, -2:])
    np.savez_compressed(os.path.join(OUT, "time_features.npz"), **out)
    print("wrote time_features.npz")


if __name__ == "__main__":
    torch.set_num_threads(1)
    os.chdir(REF)
    combos = [("conv", "conv", "mlp"), ("gru", "gru", "gru"), ("conv", "gru", "mlp")]
    for flavor in ("per", "std"):
        for enc in combos:
            gen_qnet(flavor, enc, "relu")
        gen_qnet(flavor, ("transformer", "transformer", "gru"), "relu", store_weights=False)
    gen_qnet("per", ("gru", "gru", "gru"), "tanh")
    gen_qnet("per", ("gru", "conv", "gru"), "relu")
    gen_qnet("per", ("transformer", "gru", "mlp"), "relu", store_weights=False)
    gen_update("per", ("gru", "gru", "gru"))
    gen_update("per", ("conv", "conv", "mlp"))
    gen_update("std", ("conv", "gru", "mlp"))
    gen_per()
    gen_time_features()
