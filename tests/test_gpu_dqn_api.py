"""GPU: the reference-shaped Python API (mcts_q_amd.dqn / src.mctsq_config_dqn*.py) driven exactly as
the reference's callers drive it - construct with a seed, push observation dicts, update(),
update_target_network(), save / load - against the golden vectors recorded from the reference's own
DQNModel run on the same seed and pushes (tests/golden/gen_golden.py: gen_update)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROC_KEYS = ["T_CAT", "H2_in_MolarFlow", "CH4_syn_MolarFlow", "H2_res_MolarFlow", "H2O_DE_MassFlow", "Elec_Heating"]


def make_obs(rng):
    obs = {"Elec_Price": rng.random_sample(25)}
    for k in PROC_KEYS:
        obs[k] = rng.random_sample(10)
    obs["Gas_Price"] = rng.random_sample(3)
    obs["EUA_Price"] = rng.random_sample(3)
    return obs


def build(cfg, capacity=64, B=8):
    if cfg["flavor"] == "per":
        from src.mctsq_config_dqn_per import DQNModel
        extra = 0
    else:
        from src.mctsq_config_dqn import DQNModel
        extra = 2
    return DQNModel(el_input_dim=1 + extra, process_input_dim=6 + extra, gas_eua_input_dim=2 + extra, action_dim=5,
                    embed_dim=64, hidden_layers=2, hidden_units=128, buffer_capacity=capacity, batch_size=B,
                    gamma=0.99, lr=1e-3, price_encoder_type=cfg["price"], process_encoder_type=cfg["process"],
                    gas_eua_encoder_type=cfg["gas"], activation="relu", learning_starts=10, seed=cfg["seed"])


UPDATE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "update_*.npz")))


@pytest.mark.parametrize("path", UPDATE_FILES, ids=[os.path.basename(p)[7:-4] for p in UPDATE_FILES])
def test_reference_call_sequence_matches_golden(path, tmp_path):
    z = np.load(path)
    cfg = json.loads(str(z["cfg"]))
    model = build(cfg)
    # seeded default init == the reference's initial weights, bit for bit
    sd = model.policy_net.state_dict()
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), z["sd_pre/" + k]), k
    assert model.update() is None                                   # guard (:406-409)
    rng = np.random.RandomState(77)
    n_push = len(z["states"])
    obs = [make_obs(rng) for _ in range(n_push + 1)]
    for i in range(n_push):
        a = int(rng.randint(5)); r = float(rng.random_sample()); d = bool(rng.random_sample() < 0.1)
        model.replay_buffer.push(obs[i], a, r, obs[i + 1], d)
    assert len(model.replay_buffer) == n_push
    pr = np.abs(rng.standard_normal(n_push)).astype(np.float32)
    model.replay_buffer.update_priorities(np.arange(n_push - 8), pr[:n_push - 8])
    assert np.array_equal(np.array(model.replay_buffer.priorities), z["prio_before"])
    loss = model.update()                                            # draws from the global numpy RNG
    np.testing.assert_allclose(float(loss.item()), float(z["loss"]), rtol=5e-6)
    if cfg["flavor"] == "per":
        model.update_target_network(0.005)
    else:
        model.update_target_network()
    np.testing.assert_allclose(np.array(model.replay_buffer.priorities), z["prio_after"], rtol=1e-4, atol=1e-6)
    post, tgt = model.policy_net.state_dict(), model.target_net.state_dict()
    for k in post:
        quiet = np.abs(z["grad_clipped/" + k]) < 1e-6
        for got, ref in ((post[k].numpy(), z["sd_post/" + k]), (tgt[k].numpy(), z["tgt_post/" + k])):
            err = np.abs(got - ref)
            assert err[~quiet].max(initial=0) <= 2e-6, k
            assert err[quiet].max(initial=0) <= 2.0 * cfg["lr"], k
    # optimizer state in torch.optim.Adam's layout
    osd = model.optimizer.state_dict()
    names = cfg["param_order"]
    assert osd["param_groups"][0]["params"] == list(range(len(names)))
    gmax = max(float(np.abs(z["grad_clipped/" + k]).max()) for k in names)
    for i, k in enumerate(names):
        gs = max(np.abs(z["grad_clipped/" + k]).max(), 1e-3 * gmax)
        np.testing.assert_allclose(osd["state"][i]["exp_avg"].numpy(), z["exp_avg/" + k], rtol=1e-4, atol=1e-5 * gs)
        assert float(osd["state"][i]["step"]) == 1.0
    # save -> load into a fresh model: identical state and identical next sample
    f = str(tmp_path / "ckpt.pt")
    model.save(f)
    other = build(dict(cfg, seed=1))
    other.load(f)
    for k, v in other.policy_net.state_dict().items():
        assert np.array_equal(v.numpy(), post[k].numpy())
    assert np.array_equal(np.array(other.replay_buffer.priorities), np.array(model.replay_buffer.priorities))
    assert len(other.replay_buffer) == len(model.replay_buffer)
    u = np.random.RandomState(5).random_sample(16)
    i1, p1, w1 = model.replay_buffer.dev.sample_indices(u)
    i2, p2, w2 = other.replay_buffer.dev.sample_indices(u)
    assert torch.equal(i1, i2) and torch.equal(w1, w2)
    b1, b2 = model.replay_buffer.dev.gather(p1), other.replay_buffer.dev.gather(p2)
    for k in b1:
        assert torch.equal(b1[k], b2[k]), k
    st = list(model.replay_buffer.buffer)
    assert np.allclose(st[0][0]["Elec_Price"], obs[0]["Elec_Price"].astype(np.float32))
    assert st[3][1] in range(5) and isinstance(st[3][4], bool)


def test_policy_net_call_conventions():
    """CPU tensors in -> CPU Q out (what MCTSQ._evaluate/_select do), device tensors stay on device;
    bad encoder / activation names raise ValueError like the reference."""
    from mcts_q_amd.dqn import TripleEncoderDQN
    from mcts_q_amd.leafbatch import pack_leaves
    torch.manual_seed(3)
    net = TripleEncoderDQN(1, 6, 2, 64, 2, 128, 5, "gru", "gru", "gru", "relu").eval()
    rng = np.random.RandomState(0)
    obs = [make_obs(rng) for _ in range(7)]
    price, process, gas = pack_leaves(obs)
    out = net.evaluate_leaves(price, process, gas)
    q_cpu = net(price.cpu(), process.cpu(), gas.cpu())
    assert q_cpu.device.type == "cpu" and q_cpu.shape == (7, 5)
    assert torch.equal(q_cpu, out["q"].cpu())
    assert torch.equal(out["vmax"].cpu(), q_cpu.max(1)[0])
    assert torch.equal(out["argmax"].cpu().long(), q_cpu.argmax(1))
    np.testing.assert_allclose(out["prior"].cpu().numpy(), torch.softmax(q_cpu, -1).numpy(), rtol=1e-5, atol=1e-7)
    # single leaf, as the reference calls it (mcts_per.py:367-370)
    q1 = net(price[:1].cpu(), process[:1].cpu(), gas[:1].cpu())
    assert abs(q1.max().item() - out["vmax"][0].item()) <= 1e-6
    with pytest.raises(ValueError):
        TripleEncoderDQN(1, 6, 2, 64, 2, 128, 5, "lstm", "gru", "gru", "relu")
    with pytest.raises(ValueError):
        TripleEncoderDQN(1, 6, 2, 64, 2, 128, 5, "gru", "gru", "conv", "relu")
    with pytest.raises(ValueError):
        TripleEncoderDQN(1, 6, 2, 64, 2, 128, 5, "gru", "gru", "gru", "gelu")


def test_loads_a_checkpoint_written_by_the_reference(tmp_path):
    """f-3 interchange (dqn_per.py:478-508): tests/golden/ckpt_ref_conv_conv_mlp.pt was written by the REFERENCE's
    DQNModel.save after 40 pushes and one update (tests/golden/gen_golden_r2.py).  DQNModel.load (tensor-only
    unpickling) restores networks, Adam state and the replay buffer; the next update() then reproduces the
    reference's own next update: loss, sampled priorities written back, post-step parameters."""
    z = np.load(os.path.join(GOLDEN, "ckpt_ref_conv_conv_mlp.npz"))
    cfg = json.loads(str(z["cfg"]))
    model = build(dict(cfg, seed=11), capacity=cfg["capacity"], B=cfg["B"])      # different seed: all state comes from the file
    model.load(os.path.join(GOLDEN, "ckpt_ref_conv_conv_mlp.pt"))
    for k, v in model.policy_net.state_dict().items():
        assert np.array_equal(v.numpy(), z["sd_saved/" + k]), k
    for k, v in model.target_net.state_dict().items():
        assert np.array_equal(v.numpy(), z["tgt_saved/" + k]), k
    assert len(model.replay_buffer) == len(z["prio_saved"])
    assert np.array_equal(np.array(model.replay_buffer.priorities), z["prio_saved"])
    rows = np.stack([np.concatenate([t[0]["Elec_Price"], np.stack([t[0][k] for k in PROC_KEYS], -1).ravel(),
                                     np.stack([t[0]["Gas_Price"], t[0]["EUA_Price"]], -1).ravel()])
                     for t in model.replay_buffer.buffer])
    assert np.array_equal(rows, z["rows_saved"].astype(np.float32))
    assert model.learner.step_count == 1                                          # Adam step counter from the file
    np.random.seed(cfg["np_seed_before_update"])
    loss = model.update()
    np.testing.assert_allclose(float(loss.item()), float(z["loss2"]), rtol=5e-6)
    np.testing.assert_allclose(np.array(model.replay_buffer.priorities), z["prio_post"], rtol=1e-4, atol=1e-6)
    post = model.policy_net.state_dict()
    for k in post:
        assert np.abs(post[k].numpy() - z["sd_post/" + k]).max() <= 2.0 * 1e-3, k      # |Adam step| <= lr
        moved = np.abs(z["sd_post/" + k] - z["sd_saved/" + k]) > 5e-4                 # entries with a clear gradient
        assert np.abs(post[k].numpy() - z["sd_post/" + k])[moved].max(initial=0) <= 5e-6, k
    # and a file written here has the reference's layout (the CPU test test_reference_loads_our_checkpoint reads a
    # committed one with the reference's own loader)
    f = str(tmp_path / "ours.pt")
    model.save(f)
    ck = torch.load(f, map_location="cpu", weights_only=False)
    assert sorted(ck) == ["optimizer_state_dict", "policy_net_state_dict", "replay_buffer_data", "target_net_state_dict"]
    assert sorted(ck["replay_buffer_data"]) == ["alpha", "buffer", "capacity", "priorities"]
    assert list(ck["policy_net_state_dict"]) == list(post)
