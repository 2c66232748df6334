#!/usr/bin/env python3
"""Round-2 golden vectors (dev container only; imports the reference like gen_golden.py does).

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_r2.py

  encoders.npz                  the four module-level encoder classes alone: seeded default init ->
                                state_dict, input [8,S,F], output [8,64]   (dqn_per.py:21-130)
  ckpt_ref_conv_conv_mlp.pt     a checkpoint written by the REFERENCE's DQNModel.save after 40 pushes and
                                one update (dqn_per.py:478-493)
  ckpt_ref_conv_conv_mlp.npz    what the reference computes from that state next: np.random.seed(4321),
                                update() -> loss, |td| priorities, sampled indices, post-step parameters

Only data leaves this script (arrays, and a torch checkpoint of tensors / lists / numpy arrays).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import gen_golden as g1  # noqa: E402  (puts /root/reference on sys.path and imports its modules)

ref_per = g1.ref_per


def gen_encoders():
    out, meta = {}, {}
    gen = torch.Generator().manual_seed(99)
    cases = [("conv", lambda: ref_per.ConvAttentionEnc(1, 64), (8, 25, 1)),
             ("conv6", lambda: ref_per.ConvAttentionEnc(6, 64), (8, 10, 6)),
             ("gru", lambda: ref_per.GRUAttentionEnc(6, 64), (8, 10, 6)),
             ("gru1", lambda: ref_per.GRUAttentionEnc(1, 64), (5, 25, 1)),
             ("transformer", lambda: ref_per.TransformerEnc(6, 64, 4, 2), (8, 10, 6)),
             ("gas_mlp", lambda: ref_per.GasEUAEncoder(2, 64, "mlp"), (8, 3, 2)),
             ("gas_gru", lambda: ref_per.GasEUAEncoder(2, 64, "gru"), (8, 3, 2))]
    for i, (tag, make, shape) in enumerate(cases):
        seed = 500 + i
        torch.manual_seed(seed)
        m = make().eval()
        x = torch.rand(*shape, generator=gen)
        with torch.no_grad():
            y = m(x)
        sd = g1.sd_to_np(m.state_dict())
        meta[tag] = dict(seed=seed, shape=list(shape),
                         sd_sum={k: float(v.astype(np.float64).sum()) for k, v in sd.items()},
                         sd_digest={k: g1.tensor_digest(v) for k, v in sd.items()})
        if tag != "transformer":                    # 2.4 MB: rebuilt from the seed by the test, digests checked
            out.update({f"{tag}/sd/{k}": v for k, v in sd.items()})
        out[f"{tag}/x"], out[f"{tag}/y"] = x.numpy(), y.numpy()
        print(tag, "y[0,:3] =", y[0, :3].numpy())
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "encoders.npz"), **out)
    print("wrote encoders.npz")


def gen_checkpoint(enc=("conv", "conv", "mlp"), n_push=40, B=8):
    model = g1.build_model(ref_per, "per", enc, "relu", batch_size=B, capacity=64)
    model.policy_net.eval()
    rng = np.random.RandomState(78)
    obs = [g1.make_obs(rng) for _ in range(n_push + 1)]
    for i in range(n_push):
        model.replay_buffer.push(obs[i], int(rng.randint(5)), float(rng.random_sample()), obs[i + 1],
                                 bool(rng.random_sample() < 0.1))
    pr = np.abs(rng.standard_normal(n_push)).astype(np.float32)
    model.replay_buffer.update_priorities(np.arange(n_push - 8), pr[:n_push - 8])
    loss1 = model.update()
    model.update_target_network(0.005)
    name = "ckpt_ref_" + "_".join(enc)
    model.save(os.path.join(HERE, name + ".pt"))
    out = {"sd_saved/" + k: v for k, v in g1.sd_to_np(model.policy_net.state_dict()).items()}
    out.update({"tgt_saved/" + k: v for k, v in g1.sd_to_np(model.target_net.state_dict()).items()})
    out["prio_saved"] = np.array(model.replay_buffer.priorities, dtype=np.float64)
    out["rows_saved"] = np.array([g1.obs_to_row(t[0]) for t in model.replay_buffer.buffer])
    np.random.seed(4321)
    st = np.random.get_state()
    u = np.random.random_sample(B)
    np.random.set_state(st)
    loss2 = model.update()
    out.update({"sd_post/" + k: v for k, v in g1.sd_to_np(model.policy_net.state_dict()).items()})
    out.update(u=u, loss1=loss1.detach().numpy(), loss2=loss2.detach().numpy(),
               prio_post=np.array(model.replay_buffer.priorities, dtype=np.float64))
    cfg = dict(flavor="per", price=enc[0], process=enc[1], gas=enc[2], act="relu", B=B, seed=g1.SEED,
               np_seed_before_update=4321, capacity=64)
    out["cfg"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "loss1", float(loss1), "loss2", float(loss2))


if __name__ == "__main__":
    torch.set_num_threads(1)
    os.chdir(g1.REF)
    gen_encoders()
    gen_checkpoint()
