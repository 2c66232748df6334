"""CPU: pin the oracle (oracle/*.py) against vectors produced by the reference itself
(tests/golden/*.npz, generator: tests/golden/gen_golden.py)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import per_numpy, qnet_numpy, torch_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
QNET_FILES = sorted(glob.glob(os.path.join(GOLDEN, "qnet_*.npz")))
UPDATE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "update_*.npz")))


def load_case(path):
    z = np.load(path)
    cfg = json.loads(str(z["cfg"]))
    return z, cfg


def rebuild_sd(z, cfg):
    """Weights: stored, or (transformer cases) rebuilt from the seed and checked by digest."""
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    if sd:
        return sd
    import hashlib
    torch.manual_seed(cfg["seed"])
    net = torch_ref.RefQNet(cfg)
    sd = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    for k, dig in cfg["sd_digest"].items():
        got = hashlib.sha256(np.ascontiguousarray(sd[k]).tobytes()).hexdigest()[:16]
        assert got == dig, f"seed-rebuilt weight {k} differs from the reference's"
    return sd


@pytest.mark.parametrize("path", QNET_FILES, ids=[os.path.basename(p)[5:-4] for p in QNET_FILES])
def test_numpy_forward_matches_reference(path):
    z, cfg = load_case(path)
    sd = rebuild_sd(z, cfg)
    q32 = qnet_numpy.q_forward(sd, cfg, z["price"], z["process"], z["gas"], np.float32)
    q64 = qnet_numpy.q_forward(sd, cfg, z["price"], z["process"], z["gas"], np.float64)
    scale = np.abs(z["Q"]).max()
    assert np.abs(q32 - z["Q"]).max() <= 1e-5 * scale
    assert np.abs(q64 - z["Q"]).max() <= 1e-5 * scale
    v, a, pr = qnet_numpy.leaf_outputs(q32)
    assert np.array_equal(a, z["argmax"])
    np.testing.assert_allclose(v, z["vmax"], rtol=0, atol=1e-5 * scale)
    np.testing.assert_allclose(pr, z["prior"], rtol=1e-5, atol=1e-7)


TRAINED_FILES = sorted(glob.glob(os.path.join(GOLDEN, "trained_*.npz")))


@pytest.mark.parametrize("path", TRAINED_FILES, ids=[os.path.basename(p)[8:-4] for p in TRAINED_FILES])
def test_oracle_forward_on_trained_weights(path):
    """The oracle against the reference's forward (fp32 and module.double()) on weights the reference trained for
    300 updates (tests/golden/gen_golden_r3.py): the fp64 restatement reproduces the reference's fp64 evaluation to
    ~1e-12, the torch restatement its fp32 Q bit for bit."""
    z, cfg = load_case(path)
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    assert len(TRAINED_FILES) >= 4 and max(cfg["drift"].values()) > 1.0      # the weights did move
    q64 = qnet_numpy.q_forward(sd, cfg, z["price"], z["process"], z["gas"], np.float64)
    scale = np.abs(z["Q"]).max()
    if "Q64" in z.files:
        assert np.abs(q64 - z["Q64"]).max() <= 1e-10 * scale
    else:
        # the non-_per module casts to float32 inside its forward (time-feature round trip): no fp64 evaluation of the
        # reference exists, the fp64 oracle must sit within fp32 rounding of the reference's fp32 Q
        assert cfg["flavor"] == "std" and np.abs(q64 - z["Q"]).max() <= 2e-6 * scale
    with torch.no_grad():
        q32 = torch_ref.RefQNet.from_state_dict(cfg, sd)(*(torch.from_numpy(z[k]) for k in ("price", "process", "gas")))
    assert np.array_equal(q32.numpy(), z["Q"])
    assert np.array_equal(q64.argmax(1), z["argmax"])


@pytest.mark.parametrize("path", QNET_FILES, ids=[os.path.basename(p)[5:-4] for p in QNET_FILES])
def test_torch_ref_forward_and_grads_match_reference(path):
    z, cfg = load_case(path)
    sd = rebuild_sd(z, cfg)
    net = torch_ref.RefQNet.from_state_dict(cfg, sd)
    s = tuple(torch.from_numpy(z[k]) for k in ("price", "process", "gas"))
    with torch.no_grad():
        q = net(*s).numpy()
    scale = np.abs(z["Q"]).max()
    assert np.abs(q - z["Q"]).max() <= 2e-6 * scale
    loss, td, grads = torch_ref.td_loss_and_grads(net, s, torch.from_numpy(z["a"]),
                                                  torch.from_numpy(z["y"]), torch.from_numpy(z["w"]))
    np.testing.assert_allclose(loss.numpy(), z["loss"], rtol=1e-6)
    gmax = max(float(np.abs(z["grad/" + k]).max()) for k in grads)
    for k, g in grads.items():
        ref = z["grad/" + k]
        got = g.numpy()
        if ref.size != got.size:                      # transformer: first 4096 elements stored
            got = got.ravel()[:4096]
            assert abs(float(torch.sqrt((g.double() ** 2).sum())) - cfg["grad_norm"][k]) <= \
                1e-4 * cfg["grad_norm"][k] + 1e-9
        gs = max(np.abs(ref).max(), 1e-3 * gmax)   # d/d(score bias) is exactly 0 in maths
        assert np.abs(got.reshape(ref.shape) - ref).max() <= 1e-4 * gs, k


@pytest.mark.parametrize("path", UPDATE_FILES, ids=[os.path.basename(p)[7:-4] for p in UPDATE_FILES])
def test_full_update_matches_reference(path):
    z, cfg = load_case(path)
    pre = {k[7:]: z[k] for k in z.files if k.startswith("sd_pre/")}
    tpre = {k[8:]: z[k] for k in z.files if k.startswith("tgt_pre/")}
    net = torch_ref.RefQNet.from_state_dict(cfg, pre)
    tgt = torch_ref.RefQNet.from_state_dict(cfg, tpre)
    # PER sample with the uniforms the reference drew
    idx, w = per_numpy.reference_sample(z["prio_before"], z["u"], cfg["alpha"], cfg["beta"])
    batch = (z["states"][idx], z["actions"][idx], z["rewards"][idx], z["next_states"][idx],
             z["dones"][idx])
    m = {k: torch.zeros_like(p) for k, p in net.named_parameters()}
    v = {k: torch.zeros_like(p) for k, p in net.named_parameters()}
    out = torch_ref.dqn_update(net, tgt, m, v, 1, batch, w, cfg["gamma"], cfg["lr"])
    np.testing.assert_allclose(out["loss"].numpy(), z["loss"], rtol=2e-6)
    # priority write-back, sequential (last duplicate wins)
    prio = z["prio_before"].copy()
    for i, p in zip(idx, out["new_prio"]):
        prio[i] = p
    np.testing.assert_allclose(prio, z["prio_after"], rtol=1e-4, atol=1e-7)
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad_clipped/"))
    for k, p in net.named_parameters():
        gs = max(np.abs(z["grad_clipped/" + k]).max(), 1e-3 * gmax)
        assert np.abs(out["grads"][k].numpy() - z["grad_clipped/" + k]).max() <= 1e-4 * gs, k
        np.testing.assert_allclose(m[k].numpy(), z["exp_avg/" + k], rtol=1e-4, atol=1e-4 * gs * 0.1)
        # Adam's first step is lr*g/(|g|+eps): where |g| is at rounding-noise level (e.g. the
        # softmax-invariant score bias, whose exact gradient is 0) the step is noise/eps-sized.
        quiet = np.abs(z["grad_clipped/" + k]) < 1e-6
        err = np.abs(p.detach().numpy() - z["sd_post/" + k])
        assert err[~quiet].max(initial=0) <= 2e-6, k
        assert err[quiet].max(initial=0) <= 2.0 * cfg["lr"], k
    if cfg["flavor"] == "per":
        torch_ref.polyak(net, tgt, cfg["tau"])
    else:
        torch_ref.hard_sync(net, tgt)
    for k, p in tgt.named_parameters():
        quiet = np.abs(z["grad_clipped/" + k]) < 1e-6
        err = np.abs(p.detach().numpy() - z["tgt_post/" + k])
        assert err[~quiet].max(initial=0) <= 2e-6, k
        assert err[quiet].max(initial=0) <= 2.0 * cfg["lr"], k


def test_time_features():
    z = np.load(os.path.join(GOLDEN, "time_features.npz"))
    assert np.array_equal(qnet_numpy.time_features(25), z["price"])
    assert np.array_equal(qnet_numpy.time_features(10), z["process"])
    assert np.array_equal(qnet_numpy.gas_time_features(), z["gas"])


PER_TAGS = ["small", "wrap", "big", "partial"]


@pytest.mark.parametrize("tag", PER_TAGS)
def test_per_reference_form_bit_exact(tag):
    z = np.load(os.path.join(GOLDEN, "per_cases.npz"))
    idx, w = per_numpy.reference_sample(z[f"{tag}/prio"], z[f"{tag}/u"])
    assert np.array_equal(idx, z[f"{tag}/idx"])
    assert np.array_equal(w, z[f"{tag}/w"])


@pytest.mark.parametrize("tag", PER_TAGS)
def test_per_fixed_point_form_matches_reference(tag):
    z = np.load(os.path.join(GOLDEN, "per_cases.npz"))
    idx, w, _ = per_numpy.fixed_point_sample(z[f"{tag}/prio"], z[f"{tag}/u"],
                                             capacity=int(z[f"{tag}/capacity"]))
    assert np.array_equal(idx, z[f"{tag}/idx"])
    # weights: within one fp32 ulp of the reference's
    assert np.abs(w - z[f"{tag}/w"]).max() <= 1.2e-7 * 1.0


@pytest.mark.parametrize("tag", PER_TAGS)
def test_per_push_update_semantics(tag):
    z = np.load(os.path.join(GOLDEN, "per_cases.npz"))
    cap, n_push = int(z[f"{tag}/capacity"]), int(z[f"{tag}/n_push"])
    buf = per_numpy.RefPER(cap)
    for i in range(n_push):
        buf.push(i)
    buf.update_priorities(z[f"{tag}/upd_idx"], z[f"{tag}/upd_prio"])
    assert np.array_equal(np.array(buf.prio), z[f"{tag}/prio"])
    assert buf.items[0] == int(z[f"{tag}/oldest"])
    assert np.array_equal(np.array([buf.items[i] for i in z[f"{tag}/idx"]]), z[f"{tag}/payload"])
    m = int(np.argmax(buf.prio))
    buf.update_priorities([m], [np.float32(0.25)])
    assert max(buf.prio) == float(z[f"{tag}/max_before"])
    buf.push(-1)
    assert buf.prio[-1] == float(z[f"{tag}/pushed_prio"])
    assert len(buf) == int(z[f"{tag}/len_after"])


def test_pow_det_close_to_libm():
    rng = np.random.RandomState(0)
    x = np.concatenate([np.abs(rng.standard_normal(20000)).astype(np.float32).astype(np.float64),
                        10.0 ** rng.uniform(-30, 4, 20000), [1.0, 0.0, 0.5, 2.0, 256.0]])
    for a in (0.6, -0.4, 1.0, 0.3):
        with np.errstate(all="ignore"):
            ref = np.where(x == 0, 0.0 if a > 0 else np.inf, x ** a)
        got = per_numpy.pow_det(x, a)
        ok = np.isfinite(ref) & (ref != 0)
        rel = np.abs(got[ok] / ref[ok] - 1)
        assert rel.max() < 1e-14                                  # ~1e-29 .. 1e4
        assert rel[(x[ok] > 1e-6) & (x[ok] < 1e3)].max() < 2.5e-15  # the range priorities live in
        assert np.array_equal(got[~ok], ref[~ok])
