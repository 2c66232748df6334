"""GPU parity of the learner step (through the C ABI): flat gradient of the IS-weighted TD loss vs
the reference's autograd gradients (golden), one full DQNModel.update() vs the reference's
(golden: loss, clipped grads, Adam moments, parameters, priorities, target sync), and the
clip/Adam/Polyak kernel vs the torch oracle.

Tolerances: gradients 1e-4 of the largest gradient entry of the tensor (same bar the oracle is
pinned with); post-step parameters 2e-6 absolute where the gradient is above rounding noise."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import per_numpy, torch_ref

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sd(z, cfg):
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    if not sd:
        torch.manual_seed(cfg["seed"])
        sd = {k: v.detach().numpy() for k, v in torch_ref.RefQNet(cfg).state_dict().items()}
    return sd


QNET_FILES = sorted(glob.glob(os.path.join(GOLDEN, "qnet_*.npz")))
# One golden input puts ONE FFN pre-activation of the process transformer's first layer within fp32
# rounding of the ReLU kink: the HIP forward (k-ordered fmaf chain) and torch's GEMM land on opposite
# sides of zero for it, so d(linear1) of that layer differs by that single element's contribution
# (3e-3 of the tensor max) and everything upstream of it by <= 4e-4.  tools/debug_grads.py shows every
# other tensor of this case, and all tensors on other inputs, within 2e-6 of an fp64 evaluation; the
# test below (test_transformer_gradients_vs_fp64) pins that.
RELU_KINK_CASES = {"transformer_transformer_gru_per_relu": "process_encoder."}


@pytest.mark.parametrize("path", QNET_FILES, ids=[os.path.basename(p)[5:-4] for p in QNET_FILES])
def test_td_gradient_matches_reference_golden(path):
    from mcts_q_amd import NetSpec
    from mcts_q_amd.learner import DeviceLearner
    z = np.load(path)
    cfg = json.loads(str(z["cfg"]))
    spec = NetSpec.from_cfg(cfg)
    L = DeviceLearner(spec).load_state_dict(_sd(z, cfg))
    dev = {k: torch.from_numpy(z[k]).cuda() for k in ("price", "process", "gas")}
    B = z["price"].shape[0]
    # y is given: done = 1 makes target = reward = y (dqn_per.py:464)
    batch = dict(sprice=dev["price"], sprocess=dev["process"], sgas=dev["gas"], nprice=dev["price"],
                 nprocess=dev["process"], ngas=dev["gas"], act=torch.from_numpy(z["a"].astype(np.int32)).cuda(),
                 rew=torch.from_numpy(z["y"]).cuda(), done=torch.ones(B, device="cuda"))
    abs_td = L.compute_grad(batch, torch.from_numpy(z["w"]).cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(L.loss.item(), float(z["loss"]), rtol=2e-6)
    np.testing.assert_allclose(abs_td.cpu().numpy(), np.abs(z["td"]), rtol=0, atol=1e-5 * np.abs(z["Q"]).max())
    flat = L.grad.cpu().numpy()
    got = spec.unpack(flat)
    gmax = max(float(np.abs(z["grad/" + k]).max()) for k in got)
    for k, g in got.items():
        ref = z["grad/" + k]
        if ref.size != g.size:                       # transformer: first 4096 elements stored
            g = g.ravel()[:4096]
        gs = max(np.abs(ref).max(), 1e-3 * gmax)
        tol = 1e-4 * gs
        if os.path.basename(path)[5:-4] in RELU_KINK_CASES and k.startswith(RELU_KINK_CASES[os.path.basename(path)[5:-4]]):
            tol = 5e-3 * np.abs(ref).max()
        assert np.abs(g.reshape(ref.shape) - ref).max() <= tol, k
    # alignment pads of the flat buffer stay zero
    offs, sizes, total = spec.layout()
    mask = np.ones(total, dtype=bool)
    for o, n in zip(offs, sizes):
        mask[o:o + n] = False
    assert not flat[mask].any()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_transformer_gradients_vs_fp64(seed):
    """Every gradient tensor of the transformer/transformer/gru learner step within 2e-5 (relative to
    the tensor max) of an fp64 autograd evaluation, on random inputs.  One exception is inherent to fp32: a ReLU unit
    of the feed-forward block whose pre-activation for some token is within rounding of zero is on in one evaluation
    and off in the other, which moves that unit's row of d(linear1) by one token's contribution (seed 2 has such a
    unit: tools/tfgrad_probe.py names the row).  At most one row per linear1 tensor may therefore miss the bar, and
    when one does, the tensors upstream of it are held to 1e-4 (the missing token's knock-on, measured 3e-5) instead
    of 2e-5; without such a row every tensor has to meet 2e-5."""
    from mcts_q_amd import NetSpec
    from mcts_q_amd.learner import DeviceLearner
    z = np.load(os.path.join(GOLDEN, "qnet_transformer_transformer_gru_per_relu.npz"))
    cfg = json.loads(str(z["cfg"]))
    sd = _sd(z, cfg)
    spec = NetSpec.from_cfg(cfg)
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
    gmax = max(float(v.abs().max()) for v in g64.values())
    flipped = []
    for k, ref in g64.items():
        if ".linear1." in k:
            scale = max(float(ref.abs().max()), 1e-3 * gmax)
            bad = np.unique(np.argwhere(np.abs(got[k] - ref.numpy()) > 2e-5 * scale)[:, 0])
            assert len(bad) <= 1, (k, bad)
            flipped += [(k, int(r)) for r in bad]
    assert len(flipped) <= 2, flipped
    bar = 1e-4 if flipped else 2e-5
    for k, ref in g64.items():
        scale = max(float(ref.abs().max()), 1e-3 * gmax)      # d(score bias) is exactly 0 in maths
        err = np.abs(got[k] - ref.numpy())
        if ".linear1." in k:
            err = np.delete(err, [r for kk, r in flipped if kk == k], axis=0)
        assert err.max() <= bar * scale, (k, flipped)


UPDATE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "update_*.npz")))


@pytest.mark.parametrize("path", UPDATE_FILES, ids=[os.path.basename(p)[7:-4] for p in UPDATE_FILES])
def test_full_update_matches_reference_golden(path):
    """sample -> gather -> grad -> clip/Adam -> priority write-back -> target sync, all on the device."""
    from mcts_q_amd import NetSpec
    from mcts_q_amd.learner import DeviceLearner
    from mcts_q_amd.per import DevicePER
    z = np.load(path)
    cfg = json.loads(str(z["cfg"]))
    spec = NetSpec.from_cfg(cfg)
    pre = {k[7:]: z[k] for k in z.files if k.startswith("sd_pre/")}
    tpre = {k[8:]: z[k] for k in z.files if k.startswith("tgt_pre/")}
    L = DeviceLearner(spec, gamma=cfg["gamma"], lr=cfg["lr"]).load_state_dict(pre, tpre)
    n = len(z["states"])
    buf = DevicePER(64, alpha=cfg["alpha"])
    buf.push_rows(z["states"].astype(np.float32), z["actions"], z["rewards"].astype(np.float32),
                  z["next_states"].astype(np.float32), z["dones"].astype(np.float32))
    # install the reference's pre-update priorities (fp64 values that are exact fp32 or 1.0)
    assert np.array_equal(z["prio_before"], z["prio_before"].astype(np.float32).astype(np.float64))
    buf.update_priorities(np.arange(n), z["prio_before"].astype(np.float32))
    idx, phys, w = buf.sample_indices(z["u"], beta=cfg["beta"])
    idx_ref, w_ref = per_numpy.reference_sample(z["prio_before"], z["u"], cfg["alpha"], cfg["beta"])
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    batch = buf.gather(phys)
    abs_td = L.compute_grad(batch, w)
    L.apply_grad(tau=cfg["tau"] if cfg["flavor"] == "per" else 1.0)
    buf.update_priorities(idx, abs_td)
    torch.cuda.synchronize()
    np.testing.assert_allclose(L.loss.item(), float(z["loss"]), rtol=5e-6)
    np.testing.assert_allclose(buf.logical_priorities(), z["prio_after"], rtol=1e-4, atol=1e-6)
    post = spec.unpack(L.policy.flat.cpu().numpy())
    tgt = spec.unpack(L.target.flat.cpu().numpy())
    m = spec.unpack(L.exp_avg.cpu().numpy())
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad_clipped/"))
    for k in post:
        gc = z["grad_clipped/" + k]
        gs = max(np.abs(gc).max(), 1e-3 * gmax)
        np.testing.assert_allclose(m[k], z["exp_avg/" + k], rtol=1e-4, atol=1e-5 * gs)
        quiet = np.abs(gc) < 1e-6                     # Adam's first step is lr*sign(g): noise-level g flips
        for got, ref in ((post[k], z["sd_post/" + k]), (tgt[k], z["tgt_post/" + k])):
            err = np.abs(got - ref)
            assert err[~quiet].max(initial=0) <= 2e-6, k
            assert err[quiet].max(initial=0) <= 2.0 * cfg["lr"], k


def test_clip_adam_polyak_matches_torch():
    """Several steps of clip + Adam + Polyak on random flat buffers vs torch.optim semantics."""
    from mcts_q_amd._lib import lib, ptr
    rng = np.random.RandomState(0)
    n = 4096 + 16 * 37
    p0 = rng.standard_normal(n).astype(np.float32)
    t0 = rng.standard_normal(n).astype(np.float32)
    p, t = torch.from_numpy(p0).cuda(), torch.from_numpy(t0).cuda()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    ws = torch.zeros(8192, dtype=torch.uint8, device="cuda")
    gn = torch.zeros(1, device="cuda")
    rp = {"w": torch.from_numpy(p0.copy())}
    rt = torch.from_numpy(t0.copy())
    rm, rv = {"w": torch.zeros(n)}, {"w": torch.zeros(n)}
    st = torch.cuda.current_stream().cuda_stream
    for step in range(1, 6):
        g = (rng.standard_normal(n) * (3.0 if step % 2 else 0.01)).astype(np.float32)   # clipped and unclipped steps
        gd = torch.from_numpy(g).cuda()
        lib.check(lib.mctsq_clip_adam_polyak(ptr(p), ptr(gd), ptr(m), ptr(v), ptr(t), n, step, 1e-3, 0.9, 0.999,
                                             1e-8, 10.0, 0.005, ptr(gn), ptr(ws), ws.numel(), st))
        grads = {"w": torch.from_numpy(g)}
        coef, total = torch_ref.clip_coef(grads, 10.0)
        torch_ref.adam_step(rp, {"w": grads["w"] * coef}, rm, rv, step)
        rt = 0.005 * rp["w"] + (1.0 - 0.005) * rt
        torch.cuda.synchronize()
        np.testing.assert_allclose(gn.item(), total, rtol=1e-6)
        np.testing.assert_allclose(p.cpu().numpy(), rp["w"].numpy(), rtol=2.4e-7, atol=2e-7)
        np.testing.assert_allclose(m.cpu().numpy(), rm["w"].numpy(), rtol=1e-6, atol=5e-8)   # fp32 rounding of O(0.3) terms
        np.testing.assert_allclose(v.cpu().numpy(), rv["w"].numpy(), rtol=2e-6, atol=1e-10)
        np.testing.assert_allclose(t.cpu().numpy(), rt.numpy(), rtol=2.4e-7, atol=2e-7)


@pytest.mark.parametrize("combo", ["gru/gru/gru", "conv/conv/mlp", "gru/conv/gru", "transformer/transformer/gru",
                                   "transformer/gru/mlp", "gru/gru/gru+sharp"])
@pytest.mark.parametrize("B", [5, 1024])
def test_gradient_matches_torch_autograd_batched(B, combo):
    """Ragged and BASELINE config-3 batch sizes vs autograd of the torch oracle, real target path.
    "gru/gru/gru+sharp": attention score weights scaled up so the GRU kernels take the running-max
    softmax path (gru_core.hpp) in the training forward as well."""
    from mcts_q_amd import NetSpec
    from mcts_q_amd.learner import DeviceLearner
    sharp = combo.endswith("+sharp")
    pe, pr, ga = combo.replace("+sharp", "").split("/")
    cfg = dict(price=pe, process=pr, gas=ga, act="relu", flavor="per")
    torch.manual_seed(7)
    net, tgt = torch_ref.RefQNet(cfg), torch_ref.RefQNet(cfg)
    if sharp:
        with torch.no_grad():
            for m in (net, tgt):
                m.price_encoder.attention_weights.weight *= 60.0
                m.process_encoder.attention_weights.weight *= 60.0
    spec = NetSpec.from_cfg(cfg)
    L = DeviceLearner(spec).load_state_dict({k: v.detach().numpy() for k, v in net.state_dict().items()},
                                            {k: v.detach().numpy() for k, v in tgt.state_dict().items()})
    if "transformer" in combo and B > 100:
        B = 83                                   # keeps the CPU autograd oracle to a few seconds
    g = torch.Generator().manual_seed(B)
    rows, nrows = torch.rand(B, 91, generator=g), torch.rand(B, 91, generator=g)
    a = torch.randint(0, 5, (B,), generator=g)
    r, d = torch.rand(B, generator=g), (torch.rand(B, generator=g) < 0.2).float()
    w = torch.rand(B, generator=g) * 0.9 + 0.1
    s, s2 = torch_ref.split_state_rows(rows.numpy()), torch_ref.split_state_rows(nrows.numpy())
    with torch.no_grad():
        y = r + 0.99 * tgt(*s2).max(1)[0] * (1 - d)
    loss, td, grads = torch_ref.td_loss_and_grads(net, s, a, y, w)
    batch = dict(sprice=s[0].cuda(), sprocess=s[1].cuda(), sgas=s[2].cuda(), nprice=s2[0].cuda(),
                 nprocess=s2[1].cuda(), ngas=s2[2].cuda(), act=a.int().cuda(), rew=r.cuda(), done=d.cuda())
    abs_td = L.compute_grad(batch, w.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(L.loss.item(), float(loss), rtol=1e-5)
    np.testing.assert_allclose(abs_td.cpu().numpy(), td.abs().numpy(), rtol=0, atol=2e-5)
    got = spec.unpack(L.grad.cpu().numpy())
    gmax = max(float(v.abs().max()) for v in grads.values())
    for k, ref in grads.items():
        gs = max(float(ref.abs().max()), 1e-3 * gmax)
        assert np.abs(got[k] - ref.numpy()).max() <= 2e-4 * gs, k


@pytest.mark.parametrize("case", [
    dict(price="gru", process="gru", gas="gru", seq=(1, 2, 3), action_dim=3),
    dict(price="gru", process="gru", gas="gru", seq=(32, 16, 7), action_dim=16),
    dict(price="gru", process="gru", gas="gru", seq=(25, 10, 3), hidden_units=64, hidden_layers=3),
    dict(price="conv", process="conv", gas="mlp", seq=(26, 1, 3), action_dim=7),
    dict(price="transformer", process="gru", gas="mlp", seq=(32, 1, 3), act="tanh"),
])
def test_gradient_nonstandard_shapes(case):
    """TD-loss gradient for window lengths at the ABI's limits, 1..16 actions and a head shape that takes the generic
    head kernels, against autograd of the torch oracle (ragged batch of 21)."""
    from mcts_q_amd import NetSpec
    from mcts_q_amd.learner import DeviceLearner
    case = dict(case)
    seq = case.pop("seq")
    cfg = {"act": "relu", "flavor": "per", "action_dim": 5, **case}
    torch.manual_seed(13)
    net, tgt = torch_ref.RefQNet(cfg), torch_ref.RefQNet(cfg)
    spec = NetSpec(price=cfg["price"], process=cfg["process"], gas=cfg["gas"], act=cfg["act"],
                   hidden_layers=cfg.get("hidden_layers", 2), hidden_units=cfg.get("hidden_units", 128),
                   action_dim=cfg["action_dim"], seq=seq)
    L = DeviceLearner(spec).load_state_dict({k: v.detach().numpy() for k, v in net.state_dict().items()},
                                            {k: v.detach().numpy() for k, v in tgt.state_dict().items()})
    B = 21
    g = torch.Generator().manual_seed(2)
    mk = lambda: (torch.rand(B, seq[0], 1, generator=g), torch.rand(B, seq[1], 6, generator=g), torch.rand(B, seq[2], 2, generator=g))
    s, s2 = mk(), mk()
    a = torch.randint(0, cfg["action_dim"], (B,), generator=g)
    r, d = torch.rand(B, generator=g), (torch.rand(B, generator=g) < 0.2).float()
    w = torch.rand(B, generator=g) * 0.9 + 0.1
    with torch.no_grad():
        y = r + 0.99 * tgt(*s2).max(1)[0] * (1 - d)
    loss, td, grads = torch_ref.td_loss_and_grads(net, s, a, y, w)
    batch = dict(sprice=s[0].cuda(), sprocess=s[1].cuda(), sgas=s[2].cuda(), nprice=s2[0].cuda(),
                 nprocess=s2[1].cuda(), ngas=s2[2].cuda(), act=a.int().cuda(), rew=r.cuda(), done=d.cuda())
    abs_td = L.compute_grad(batch, w.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(L.loss.item(), float(loss), rtol=1e-5)
    np.testing.assert_allclose(abs_td.cpu().numpy(), td.abs().numpy(), rtol=0, atol=2e-5)
    got = spec.unpack(L.grad.cpu().numpy())
    gmax = max(float(v.abs().max()) for v in grads.values())
    for k, ref in grads.items():
        gs = max(float(ref.abs().max()), 1e-3 * gmax)
        assert np.abs(got[k] - ref.numpy()).max() <= 2e-4 * gs, k


def _filled_replay(n=3000, cap=4096, seed=2):
    from mcts_q_amd.per import DevicePER
    g = torch.Generator().manual_seed(seed)
    buf = DevicePER(cap)
    buf.push_rows(torch.rand(n, 91, generator=g).numpy(), torch.randint(0, 5, (n,), generator=g).numpy(),
                  torch.rand(n, generator=g).numpy(), torch.rand(n, 91, generator=g).numpy(),
                  (torch.rand(n, generator=g) < 0.1).float().numpy())
    buf.update_priorities(np.arange(n), torch.rand(n, generator=g).numpy() + 0.01)
    return buf


@pytest.mark.parametrize("combo,B", [(("gru", "gru", "gru"), 100), (("conv", "gru", "mlp"), 100),
                                     (("gru", "gru", "gru"), 1024), (("gru", "gru", "gru"), 2048)])
def test_one_call_step_equals_the_separate_entry_points(combo, B):
    """mctsq_per_dqn_step (DQNModel.update as one library call) against sample_batch -> dqn_grad_prepared ->
    clip_adam_polyak_ex -> prepare -> per_update called one by one: parameters, Adam moments, target, priorities
    and loss bit for bit over three steps.  B = 100: ragged batch (not a multiple of the 16-row tile), every GRU
    window on 8-sample tiles in one launch; 1024: the long window on 8-sample tiles next to the short ones on
    16-sample tiles in one launch; 2048: the windows no longer fit one wave of workgroups - two lanes."""
    from mcts_q_amd import NetSpec
    from mcts_q_amd.learner import DeviceLearner
    cfg = dict(price=combo[0], process=combo[1], gas=combo[2], act="relu", flavor="per")
    torch.manual_seed(5)
    sd = {k: v.detach().numpy() for k, v in torch_ref.RefQNet(cfg).state_dict().items()}
    spec = NetSpec.from_cfg(cfg)
    A, Bm = DeviceLearner(spec).load_state_dict(sd), DeviceLearner(spec).load_state_dict(sd)
    ra, rb = _filled_replay(), _filled_replay()
    rng = np.random.RandomState(8)
    prev_prep = None
    for it in range(3):
        u = rng.random_sample(B)
        A.step(ra, u, tau=0.005)
        batch, w, idx, _ = rb.sample_batch(u)
        abs_td = Bm.compute_grad(batch, w)
        Bm.apply_grad(tau=0.005)
        rb.update_priorities(idx, abs_td)
        torch.cuda.synchronize()
        assert torch.equal(A.loss, Bm.loss) and float(A.loss) > 0
        for x, y in ((A.policy.flat, Bm.policy.flat), (A.target.flat, Bm.target.flat), (A.exp_avg, Bm.exp_avg),
                     (A.exp_avg_sq, Bm.exp_avg_sq), (A.grad, Bm.grad), (ra.prio, rb.prio), (ra.tree, rb.tree),
                     (A.grad_norm, Bm.grad_norm)):
            assert torch.equal(x, y)
        # deferred re-split: the one-call step re-splits the weight fragments at its START (helper lane, under the
        # sampling) and leaves them stale: what it holds now are the fragments of the parameters BEFORE this step
        if prev_prep is not None:
            for x, y in zip((A.policy._prep, A.target._prep, A._prep_bwd), prev_prep):
                assert torch.equal(x, y)
        assert A.policy._prep_version is None               # ... and it says so: the next user re-splits
        prev_prep = [t.clone() for t in (Bm.policy._prep, Bm.target._prep, Bm._prep_bwd)]
    assert A.step_count == Bm.step_count == 3
    # a leaf evaluation behind the step re-splits first (DeviceQNet._prepared) and sees the new parameters
    p, c, g = torch.rand(7, 25, 1).cuda(), torch.rand(7, 10, 6).cuda(), torch.rand(7, 3, 2).cuda()
    q1 = A.policy(p, c, g)["q"].clone()
    A.policy.mark_dirty()
    assert torch.equal(q1, A.policy(p, c, g)["q"])


def test_unprepared_gradient_entry_point_gives_the_same_bits():
    """mctsq_dqn_grad (splits the weights itself, into its workspace) == mctsq_dqn_grad_prepared."""
    import ctypes as C
    from mcts_q_amd import NetSpec
    from mcts_q_amd._lib import lib, ptr
    from mcts_q_amd.learner import DeviceLearner
    cfg = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
    torch.manual_seed(6)
    sd = {k: v.detach().numpy() for k, v in torch_ref.RefQNet(cfg).state_dict().items()}
    L = DeviceLearner(NetSpec.from_cfg(cfg)).load_state_dict(sd)
    buf = _filled_replay()
    batch, w, _, _ = buf.sample_batch(np.random.RandomState(1).random_sample(48))
    abs_td = L.compute_grad(batch, w)
    g_ref, loss_ref = L.grad.clone(), L.loss.clone()
    need = lib.check(lib.mctsq_dqn_grad_workspace_bytes(C.byref(L.desc), 48))
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    grad, loss, td = torch.empty_like(L.grad), torch.zeros(1, device="cuda"), torch.empty(48, device="cuda")
    lib.check(lib.mctsq_dqn_grad(C.byref(L.desc), ptr(L.policy.flat), ptr(L.target.flat), ptr(batch["sprice"]),
                                 ptr(batch["sprocess"]), ptr(batch["sgas"]), ptr(batch["nprice"]), ptr(batch["nprocess"]),
                                 ptr(batch["ngas"]), ptr(batch["act"]), ptr(batch["rew"]), ptr(batch["done"]), ptr(w),
                                 L.gamma, 1.0, 48, ptr(grad), ptr(loss), ptr(td), None, None, ptr(ws), ws.numel(),
                                 torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(grad, g_ref) and torch.equal(loss, loss_ref) and torch.equal(td, abs_td)


def test_half_tile_forward_and_gradient_are_bit_identical():
    """The gru/gru/gru forward runs on 8-sample half tiles when 16-sample tiles would leave half the CUs idle
    (fused_gru_head.hip: half_tiles).  Same arithmetic per row, so Q, argmax, priors, |td|, loss and the flat gradient
    (i.e. every record the backward reads) must not change by a bit: tools/fwd_rpl_check.py runs both modes in child
    processes (the mode is read once per process from MCTSQ_FWD_RPL) and compares the dumps."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fwd_rpl_check.py"), "21", "1000"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "B=21: identical" in r.stdout and "B=1000: identical" in r.stdout
