"""BASELINE config 5 on the real kernels (SURVEY.md section 8e, section 4 tier 4): the data-parallel PER-DQN step with
MORE THAN ONE RANK.  Two fresh child processes (torch.distributed.run, gloo, both on cuda:0; the box has one GPU, so
RCCL itself is not exercised here) each own a replay shard and a learner replica and take three steps through
DeviceLearner.step_sharded = mctsq_per_dqn_step_begin -> ONE all-reduce over [gradient | loss | weight maxima] ->
mctsq_per_dqn_step_finish.  Checked:
  (a) the replicas end bit-identical (parameters, Adam moments, target network, loss);
  (b) against ONE process stepping over the concatenated buffer with the same uniforms (mctsq_per_dqn_step): the rows
      every rank serves are exactly the single buffer's draws that fall into its shard, the priorities written by the
      first step are bit-identical, the exchanged weight maximum is the single buffer's, and loss / gradient norm /
      parameters / moments agree to fp32 rounding (the folded form multiplies fp32(w_raw) by 1 / max instead of
      rounding w_raw / max, and sums the ranks' gradients in a different order)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _close(a, b, rel):
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) <= rel * scale


def _params_close(p, p_ref, t, t_ref, grad0, steps, lr=1e-3, tau=0.005):
    """Adam's first steps move a weight by ~lr whatever the size of its gradient, so where the gradient is rounding
    noise (|g| far below the largest entry) a last-bit difference decides the direction: there the bound is the
    distance Adam can travel; where the gradient is significant the parameters must agree to fp32 rounding of lr."""
    clear = np.abs(grad0) > 1e-3 * np.abs(grad0).max()
    dp = np.abs(p - p_ref)
    assert clear.sum() > 1000 and float(dp[clear].max()) <= 1e-3 * lr * steps, float(dp[clear].max())
    assert float(dp.max()) <= 2.0 * lr * steps
    assert float(np.abs(t - t_ref).max()) <= tau * steps * max(float(dp.max()), 1e-7) * 1.01


def _single_process_reference():
    import dp_worker as W
    from mcts_q_amd import NetSpec
    from mcts_q_amd.learner import DeviceLearner
    from mcts_q_amd.per import DevicePER
    dev = torch.device("cuda", 0)
    world = len(W.SIZES)
    L = DeviceLearner(NetSpec.from_cfg(W.CFG), dev).load_state_dict(W.seeded_weights())
    buf = DevicePER(W.CAP * world, device=dev)
    prios = []
    for r, n in enumerate(W.SIZES):
        rows, act, rew, nrows, done, prio = W.shard_content(r, n)
        buf.push_rows(rows, act, rew, nrows, done)
        prios.append(prio)
    buf.update_priorities(np.arange(sum(W.SIZES)), np.concatenate(prios))
    u_all = np.random.RandomState(5).random_sample((W.STEPS, W.B))
    out = []
    for i in range(W.STEPS):
        idx, _, w = buf.sample_indices(u_all[i])            # what the step is about to draw (sampling changes nothing)
        wmax, mass = float(buf.last_local_wmax.item()), int(buf.total_mass().item())
        L.step(buf, u_all[i], tau=W.TAU)
        torch.cuda.synchronize()
        out.append(dict(idx=idx.cpu().numpy(), w=w.cpu().numpy(), wmax=wmax, mass=mass, loss=float(L.loss.item()),
                        gnorm=float(L.grad_norm.item()), prio=buf.logical_priorities().copy(),
                        grad=L.grad.cpu().numpy().copy()))
    fin = dict(params=L.policy.flat.cpu().numpy(), target=L.target.flat.cpu().numpy(), m=L.exp_avg.cpu().numpy(),
               v=L.exp_avg_sq.cpu().numpy())
    return out, fin


def test_two_ranks_on_the_real_kernels_equal_the_single_process_step(tmp_path):
    import dp_worker as W
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dp_worker.py"), str(tmp_path)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    ranks = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2)]
    # (a) replicas bit-identical
    for k in ("params", "target", "m", "v"):
        assert np.array_equal(ranks[0][k], ranks[1][k]), k
    for i in range(W.STEPS):
        for k in ("loss", "gnorm", "exchange", "grad"):
            assert np.array_equal(ranks[0][f"{k}{i}"], ranks[1][f"{k}{i}"]), (k, i)
    # (b) against the single-process global-batch step
    steps, fin = _single_process_reference()
    offs = np.concatenate([[0], np.cumsum(W.SIZES)])
    for i, ref in enumerate(steps):
        served = np.zeros(W.B, dtype=np.int64)
        for r, z in enumerate(ranks):
            kept, dropped = z[f"count{i}"]
            mine = np.nonzero((ref["idx"] >= offs[r]) & (ref["idx"] < offs[r + 1]))[0]
            assert dropped == 0 and kept == len(mine), (i, r)
            d = z[f"draw{i}"]
            assert np.array_equal(d[:kept], mine) and (d[kept:] == -1).all()
            served[d[:kept]] += 1
        assert (served == 1).all()
        ms = ranks[0][f"masses{i}"]
        assert np.array_equal(ms, ranks[1][f"masses{i}"]) and ms[1].tolist() == list(W.SIZES)
        # steps 0 and 1 sample from bit-identical priorities (the parameters first differ, by rounding, after step 0,
        # so the |td| written by step 1 differ in the last bits): exact there, close afterwards
        exact = i <= 1
        msum = int(ms[0, 0]) + int(ms[0, 1])
        assert msum == ref["mass"] if exact else abs(msum - ref["mass"]) <= 1e-6 * ref["mass"], (i, ms, ref["mass"])
        ex = ranks[0][f"exchange{i}"].astype(np.float64)
        gmax = max(ex[0] + ex[1], ex[2] + ex[3])
        assert abs(gmax - ref["wmax"]) <= (1e-13 if exact else 1e-6) * ref["wmax"], (i, gmax, ref["wmax"])
        prio = np.concatenate([z[f"prio{i}"] for z in ranks])
        if i == 0:
            assert np.array_equal(prio, ref["prio"])            # same parameters, row-wise forward: same |td| bits
        else:
            assert _close(prio, ref["prio"], 2e-5)
        assert abs(float(ranks[0][f"loss{i}"][0]) - ref["loss"]) <= 2e-6 * abs(ref["loss"])
        assert abs(float(ranks[0][f"gnorm{i}"][0]) - ref["gnorm"]) <= 1e-5 * ref["gnorm"]
        assert _close(ranks[0][f"grad{i}"], ref["grad"], 1e-5 if i == 0 else 1e-4)
    assert _close(ranks[0]["m"], fin["m"], 1e-5) and _close(ranks[0]["v"], fin["v"], 1e-5)
    _params_close(ranks[0]["params"], fin["params"], ranks[0]["target"], fin["target"], steps[0]["grad"], W.STEPS)


def test_begin_finish_at_world_one_equals_the_one_call_step():
    """The two-call form alone in this process (world = 1: no collective): same draws, bit-identical priorities
    after the first step, parameters to fp32 rounding - and, with every draw its own, the (kept, dropped) words."""
    import dp_worker as W
    from mcts_q_amd import NetSpec
    from mcts_q_amd.learner import DeviceLearner
    dev = torch.device("cuda", 0)
    sd = W.seeded_weights()
    runs = []
    for mode in ("one", "two"):
        L = DeviceLearner(NetSpec.from_cfg(W.CFG), dev).load_state_dict(sd)
        buf = W.build_shard(0, 2500, 2500, 2500, dev)
        u_all = np.random.RandomState(8).random_sample((3, 128))
        pr = []
        for i in range(3):
            if mode == "one":
                L.step(buf, u_all[i], tau=W.TAU)
            else:
                L.step_sharded(buf, u_all[i], tau=W.TAU)
                assert L._dp["count"].tolist() == [128, 0]
            pr.append(buf.logical_priorities().copy())
            if i == 0:
                g0 = L.grad.cpu().numpy().copy()
        runs.append((L.policy.flat.cpu().numpy(), L.target.flat.cpu().numpy(), pr, float(L.loss.item()), g0))
    (p1, t1, pr1, l1, g1), (p2, t2, pr2, l2, _) = runs
    assert np.array_equal(pr1[0], pr2[0])
    assert _close(pr2[2], pr1[2], 2e-5) and abs(l1 - l2) <= 1e-5 * abs(l1)
    _params_close(p2, p1, t2, t1, g1, 3)
