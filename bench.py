#!/usr/bin/env python3
"""bench.py - headline benchmark of the MCTS_Q hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input on every rank: the batched
leaf Q evaluation of BASELINE.json config 2 (gru/gru/gru, 4096 leaf windows per GPU, seeded
default-init weights), inputs resident in HBM before the timed region.  Ranks hold independent
trees/leaves (no data-path collective), so scaling is weak.  Rank 0 prints ONE JSON line; `value`
is whole-job leaf evaluations per second.

The second half of BASELINE's metric, PER-DQN train steps/s (config 3: 1M-entry replay, batch 1024
TD updates; data-parallel over ranks with one flat-gradient RCCL all-reduce), is timed in a second
region and reported under "train".  "roofline" is the dominant kernel of the leaf path timed live
with HIP events; "cpu_baseline" is the torch-CPU oracle on the host cores of the same box.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
HBM_PEAK_GBS = 8000.0


def forward_flops(spec):
    """Algorithmic forward FLOPs per leaf: (encoders, head)."""
    enc = 0
    for kind, S, F in zip((spec.price, spec.process, spec.gas), spec.seq, spec.feat_dims):
        if kind == "gru":
            enc += S * (2 * 192 * 64 + 2 * 192 * F) + S * 4 * 64
        elif kind == "conv":
            enc += S * (2 * 64 * F * 3 + 2 * 64 * 64 * 3) + S * 4 * 64
        elif kind == "mlp":
            enc += 2 * 64 * 3 * F + 2 * 64 * 64
        elif kind == "transformer":
            per_tok = 2 * 64 * F + 2 * (2 * 192 * 64 + 2 * 64 * 64 + 4 * 64 * 2048) + 2 * 4 * S * 64
            enc += S * per_tok
    head, d = 0, 192
    for _ in range(spec.hidden_layers):
        head += 2 * d * spec.hidden_units
        d = spec.hidden_units
    head += 2 * d * spec.action_dim
    return enc, head


def seeded_net(cfg, seed=3654):
    """Random-init weights of the configured architecture (same torch.nn containers and seed as
    DQNModel(seed=3654); there are no checkpoints to load)."""
    from oracle import torch_ref
    torch.manual_seed(seed)
    net = torch_ref.RefQNet(cfg)
    return net, {k: v.detach().numpy() for k, v in net.state_dict().items()}


def cpu_baseline(cfg, ref_net, B, train_B, replay_entries, budget_s=20.0):
    """Torch-CPU oracle (the arithmetic the reference runs) on the host cores: batched forwards at
    the GPU's batch size, the as-called pattern (one leaf per call, mcts_per.py:367-370), and a
    few full PER-DQN updates at the train batch size."""
    from oracle import per_numpy, torch_ref
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    price, process, gas = torch.rand(B, 25, 1, generator=g), torch.rand(B, 10, 6, generator=g), \
        torch.rand(B, 3, 2, generator=g)
    with torch.no_grad():
        ref_net(price, process, gas)                              # warm-up
        times, t_end = [], time.perf_counter() + budget_s * 0.35
        while len(times) < 5 or (time.perf_counter() < t_end and len(times) < 30):
            t0 = time.perf_counter(); ref_net(price, process, gas); times.append(time.perf_counter() - t0)
        batched = B / float(np.median(times))
        p1, q1, g1 = price[:1], process[:1], gas[:1]
        ref_net(p1, q1, g1)
        n1, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s * 0.15:
            ref_net(p1, q1, g1).max().item(); n1 += 1
        single = n1 / (time.perf_counter() - t0)
    # train: reference formulation of PER sampling over N priorities + one update, torch CPU
    N = replay_entries                      # same replay size as the GPU side (BASELINE config 3: 1M entries)
    rng = np.random.RandomState(3654)
    prio = np.abs(rng.standard_normal(N))
    rows, nrows = rng.random_sample((N, 91), ).astype(np.float32), rng.random_sample((N, 91)).astype(np.float32)
    acts, rews = rng.randint(0, 5, N), rng.random_sample(N).astype(np.float32)
    tgt = torch_ref.RefQNet.from_state_dict(cfg, {k: v.detach().numpy() for k, v in ref_net.state_dict().items()})
    net = torch_ref.RefQNet.from_state_dict(cfg, {k: v.detach().numpy() for k, v in ref_net.state_dict().items()})
    m = {k: torch.zeros_like(p) for k, p in net.named_parameters()}
    v = {k: torch.zeros_like(p) for k, p in net.named_parameters()}
    tt = []
    for step in range(1, 4):
        t0 = time.perf_counter()
        idx, w = per_numpy.reference_sample(prio, rng.random_sample(train_B))
        out = torch_ref.dqn_update(net, tgt, m, v, step, (rows[idx], acts[idx], rews[idx], nrows[idx],
                                                          np.zeros(train_B, np.float32)), w)
        prio[idx] = out["new_prio"]
        torch_ref.polyak(net, tgt)
        tt.append(time.perf_counter() - t0)
        if sum(tt) > budget_s * 0.5:
            break
    return {"value": batched, "unit": "leaf evals/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} batched forwards of B={B} (median) + {n1} single-leaf calls + {len(tt)} "
                      f"PER-DQN updates of B={train_B} over a {N}-entry buffer; oracle/torch_ref (stock torch CPU "
                      "ops) + oracle/per_numpy",
            "as_called_b1_evals_per_s": single, "train_steps_per_s": 1.0 / float(np.median(tt))}


def profile_numbers():
    """Numbers that need rocprofv3 and therefore come from committed profiles of this same command
    (tools/profile_bench.sh), newest round first: HBM bytes per launch of the dominant kernel from the PMC passes
    (FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM) and the share of SIMD time its matrix pipe is busy."""
    out = {"traffic": None, "traffic_source": None, "mfma_busy": None, "mfma_busy_source": None}
    for rnd in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
        if out["traffic"] is None and os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            out["traffic"] = d.get("hbm_bytes_per_launch")
            out["traffic_source"] = f"profiles/{rnd}_pmc_traffic.json"
        path = os.path.join(ROOT, "profiles", f"{rnd}_sq_counters.json")
        if out["mfma_busy"] is None and os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            out["mfma_busy"] = d.get("mfma_busy_frac")
            out["mfma_busy_source"] = f"profiles/{rnd}_sq_counters.json"
    return out


BF16_MATRIX_PEAK_TFLOPS = 2500.0     # dense, MI355X_MICROARCH.md (fp16 MFMA: same rate)
SPLIT2_FP16_CEILING = BF16_MATRIX_PEAK_TFLOPS / 3.0      # two-way fp16 split: three products per fp32 product = 833
SPLIT3_BF16_CEILING = BF16_MATRIX_PEAK_TFLOPS / 6.0      # three-way bf16 split: six products = 417


def flop_classes(spec):
    """Forward FLOPs per leaf by the matrix instruction that EXECUTES them:
    'fp16x3'  two-way fp16 split (GRU recurrence, the conv encoder's second convolution, the transformer FFN and the
              attention block's in_proj / out_proj),
    'bf16x6'  exact three-way bf16 split (the 192-128-128-A head),
    'fp32'    v_mfma_f32_16x16x4_f32 and vector arithmetic (input projections, conv1, embedding, per-head attention,
              the gas/EUA MLP, pooling)."""
    c = {"fp16x3": 0, "bf16x6": 0, "fp32": 0}
    for kind, S, F in zip((spec.price, spec.process, spec.gas), spec.seq, spec.feat_dims):
        if kind == "gru":
            c["fp16x3"] += S * 2 * 192 * 64
            c["fp32"] += S * 2 * 192 * F + S * 4 * 64
        elif kind == "conv":
            c["fp16x3"] += S * 2 * 64 * 64 * 3
            c["fp32"] += S * 2 * 64 * F * 3 + S * 4 * 64
        elif kind == "mlp":
            c["fp32"] += 2 * 64 * 3 * F + 2 * 64 * 64
        elif kind == "transformer":
            c["fp16x3"] += S * 2 * (4 * 64 * 2048) + S * 2 * (2 * 192 * 64 + 2 * 64 * 64)
            c["fp32"] += S * (2 * 64 * F + 2 * 4 * S * 64)
    enc, head = forward_flops(spec)
    assert sum(c.values()) == enc
    shipped_head = spec.hidden_layers == 2 and spec.hidden_units == 128
    c["bf16x6" if shipped_head else "fp32"] += head
    return c


def executed_pipe_peak(spec):
    """Ceiling of the instructions the forward actually issues, in fp32-equivalent TFLOP/s: the classes of
    flop_classes() weighted by the time each would take at its own peak (833 / 417 / 157.3).  Returns (ceiling, share
    of the FLOPs on the 16-bit matrix instructions)."""
    c = flop_classes(spec)
    total = sum(c.values())
    t = c["fp16x3"] / SPLIT2_FP16_CEILING + c["bf16x6"] / SPLIT3_BF16_CEILING + c["fp32"] / FP32_MATRIX_PEAK_TFLOPS
    return total / t, (c["fp16x3"] + c["bf16x6"]) / total


DOMINANT_KERNEL = {"gru/gru/gru": "gru3_head_fwd_kernel<false, 4>", "conv/conv/mlp": "conv_enc_fwd_kernel",
                   "transformer/transformer/gru": "tf_ffn_pair_kernel"}
TRAIN_DOMINANT = {"gru/gru/gru": "gru3_head_fwd_pair_kernel<2>", "conv/conv/mlp": "conv_enc_bwd_kernel",
                  "transformer/transformer/gru": "tf_ffn_bwd_kernel"}


def other_configs(args, dev, buf, settle_fn, steps, warmup):
    """The BASELINE configurations beside the headline one, each on one GPU (rank 0): forward of conv/conv/mlp at 4096
    leaves (config 1's network at config 2's batch) and of transformer/transformer/gru at 16384 leaves (config 4), and
    one PER-DQN step each for conv and transformer at B = 1024 over the 1M-entry replay (config 3's step on the other
    encoder families).  Fractions are quoted against the fp32-MFMA peak on ALGORITHMIC FLOPs (can exceed 1 for split
    16-bit execution: labelled as such, not a roofline fraction) and against the executed-pipe ceiling (the roofline
    fraction proper)."""
    from mcts_q_amd import NetSpec
    from mcts_q_amd.hipevents import HipEvent
    from mcts_q_amd.learner import DeviceLearner
    from mcts_q_amd.qnet import DeviceQNet
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = []
    for combo, B, do_train in (("conv/conv/mlp", 4096, True), ("transformer/transformer/gru", 16384, True)):
        pe, pr, ga = combo.split("/")
        cfg = dict(price=pe, process=pr, gas=ga, act="relu", flavor="per")
        spec = NetSpec.from_cfg(cfg)
        _, sd = seeded_net(cfg)
        net = DeviceQNet(spec, dev).load_state_dict(sd)
        g = torch.Generator().manual_seed(7)
        x = [torch.rand(B, 25, 1, generator=g).to(dev), torch.rand(B, 10, 6, generator=g).to(dev),
             torch.rand(B, 3, 2, generator=g).to(dev)]
        o = net(*x)
        t_end = time.perf_counter() + args.settle_ms * 1e-3        # clock settle under THIS kernel mix
        while time.perf_counter() < t_end:
            for _ in range(10):
                net(*x, out=o)
            torch.cuda.synchronize()
        n = max(10, min(steps, 50)) if B >= 16384 else max(20, min(steps, 200))
        for _ in range(max(2, warmup // 4)):
            net(*x, out=o)
        e0, e1 = HipEvent(), HipEvent()
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(n):
            net(*x, out=o)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_ms(e1) / n
        enc_fl, head_fl = forward_flops(spec)
        flops = (enc_fl + head_fl) * B
        ach = flops / (ms * 1e-3) / 1e12
        ceil, share = executed_pipe_peak(spec)
        rec = {"workload": f"{combo} leaf evaluation, {B} leaves", "encoders": combo, "batch": B, "steps": n,
               "ms_per_forward": ms, "value": B / (ms * 1e-3), "unit": "leaf evals/s",
               "dominant_kernel": DOMINANT_KERNEL[combo], "flops_per_leaf": enc_fl + head_fl, "flops": flops,
               "achieved": ach, "achieved_unit": "TFLOP/s (algorithmic fp32 FLOPs)",
               "executed_pipe_peak": ceil, "frac_executed_pipe": ach / ceil, "split_16bit_share_of_flops": share,
               "algorithmic_over_fp32_mfma_peak": ach / FP32_MATRIX_PEAK_TFLOPS,
               "flop_classes_per_leaf": flop_classes(spec),
               "hbm_GBs_algorithmic": (91 * 4 + 5 * 4 + 4 + 4 + 5 * 4) * B / (ms * 1e-3) / 1e9}
        if do_train and buf is not None:
            Bt = args.train_batch
            L = DeviceLearner(spec, dev).load_state_dict(sd)
            tsteps = max(10, min(steps, 30 if "transformer" in combo else 100))
            u_all = torch.from_numpy(np.random.RandomState(11).random_sample((tsteps + 5, Bt))).to(dev)
            for i in range(5):
                L.step(buf, u_all[i], tau=0.005)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(tsteps):
                L.step(buf, u_all[5 + i], tau=0.005)
            torch.cuda.synchronize()
            tw = (time.perf_counter() - t0) / tsteps
            sf = 4 * (enc_fl + head_fl) * Bt
            rec["train"] = {"workload": f"PER-DQN step, {combo}, batch {Bt}, {buf.size}-entry replay", "steps": tsteps,
                            "ms_per_step": tw * 1e3, "value": 1.0 / tw, "unit": "train steps/s",
                            "dominant_kernel": TRAIN_DOMINANT[combo], "flops_per_step": sf,
                            "achieved": sf / tw / 1e12, "achieved_unit": "TFLOP/s (algorithmic: 4 x forward FLOPs)",
                            "frac_executed_pipe": sf / tw / 1e12 / ceil,
                            "executed_pipe_peak": ceil,
                            "note": "ceiling = the FORWARD's instruction mix; the backward runs its GRU / head products on "
                                    "the six-term bf16 split and the weight gradients on fp32 MFMA, so this overstates "
                                    "the ceiling and understates the fraction",
                            "loss_last": float(L.loss.item())}
            del L
        out.append(rec)
        del net
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=300.0, dest="settle_ms",
                    help="untimed device load ahead of the warm-up steps, until the clocks have settled")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--train-batch", type=int, default=1024, help="GLOBAL train batch (split over ranks)")
    ap.add_argument("--replay", type=int, default=1_000_000, help="replay entries per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-trees", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the conv / transformer configurations")
    ap.add_argument("--trees", type=int, default=512, help="MCTS trees per rank (BASELINE config 5)")
    ap.add_argument("--backend", default="nccl", help="collective backend (nccl = RCCL; gloo only to rehearse "
                                                     "several ranks on one GPU)")
    args = ap.parse_args()
    if os.environ.get("MCTSQ_BENCH_DEBUG"):          # dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MCTSQ_BENCH_DEBUG"]), exit=True)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU) as a CHILD job, before
        # anything in this process touches the GPU, and relay its one JSON line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from mcts_q_amd import NetSpec, parallel
    from mcts_q_amd.qnet import DeviceQNet
    from mcts_q_amd.hipevents import HipEvent

    cfg = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
    spec = NetSpec.from_cfg(cfg)
    ref_net, sd = seeded_net(cfg)
    net = DeviceQNet(spec, dev).load_state_dict(sd)
    B = args.batch
    g = torch.Generator().manual_seed(rank)          # each rank evaluates its own trees' leaves
    price = torch.rand(B, 25, 1, generator=g).to(dev)
    process = torch.rand(B, 10, 6, generator=g).to(dev)
    gas = torch.rand(B, 3, 2, generator=g).to(dev)
    out = net(price, process, gas)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=dev)
            parallel.all_reduce(t, dist.ReduceOp.MAX, dist.group.WORLD)
            return float(t.item())
        return x

    # clock settle: the device ramps its clocks over the first tenths of a second of load; a 20-step run on a cold
    # device read 43 us per launch against 39.5-41 in steady state.  Untimed, like the W warm-up steps behind it.
    def settle():
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for _ in range(50):
                net(price, process, gas, out=out)
            torch.cuda.synchronize()

    settle()
    for _ in range(args.warmup):
        net(price, process, gas, out=out)
    # --- timed region: exactly K steps ---------------------------------------------------------
    ev0, ev1 = HipEvent(), HipEvent()
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        net(price, process, gas, out=out)
    ev1.record(stream)
    barrier()
    wall = max_over_ranks(time.perf_counter() - t0)
    gpu_ms_total = ev0.elapsed_ms(ev1)

    # --- dominant kernel (fused GRU encoders + head: the whole forward is one launch) timed live
    # with HIP events on the launch stream ---------------------------------------------------------
    reps = max(20, min(args.steps, 200))
    evs = [(HipEvent(), HipEvent()) for _ in range(reps)]
    for a, b in evs:
        a.record(stream)
        net(price, process, gas, out=out, mid_event=b.ptr)
    torch.cuda.synchronize()
    k_single_ms = float(np.median([a.elapsed_ms(b) for a, b in evs]))   # one launch between two events: + ~2 us of event packets
    # the kernel's average launch duration over the TIMED region (two events around its K back-to-back launches on the
    # launch stream): the figure rocprofv3's per-kernel average of the same command agrees with
    k_ms = gpu_ms_total / args.steps
    enc_fl, head_fl = forward_flops(spec)
    flops = (enc_fl + head_fl) * B
    achieved = flops / (k_ms * 1e-3) / 1e12
    alg_bytes = (91 * 4 + 5 * 4 + 4 + 4 + 5 * 4) * B             # windows in; Q, vmax, argmax, prior out
    prof = profile_numbers()
    pipe_peak, bf16_share = executed_pipe_peak(spec)
    roofline = {"bound": "mfma", "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP32_MATRIX_PEAK_TFLOPS, "traffic": prof["traffic"],
                "traffic_source": prof["traffic_source"],
                "kernel": "gru3_head_fwd_kernel<false, 4>", "kernel_ms": k_ms,
                "kernel_ms_single_launch_between_events": k_single_ms, "flops_per_launch": flops,
                "algorithmic_bytes_per_launch": alg_bytes,
                "hbm_GBs_algorithmic": alg_bytes / (k_ms * 1e-3) / 1e9,
                "frac_executed_pipe": achieved / pipe_peak, "executed_pipe_peak": pipe_peak,
                "split_16bit_share_of_flops": bf16_share,
                "mfma_busy_measured": prof["mfma_busy"], "mfma_busy_source": prof["mfma_busy_source"],
                "note": "intensity ~2.5 kFLOP/B: matrix/vector issue bounds it, not HBM.  `frac` = ALGORITHMIC fp32 "
                        "FLOPs / time against the dense fp32 MFMA peak (157.3).  The recurrent products execute as three "
                        "fp16 MFMAs on a two-way split of operands scaled into the fp16 range (ceiling 2500/3 = 833 "
                        "TFLOP/s fp32-equivalent), the head as six bf16 MFMAs on an exact 3-way split (2500/6 = 417), "
                        "both fp32-accurate; `frac_executed_pipe` prices the kernel against those ceilings blended "
                        "with the fp32-MFMA share (input projections)"}

    # --- the same K steps with the D2H of value / greedy action inside the timed region: the kernel's
    # epilogue stores vmax / argmax straight into pinned (device-mapped) host memory, so the transfer is part of the
    # launch and needs no copy of its own (an async copy per step was measured at +47 us per step) -----------
    host_va = torch.empty(2 * B, dtype=torch.float32, pin_memory=True)
    out_d = dict(q=out["q"], vmax=host_va[:B], argmax=host_va[B:].view(torch.int32), prior=out["prior"])
    for _ in range(max(3, args.warmup // 4)):
        net(price, process, gas, out=out_d)
    barrier()
    assert torch.equal(host_va[:B], out["vmax"].cpu()) and torch.equal(host_va[B:].view(torch.int32), out["argmax"].cpu())
    t0d = time.perf_counter()
    for _ in range(args.steps):
        net(price, process, gas, out=out_d)
    barrier()
    wall_d2h = max_over_ranks(time.perf_counter() - t0d)

    # --- PER-DQN train steps (config 3), data-parallel over ranks ---------------------------------
    train = train_weak = buf = None
    if not args.no_train:
        from mcts_q_amd.learner import DeviceLearner
        from mcts_q_amd.per import DevicePER
        N = args.replay
        group = dist.group.WORLD if world > 1 else None
        L = DeviceLearner(spec, dev, process_group=group).load_state_dict(sd)
        buf = DevicePER(N, device=dev, mass_capacity=N * world)
        gen = torch.Generator(device=dev).manual_seed(100 + rank)
        chunk = 250_000
        for s0 in range(0, N, chunk):
            k = min(chunk, N - s0)
            buf.push_rows(torch.rand(k, 91, device=dev, generator=gen),
                          torch.randint(0, 5, (k,), device=dev, generator=gen),
                          torch.rand(k, device=dev, generator=gen), torch.rand(k, 91, device=dev, generator=gen),
                          torch.zeros(k, device=dev))
        pr = np.abs(np.random.RandomState(3654 + rank).standard_normal(N)).astype(np.float32)
        pr[: N // 100] = 1.0
        buf.update_priorities(torch.arange(N, device=dev), torch.from_numpy(pr).to(dev))
        tsteps = max(10, min(args.steps, 100))
        twarm = max(3, min(args.warmup, 10))

        def run_train(Bg, kind):
            # the SAME uniforms on every rank (shared seed): each rank serves the draws in its mass interval
            u_all = torch.from_numpy(np.random.RandomState(3654).random_sample((tsteps + twarm, Bg))).to(dev)
            bcap = parallel.draw_capacity(Bg, world)

            def train_step(i):
                if world == 1:
                    L.step(buf, u_all[i], tau=0.005)                       # DQNModel.update() + target sync: one call
                else:
                    # two library calls around ONE all-reduce over [gradient | loss | weight maxima], plus the
                    # all-gather of the shards' (mass, size) pairs; no host read of device memory
                    L.step_sharded(buf, u_all[i], tau=0.005, bcap=bcap)

            settle()                                           # the replay set-up above left the device idle
            for i in range(twarm):
                train_step(i)
            tev0, tev1 = HipEvent(), HipEvent()
            barrier()
            t0 = time.perf_counter()
            tev0.record(stream)
            for i in range(tsteps):
                train_step(twarm + i)
            tev1.record(stream)
            barrier()
            twall = max_over_ranks(time.perf_counter() - t0)
            enc_fl, head_fl = forward_flops(spec)
            step_flops = 4 * (enc_fl + head_fl) * Bg           # policy fwd + 2 x bwd + target fwd
            rows = Bg if world == 1 else bcap
            act_bytes = sum(S * rows * 64 * 4 * 5 for S in spec.seq)      # r, z, n, W_hn h, h per GRU step: written, read back
            # + the per-workgroup weight-gradient partials of the GRU backward (one workgroup per 8 rows of the longest
            # window, per 16 rows of the others; ~12.7 k floats each, written and read by the reduction), the batch
            # rows in and out of the sampler, parameters / moments / target / gradient through the optimizer
            wg = sum((rows // 8 if S == max(spec.seq) else rows // 16) * (192 * 8 + 192 + 128 * 65 + 64 * 65) for S in spec.seq)
            step_bytes = 2 * act_bytes + 2 * 4 * wg + 2 * 364 * rows + 16 * spec.n_flat * 4
            ach = step_flops / (twall / tsteps) / 1e12
            return {"metric": "PER-DQN train steps/sec (gru/gru/gru, 1M-entry sum-tree replay per rank, "
                              f"global batch {Bg})", "value": tsteps / twall, "unit": "train steps/s",
                    "steps": tsteps, "warmup": twarm, "ms_per_step": twall / tsteps * 1e3,
                    "gpu_ms_per_step_events_rank0": tev0.elapsed_ms(tev1) / tsteps,
                    "global_batch": Bg, "rows_per_rank": rows, "replay_entries_per_gpu": N,
                    "scaling": kind, "transitions_per_s": tsteps * Bg / twall, "loss_last": float(L.loss.item()),
                    "collectives_per_step": (0 if world == 1 else 2),
                    "roofline": {"bound": "latency chain (64 tiles of 16 transitions on 256 CUs; see DESIGN 5b)",
                                 "flops_per_step": step_flops, "achieved": ach, "unit": "TFLOP/s",
                                 "peak": FP32_MATRIX_PEAK_TFLOPS, "frac": ach / FP32_MATRIX_PEAK_TFLOPS,
                                 "frac_executed_pipe": ach / executed_pipe_peak(spec)[0],
                                 "activation_record_bytes": act_bytes, "algorithmic_bytes_per_step": step_bytes,
                                 "hbm_GBs_algorithmic": step_bytes / (twall / tsteps) / 1e9,
                                 "hbm_frac": step_bytes / (twall / tsteps) / 1e9 / HBM_PEAK_GBS,
                                 "per_kernel": "profiles/r03_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this "
                                               "program), profiles/r03_pmc_traffic.json, profiles/r03_sq_counters.json"}}

        train = run_train(args.train_batch, "strong (global batch fixed; replay shards sampled as one buffer, each rank "
                          "back-propagates its own draws; one flat-gradient all-reduce)" if world > 1 else "single GPU")
        if world > 1:
            train_weak = run_train(args.train_batch * world, "weak (global batch = %d x ranks)" % args.train_batch)

    # --- the other BASELINE configurations (conv / transformer forward, their PER-DQN steps), rank 0 only -----------
    configs = None
    if not args.no_configs and rank == 0:
        configs = other_configs(args, dev, buf if not args.no_train else None, settle, args.steps, args.warmup)
    if world > 1:
        dist.barrier()

    # --- lock-step MCTS in situ (config 5 shape per rank: `trees` trees x 50 iterations): the C search iteration over
    # the vectorised PtG environment (real step arithmetic on synthetic tables of the real sizes), leaves staged in
    # pinned memory, ONE forward launch per block of trees and iteration, two blocks alternating between host and
    # device; search only, weights fixed -------------------------------------------------------------------
    trees = None
    if not args.no_trees:
        from mcts_q_amd.mcts import DeviceLeafEvaluator
        from mcts_q_amd.ptgenv import PtGBatchEnv
        from mcts_q_amd.synthetic_env import synthetic_ptg_kwargs
        from mcts_q_amd.treesearch import LockstepMCTS

        benv = PtGBatchEnv(synthetic_ptg_kwargs(0, scale=1.0, train_days=46, env_values=dict(eps_len_d=40)), seed=rank)
        K, iters, tsteps = args.trees, 50, 10
        halves = 2 if K >= 64 else 1
        drv = LockstepMCTS(K, DeviceLeafEvaluator(net, -(-K // halves), halves, zero_copy=True), iters, 42, 19652, 1.25,
                           batch_env=benv, seed=rank)
        st = benv.reset(np.arange(K) + 100000 * rank)
        a = drv.predict(st, train=False)                      # warm-up search
        barrier()
        t0 = time.perf_counter()
        for _ in range(tsteps):
            st = benv.step_batch(st, a)[0]
            a = drv.predict(st, train=False)
        barrier()
        swall = max_over_ranks(time.perf_counter() - t0)
        trees = {"metric": "leaf evals/s inside the lock-step tree search (C search iteration + vectorised PtG environment "
                           "step on the host, one leaf batch per block of trees and iteration on the device)",
                 "trees_per_gpu": K, "iterations": iters, "searches": tsteps, "s_per_search_of_all_trees": swall / tsteps,
                 "value": world * tsteps * K * (iters + 1) / swall, "unit": "leaf evals/s", "scaling": "weak",
                 "with_learner_and_replay_in_the_loop": "tools/tree_bench.py 512 5 50 ptg0 (DESIGN.md section 7)"}

    if rank == 0:
        res = {
            "metric": "Q leaf evals/sec (gru/gru/gru, batch 4096 per GPU)",
            "value": world * B * args.steps / wall, "unit": "leaf evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "clock_settle_ms": args.settle_ms,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 via split 16-bit MFMA (fp16 x2 planes / 3 products for the recurrence, bf16 x3 planes / 6 products "
                     "for the head; fp32 accumulate; ~22-bit products, measured 4e-7..1.1e-6 of max|Q| against fp64)",
            "data": "synthetic U[0,1) windows, seeded default-init weights",
            "config": {"workload": "BASELINE config 2: GRU encoder batched leaf evaluation, "
                                   f"{B} price/process/gas windows per GPU", "encoders": "gru/gru/gru",
                       "batch_per_gpu": B, "parallelism": f"trees sharded over {world} GPU(s), no collective"},
            "gpu_ms_per_step_events": gpu_ms_total / args.steps,
            "with_d2h_of_value_and_action": {"value": world * B * args.steps / wall_d2h, "unit": "leaf evals/s",
                                             "ms_per_step": wall_d2h / args.steps * 1e3,
                                             "note": "same K steps with vmax / argmax (32 KB per step) stored by the "
                                                     "kernel's epilogue into pinned host memory: the D2H is inside "
                                                     "the launch, results checked against the device copy"},
            "collective_backend": (dist.get_backend() if world > 1 else None),
            "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
            "roofline": roofline,
        }
        if train is not None:
            res["train"] = train
        if train_weak is not None:
            res["train_weak"] = train_weak
        if trees is not None:
            res["tree_search"] = trees
        if configs is not None:
            res["configs"] = configs
        if not args.no_cpu_baseline and world == 1:          # the CPU path is timed beside the one-GPU run only
            res["cpu_baseline"] = cpu_baseline(cfg, ref_net, B, args.train_batch, args.replay)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
