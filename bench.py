#!/usr/bin/env python3
"""bench.py - headline benchmark of the MCTS_Q hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input on every rank:
the batched leaf Q evaluation of BASELINE.json config 2 (gru/gru/gru, 4096 leaf windows per GPU,
weights from a seeded default init), inputs resident in HBM before the timed region.  Ranks hold
independent trees/leaves (no data-path collective), so scaling is weak.  Rank 0 prints ONE JSON
line; `value` is whole-job leaf evaluations per second.  Extra keys report the PER-DQN train-step
rate (config 3) when the learner path is built, the roofline of the dominant kernel measured with
HIP events, and the CPU oracle timed on the host cores of the same box.
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


def encoder_flops(spec):
    """Algorithmic forward FLOPs per leaf of the encoder stage and of the head."""
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


def seeded_state_dict(cfg, seed=3654):
    """Random-init weights of the configured architecture (DQNModel(seed=3654) equivalent init
    through the same torch.nn containers; there are no checkpoints to load)."""
    from oracle import torch_ref
    torch.manual_seed(seed)
    net = torch_ref.RefQNet(cfg)
    return net, {k: v.detach().numpy() for k, v in net.state_dict().items()}


def cpu_baseline(ref_net, B, budget_s=12.0):
    """Torch-CPU oracle (the arithmetic the reference runs) on the host cores: batched pattern at
    the GPU's batch size and the as-called pattern (one leaf per call, mcts_per.py:367-370)."""
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    price, process, gas = torch.rand(B, 25, 1, generator=g), torch.rand(B, 10, 6, generator=g), \
        torch.rand(B, 3, 2, generator=g)
    with torch.no_grad():
        ref_net(price, process, gas)                              # warm-up
        times, t_end = [], time.perf_counter() + budget_s * 0.6
        while len(times) < 10 or (time.perf_counter() < t_end and len(times) < 30):
            t0 = time.perf_counter(); ref_net(price, process, gas); times.append(time.perf_counter() - t0)
        batched = B / float(np.median(times))
        p1, q1, g1 = price[:1], process[:1], gas[:1]
        ref_net(p1, q1, g1)
        n1, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s * 0.3:
            ref_net(p1, q1, g1).max().item(); n1 += 1
        single = n1 / (time.perf_counter() - t0)
    return {"value": batched, "unit": "leaf evals/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} batched forwards of B={B} (median) + {n1} single-leaf calls; "
                      "oracle/torch_ref.RefQNet (stock torch CPU ops)",
            "as_called_b1_evals_per_s": single}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from mcts_q_amd import NetSpec
    from mcts_q_amd.qnet import DeviceQNet
    from mcts_q_amd.hipevents import HipEvent

    cfg = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
    spec = NetSpec.from_cfg(cfg)
    ref_net, sd = seeded_state_dict(cfg)
    net = DeviceQNet(spec, dev).load_state_dict(sd)
    B = args.batch
    g = torch.Generator().manual_seed(rank)          # each rank evaluates its own trees' leaves
    price = torch.rand(B, 25, 1, generator=g).to(dev)
    process = torch.rand(B, 10, 6, generator=g).to(dev)
    gas = torch.rand(B, 3, 2, generator=g).to(dev)
    out = net(price, process, gas)
    stream = torch.cuda.current_stream(dev).cuda_stream

    for _ in range(args.warmup):
        net(price, process, gas, out=out)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # --- timed region: exactly K steps -----------------------------------------------------
    ev0, ev1 = HipEvent(), HipEvent()
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        net(price, process, gas, out=out)
    ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    gpu_ms_total = ev0.elapsed_ms(ev1)

    # --- dominant kernel (fused GRU encoders) timed live with HIP events, same stream ------
    reps = max(20, min(args.steps, 200))
    evs = [(HipEvent(), HipEvent()) for _ in range(reps)]
    for a, b in evs:
        a.record(stream)
        net(price, process, gas, out=out, mid_event=b.ptr)
    torch.cuda.synchronize()
    enc_ms = float(np.median([a.elapsed_ms(b) for a, b in evs]))
    enc_fl, head_fl = encoder_flops(spec)
    achieved = enc_fl * B / (enc_ms * 1e-3) / 1e12
    bytes_per_leaf = 91 * 4 + 192 * 4                    # windows in + feature row out (encoder kernel)
    roofline = {"bound": "mfma", "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP32_MATRIX_PEAK_TFLOPS, "traffic": None,
                "kernel": "gru_enc_fwd_kernel<false>", "kernel_ms": enc_ms,
                "flops_per_launch": enc_fl * B,
                "hbm_GBs_algorithmic": bytes_per_leaf * B / (enc_ms * 1e-3) / 1e9}

    if rank == 0:
        res = {
            "metric": "Q leaf evals/sec (gru/gru/gru, batch 4096 per GPU)",
            "value": world * B * args.steps / wall, "unit": "leaf evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic U[0,1) windows, seeded default-init weights",
            "config": {"workload": "BASELINE config 2: GRU encoder batched leaf evaluation, "
                                   f"{B} price/process/gas windows per GPU", "encoders": "gru/gru/gru",
                       "batch_per_gpu": B, "parallelism": f"trees sharded over {world} GPU(s), no collective"},
            "gpu_ms_per_step_events": gpu_ms_total / args.steps,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(ref_net, B)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
