#!/usr/bin/env python3
"""20 forwards of one encoder combo at one batch size (for rocprofv3): bench_one.py combo B"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcts_q_amd import NetSpec
from mcts_q_amd.qnet import DeviceQNet
from bench import seeded_net
combo, B = sys.argv[1], int(sys.argv[2])
pe, pr, ga = combo.split("/")
cfg = dict(price=pe, process=pr, gas=ga, act="relu", flavor="per")
_, sd = seeded_net(cfg)
net = DeviceQNet(NetSpec.from_cfg(cfg)).load_state_dict(sd)
x = [torch.rand(B, 25, 1).cuda(), torch.rand(B, 10, 6).cuda(), torch.rand(B, 3, 2).cuda()]
out = net(*x)
for _ in range(20):
    net(*x, out=out)
torch.cuda.synchronize()
