import sys, time, torch
sys.path.insert(0,'/root/repo')
from mcts_q_amd import NetSpec
from mcts_q_amd.qnet import DeviceQNet
from bench import seeded_net
cfg = dict(price="gru", process="gru", gas="gru", act="relu", flavor="per")
_, sd = seeded_net(cfg)
net = DeviceQNet(NetSpec.from_cfg(cfg)).load_state_dict(sd)
B=256
x=(torch.rand(B,25,1,device='cuda'),torch.rand(B,10,6,device='cuda'),torch.rand(B,3,2,device='cuda'))
out=net.forward(*x)
torch.cuda.synchronize()
for rep in range(3):
    t0=time.perf_counter()
    for _ in range(2000): net.forward(*x, out=out)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(f"host enqueue {(t1-t0)/2000*1e6:.2f} us/call, total {(t2-t0)/2000*1e6:.2f} us/call")
