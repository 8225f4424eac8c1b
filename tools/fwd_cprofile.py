#!/usr/bin/env python
"""cProfile of the trainer thread over whole training steps (forward, loss, optimizer; the backward pass runs on autograd's own thread and shows up
as time inside `run_backward`): where the Python side of a step goes, by function (tottime)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import ops, timer  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402
from planerecnet_amd.targets import DeviceTargetBuilder  # noqa: E402

timer.disable_all()
torch.set_num_threads(4)
dev = torch.device("cuda:0")
set_cfg("PlaneRecNet_101_config")
torch.manual_seed(0)
net = PlaneRecNet(cfg)
net.init_head_weights()
net = net.to(dev).train()
crit = PlaneRecNetLoss().to(dev)
opt = __import__("planerecnet_amd.optim", fromlist=["FusedAdam"]).FusedAdam(net.parameters(), lr=1e-4)
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
pf = DeviceTargetBuilder(crit)
pf.submit(inst, (480, 640))
pf.submit(inst, (480, 640))
ops.set_wgrad_async(True)


def step():
    opt.zero_grad(set_to_none=True)
    t = pf.get(depths, dev, overlap=True)
    pf.submit(inst, (480, 640))
    out = net(images)
    losses = crit(net, *out, inst, depths, targets=t)
    sum(losses.values()).sum().backward()
    ops.wgrad_join()
    opt.step()


for _ in range(6):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
N = 10
for _ in range(N):
    torch.cuda.synchronize()
    torch.cuda._sleep(int(0.25 * 2.4e9))
    pr.enable()
    step()
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(40)
print("(all times are totals over %d steps)" % N)
pf.close()
