#!/usr/bin/env python
"""GPU time of the loss alone (forward, and backward down to the network outputs) inside a training step, per term."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import ops, timer  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss, TargetPrefetcher  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402

timer.disable_all()
torch.set_num_threads(4)
dev = torch.device("cuda:0")
set_cfg("PlaneRecNet_101_config")
torch.manual_seed(0)
net = PlaneRecNet(cfg)
net.init_head_weights()
net = net.to(dev).train()
crit = PlaneRecNetLoss().to(dev)
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
pf = TargetPrefetcher(crit)
pf.submit(inst, (480, 640))
t = pf.get(depths, dev)
with torch.no_grad():
    out = net(images)
leaves = [out[0].detach().requires_grad_(True)] + [c.detach().requires_grad_(True) for c in out[1]] + [k.detach().requires_grad_(True) for k in out[2]] + \
         [out[3].detach().requires_grad_(True)]


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def run(which=None):
    losses = crit(net, leaves[0], leaves[1:5], leaves[5:9], leaves[9], inst, depths, targets=t)
    e1 = ev()
    tot = sum(v.sum() for k, v in losses.items() if which is None or k == which)
    torch.autograd.grad(tot, leaves, allow_unused=True)
    e2 = ev()
    return e1, e2


for _ in range(3):
    run()
torch.cuda.synchronize()
n = 10
f = b = 0.0
for _ in range(n):
    torch.cuda._sleep(int(0.02 * 2.4e9))
    e0 = ev()
    e1, e2 = run()
    torch.cuda.synchronize()
    f += e0.elapsed_time(e1)
    b += e1.elapsed_time(e2)
print("loss forward %.2f ms, backward to the network outputs %.2f ms (GPU time, stream parked while enqueueing)" % (f / n, b / n))
for k in ("ins", "cat", "dpt", "pln", "lav"):
    bb = 0.0
    for _ in range(5):
        torch.cuda._sleep(int(0.02 * 2.4e9))
        e1, e2 = run(k)
        torch.cuda.synchronize()
        bb += e1.elapsed_time(e2)
    print("  backward of %s alone: %.2f ms" % (k, bb / 5))
pf.close()
