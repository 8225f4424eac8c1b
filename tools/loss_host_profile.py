#!/usr/bin/env python
"""cProfile of the loss forward on the main thread (host enqueue cost per step)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import timer  # noqa: E402
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
with torch.no_grad():
    out = net(images)
out = [o.detach().requires_grad_(True) if torch.is_tensor(o) else [t.detach().requires_grad_(True) for t in o] for o in out]
pr = cProfile.Profile()
for it in range(6):
    t = pf.get(depths, dev)
    pf.submit(inst, (480, 640))
    if it >= 2:
        pr.enable()
    losses = crit(net, *out, inst, depths, targets=t)
    if it >= 2:
        pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumtime").print_stats(30)
pf.close()
