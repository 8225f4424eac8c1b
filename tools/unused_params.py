#!/usr/bin/env python
"""Which parameters receive no gradient in a training step (they would hold back the in-order gradient buckets)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import ops, timer  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402

timer.disable_all()
dev = torch.device("cuda:0")
set_cfg("PlaneRecNet_101_config")
net = PlaneRecNet(cfg)
net.init_head_weights()
net = net.to(dev).train()
crit = PlaneRecNetLoss().to(dev)
images, inst, depths = bench.synth_batch(2, 480, 640, 1000, dev)
out = net(images)
losses = crit(net, *out, inst, depths)
sum(losses.values()).sum().backward()
ps = list(net.named_parameters())
none = [n for n, p in ps if p.requires_grad and p.grad is None]
print(len(ps), "parameters,", len(none), "without gradient:", none)
