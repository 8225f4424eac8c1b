#!/usr/bin/env python
"""cProfile of the AUTOGRAD thread over the backward pass of whole training steps (the profiler is switched on by a gradient hook on the loss, i.e. on
the thread that runs the backward nodes, and off by the post-accumulate hook of the stem convolution's weight, the last node of the pass)."""
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
pr = cProfile.Profile()
ON = [False]
net.backbone.conv1.weight.register_post_accumulate_grad_hook(lambda p: pr.disable() if ON[0] else None)


def step():
    opt.zero_grad(set_to_none=True)
    t = pf.get(depths, dev, overlap=True)
    pf.submit(inst, (480, 640))
    out = net(images)
    losses = crit(net, *out, inst, depths, targets=t)
    tot = sum(losses.values()).sum()
    if ON[0]:
        tot.register_hook(lambda g: pr.enable())
    tot.backward()
    ops.wgrad_join()
    opt.step()


for _ in range(6):
    step()
torch.cuda.synchronize()
N = 10
ON[0] = True
for _ in range(N):
    torch.cuda.synchronize()
    torch.cuda._sleep(int(0.25 * 2.4e9))
    step()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(50)
st.sort_stats("cumtime").print_stats(45)
print("(all times are totals over %d backward passes)" % N)
pf.close()
