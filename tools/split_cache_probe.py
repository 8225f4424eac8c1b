#!/usr/bin/env python
"""Does the operand-image cache of the 16-bit-pipe launches churn in steady state?  Counts, per training step of the bench's loop: per-launch cuts, cache hits, uncached (temporary)
operands, rebuilds of the batched refresh's item table (each one a host loop over ~300 entries and a host-to-device copy), operand-epoch bumps (each one makes every block re-resolve its
parameter table).   python tools/split_cache_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import blocks, ops, timer  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss  # noqa: E402
from planerecnet_amd.optim import FusedAdam  # noqa: E402
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
opt = FusedAdam(net.parameters(), lr=1e-5)
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
pf = DeviceTargetBuilder(crit)
pf.submit(inst, (480, 640))
pf.submit(inst, (480, 640))
ops.set_wgrad_async(True)
builds = [0]
_orig = ops._prep_items


def _counted(entries, d):
    builds[0] += 1
    return _orig(entries, d)


ops._prep_items = _counted


def step():
    opt.zero_grad(set_to_none=True)
    t = pf.get(depths, dev, overlap=True)
    pf.submit(inst, (480, 640))
    out = net(images)
    losses = crit(net, *out, inst, depths, targets=t)
    sum(losses.values()).sum().backward()
    ops.wgrad_join()
    opt.step()


for i in range(14):
    s0 = dict(ops.SPLIT_STATS)
    b0, e0, r0, v0 = builds[0], ops.OPERAND_EPOCH[0], blocks.STATS["resolved"], blocks.STATS["vouched"]
    step()
    s1 = ops.SPLIT_STATS
    print("step %2d: cuts %3d hits %4d uncached %3d refreshes %d | item-table rebuilds %d | operand-epoch bumps %3d | block tables re-resolved %2d vouched %2d | cached entries %d" %
          (i, s1["cuts"] - s0["cuts"], s1["hits"] - s0["hits"], s1["uncached"] - s0["uncached"], s1["refreshes"] - s0.get("refreshes", 0), builds[0] - b0,
           ops.OPERAND_EPOCH[0] - e0, blocks.STATS["resolved"] - r0, blocks.STATS["vouched"] - v0, len(ops._SPLIT_IMG)), flush=True)
pf.close()
