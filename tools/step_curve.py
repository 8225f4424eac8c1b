#!/usr/bin/env python
"""Wall time of every training step from a cold start (device-synchronised per step): how long until the step time settles?"""
import os
import sys
import time

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
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
pf = TargetPrefetcher(crit)
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


if os.environ.get("NOGC"):
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
evs, host = [], []
e0 = torch.cuda.Event(enable_timing=True)
e0.record()
allocs = []
for i in range(n):                                   # no synchronisation inside the loop: one event per step on the GPU timeline
    t0 = time.perf_counter()
    step()
    ms_ = torch.cuda.memory_stats()
    allocs.append((ms_.get("num_device_alloc", 0), ms_.get("num_device_free", 0)))
    host.append((time.perf_counter() - t0) * 1e3)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
torch.cuda.synchronize()
prev, ts = e0, []
for e in evs:
    ts.append(prev.elapsed_time(e))
    prev = e
print("gpu  ms/step:", " ".join("%.1f" % t for t in ts))
print("host ms/step:", " ".join("%.1f" % t for t in host))
print("device mallocs/frees after steps 5, 10, 20, last:", [allocs[i] for i in (5, 10, 20, len(allocs) - 1) if i < len(allocs)])
print("reserved MB", torch.cuda.memory_reserved() >> 20, "alloc retries", torch.cuda.memory_stats().get("num_alloc_retries"), "segments", torch.cuda.memory_stats().get("segment.all.current"))
pf.close()
