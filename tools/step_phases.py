#!/usr/bin/env python
"""Host-side phase times of the bench step (no device syncs inside the step): where the main thread spends its time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss, TargetPrefetcher  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402
from planerecnet_amd import ops, timer  # noqa: E402

timer.disable_all()
ops.set_wgrad_async(True)

torch.set_num_threads(int(os.environ.get("HOST_THREADS", "4")))
dev = torch.device("cuda:0")
H, W = 480, 640
PB = int(os.environ.get("PHASE_B", "8"))   # PHASE_B=1 exposes the pure host cost (the GPU then finishes long before the host)
set_cfg("PlaneRecNet_101_config")
torch.manual_seed(0)
net = PlaneRecNet(cfg)
net.init_head_weights()
net = net.to(dev).train()
crit = PlaneRecNetLoss().to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
images, inst, depths = bench.synth_batch(PB, H, W, 1000, dev)
pf = TargetPrefetcher(crit)
pf.submit(inst, (H, W))
pf.submit(inst, (H, W))
acc = {}


def mark(name, t0):
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3


def step():
    t0 = time.perf_counter(); opt.zero_grad(set_to_none=True); ft, fv = pf.queue.popleft(); h = ft.result(); h["vnl"] = fv.result(); mark("wait_worker", t0)
    t0 = time.perf_counter()
    mark("pin", t0)
    t0 = time.perf_counter(); targets = crit.upload(h, depths, dev); pf.submit(inst, (H, W)); mark("upload", t0)
    t0 = time.perf_counter(); out = net(images); mark("net_fwd", t0)
    t0 = time.perf_counter(); losses = crit(net, *out, inst, depths, targets=targets); loss = sum(losses.values()).sum(); mark("loss_fwd", t0)
    t0 = time.perf_counter(); loss.backward(); ops.wgrad_join(); mark("backward", t0)
    t0 = time.perf_counter(); opt.step(); mark("adam", t0)


for _ in range(3):
    step()
torch.cuda.synchronize()
acc.clear()
N = 8
t0 = time.perf_counter()
for _ in range(N):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print({k: round(v / N, 1) for k, v in acc.items()}, "| host enqueue/step %.1f ms, wall/step %.1f ms" % (t_enq / N * 1e3, t_all / N * 1e3))
pf.close()
