#!/usr/bin/env python
"""Where do the zero-fills and buffer copies of a training step come from?  torch.profiler (CPU activity, Python stacks) over two steps of the
bench's default loop (device target builder); aten::fill_ / zero_ / copy_ events grouped by the innermost frame of this package (or the autograd
node) that issued them.  Complements tools/aten_trace.py, which does not see fills issued inside composite operators.   python tools/fill_probe.py"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import ops, timer  # noqa: E402
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


def step():
    opt.zero_grad(set_to_none=True)
    t = pf.get(depths, dev, overlap=True)
    pf.submit(inst, (480, 640))
    out = net(images)
    losses = crit(net, *out, inst, depths, targets=t)
    sum(losses.values()).sum().backward()
    ops.wgrad_join()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0])
for ev in prof.events():
    if ev.name not in ("aten::fill_", "aten::zero_", "aten::copy_"):
        continue
    shapes = ev.input_shapes[0] if ev.input_shapes else []
    n = 1
    for s in shapes:
        n *= s
    chain, q = [], ev.cpu_parent                           # the operators this one was issued from, outermost last (Python stacks are not recorded on this build)
    while q is not None:
        chain.append(q.name)
        q = q.cpu_parent
    where = " <- ".join(chain[:4]) if chain else "(top level: a direct call)"
    e = acc[(ev.name, where)]
    e[0] += 1
    e[1] += n
print("%-14s %-110s %8s %14s" % ("op", "issued from", "calls/step", "elements/step"))
for (name, where), (c, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-14s %-110s %8.1f %14d" % (name, where[:110], c / STEPS, n // STEPS))
pf.close()
