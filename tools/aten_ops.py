#!/usr/bin/env python
"""Which ATen operators (i.e. kernels outside libprn_hip.so) does a training step launch, how often, and from which line of
this package?  (torch.profiler; forward-side ops are attributed to the innermost planerecnet_amd / bench frame, backward-side
ops to the autograd node that issued them.)"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

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
opt = __import__("planerecnet_amd.optim", fromlist=["FusedAdam"]).FusedAdam(net.parameters(), lr=1e-4)
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
pf = TargetPrefetcher(crit)
pf.submit(inst, (480, 640))
ops.set_wgrad_async(True)


from torch.profiler import record_function  # noqa: E402


def step():
    opt.zero_grad(set_to_none=True)
    with record_function("PH_targets"):
        t = pf.get(depths, dev)
        pf.submit(inst, (480, 640))
    with record_function("PH_net_forward"):
        out = net(images)
    with record_function("PH_loss_forward"):
        losses = crit(net, *out, inst, depths, targets=t)
        tot = sum(losses.values()).sum()
    with record_function("PH_backward"):
        tot.backward()
        ops.wgrad_join()
    with record_function("PH_adam"):
        opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
evs = prof.events()
kernels = [e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA]
print("device kernels in the step: %d, %.2f ms" % (len(kernels), sum(e.device_time for e in kernels) / 1e3))
ours = [e for e in kernels if "anonymous namespace" in e.name and "at::" not in e.name]
print("  libprn_hip kernels: %d, %.2f ms" % (len(ours), sum(e.device_time for e in ours) / 1e3))
print("  other (ATen / rocprim / copies): %d, %.2f ms" % (len(kernels) - len(ours), sum(e.device_time for e in kernels if e not in set(ours)) / 1e3))
agg = collections.defaultdict(lambda: [0, 0.0])
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    dt = sum(k.duration for k in e.kernels) if e.kernels else 0.0
    if not e.kernels:
        continue
    where = "(autograd / other)"
    for fr in e.stack or []:
        if ("planerecnet_amd/" in fr or "bench.py" in fr or "tools/" in fr) and "profiler" not in fr:
            where = fr.strip().split("/")[-1][:70]
            break
    a = agg[(e.name, where)]
    a[0] += len(e.kernels)
    a[1] += dt
# the same by phase of the step (forward-side ops nest inside the phase's CPU range; backward-side ops run on the autograd
# thread during PH_backward's time span)
phases = [(e.name, e.time_range.start, e.time_range.end) for e in evs if e.name.startswith("PH_")]
byph = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::") or not e.kernels:
        continue
    ph = "(outside)"
    for name, a0, a1 in phases:
        if a0 <= e.time_range.start <= a1:
            ph = name
            break
    a = byph[ph][e.name]
    a[0] += len(e.kernels)
    a[1] += sum(k.duration for k in e.kernels)
for ph, d in byph.items():
    print("== %s: %d ATen launches, %.1f us" % (ph, sum(v[0] for v in d.values()), sum(v[1] for v in d.values())))
    for name, (n, us) in sorted(d.items(), key=lambda kv: -kv[1][0])[:14]:
        print("      %-28s %5d %9.1f us" % (name, n, us))
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("%-28s %-72s %6s %9s" % ("op", "issued from", "launch", "us/step"))
for (name, where), (n, us) in rows[:70]:
    print("%-28s %-72s %6d %9.1f" % (name, where, n, us))
pf.close()
