#!/usr/bin/env python
"""Is a training step bit-stable run to run?  The same weights, batch and loss targets through forward + loss + backward N times (PlaneRecNet_101, B = 2 by default, the
benchmark's plan with BATCH=8): every loss term and every parameter gradient compared bit for bit with the first repetition; prints what differs and by how much.
    python tools/determinism_probe.py [repetitions=4]      (BATCH, CONFIG in the environment)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import ops, timer  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402
from planerecnet_amd.targets import DeviceTargetBuilder  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(os.environ.get("BATCH", "2"))
timer.disable_all()
torch.set_num_threads(4)
dev = torch.device("cuda:0")
set_cfg(os.environ.get("CONFIG", "PlaneRecNet_101_config"))
torch.manual_seed(0)
net = PlaneRecNet(cfg)
net.init_head_weights()
with torch.no_grad():                                      # offsets of the deformable layers off their zero initialisation: every bilinear corner in play
    for n, p in net.named_parameters():
        if "offset_conv" in n or "modulator_conv" in n:
            p.normal_(0.0, 0.02)
net = net.to(dev).train()
crit = PlaneRecNetLoss().to(dev)
images, inst, depths = bench.synth_batch(B, 480, 640, 1000, dev)
pf = DeviceTargetBuilder(crit)
pf.submit(inst, (480, 640))
targets = pf.get(depths, dev)
ops.set_wgrad_async(True)
params = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
state = {k: v.detach().clone() for k, v in net.state_dict().items()}
first = None
for rep in range(REPS):
    with torch.no_grad():
        for k, v in net.state_dict().items():              # (BatchNorm running statistics back to the same start)
            v.copy_(state[k])
    for _, p in params:
        p.grad = None
    out = net(images)
    losses = crit(net, *out, inst, depths, targets=targets)
    sum(losses.values()).sum().backward()
    ops.wgrad_join()
    torch.cuda.synchronize()
    snap = {"loss " + k: v.detach().clone() for k, v in losses.items()}
    snap.update({n: p.grad.detach().clone() for n, p in params if p.grad is not None})
    if first is None:
        first = snap
        print("%d loss terms, %d parameter gradients; losses %s" % (len(losses), len(snap) - len(losses), {k: float(v) for k, v in losses.items()}))
        continue
    diff = []
    for k, v in snap.items():
        if not torch.equal(v, first[k]):
            d = (v.double() - first[k].double())
            diff.append((float(d.norm() / (first[k].double().norm() + 1e-300)), k, int((v != first[k]).sum()), v.numel()))
    print("repetition %d: %d of %d tensors differ from the first" % (rep, len(diff), len(snap)))
    for r, k, n, tot in sorted(diff, reverse=True)[:12]:
        print("    %-64s rel-L2 %.2e   %d of %d elements" % (k, r, n, tot))
pf.close()
