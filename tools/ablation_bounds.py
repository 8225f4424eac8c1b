#!/usr/bin/env python
"""Upper bounds for the fusions that were costed and not built: the training step of the headline workload timed with the
helper launches a fusion would REMOVE simply left out (include/prn.h: prn_debug_skip_launches; results are wrong while a bit is set -- the consumers
read stale sums -- the launches and bytes of everything else are unchanged).  What a leg gains is what the fused form could gain if its epilogue work
were free; the real gain is smaller.  Legs are interleaved with plain legs so that clock drift shows.   python tools/ablation_bounds.py [steps=20]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import ops, timer  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss  # noqa: E402
from planerecnet_amd.optim import FusedAdam  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402
from planerecnet_amd.targets import DeviceTargetBuilder  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
timer.disable_all()
torch.set_num_threads(4)                                # (host-side tensor ops are small: a wide OpenMP team only adds fork / join latency and noise)
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


def restore():
    """Weights, BatchNorm buffers and optimiser state as they were before an ablated leg (which may have driven them to NaN)."""
    global opt
    with torch.no_grad():
        for k, v in net.state_dict().items():
            v.copy_(SNAP[k])
    opt = FusedAdam(net.parameters(), lr=1e-5)


DIRTY = [False]


def leg(mask):
    if DIRTY[0]:
        restore()
    DIRTY[0] = 0 < mask < 256
    torch.cuda.synchronize()
    ops.lib.prn_debug_skip_launches(mask)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(STEPS):
        step()
    e1.record()
    torch.cuda.synchronize()
    ops.lib.prn_debug_skip_launches(0)
    return e0.elapsed_time(e1) / STEPS


for _ in range(40):
    step()
SNAP = {k: v.detach().clone() for k, v in net.state_dict().items()}
NAMES = {1: "forward BatchNorm statistics pass (large maps)", 2: "backward BatchNorm sums pass (large maps)", 4: "sums of weight-gradient split partials",
         8: "prn_channel_sum (bias gradients, sums for lazily-summed BatchNorm inputs)", 16: "x0.5 resize / adjoint (FPN levels)", 31: "all five families"}
plain, rows = [], []
for mask in (1 << 8, 2 << 8, 4 << 8, 8 << 8, 16 << 8, 31 << 8, 4):
    plain.append(leg(0))
    rows.append((mask, leg(mask)))
plain.append(leg(0))
print("plain legs (ms/step, %d steps each, interleaved): %s" % (STEPS, " ".join("%.2f" % p for p in plain)))
for i, (mask, ms) in enumerate(rows):
    ref = 0.5 * (plain[i] + plain[i + 1])
    if mask >= 256:
        print("TWICE    %-78s %.2f ms/step   cost in the step %.2f ms (%.1f %%)" % (NAMES[mask >> 8], ms, ms - ref, 100 * (ms - ref) / ref))
    else:
        print("LEFT OUT %-78s %.2f ms/step   gain <= %.2f ms (%.1f %%)" % (NAMES[mask], ms, ref - ms, 100 * (ref - ms) / ref))
pf.close()
