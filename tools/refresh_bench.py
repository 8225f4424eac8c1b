"""GPU time of the per-step weight preparation (dgrad layouts, Winograd transform-domain operands, split-GEMM images) of PlaneRecNet_101:
python tools/refresh_bench.py   (PRN_LIB=<other build> for an A/B)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from planerecnet_amd import ops  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402

set_cfg("PlaneRecNet_101_config")
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = PlaneRecNet(cfg)
net.init_head_weights()
net = net.to(dev).train()
crit = PlaneRecNetLoss().to(dev)
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
for _ in range(4):                      # the caches learn which operands the step asks for
    net.zero_grad(set_to_none=True)
    out = net(images)
    sum(crit(net, *out, inst, depths).values()).sum().backward()
torch.cuda.synchronize()
for p in net.parameters():              # as after an optimizer step: every derived operand is stale
    p.data.mul_(1.0)
    p._version  # noqa: B018
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(10):
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.0)                 # bumps the version counters
    torch.cuda.synchronize()
    s.record()
    net._refresh_dgrad_weights()
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
ts.sort()
print("per-step weight preparation (flip + Winograd operands + split images): median %.3f ms, min %.3f ms of GPU time" % (ts[len(ts) // 2], ts[0]))
