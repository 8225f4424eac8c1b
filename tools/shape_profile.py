#!/usr/bin/env python
"""Per-shape time of the conv launches inside a real training step (events around every conv_fwd_raw / conv_wgrad_raw
call, keyed by the launch geometry): which layers the MFMA time goes to, and at what rate each runs."""
import collections
import os
import sys

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
opt = __import__("planerecnet_amd.optim", fromlist=["FusedAdam"]).FusedAdam(net.parameters(), lr=1e-4)
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
pf = TargetPrefetcher(crit)
pf.submit(inst, (480, 640))
REC = []
_f, _w = ops.conv_fwd_raw, ops.conv_wgrad_raw
recording = False


def fwd(x, w2d, bias, addend, M, K, stride, pad, Ho, Wo, mode=0, dil=1, epi=0, scatter2=None):
    if not recording:
        return _f(x, w2d, bias, addend, M, K, stride, pad, Ho, Wo, mode, dil, epi, scatter2)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    y = _f(x, w2d, bias, addend, M, K, stride, pad, Ho, Wo, mode, dil, epi, scatter2)
    e.record()
    B, C, H, W = x.shape
    REC.append((("fwd", C, H, W, M, K, stride, mode, dil), 2.0 * M * C * K * K * B * Ho * Wo / (dil * dil), s, e))
    return y


def wg(x, dy, M, K, stride, pad, mode):
    if not recording:
        return _w(x, dy, M, K, stride, pad, mode)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    y = _w(x, dy, M, K, stride, pad, mode)
    e.record()
    B, C, H, W = x.shape
    REC.append((("wgrad", C, H, W, M, K, stride, mode, 1), 2.0 * M * C * K * K * B * dy.shape[2] * dy.shape[3], s, e))
    return y


ops.conv_fwd_raw, ops.conv_wgrad_raw = fwd, wg


def step():
    opt.zero_grad(set_to_none=True)
    t = pf.get(depths, dev)
    pf.submit(inst, (480, 640))
    out = net(images)
    losses = crit(net, *out, inst, depths, targets=t)
    sum(losses.values()).sum().backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
recording = True
N = 3
for _ in range(N):
    step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, fl, s, e in REC:
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += s.elapsed_time(e)
    a[2] += fl
tot = sum(a[1] for a in agg.values()) / N
print("total conv time / step: %.2f ms" % tot)
print("%-6s %5s %4s %4s %5s %2s %2s %4s %3s | %6s %9s %8s %7s" % ("kind", "C", "H", "W", "M", "K", "s", "mode", "dil", "calls", "ms/step", "us/call", "TF/s"))
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-6s %5d %4d %4d %5d %2d %2d %4d %3d | %6.1f %9.3f %8.1f %7.1f" % (key + (n / N, ms / N, ms / n * 1e3, fl / (ms * 1e-3) / 1e12)))
pf.close()
