#!/usr/bin/env python
"""Train-mode BatchNorm (+ReLU / +residual) forward and backward per shape: PRN_BN_SMALL_MAX=12288 python tools/bn_bench.py for the
two-launch path on the 38400-value maps."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import ops  # noqa: E402
from tools.conv_bench import timeit  # noqa: E402

B = 8
for (C, H, W, res) in [(128, 60, 80, False), (512, 60, 80, True), (512, 60, 80, False), (256, 30, 40, False), (64, 120, 160, False), (256, 120, 160, True)]:
    x = torch.randn(B, C, H, W, device="cuda")
    r = torch.randn(B, C, H, W, device="cuda") if res else None
    g, b = torch.ones(C, device="cuda").requires_grad_(True), torch.zeros(C, device="cuda").requires_grad_(True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    xg = x.clone().requires_grad_(True)
    y = ops.batch_norm(xg, g, b, rm, rv, True, 1e-5, 0.1, r, True)
    dy = torch.randn_like(y)
    tf = timeit(lambda: ops.batch_norm(x, g, b, rm, rv, True, 1e-5, 0.1, r, True), reps=20)
    tb = timeit(lambda: torch.autograd.grad(y, [xg, g, b], dy, retain_graph=True), reps=20)
    n = x.numel() * 4
    print("C=%4d %3dx%3d res=%d   fwd %6.1f us (%4.2f TB/s of 2 passes)   bwd %6.1f us (%4.2f TB/s of 3 passes)" % (C, H, W, res, tf * 1e6, (2 + res) * n / tf / 1e12, tb * 1e6, (3 + res) * n / tb / 1e12), flush=True)
