#!/usr/bin/env python
"""Micro-benchmark of the DCNv2 sampling kernels on the five call shapes of PlaneRecNet_101 @480x640, B=8."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import ops  # noqa: E402
from tools.conv_bench import timeit  # noqa: E402

B = 8
SHAPES = [("128ch 120x160 s2", 128, 120, 160, 2), ("128ch 60x80 s1", 128, 60, 80, 1), ("256ch 60x80 s2", 256, 60, 80, 2),
          ("256ch 30x40 s1", 256, 30, 40, 1), ("512ch 30x40 s2", 512, 30, 40, 2)]
dev = torch.device("cuda:0")
print("%-20s %10s %10s | %10s %10s" % ("shape", "fwd us", "GB/s", "bwd us", "GB/s"))
for name, C, H, W, s in SHAPES:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(B, C, H, W, device=dev)
    om = torch.randn(B, 27, Ho, Wo, device=dev) * 0.7
    w = torch.randn(C, C, 3, 3, device=dev) * 0.02
    xr = x.clone().requires_grad_(True)
    omr = om.clone().requires_grad_(True)
    cols = torch.empty(B, C * 9, Ho, Wo, device=dev)
    from planerecnet_amd.ops import lib, _p, _stream, check
    tf = timeit(lambda: check(lib.prn_dcn_sample(_p(x), _p(om), _p(cols), B, C, H, W, Ho, Wo, s, max(H, W) / 4.0, _stream()), "f"))
    dcols = torch.randn_like(cols)
    dx, dom = torch.empty_like(x), torch.empty_like(om)
    ws = torch.empty(lib.prn_dcn_sample_bwd_ws_bytes(B, C, H, W, Ho, Wo) // 4, device=dev)
    tb = timeit(lambda: check(lib.prn_dcn_sample_bwd(_p(x), _p(om), _p(dcols), _p(dx), _p(dom), _p(ws), B, C, H, W, Ho, Wo, s, max(H, W) / 4.0, _stream()), "b"))
    bf = 4.0 * (x.numel() + om.numel() + cols.numel())
    bb = 4.0 * (2 * x.numel() + 2 * om.numel() + cols.numel())
    print("%-20s %10.1f %10.1f | %10.1f %10.1f" % (name, tf * 1e6, bf / tf / 1e9, tb * 1e6, bb / tb / 1e9))
