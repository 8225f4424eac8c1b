#!/usr/bin/env python
"""Probe: fixed cost vs per-K-step cost of the implicit-GEMM forward kernel (time = a + b*K at a fixed grid).
    PRN_CONV_FORCE=2,2,16,1 python tools/quant_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import ops  # noqa: E402
from tools.conv_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")


def run(C, M, H, W, K=1, B=8):
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(M, C, K, K, device=dev) * 0.02
    t = timeit(lambda: ops.conv_fwd_raw(x, w, None, None, M, K, 1, K // 2, H, W, 0), reps=20)
    fl = 2.0 * M * C * K * K * B * H * W
    print("C=%4d M=%4d %3dx%3d K=%d N=%6d  %7.1f us  %6.1f TF/s" % (C, M, H, W, K, B * H * W, t * 1e6, fl / t / 1e12), flush=True)


for (H, W) in [(32, 64), (30, 40)]:
    for C in [16, 64, 256, 512, 1024, 2048, 4096]:
        run(C, 256, H, W)
