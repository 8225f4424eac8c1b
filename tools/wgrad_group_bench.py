#!/usr/bin/env python
"""Weight gradients of G same-shape 1x1 layers: G single launches (each with its own pixel splits + reduction) against one grouped
launch (include/prn.h: prn_conv2d_wgrad_grouped)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import ops  # noqa: E402
from tools.conv_bench import timeit  # noqa: E402

B = 8
for (C, H, W, M) in [(1024, 30, 40, 256), (256, 30, 40, 1024), (2048, 15, 20, 512), (512, 60, 80, 128), (256, 120, 160, 64)]:
    for G in (1, 2, 4, 8, 16):
        xs = [torch.randn(B, C, H, W, device="cuda") for _ in range(G)]
        dys = [torch.randn(B, M, H, W, device="cuda") for _ in range(G)]
        t1 = timeit(lambda: [ops.conv_wgrad_raw(x, dy, M, 1, 1, 0, 0) for x, dy in zip(xs, dys)])
        t2 = timeit(lambda: ops.conv_wgrad_grouped_raw(xs, dys, M, 1, 1, 0, 0))
        fl = 2.0 * M * C * B * H * W * G
        print("%4d->%4d @%3dx%3d  G=%2d   singles %7.1f us (%5.1f TF/s)   grouped %7.1f us (%5.1f TF/s)" % (C, M, H, W, G, t1 * 1e6, fl / t1 / 1e12, t2 * 1e6, fl / t2 / 1e12), flush=True)
