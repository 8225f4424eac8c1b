#!/usr/bin/env python
"""Run ONE conv shape repeatedly (for rocprofv3 --pmc runs): python tools/conv_one.py "<name substring>" [fwd|dgrad|wgrad] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import ops  # noqa: E402
from tools.conv_bench import B, SHAPES  # noqa: E402

name, which, reps = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "fwd"), int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
for nm, C, H, W, M, K, stride, pad, mode in SHAPES:
    if name not in nm:
        continue
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(M, C, K, K, device=dev) * (C * K * K) ** -0.5
    Ho, Wo = ops._out_hw(H, W, K, stride, pad, mode)
    dy = torch.randn(B, M, Ho, Wo, device=dev)
    for _ in range(reps):
        if which == "fwd":
            ops.conv_fwd_raw(x, w, None, None, M, K, stride, pad, Ho, Wo, mode)
        elif which == "dgrad":
            ops.conv_dgrad_raw(dy, w, x.shape, stride, pad, mode)
        else:
            ops.conv_wgrad_raw(x, dy, M, K, stride, pad, mode)
    torch.cuda.synchronize()
