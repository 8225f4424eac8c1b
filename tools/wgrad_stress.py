#!/usr/bin/env python
"""One weight-gradient launch sequence repeated thousands of times, bit-compared with its first result -- alone, and with unrelated work running on a second stream
(different wave scheduling, cache and memory-system timing): an intra-kernel hazard (a missing barrier, an LDS buffer re-used a slice early) shows up as a sporadic
difference; arithmetic never does.  The shapes are the stride-1 3x3 layers of PlaneRecNet_101's heads at B = 2 with every launch forced onto the 16-bit pipe
(`all-f16` of the parity tests) and under the default plan.      python tools/wgrad_stress.py [repetitions=3000]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import ops  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
SHAPES = [(2, 256, 30, 40, 128), (2, 256, 60, 80, 128), (2, 256, 30, 40, 256), (2, 128, 60, 80, 128), (2, 256, 15, 20, 128), (8, 256, 30, 40, 256)]
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev)
for arith in ({"mode": 2, "kind": "f16"}, {}):
    old = ops.set_split_gemm(**arith)
    try:
        for (B, C, H, W, M) in SHAPES:
            x = torch.randn(B, C, H, W, generator=g).to(dev)
            dy = torch.randn(B, M, H, W, generator=g).to(dev)
            w = (torch.randn(M, C, 3, 3, generator=g) * 0.05).to(dev)
            for use_v in (True, False):
                keep = []
                ops.conv3x3_winograd_raw(x, ops.winograd_weights(w)[0], None, None, M, ops.IN_ZERO, ops.EPI_NONE, keep)
                V = keep[0] if (use_v and keep) else None
                for busy in (False, True):
                    first, bad = None, torch.zeros((), device=dev, dtype=torch.int64)
                    for rep in range(REPS):
                        if busy and rep % 8 == 0:
                            with torch.cuda.stream(side):
                                torch.mm(a, a)
                        dw = ops.conv3x3_winograd_wgrad_raw(x, dy, M, ops.IN_ZERO, V)
                        if first is None:
                            first = dw.clone()
                        else:
                            bad += (dw != first).any()                  # (on the device: no synchronisation between repetitions)
                    torch.cuda.synchronize()
                    print("arith %-22s B %d C %3d %3dx%-3d M %3d  V kept %-5s  second stream busy %-5s: %d of %d results differ" %
                          (arith or "default", B, C, H, W, M, V is not None, busy, int(bad), REPS - 1), flush=True)
    finally:
        ops.set_split_gemm(**old)
