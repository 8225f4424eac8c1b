"""One split16 shape, 40 launches (for PMC passes: tools/pmc_split16_stalls.sh).  S16_SHAPE=0..4 as in tools/split16_phase_timing.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from planerecnet_amd import ops  # noqa: E402

B = 8
SHAPES = [(1024, 256, 30, 40, True), (256, 1024, 30, 40, False), (512, 128, 60, 80, True), (128, 512, 60, 80, False), (256, 256, 120, 160, False)]
M, C, H, W, add = SHAPES[int(os.environ.get("S16_SHAPE", "0"))]
ops.set_split_gemm(mode=2)
x = torch.relu(torch.randn(B, C, H, W, device="cuda"))
w = torch.randn(M, C, 1, 1, device="cuda") * 0.05
addend = torch.randn(B, M, H, W, device="cuda") if add else None
for _ in range(40):
    ops.conv2d(x, w, addend=addend)
torch.cuda.synchronize()
