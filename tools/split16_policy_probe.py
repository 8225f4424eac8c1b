"""Kernel duration of one split16 shape under rocprofv3 (PRN_SPLIT_STORE_POLICY experiment).  S16_SHAPE=0..4"""
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
bn = torch.nn.functional.relu
for _ in range(60):
    y = ops.conv2d(x, w, addend=addend)
    z = bn(y)                      # a consumer kernel after each launch, as in the network
torch.cuda.synchronize()
