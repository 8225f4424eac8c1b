#!/usr/bin/env python
"""Per-shape micro-benchmark of the implicit-GEMM conv kernels on the shapes PlaneRecNet_101 @480x640, B=8 launches.
Times each launch with HIP events on the launch stream (10 reps) and prints TFLOP/s (algorithmic FLOPs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import ops  # noqa: E402

B = 8
# name, C, H, W, M, K, stride, pad, mode
SHAPES = [
    ("stem 7x7 s2", 3, 480, 640, 64, 7, 2, 3, 0),
    ("l0 1x1 64->64", 64, 120, 160, 64, 1, 1, 0, 0),
    ("l0 3x3 64", 64, 120, 160, 64, 3, 1, 1, 0),
    ("l0 1x1 64->256", 64, 120, 160, 256, 1, 1, 0, 0),
    ("l0 1x1 256->64", 256, 120, 160, 64, 1, 1, 0, 0),
    ("l1 1x1 512->128", 512, 60, 80, 128, 1, 1, 0, 0),
    ("l1 3x3 128", 128, 60, 80, 128, 3, 1, 1, 0),
    ("l1 1x1 128->512", 128, 60, 80, 512, 1, 1, 0, 0),
    ("l2 1x1 1024->256", 1024, 30, 40, 256, 1, 1, 0, 0),
    ("l2 3x3 256", 256, 30, 40, 256, 3, 1, 1, 0),
    ("l2 1x1 256->1024", 256, 30, 40, 1024, 1, 1, 0, 0),
    ("l2 dcn gemm 2304->256", 2304, 30, 40, 256, 1, 1, 0, 0),
    ("l2 offset conv 256->27", 256, 30, 40, 27, 3, 1, 1, 0),
    ("l3 1x1 2048->512", 2048, 15, 20, 512, 1, 1, 0, 0),
    ("l3 3x3 512", 512, 15, 20, 512, 3, 1, 1, 0),
    ("l3 1x1 512->2048", 512, 15, 20, 2048, 1, 1, 0, 0),
    ("fpn 3x3 256 @120x160", 256, 120, 160, 256, 3, 1, 1, 0),
    ("fpn 3x3 256 @60x80", 256, 60, 80, 256, 3, 1, 1, 0),
    ("head 3x3 256 @40x40", 256, 40, 40, 256, 3, 1, 1, 0),
    ("head 3x3 256 @16x16", 256, 16, 16, 256, 3, 1, 1, 0),
    ("mask 3x3 256->128 @120x160", 256, 120, 160, 128, 3, 1, 1, 0),
    ("dec deconv4 up2 256->64 @120x160", 256, 120, 160, 64, 3, 1, 1, 2),
    ("dec conv4 refl 256->128 @120x160", 256, 120, 160, 128, 3, 1, 1, 1),
    ("dec deconv3 up2 256->128 @60x80", 256, 60, 80, 128, 3, 1, 1, 2),
    ("dec depth_pred 64->1 @240x320", 64, 240, 320, 1, 3, 1, 1, 1),
    ("prior 1x1 3728->256 @30x40", 3728, 30, 40, 256, 1, 1, 0, 0),
    ("gemm-like 1x1 4096->4096 @64x64", 4096, 64, 64, 4096, 1, 1, 0, 0),
    ("gemm-like 3x3 512->512 @64x64", 512, 64, 64, 512, 3, 1, 1, 0),
]


_FLUSH = None


def timeit_cold(fn, reps=6):
    """COLD=1: every timed launch runs after a 1 GiB fill that evicts the operands from L2 / Infinity Cache (inside a
    training step a layer's activations and weights are not cache-resident the way they are in a back-to-back loop)."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(256 << 20, device="cuda", dtype=torch.float32)
    fn()
    torch.cuda.synchronize()
    tot = 0.0
    for i in range(reps):
        _FLUSH.fill_(float(i))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / reps * 1e-3


def timeit(fn, reps=10):
    if os.environ.get("COLD"):
        return timeit_cold(fn)
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    dev = torch.device("cuda:0")
    only = sys.argv[1] if len(sys.argv) > 1 else None
    print("%-36s %8s | %9s %7s | %9s %7s | %9s %7s" % ("shape", "GFLOP", "fwd us", "TF/s", "dgrad us", "TF/s", "wgrad us", "TF/s"))
    tot = [0.0, 0.0, 0.0, 0.0]
    for name, C, H, W, M, K, stride, pad, mode in SHAPES:
        if only and only not in name:
            continue
        x = torch.randn(B, C, H, W, device=dev)
        w = torch.randn(M, C, K, K, device=dev) * (C * K * K) ** -0.5
        if os.environ.get("ZERO"):            # DVFS probe: all-zero operands draw less power -> higher clock (MI355X_MICROARCH.md)
            x.zero_(); w.zero_()
        Ho, Wo = ops._out_hw(H, W, K, stride, pad, mode)
        dy = torch.randn(B, M, Ho, Wo, device=dev)
        fl = 2.0 * M * C * K * K * B * Ho * Wo
        tf = timeit(lambda: ops.conv_fwd_raw(x, w, None, None, M, K, stride, pad, Ho, Wo, mode))
        td = timeit(lambda: ops.conv_dgrad_raw(dy, w, x.shape, stride, pad, mode)) if C > 3 else float("nan")
        tw = timeit(lambda: ops.conv_wgrad_raw(x, dy, M, K, stride, pad, mode))
        print("%-36s %8.2f | %9.1f %7.1f | %9.1f %7.1f | %9.1f %7.1f" % (name, fl / 1e9, tf * 1e6, fl / tf / 1e12, td * 1e6, fl / td / 1e12, tw * 1e6, fl / tw / 1e12))


if __name__ == "__main__":
    main()
