#!/usr/bin/env python
"""Micro-benchmark of the fused DCNv2 operator (include/prn.h: prn_dcnv2_*) on the five call shapes of PlaneRecNet_101
@480x640, B=8, next to the column-tensor path it replaces (prn_dcn_sample + 1x1 GEMM) and a plain 3x3 conv of the same size.

    python tools/dcn_bench.py [--offsets 0.6]      (r.m.s. offset in pixels; 0 = the init state of the reference)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import ops  # noqa: E402
from planerecnet_amd.ops import lib, _p, _stream, check  # noqa: E402
from tools.conv_bench import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--offsets", type=float, default=0.6)
ap.add_argument("--batch", type=int, default=8)
args = ap.parse_args()
B = args.batch
SHAPES = [("128ch 120x160 s2", 128, 120, 160, 2), ("128ch 60x80 s1", 128, 60, 80, 1), ("256ch 60x80 s2", 256, 60, 80, 2),
          ("256ch 30x40 s1", 256, 30, 40, 1), ("512ch 30x40 s2", 512, 30, 40, 2)]
dev = torch.device("cuda:0")
print("offsets r.m.s. %.2f px, B = %d; TF/s = 2*M*9C*N / time; algorithmic bytes fwd = 4*(x + om + y + w)" % (args.offsets, B))
print("%-18s %6s | %8s %6s %7s | %8s %6s | %9s %9s | %8s | %9s %9s" % ("shape", "GF", "fused us", "TF/s", "alg GB/s", "wgrad us", "TF/s", "dcolGEMM", "dx+dom us",
                                                                         "table us", "old fwd", "conv3x3"))
for name, C, H, W, s in SHAPES:
    M = C
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    mo = max(H, W) / 4.0
    x = torch.randn(B, C, H, W, device=dev)
    om = torch.randn(B, 27, Ho, Wo, device=dev) * args.offsets
    w = torch.randn(M, C, 3, 3, device=dev) * 0.02
    dy = torch.randn(B, M, Ho, Wo, device=dev)
    gf = 2.0 * M * C * 9 * B * Ho * Wo
    table = ops.dcn_table(x.shape, M, om, None, s, 1, 1, mo)
    t_tab = timeit(lambda: ops.dcn_table(x.shape, M, om, None, s, 1, 1, mo))
    t_f = timeit(lambda: ops.dcn_fwd_raw(x, table, w, None, s, 1, 1, mo))
    t_w = timeit(lambda: ops.dcn_wgrad_raw(x, table, dy, M, s, 1, 1, mo))
    t_d = timeit(lambda: ops.dcn_data_grads_raw(x, om, None, w, dy, s, 1, 1, mo))
    t_g = timeit(lambda: ops.dcn_data_grads_raw(x, om, None, w, dy, s, 1, 1, mo, need_x=False, need_om=False))
    # the path this replaces: sampler -> column tensor -> 1x1 GEMM
    cols = torch.empty(B, C * 9, Ho, Wo, device=dev)

    def old():
        check(lib.prn_dcn_sample(_p(x), _p(om), _p(cols), B, C, H, W, Ho, Wo, s, mo, _stream()), "f")
        return ops.conv_fwd_raw(cols, w, None, None, M, 1, 1, 0, Ho, Wo)
    t_o = timeit(old)
    t_c = timeit(lambda: ops.conv_fwd_raw(x, w, None, None, M, 3, s, 1, Ho, Wo))
    alg = 4.0 * (x.numel() + om.numel() + B * M * Ho * Wo + w.numel())
    print("%-18s %6.2f | %8.1f %6.1f %7.1f | %8.1f %6.1f | %9.1f %9.1f | %8.1f | %9.1f %9.1f"
          % (name, gf / 1e9, t_f * 1e6, gf / t_f / 1e12, alg / t_f / 1e9, t_w * 1e6, gf / t_w / 1e12, t_g * 1e6, (t_d - t_g) * 1e6, t_tab * 1e6, t_o * 1e6,
             t_c * 1e6))
