#!/usr/bin/env python
"""Direct implicit-GEMM kernel vs the Winograd F(4x4, 3x3) path on the 3x3 / stride-1 shapes of PlaneRecNet_101 @480x640,
B=8: whole-call time and the three stages (input transform, 36 batched GEMMs, output transform).  COLD=1 as in conv_bench.py;
PRN_WINO_TILE="tm,tn" forces the GEMM tile."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import ops  # noqa: E402
from planerecnet_amd.ops import lib, _p, _stream  # noqa: E402
from conv_bench import timeit  # noqa: E402

B = int(os.environ.get("BATCH", "8"))
SHAPES = [
    ("l0 3x3 64 @120x160", 64, 120, 160, 64),
    ("l1 3x3 128 @60x80", 128, 60, 80, 128),
    ("l2 3x3 256 @30x40", 256, 30, 40, 256),
    ("l3 3x3 512 @15x20", 512, 15, 20, 512),
    ("fpn 3x3 256 @120x160", 256, 120, 160, 256),
    ("fpn 3x3 256 @60x80", 256, 60, 80, 256),
    ("fpn 3x3 256 @30x40", 256, 30, 40, 256),
    ("mask 3x3 128->256 @120x160", 128, 120, 160, 256),
    ("mask 3x3 256->128 @120x160", 256, 120, 160, 128),
    ("dec 3x3 128->256 @60x80", 128, 60, 80, 256),
]


def main():
    dev = torch.device("cuda:0")
    print(f"{'shape':34s} {'GFLOP':>7s} | {'direct us':>9s} {'TF/s':>6s} | {'wino us':>8s} {'TF/s':>6s} {'x':>5s} | {'in us':>6s} {'gemm us':>8s} {'TF/s':>6s} {'out us':>7s}")
    for name, C, H, W, M in SHAPES:
        x = torch.randn(B, C, H, W, device=dev)
        w = torch.randn(M, C, 3, 3, device=dev) * (C * 9) ** -0.5
        U, _ = ops.winograd_weights(w)
        fl = 2.0 * M * C * 9 * B * H * W
        td = timeit(lambda: ops.conv_fwd_raw(x, w, None, None, M, 3, 1, 1, H, W))
        tw = timeit(lambda: ops.conv3x3_winograd_raw(x, U, None, None, M))
        P = lib.prn_winograd_tiles(B, H, W)
        ws = torch.empty(36 * (C + M) * P, device=dev)
        V, Yt = ws[:36 * C * P], ws[36 * C * P:]
        gws = torch.empty(max(lib.prn_gemm_batched_ws_bytes(M, C, P, 36, ops.opts_ref()), 16) // 4, device=dev)
        y = torch.empty(B, M, H, W, device=dev)
        ti = timeit(lambda: lib.prn_winograd_input(_p(x), _p(V), B, C, H, W, 0, _stream()))
        tg = timeit(lambda: lib.prn_gemm_batched(M, C, P, 36, _p(U), None, _p(V), _p(Yt), _p(gws), ops.opts_ref(), _stream()))
        to = timeit(lambda: lib.prn_winograd_output(_p(Yt), None, None, _p(y), B, M, H, W, 0, _stream()))
        gfl = 2.0 * 36 * M * C * P
        print(f"{name:34s} {fl / 1e9:7.2f} | {td * 1e6:9.1f} {fl / td / 1e12:6.1f} | {tw * 1e6:8.1f} {fl / tw / 1e12:6.1f} {td / tw:5.2f} | "
              f"{ti * 1e6:6.1f} {tg * 1e6:8.1f} {gfl / tg / 1e12:6.1f} {to * 1e6:7.1f}", flush=True)


if __name__ == "__main__":
    main()


def wgrad_main():
    dev = torch.device("cuda:0")
    print(f"{'wgrad shape':34s} {'GFLOP':>7s} | {'direct us':>9s} {'TF/s':>6s} | {'wino us':>8s} {'TF/s':>6s} {'x':>5s} | {'xform us':>8s} {'gemm us':>8s} {'TF/s':>6s} {'dw us':>7s} splits")
    for name, C, H, W, M in SHAPES:
        x = torch.randn(B, C, H, W, device=dev)
        dy = torch.randn(B, M, H, W, device=dev)
        fl = 2.0 * M * C * 9 * B * H * W
        td = timeit(lambda: ops.conv_wgrad_raw(x, dy, M, 3, 1, 1, 0))
        tw = timeit(lambda: ops.conv3x3_winograd_wgrad_raw(x, dy, M))
        P = lib.prn_winograd_tiles(B, H, W)
        ws = torch.empty(lib.prn_winograd_wgrad_ws_bytes(B, C, H, W, M, ops.opts_ref()) // 4, device=dev)
        dw = torch.empty(M, C, 3, 3, device=dev)
        args = (_p(x), _p(dy), _p(dw), _p(ws), B, C, H, W, M, 0, ops.opts_ref(), _stream())
        t1 = timeit(lambda: lib.prn_conv3x3_winograd_wgrad(*args, 1))
        t2 = timeit(lambda: lib.prn_conv3x3_winograd_wgrad(*args, 2))
        t3 = timeit(lambda: lib.prn_conv3x3_winograd_wgrad(*args, 3))
        gfl = 2.0 * 36 * M * C * P
        print(f"{name:34s} {fl / 1e9:7.2f} | {td * 1e6:9.1f} {fl / td / 1e12:6.1f} | {tw * 1e6:8.1f} {fl / tw / 1e12:6.1f} {td / tw:5.2f} | "
              f"{t1 * 1e6:8.1f} {t2 * 1e6:8.1f} {gfl / t2 / 1e12:6.1f} {t3 * 1e6:7.1f} {lib.prn_gemm_batched_nt_splits(M, C, P, 36, ops.opts_ref())}", flush=True)


if __name__ == "__main__":
    wgrad_main()
