"""Weight gradient of the plain-GEMM layers: the fp16-piece kernel (csrc/prn_wgrad16.hip) against the fp32 MFMA kernel, shape by shape,
as single launches, as grouped launches of 8 layers, and as the 36 batched products of the Winograd weight gradient.  B = 8, 480x640 maps.
    python tools/wgrad16_bench.py            (PRN_WGRAD_WGS=512 plans both for the deferred side stream, as in the training step)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from planerecnet_amd import ops  # noqa: E402

lib, _p, _stream = ops.lib, ops._p, ops._stream
B = 8
SHAPES = [("s3 1024<-256 @30x40", 1024, 256, 30, 40), ("s3 256<-1024 @30x40", 256, 1024, 30, 40), ("s2 512<-128 @60x80", 512, 128, 60, 80),
          ("s2 128<-512 @60x80", 128, 512, 60, 80), ("s1 256<-64 @120x160", 256, 64, 120, 160), ("s1 64<-256 @120x160", 64, 256, 120, 160),
          ("s4 2048<-512 @15x20", 2048, 512, 15, 20), ("s4 512<-2048 @15x20", 512, 2048, 15, 20), ("fpn 256<-256 @120x160", 256, 256, 120, 160),
          ("fpn 256<-512 @60x80", 256, 512, 60, 80), ("prior 256<-3728 @30x40", 256, 3728, 30, 40), ("mask 128<-256 @120x160", 128, 256, 120, 160)]


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def main():
    print("%-26s %7s | %9s %6s | %9s %6s %5s | %11s %11s" % ("dW shape", "GFLOP", "fp32 us", "TF/s", "fp16x2 us", "TF/s", "x", "grp8 fp32", "grp8 fp16x2"))
    for name, M, C, H, W in SHAPES:
        x = torch.relu(torch.randn(B, C, H, W, device="cuda"))
        dy = torch.randn(B, M, H, W, device="cuda")
        fl = 2.0 * M * C * B * H * W
        t = {}
        for mode in (0, 2):
            ops.set_split_gemm(wgrad=mode)
            t[mode] = timeit(lambda: ops.conv_wgrad_raw(x, dy, M, 1, 1, 0, ops.IN_ZERO))
        tg = {}
        if B * H * W <= 40000:
            xs = [torch.relu(torch.randn(B, C, H, W, device="cuda")) for _ in range(8)]
            dys = [torch.randn(B, M, H, W, device="cuda") for _ in range(8)]
            for mode in (0, 2):
                ops.set_split_gemm(wgrad=mode)
                tg[mode] = timeit(lambda: ops.conv_wgrad_grouped_raw(xs, dys, M, 1, 1, 0, ops.IN_ZERO), 10) / 8
        print("%-26s %7.2f | %9.1f %6.1f | %9.1f %6.1f %5.2f | %11s %11s" % (name, fl / 1e9, t[0] * 1e6, fl / t[0] / 1e12, t[2] * 1e6, fl / t[2] / 1e12, t[0] / t[2],
                                                                             ("%.1f" % (tg[0] * 1e6)) if tg else "-", ("%.1f" % (tg[2] * 1e6)) if tg else "-"), flush=True)
    print("\nWinograd weight-gradient products: 36 x [M x P] x [C x P]^T (partials only)")
    for name, C, H, W, M in [("256->256 @30x40", 256, 30, 40, 256), ("256->256 @120x160", 256, 120, 160, 256), ("128->128 @60x80", 128, 60, 80, 128),
                             ("256->128 @120x160", 256, 120, 160, 128), ("512->512 @15x20", 512, 15, 20, 512), ("64->64 @120x160", 64, 120, 160, 64)]:
        P = lib.prn_winograd_tiles(B, H, W)
        A = torch.randn(36, M, P, device="cuda")
        Bm = torch.randn(36, C, P, device="cuda")
        fl = 2.0 * 36 * M * C * P
        t, sp = {}, {}
        for mode in (0, 2):
            ops.set_split_gemm(wgrad=mode)
            oref = ops.opts_ref()
            sp[mode] = lib.prn_gemm_batched_nt_splits(M, C, P, 36, oref)
            ws = torch.empty(sp[mode] * 36 * M * C, device="cuda")
            t[mode] = timeit(lambda: lib.prn_gemm_batched_nt(M, C, P, 36, _p(A), _p(Bm), _p(ws), oref, _stream()))
        print("%-26s %7.2f | %9.1f %6.1f (S %3d) | %9.1f %6.1f (S %3d) %5.2f" % (name, fl / 1e9, t[0] * 1e6, fl / t[0] / 1e12, sp[0], t[2] * 1e6, fl / t[2] / 1e12, sp[2], t[0] / t[2]), flush=True)


if __name__ == "__main__":
    main()
