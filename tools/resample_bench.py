"""The gather-type elementwise kernels (bilinear resize forward / adjoint, reflect- and replicate-pad folds) at the shapes of the training step:
time and bytes moved per launch.   python tools/resample_bench.py   (PRN_LIB=<other build> for an A/B)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from planerecnet_amd import ops  # noqa: E402

lib, _p, _s = ops.lib, ops._p, ops._stream


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


def row(name, t, nbytes):
    print("%-52s %8.1f us  %7.1f MB  %6.0f GB/s" % (name, t * 1e6, nbytes / 1e6, nbytes / t / 1e9), flush=True)


def main():
    dev = "cuda"
    for (B, C, H, W, up2) in [(8, 64, 240, 320, 0), (8, 256, 120, 160, 1), (8, 128, 120, 160, 0)]:
        Hv, Wv = (2 * H, 2 * W) if up2 else (H, W)
        dp = torch.randn(B, C, Hv + 2, Wv + 2, device=dev)
        dx = torch.empty(B, C, H, W, device=dev)
        row("pad_fold %dx%dx%dx%d up2=%d" % (B, C, H, W, up2), timeit(lambda: lib.prn_pad_fold(_p(dp), _p(dx), B, C, H, W, up2, _s())), 4.0 * (dp.numel() + dx.numel()))
    for (B, C, H, W) in [(8, 64, 240, 320), (8, 128, 120, 160)]:
        dp = torch.randn(B, C, H + 2, W + 2, device=dev)
        dx = torch.empty(B, C, H, W, device=dev)
        row("replicate_fold %dx%dx%dx%d" % (B, C, H, W), timeit(lambda: lib.prn_replicate_fold(_p(dp), _p(dx), B, C, H, W, _s())), 4.0 * (dp.numel() + dx.numel()))
    for (BC, H, W, Ho, Wo) in [(8 * 128, 60, 80, 120, 160), (8 * 256, 30, 40, 120, 160), (8 * 128, 30, 40, 60, 80), (8 * 256, 15, 20, 40, 40), (8 * 256, 60, 80, 40, 40), (8 * 256, 60, 80, 36, 36), (8 * 256, 30, 40, 24, 24)]:
        x = torch.randn(BC, H, W, device=dev)
        y = torch.empty(BC, Ho, Wo, device=dev)
        add = torch.randn(BC, Ho, Wo, device=dev)
        row("resize fwd   %dx(%dx%d -> %dx%d)" % (BC, H, W, Ho, Wo), timeit(lambda: lib.prn_resize_bilinear_fwd(_p(x), _p(y), BC, H, W, Ho, Wo, _s())), 4.0 * (x.numel() + y.numel()))
        row("resize fwd + addend", timeit(lambda: lib.prn_resize_bilinear_add_fwd(_p(x), _p(add), _p(y), BC, H, W, Ho, Wo, _s())), 4.0 * (x.numel() + 2 * y.numel()))
        dx = torch.empty(BC, H, W, device=dev)
        row("resize adjoint %dx(%dx%d <- %dx%d)" % (BC, H, W, Ho, Wo), timeit(lambda: lib.prn_resize_bilinear_bwd(_p(y), _p(dx), BC, H, W, Ho, Wo, _s())), 4.0 * (x.numel() + y.numel()))


if __name__ == "__main__":
    main()
