"""Where a windowed DCNv2 forward launch (csrc/prn_dcnv2.hip: dcnv2_fwd2_kernel) spends its time: wave 0 of every workgroup sums the shader
clocks of the five phases of its K loop (csrc built with -DPRN_DCN2_TIMING into planerecnet_amd/build/libprn_dcn2timing.so -- a side build, not
the product library).
    python tools/dcn2_phase_timing.py --build
    PRN_LIB=planerecnet_amd/build/libprn_dcn2timing.so python tools/dcn2_phase_timing.py"""
import ctypes
import glob
import os
import subprocess
import sys

if "--build" in sys.argv:
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "planerecnet_amd")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = os.path.join(pkg, "build", "prn_dcnv2_timing.o")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-DPRN_DCN2_TIMING", "-c",
                           os.path.join(pkg, "csrc", "prn_dcnv2.hip"), "-o", obj])
    others = [o for o in glob.glob(os.path.join(pkg, "build", "prn_*.o")) if not o.endswith(("prn_dcnv2.o", "_timing.o", "_old.o"))]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(pkg, "build", "libprn_dcn2timing.so")] + others + [obj])
    print("built", os.path.join(pkg, "build", "libprn_dcn2timing.so"))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from planerecnet_amd import ops  # noqa: E402
from planerecnet_amd._lib import lib  # noqa: E402

B = 8
SHAPES = [("128ch 120x160 s2", 128, 120, 160, 2), ("128ch 60x80 s1", 128, 60, 80, 1), ("256ch 60x80 s2", 256, 60, 80, 2), ("256ch 30x40 s1", 256, 30, 40, 1),
          ("512ch 30x40 s2", 512, 30, 40, 2)]


def main():
    rd = lib.prn_debug_dcn2_timing
    rd.restype = ctypes.c_int
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rms = float(sys.argv[sys.argv.index("--offsets") + 1]) if "--offsets" in sys.argv else 0.6
    for name, C, H, W, s in SHAPES:
        M = C
        Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
        x = torch.relu(torch.randn(B, C, H, W, device="cuda"))
        w = torch.randn(M, C, 3, 3, device="cuda") * (9 * C) ** -0.5
        om = torch.randn(B, 27, Ho, Wo, device="cuda") * rms
        mo = max(H, W) / 4.0
        table = ops.dcn_table(x.shape, M, om, None, s, 1, 1, mo)
        for _ in range(5):
            ops.dcn_fwd_raw(x, table, w, None, s, 1, 1, mo)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.dcn_fwd_raw(x, table, w, None, s, 1, 1, mo)
        e.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(e) / 20 * 1e3
        buf = np.zeros(4096 * 10, dtype=np.int64)
        assert rd(buf.ctypes.data, buf.size) == 0
        d = buf.reshape(4096, 10)
        d = d[d[:, 6] != 0]
        d = d[d[:, 6] > d[:, 6].max() - 100000]
        it = d[:, 5].astype(np.float64)
        print("%-18s %6.1f us/launch (+ reduce); %d workgroups, %.1f iterations each; window classes %s; loop wall (100 MHz) median %.1f us, first->last %.1f us"
              % (name, us, len(d), it.mean(), sorted(set(d[:, 8].tolist())), np.median(d[:, 7] - d[:, 6]) * 0.01, (d[:, 7].max() - d[:, 6].min()) * 0.01))
        tot = d[:, :5].sum(axis=1) / np.maximum(it, 1)
        for lab, i in (("load issue", 0), ("sample", 1), ("mfma loop", 2), ("lds stores", 3), ("barrier", 4)):
            v = d[:, i] / np.maximum(it, 1)
            print("   %-11s cycles per iteration: p10 %7.0f  median %7.0f  p90 %7.0f" % ((lab,) + tuple(np.percentile(v, [10, 50, 90]))))
        print("   %-11s cycles per iteration: p10 %7.0f  median %7.0f  p90 %7.0f   (pure MFMA time of a wave: %d)" % (("total",) + tuple(np.percentile(tot, [10, 50, 90])) + (9 * (4 if M > 128 else 2) * 64,)))
        wnd = d[:, 9]
        print("   windows (rows x cols): min %dx%d  max %dx%d" % (wnd.min() // 1000, wnd.min() % 1000, wnd.max() // 1000, wnd.max() % 1000))


if __name__ == "__main__":
    main()
