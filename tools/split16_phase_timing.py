"""Where a split16 GEMM launch spends its time: per workgroup, the 100 MHz wall clock at kernel entry / loop entry / loop exit / kernel exit and
the compute unit it ran on (csrc/prn_gemm_split.hip built with -DPRN_S16_TIMING into planerecnet_amd/build/libprn_s16timing.so -- a side build,
not the product library).  Build the side library here or on the GPU box (after the product build, whose objects it links), then run on the GPU box:
    python tools/split16_phase_timing.py --build
    PRN_LIB=planerecnet_amd/build/libprn_s16timing.so python tools/split16_phase_timing.py"""
import ctypes
import glob
import os
import subprocess
import sys

if "--build" in sys.argv:
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "planerecnet_amd")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = os.path.join(pkg, "build", "prn_gemm_split_timing.o")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-DPRN_S16_TIMING", "-c",
                           os.path.join(pkg, "csrc", "prn_gemm_split.hip"), "-o", obj])
    others = [o for o in glob.glob(os.path.join(pkg, "build", "prn_*.o")) if not o.endswith(("prn_gemm_split.o", "_timing.o", "_old.o"))]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(pkg, "build", "libprn_s16timing.so")] + others + [obj])
    print("built", os.path.join(pkg, "build", "libprn_s16timing.so"))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from planerecnet_amd import ops  # noqa: E402
from planerecnet_amd._lib import lib  # noqa: E402

B = 8
SHAPES = [("s3 1024<-256 @30x40 +addend", 1024, 256, 30, 40, True), ("s3 256<-1024 @30x40", 256, 1024, 30, 40, False),
          ("s2 512<-128 @60x80 +addend", 512, 128, 60, 80, True), ("s2 128<-512 @60x80", 128, 512, 60, 80, False),
          ("fpn 256<-256 @120x160", 256, 256, 120, 160, False)]


def main():
    rd = lib.prn_debug_s16_timing
    rd.restype = ctypes.c_int
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rit = lib.prn_debug_s16_iter
    rit.restype = ctypes.c_int
    rit.argtypes = [ctypes.c_void_p, ctypes.c_int]
    ops.set_split_gemm(mode=2)
    for name, M, C, H, W, add in SHAPES:
        x = torch.relu(torch.randn(B, C, H, W, device="cuda"))
        w = torch.randn(M, C, 1, 1, device="cuda") * 0.05
        addend = torch.randn(B, M, H, W, device="cuda") if add else None
        for _ in range(5):
            ops.conv2d(x, w, addend=addend)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.conv2d(x, w, addend=addend)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        buf = np.zeros(8192 * 6, dtype=np.int64)
        assert rd(buf.ctypes.data, buf.size) == 0
        d = buf.reshape(8192, 6)
        d = d[d[:, 0] != 0]
        d = d[d[:, 0] > d[:, 0].max() - 30000]                               # (slots of an earlier, larger grid keep their old stamps: 300 us)
        t0 = d[:, 0].min()
        ent, lp0, lp1, ex = [(d[:, i] - t0) * 0.01 for i in range(4)]        # us
        cu = d[:, 4]
        print("%-30s %6.1f us/launch (back to back)  %d workgroups on %d distinct CUs" % (name, us, len(d), len(set(cu.tolist()))))
        print("   kernel span (first entry -> last exit) %6.1f us" % ex.max())
        for lab, v in (("entry", ent), ("loop entry", lp0), ("loop exit", lp1), ("exit", ex), ("prologue", lp0 - ent), ("loop", lp1 - lp0), ("epilogue", ex - lp1), ("whole", ex - ent)):
            print("   %-10s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % ((lab,) + tuple(np.percentile(v, [0, 10, 50, 90, 100]))))
        per = {}
        for c in cu.tolist():
            per[c] = per.get(c, 0) + 1
        hist = {}
        for n in per.values():
            hist[n] = hist.get(n, 0) + 1
        print("   workgroups per CU: %s" % sorted(hist.items()))
        it = np.zeros(2 * 64 * 4, dtype=np.int64)
        assert rit(it.ctypes.data, it.size) == 0
        it = it.reshape(2, 64, 4)
        for g in range(2):
            rows = it[g][it[g][:, 0] > d[:, 0].max() - 30000]
            if len(rows) == 0:
                continue
            base = rows[0, 0]
            print("   workgroup %s wave 0, per iteration [after barrier, MFMAs issued, next activations arrived, end] in us since its first barrier:" % ("0" if g == 0 else "mid-grid"))
            for i, rw in enumerate(rows[:12]):
                print("      it %2d: %s   (barrier->mfma issued %.2f, ->loads in %.2f, cut+issue %.2f)" % (i, " ".join("%6.2f" % ((v - base) * 0.01) for v in rw), (rw[1] - rw[0]) * 0.01, (rw[2] - rw[1]) * 0.01, (rw[3] - rw[2]) * 0.01))


if __name__ == "__main__":
    main()
