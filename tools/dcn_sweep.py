#!/usr/bin/env python
"""Tile / split sweep of the fused DCNv2 forward and weight-gradient kernels (PRN_DCN_FWD / PRN_DCN_WGRAD are read once per
process, so every point runs in a fresh interpreter).   python tools/dcn_sweep.py [C H W stride]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONE = r'''
import sys, torch
sys.path.insert(0, %r)
from planerecnet_amd import ops
from tools.conv_bench import timeit
C, H, W, s, which = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
B, M = 8, C
dev = torch.device("cuda:0")
Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
mo = max(H, W) / 4.0
x = torch.randn(B, C, H, W, device=dev); om = torch.randn(B, 27, Ho, Wo, device=dev) * 0.6
w = torch.randn(M, C, 3, 3, device=dev) * 0.02; dy = torch.randn(B, M, Ho, Wo, device=dev)
table = ops.dcn_table(x.shape, M, om, None, s, 1, 1, mo)
if which == "f":
    t = timeit(lambda: ops.dcn_fwd_raw(x, table, w, None, s, 1, 1, mo))
else:
    t = timeit(lambda: ops.dcn_wgrad_raw(x, table, dy, M, s, 1, 1, mo))
print("%%.1f" %% (t * 1e6))
''' % ROOT

shape = sys.argv[1:5] if len(sys.argv) >= 5 else ["256", "30", "40", "1"]
for which, var, tms, splits in (("f", "PRN_DCN_FWD", (1, 2, 4), (1, 2, 3, 4, 5, 6, 8)), ("w", "PRN_DCN_WGRAD", (1, 2, 4), (4, 8, 12, 16, 21, 28, 33))):
    print("fused %s, %s: rows = tile height / 64, columns = splits %s (us)" % ("forward" if which == "f" else "weight gradient", "x".join(shape), splits))
    for tm in tms:
        row = []
        for sp in splits:
            env = dict(os.environ, **{var: "%d,%d" % (tm, sp)})
            r = subprocess.run([sys.executable, "-c", ONE] + shape + [which], capture_output=True, text=True, env=env)
            row.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "err")
        print("  tm=%d: %s" % (tm, "  ".join("%7s" % v for v in row)), flush=True)
