#!/usr/bin/env python
"""Sweep PRN_CONV_FORCE configurations over the conv shapes (forward launches only); one subprocess per configuration."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
from planerecnet_amd import ops
from tools.conv_bench import B, SHAPES, timeit
dev = torch.device("cuda:0")
out = []
for name, C, H, W, M, K, stride, pad, mode in SHAPES:
    x = torch.randn(B, C, H, W, device=dev); w = torch.randn(M, C, K, K, device=dev) * (C*K*K) ** -0.5
    Ho, Wo = ops._out_hw(H, W, K, stride, pad, mode)
    t = timeit(lambda: ops.conv_fwd_raw(x, w, None, None, M, K, stride, pad, Ho, Wo, mode), reps=5)
    out.append("%%.1f" %% (t * 1e6))
print("RES " + " ".join(out))
''' % ROOT

configs = [None] + [(tm, tn, bk, s) for (tm, tn, bk) in ((1, 1, 16), (2, 1, 16), (2, 2, 16)) for s in (1, 2, 3, 4)]
sys.path.insert(0, ROOT)
from tools.conv_bench import SHAPES  # noqa: E402
rows = {}
for cfg in configs:
    env = dict(os.environ)
    if cfg:
        env["PRN_CONV_FORCE"] = ",".join(map(str, cfg))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RES ")]
    rows[cfg] = line[0].split()[1:] if line else None
    if not line:
        print("FAILED", cfg, r.stderr[-300:])
print("%-34s" % "shape" + "".join("%11s" % ("auto" if c is None else "%d%d/%d/s%d" % c) for c in configs))
for i, sh in enumerate(SHAPES):
    vals = [float(rows[c][i]) if rows[c] else float("nan") for c in configs]
    best = min(v for v in vals[1:] if v == v)
    print("%-34s" % sh[0][:34] + "".join(("%10.0f%s" % (v, "*" if v == best else " ")) for v in vals))
