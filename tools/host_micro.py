#!/usr/bin/env python
"""Host cost of the pieces of one operator call on the GPU box (small tensors: the GPU is never the limit)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import ops, _lib  # noqa: E402
from planerecnet_amd.ops import lib, _p, _stream  # noqa: E402

dev = torch.device("cuda:0")


def bench(f, n=3000):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    dt = (time.perf_counter() - t) / n * 1e6
    torch.cuda.synchronize()
    return dt


B, C, H, W = 2, 64, 16, 16
x = torch.randn(B, C, H, W, device=dev)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
w1 = torch.randn(C, C, 1, 1, device=dev)
w3 = torch.randn(C, C, 3, 3, device=dev)
print("torch.empty_like(x) cuda        %.1f us" % bench(lambda: torch.empty_like(x)))
print("torch.empty(n, device, dtype)   %.1f us" % bench(lambda: torch.empty(2 * C, device=dev, dtype=torch.float32)))
print("x.new_empty(n)                  %.1f us" % bench(lambda: x.new_empty(2 * C)))
print("_stream()                       %.1f us" % bench(lambda: _stream()))
y = torch.empty_like(x)
stats = torch.empty(2 * C, device=dev)
ws = torch.empty(2 * C * _lib.BN_SPLITS, device=dev, dtype=torch.float64)
st = _stream()
print("lib.prn_bn_train_fwd only       %.1f us" % bench(lambda: lib.prn_bn_train_fwd(_p(x), _p(stats), _p(g), _p(b), None, _p(y), _p(rm), _p(rv), _p(ws), B, C, H * W, 1e-5, 0.1, 1, st)))
with torch.no_grad():
    print("ops.batch_norm (no grad)        %.1f us" % bench(lambda: ops.batch_norm(x, g, b, rm, rv, True, relu=True)))
    print("ops.conv2d 1x1 (no grad)        %.1f us" % bench(lambda: ops.conv2d(x, w1)))
    ops.WINOGRAD_MIN_TILES = 1
    print("ops.conv2d 3x3 winograd (nograd)%.1f us" % bench(lambda: ops.conv2d(x, w3, None, 1, 1)))
    ops.WINOGRAD = False
    print("ops.conv2d 3x3 direct (nograd)  %.1f us" % bench(lambda: ops.conv2d(x, w3, None, 1, 1)))
    ops.WINOGRAD = True
xg = x.clone().requires_grad_(True)
gg = g.clone().requires_grad_(True)
print("ops.batch_norm (grad)           %.1f us" % bench(lambda: ops.batch_norm(xg, gg, b, rm, rv, True, relu=True)))
w1g = w1.clone().requires_grad_(True)
print("ops.conv2d 1x1 (grad)           %.1f us" % bench(lambda: ops.conv2d(xg, w1g)))
a = torch.randn(1 << 16, device=dev)
print("aten add (1 launch)             %.1f us" % bench(lambda: a + a))
print("aten add_ in place              %.1f us" % bench(lambda: a.add_(1.0)))
e = torch.cuda.Event()
print("event record                    %.1f us" % bench(lambda: e.record()))
s2 = torch.cuda.Stream()
print("s2.wait_stream(cur)             %.1f us" % bench(lambda: s2.wait_stream(torch.cuda.current_stream())))
def ctxsw():
    with torch.cuda.stream(s2):
        pass
print("with torch.cuda.stream(s2)      %.1f us" % bench(ctxsw))
print("x.record_stream(s2)             %.1f us" % bench(lambda: x.record_stream(s2)))
