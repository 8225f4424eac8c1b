#!/usr/bin/env python
"""Host time per autograd Function class (forward on the main thread, backward on the autograd thread) for one training
step enqueued while the GPU is parked: where the Python side of the step goes."""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import blocks, ops, timer, losses as L  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss, TargetPrefetcher  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402

timer.disable_all()
torch.set_num_threads(4)
dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "8"))
set_cfg("PlaneRecNet_101_config")
torch.manual_seed(0)
net = PlaneRecNet(cfg)
net.init_head_weights()
net = net.to(dev).train()
crit = PlaneRecNetLoss().to(dev)
opt = __import__("planerecnet_amd.optim", fromlist=["FusedAdam"]).FusedAdam(net.parameters(), lr=1e-4)
images, inst, depths = bench.synth_batch(B, 480, 640, 1000, dev)
pf = TargetPrefetcher(crit)
pf.submit(inst, (480, 640))
pf.submit(inst, (480, 640))
ops.set_wgrad_async(True)

acc = defaultdict(lambda: [0, 0.0])
ON = [False]


def wrap(cls, name):
    f = getattr(cls, name)

    def timed(*a, **k):
        if not ON[0]:
            return f(*a, **k)
        t0 = time.perf_counter()
        r = f(*a, **k)
        e = acc[cls.__name__ + "." + name]
        e[0] += 1
        e[1] += time.perf_counter() - t0
        return r
    setattr(cls, name, staticmethod(timed))


for mod in (ops, L, blocks):
    for v in list(vars(mod).values()):
        if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function:
            wrap(v, "forward")
            wrap(v, "backward")

# C-ABI calls: time inside the library (HIP launches) per entry point
class _TimedLib:
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)
        object.__setattr__(self, "_cache", {})

    def __getattr__(self, name):
        c = self._cache.get(name)
        if c is None:
            f = getattr(self._lib, name)

            def timed(*a):
                if not ON[0]:
                    return f(*a)
                t0 = time.perf_counter()
                r = f(*a)
                e = acc["lib." + name]
                e[0] += 1
                e[1] += time.perf_counter() - t0
                return r
            c = self._cache[name] = timed
        return c


ops.lib = _TimedLib(ops.lib)
blocks.lib = ops.lib
_te, _tel = torch.empty, torch.empty_like


def _timed_empty(*a, **k):
    if not ON[0]:
        return _te(*a, **k)
    t0 = time.perf_counter()
    r = _te(*a, **k)
    e = acc["torch.empty"]
    e[0] += 1
    e[1] += time.perf_counter() - t0
    return r


def _timed_empty_like(*a, **k):
    if not ON[0]:
        return _tel(*a, **k)
    t0 = time.perf_counter()
    r = _tel(*a, **k)
    e = acc["torch.empty_like"]
    e[0] += 1
    e[1] += time.perf_counter() - t0
    return r


torch.empty, torch.empty_like = _timed_empty, _timed_empty_like
phase = defaultdict(float)
FIXED = {}


def step():
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    if os.environ.get("FIXED_TARGETS"):
        if "t" not in FIXED:
            FIXED["t"] = pf.get(depths, dev)
        t = FIXED["t"]
    else:
        t = pf.get(depths, dev)
        pf.submit(inst, (480, 640))
    t1 = time.perf_counter()
    out = net(images)
    t2 = time.perf_counter()
    losses = crit(net, *out, inst, depths, targets=t)
    tot = sum(losses.values()).sum()
    t3 = time.perf_counter()
    tot.backward()
    ops.wgrad_join()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    if ON[0]:
        for k, v in (("prep", t1 - t0), ("net_fwd", t2 - t1), ("loss_fwd", t3 - t2), ("backward", t4 - t3), ("adam", t5 - t4)):
            phase[k] += v


for _ in range(6):
    step()
N = 5
ON[0] = True
for _ in range(N):
    torch.cuda.synchronize()
    torch.cuda._sleep(int(0.25 * 2.4e9))
    step()
torch.cuda.synchronize()
print("phases (ms/step):", {k: round(v / N * 1e3, 1) for k, v in phase.items()}, "total", round(sum(phase.values()) / N * 1e3, 1))
print("%-34s %8s %10s %9s" % ("function", "calls", "ms/step", "us/call"))
for k, (c, t) in sorted(acc.items(), key=lambda x: -x[1][1]):
    print("%-34s %8.1f %10.2f %9.1f" % (k, c / N, t / N * 1e3, t / c * 1e6))
print("sum inside Functions: %.1f ms/step" % (sum(t for _, t in acc.values()) / N * 1e3))
pf.close()
