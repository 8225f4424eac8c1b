#!/usr/bin/env python
"""Is the training step bound by the host or by the GPU?  Each measured step is enqueued while the GPU is parked behind a
spin kernel: H = host time to enqueue the whole step with an idle-waiting GPU (no back-pressure), G = GPU time to drain it
once released (no host dependency).  S = the ordinary free-running step time for comparison."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import ops, timer  # noqa: E402
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
if os.environ.get("TARGETS", "device") == "device":      # GT-only loss preparation: device kernels (default) or the host workers
    from planerecnet_amd.targets import DeviceTargetBuilder
    pf = DeviceTargetBuilder(crit)
else:
    pf = TargetPrefetcher(crit)
pf.submit(inst, (480, 640))
pf.submit(inst, (480, 640))
ops.set_wgrad_async(True)


FIXED = [None]


PH = {}


def step():
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t1 = time.perf_counter()
    if os.environ.get("FIXED_TARGETS"):                 # targets computed once: no worker thread competing for the GIL
        if FIXED[0] is None:
            FIXED[0] = pf.get(depths, dev)
        t = FIXED[0]
    else:
        t = pf.get(depths, dev, overlap=True)
        pf.submit(inst, (480, 640))
    t2 = time.perf_counter()
    out = net(images)
    t3 = time.perf_counter()
    losses = crit(net, *out, inst, depths, targets=t)
    tot = sum(losses.values()).sum()
    t4 = time.perf_counter()
    tot.backward()
    t5 = time.perf_counter()
    ops.wgrad_join()
    t6 = time.perf_counter()
    opt.step()
    t7 = time.perf_counter()
    for k, v in (("zero_grad", t1 - t0), ("targets", t2 - t1), ("forward", t3 - t2), ("loss", t4 - t3), ("backward", t5 - t4), ("join", t6 - t5), ("adam", t7 - t6)):
        PH.setdefault(k, []).append(v * 1e3)


HOST_ONLY = bool(os.environ.get("HOST_ONLY"))          # several copies of this script share ONE GPU (tools/host_probe_8ranks.sh): only H means something
for _ in range(3 if HOST_ONLY else 6):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(0 if HOST_ONLY else 10):
    step()
torch.cuda.synchronize()
S = (time.perf_counter() - t0) / 10 * 1e3
Hs, Gs, Cs = [], [], []
for _ in range(6):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(0.25 * 2.4e9))            # ~250 ms at 2.4 GHz... clock-dependent: only has to outlast the enqueue
    e0.record()
    t0, c0 = time.perf_counter(), time.process_time()
    step()
    Hs.append((time.perf_counter() - t0) * 1e3)
    Cs.append((time.process_time() - c0) * 1e3)
    e1.record()
    torch.cuda.synchronize()
    Gs.append(e0.elapsed_time(e1))
print("free-running step S = %.1f ms" % S)
print("host phases of the parked steps (ms): " + ", ".join("%s %.2f" % (k, sum(v[-6:]) / 6) for k, v in PH.items()))
print("host enqueue H (GPU parked): " + " ".join("%.1f" % h for h in Hs))
print("process CPU during the enqueue (all threads): " + " ".join("%.1f" % c for c in Cs))
print("GPU drain    G (no host dep): " + " ".join("%.1f" % g for g in Gs))
if hasattr(pf, "host_ms"):
    print("target preparation, host side (submit + get): %.1f ms wall / %.1f ms CPU per step (mode %s)" % (pf.host_ms / pf.calls, pf.host_cpu_ms / pf.calls, os.environ.get("TARGETS", "device")))
pf.close()
