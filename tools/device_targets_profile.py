"""Where the host time of the device target preparation goes (targets.DeviceTargetBuilder.submit / get): wall + CPU time per call, a cProfile
of each, on the benchmark's batch (8 x 480x640, 3-8 planes per image).  Run on the GPU box: python tools/device_targets_profile.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from planerecnet_amd.config import set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss  # noqa: E402
from planerecnet_amd.targets import DeviceTargetBuilder  # noqa: E402

set_cfg("PlaneRecNet_101_config")
torch.set_num_threads(int(os.environ.get("PRN_HOST_THREADS", "4")))
dev = torch.device("cuda:0")
crit = PlaneRecNetLoss().to(dev)
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
tb = DeviceTargetBuilder(crit, seed=0)
hw = (480, 640)
for _ in range(3):
    tb.submit(inst, hw)
for _ in range(5):                                             # warm: allocator pools, pinned buffers
    tb.get(depths, dev, overlap=True)
    tb.submit(inst, hw)
torch.cuda.synchronize()


def timed(fn, n=30):
    w, c = [], []
    for _ in range(n):
        torch.cuda.synchronize()
        t0, c0 = time.perf_counter(), time.thread_time()
        fn()
        w.append(time.perf_counter() - t0)
        c.append(time.thread_time() - c0)
    w.sort(); c.sort()
    return 1e3 * w[len(w) // 2], 1e3 * c[len(c) // 2]


sw, sc = timed(lambda: (tb.get(depths, dev, overlap=True), None)[1] or tb.submit(inst, hw))
print("get + submit per step: wall %.2f ms, thread CPU %.2f ms (median of 30, device idle at the start of each)" % (sw, sc))
for _ in range(12):
    tb.submit(inst, hw)
for name, fn in (("submit", lambda: tb.submit(inst, hw)), ("get", lambda: tb.get(depths, dev, overlap=True))):
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        fn()
    pr.disable()
    print("---- cProfile of 5 x %s (tottime)" % name)
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
