#!/usr/bin/env python
"""Main-thread cost of the per-step target hand-over (TargetPrefetcher.get / submit), piece by piece."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss, TargetPrefetcher  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    set_cfg("PlaneRecNet_101_config")
    crit = PlaneRecNetLoss().to(dev)
    images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
    pf = TargetPrefetcher(crit)
    pf.submit(inst, (480, 640))
    pf.submit(inst, (480, 640))
    acc = {}
    for it in range(12):
        time.sleep(0.06)
        t0 = time.perf_counter()
        ft, fv = pf.queue.popleft()
        h = ft.result()
        t1 = time.perf_counter()
        h["vnl"] = fv.result()
        t2 = time.perf_counter()
        t3 = time.perf_counter()
        t = crit.upload(h, depths, dev)
        t4 = time.perf_counter()
        pf.submit(inst, (480, 640))
        t5 = time.perf_counter()
        torch.cuda.synchronize()
        if it >= 4:
            for k, v in (("recv targets", t1 - t0), ("recv vnl", t2 - t1), ("pin", t3 - t2), ("upload", t4 - t3), ("submit", t5 - t4)):
                acc[k] = acc.get(k, 0.0) + v * 1e3 / 8
    print({k: round(v, 2) for k, v in acc.items()}, "total %.2f ms" % sum(acc.values()))
    pf.close()


if __name__ == "__main__":
    main()
