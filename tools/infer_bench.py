#!/usr/bin/env python
"""Forward-only timings of the other BASELINE.json configurations (not bench.py lines; recorded in DESIGN.md):
  C2  PlaneRecNet_50_config, batch 8, 480x640, eval-mode forward + on-device post-process
  C5  PlaneRecNet_101_config, max_size=960 -> 736x960, batch 4, eval-mode forward + post-process"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import timer  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402

timer.disable_all()
dev = torch.device("cuda:0")
for name, config, B, H, W in (("C2", "PlaneRecNet_50_config", 8, 480, 640), ("C5", "PlaneRecNet_101_config", 4, 736, 960)):
    set_cfg(config)
    torch.manual_seed(0)
    net = PlaneRecNet(cfg)
    net.init_head_weights()
    net = net.to(dev).eval()
    x = torch.randn(B, 3, H, W, device=dev)
    with torch.no_grad():
        for _ in range(3):
            out = net(x)
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            out = net(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    dets = sum(0 if r["pred_scores"] is None else len(r["pred_scores"]) for r in out)
    print("%s %s B=%d %dx%d: %.1f ms / batch, %.1f img/s (detections in last batch: %d)" % (name, config, B, H, W, dt * 1e3, B / dt, dets))
