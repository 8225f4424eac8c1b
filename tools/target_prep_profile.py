import sys, time, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from planerecnet_amd.config import cfg, set_cfg
from planerecnet_amd.losses import PlaneRecNetLoss
import bench
set_cfg("PlaneRecNet_101_config")
torch.set_num_threads(2)
crit = PlaneRecNetLoss()
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, torch.device("cpu"))
for name, fn in (("targets", lambda: crit.prepare_host(inst, (480, 640), None, False, pin=False)),
                 ("vnl", lambda: crit.vnl.prepare_host([{k: g[k] for k in ("masks", "plane_paras", "k_matrix")} for g in inst], (480, 640), pin=False))):
    fn()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    print(name, "ms per batch", (time.perf_counter() - t0) / 3 * 1e3)
    pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(8)
