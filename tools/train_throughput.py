#!/usr/bin/env python
"""Wall-clock throughput of train.py on synthetic data: time(N2 iterations) - time(N1 iterations), both modes of the
non-finite-loss check (device-side found_inf vs the reference's per-step host read)."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(iters, sync):
    env = dict(os.environ)
    if sync:
        env["PRN_TRAIN_SYNC_LOSS"] = "1"
    else:
        env.pop("PRN_TRAIN_SYNC_LOSS", None)
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--config", "PlaneRecNet_101_config", "--dataset", "synthetic", "--batch_size", "8",
                        "--save_folder", d + "/", "--num_workers", os.environ.get("TRAIN_WORKERS", "2"), "--no_tensorboard", "--synthetic_size", "8000", "--max_iter", str(iters),
                        "--save_interval", "100000"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        return time.time() - t0


if __name__ == "__main__":
    n1, n2 = 101, 501
    for sync in ((False,) if os.environ.get("TRAIN_MODES") == "device" else (False, True)):
        a, b = run(n1, sync), run(n2, sync)
        print("%s: %.1f s for %d it, %.1f s for %d it -> %.1f ms/iteration" % ("host read" if sync else "device skip", a, n1, b, n2, (b - a) / (n2 - n1) * 1e3))
