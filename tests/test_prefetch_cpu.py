"""CPU: the process-based target prefetcher (losses.TargetPrefetcher, workers="process") produces exactly what the in-process
one does -- SOLOv2 targets and the virtual-normal triplet draws (numpy global RNG stream continued in the worker) -- over
consecutive batches, and a worker start does not re-run the caller's script."""
import numpy as np
import torch

import bench
from planerecnet_amd.config import set_cfg
from planerecnet_amd.losses import PlaneRecNetLoss, TargetPrefetcher


def _host_results(workers, inst, n):
    np.random.seed(5)
    crit = PlaneRecNetLoss()
    pf = TargetPrefetcher(crit, workers=workers)
    out = []
    try:
        for _ in range(n):
            pf.submit(inst, (480, 640))
            ft, fv = pf.queue.popleft()
            out.append((ft.result(), fv.result() if fv is not None else None))
    finally:
        pf.close()
    return out


def _same(a, b):
    if torch.is_tensor(a):
        return torch.equal(a, b)
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return np.array_equal(a, b)
    return a == b


def test_process_prefetcher_matches_thread_prefetcher():
    set_cfg("PlaneRecNet_50_config")
    _, inst, _ = bench.synth_batch(2, 480, 640, 1000, torch.device("cpu"))
    pr = _host_results("process", inst, 2)
    th = _host_results("thread", inst, 2)
    for (ta, va), (tb, vb) in zip(pr, th):
        assert _same(ta, tb), "SOLOv2 targets differ between worker process and worker thread"
        assert _same(va, vb), "virtual-normal triplets differ: RNG stream not continued in the worker process"
    assert not _same(pr[0][1]["gid"], pr[1][1]["gid"])       # consecutive batches draw different triplets
