"""CPU: the process-based target prefetcher (losses.TargetPrefetcher, workers="process") produces exactly what the in-process
one does -- SOLOv2 targets and the virtual-normal triplet draws (numpy global RNG stream continued in the worker) -- over
consecutive batches, and a worker start does not re-run the caller's script."""
import numpy as np
import pytest
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
            ft, fv, early = pf.queue.popleft()
            if early is not None:                                # the receiver thread owns the pipes until it is done
                early.result()
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


@pytest.mark.parametrize("early", ["0", "1"])
def test_prefetcher_is_fifo_and_discard_resynchronises(early, monkeypatch):
    """get() order == submit() order with several batches in flight; discard() leaves no stale batch behind.  early=1: worker
    results are received by the helper thread (PRN_PREFETCH_EARLY) while the caller asks for the same futures."""
    monkeypatch.setenv("PRN_PREFETCH_EARLY", early)
    set_cfg("PlaneRecNet_50_config")
    crit = PlaneRecNetLoss()
    batches = [bench.synth_batch(1, 480, 640, 1000 + i, torch.device("cpu"))[1] for i in range(3)]
    for workers in ("thread", "process"):
        pf = TargetPrefetcher(crit, workers=workers)
        try:
            want = [crit.prepare_host(b, (480, 640), None, False, pin=False)["ins_labels"] for b in batches]
            for b in batches:
                pf.submit(b, (480, 640))
            got = []
            while pf.queue:
                ft, fv = pf.queue.popleft()[:2]
                got.append(ft.result()["ins_labels"])
                fv.result()
            assert all(torch.equal(a, b) for a, b in zip(got, want)), workers
            pf.submit(batches[0], (480, 640))
            pf.submit(batches[1], (480, 640))
            pf.discard()
            assert not pf.queue
            pf.submit(batches[2], (480, 640))
            ft, fv, early = pf.queue.popleft()
            if early is not None:                                # the receiver thread owns the pipes until it is done
                early.result()
            assert torch.equal(ft.result()["ins_labels"], want[2]), workers
            fv.result()
        finally:
            pf.close()


def test_pack_tree_round_trip_keeps_dtypes_shapes_and_alignment():
    """Worker results travel as ONE byte tensor + a skeleton (losses._pack_tree / _unpack_tree)."""
    from planerecnet_amd.losses import _pack_tree, _unpack_tree
    g = torch.Generator().manual_seed(0)
    tree = {"a": torch.randint(0, 255, (7, 3, 5), generator=g, dtype=torch.uint8), "n": 3, "s": "deferred",
            "l": [torch.randn(4, generator=g, dtype=torch.float64), torch.arange(9, dtype=torch.int32).view(3, 3).t()],     # non-contiguous view
            "e": torch.zeros(0, dtype=torch.int64), "nested": {"b": torch.tensor([True, False, True]), "t": (1, 2.5, None)}}
    skel, blob = _pack_tree(tree)
    assert blob.dtype == torch.uint8 and blob.dim() == 1
    back = _unpack_tree(skel, blob.clone())
    assert back["n"] == 3 and back["s"] == "deferred" and back["nested"]["t"] == (1, 2.5, None)
    for got, want in ((back["a"], tree["a"]), (back["l"][0], tree["l"][0]), (back["l"][1], tree["l"][1]), (back["e"], tree["e"]),
                      (back["nested"]["b"], tree["nested"]["b"])):
        assert got.dtype == want.dtype and got.shape == want.shape and torch.equal(got, want)

    def offsets(v):
        if isinstance(v, tuple) and len(v) == 4 and v[0] == "__t__":
            yield v[1]
        elif isinstance(v, dict):
            for x in v.values():
                yield from offsets(x)
        elif isinstance(v, (list, tuple)):
            for x in v:
                yield from offsets(x)
    assert all(o % 16 == 0 for o in offsets(skel))
