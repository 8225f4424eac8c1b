"""Depth-error metrics (reference eval.py:164-207): the oracle against the reference's golden values (CPU), the fused HIP
reduction against the oracle and the golden values (GPU)."""
import os

import numpy as np
import pytest
import torch


def make_pair(seed, H=480, W=640):          # the generator of tests/golden/make_golden_metrics.py
    g = torch.Generator().manual_seed(seed)
    gt = 0.3 + 6.0 * torch.rand(1, H, W, generator=g)
    gt[torch.rand(1, H, W, generator=g) < 0.1] = 0.0
    pred = gt * (1.0 + 0.2 * torch.randn(1, H, W, generator=g)) + 0.05
    pred[torch.rand(1, H, W, generator=g) < 0.02] = 75.0
    pred[torch.rand(1, H, W, generator=g) < 0.02] = 0.2
    return pred, gt


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_metrics_equal_reference_golden(golden_dir, seed):
    from oracle.metrics_ref import compute_depth_metrics_ref
    fx = np.load(os.path.join(golden_dir, "depth_metrics.npz"))
    pred, gt = make_pair(seed)
    got = [float(v) for v in compute_depth_metrics_ref(pred, gt, float(fx["min_depth"]), float(fx["max_depth"]))]
    assert np.allclose(got, fx["seed%d" % seed], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,shape", [(0, (480, 640)), (1, (480, 640)), (2, (480, 640)), (3, (17, 23)), (4, (736, 960))])
def test_device_metrics_equal_oracle_and_golden(golden_dir, seed, shape):
    from oracle.metrics_ref import compute_depth_metrics_ref
    from planerecnet_amd.config import cfg, set_cfg
    from planerecnet_amd.metrics import compute_depth_metrics
    set_cfg("PlaneRecNet_50_config")
    fx = np.load(os.path.join(golden_dir, "depth_metrics.npz"))
    assert float(fx["min_depth"]) == cfg.dataset.min_depth and float(fx["max_depth"]) == cfg.dataset.max_depth
    pred, gt = make_pair(seed, *shape)
    got = compute_depth_metrics(pred.cuda(), gt.cuda(), median_scaling=True)
    assert all(not t.is_cuda for t in got) and len(got) == 8           # the reference returns CPU tensors
    got = [float(v) for v in got]
    ref = [float(v) for v in compute_depth_metrics_ref(pred, gt, cfg.dataset.min_depth, cfg.dataset.max_depth)]
    # fp32 terms summed in fp64 on the device vs fp32 pairwise sums on the CPU: 2e-5 relative; the threshold ratios are counts
    assert np.allclose(got, ref, rtol=2e-5, atol=1e-7), (got, ref)
    if shape == (480, 640) and seed < 3:
        assert np.allclose(got, fx["seed%d" % seed], rtol=2e-5, atol=1e-7)
    assert float(compute_depth_metrics(pred.cuda(), gt.cuda(), median_scaling=False)[7]) == 0.0


@pytest.mark.gpu
def test_device_metrics_reject_cpu_tensors():
    from planerecnet_amd.metrics import compute_depth_metrics
    with pytest.raises(RuntimeError):
        compute_depth_metrics(torch.ones(1, 4, 4), torch.ones(1, 4, 4))
