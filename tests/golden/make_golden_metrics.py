"""Golden vectors for the depth-error metrics -- runs ONLY in the build container (needs /root/reference).

    python tests/golden/make_golden_metrics.py

Calls the reference's own `compute_depth_metrics` (eval.py:164-207, imported under oracle/ref_shim.py; its
`.type(torch.cuda.DoubleTensor)` casts are mapped to the CPU type) on seeded depth pairs, asserts the oracle restatement
(oracle/metrics_ref.py) returns the same eight numbers, and writes them to tests/golden/depth_metrics.npz.  Inputs are
regenerated from the seed by `make_pair` (duplicated in the tests)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
from oracle import ref_shim  # noqa: E402
from oracle.metrics_ref import compute_depth_metrics_ref  # noqa: E402


def make_pair(seed, H=480, W=640):
    """Ground truth with holes (0 = no measurement) and values under 0.5 m, a prediction with noise, outliers and values outside
    [min_depth, max_depth]."""
    g = torch.Generator().manual_seed(seed)
    gt = 0.3 + 6.0 * torch.rand(1, H, W, generator=g)
    gt[torch.rand(1, H, W, generator=g) < 0.1] = 0.0
    pred = gt * (1.0 + 0.2 * torch.randn(1, H, W, generator=g)) + 0.05
    pred[torch.rand(1, H, W, generator=g) < 0.02] = 75.0
    pred[torch.rand(1, H, W, generator=g) < 0.02] = 0.2
    return pred, gt


def main():
    ref = ref_shim.load_reference("PlaneRecNet_50_config")
    torch.cuda.DoubleTensor = torch.DoubleTensor            # eval.py:193-195 cast through the CUDA tensor type
    # eval.py as a module drags in cv2 / pycocotools / tensorboardX / an old-numpy dependency at import time, none of which the
    # function touches: compile ONLY the reference's `compute_depth_metrics` definition, read from the reference file where it
    # lies (nothing of it is stored in this repository), in a namespace holding torch and the reference's cfg.
    import ast
    src = open(os.path.join(ref_shim.REF_ROOT, "eval.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "compute_depth_metrics"][0]
    ns = {"torch": torch, "cfg": ref["config"].cfg}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), os.path.join(ref_shim.REF_ROOT, "eval.py"), "exec"), ns)
    ev = types.SimpleNamespace(compute_depth_metrics=ns["compute_depth_metrics"])
    cfg = ref["config"].cfg
    fix = {"min_depth": cfg.dataset.min_depth, "max_depth": cfg.dataset.max_depth}
    for seed in (0, 1, 2):
        pred, gt = make_pair(seed)
        r = [float(v) for v in ev.compute_depth_metrics(pred.clone(), gt.clone(), median_scaling=True)]
        o = [float(v) for v in compute_depth_metrics_ref(pred, gt, cfg.dataset.min_depth, cfg.dataset.max_depth)]
        print(seed, ["%.6f" % v for v in r])
        assert np.allclose(r, o, rtol=1e-6, atol=1e-7), (r, o)
        fix["seed%d" % seed] = np.array(r)
    np.savez(os.path.join(HERE, "depth_metrics.npz"), **fix)
    print("written", os.path.join(HERE, "depth_metrics.npz"))


if __name__ == "__main__":
    main()
