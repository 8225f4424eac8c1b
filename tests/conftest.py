import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")


# PRN_TEST_POISON=1: every float tensor the python side allocates uninitialised (torch.empty / empty_like: outputs, workspaces, partial
# buffers) is filled with NaN first, so a kernel that reads memory no launch wrote shows up as NaN in a checked result instead of depending on
# what the caching allocator handed out.  (A debugging mode for the GPU tier; the extra fills make the suite slower.)
if os.environ.get("PRN_TEST_POISON"):
    import torch as _torch
    _empty, _empty_like = _torch.empty, _torch.empty_like

    def _poison(t):
        return t.fill_(float("nan")) if (t.is_cuda and t.dtype.is_floating_point and t.numel()) else t

    _torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
    _torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# The arithmetic the plain GEMMs run in (csrc/prn_gemm_split.hip, include/prn.h: prn_gemm_opts):
#   default   what bench.py times: launches of >= 300 tiles and >= 4 GFLOP on the 16-bit matrix pipe as fp16-piece products -- at the
#             B = 1 / 2 of the parity tests only the 120x160-map products qualify
#   all-f16 / all-bf16   PRN_SPLIT_ALWAYS: EVERY launch the split kernel can take runs on it (stage-3 / stage-4 1x1 layers, every Winograd
#             product, the DCN column gradients, the plane prior) -- so that B = 2 covers what B = 8 times, in both piece formats
#   fp32      fp32 MFMA everywhere (PRN_SPLIT_GEMM=0)
# The model-level parity tests run under all four with the SAME bounds and print their error percentiles per arithmetic.
#   b8-plan   the launch plan of the BENCHMARK's batch applied to the tests' batch of 2: plan mode with the tile and FLOP floors divided by 4
#             (tile counts and FLOPs are linear in the batch), so exactly the launches that run on the 16-bit pipe at B = 8 run on it here
GEMM_ARITHMETICS = {"default": {}, "b8-plan": {"mode": 1, "kind": "f16", "min_tiles": 75, "min_gflop": 1.0},
                    "all-f16": {"mode": 2, "kind": "f16"}, "all-bf16": {"mode": 2, "kind": "bf16"}, "fp32": {"mode": 0}}


@pytest.fixture(params=list(GEMM_ARITHMETICS))
def gemm_arith(request):
    from planerecnet_amd import ops
    kw = dict(GEMM_ARITHMETICS[request.param])
    if os.environ.get("PRN_TEST_WGRAD") is not None and kw.get("mode", 1) != 0:      # (bisecting aid: the weight-gradient kernel on / off under any arithmetic)
        kw["wgrad"] = int(os.environ["PRN_TEST_WGRAD"])
    old = ops.set_split_gemm(**kw)
    ops.SPLIT_STATS.update({"hits": 0, "cuts": 0, "uncached": 0})
    yield request.param
    ops.set_split_gemm(**old)
