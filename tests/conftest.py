import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The CPU side of the tests (the oracle: an fp32 / fp64 PyTorch-CPU restatement, oracle/) runs hundreds of small operators per pass; on the 128-core / 256-thread
# hosts of the GPU boxes a full-width OpenMP team spends its time forking and joining (bench.py cpu_baseline measured the same model ~500x slower at full width than
# at 16 threads) and makes the suite's run time depend on what else the host is doing.  16 threads (PRN_TEST_THREADS overrides) for every test.
try:
    import torch as _t
    _t.set_num_threads(int(os.environ.get("PRN_TEST_THREADS", str(max(1, min(16, os.cpu_count() or 1))))))
except Exception:                                              # noqa: BLE001
    pass


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


# PRN_TEST_GUARD=1: every 4-byte device tensor the python side allocates uninitialised (torch.empty / empty_like: the library's outputs, workspaces, saved
# buffers) sits between two 4 KB guard bands filled with a pattern; after every test the bands of all tensors allocated during it are compared with the
# pattern.  A kernel that writes past the end (or before the start) of its buffer -- harmless or not depending on what the caching allocator placed next to it,
# i.e. on the history of the session -- damages its OWN band and is reported with the allocation's shape and call site, in every run.  (A debugging mode for the
# GPU tier: tensors stay alive until the end of their test, so memory use is the sum of a test's allocations.)
if os.environ.get("PRN_TEST_GUARD"):
    import traceback as _tb
    import torch as _torch
    _g_empty, _g_empty_like = _torch.empty, _torch.empty_like
    _G = 1024                                                  # elements per band (4 KB: a multiple of every alignment the library asks for)
    _PAT = 0x7FA5C3D1                                          # (as a float: a NaN with a payload nothing computes)
    _GUARDED = []                                              # (base tensor, numel, shape, site)

    def _site():
        for fr in reversed(_tb.extract_stack(limit=12)[:-2]):
            if "planerecnet_amd" in fr.filename or "/tests/" in fr.filename:
                return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
        return "?"

    def _guard(t):
        if not (t.is_cuda and t.dtype.itemsize == 4 and t.numel() and t.is_contiguous()):
            return t
        n = t.numel()
        pad = (-n) % 64                                        # the tail band starts 256-byte aligned
        base = _g_empty(n + pad + 2 * _G, device=t.device, dtype=_torch.int32)
        base[:_G].fill_(_PAT)
        base[_G + n:].fill_(_PAT)
        _GUARDED.append((base, n, tuple(t.shape), _site()))
        return base[_G:_G + n].view(t.dtype).view(t.shape)

    _torch.empty = lambda *a, **k: _guard(_g_empty(*a, **k))
    _torch.empty_like = lambda *a, **k: _guard(_g_empty_like(*a, **k))

    @pytest.fixture(autouse=True)
    def _check_guard_bands():
        del _GUARDED[:]
        yield
        if not _GUARDED or not _torch.cuda.is_available():
            return
        _torch.cuda.synchronize()
        damaged = []
        flags = _torch.stack([((b[:_G] != _PAT).any() | (b[_G + n:] != _PAT).any()) for b, n, _, _ in _GUARDED]).cpu()
        for (b, n, shape, site), f in zip(_GUARDED, flags.tolist()):
            if f:
                head = int((b[:_G] != _PAT).sum())
                tail = b[_G + n:]
                bad = (tail != _PAT).nonzero().flatten()
                damaged.append("%s allocated at %s: %d words before the start, %d words past the end (first at +%d, last at +%d)" %
                               (shape, site, head, bad.numel(), int(bad[0]) if bad.numel() else -1, int(bad[-1]) if bad.numel() else -1))
        n_all = len(_GUARDED)
        del _GUARDED[:]
        assert not damaged, "%d of %d guarded allocations were written outside their bounds:\n  %s" % (len(damaged), n_all, "\n  ".join(damaged[:20]))


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
