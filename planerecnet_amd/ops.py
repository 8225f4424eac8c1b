"""Operator layer: torch.autograd.Function wrappers over the HIP C ABI (include/prn.h).

Mirrors the operator surface the reference's modules call (torch.nn.functional conv2d / batch_norm /
group_norm / interpolate / max_pool2d and torchvision.ops.deform_conv2d) with the fusions the MI355X
kernels provide (padding / reflection / nearest-x2 inside the conv gather, residual + ReLU inside BN,
ReLU inside GroupNorm).  PyTorch supplies device memory, the current HIP stream and the autograd
graph; all arithmetic of these ops runs in libprn_hip.so.  Non-device tensors raise: there is no
CPU path in the product.
"""
import ctypes

import os
import threading
import weakref

import torch

from . import _lib, profiling
from ._lib import ConvDesc, GemmOpts, IN_ZERO, IN_REFLECT, IN_UP2_REFLECT, IN_DILATED, IN_UP2_PHASE, IN_EMBED1, EPI_NONE, EPI_RELU, EPI_SIGMOID, lib, check


_SIDE = {}
# Deferred weight gradients.  In the backward pass only the input-gradient chain is on the critical path; the weight
# gradient of a layer is needed by nobody until the optimizer (or the gradient exchange) runs.  With this mode on, every conv
# weight gradient is launched on a side HIP stream that waits for dy but that the main stream never waits for inside backward:
# the ~190 wgrad GEMMs + reductions of a step fill the CUs that the small launches of the main chain leave idle (launch ramp,
# first-load latency, drain: ~25 % of a small kernel's duration).  Measured 83.3 -> 80.8 ms/step.  The gradient is written
# to `weight.grad` directly (on the side stream) and the op returns None for it, so this is OPT-IN for training loops that
# (1) read gradients only through `.grad`, (2) call wgrad_join() after backward() and before anything consumes `.grad`
# (bench.py, train.py).  Post-accumulate-grad hooks (the data-parallel bucketing) keep firing from the autograd engine;
# GradAllReduce makes its exchange stream wait for the side stream as well.  torch.autograd.grad() callers keep the default (off).
# A parameter must receive its gradients either all through deferred ops or all through autograd (a leaf weight that is ALSO
# used through a non-leaf expression would be accumulated by autograd on the main stream while the side stream adds to it).
# (Also tried on the side stream and dropped: the bias gradients -- no change -- and the once-per-step batched weight flip,
# whose per-layer event waits cost 1.5 ms/step more than the 0.5 ms it hides.)
WGRAD_ASYNC = bool(int(os.environ.get("PRN_WGRAD_ASYNC", "0")))


# Workgroups a deferred weight-gradient launch is planned for (see plan_wgrad in csrc/prn_conv.hip): the side stream shares the CUs
# with the main chain, and a launch that fills every CU's register file stalls the main chain's small kernels.  0: the library's
# stand-alone plan.  It travels to the library in every call's prn_gemm_opts (wgrad_wgs); cached workspace sizes are keyed by the options.
WGRAD_WGS_ASYNC = os.environ.get("PRN_WGRAD_WGS_ASYNC", "512")


# ------------------------------------------------------------------------------------------ execution options (prn_gemm_opts)
# The library keeps NO state between calls (include/prn.h): which matrix pipe a plain GEMM takes, the piece format, the thresholds and the
# weight-gradient launch size are this host layer's policy and go into every call -- inside the descriptors or as the `opts` argument.
# PRN_SPLIT_GEMM: 0 = fp32 MFMA everywhere, 1 = where the plan says so (default), 2 = wherever the split kernel applies.
# PRN_SPLIT_KIND: f16 (default: two fp16 pieces, three products, power-of-two scaling) / bf16 (three pieces, six products).
_POLICY = {"split_mode": int(os.environ.get("PRN_SPLIT_GEMM", "1")),
           "kind": _lib.PIECES_BF16 if os.environ.get("PRN_SPLIT_KIND", "f16") == "bf16" else _lib.PIECES_F16,
           "products": int(os.environ.get("PRN_SPLIT16_PRODUCTS", "3")),
           "min_gflop": float(os.environ.get("PRN_SPLIT_MIN_GFLOP", "4.0")),
           # (the eval entry of the train / eval thresholds below: what applies until a model's train() / eval() calls split_gemm_policy)
           "min_tiles": int(os.environ.get("PRN_SPLIT_MIN_TILES", os.environ.get("PRN_SPLIT_MIN_TILES_EVAL", "300"))),
           "wgrad_wgs": int(os.environ.get("PRN_WGRAD_WGS", "0") or 0),
           "wgrad_target": int(os.environ.get("PRN_WGRAD_TARGET", "0") or 0),
           # PRN_WGRAD_SPLIT: the weight gradients of the plain-GEMM layers on the 16-bit pipe (csrc/prn_wgrad16.hip): 0 = fp32 MFMA, 1 = by plan, 2 = wherever it applies
           "wgrad_split": int(os.environ.get("PRN_WGRAD_SPLIT", "1"))}
_OPTS = {}          # policy tuple -> (GemmOpts, byref)
_WGS = [None]       # (kept for cache keys: the current options' key)


def _opts_entry(no_split=False):
    key = (0 if no_split else _POLICY["split_mode"], _POLICY["kind"], _POLICY["products"], _POLICY["min_tiles"], _POLICY["min_gflop"], _POLICY["wgrad_wgs"], _POLICY["wgrad_target"],
           _POLICY["wgrad_split"] if _POLICY["kind"] == _lib.PIECES_F16 else 0)          # (the weight-gradient kernel exists for the fp16 pieces only)
    e = _OPTS.get(key)
    if e is None:
        o = GemmOpts(*key[:4], key[4], key[5], key[6], key[7])
        e = _OPTS[key] = (o, ctypes.byref(o), key)
    _WGS[0] = key
    return e


def opts_ref():
    """byref(prn_gemm_opts) of the current policy: the `opts` argument of the entry points that take one."""
    return _opts_entry()[1]


def opts_key():
    return _opts_entry()[2]


def set_split_gemm(mode=None, kind=None, products=None, min_gflop=None, min_tiles=None, wgrad=None):
    """Change this process's split-GEMM policy (tests, tools, bench.py's fp32-only leg); returns the previous values as a dict that can be
    passed back as keyword arguments.  kind: 'f16' / 'bf16' or the PRN_PIECES_* value."""
    old = {"mode": _POLICY["split_mode"], "kind": _POLICY["kind"], "products": _POLICY["products"], "min_gflop": _POLICY["min_gflop"],
           "min_tiles": (_SPLIT_POLICY["train"], _SPLIT_POLICY["eval"]), "wgrad": _POLICY["wgrad_split"]}
    if mode is not None:
        _POLICY["split_mode"] = int(mode)
        if wgrad is None:
            wgrad = int(mode)                                    # (one switch for tests / tools unless told otherwise)
    if wgrad is not None:
        _POLICY["wgrad_split"] = int(wgrad)
    if kind is not None:
        _POLICY["kind"] = {"f16": _lib.PIECES_F16, "bf16": _lib.PIECES_BF16}.get(kind, kind)
    if products is not None:
        _POLICY["products"] = int(products)
    if min_gflop is not None:
        _POLICY["min_gflop"] = float(min_gflop)
    if min_tiles is not None:                                    # an int (both phases) or (train, eval): the thresholds split_gemm_policy switches between
        tr, ev = (min_tiles if isinstance(min_tiles, (tuple, list)) else (min_tiles, min_tiles))
        _SPLIT_POLICY["train"], _SPLIT_POLICY["eval"] = int(tr), int(ev)
        _POLICY["min_tiles"] = _SPLIT_POLICY[_SPLIT_POLICY.get("mode", "eval")]
        _SPLIT_POLICY["current"] = _POLICY["min_tiles"]
    return old


def split_mode():
    return _POLICY["split_mode"]


def _set_wgs(want):
    _POLICY["wgrad_wgs"] = int(want) if want else 0


def set_wgrad_async(on):
    global WGRAD_ASYNC
    WGRAD_ASYNC = bool(on)
    _set_wgs(WGRAD_WGS_ASYNC if (WGRAD_ASYNC and WGRAD_WGS_ASYNC not in ("", "0")) else None)


def wgrad_streams():
    return list(_SIDE.values())


def wgrad_join():
    """Make the current stream wait for the deferred weight gradients.  Call after backward(), before reading any `.grad`."""
    wgrad_flush()
    if _SIDE:
        main = torch.cuda.current_stream()
        for st in _SIDE.values():
            main.wait_stream(st)
    del _HELD[:]                                            # (inputs of the fast deferred launches: the current stream is now ordered behind their readers)


def _defer(needs_grad, w):
    return WGRAD_ASYNC and needs_grad and w.is_leaf and w.requires_grad and not profiling.active()


# Bias gradients (a per-channel sum of dy: 62 launches, 1 ms per step) have no consumer inside the backward pass either: with the
# weight gradients deferred they go to the side stream too (PRN_BIAS_ASYNC=0: keep them on the main stream).
BIAS_ASYNC = bool(int(os.environ.get("PRN_BIAS_ASYNC", "1")))


def _small_param_grads(leaves, inputs, compute, fast=False):
    """Gradients of a few small parameters (GroupNorm affine pairs, ragged-conv biases) from tensors the backward pass already holds:
    `compute()` -> tuple matching `leaves`.  Deferred to the side stream with the weight gradients when possible (then None)."""
    if BIAS_ASYNC and all(p_ is not None and _defer(True, p_) for p_ in leaves):
        _deferred_wgrad(list(leaves), inputs, compute, fast=fast)
        return None
    return compute()


def sum_rows(part):
    """part [nb, R, N] -> [nb, N] (rows summed in order) in one library launch (include/prn.h: prn_sum_rows)."""
    nb, R, N = part.shape
    out = torch.empty(nb, N, device=part.device, dtype=torch.float32)
    check(lib.prn_sum_rows(_p(part), _p(out), nb, R, N, _stream()), "prn_sum_rows")
    return out


def _bias_grad(needs_grad, bias, dy):
    """d bias = channel_sum(dy): returned for autograd, or queued for the side stream (then None is returned)."""
    if not needs_grad or bias is None:
        return None
    if BIAS_ASYNC and _defer(True, bias):
        _deferred_wgrad(bias, (dy,), lambda: channel_sum(dy), fast=True)
        return None
    return channel_sum(dy)


# Deferred weight gradients are queued per originating stream and moved to the side stream a few at a time: the stream
# switch, the event pair of wait_stream and the record_stream calls cost ~25 us of host time per layer when done one by one
# (128 layers per step on the autograd thread).  Anything that reads `.grad` (wgrad_join, the gradient exchange) flushes first.
WGRAD_BATCH = int(os.environ.get("PRN_WGRAD_BATCH", "4"))       # (3-4: 50.5-50.7 ms/step, 6: 50.7-51.2, 12: 51.3-51.4)
# The gradients are allocated under the side stream and read under the main stream (gradient exchange, optimizer).  Telling the
# allocator so (record_stream) makes it record one event on the MAIN stream per gradient when zero_grad() releases them: ~330
# marker packets in a row, 0.9-1.1 ms in which the GPU does nothing between the optimizer and the next forward pass (bench.py
# PRN_BENCH_GAP=1).  It is not needed: a released block goes back to the side stream's pool, and everything this module (and
# TargetPrefetcher.get) puts on the side stream is preceded by side.wait_stream(main) -- issued after the release, since the
# release happens between steps -- so the next writer of the block is ordered after its last reader.  1 restores the records.
GRAD_RECORD_STREAM = bool(int(os.environ.get("PRN_GRAD_RECORD_STREAM", "0")))
_PENDING = {}       # raw stream handle -> (stream object, [(weight, inputs, compute)])


# Layers of one shape (the 1x1 convolutions of a ResNet stage: 23 blocks in stage 3) are held back until WGRAD_GROUP of them are
# queued and then computed by ONE grouped launch (conv_wgrad_grouped_raw): a single small-map layer needs ~30 pixel splits to
# fill the CUs, a group of 8 needs 4 -- less partial traffic, one reduction instead of 8.  PRN_WGRAD_GROUP=1 switches it off.
WGRAD_GROUP = min(int(os.environ.get("PRN_WGRAD_GROUP", "8")), 16)
WGRAD_GROUP_AGE = int(os.environ.get("PRN_WGRAD_GROUP_AGE", "9"))      # a group that saw no new layer for this many deferred ops goes out as it is
WGRAD_GROUP_PIXELS = 40000       # only maps up to this many pixels per batch (stages 2-4): larger ones need few splits anyway, and the
                                 # last layers of the backward pass must not wait for a group to fill (they would run after it, alone)
_GROUPS = {}        # raw stream handle -> {shape key: [[(weight, (x, dy)), ...], sequence number of the last append]}
_DEFER_SEQ = [0]


def _deferred_wgrad(w, inputs, compute, group=None, hold=None, fast=False):
    """inputs: what `compute` reads -- for a shape group (x, dy) of the layer, tensors or blocks._Ref (pointer + shape).  hold: the tensors that
    own that memory (default: the inputs themselves).  fast: `compute` consists of this library's launches only (no ATen kernels, no temporaries
    besides its results and workspaces taken through _wspace): it is then issued on the side stream's raw handle WITHOUT switching PyTorch's current
    stream -- results come from the calling stream's pool, workspaces from the side stream's persistent scratch, `hold` stays referenced until
    wgrad_join() (see _flush_one)."""
    if hold is None:
        hold = inputs
    handle = _raw_stream(_cur_dev())                        # (the Stream object is built once per stream: torch.cuda.current_stream() costs ~4 us per call)
    e = _PENDING.get(handle)
    if e is None:
        e = _PENDING[handle] = (torch.cuda.current_stream(), [])
    _DEFER_SEQ[0] += 1
    groups = _GROUPS.get(handle)
    if group is not None and WGRAD_GROUP > 1:
        if groups is None:
            groups = _GROUPS[handle] = {}
        pend = groups.setdefault(group, [[], 0])
        pend[0].append((w, inputs, hold, fast))
        pend[1] = _DEFER_SEQ[0]
        if len(pend[0]) >= WGRAD_GROUP:
            _emit_group(e, handle, group)
    else:
        e[1].append((w, hold, compute, fast))
    if groups:
        for key in [k for k, v in groups.items() if _DEFER_SEQ[0] - v[1] >= WGRAD_GROUP_AGE]:       # the backward pass has left that stage
            _emit_group(e, handle, key)
    if len(e[1]) >= WGRAD_BATCH:
        _flush_one(e)


def queue_wgrads(items, hold, need, base, grads):
    """The parameter gradients of a composite node (blocks._BottleneckFn.backward).  items: (index | [indices] into the node's parameter list,
    parameter | [parameters], inputs, compute, shape-group key); need[base + index]: autograd wants that gradient.  Each is deferred to the side
    stream where the training loop allows it (then `.grad` is written there and autograd gets None) or computed now into grads[index]."""
    for idx, w, inputs, compute, gkey in items:
        if isinstance(idx, list):                           # one launch pair, several parameters (a DCN block's offset | modulator convolution)
            if not any(need[base + k] for k in idx):
                continue
            if BIAS_ASYNC and all(need[base + k] for k in idx) and all(_defer(True, p_) for p_ in w):
                _deferred_wgrad(list(w), inputs, compute, None, hold, True)
            else:
                for k, g in zip(idx, compute()):
                    grads[k] = g
        elif need[base + idx]:
            if _defer(True, w) and (BIAS_ASYNC or w.dim() != 1):
                _deferred_wgrad(w, inputs, compute, gkey, hold, True)
            else:
                grads[idx] = compute()


# What the model's per-step refresh vouches for (PlaneRecNet._refresh_dgrad_weights -> vouch_refreshed): id(weight) -> the version its derived layouts
# were last rewritten for, and the addresses of the buffers those batch refreshers rewrite in place (FlippedWeights / WinogradWeights views).
REFRESHED = {}
BATCHED = {}


def vouch_refreshed(weights):
    """Call after FlippedWeights.refresh / WinogradWeights.refresh / split_refresh_all ran for `weights` at their current versions."""
    for w in weights:
        REFRESHED[id(w)] = w._version


def _emit_group(e, handle, key):
    """Move a shape group to the launch queue as ONE item (its compute returns the [G, ...] gradient stack)."""
    pend = _GROUPS[handle].pop(key)[0]
    ws_, ins = [w for w, _, _, _ in pend], [i for _, i, _, _ in pend]
    M, K, stride, pad, mode = key[-5:]
    e[1].append((ws_, tuple(t for _, _, h, _ in pend for t in h), lambda: conv_wgrad_grouped_raw([i[0] for i in ins], [i[1] for i in ins], M, K, stride, pad, mode),
                 all(f for _, _, _, f in pend)))


# Launching a deferred gradient used to mean: switch PyTorch's current stream to the side stream (a context manager, ~10 us), allocate result and
# workspace there, tell the allocator about every input (`record_stream`, ~1.2 us each; a block has three per layer) -- ~50 times per step.  A `fast`
# item needs none of it: it only calls this library, which takes the stream as an argument (_stream() returns the side stream's handle while
# _TLS.side is set); its RESULT is allocated from the calling stream's pool -- the first write on the side stream is ordered after everything the
# calling stream had enqueued when the block was released (side waits for an event recorded on it first), the readers come after wgrad_join();
# its WORKSPACE is the side stream's persistent scratch (launches on one stream are ordered); its INPUTS stay referenced in _HELD until
# wgrad_join() has made the calling stream wait for the side stream -- whoever re-uses their memory afterwards is ordered behind the last reader.
_TLS = threading.local()
_HELD = []
_SIDE_SCRATCH = {}      # raw side-stream handle -> float32 tensor
_SIDE_EVENTS = {}       # raw handle of the originating stream -> reusable event
FAST_WGRAD = os.environ.get("PRN_WGRAD_FAST", "1") == "1"      # 0: every deferred launch under a PyTorch stream context (A/B)


class _Ws:
    __slots__ = ("ptr",)

    def __init__(self, ptr):
        self.ptr = ptr

    def data_ptr(self):
        return self.ptr


def _wspace(nbytes, dev, dtype=torch.float32):
    """Workspace of a weight-gradient launch: the side stream's scratch inside a fast deferred launch, a fresh allocation otherwise."""
    side = getattr(_TLS, "side", None)
    if side is None:
        return torch.empty(max((nbytes + dtype.itemsize - 1) // dtype.itemsize, 1), device=dev, dtype=dtype)
    t = _SIDE_SCRATCH.get(side)
    if t is None or t.numel() * 4 < nbytes:
        with torch.cuda.stream(_TLS.side_obj):
            t = _SIDE_SCRATCH[side] = torch.empty((int(nbytes * 1.5) + (1 << 20)) // 4, device=dev, dtype=torch.float32)
    return _Ws(t.data_ptr())


def _assign_grads(w, dw, main, side_ctx):
    # a list of weights: a shape group (one weight per slice of the gradient stack) or a node with several parameters
    pairs = zip(w, dw if isinstance(dw, (tuple, list)) else dw.unbind(0)) if isinstance(w, list) else [(w, dw)]
    for w_, dw_ in pairs:
        if dw_.shape != w_.shape:
            dw_ = dw_.view_as(w_)
        if GRAD_RECORD_STREAM and side_ctx:
            dw_.record_stream(main)                 # read by the optimizer on the main stream after wgrad_join()
        if w_.grad is None:
            w_.grad = dw_
        elif side_ctx:
            w_.grad.add_(dw_)
        else:                                       # (gradient accumulation: the sum is an ATen kernel and must follow the launch on the side stream)
            with torch.cuda.stream(_TLS.side_obj):
                w_.grad.add_(dw_)
            _HELD.append(dw_)
    # (post-accumulate-grad hooks still fire: the engine runs the parameter's AccumulateGrad node -- a no-op for the
    # undefined gradient this op returns -- and its hooks once all uses of the parameter have been processed)


def _flush_one(e, everything=False):
    main, items = e
    if everything:
        for key in list(_GROUPS.get(main.cuda_stream, {})):     # incomplete shape groups go out as they are
            _emit_group(e, main.cuda_stream, key)
    if not items:
        return
    todo = items[:]
    del items[:]
    first = todo[0][0][0] if isinstance(todo[0][0], list) else todo[0][0]
    side = _side_stream(first.device, main)                 # (keyed by the originating stream)
    ev = _SIDE_EVENTS.get(main.cuda_stream)
    if ev is None:
        ev = _SIDE_EVENTS[main.cuda_stream] = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    slow = [t for t in todo if not (t[3] and FAST_WGRAD)]
    if len(slow) < len(todo):
        _TLS.side, _TLS.side_obj = side.cuda_stream, side
        try:
            for w, hold, compute, fast in todo:
                if fast and FAST_WGRAD:
                    _assign_grads(w, compute(), main, False)
                    _HELD.append(hold)
        finally:
            _TLS.side = None
    if slow:
        with torch.cuda.stream(side), torch.no_grad():
            for w, _, compute, _ in slow:
                _assign_grads(w, compute(), main, True)
        for _, inputs, _, _ in slow:
            for t in inputs:
                t.record_stream(side)                      # (their release puts marker packets on the SIDE stream)


def wgrad_flush():
    """Launch every queued weight gradient (on its side stream).  Called by wgrad_join() and by the gradient exchange
    before it reads a bucket's gradients.  (Planning the last launches of a step -- which run after the backward pass, alone -- for
    full residency again changed nothing: 52.1 vs 52.1 ms; neither did running every other one of them on the compute stream itself,
    which idles ~0.5 ms at the join: 50.9-51.4 vs 51.05-51.1 ms.)"""
    for e in list(_PENDING.values()):
        _flush_one(e, everything=True)


# Off by default.  With the extra streams on, the SAME binary ran either 60.5 or 65.7 ms/step (bimodal between runs, minutes
# apart on one box; kernel-time sums identical, i.e. idle gaps): HIP multiplexes all streams of a process onto 4 hardware
# queues, and whether a branch stream shares a queue with the main chain or with the weight-gradient stream is not under our
# control.  With main + weight-gradient stream only: 61.1-61.4 ms in every run (a 0.6 ms best-case gain against a 5 ms risk).
BRANCH_STREAMS = bool(int(os.environ.get("PRN_BRANCH_STREAMS", "0")))
_BRANCH_POOL = {}


def _tensors(o):
    if isinstance(o, torch.Tensor):
        yield o
    elif isinstance(o, (list, tuple)):
        for v in o:
            yield from _tensors(v)
    elif isinstance(o, dict):
        for v in o.values():
            yield from _tensors(v)


def run_branches(fns):
    """Run independent sub-graphs (callables without arguments) each on its own HIP stream and join them.

    PlaneRecNet's heads are many short chains of small launches (SOLO grids of 12^2 .. 40^2 cells, 1/8 .. 1/32-scale mask
    levels, the decoder's lateral branches): a single in-order stream leaves most CUs idle through every launch ramp and
    tail, while kernels of independent chains can fill each other's gaps.  Measured (40-step runs): 85.8 -> 84.9 ms/step with
    the five instance-head levels forked; forking the mask-head levels and decoder branches as well gives the gain back.  autograd replays each backward node on the stream its forward ran on, so the backward
    pass overlaps in the same way.  Per-op forks (dgrad || wgrad) do NOT pay: two event waits per op cost more than the
    overlap returns (97.6 vs 93.4 ms/step)."""
    if not BRANCH_STREAMS or len(fns) < 2 or not torch.cuda.is_available():
        return [f() for f in fns]
    main = torch.cuda.current_stream()
    pool = _BRANCH_POOL.setdefault(torch.cuda.current_device(), [])
    while len(pool) < len(fns):
        pool.append(torch.cuda.Stream())
    outs = []
    for f, st in zip(fns, pool):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            outs.append(f())
    for o, st in zip(outs, pool):
        main.wait_stream(st)
        for t in _tensors(o):
            t.record_stream(main)
    return outs


def _side_stream(device, main=None):
    key = (device.index, (main if main is not None else torch.cuda.current_stream(device)).cuda_stream)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


_raw_stream = torch._C._cuda_getCurrentRawStream      # raw hipStream_t of the calling thread's current stream (0.3 us vs 10 us)


_cur_dev = torch._C._cuda_getDevice                       # (torch.cuda.current_device() without its lazy-init check: 0.2 against 0.9 us, once per launch)


def _stream():
    side = getattr(_TLS, "side", None)                      # inside a fast deferred launch: the side stream (see _flush_one)
    return ctypes.c_void_p(side if side is not None else _raw_stream(_cur_dev()))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _ptr(t):
    return None if t is None else t.data_ptr()


# Bumped whenever a cache of derived operands (cut images, input-gradient layouts, Winograd-domain weights) gains, drops or re-allocates an entry:
# whoever holds raw pointers into those caches (blocks._State: the parameter table of a Bottleneck call) looks them up again.
OPERAND_EPOCH = [0]


def _dev(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or t.dtype != torch.float32):
            raise RuntimeError("planerecnet_amd ops need fp32 tensors resident on the MI355X (got %s %s); "
                               "there is no CPU fallback" % (t.device, t.dtype))


def _c(t):
    return t if (t is None or t.is_contiguous()) else t.contiguous()


_DESC = {}        # (shape key) -> (ConvDesc, byref, fwd workspace bytes, wgrad workspace bytes): built once per distinct launch shape


def _desc(B, C, H, W, M, K, stride, pad, Ho, Wo, mode=IN_ZERO, dil=1, epi=EPI_NONE, ystride=0, yH=0, yW=0):
    oe = _opts_entry("conv1x1" in SPLIT_SKIP)
    key = (B, C, H, W, M, K, stride, pad, Ho, Wo, mode, dil, epi, ystride, yH, yW, oe[2])
    e = _DESC.get(key)
    if e is None:
        d = ConvDesc(B, C, H, W, M, K, K, stride, pad, Ho, Wo, mode, dil, epi, ystride, yH, yW, 0, oe[0])
        ref = ctypes.byref(d)
        fb = lib.prn_conv2d_fwd_ws_bytes(ref)
        wb = lib.prn_conv2d_wgrad_ws_bytes(ref) if (mode != IN_DILATED and K != 4 and ystride <= 1) else 0
        if fb < 0 or wb < 0:
            raise RuntimeError(lib.prn_last_error().decode())
        d.kind = lib.prn_conv2d_kernel_kind(ref)             # 2 / 3: plain GEMM on the split kernel
        off = ctypes.c_int64(0)
        parts = lib.prn_conv2d_fwd_partials(ref, ctypes.byref(off)) if ystride <= 1 else 0
        e = _DESC[key] = (d, ref, fb, wb, (parts, off.value // 4) if parts > 1 else None)
    return e


# (Round 6: the producer -> BatchNorm hand-overs of DESIGN.md 11.6 -- K-split sums summed by the BatchNorm kernel, Winograd transforms inside the BatchNorm
# launches -- are internal to the block entry points now (include/prn.h: prn_bottleneck_train_fwd / _bwd, planerecnet_amd/blocks.py).  Every operator of this
# module writes its own result: no tensor is ever handed out unwritten, and nothing is keyed by a data pointer between two call sites.)


def _out_hw(H, W, K, stride, pad, mode):
    if mode == IN_REFLECT:
        return H, W
    if mode == IN_UP2_REFLECT:
        return 2 * H, 2 * W
    return (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1


FUSED_SPLIT_SUM = os.environ.get("PRN_CONV_FUSED_REDUCE", "0") == "1"      # see fused_reduce_ok in csrc/prn_conv.hip: neutral on the step
_COUNTERS = {}     # (device index, raw stream) -> zeroed uint32[PRN_TILE_COUNTERS]: the tile-arrival counters of prn_conv2d_fwd_counted
TILE_COUNTERS = 4096


def _counters(dev):
    """One counter buffer per stream: launches on a stream are ordered, and every launch leaves the counters at zero."""
    key = (dev.index, torch._C._cuda_getCurrentRawStream(dev.index))
    c = _COUNTERS.get(key)
    if c is None:
        c = _COUNTERS[key] = torch.zeros(TILE_COUNTERS, device=dev, dtype=torch.int32)
    return c


def split_products():
    """Piece products the split GEMM kernel issues per fp32 multiply-add (csrc/prn_gemm_split.hip): 3 with the fp16 pieces, 6 with the bf16 ones."""
    return float(max(_POLICY["products"], 3)) if _POLICY["kind"] == _lib.PIECES_F16 else 6.0


def _gemm_family(M, K, B, HW, nz, flops):
    """(family, executed FLOPs, reference FLOPs) of a plain GEMM launch for the profiler: on the split kernel the launch executes six
    bf16 products per fp32 multiply-add on the bf16 matrix pipe; `flops` (2*M*K*N) stays the reference operator's work."""
    if gemm_pipe(M, K, B, HW, nz) >= 1:
        return "split_gemm_kernel", split_products() * flops, flops
    return "conv_igemm_kernel", flops, flops



# ------------------------------------------------------------------------------------------ split-GEMM weight images
# The split GEMM kernel (csrc/prn_gemm_split.hip, both piece formats) reads its weight operand as pre-cut "images".  Left alone the library cuts the
# weight inside every launch (one more small kernel in front of each GEMM); for operands that persist -- parameters, the per-step flipped
# dgrad layouts, the Winograd transform-domain weights -- this cache keeps the images (the caller's memory: their pointer travels
# with each call, include/prn.h `w_images`; the library holds no table) and re-cuts ALL of them with one launch per training step (split_refresh_all, called by the model next to
# FlippedWeights / WinogradWeights); at inference they are cut once.  Every launch site checks its operand's entry against the version
# counter of the parameter (or the generation stamp of the derived buffer) first, so a stale image is never read.
# "auto" (default): while training every operand's images are kept and re-cut by ONE batched pass per step (two launches) -- the launches then
# run without their two cutting kernels, i.e. without two dependent-launch gaps each (~6 us of a 45 us launch: 50.5 -> 49.3 ms per step with the
# fp16 pieces; that the first measurement of this cache, with the bf16 pieces, read slower was a comparison across boxes); at inference images
# are kept for the large launches only (SPLIT_CACHE_MIN_TILES).  "1": always, every launch; "eval": only while autograd is off; "0": never.
SPLIT_CACHE = os.environ.get("PRN_SPLIT_CACHE", "auto")
# ... and only for launches of at least this many output tiles: below it the cutting kernels pay for themselves by leaving the images in L2 /
# Infinity Cache in front of the GEMM that streams them (PlaneRecNet_50 B = 8 at 480x640: 580 img/s cutting per launch, 571 with kept images;
# PlaneRecNet_101 B = 4 at 736x960: 246 against 253)
SPLIT_CACHE_MIN_TILES = int(os.environ.get("PRN_SPLIT_CACHE_MIN_TILES", "2500"))
_SPLIT_IMG = {}    # operand data_ptr -> _SplitEntry
_STAMP = {}        # data_ptr of a derived persistent operand (flipped weight view, Winograd U / Ut) -> (generation of its contents, weakref to the stamped tensor)
_STAMP_GEN = [0]   # generations are unique across operands: an address re-stamped by another tensor never repeats a stamp
_SPLIT_ITEMS = [None, None, 0]     # [item table on the device, the entry set it was built for, total blocks]
_PIPE = {}
SPLIT_STATS = {"hits": 0, "cuts": 0, "uncached": 0, "refreshes": 0}     # launches on current images / single re-cuts / temporaries / batched refreshes


_SPLIT_POLICY = {"train": int(os.environ.get("PRN_SPLIT_MIN_TILES", os.environ.get("PRN_SPLIT_MIN_TILES_TRAIN", "300"))),
                 "eval": int(os.environ.get("PRN_SPLIT_MIN_TILES", os.environ.get("PRN_SPLIT_MIN_TILES_EVAL", "300")))}
_SPLIT_POLICY["current"] = _POLICY["min_tiles"]                  # (== the eval entry: direct ops calls from tools / tests honour the env threshold too)


def split_gemm_policy(which):
    """'train' / 'eval': how small a plain GEMM may be and still take the bf16-split kernel (csrc/prn_gemm_split.hip: g_min_tiles has the
    measurements).  A model calls this from train() / eval(); the cached launch plans and workspace sizes are dropped on a change."""
    n = _SPLIT_POLICY[which]
    _SPLIT_POLICY["mode"] = which                                # (autograd functions run with grad mode off: this, not torch.is_grad_enabled(), tells training from inference)
    if _SPLIT_POLICY.get("current") != n:
        _SPLIT_POLICY["current"] = n
        _POLICY["min_tiles"] = n                                 # (descriptor / plan caches are keyed by the options: nothing to drop)


def autotune_split_policy(run_step, which, alternatives=None, steps=10, skip=3, rounds=2, margin=0.015, reduce_max=None):
    """The threshold of split_gemm_policy is a trade against the BOARD's clock management, and boards differ (one box of
    five ran the training step 3.6 % faster with the split kernel on every launch it wins alone, the others 2 % slower).  Times `run_step`
    under the default threshold and under each alternative (A B A B ..., `skip` untimed steps after every switch) and keeps an alternative
    only if its SLOWEST block beats the default's FASTEST by `margin`.  reduce_max: callable(list of floats) -> list, the maximum over ranks
    (so that every rank of a job decides alike).  Returns a dict for the log.
    CAVEAT (measured): the firmware lowers the clock over SECONDS; with ten-step blocks the broad setting looks 0.2 ms slower where
    separate half-minute runs say 1.3 ms -- use steps >= 60 for a decision that holds, or trust the defaults."""
    import time
    base = _SPLIT_POLICY[which]
    alts = [a for a in (alternatives if alternatives is not None else ((300,) if which == "train" else (10 ** 9,))) if a != base]
    if _POLICY["split_mode"] != 1 or "PRN_SPLIT_MIN_TILES" in os.environ or not alts:
        return {"tuned": False, "min_tiles": base}
    times = {n: [] for n in [base] + alts}
    for _ in range(rounds):
        for n in [base] + alts:
            _SPLIT_POLICY[which] = n
            for _ in range(skip):
                run_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                run_step()
            torch.cuda.synchronize()
            times[n].append((time.perf_counter() - t0) / steps)
    if reduce_max is not None:
        flat = reduce_max([t for n in [base] + alts for t in times[n]])
        for i, n in enumerate([base] + alts):
            times[n] = flat[i * rounds:(i + 1) * rounds]
    best = base
    for n in alts:
        if max(times[n]) < (1.0 - margin) * min(times[best]):
            best = n
    _SPLIT_POLICY[which] = best
    return {"tuned": True, "min_tiles": best, "default": base, "ms_per_step": {str(n): [round(1e3 * t, 2) for t in v] for n, v in times.items()}}


def gemm_pipe(M, K, B, HW, nz):
    oe = _opts_entry()
    key = (M, K, B, HW, nz, oe[2])
    v = _PIPE.get(key)
    if v is None:
        v = _PIPE[key] = lib.prn_gemm_pipe(int(M), int(K), int(B), int(HW), int(nz), oe[1])
    return v


class _SplitEntry:
    __slots__ = ("owner", "tensor", "stamp", "images", "dims", "ptr")


def _stamp(t):
    """A derived persistent operand (one the caches below keep and rewrite in place) got new contents."""
    p = t.data_ptr()
    _STAMP_GEN[0] += 1
    _STAMP[p] = (_STAMP_GEN[0], weakref.ref(t))


def _stamp_of(ptr):
    """Generation of the live stamped operand at this address, or None: a stamp whose tensor died says nothing about whatever tensor the
    allocator put there next (a temporary must not inherit it and get its images cached)."""
    s = _STAMP.get(ptr)
    if s is None:
        return None
    if s[1]() is None:
        _STAMP.pop(ptr, None)
        return None
    return s[0]


def _split_drop(ptr, stamp=True):
    if stamp:
        _STAMP.pop(ptr, None)
    if _SPLIT_IMG.pop(ptr, None) is not None:
        _SPLIT_ITEMS[0] = None
        OPERAND_EPOCH[0] += 1


def _split_state(e):
    """Current content stamp of an entry's operand, or None when its owner is gone."""
    if e.owner is not None:
        o = e.owner()
        if o is None or o.data_ptr() != e.ptr:                   # owner gone, or its storage moved (p.data = ..., module.to()): the address may be reused
            return None
        return o._version
    return _stamp_of(e.ptr)


def _prep_items(entries, dev):
    import numpy as np
    items = np.zeros(len(entries), dtype=np.dtype([("src", "u8"), ("dst", "u8"), ("M", "i4"), ("K", "i4"), ("nz", "i4"), ("pad", "i4"), ("zw", "i8"), ("first", "i8")]))
    blocks = rblocks = 0
    for i, e in enumerate(entries):
        M, K, nz, _ = e.dims
        items[i] = (e.ptr, e.images.data_ptr(), M, K, nz, 0, rblocks, blocks)      # ("zw" slot: first block of four rows, weights are dense)
        blocks += (nz * ((M + 127) // 128) * ((K + 31) // 32) * 512 + 255) // 256
        rblocks += (nz * ((M + 127) // 128) * 128 + 3) // 4
    return torch.from_numpy(items.view(np.uint8).reshape(-1).copy()).to(dev), blocks, rblocks


def split_images(t, M, K, nz, cols=None):
    """Call in front of a launch that takes the split kernel with weight operand t [nz, M, K] (dense): returns the device pointer of CURRENT
    images of t (cut now if needed) when t persists (a parameter / a view of one / a stamped derived buffer) -- the `*_images` argument of
    the launch -- or None: the launch then cuts t itself, into its workspace.
    cols = (B, HW) of the activation side: small launches keep cutting per launch (SPLIT_CACHE_MIN_TILES)."""
    training = _SPLIT_POLICY.get("mode", "eval") == "train"
    if SPLIT_CACHE in ("0", False) or (SPLIT_CACHE == "eval" and training):
        return None
    floor = SPLIT_CACHE_MIN_TILES if (SPLIT_CACHE == "eval" or (SPLIT_CACHE == "auto" and not training)) else 0
    if cols is not None and ((M + 127) // 128) * ((cols[1] + 127) // 128) * cols[0] * nz < floor:
        if t.data_ptr() in _SPLIT_IMG:
            _split_drop(t.data_ptr(), stamp=False)               # (the same weight seen earlier with a larger batch)
        return None
    kind = _POLICY["kind"]
    ptr = t.data_ptr()
    e = _SPLIT_IMG.get(ptr)
    if e is not None:
        cur = _split_state(e)
        if cur is not None and cur == e.stamp and e.dims == (M, K, nz, kind):
            SPLIT_STATS["hits"] += 1
            return _p(e.images)                                 # current
        if cur is None or e.dims != (M, K, nz, kind):           # owner gone (address reused), another view of the storage, other piece format
            _split_drop(ptr, stamp=False)
            e = None
    if e is None:
        owner = t if isinstance(t, torch.nn.Parameter) else (t._base if isinstance(t._base, torch.nn.Parameter) else None)
        if owner is None and _stamp_of(ptr) is None:
            SPLIT_STATS["uncached"] += 1
            return None                                         # a temporary: the launch cuts it
        if owner is not None and owner.data_ptr() != ptr:
            return None                                         # a view that does not start at the parameter's first element
        e = _SplitEntry()
        e.owner = weakref.ref(owner) if owner is not None else None
        e.tensor = None if owner is not None else t              # derived buffers are kept alive by the entry (no address reuse)
        e.ptr, e.dims = ptr, (M, K, nz, kind)
        nb = lib.prn_split_images_bytes(M, K, nz)
        e.images = torch.empty(nb, device=t.device, dtype=torch.uint8)
        _SPLIT_IMG[ptr] = e
        _SPLIT_ITEMS[0] = None
        OPERAND_EPOCH[0] += 1
        if owner is not None:
            weakref.finalize(owner, lambda p=ptr, r=e: _split_drop(p) if _SPLIT_IMG.get(p) is r else None)
    check(lib.prn_split_prepare(ctypes.c_void_p(ptr), _p(e.images), M, K, nz, kind, _stream()), "prn_split_prepare")
    e.stamp = _split_state(e)
    SPLIT_STATS["cuts"] += 1
    if os.environ.get("PRN_SPLIT_DEBUG"):
        SPLIT_STATS.setdefault("who", {}).setdefault((M, K, nz, "param" if e.owner is not None else "derived"), []).append(SPLIT_STATS["refreshes"])
    return _p(e.images)


def split_images_ptr(t, M, K, nz, cols=None):
    """split_images as a plain integer address (or None): for parameter tables filled field by field."""
    r = split_images(t, M, K, nz, cols)
    return None if r is None else r.value


def split_refresh_all():
    """Re-cut every cached operand with ONE launch (a model calls this once per training step, after the optimizer changed the weights
    and after the flipped / transform-domain layouts were refreshed)."""
    if SPLIT_CACHE == "eval" or SPLIT_CACHE in ("0", False):     # (inference-only images are validated launch by launch, never refreshed)
        return
    if _SPLIT_POLICY.get("mode", "eval") != "train":
        return
    dead = [p for p, e in _SPLIT_IMG.items() if _split_state(e) is None]
    for p in dead:
        _split_drop(p)
    if not _SPLIT_IMG:
        return
    kind = _POLICY["kind"]
    for e in [e for e in _SPLIT_IMG.values() if e.dims[3] != kind]:          # images of the other piece format: start over at their next launch
        _split_drop(e.ptr, stamp=False)
    entries = list(_SPLIT_IMG.values())
    if not entries:
        return
    if _SPLIT_ITEMS[0] is None or _SPLIT_ITEMS[1] != [id(e) for e in entries]:
        items, blocks, rblocks = _prep_items(entries, entries[0].images.device)
        _SPLIT_ITEMS[0], _SPLIT_ITEMS[1], _SPLIT_ITEMS[2] = items, [id(e) for e in entries], (blocks, rblocks)
    check(lib.prn_split_prepare_batched(_p(_SPLIT_ITEMS[0]), len(entries), _SPLIT_ITEMS[2][0], _SPLIT_ITEMS[2][1], kind, _stream()), "prn_split_prepare_batched")
    for e in entries:
        e.stamp = _split_state(e)
    SPLIT_STATS["refreshes"] += 1


# ------------------------------------------------------------------------------------------ raw launches
def conv_fwd_raw(x, w2d, bias, addend, M, K, stride, pad, Ho, Wo, mode=IN_ZERO, dil=1, epi=EPI_NONE, scatter2=None):
    """scatter2=(yH, yW): store output pixel (oh, ow) at (2*oh, 2*ow) of a zero-filled [B, M, yH, yW] tensor."""
    B, C, H, W = x.shape
    if scatter2 is None:
        y = torch.empty(B, M, Ho, Wo, device=x.device, dtype=torch.float32)
        d_, ref, nbytes, _, _ = _desc(B, C, H, W, M, K, stride, pad, Ho, Wo, mode, dil, epi)
        wimg = split_images(w2d, M, C, 1, (B, Ho * Wo)) if (d_.kind >= 2 and K == 1 and stride == 1) else None      # (tap gather -- 4x4 / stride 2, 1x1 / stride 2 --: images cut per call, tap-major)
    else:
        wimg = None
        y = torch.zeros(B, M, scatter2[0], scatter2[1], device=x.device, dtype=torch.float32)
        _, ref, nbytes, _, _ = _desc(B, C, H, W, M, K, stride, pad, Ho, Wo, mode, dil, epi, 2, scatter2[0], scatter2[1])
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32) if nbytes else None
    cnt = _counters(x.device) if (nbytes and FUSED_SPLIT_SUM) else None      # K-split layers: the sum inside the GEMM launch (opt-in)
    if profiling._enabled:
        # algorithmic FLOPs of the reference convolution this launch evaluates (a dilated-input dgrad is credited with the
        # FLOPs of the strided forward conv it differentiates: a quarter of the MACs the kernel issues)
        kind = lib.prn_conv2d_kernel_kind(ref)
        direct = kind == 1       # the one- / two-channel 3x3 layers run on HBM-bound direct kernels, not on the GEMM
        split = kind >= 2        # plain GEMM on the bf16-split kernel (3: with a K split)
        nb_ = 4.0 * (x.numel() + w2d.numel() + y.numel() + (addend.numel() if addend is not None else 0))
        fl = 2.0 * M * C * K * K * B * Ho * Wo / (dil * dil)
        with profiling.span("conv3x3_direct" if direct else ("split_gemm_kernel" if split else "conv_igemm_kernel"), "hbm" if direct else "mfma",
                            nb_ if direct else (split_products() * fl if split else fl), ref=None if direct else fl, nbytes=nb_,
                            tag=None if direct else ("conv", C, H, W, M, K, stride, mode, dil, B)):
            check(lib.prn_conv2d_fwd_counted(ref, _p(x), _p(w2d), wimg, _p(bias), _p(addend), _p(y), _p(ws), _p(cnt), _stream(), 1), "prn_conv2d_fwd")
        if nbytes and kind != 2:
            with profiling.span("reduce_epilogue_kernel", "hbm", float(nbytes) + 4.0 * y.numel()):
                check(lib.prn_conv2d_fwd_counted(ref, _p(x), _p(w2d), wimg, _p(bias), _p(addend), _p(y), _p(ws), _p(cnt), _stream(), 2), "prn_conv2d_fwd")
    else:
        check(lib.prn_conv2d_fwd_counted(ref, _p(x), _p(w2d), wimg, _p(bias), _p(addend), _p(y), _p(ws), _p(cnt), _stream(), 0), "prn_conv2d_fwd")
    return y


def conv_wgrad_raw(x, dy, M, K, stride, pad, mode):
    B, C, H, W = x.shape
    if mode == IN_UP2_PHASE:                                 # dy: phase-major [4, B, M, H, W]; dw: [4, M, C, 2, 2]
        Ho, Wo = 2 * H, 2 * W
        dw = torch.empty(4, M, C, 2, 2, device=x.device, dtype=torch.float32)
    else:
        Ho, Wo = dy.shape[2:]
        dw = torch.empty(M, C, K, K, device=x.device, dtype=torch.float32)
    _, ref, _, nbytes, _ = _desc(B, C, H, W, M, K, stride, pad, Ho, Wo, mode)
    ws = _wspace(nbytes, x.device)
    if profiling._enabled:
        wkind = lib.prn_conv2d_wgrad_kernel_kind(ref, 1)               # 1: the one- / two-channel 3x3 layers (direct HBM-bound kernel); 2: fp16-piece kernel
        direct = wkind == 1
        fl_w = 2.0 * M * C * K * K * B * Ho * Wo
        with profiling.span("conv3x3_direct" if direct else ("wgrad16_kernel" if wkind == 2 else "conv_wgrad_kernel"), "hbm" if direct else "mfma",
                            4.0 * (x.numel() + dy.numel()) if direct else (split_products() * fl_w if wkind == 2 else fl_w), ref=None if direct else fl_w,
                            nbytes=4.0 * (x.numel() + dy.numel() + dw.numel()),
                            tag=None if direct else ("wgrad", C, H, W, M, K, stride, mode, 1, B)):
            check(lib.prn_conv2d_wgrad_phase(ref, _p(x), _p(dy), _p(dw), _p(ws), _stream(), 1), "prn_conv2d_wgrad")
        if nbytes:
            with profiling.span("reduce_splits_kernel", "hbm", float(nbytes) + 4.0 * dw.numel()):
                check(lib.prn_conv2d_wgrad_phase(ref, _p(x), _p(dy), _p(dw), _p(ws), _stream(), 2), "prn_conv2d_wgrad")
    else:
        check(lib.prn_conv2d_wgrad(ref, _p(x), _p(dy), _p(dw), _p(ws), _stream()), "prn_conv2d_wgrad")
    return dw


WGRAD_GROUP_MAX = 16


def conv_wgrad_grouped_raw(xs, dys, M, K, stride, pad, mode):
    """Weight gradients of len(xs) layers of ONE shape in one launch (include/prn.h: prn_conv2d_wgrad_grouped) -> [G, M, C, K, K]."""
    G = len(xs)
    B, C, H, W = xs[0].shape
    Ho, Wo = dys[0].shape[2:]
    dev = xs[0].device
    _, ref, _, _, _ = _desc(B, C, H, W, M, K, stride, pad, Ho, Wo, mode)
    nbytes = lib.prn_conv2d_wgrad_grouped_ws_bytes(ref, G)
    if nbytes < 0:
        raise RuntimeError(lib.prn_last_error().decode())
    dw = torch.empty(G, M, C, K, K, device=dev, dtype=torch.float32)
    ws = _wspace(nbytes, dev)
    px = (ctypes.c_void_p * G)(*[t.data_ptr() for t in xs])
    pdy = (ctypes.c_void_p * G)(*[t.data_ptr() for t in dys])
    check(lib.prn_conv2d_wgrad_grouped(ref, G, px, pdy, _p(dw), _p(ws), _stream()), "prn_conv2d_wgrad_grouped")
    return dw


_FLIP_USED = set()   # data_ptr of every weight whose dgrad layout was asked for (a model prunes its per-step batch with it)


def flip_transpose(w):
    _FLIP_USED.add(w.data_ptr())
    e = _FLIPPED.get(w.data_ptr())
    if e is not None and tuple(t._version for t in e[0]) == e[1] and e[2].shape[1] == w.shape[0]:
        return e[2]                                         # flipped by FlippedWeights.refresh() since the last weight update
    M, C, KH, KW = w.shape
    wt = torch.empty(C, M, KH, KW, device=w.device, dtype=torch.float32)
    check(lib.prn_weight_flip_transpose(_p(w), _p(wt), M, C, KH, KW, _stream()), "prn_weight_flip_transpose")
    return wt


_FLIPPED = {}       # weight data_ptr -> (tensors whose version counters guard the entry, their versions at flip time, flipped tensor)


class FlippedWeights:
    """All dgrad operand layouts (include/prn.h: prn_weight_flip_transpose) of a model's conv weights, refreshed with ONE
    launch per training step instead of one launch per conv per backward (180 launches for PlaneRecNet_101).
    `weights` is a list of (parameter, (M, C, KH, KW)) -- the shape the dgrad sees (a DCN weight is [M, C*9, 1, 1])."""

    def __init__(self, weights):
        # (parameter, shape) or (parameter, shape, data[, more parameters]): `data` is the tensor actually read (a DCN block's merged
        # [27, C, 3, 3] offset + modulator weight, of which the two parameters are views); the version counters of the parameter AND of
        # the extra ones guard the cached layout (an in-place update of the modulator weight alone must invalidate it too)
        self.weights = [(e[0], tuple(int(v) for v in e[1])) for e in weights]
        self.data = [e[2] if len(e) > 2 else e[0] for e in weights]
        self.guards = [(e[0],) + tuple(e[3] if len(e) > 3 else ()) for e in weights]
        self.ptrs = None

    def _build(self):
        import numpy as np
        dev = self.weights[0][0].device
        total = sum(M * C * KH * KW for _, (M, C, KH, KW) in self.weights)
        self.flat = torch.empty(total, device=dev, dtype=torch.float32)
        items = np.zeros(len(self.weights), dtype=np.dtype([("src", "u8"), ("dst", "u8"), ("M", "i4"), ("C", "i4"), ("KH", "i4"), ("KW", "i4"),
                                                             ("first", "i8")]))
        self.views, first, blocks = [], 0, 0
        for i, ((_, (M, C, KH, KW)), w) in enumerate(zip(self.weights, self.data)):
            assert w.is_contiguous() and w.numel() == M * C * KH * KW and w.dtype == torch.float32
            v = self.flat[first:first + w.numel()].view(C, M, KH, KW)
            items[i] = (w.data_ptr(), v.data_ptr(), M, C, KH, KW, blocks)       # `first` counts 32x32 (M, C) blocks
            self.views.append(v)
            first += w.numel()
            blocks += ((M + 31) // 32) * ((C + 31) // 32)
        self.items = torch.from_numpy(items.view(np.uint8).reshape(-1).copy()).to(dev)
        self.total = blocks
        self.ptrs = [w.data_ptr() for w in self.data]
        OPERAND_EPOCH[0] += 1
        for k in [k for k, o in BATCHED.items() if o == id(self)]:
            del BATCHED[k]
        for v in self.views:                                    # rewritten in place by every refresh(): see blocks.py
            BATCHED[v.data_ptr()] = id(self)

    def refresh(self):
        """Call after the weights changed (once per step, before backward)."""
        if not self.weights:
            return
        if self.ptrs is None or any(w.data_ptr() != p for w, p in zip(self.data, self.ptrs)):
            for p in (self.ptrs or []):
                _FLIPPED.pop(p, None)
            for v in (getattr(self, "views", None) or []):
                _split_drop(v.data_ptr())
            self._build()
        check(lib.prn_weight_flip_transpose_batched(_p(self.items), len(self.weights), self.total, _stream()), "prn_weight_flip_transpose_batched")
        for gd, d, v in zip(self.guards, self.data, self.views):
            _FLIPPED[d.data_ptr()] = (gd, tuple(t._version for t in gd), v)
            _stamp(v)


# ------------------------------------------------------------------------------------------ Winograd F(4x4, 3x3)
WINOGRAD = bool(int(os.environ.get("PRN_WINOGRAD", "1")))       # 0: every 3x3 conv takes the direct implicit-GEMM kernel
WINOGRAD_WGRAD = bool(int(os.environ.get("PRN_WINOGRAD_WGRAD", "1")))   # 0: weight gradients stay on the direct kernel
WINOGRAD_MIN_TILES = int(os.environ.get("PRN_WINOGRAD_MIN_TILES", "128"))   # 4x4 output tiles in the batch below which the direct kernel runs
_WINO = {}          # weight data_ptr -> (weight, version at transform time, U [36,M,C], Ut [36,C,M])


def winograd_ok(B, C, H, W, M, K, stride, pad, mode, epi):
    """Shapes the Winograd path takes (include/prn.h): 3x3 / stride 1 / pad 1, zero or reflect padding, W % 4 == 0, wide
    enough in channels and tiles that 36 GEMMs of [M x C] x [C x tiles] fill the GPU."""
    return (WINOGRAD and K == 3 and stride == 1 and pad == 1 and mode in (IN_ZERO, IN_REFLECT) and epi in (EPI_NONE, EPI_RELU)
            and W % 4 == 0 and H >= 8 and C >= 64 and M >= 64 and B * ((H + 3) // 4) * (W // 4) >= WINOGRAD_MIN_TILES)


def _winograd_items(entries, dev):
    """entries: [(w, M, C, U or None, Ut or None)] -> (device item array, total 32x32 blocks)"""
    import numpy as np
    items = np.zeros(len(entries), dtype=np.dtype([("src", "u8"), ("u", "u8"), ("ut", "u8"), ("M", "i4"), ("C", "i4"), ("first", "i8")]))
    blocks = 0
    for i, (w, M, C, U, Ut) in enumerate(entries):
        items[i] = (w.data_ptr(), U.data_ptr() if U is not None else 0, Ut.data_ptr() if Ut is not None else 0, M, C, blocks)
        blocks += ((M + 31) // 32) * ((C + 31) // 32)
    return torch.from_numpy(items.view(np.uint8).reshape(-1).copy()).to(dev), blocks


def _wino_store(w, U, Ut):
    """Cache entry keyed by the weight's address; holds the weight weakly and disappears with it (the operands are 8x the
    weight's size -- a dead model must not pin them)."""
    ptr = w.data_ptr()
    new = ptr not in _WINO or _WINO[ptr][0]() is not w
    old = _WINO.get(ptr)
    if old is not None and (old[2] is not U or old[3] is not Ut):
        _split_drop(old[2].data_ptr()); _split_drop(old[3].data_ptr())
    if old is None or old[2] is not U or old[3] is not Ut:
        OPERAND_EPOCH[0] += 1
    _WINO[ptr] = (weakref.ref(w), w._version, U, Ut)
    _stamp(U); _stamp(Ut)
    if new:
        weakref.finalize(w, lambda p=ptr, r=_WINO[ptr][0]: _wino_pop(p) if (_WINO.get(p) or (None,))[0] is r else None)


def _wino_pop(ptr):
    e = _WINO.pop(ptr, None)
    if e is not None:
        OPERAND_EPOCH[0] += 1
        _split_drop(e[2].data_ptr()); _split_drop(e[3].data_ptr())


def winograd_weights(w):
    """(U, Ut) of a [M, C, 3, 3] weight: from the per-step batch (WinogradWeights.refresh) when current, else computed here."""
    e = _WINO.get(w.data_ptr())
    if e is not None and e[0]() is w and w._version == e[1]:
        return e[2], e[3]
    M, C = w.shape[:2]
    U = torch.empty(36, M, C, device=w.device, dtype=torch.float32)
    Ut = torch.empty(36, C, M, device=w.device, dtype=torch.float32)
    items, blocks = _winograd_items([(w, M, C, U, Ut)], w.device)
    check(lib.prn_winograd_weights_batched(_p(items), 1, blocks, _stream()), "prn_winograd_weights_batched")
    _wino_store(w, U, Ut)                                    # valid until the weight is modified in place (inference: for good)
    if isinstance(w, torch.nn.Parameter):
        _WINO_SEEN[w.data_ptr()] = weakref.ref(w)
    return U, Ut


_WINO_SEEN = {}     # data_ptr -> weakref(Parameter): weights the Winograd path was asked for (a model batches these per step)


def winograd_seen():
    return [w for w in (r() for r in _WINO_SEEN.values()) if w is not None]


class WinogradWeights:
    """Transform-domain operands (include/prn.h: prn_winograd_weights_batched) of a model's 3x3 weights, forward and
    input-gradient form, refreshed with ONE launch per training step."""

    def __init__(self, weights):
        self.weights = [w for w in weights]
        self.ptrs = None

    def _build(self):
        dev = self.weights[0].device
        total = sum(w.numel() * 4 for w in self.weights)                      # 36 / 9 floats per tap set
        self.flat_u = torch.empty(total, device=dev, dtype=torch.float32)
        self.flat_ut = torch.empty(total, device=dev, dtype=torch.float32)
        self.views, entries, first = [], [], 0
        for w in self.weights:
            M, C = w.shape[:2]
            assert w.is_contiguous() and tuple(w.shape[2:]) == (3, 3) and w.dtype == torch.float32
            n = 36 * M * C
            U, Ut = self.flat_u[first:first + n].view(36, M, C), self.flat_ut[first:first + n].view(36, C, M)
            self.views.append((U, Ut))
            entries.append((w, M, C, U, Ut))
            first += n
        self.items, self.total = _winograd_items(entries, dev)
        self.ptrs = [w.data_ptr() for w in self.weights]
        OPERAND_EPOCH[0] += 1
        for k in [k for k, o in BATCHED.items() if o == id(self)]:
            del BATCHED[k]
        for U, Ut in self.views:
            BATCHED[U.data_ptr()] = BATCHED[Ut.data_ptr()] = id(self)

    def refresh(self):
        """Call after the weights changed (once per step, before forward)."""
        if not self.weights or not WINOGRAD:
            return
        if self.ptrs is None or any(w.data_ptr() != p for w, p in zip(self.weights, self.ptrs)):
            for p in (self.ptrs or []):
                _wino_pop(p)
            self._build()
        check(lib.prn_winograd_weights_batched(_p(self.items), len(self.weights), self.total, _stream()), "prn_winograd_weights_batched")
        for w, (U, Ut) in zip(self.weights, self.views):
            _wino_store(w, U, Ut)


_WINO_WG_WS = {}
# the 36 transform-domain products of a layer take the split kernel only from this many 128 x 128 output tiles (0: whenever the plan says so)
WINOGRAD_SPLIT_MIN_TILES = int(os.environ.get("PRN_SPLIT_WINO_MIN_TILES", "0"))
SPLIT_SKIP = set(filter(None, os.environ.get("PRN_SPLIT_SKIP", "").split(",")))      # bisecting aid: launch families kept on the fp32 kernel: conv1x1, colgrad, prior, wino
WINOGRAD_KEEP_V = int(os.environ.get("PRN_WINOGRAD_KEEP_V", str(128 << 20)))    # keep B^T x B for the weight gradient up to this many bytes per layer


def conv3x3_winograd_raw(x, U, bias, addend, M, mode=IN_ZERO, epi=EPI_NONE, keep=None):
    """3x3 / stride 1 / pad 1 convolution of x [B,C,H,W] with transform-domain weights U [36,M,C].
    keep: a list that receives the workspace (whose head is V = B^T x B) for conv3x3_winograd_wgrad_raw(..., V=...).
    mode IN_EMBED1: x is the block at (1, 1) of a virtual zero tensor [B,C,H+2,W+4]; the result has that size (include/prn.h)."""
    B, C, H, W = x.shape
    if mode == IN_EMBED1:
        H, W = H + 2, W + 4
    P = lib.prn_winograd_tiles(B, H, W)
    y = torch.empty(B, M, H, W, device=x.device, dtype=torch.float32)
    split_ok = "wino" not in SPLIT_SKIP and gemm_pipe(M, C, 1, P, 36) >= 1 and ((M + 127) // 128) * ((P + 127) // 128) * 36 >= WINOGRAD_SPLIT_MIN_TILES
    uimg = split_images(U, M, C, 36, (1, P)) if split_ok else None
    oref = opts_ref() if split_ok else None
    wkey = ("wino-fwd", B, C, H, W, M, opts_key() if split_ok else None)
    nb_ws = _WINO_WG_WS.get(wkey)
    if nb_ws is None:
        nb_ws = _WINO_WG_WS[wkey] = lib.prn_conv3x3_winograd_ws_bytes(B, C, H, W, M, oref)
        if nb_ws < 0:
            raise RuntimeError(lib.prn_last_error().decode())
    ws = torch.empty(nb_ws // 4, device=x.device, dtype=torch.float32)
    if profiling._enabled:
        V, Yt = ws[:36 * C * P], ws[36 * C * P:36 * (C + M) * P]
        gws = ws[(36 * (C + M) * P + 63) // 64 * 64:]
        with profiling.span("winograd_input_kernel", "hbm", 4.0 * x.numel() + 4.0 * V.numel(), 0.0):
            check(lib.prn_winograd_input(_p(x), _p(V), B, C, H, W, mode, _stream()), "prn_winograd_input")
        fam, ex, _ = _gemm_family(M, C, 1, P, 36, 2.0 * 36 * M * C * P) if split_ok else ("conv_igemm_kernel", 2.0 * 36 * M * C * P, None)
        with profiling.span(fam, "mfma", ex, 2.0 * 9 * M * C * B * H * W, nbytes=4.0 * 36 * (C * P + M * C + M * P),
                            tag=("wino-products", C, H, W, M, 3, 1, mode, 1, B)):
            check(lib.prn_gemm_batched(M, C, P, 36, _p(U), uimg, _p(V), _p(Yt), _p(gws) if gws.numel() else None, oref, _stream()), "prn_gemm_batched")
        with profiling.span("winograd_output_kernel", "hbm", 4.0 * Yt.numel() + 4.0 * y.numel() * (2 if addend is not None else 1), 0.0):
            check(lib.prn_winograd_output(_p(Yt), _p(bias), _p(addend), _p(y), B, M, H, W, epi, _stream()), "prn_winograd_output")
    else:
        check(lib.prn_conv3x3_winograd(_p(x), _p(U), uimg, _p(bias), _p(addend), _p(y), _p(ws), B, C, H, W, M, mode, epi, oref, _stream()), "prn_conv3x3_winograd")
    if keep is not None and mode == IN_ZERO and 4 * 36 * C * P <= WINOGRAD_KEEP_V:
        keep.append(ws)
    return y


def conv3x3_winograd_wgrad_raw(x, dy, M, mode=IN_ZERO, V=None):
    """Weight gradient [M,C,3,3] of a 3x3 / stride 1 / pad 1 convolution on the Winograd path.  V: the forward call's kept
    workspace (see conv3x3_winograd_raw) -- the input transform is then not recomputed."""
    B, C, H, W = x.shape
    oref = opts_ref()
    key = (B, C, H, W, M, opts_key())
    nbytes = _WINO_WG_WS.get(key)
    if nbytes is None:
        nbytes = _WINO_WG_WS[key] = lib.prn_winograd_wgrad_ws_bytes(B, C, H, W, M, oref)
    ws = _wspace(nbytes, x.device)
    dw = torch.empty(M, C, 3, 3, device=x.device, dtype=torch.float32)
    if V is not None and not profiling._enabled:
        check(lib.prn_conv3x3_winograd_wgrad_v(_p(V), _p(dy), _p(dw), _p(ws), B, C, H, W, M, oref, _stream()), "prn_conv3x3_winograd_wgrad_v")
        return dw
    args = (_p(x), _p(dy), _p(dw), _p(ws), B, C, H, W, M, mode, oref, _stream())
    if profiling._enabled:
        P = lib.prn_winograd_tiles(B, H, W)
        with profiling.span("winograd_wgrad_transforms", "hbm", 4.0 * (x.numel() + dy.numel()) + 4.0 * 36 * (C + M) * P, 0.0):
            check(lib.prn_conv3x3_winograd_wgrad(*args, 1), "prn_conv3x3_winograd_wgrad")
        w16 = lib.prn_gemm_batched_nt_kind(M, C, P, 36, oref) == 2
        with profiling.span("wgrad16_kernel" if w16 else "conv_wgrad_kernel", "mfma", (split_products() if w16 else 1.0) * 2.0 * 36 * M * C * P, 2.0 * 9 * M * C * B * H * W,
                            tag=("wino-wgrad-products", C, H, W, M, 3, 1, mode, 1, B)):
            check(lib.prn_conv3x3_winograd_wgrad(*args, 2), "prn_conv3x3_winograd_wgrad")
        with profiling.span("winograd_dw_kernel", "hbm", float(nbytes) - 4.0 * 36 * (C + M) * P + 4.0 * dw.numel(), 0.0):
            check(lib.prn_conv3x3_winograd_wgrad(*args, 3), "prn_conv3x3_winograd_wgrad")
    else:
        check(lib.prn_conv3x3_winograd_wgrad(*args, 0), "prn_conv3x3_winograd_wgrad")
    return dw


def channel_sum(g):
    B, C, H, W = g.shape
    out = torch.empty(C, device=g.device, dtype=torch.float32)
    ws = _wspace(8 * C * _lib.BN_SPLITS, g.device, torch.float64)
    check(lib.prn_channel_sum(_p(g), _p(out), _p(ws), B, C, H * W, _stream()), "prn_channel_sum")
    return out


def conv_dgrad_raw(dy, w, x_shape, stride, pad, mode, addend=None):
    """Gradient w.r.t. the conv input: the same implicit-GEMM kernel run over dy with flipped/transposed weights.
    `addend` (another gradient of the same input, e.g. the residual branch) is summed in the kernel epilogue."""
    B, C, H, W = x_shape
    M, _, K, _ = w.shape
    if mode == IN_ZERO and winograd_ok(B, M, H, W, C, K, stride, pad, IN_ZERO, EPI_NONE) and tuple(dy.shape[2:]) == (H, W):
        return conv3x3_winograd_raw(dy, winograd_weights(w)[1], None, addend, C)
    wt = flip_transpose(w)                                  # [C, M, K, K]
    if mode == IN_REFLECT and W % 4 == 0 and tuple(dy.shape[2:]) == (H, W) and winograd_ok(B, M, H + 2, W + 4, C, K, 1, 1, IN_ZERO, EPI_NONE):
        # gradient of the reflect-padded tensor = FULL correlation of dy, on the Winograd path (dy embedded at (1, 1) of a
        # zero tensor by the input transform's index map), then folded onto the unpadded tensor
        dp = conv3x3_winograd_raw(dy, winograd_weights(w)[1], None, None, C, IN_EMBED1)          # [B, C, H+2, W+4]
        dx = torch.empty(B, C, H, W, device=dy.device, dtype=torch.float32)
        check(lib.prn_pad_fold_pitched(_p(dp), _p(dx), B, C, H, W, W + 4, _stream()), "prn_pad_fold_pitched")
        return dx if addend is None else dx + addend
    if mode in (IN_REFLECT, IN_UP2_REFLECT):
        Hv, Wv = (2 * H, 2 * W) if mode == IN_UP2_REFLECT else (H, W)
        dp = conv_fwd_raw(dy, wt, None, None, C, K, 1, 2, Hv + 2, Wv + 2)       # grad of the virtual padded tensor
        dx = torch.empty(B, C, H, W, device=dy.device, dtype=torch.float32)
        check(lib.prn_pad_fold(_p(dp), _p(dx), B, C, H, W, 1 if mode == IN_UP2_REFLECT else 0, _stream()), "prn_pad_fold")
        return dx if addend is None else dx + addend
    if stride == 1:
        return conv_fwd_raw(dy, wt, None, addend, C, K, 1, K - 1 - pad, H, W)
    if stride != 2:
        raise RuntimeError("conv dgrad: stride %d not implemented" % stride)
    if K == 1 and pad == 0 and addend is None:
        # only the even positions of dx are non-zero: run the GEMM over dy's own pixel grid and scatter (4x fewer MACs
        # than gathering through the zero-dilated view)
        return conv_fwd_raw(dy, wt, None, None, C, 1, 1, 0, dy.shape[2], dy.shape[3], scatter2=(H, W))
    if K == 1 and pad == 0:
        # ... and with another gradient of the same input: scattered, then summed (the block entry points add INTO a gradient they provably own instead:
        # blocks._BottleneckFn.backward; a plain operator does not write into a tensor autograd handed it)
        return conv_fwd_raw(dy, wt, None, None, C, 1, 1, 0, dy.shape[2], dy.shape[3], scatter2=(H, W)) + addend
    return conv_fwd_raw(dy, wt, None, addend, C, K, 1, K - 1 - pad, H, W, IN_DILATED, 2)


# ------------------------------------------------------------------------------------------ conv2d
class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, addend, stride, pad, mode, epi, fork=False):
        _dev(x, w, bias, addend)
        x0, bias_param = x, bias
        x, w, bias, addend = _c(x), _c(w), _c(bias), _c(addend)
        M, C, K, _ = w.shape
        assert x.shape[1] == C, (x.shape, w.shape)
        Ho, Wo = _out_hw(x.shape[2], x.shape[3], K, stride, pad, mode)
        ctx.wino_v = None
        if winograd_ok(x.shape[0], C, x.shape[2], x.shape[3], M, K, stride, pad, mode, epi):
            keep = [] if (WINOGRAD_WGRAD and ctx.needs_input_grad[1]) else None
            y = conv3x3_winograd_raw(x, winograd_weights(w)[0], bias, addend, M, mode, epi, keep)
            if keep:
                ctx.wino_v = keep[0]
        else:
            y = conv_fwd_raw(x, w, bias, addend, M, K, stride, pad, Ho, Wo, mode, 1, epi)
        ctx.save_for_backward(x, w, y if epi != EPI_NONE else None)
        ctx.cfg = (stride, pad, mode, epi, bias is not None, addend is not None)
        ctx.bias = bias_param
        ctx.fork = fork
        if fork:
            # second output = the input itself: whatever else consumes x (the residual add) takes it from here, so both
            # gradients of x reach THIS node and are summed in the dgrad epilogue instead of by a separate add kernel
            ctx.set_materialize_grads(False)
            return y, x0
        return y

    @staticmethod
    def backward(ctx, dy, dfork=None):
        x, w, y = ctx.saved_tensors
        stride, pad, mode, epi, has_bias, has_add = ctx.cfg
        if dy is None:                                      # only the forked identity was used
            return (dfork,) + (None,) * 8
        dy = _c(dy)
        dfork = _c(dfork)
        if epi == EPI_RELU:
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)      # dy * (y > 0) in one pass
        elif epi == EPI_SIGMOID:
            dy = dy * y * (1 - y)
        M, C, K, _ = w.shape
        if WINOGRAD_WGRAD and winograd_ok(x.shape[0], C, x.shape[2], x.shape[3], M, K, stride, pad, mode, EPI_NONE):
            V, ctx.wino_v = ctx.wino_v, None                # (attributes of ctx are not released by autograd after backward)

            def wgrad():
                return conv3x3_winograd_wgrad_raw(x, dy, M, mode, V)
        else:
            V = None

            def wgrad():
                return conv_wgrad_raw(x, dy, M, K, stride, pad, mode)
        if _defer(ctx.needs_input_grad[1], w):
            # every buffer the deferred launch reads must be listed: it is released here, on the main stream, possibly before
            # the side stream has run (a kept Winograd operand is one of them)
            gkey = None
            if V is None and K in (1, 3, 7) and mode in (IN_ZERO, IN_REFLECT) and M > 2 and dy.shape[0] * dy.shape[2] * dy.shape[3] <= WGRAD_GROUP_PIXELS:
                gkey = ("conv", tuple(x.shape), tuple(dy.shape[2:]), M, K, stride, pad, mode)
            _deferred_wgrad(w, (x, dy) if V is None else (x, dy, V), wgrad, gkey, fast=True)
            dx = conv_dgrad_raw(dy, w, x.shape, stride, pad, mode, dfork) if ctx.needs_input_grad[0] else None
            dfork = None
            dw = None
        else:
            dx = conv_dgrad_raw(dy, w, x.shape, stride, pad, mode, dfork) if ctx.needs_input_grad[0] else None
            dfork = None
            dw = wgrad() if ctx.needs_input_grad[1] else None
        if dfork is not None and dx is not None:
            dx = dx + dfork
        db = _bias_grad(has_bias and ctx.needs_input_grad[2], ctx.bias, dy)
        da = dy if (has_add and ctx.needs_input_grad[3]) else None
        return dx, dw, db, da, None, None, None, None, None


class _ConvUp2(torch.autograd.Function):
    """Upsample(x2, nearest) -> ReflectionPad2d(1) -> Conv3x3 in its sub-pixel form (include/prn.h: PRN_IN_UP2_PHASE):
    four 2x2 convolutions of the source map instead of one 3x3 convolution of the upsampled one (2.25x fewer MACs in
    forward, input gradient and weight gradient alike)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        _dev(x, w, bias)
        bias_param = bias
        x, w, bias = _c(x), _c(w), _c(bias)
        B, C, H, W = x.shape
        M = w.shape[0]
        assert w.shape[1:] == (C, 3, 3), (x.shape, w.shape)
        wp = torch.empty(4, M, C, 2, 2, device=x.device, dtype=torch.float32)
        check(lib.prn_up2_phase_weights(_p(w), _p(wp), M, C, _stream()), "prn_up2_phase_weights")
        y = conv_fwd_raw(x, wp, bias, None, M, 2, 1, 0, 2 * H, 2 * W, IN_UP2_PHASE)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.bias = bias_param
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        B, C, H, W = x.shape
        M = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            kd = torch.empty(C, M, 4, 4, device=x.device, dtype=torch.float32)
            check(lib.prn_up2_dgrad_weights(_p(w), _p(kd), M, C, _stream()), "prn_up2_dgrad_weights")
            dp = conv_fwd_raw(dy, kd, None, None, C, 4, 2, 3, H + 2, W + 2)           # gradient of the replicate-padded source
            dx = torch.empty_like(x)
            check(lib.prn_replicate_fold(_p(dp), _p(dx), B, C, H, W, _stream()), "prn_replicate_fold")
        def wgrad():
            dyp = torch.empty(4, B, M, H, W, device=x.device, dtype=torch.float32)
            check(lib.prn_space_to_depth2(_p(dy), _p(dyp), B, M, H, W, _stream()), "prn_space_to_depth2")
            dwp = conv_wgrad_raw(x, dyp, M, 2, 1, 0, IN_UP2_PHASE)
            dwo = torch.empty_like(w)
            check(lib.prn_up2_wgrad_combine(_p(dwp), _p(dwo), M, C, _stream()), "prn_up2_wgrad_combine")
            if getattr(_TLS, "side", None) is not None:          # a fast deferred launch: its temporaries live until wgrad_join() like its inputs (_flush_one)
                _HELD.append((dyp, dwp))
            return dwo
        if _defer(ctx.needs_input_grad[1], w):
            _deferred_wgrad(w, (x, dy), wgrad, fast=True)
        elif ctx.needs_input_grad[1]:
            dw = wgrad()
        db = _bias_grad(ctx.has_bias and ctx.needs_input_grad[2], ctx.bias, dy)
        return dx, dw, db


UP2_SUBPIXEL = bool(int(os.environ.get("PRN_UP2_SUBPIXEL", "1")))      # 0: the PRN_IN_UP2_REFLECT gather at output resolution


def conv2d(x, w, bias=None, stride=1, pad=0, in_mode=IN_ZERO, epilogue=EPI_NONE, addend=None):
    """F.conv2d replacement (reference: every nn.Conv2d call; see include/prn.h for the call-site list)."""
    if in_mode == IN_UP2_REFLECT and UP2_SUBPIXEL and epilogue == EPI_NONE and addend is None and x.shape[2] > 1 and x.shape[3] > 1:
        return _ConvUp2.apply(x, w, bias)
    return _Conv2d.apply(x, w, bias, addend, stride, pad, in_mode, epilogue, False)


_UP2_PHASE = {}     # weight data_ptr -> (weakref(weight), version, phase weights): inference only


def conv_up2_inference(x, w, bias=None, relu=False):
    """Upsample(x2, nearest) -> ReflectionPad2d(1) -> Conv3x3 (+ bias, + ReLU) in its sub-pixel form WITHOUT autograd: the
    eval-mode decoder blocks with BatchNorm folded into (w, bias) -- one launch per block instead of conv + rsqrt / cat / BatchNorm
    kernels; the four 2x2 phase kernels are rebuilt only when the weight changes."""
    _dev(x, w, bias)
    x, w, bias = _c(x), _c(w), _c(bias)
    B, C, H, W = x.shape
    M = w.shape[0]
    assert w.shape[1:] == (C, 3, 3), (x.shape, w.shape)
    if not (UP2_SUBPIXEL and H > 1 and W > 1):
        return conv2d(x, w, bias, 1, 1, IN_UP2_REFLECT, EPI_RELU if relu else EPI_NONE)
    e = _UP2_PHASE.get(w.data_ptr())
    if e is None or e[0]() is not w or e[1] != w._version:
        wp = torch.empty(4, M, C, 2, 2, device=x.device, dtype=torch.float32)
        check(lib.prn_up2_phase_weights(_p(w), _p(wp), M, C, _stream()), "prn_up2_phase_weights")
        ptr = w.data_ptr()
        e = _UP2_PHASE[ptr] = (weakref.ref(w), w._version, wp)
        weakref.finalize(w, lambda p=ptr, r=e[0]: _UP2_PHASE.pop(p, None) if (_UP2_PHASE.get(p) or (None,))[0] is r else None)
    return conv_fwd_raw(x, e[2], bias, None, M, 2, 1, 0, 2 * H, 2 * W, IN_UP2_PHASE, 1, EPI_RELU if relu else EPI_NONE)


def conv2d_fork(x, w, bias=None, stride=1, pad=0, in_mode=IN_ZERO, epilogue=EPI_NONE, addend=None):
    """conv2d that also hands its input back: `y, x_id = conv2d_fork(x, w)`.  Use x_id wherever else x is consumed
    (the identity branch of a residual block): the two gradients of x are then summed inside the input-gradient GEMM's
    epilogue rather than by autograd's separate accumulation kernel (one full-tensor read-read-write pass per block)."""
    return _Conv2d.apply(x, w, bias, addend, stride, pad, in_mode, epilogue, True)


# ------------------------------------------------------------------------------------------ DCNv2
# One operator (include/prn.h: prn_dcnv2_*): the bilinear sampler is the operand loader of the MFMA contraction in the
# forward pass and in the weight gradient; no [B, 9*Cin, Ho, Wo] column tensor exists.  The column GRADIENT W^T dy is the
# one transient intermediate of the backward pass (it feeds both d-input and d-offset / d-mask).
_DCN = {}          # geometry key -> (DcnDesc, byref, table bytes, fwd ws bytes, wgrad ws bytes, data-gradient ws bytes)


def _dcn_desc(B, C, H, W, M, stride, pad, raw, max_offset, epi=EPI_NONE):
    oe = _opts_entry("colgrad" in SPLIT_SKIP)
    key = (B, C, H, W, M, stride, pad, raw, float(max_offset), epi, oe[2])
    e = _DCN.get(key)
    if e is None:
        Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
        d = _lib.DcnDesc(B, C, H, W, M, stride, pad, Ho, Wo, int(raw), float(max_offset), epi, oe[0])
        ref = ctypes.byref(d)
        sizes = [lib.prn_dcnv2_table_bytes(ref), lib.prn_dcnv2_fwd_ws_bytes(ref), lib.prn_dcnv2_bwd_weight_ws_bytes(ref), lib.prn_dcnv2_bwd_ws_bytes(ref)]
        if min(sizes) < 0:
            raise RuntimeError(lib.prn_last_error().decode())
        e = _DCN[key] = (d, ref, sizes[0], sizes[1], sizes[2], sizes[3])
    return e


def _f32(nbytes, dev):
    return torch.empty(max(nbytes // 4, 1), device=dev, dtype=torch.float32)


def dcn_table(x_shape, M, offset, mask, stride, pad, raw, max_offset):
    """(offset, mask) -> gather table of the fused operator (include/prn.h: prn_dcnv2_table)."""
    B, C, H, W = x_shape
    _, ref, tb, _, _, _ = _dcn_desc(B, C, H, W, M, stride, pad, raw, max_offset)
    table = _f32(tb, offset.device)
    check(lib.prn_dcnv2_table(ref, _p(offset), _p(mask), _p(table), _stream()), "prn_dcnv2_table")
    return table


def dcn_fwd_raw(x, table, w, bias, stride, pad, raw, max_offset, epi=EPI_NONE):
    B, C, H, W = x.shape
    M = w.shape[0]
    d, ref, _, fb, _, _ = _dcn_desc(B, C, H, W, M, stride, pad, raw, max_offset, epi)
    y = torch.empty(B, M, d.Ho, d.Wo, device=x.device, dtype=torch.float32)
    ws = _f32(fb, x.device) if fb else None
    if profiling._enabled:
        with profiling.span("dcnv2_fwd_kernel", "mfma", 2.0 * M * C * 9 * B * d.Ho * d.Wo,
                            nbytes=4.0 * (x.numel() + w.numel() + y.numel()) + 32.0 * 9 * B * d.Ho * d.Wo):
            check(lib.prn_dcnv2_fwd_phase(ref, _p(x), _p(table), _p(w), _p(bias), _p(y), _p(ws), _stream(), 1), "prn_dcnv2_fwd")
        if fb:
            with profiling.span("reduce_epilogue_kernel", "hbm", float(fb) + 4.0 * y.numel()):
                check(lib.prn_dcnv2_fwd_phase(ref, _p(x), _p(table), _p(w), _p(bias), _p(y), _p(ws), _stream(), 2), "prn_dcnv2_fwd")
    else:
        check(lib.prn_dcnv2_fwd(ref, _p(x), _p(table), _p(w), _p(bias), _p(y), _p(ws), _stream()), "prn_dcnv2_fwd")
    return y


def dcn_wgrad_raw(x, table, dy, M, stride, pad, raw, max_offset):
    B, C, H, W = x.shape
    _, ref, _, _, wb, _ = _dcn_desc(B, C, H, W, M, stride, pad, raw, max_offset)
    dw = torch.empty(M, C, 3, 3, device=x.device, dtype=torch.float32)
    ws = _wspace(wb, x.device) if wb else None
    if profiling._enabled:
        with profiling.span("dcnv2_wgrad_kernel", "mfma", 2.0 * M * C * 9 * dy.shape[0] * dy.shape[2] * dy.shape[3],
                            nbytes=4.0 * (x.numel() + dy.numel() + dw.numel()) + 32.0 * 9 * dy.shape[0] * dy.shape[2] * dy.shape[3]):
            check(lib.prn_dcnv2_bwd_weight_phase(ref, _p(x), _p(table), _p(dy), _p(dw), _p(ws), _stream(), 1), "prn_dcnv2_bwd_weight")
        if wb:
            with profiling.span("reduce_splits_kernel", "hbm", float(wb) + 4.0 * dw.numel()):
                check(lib.prn_dcnv2_bwd_weight_phase(ref, _p(x), _p(table), _p(dy), _p(dw), _p(ws), _stream(), 2), "prn_dcnv2_bwd_weight")
    else:
        check(lib.prn_dcnv2_bwd_weight(ref, _p(x), _p(table), _p(dy), _p(dw), _p(ws), _stream()), "prn_dcnv2_bwd_weight")
    return dw


def dcn_data_grads_raw(x, offset, mask, w, dy, stride, pad, raw, max_offset, need_x=True, need_om=True):
    """-> (dx, d_offset, d_mask): the column gradient W^T dy goes to the head of one workspace (1x1 MFMA GEMM), d-input is
    gathered from it through the per-call CSR inversion of the sampling pattern, d-offset / d-mask are reduced from it."""
    B, C, H, W = x.shape
    M = w.shape[0]
    d, ref, _, _, _, db = _dcn_desc(B, C, H, W, M, stride, pad, raw, max_offset)
    wt = flip_transpose(w.view(M, C * 9, 1, 1))                   # [9C, M, 1, 1]
    ws = _f32(db, x.device)
    # the column-gradient GEMM's weight operand: current images, or None (the launch cuts wt itself)
    wimg = split_images(wt, 9 * C, M, 1, (B, d.Ho * d.Wo)) if gemm_pipe(9 * C, M, B, d.Ho * d.Wo, 1) >= 1 else None
    dx = torch.empty_like(x) if need_x else None
    ncols = 4.0 * B * C * 9 * d.Ho * d.Wo
    if profiling._enabled:
        # every launch bracketed on its own: the column-gradient GEMM, its K-split sum, the CSR gather of dx (phases 1 / 2 / 3)
        fam, ex, fl = _gemm_family(9 * C, M, B, d.Ho * d.Wo, 1, 2.0 * M * C * 9 * B * d.Ho * d.Wo)
        with profiling.span(fam, "mfma", ex, fl, nbytes=4.0 * (dy.numel() + wt.numel()) + ncols,
                            tag=("dcn-colgrad", M, d.Ho, d.Wo, 9 * C, 1, 1, 0, 1, B)):
            check(lib.prn_dcnv2_bwd_input_phase(ref, _p(dy), _p(wt), wimg, _p(offset), _p(mask), _p(dx), _p(ws), _stream(), 1), "prn_dcnv2_bwd_input")
        with profiling.span("reduce_epilogue_kernel", "hbm", 0.0):      # (no-op without a K split: an empty bracket, ~0 us)
            check(lib.prn_dcnv2_bwd_input_phase(ref, _p(dy), _p(wt), wimg, _p(offset), _p(mask), _p(dx), _p(ws), _stream(), 2), "prn_dcnv2_bwd_input")
        if need_x:
            with profiling.span("dcnv2_bwd_input", "hbm", ncols + 4.0 * x.numel() + 8.0 * offset.numel()):
                check(lib.prn_dcnv2_bwd_input_phase(ref, _p(dy), _p(wt), wimg, _p(offset), _p(mask), _p(dx), _p(ws), _stream(), 3), "prn_dcnv2_bwd_input")
    else:
        check(lib.prn_dcnv2_bwd_input(ref, _p(dy), _p(wt), wimg, _p(offset), _p(mask), _p(dx), _p(ws), _stream()), "prn_dcnv2_bwd_input")
    d_off = d_msk = None
    if need_om:
        d_off = torch.empty_like(offset)
        d_msk = torch.empty_like(mask) if (mask is not None and not raw) else None
        if not raw and mask is None:                                # (no modulation given: its gradient is computed and dropped)
            d_msk_tmp = torch.empty(B, 9, d.Ho, d.Wo, device=x.device, dtype=torch.float32)
        else:
            d_msk_tmp = d_msk
        with profiling.span("dcnv2_bwd_offset_mask", "hbm", ncols + 4.0 * x.numel() + 8.0 * offset.numel()):
            check(lib.prn_dcnv2_bwd_offset_mask(ref, _p(x), _p(offset), _p(mask), _p(d_off), _p(d_msk_tmp), _p(ws), _stream()), "prn_dcnv2_bwd_offset_mask")
    return dx, d_off, d_msk


class _DeformConv(torch.autograd.Function):
    """torchvision.ops.deform_conv2d(input, offset, weight, bias, stride, padding, mask=mask) (models/dcn.py:59-66)."""

    @staticmethod
    def forward(ctx, x, offset, weight, bias, mask, stride, pad):
        _dev(x, offset, weight, bias, mask)
        ctx.bias = bias
        x, offset, weight, bias, mask = _c(x), _c(offset), _c(weight), _c(bias), _c(mask)
        M = weight.shape[0]
        table = dcn_table(x.shape, M, offset, mask, stride, pad, 0, 0.0)
        y = dcn_fwd_raw(x, table, weight, bias, stride, pad, 0, 0.0)
        ctx.save_for_backward(x, offset, mask, weight, table)
        ctx.cfg = (stride, pad, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, offset, mask, w, table = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        dy = _c(dy)
        M = w.shape[0]
        ni = ctx.needs_input_grad
        dx = d_off = d_msk = dw = db = None
        if ni[0] or ni[1] or ni[4]:
            dx, d_off, d_msk = dcn_data_grads_raw(x, offset, mask, w, dy, stride, pad, 0, 0.0, need_x=ni[0], need_om=ni[1] or ni[4])
        if _defer(ni[2], w):
            _deferred_wgrad(w, (x, dy, table), lambda: dcn_wgrad_raw(x, table, dy, M, stride, pad, 0, 0.0), fast=True)
        elif ni[2]:
            dw = dcn_wgrad_raw(x, table, dy, M, stride, pad, 0, 0.0)
        db = _bias_grad(has_bias and ni[3], ctx.bias, dy)
        return dx, d_off, dw, db, d_msk, None, None


def _pair(v):
    if isinstance(v, (tuple, list)):
        if len(v) != 2 or v[0] != v[1]:
            raise NotImplementedError("deform_conv2d: symmetric stride / padding / dilation only (got %r)" % (v,))
        return int(v[0])
    return int(v)


def deform_conv2d(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mask=None):
    """Drop-in for `torchvision.ops.deform_conv2d` -- same signature, argument meaning and return value -- for what the
    reference uses (models/dcn.py:59-66): 3x3 kernel, groups = offset groups = 1, dilation 1, symmetric stride / padding.
    offset [B, 18, Ho, Wo] (channel 2k = dy, 2k+1 = dx of tap k), mask [B, 9, Ho, Wo] or None, weight [M, C, 3, 3]."""
    if _pair(dilation) != 1:
        raise NotImplementedError("deform_conv2d: dilation 1 only")
    if tuple(weight.shape[2:]) != (3, 3) or weight.shape[1] != input.shape[1]:
        raise NotImplementedError("deform_conv2d: 3x3 kernels with groups = 1 only (weight %s, input %s)" % (tuple(weight.shape), tuple(input.shape)))
    s, p = _pair(stride), _pair(padding)
    B, _, H, W = input.shape
    Ho, Wo = (H + 2 * p - 3) // s + 1, (W + 2 * p - 3) // s + 1
    if tuple(offset.shape) != (B, 18, Ho, Wo) or (mask is not None and tuple(mask.shape) != (B, 9, Ho, Wo)):
        raise RuntimeError("deform_conv2d: offset %s / mask %s do not match [%d, 18|9, %d, %d] (one offset group)"
                           % (tuple(offset.shape), None if mask is None else tuple(mask.shape), B, Ho, Wo))
    return _DeformConv.apply(input, offset, weight, bias, mask, s, p)


class _DeformConvBlock(torch.autograd.Function):
    """models/dcn.py:52-67 as one node: 27-channel offset|modulator conv + DCNv2 on its RAW output (clamp and 2*sigmoid
    folded into the gather table).  x feeds both, so its two gradients (through the sampler and through the offset conv) are
    summed in the offset conv's input-gradient epilogue.  The offset / modulator parameters are passed for autograd; the
    arithmetic reads the merged [27, C, 3, 3] / [27] tensors they are views of (dcn.DeformableConv2d._merged), and their
    gradients are returned as views of one 27-channel gradient."""

    @staticmethod
    def forward(ctx, x, w_off, w_mod, b_off, b_mod, w27, b27, w, bias, stride, max_offset):
        _dev(x, w27, b27, w, bias)
        ctx.leaves = (w_off, w_mod, b_off, b_mod, bias)
        x, w, bias = _c(x), _c(w), _c(bias)
        B, C, H, W = x.shape
        M = w.shape[0]
        Ho, Wo = _out_hw(H, W, 3, stride, 1, IN_ZERO)
        om = conv_fwd_raw(x, w27, b27, None, 27, 3, stride, 1, Ho, Wo)
        table = dcn_table(x.shape, M, om, None, stride, 1, 1, max_offset)
        y = dcn_fwd_raw(x, table, w, bias, stride, 1, 1, max_offset)
        ctx.save_for_backward(x, om, w27, w, table)
        ctx.cfg = (stride, float(max_offset), bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, om, w27, w, table = ctx.saved_tensors
        stride, max_offset, has_bias = ctx.cfg
        dy = _c(dy)
        M = w.shape[0]
        ni = ctx.needs_input_grad
        dx1, dom, _ = dcn_data_grads_raw(x, om, None, w, dy, stride, 1, 1, max_offset)
        if _defer(ni[7], w):
            _deferred_wgrad(w, (x, dy, table), lambda: dcn_wgrad_raw(x, table, dy, M, stride, 1, 1, max_offset), fast=True)
            dw = None
        else:
            dw = dcn_wgrad_raw(x, table, dy, M, stride, 1, 1, max_offset) if ni[7] else None
        w_off, w_mod, b_off, b_mod, bias_p = ctx.leaves
        db = _bias_grad(has_bias and ni[8], bias_p, dy)
        if BIAS_ASYNC and ni[1] and ni[2] and ni[3] and ni[4] and all(_defer(True, p_) for p_ in (w_off, w_mod, b_off, b_mod)):
            # the offset / modulator conv's parameter gradients (one 27-channel weight gradient, one channel sum of d om) on the side
            # stream as well; each lands in two parameters (views of the 27-row results)
            def om_grads():
                dw27_ = conv_wgrad_raw(x, dom, 27, 3, stride, 1, IN_ZERO)
                db27_ = channel_sum(dom)
                return dw27_[:18], dw27_[18:], db27_[:18], db27_[18:]
            _deferred_wgrad([w_off, w_mod, b_off, b_mod], (x, dom), om_grads, fast=True)
            dw27 = db27 = None
        else:
            dw27 = conv_wgrad_raw(x, dom, 27, 3, stride, 1, IN_ZERO) if (ni[1] or ni[2]) else None
            db27 = channel_sum(dom) if (ni[3] or ni[4]) else None
        dx = conv_dgrad_raw(dom, w27, x.shape, stride, 1, IN_ZERO, dx1) if ni[0] else None
        return (dx, None if dw27 is None else dw27[:18], None if dw27 is None else dw27[18:], None if db27 is None else db27[:18],
                None if db27 is None else db27[18:], None, None, dw, db, None, None)


def deform_conv_block(x, w_off, w_mod, b_off, b_mod, w27, b27, weight, bias, stride, max_offset):
    """models/dcn.py:52-67 in one node: om = conv3x3(x; [offset | modulator] weights), y = deform_conv2d(x, clamp(om[:18]), weight,
    mask = 2 * sigmoid(om[18:])).  w27 / b27: the merged storage the four offset / modulator parameters are views of."""
    return _DeformConvBlock.apply(x, w_off, w_mod, b_off, b_mod, w27, b27, weight, bias, stride, max_offset)


def deform_conv2d_raw_relu(x, om_raw, weight, bias, stride, max_offset):
    """Inference only (no autograd): DCNv2 on the raw offset|modulator map with ReLU in the contraction's epilogue (the
    caller folded the eval-mode BatchNorm into weight / bias)."""
    assert not torch.is_grad_enabled()
    x, om_raw, weight, bias = _c(x), _c(om_raw), _c(weight), _c(bias)
    _dev(x, om_raw, weight, bias)
    table = dcn_table(x.shape, weight.shape[0], om_raw, None, stride, 1, 1, max_offset)
    return dcn_fwd_raw(x, table, weight, bias, stride, 1, 1, max_offset, EPI_RELU)


# ------------------------------------------------------------------------------------------ composite blocks
class _PlanePrior(torch.autograd.Function):
    """Plane prior of the depth decoder as one C call (include/prn.h: prn_plane_prior_fwd).  seg / kernels are detached in the
    reference (planerecnet.py:589,592): only conv1x1's weight and bias receive gradients."""

    @staticmethod
    def forward(ctx, seg, kernels, w1, b1):
        _dev(seg, kernels, w1, b1)
        seg, kernels, w1, b1 = _c(seg), _c(kernels), _c(w1), _c(b1)
        B, E, h, w = seg.shape
        NK, F = kernels.shape[1], w1.shape[0]
        oref = _opts_entry("prior" in SPLIT_SKIP)[1]              # (the block's split-kernel launches cut their weights per call: nothing cached, nothing stale)
        nb = lib.prn_plane_prior_ws_bytes(B, E, h, w, NK, F, oref)
        if nb < 0:
            raise RuntimeError(lib.prn_last_error().decode())
        ws = _f32(nb, seg.device)
        pooled = torch.empty(B, NK, h // 4, w // 4, device=seg.device, dtype=torch.float32)
        out = torch.empty(B, F, h // 4, w // 4, device=seg.device, dtype=torch.float32)
        if profiling._enabled:
            args = (_p(seg), _p(kernels), _p(w1), _p(b1), _p(pooled), _p(out), _p(ws), B, E, h, w, NK, F, oref, _stream())
            npx = (h // 2) * (w // 2)
            g1 = _gemm_family(NK, E, 1, npx, B, 2.0 * NK * E * B * npx)
            g2 = _gemm_family(F, NK, B, npx // 4, 1, 2.0 * F * NK * B * npx / 4)
            for ph, (fam, bound, work, rf, nb_, tag) in enumerate((
                    ("plane_prior_centre_gather", "hbm", 4.0 * 2 * B * E * npx, None, 0.0, None),
                    (g1[0], "mfma", g1[1], g1[2], 4.0 * B * (E * npx + NK * E + NK * npx), ("prior-dynamic", E, h // 2, w // 2, NK, 1, 1, 0, 1, B)),
                    ("resize_down2", "hbm", 4.0 * B * NK * npx * 1.25, None, 0.0, None),
                    (g2[0], "mfma", g2[1], g2[2], 4.0 * (B * NK * npx / 4 + F * NK + B * F * npx / 4), ("conv", NK, h // 4, w // 4, F, 1, 1, 0, 1, B))), 1):
                with profiling.span(fam, bound, work, rf, nbytes=nb_, tag=tag):
                    check(lib.prn_plane_prior_fwd_phase(*args, ph), "prn_plane_prior_fwd")
        else:
            check(lib.prn_plane_prior_fwd(_p(seg), _p(kernels), _p(w1), _p(b1), _p(pooled), _p(out), _p(ws), B, E, h, w, NK, F, oref, _stream()),
                  "prn_plane_prior_fwd")
        ctx.save_for_backward(pooled, w1)
        ctx.dims = (B, h, w, NK, F, b1 is not None)
        ctx.bias = b1
        return out

    @staticmethod
    def backward(ctx, d_out):
        pooled, w1 = ctx.saved_tensors
        B, h, w, NK, F, has_bias = ctx.dims
        d_out = _c(d_out)

        def wgrad():
            dw = torch.empty(F, NK, 1, 1, device=d_out.device, dtype=torch.float32)
            oref = opts_ref()
            ws = _wspace(lib.prn_plane_prior_wgrad_ws_bytes(B, h, w, NK, F, oref), d_out.device)
            with profiling.span("conv_wgrad_kernel", "mfma", 2.0 * F * NK * B * (h // 4) * (w // 4), nbytes=4.0 * (pooled.numel() + d_out.numel() + dw.numel()),
                                tag=("wgrad", NK, h // 4, w // 4, F, 1, 1, 0, 1, B)):      # (GEMM + split sum in one bracket)
                check(lib.prn_plane_prior_wgrad(_p(pooled), _p(d_out), _p(dw), _p(ws), B, h, w, NK, F, oref, _stream()), "prn_plane_prior_wgrad")
            return dw
        dw = None
        if _defer(ctx.needs_input_grad[2], w1):
            _deferred_wgrad(w1, (pooled, d_out), wgrad, fast=True)
        elif ctx.needs_input_grad[2]:
            dw = wgrad().view_as(w1)
        db = _bias_grad(has_bias and ctx.needs_input_grad[3], ctx.bias, d_out)
        return None, None, dw, db


def plane_prior(seg, kernels, w1, b1):
    """planerecnet.py:586-594 in its exact reduced form (include/prn.h: prn_plane_prior_fwd): seg [B,E,h,w], kernels [B,NK,E]."""
    return _PlanePrior.apply(seg, kernels, w1, b1)


def fpn_level(x, w_lat, b_lat, prev, w_out, b_out, relu):
    """One FPN level without autograd (inference): lateral 1x1 (+ resized finer lateral) and the 3x3 output conv as one C call
    (include/prn.h: prn_fpn_level_fwd).  -> (lateral, p_out)."""
    assert not torch.is_grad_enabled()
    _dev(x, w_lat, b_lat, prev, w_out, b_out)
    x, prev = _c(x), _c(prev)
    B, C, H, W = x.shape
    F = w_lat.shape[0]
    U = winograd_weights(w_out)[0] if winograd_ok(B, F, H, W, F, 3, 1, 1, IN_ZERO, EPI_RELU if relu else EPI_NONE) else None
    oref = opts_ref()                  # (split-kernel launches inside the block cut w_lat / U per call: no kept images to go stale)
    nb = lib.prn_fpn_level_ws_bytes(B, C, H, W, F, int(relu), int(prev is not None), int(U is not None), oref)
    if nb < 0:
        raise RuntimeError(lib.prn_last_error().decode())
    ws = _f32(nb, x.device) if nb else None
    lateral = torch.empty(B, F, H, W, device=x.device, dtype=torch.float32)
    p_out = torch.empty(B, F, H, W, device=x.device, dtype=torch.float32)
    Hp, Wp = (prev.shape[2], prev.shape[3]) if prev is not None else (0, 0)
    check(lib.prn_fpn_level_fwd(_p(x), _p(w_lat), _p(b_lat), _p(prev), Hp, Wp, _p(w_out), _p(U), _p(b_out), _p(lateral), _p(p_out), _p(ws), B, C, H, W, F,
                                int(relu), oref, _stream()), "prn_fpn_level_fwd")
    return lateral, p_out


# ------------------------------------------------------------------------------------------ BatchNorm
class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, residual, training, eps, momentum, relu):
        _dev(x, gamma, beta, rmean, rvar, residual)
        x, residual = _c(x), _c(residual)
        B, C, H, W = x.shape
        HW = H * W
        y = torch.empty_like(x)
        if training:
            stats = torch.empty(2 * C, device=x.device, dtype=torch.float32)
            # (the fp64 partial sums exist only on the two-launch path; the library checks the pointer there)
            ws = torch.empty(2 * C * _lib.BN_SPLITS, device=x.device, dtype=torch.float64) if lib.prn_bn_kernel_kind(B, HW) != 1 else None
            if profiling._enabled:
                small = lib.prn_bn_kernel_kind(B, HW) == 1        # one pass (x read once) or statistics pass + apply pass
                r_ = 1 if residual is not None else 0
                with profiling.span("bn_small_fwd" if small else "bn_train_fwd", "hbm", 4.0 * x.numel() * ((2 if small else 3) + r_),
                                    ref=4.0 * x.numel() * (3 + r_)):
                    check(lib.prn_bn_train_fwd(_p(x), _p(stats), _p(gamma), _p(beta), _p(residual), _p(y), _p(rmean), _p(rvar), _p(ws),
                                               B, C, HW, eps, momentum, int(relu), _stream()), "prn_bn_train_fwd")
            else:
                check(lib.prn_bn_train_fwd(_p(x), _p(stats), _p(gamma), _p(beta), _p(residual), _p(y), _p(rmean), _p(rvar), _p(ws),
                                           B, C, HW, eps, momentum, int(relu), _stream()), "prn_bn_train_fwd")
            # the kernel updated the running statistics through raw pointers: tell autograd (version counters), so that
            # everything keyed on them -- backbone.folded_bn's inference cache -- sees the change
            torch.autograd.graph.increment_version(rmean)
            torch.autograd.graph.increment_version(rvar)
        else:
            stats = torch.cat([rmean, torch.rsqrt(rvar + eps)])
            with profiling.span("bn_apply", "hbm", 4.0 * x.numel() * (3 if residual is not None else 2)):
                check(lib.prn_bn_apply(_p(x), _p(stats), _p(gamma), _p(beta), _p(residual), _p(y), B, C, HW, int(relu), _stream()), "prn_bn_apply")
        # the ReLU mask is re-derived from x in the backward unless a residual was added (then the output's sign is needed)
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, stats, gamma, beta)
        ctx.cfg = (training, relu, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, stats, gamma, beta = ctx.saved_tensors
        training, relu, has_res = ctx.cfg
        dy = _c(dy)
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        need_affine = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dg = torch.empty(C, device=x.device, dtype=torch.float32) if need_affine else None
        db = torch.empty(C, device=x.device, dtype=torch.float32) if need_affine else None
        # executed bytes: dy and x (and y, when the ReLU mask comes from the output) are read once by the one-pass kernel, twice by the
        # two-pass pair; dx (and the residual's gradient) written once.  ref = the reference operator chain (ReLU bwd + BN bwd + add)
        small = training and lib.prn_bn_kernel_kind(B, H * W) == 1
        ws = torch.empty(2 * C * _lib.BN_SPLITS, device=x.device, dtype=torch.float64) if not small else None      # (only the two-launch path writes partial sums)
        reads = 2 + (1 if y is not None else 0)
        with profiling.span("bn_small_bwd" if small else "bn_bwd", "hbm", 4.0 * x.numel() * (reads * (1 if small else 2) + 1 + (1 if has_res else 0)),
                            ref=4.0 * x.numel() * ((3 if relu else 2) * 2 + 1 + (1 if has_res else 0))):
            check(lib.prn_bn_bwd(_p(dy), _p(x), _p(y), _p(stats), _p(gamma), _p(beta), _p(dx), _p(dres), _p(dg), _p(db), _p(ws),
                                 B, C, H * W, int(relu), int(not training), _stream()), "prn_bn_bwd")
        return dx, dg, db, None, None, dres, None, None, None, None


class _BatchNormCat(torch.autograd.Function):
    """cat([relu(bn_a(xa)), relu(bn_b(xb))], 1) in training mode: the two layers write the two channel ranges of ONE buffer and their
    backward passes read the two ranges of its gradient in place (include/prn.h: prn_bn_train_fwd_into / prn_bn_bwd_from) -- no
    concatenation kernel and no slice copies of the gradient (DepthDecoder_FPN: 56 M + 52 M elements per step)."""

    @staticmethod
    def forward(ctx, xa, ga, ba, rma, rva, xb, gb, bb, rmb, rvb, eps_a, mom_a, eps_b, mom_b):
        _dev(xa, ga, ba, rma, rva, xb, gb, bb, rmb, rvb)
        xa, xb = _c(xa), _c(xb)
        B, Ca, H, W = xa.shape
        Cb = xb.shape[1]
        assert xb.shape == (B, Cb, H, W)
        HW = H * W
        out = torch.empty(B, Ca + Cb, H, W, device=xa.device, dtype=torch.float32)
        saved = []
        for x, g, b_, rm, rv, eps, mom, c0 in ((xa, ga, ba, rma, rva, eps_a, mom_a, 0), (xb, gb, bb, rmb, rvb, eps_b, mom_b, Ca)):
            C = x.shape[1]
            stats = torch.empty(2 * C, device=x.device, dtype=torch.float32)
            ws = torch.empty(2 * C * _lib.BN_SPLITS, device=x.device, dtype=torch.float64)
            with profiling.span("bn_train_fwd", "hbm", 4.0 * x.numel() * 3):
                check(lib.prn_bn_train_fwd_into(_p(x), _p(stats), _p(g), _p(b_), None, out.data_ptr() + 4 * c0 * HW, (Ca + Cb) * HW, _p(rm), _p(rv), _p(ws),
                                                B, C, HW, eps, mom, 1, _stream()), "prn_bn_train_fwd_into")
            torch.autograd.graph.increment_version(rm)
            torch.autograd.graph.increment_version(rv)
            saved += [x, stats, g, b_]
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, dy):
        xa, sa, ga, ba, xb, sb, gb, bb = ctx.saved_tensors
        dy = _c(dy)
        B, Ca, H, W = xa.shape
        Cb, HW = xb.shape[1], H * W
        grads = []
        for x, stats, g, b_, c0, k in ((xa, sa, ga, ba, 0, 0), (xb, sb, gb, bb, Ca, 5)):
            C = x.shape[1]
            dx = torch.empty_like(x)
            need_affine = ctx.needs_input_grad[k + 1] or ctx.needs_input_grad[k + 2]
            dg = torch.empty(C, device=x.device, dtype=torch.float32) if need_affine else None
            db = torch.empty(C, device=x.device, dtype=torch.float32) if need_affine else None
            ws = torch.empty(2 * C * _lib.BN_SPLITS, device=x.device, dtype=torch.float64)
            with profiling.span("bn_bwd", "hbm", 4.0 * x.numel() * 5):
                check(lib.prn_bn_bwd_from(dy.data_ptr() + 4 * c0 * HW, (Ca + Cb) * HW, _p(x), None, _p(stats), _p(g), _p(b_), _p(dx), None, _p(dg), _p(db), _p(ws),
                                          B, C, HW, 1, 0, _stream()), "prn_bn_bwd_from")
            grads += [dx, dg, db, None, None]
        return tuple(grads) + (None, None, None, None)


def flush_batch_count(m):
    """Write a BatchNorm module's host-side batch count into its `num_batches_tracked` buffer."""
    n = m.__dict__.pop("_prn_nbt_pending", 0)
    if n and m.num_batches_tracked is not None:
        m.num_batches_tracked += n


def _count_batch(m):
    """One more training batch through BatchNorm module `m`: counted on the host (one tiny increment kernel per layer per step
    otherwise) and written to `num_batches_tracked` whenever ANY state dict containing the module is taken -- a state-dict pre-hook on
    the module itself, so net.backbone.state_dict() or a sub-module's are as current as the root's."""
    d = m.__dict__
    d["_prn_nbt_pending"] = d.get("_prn_nbt_pending", 0) + 1
    if "_prn_nbt_hook" not in d:
        d["_prn_nbt_hook"] = m.register_state_dict_pre_hook(lambda mod, prefix, keep_vars: flush_batch_count(mod))


def batch_norm_relu_cat(ma, xa, mb, xb):
    """torch.cat([relu(ma(xa)), relu(mb(xb))], 1) for two nn.BatchNorm2d modules in training mode (see _BatchNormCat)."""
    for m in (ma, mb):
        if m.track_running_stats:
            _count_batch(m)
    return _BatchNormCat.apply(xa, ma.weight, ma.bias, ma.running_mean, ma.running_var, xb, mb.weight, mb.bias, mb.running_mean, mb.running_var,
                               float(ma.eps), float(ma.momentum), float(mb.eps), float(mb.momentum))


def batch_norm_module(m, x, residual=None, relu=False):
    """nn.BatchNorm2d.forward (+ residual add + ReLU) on the HIP kernels, including the module's bookkeeping: in training mode
    `num_batches_tracked` advances like nn.BatchNorm2d's (counted on the host and written to the buffer when a state dict
    is read, see _count_batch)."""
    if m.training and m.track_running_stats:
        _count_batch(m)
    return batch_norm(x, m.weight, m.bias, m.running_mean, m.running_var, m.training, m.eps, m.momentum, residual, relu)


def batch_norm(x, gamma, beta, running_mean, running_var, training, eps=1e-5, momentum=0.1, residual=None, relu=False):
    """F.batch_norm (+ residual add + ReLU) replacement. training=True uses batch statistics and updates the
    running buffers in place (momentum, unbiased variance) like nn.BatchNorm2d."""
    return _BatchNorm.apply(x, gamma, beta, running_mean, running_var, residual, bool(training), float(eps), float(momentum), bool(relu))


# ------------------------------------------------------------------------------------------ GroupNorm + ReLU
class _GroupNormReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps):
        _dev(x, gamma, beta)
        gamma_leaf, beta_leaf = gamma, beta
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(B * groups * 2, device=x.device, dtype=torch.float32)
        with profiling.span("gn_relu_fwd", "hbm", 4.0 * x.numel() * 2):
            check(lib.prn_gn_relu_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), B, C, H * W, groups, eps, _stream()), "prn_gn_relu_fwd")
        ctx.save_for_backward(x, beta, stats, gamma)       # (the backward re-derives the ReLU mask from x: y is not kept)
        ctx.groups = groups
        ctx.leaves = (gamma_leaf, beta_leaf)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, beta, stats, gamma = ctx.saved_tensors
        dy = _c(dy)
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        part = torch.empty(2, B, C, device=x.device, dtype=torch.float32)       # per-image partials of d gamma / d beta: ONE reduction for both
        with profiling.span("gn_relu_bwd", "hbm", 4.0 * x.numel() * 4):
            check(lib.prn_gn_relu_bwd(_p(dy), _p(x), _p(beta), _p(stats), _p(gamma), _p(dx), _p(part[0]), _p(part[1]), B, C, H * W, ctx.groups, _stream()),
                  "prn_gn_relu_bwd")
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            dgb = _small_param_grads(ctx.leaves, (part,), lambda: sum_rows(part).unbind(0), fast=True)
            return (dx, None, None, None, None) if dgb is None else (dx, dgb[0], dgb[1], None, None)
        dgb = sum_rows(part)
        return dx, dgb[0], dgb[1], None, None


def group_norm_relu(x, gamma, beta, groups=32, eps=1e-5):
    return _GroupNormReLU.apply(x, gamma, beta, groups, eps)


# ------------------------------------------------------------------------------------------ ragged batches
class RaggedShape:
    """Geometry of a ragged batch: segments [B, C, H_s, W_s] stored back to back in one flat tensor (include/prn.h: prn_ragged)."""

    def __init__(self, B, sizes):
        self.B, self.sizes = int(B), [(int(h), int(w)) for h, w in sizes]
        self.c = _lib.Ragged(len(self.sizes), (ctypes.c_int32 * 6)(*[h for h, _ in self.sizes]), (ctypes.c_int32 * 6)(*[w for _, w in self.sizes]))
        self.ref = ctypes.byref(self.c)
        self.hw = (ctypes.c_int32 * 6)(*[h * w for h, w in self.sizes])
        self.pixels = sum(self.B * h * w for h, w in self.sizes)
        self.key = (self.B, tuple(self.sizes))

    def supported(self):
        """The HIP path needs whole 64-pixel tiles / 16-pixel chunks per segment (see prn_ragged)."""
        return len(self.sizes) <= 6 and all((self.B * h * w) % 64 == 0 and (h * w) % 4 == 0 for h, w in self.sizes)

    def winograd(self, C, M, K):
        """Ragged 3x3 convs take the Winograd path when every segment has W % 4 == 0 (include/prn.h: prn_conv3x3_winograd_ragged)."""
        if not (WINOGRAD and K == 3 and C >= 64 and M >= 64 and all(w % 4 == 0 and h >= 8 for h, w in self.sizes)):
            return 0
        P = self.__dict__.get("_tiles")
        if P is None:
            P = self._tiles = lib.prn_winograd_tiles_ragged(self.ref, self.B)
        return P if P >= WINOGRAD_MIN_TILES else 0

    def pack(self, tensors):
        return torch.cat([t.reshape(-1) for t in tensors])

    def unpack(self, flat, C):
        # torch.split: ONE backward node that concatenates the five gradients (five slices would each allocate a zero-filled
        # full-size gradient and autograd would add them up)
        parts = torch.split(flat, [self.B * C * h * w for h, w in self.sizes])
        return [p.view(self.B, C, h, w) for p, (h, w) in zip(parts, self.sizes)]


_RDESC = {}


def _rdesc(rs, C, M, K, epi=EPI_NONE):
    oe = _opts_entry()
    key = (rs.key, C, M, K, epi, oe[2])
    e = _RDESC.get(key)
    if e is None:
        h, w = rs.sizes[0]
        d = ConvDesc(rs.B, C, h, w, M, K, K, 1, (K - 1) // 2, h, w, IN_ZERO, 1, epi, 0, 0, 0, 0, oe[0])
        ref = ctypes.byref(d)
        wb = lib.prn_conv2d_wgrad_ragged_ws_bytes(ref, rs.ref)
        if wb < 0:
            raise RuntimeError("prn_conv2d_wgrad_ragged_ws_bytes failed")
        e = _RDESC[key] = (d, ref, wb)
    return e


def _ragged_winograd_raw(xp, U, bias, addend, rs, C, M, P, epi=EPI_NONE):
    y = torch.empty(rs.pixels * M, device=xp.device, dtype=torch.float32)
    # The instance head's towers keep the fp32 MFMA kernel for their 36 products (opts = NULL): five of their parameters are near-cancelling sums
    # behind a GroupNorm backward (condition ~400, tests/test_r101_train_gpu.py) and F(4x4,3x3)'s output transform amplifies product error by its
    # coefficients (up to 8); with the fp16-piece products there, three more tower parameters left the R101 gradient bound (3.5x).
    oref = None
    key = ("wino-fwd-ragged", rs.key, C, M, None)
    nb = _WINO_WG_WS.get(key)
    if nb is None:
        nb = _WINO_WG_WS[key] = lib.prn_conv3x3_winograd_ragged_ws_bytes(rs.ref, rs.B, C, M, oref)
        if nb < 0:
            raise RuntimeError(lib.prn_last_error().decode())
    ws = torch.empty(nb // 4, device=xp.device, dtype=torch.float32)
    uimg = None
    with profiling.span("conv3x3_winograd_ragged", "mfma", 2.0 * 36 * M * C * P, 2.0 * 9 * M * C * rs.pixels):   # (three launches in one bracket)
        check(lib.prn_conv3x3_winograd_ragged(_p(xp), _p(U), uimg, _p(bias), _p(addend), _p(y), _p(ws), rs.ref, rs.B, C, M, epi, oref, _stream()),
              "prn_conv3x3_winograd_ragged")
    return y


def _ragged_conv_raw(xp, w, bias, addend, rs, C, M, K, epi=EPI_NONE):
    y = torch.empty(rs.pixels * M, device=xp.device, dtype=torch.float32)
    _, ref, _ = _rdesc(rs, C, M, K, epi)
    with profiling.span("conv_igemm_kernel", "mfma", 2.0 * M * C * K * K * rs.pixels, nbytes=4.0 * (xp.numel() + w.numel() + rs.pixels * M)):
        check(lib.prn_conv2d_fwd_ragged(ref, rs.ref, _p(xp), _p(w), _p(bias), _p(addend), _p(y), _stream()), "prn_conv2d_fwd_ragged")
    return y


class _RaggedConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, w, bias, rs):
        _dev(xp, w, bias)
        xp, w, bias = _c(xp), _c(w), _c(bias)
        M, C, K, _ = w.shape
        assert xp.numel() == rs.pixels * C, (xp.shape, w.shape, rs.sizes)
        P = rs.winograd(C, M, K)
        y = _ragged_winograd_raw(xp, winograd_weights(w)[0], bias, None, rs, C, M, P) if P else _ragged_conv_raw(xp, w, bias, None, rs, C, M, K)
        ctx.save_for_backward(xp, w)
        ctx.rs, ctx.has_bias = rs, bias is not None
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, w = ctx.saved_tensors
        rs = ctx.rs
        dy = _c(dy)
        M, C, K, _ = w.shape
        dx = dw = db = None
        P = rs.winograd(C, M, K)
        if ctx.needs_input_grad[0]:
            dx = _ragged_winograd_raw(dy, winograd_weights(w)[1], None, None, rs, M, C, P) if P else _ragged_conv_raw(dy, flip_transpose(w), None, None, rs, M, C, K)
        def wgrad():
            if P and WINOGRAD_WGRAD:
                oref = opts_ref()
                key = (rs.key, C, M, opts_key())
                nb = _WINO_WG_WS.get(key)
                if nb is None:
                    nb = _WINO_WG_WS[key] = lib.prn_winograd_wgrad_ragged_ws_bytes(rs.ref, rs.B, C, M, oref)
                ws = torch.empty(nb // 4, device=xp.device, dtype=torch.float32)
                dwo = torch.empty_like(w)
                with profiling.span("conv3x3_winograd_wgrad_ragged", "mfma", 2.0 * 36 * M * C * P, 2.0 * 9 * M * C * rs.pixels):
                    check(lib.prn_conv3x3_winograd_wgrad_ragged(_p(xp), _p(dy), _p(dwo), _p(ws), rs.ref, rs.B, C, M, oref, _stream()), "prn_conv3x3_winograd_wgrad_ragged")
                return dwo
            _, ref, nbytes = _rdesc(rs, C, M, K)
            ws = torch.empty(max(nbytes // 4, 1), device=xp.device, dtype=torch.float32)
            dwo = torch.empty_like(w)
            with profiling.span("conv_wgrad_kernel", "mfma", 2.0 * M * C * K * K * rs.pixels):
                check(lib.prn_conv2d_wgrad_ragged(ref, rs.ref, _p(xp), _p(dy), _p(dwo), _p(ws), _stream()), "prn_conv2d_wgrad_ragged")
            return dwo
        if _defer(ctx.needs_input_grad[1], w):
            _deferred_wgrad(w, (xp, dy), wgrad)
        elif ctx.needs_input_grad[1]:
            dw = wgrad()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            def bias_grad():                                     # per segment a channel sum (library launches), then the segments' sums added in order
                parts = torch.stack([channel_sum(t) for t in rs.unpack(dy, M)])
                if getattr(_TLS, "side", None) is not None:
                    _HELD.append(parts)
                return (sum_rows(parts.unsqueeze(0))[0],)
            r = _small_param_grads((ctx.bias,), (dy,), bias_grad)
            db = None if r is None else r[0]
        return dx, dw, db, None


def ragged_conv2d(xp, w, bias, rs):
    """Stride-1 'same' conv of every segment of the packed batch `xp` with the same weights, as ONE implicit GEMM."""
    return _RaggedConv.apply(xp, w, bias, rs)


class _RaggedGNReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, gamma, beta, groups, eps, rs):
        _dev(xp, gamma, beta)
        ctx.leaves = (gamma, beta)
        xp = _c(xp)
        C = gamma.numel()
        n = len(rs.sizes)
        y = torch.empty_like(xp)
        stats = torch.empty(n * rs.B * groups * 2, device=xp.device, dtype=torch.float32)
        with profiling.span("gn_relu_fwd", "hbm", 4.0 * xp.numel() * 2):
            check(lib.prn_gn_relu_fwd_ragged(_p(xp), _p(gamma), _p(beta), _p(y), _p(stats), rs.B, C, n, rs.hw, groups, eps, _stream()),
                  "prn_gn_relu_fwd_ragged")
        ctx.save_for_backward(xp, beta, stats, gamma)
        ctx.cfg = (groups, rs)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, beta, stats, gamma = ctx.saved_tensors
        groups, rs = ctx.cfg
        dy = _c(dy)
        C = gamma.numel()
        n = len(rs.sizes)
        dx = torch.empty_like(xp)
        part = torch.empty(2, n * rs.B, C, device=xp.device, dtype=torch.float32)
        with profiling.span("gn_relu_bwd", "hbm", 4.0 * xp.numel() * 4):
            check(lib.prn_gn_relu_bwd_ragged(_p(dy), _p(xp), _p(beta), _p(stats), _p(gamma), _p(dx), _p(part[0]), _p(part[1]), rs.B, C, n, rs.hw, groups,
                                             _stream()), "prn_gn_relu_bwd_ragged")
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            dgb = _small_param_grads(ctx.leaves, (part,), lambda: sum_rows(part).unbind(0), fast=True)
            return (dx, None, None, None, None, None) if dgb is None else (dx, dgb[0], dgb[1], None, None, None)
        dgb = sum_rows(part)
        return dx, dgb[0], dgb[1], None, None, None


def ragged_group_norm_relu(xp, gamma, beta, groups, eps, rs):
    return _RaggedGNReLU.apply(xp, gamma, beta, groups, eps, rs)


# ------------------------------------------------------------------------------------------ resampling
class _Resize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, addend=None, fork=False):
        _dev(x, addend)
        x0 = x
        x, addend = _c(x), _c(addend)
        B, C, H, W = x.shape
        y = torch.empty(B, C, Ho, Wo, device=x.device, dtype=torch.float32)
        if addend is not None:
            assert addend.shape == y.shape, (addend.shape, y.shape)
        check(lib.prn_resize_bilinear_add_fwd(_p(x), _p(addend), _p(y), B * C, H, W, Ho, Wo, _stream()), "prn_resize_bilinear_fwd")
        ctx.shape = (B, C, H, W, Ho, Wo)
        if fork:                                             # second output = the input itself (see _Conv2d.forward)
            ctx.set_materialize_grads(False)
            return y, x0
        return y

    @staticmethod
    def backward(ctx, dy, dfork=None):
        B, C, H, W, Ho, Wo = ctx.shape
        if dy is None:                                      # only the forked identity was used
            return dfork, None, None, None, None
        dy = _c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, C, H, W, device=dy.device, dtype=torch.float32)
            check(lib.prn_resize_bilinear_bwd_add(_p(dy), _p(_c(dfork)), _p(dx), B * C, H, W, Ho, Wo, _stream()), "prn_resize_bilinear_bwd")
        return dx, None, None, (dy if ctx.needs_input_grad[3] else None), None


def resize_bilinear(x, size, addend=None):
    """F.interpolate(mode='bilinear', align_corners=False) replacement; size = (Ho, Wo).  addend ([B, C, Ho, Wo]): returns
    resize(x) + addend from the same launch (the level sum of SOLOv2MaskHead)."""
    return _Resize.apply(x, int(size[0]), int(size[1]), addend)


RESIZE_FORK = bool(int(os.environ.get("PRN_RESIZE_FORK", "1")))      # 0: autograd sums the two gradients (cross-check)


def resize_bilinear_fork(x, size):
    """`y, x_id = resize_bilinear_fork(x, size)`: use x_id wherever else x is consumed; the gradient arriving there is then summed
    inside the resize's backward kernel instead of by autograd's accumulation pass (conv2d_fork's counterpart for the FPN maps)."""
    if not (RESIZE_FORK and x.requires_grad and torch.is_grad_enabled()):
        return _Resize.apply(x, int(size[0]), int(size[1]), None), x
    return _Resize.apply(x, int(size[0]), int(size[1]), None, True)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = _c(x)
        B, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(B, C, Ho, Wo, device=x.device, dtype=torch.float32)
        arg = torch.empty(B, C, Ho, Wo, device=x.device, dtype=torch.uint8) if ctx.needs_input_grad[0] else None
        check(lib.prn_maxpool3s2_fwd(_p(x), _p(y), _p(arg), B * C, H, W, Ho, Wo, _stream()), "prn_maxpool3s2_fwd")
        ctx.save_for_backward(arg)
        ctx.in_shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        dy = _c(dy)
        B, C, H, W = ctx.in_shape
        dx = torch.empty(B, C, H, W, device=dy.device, dtype=torch.float32)
        check(lib.prn_maxpool3s2_bwd(_p(arg), _p(dy), _p(dx), B * C, H, W, dy.shape[2], dy.shape[3], _stream()), "prn_maxpool3s2_bwd")
        return dx


def max_pool_3x3_s2(x):
    """nn.MaxPool2d(3, 2, 1) replacement (models/backbone.py:104)."""
    return _MaxPool.apply(x)


if WGRAD_ASYNC:                      # PRN_WGRAD_ASYNC=1 in the environment: same side effects as calling it
    set_wgrad_async(True)
