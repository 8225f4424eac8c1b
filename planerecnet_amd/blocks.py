"""Residual blocks as ONE host call each way (include/prn.h: prn_bottleneck_*; reference: models/backbone.py:53-73, models/dcn.py:52-67).

The reference's unit of work in the backbone is `Bottleneck.forward` (33 calls per batch for PlaneRecNet_101).  Operator by operator that is
6-9 autograd nodes per block, each with its own allocations and C call: ~20 ms of host time per training step against ~42 ms of GPU time.
Here a block in training mode is one `torch.autograd.Function`: forward = `prn_bottleneck_train_fwd`, backward = `prn_bottleneck_train_bwd`;
the C side issues the block's launch sequence, including the producer -> BatchNorm hand-overs that used to be a protocol
between Python call sites.  PyTorch supplies the buffers (per block and step: output, `save`, input gradient, `gsave`, the BatchNorm parameter
gradients; per stream: one scratch workspace); the weight gradients stay with `ops`' deferred side-stream queue (grouped launches across
blocks), fed with pointers into `save` / `gsave`.

The parameter table of a call (weights + the derived operand layouts `ops` caches for them) is filled once and re-used while it is provably
current: nothing in the caches was re-allocated (ops.OPERAND_EPOCH) and either no weight changed, or every derived operand lives in a buffer the
model's per-step refresh rewrites in place and that refresh has run for the weights' current versions (ops.REFRESHED).  Anything else looks the
operands up again, exactly as the operator-by-operator path does launch by launch.

Evaluation mode, frozen BatchNorm, profiler-bracketed runs and CPU tensors never come here (backbone.Bottleneck.forward keeps its
operator-by-operator form for them).
"""
import ctypes
import os
import sys

import torch

from . import _lib, ops, profiling
from ._lib import lib, check

ENABLED = os.environ.get("PRN_BLOCKS", "1") == "1"            # 0: every block operator by operator (A/B, cross-checks)
HANDOVER = os.environ.get("PRN_BLOCK_HANDOVER", "1") == "1"    # 0: every operator of a block writes its own result (A/B)
SCATTER_ACCUMULATE = os.environ.get("PRN_SCATTER_ACC", "1") == "1"      # stride-2 blocks add their input gradient into the gradient the stage output's other readers sent
STATS = {"fwd": 0, "bwd": 0, "resolved": 0, "vouched": 0, "scatter_acc": 0}

BLK_WINOGRAD, BLK_HANDOVER, BLK_KEEP_V, BLK_SCATTER_ACC = 1, 2, 4, 8
CONV2_DIRECT, CONV2_WINOGRAD, CONV2_DCN = 0, 1, 2
(I_SAVE, I_GSAVE, I_FWD_WS, I_BWD_WS, I_HO, I_WO, I_CONV2, I_KEEPS_V, I_A1, I_V, I_A2, I_OM, I_TABLE, I_D1, I_D2, I_D3, I_DD, I_DOM, I_BN_FLOATS, I_HANDOVERS,
 I_W1_IMG, I_W3_IMG, I_WD_IMG, I_W1T_IMG, I_W3T_IMG, I_WDT_IMG, I_U_IMG, I_COLT_IMG, INFO_COUNT) = range(29)

_SCRATCH = {}       # (device index, raw stream) -> float32 tensor: launches on a stream are ordered, and no block reads its scratch after its call
_F32 = torch.float32
_empty = torch.empty


def scratch(nbytes, dev):
    key = (dev.index, ops._raw_stream(dev.index))
    t = _SCRATCH.get(key)
    if t is None or t.numel() * 4 < nbytes:
        t = _SCRATCH[key] = torch.empty((int(nbytes * 1.25) + 1023) // 4, device=dev, dtype=torch.float32)
    return t


class _Ref:
    """A [shape] fp32 tensor at a fixed address inside a buffer this module keeps alive: what the weight-gradient launches need of a tensor
    (pointer, shape, device) without building a view (`torch.Tensor.view` costs ~3 us; a block hands out eight of these per step)."""
    __slots__ = ("ptr", "shape", "device")

    def __init__(self, ptr, shape, device):
        self.ptr, self.shape, self.device = ptr, shape, device

    def data_ptr(self):
        return self.ptr

    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


class _State:
    """Plan, parameter table and operand bookkeeping of one block for one input shape under one set of execution options."""
    __slots__ = ("plan", "plan_ref", "info", "params", "params_ref", "convw", "wver", "epoch", "batched", "keep", "hb", "bn_sizes", "out_shape", "Ho", "Wo", "conv2",
                 "dims", "wino_wgrad", "g_keys", "max_offset", "ps", "bns", "stat_bufs", "save_n", "gsave_n", "bn_n", "fwd_ws", "bwd_ws", "stride", "dcn", "ds",
                 "shapes")


def _flags():
    f = BLK_SCATTER_ACC if SCATTER_ACCUMULATE else 0
    if ops.WINOGRAD:
        f |= BLK_WINOGRAD
    if ops.WINOGRAD and ops.WINOGRAD_WGRAD:
        f |= BLK_KEEP_V
    if HANDOVER:
        f |= BLK_HANDOVER
    return f


def supported(blk):
    """Structure the block entry points cover (the reference's Bottleneck in every configuration it builds; models/backbone.py:5-52)."""
    from .dcn import DeformableConv2d
    c2 = blk.conv2
    if blk.downsample is not None:
        dc = blk.downsample[0]
        if dc.kernel_size != (1, 1) or dc.stride[0] != blk.stride or dc.bias is not None:
            return False
    elif blk.stride != 1 or blk.conv1.weight.shape[1] != blk.conv3.weight.shape[0]:
        return False
    if isinstance(c2, DeformableConv2d):
        if c2.stride != blk.stride:
            return False
    elif c2.bias is not None or c2.kernel_size != (3, 3) or c2.padding != (1, 1) or c2.stride[0] != blk.stride:
        return False
    bns = [blk.bn1, blk.bn2, blk.bn3] + ([blk.downsample[1]] if blk.downsample is not None else [])
    return (blk.stride in (1, 2) and ops.WINOGRAD_MIN_TILES == 128 and blk.conv1.bias is None and blk.conv3.bias is None and blk.conv1.kernel_size == (1, 1)
            and blk.conv3.kernel_size == (1, 1) and all(isinstance(m, torch.nn.BatchNorm2d) and m.affine and m.track_running_stats and m.momentum is not None for m in bns))


def _param_list(blk):
    from .dcn import DeformableConv2d
    c2 = blk.conv2
    ps = [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias]
    if isinstance(c2, DeformableConv2d):
        ps += [c2.offset_conv.weight, c2.modulator_conv.weight, c2.offset_conv.bias, c2.modulator_conv.bias, c2.regular_conv.weight, c2.regular_conv.bias]
    else:
        ps += [c2.weight]
    ps += [blk.bn2.weight, blk.bn2.bias, blk.conv3.weight, blk.bn3.weight, blk.bn3.bias]
    if blk.downsample is not None:
        ps += [blk.downsample[0].weight, blk.downsample[1].weight, blk.downsample[1].bias]
    return tuple(ps)


def _state(blk, x, key):
    from .dcn import DeformableConv2d
    oe = ops._opts_entry()
    B, C, H, W = x.shape
    dcn = isinstance(blk.conv2, DeformableConv2d)
    bns = (blk.bn1, blk.bn2, blk.bn3) + ((blk.downsample[1],) if blk.downsample is not None else ())
    eps = [float(m.eps) for m in bns] + [0.0] * (4 - len(bns))
    mom = [float(m.momentum) for m in bns] + [0.0] * (4 - len(bns))
    P = blk.conv1.weight.shape[0]
    d = _lib.BottleneckDesc(B, C, H, W, P, blk.stride, int(dcn), int(blk.downsample is not None), key[2], (ctypes.c_float * 4)(*eps), (ctypes.c_float * 4)(*mom),
                            max(H, W) / 4.0 if dcn else 0.0, 0, oe[0])
    st = _State()
    st.plan = ctypes.create_string_buffer(int(lib.prn_bottleneck_plan_bytes()) + 64)
    st.plan_ref = (ctypes.addressof(st.plan) + 63) & ~63
    check(lib.prn_bottleneck_plan(ctypes.byref(d), st.plan_ref), "prn_bottleneck_plan")
    info = (ctypes.c_int64 * INFO_COUNT)()
    check(lib.prn_bottleneck_plan_info(st.plan_ref, info, INFO_COUNT), "prn_bottleneck_plan_info")
    st.info = info = list(info)
    st.params = _lib.BottleneckParams()
    st.params_ref = ctypes.byref(st.params)
    st.wver, st.epoch, st.keep, st.batched = None, -1, None, False
    st.Ho, st.Wo, st.conv2 = info[I_HO], info[I_WO], info[I_CONV2]
    st.out_shape = (B, 4 * P, st.Ho, st.Wo)
    st.dims = (B, C, H, W, P)
    st.stride, st.dcn, st.ds = blk.stride, dcn, blk.downsample is not None
    st.bn_sizes = [P, P, P, P, 4 * P, 4 * P] + ([4 * P, 4 * P] if st.ds else [])
    st.hb = bool(st.ds and blk.stride == 2)
    st.wino_wgrad = bool(st.conv2 == CONV2_WINOGRAD and ops.WINOGRAD_WGRAD)
    st.max_offset = max(H, W) / 4.0
    st.save_n, st.gsave_n, st.bn_n, st.fwd_ws, st.bwd_ws = info[I_SAVE] // 4, info[I_GSAVE] // 4, info[I_BN_FLOATS], info[I_FWD_WS], info[I_BWD_WS]
    st.ps = _param_list(blk)
    st.bns = bns
    st.stat_bufs = [t for m in bns for t in (m.running_mean, m.running_var)]
    c2 = blk.conv2
    # the weights whose derived layouts the table points at (a DCN block's offset / modulator parameters guard the merged 27-channel weight)
    st.convw = (blk.conv1.weight, blk.conv3.weight) + ((c2.offset_conv.weight, c2.modulator_conv.weight, c2.regular_conv.weight) if dcn else (c2.weight,)) + (
        (blk.downsample[0].weight,) if st.ds else ())
    Ho, Wo, s = st.Ho, st.Wo, blk.stride
    Q = 4 * P
    st.shapes = ((B, P, H, W), (B, P, Ho, Wo), (B, Q, Ho, Wo), (B, 27, Ho, Wo))
    # shape-group keys of the deferred weight gradients (ops._deferred_wgrad: layers of one shape are computed by one grouped launch)
    pix = ops.WGRAD_GROUP_PIXELS
    st.g_keys = [("conv", (B, C, H, W), (H, W), P, 1, 1, 0, ops.IN_ZERO) if B * H * W <= pix else None,
                 ("conv", (B, P, H, W), (Ho, Wo), P, 3, s, 1, ops.IN_ZERO) if (st.conv2 == CONV2_DIRECT and B * Ho * Wo <= pix) else None,
                 ("conv", (B, P, Ho, Wo), (Ho, Wo), Q, 1, 1, 0, ops.IN_ZERO) if B * Ho * Wo <= pix else None,
                 ("conv", (B, C, H, W), (Ho, Wo), Q, 1, s, 0, ops.IN_ZERO) if B * Ho * Wo <= pix else None]
    return st


def _resolve(blk, st):
    """Fill the parameter table: the weights and the derived operand layouts `ops` keeps for them (cut images, input-gradient layouts,
    Winograd-domain weights) -- what the operator-by-operator path looks up launch by launch.  st.keep: the tensors the table points into;
    st.batched: every derived layout lives in a buffer the model's per-step refresh rewrites in place (ops.BATCHED)."""
    p, info = st.params, st.info
    B, C, H, W, P = st.dims
    Ho, Wo = st.Ho, st.Wo
    Q = 4 * P
    keep = []
    ptr = ops._ptr
    batched = [True]
    BATCHED = ops.BATCHED

    def derived(t):
        keep.append(t)
        if t.data_ptr() not in BATCHED:
            batched[0] = False
        return t

    def images(t, M, K, nz, cols, want):
        return ops.split_images_ptr(t, M, K, nz, cols) if want else None
    w1, w3 = blk.conv1.weight, blk.conv3.weight
    p.w1, p.w3 = ptr(w1), ptr(w3)
    p.w1_img = images(w1, P, C, 1, (B, H * W), info[I_W1_IMG])
    p.w3_img = images(w3, Q, P, 1, (B, Ho * Wo), info[I_W3_IMG])
    w1t, w3t = derived(ops.flip_transpose(w1)), derived(ops.flip_transpose(w3))
    p.w1_t, p.w3_t = ptr(w1t), ptr(w3t)
    p.w1_t_img = images(w1t, C, P, 1, (B, H * W), info[I_W1T_IMG])
    p.w3_t_img = images(w3t, P, Q, 1, (B, Ho * Wo), info[I_W3T_IMG])
    p.wd = p.wd_t = p.wd_img = p.wd_t_img = None
    if st.ds:
        wd = blk.downsample[0].weight
        wdt = derived(ops.flip_transpose(wd))
        p.wd, p.wd_t = ptr(wd), ptr(wdt)
        p.wd_img = images(wd, Q, C, 1, (B, Ho * Wo), info[I_WD_IMG])
        p.wd_t_img = images(wdt, C, Q, 1, (B, H * W), info[I_WDT_IMG])
    p.w2 = p.b2 = p.w2_t = p.u2 = p.ut2 = p.u2_img = p.ut2_img = p.w27 = p.b27 = p.w27_t = p.w2_cols_t = p.w2_cols_t_img = None
    if st.conv2 == CONV2_DCN:
        c2 = blk.conv2
        w27, b27 = c2._merged()
        w2 = c2.regular_conv.weight
        w27t = derived(ops.flip_transpose(w27))
        colt = derived(ops.flip_transpose(w2.view(P, P * 9, 1, 1)))
        p.w2, p.b2 = ptr(w2), ptr(c2.regular_conv.bias)
        p.w27, p.b27, p.w27_t, p.w2_cols_t = ptr(w27), ptr(b27), ptr(w27t), ptr(colt)
        p.w2_cols_t_img = images(colt, 9 * P, P, 1, (B, Ho * Wo), info[I_COLT_IMG])
        keep += [w27, b27]
    elif st.conv2 == CONV2_WINOGRAD:
        w2 = blk.conv2.weight
        U, Ut = ops.winograd_weights(w2)
        derived(U), derived(Ut)
        P4 = lib.prn_winograd_tiles(B, H, W)
        p.u2, p.ut2 = ptr(U), ptr(Ut)
        p.u2_img, p.ut2_img = images(U, P, P, 36, (1, P4), info[I_U_IMG]), images(Ut, P, P, 36, (1, P4), info[I_U_IMG])
        p.w2 = ptr(w2)
    else:
        w2 = blk.conv2.weight
        p.w2, p.w2_t = ptr(w2), ptr(derived(ops.flip_transpose(w2)))
    for i, m in enumerate(st.bns):
        p.gamma[i], p.beta[i], p.running_mean[i], p.running_var[i] = ptr(m.weight), ptr(m.bias), ptr(m.running_mean), ptr(m.running_var)
    st.keep, st.batched = keep, batched[0]
    STATS["resolved"] += 1


class _BottleneckFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blk, st, hand_back, *ps):
        x0 = x
        if not x.is_contiguous():
            x = x.contiguous()
        dev = x.device
        # the parameter table is re-used while it is provably current (module docstring); otherwise the operands are looked up again
        vs = [w._version for w in st.convw]
        epoch = ops.OPERAND_EPOCH[0]
        if vs != st.wver or st.epoch != epoch:
            R = ops.REFRESHED
            if st.batched and st.epoch == epoch and all([R.get(id(w)) == v for w, v in zip(st.convw, vs)]):
                STATS["vouched"] += 1
            else:
                _resolve(blk, st)
                st.epoch = ops.OPERAND_EPOCH[0]
            st.wver = vs
        y = _empty(st.out_shape, device=dev, dtype=_F32)
        save = _empty(st.save_n, device=dev, dtype=_F32)
        ws = scratch(st.fwd_ws, dev)
        rc = lib.prn_bottleneck_train_fwd(st.plan_ref, st.params_ref, x.data_ptr(), y.data_ptr(), save.data_ptr(), ws.data_ptr(), ops._raw_stream(dev.index))
        if rc:
            check(rc, "prn_bottleneck_train_fwd")
        # the kernels updated the running statistics through raw pointers: tell autograd (version counters; see ops._BatchNorm.forward)
        torch._C._increment_version(st.stat_bufs)
        for m in st.bns:
            ops._count_batch(m)
        ctx.save_for_backward(x, y, save)
        ctx.st, ctx.keep, ctx.vs = st, st.keep, vs
        STATS["fwd"] += 1
        if hand_back:
            # second output = the input itself: the block's other readers take it from here, so that their gradient arrives at THIS node
            # and the strided input gradient is added into it in place (see backward)
            ctx.set_materialize_grads(False)
            return y, x0
        return y

    @staticmethod
    def backward(ctx, dy, dfork=None):
        x, y, save = ctx.saved_tensors
        st = ctx.st
        ps = st.ps
        if dy is None:                                      # only the handed-back identity was used
            return (dfork,) + (None,) * (3 + len(ps))
        if [w._version for w in st.convw] != ctx.vs:
            raise RuntimeError("a parameter of a Bottleneck block was modified between its forward and its backward pass")
        dev = x.device
        if not dy.is_contiguous():
            dy = dy.contiguous()
        info = st.info
        acc = 0
        if dfork is not None and st.hb and SCATTER_ACCUMULATE and dfork.is_contiguous() and dfork._base is None and dfork.shape == x.shape \
                and dfork.dtype == _F32 and dfork._use_count() <= _OWNED[0] and sys.getrefcount(dfork) <= _OWNED[1]:
            dx, acc, dfork = dfork, 1, None
            STATS["scatter_acc"] += 1
        else:
            dx = torch.empty_like(x)
        gsave = _empty(st.gsave_n, device=dev, dtype=_F32)
        bn_buf = _empty(st.bn_n, device=dev, dtype=_F32)
        ws = scratch(st.bwd_ws, dev)
        rc = lib.prn_bottleneck_train_bwd(st.plan_ref, st.params_ref, x.data_ptr(), y.data_ptr(), dy.data_ptr(), dx.data_ptr(), acc, save.data_ptr(), gsave.data_ptr(),
                                          bn_buf.data_ptr(), ws.data_ptr(), ops._raw_stream(dev.index))
        if rc:
            check(rc, "prn_bottleneck_train_bwd")
        ctx.keep = None
        STATS["bwd"] += 1
        if dfork is not None:
            dx = dx + dfork
        bn = bn_buf.split(st.bn_sizes)                       # BatchNorm parameter gradients: slices of one small buffer
        # ---- weight gradients: the caller's launches (side stream, grouped across blocks) over operands in save / gsave
        B, C, H, W, P = st.dims
        Q = 4 * P
        s = st.stride
        sh1, sh2, sh3, sh27 = st.shapes
        sp, gp = save.data_ptr(), gsave.data_ptr()
        a1 = _Ref(sp + info[I_A1], sh1, dev)
        a2 = _Ref(sp + info[I_A2], sh2, dev)
        d1 = _Ref(gp + info[I_D1], sh1, dev)
        d2 = _Ref(gp + info[I_D2], sh2, dev)
        d3 = _Ref(gp + info[I_D3], sh3, dev)
        need = ctx.needs_input_grad
        grads = [None] * len(ps)
        items = []                                           # (index in ps | list of indices, weight(s), inputs, compute, group key)
        gk = st.g_keys
        conv_w = ops.conv_wgrad_raw
        ZERO = ops.IN_ZERO
        items.append((0, ps[0], (x, d1), lambda: conv_w(x, d1, P, 1, 1, 0, ZERO), gk[0]))
        grads[1], grads[2] = bn[0], bn[1]
        if st.conv2 == CONV2_DCN:
            w_off, w_mod, b_off, b_mod, w2, b2 = ps[3:9]
            dom = _Ref(gp + info[I_DOM], sh27, dev)
            table = _Ref(sp + info[I_TABLE], (1,), dev)
            mo = st.max_offset
            items.append((7, w2, (a1, d2, table), lambda: ops.dcn_wgrad_raw(a1, table, d2, P, s, 1, 1, mo), None))
            if b2 is not None:
                items.append((8, b2, (d2,), lambda: ops.channel_sum(d2), None))

            def om_grads():
                dw27 = conv_w(a1, dom, 27, 3, s, 1, ZERO)
                db27 = ops.channel_sum(dom)
                return dw27[:18], dw27[18:], db27[:18], db27[18:]
            items.append(([3, 4, 5, 6], [w_off, w_mod, b_off, b_mod], (a1, dom), om_grads, None))
            i = 9
        else:
            w2 = ps[3]
            if st.wino_wgrad:
                V = _Ref(sp + info[I_V], (1,), dev) if info[I_KEEPS_V] else None
                items.append((3, w2, (a1, d2), lambda: ops.conv3x3_winograd_wgrad_raw(a1, d2, P, ZERO, V), None))
            else:
                items.append((3, w2, (a1, d2), lambda: conv_w(a1, d2, P, 3, s, 1, ZERO), gk[1]))
            i = 4
        grads[i], grads[i + 1] = bn[2], bn[3]
        items.append((i + 2, ps[i + 2], (a2, d3), lambda: conv_w(a2, d3, Q, 1, 1, 0, ZERO), gk[2]))
        grads[i + 3], grads[i + 4] = bn[4], bn[5]
        if st.ds:
            dd = _Ref(gp + info[I_DD], sh3, dev)
            items.append((i + 5, ps[i + 5], (x, dd), lambda: conv_w(x, dd, Q, 1, s, 0, ZERO), gk[3]))
            grads[i + 6], grads[i + 7] = bn[6], bn[7]
        ops.queue_wgrads(items, (x, save, gsave), need, 4, grads)
        if not all(need[4:]):
            for k in range(len(ps)):
                if not need[4 + k]:
                    grads[k] = None
        return (dx, None, None, None) + tuple(grads)


# Adding into an incoming gradient in place is only safe when this node is its sole owner (autograd's own in-place accumulation makes the same
# check).  What "sole owner" looks like from inside backward() -- the TensorImpl's use count and the Python object's reference count -- is
# measured once, on a gradient that provably has no other holder; a tensor a hook kept, a retained gradient or one another node also returned
# elsewhere shows a higher count and is left alone (the block then writes its own dx and the two are summed).
def _calibrate_ownership():
    seen = []

    class _Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            ctx.set_materialize_grads(False)
            return x * 2, x

        @staticmethod
        def backward(ctx, dy, dfork=None):
            seen.append((dfork._use_count(), sys.getrefcount(dfork)))
            return dfork

    try:
        with torch.inference_mode(False), torch.enable_grad():          # (the package may be imported from inside a no_grad / inference_mode region)
            x = torch.zeros(2, requires_grad=True)
            _, xi = _Probe.apply(x)
            (xi * torch.ones(2)).sum().backward()
        return seen[0]
    except Exception:                                                   # no measurement, no in-place accumulation: every block writes its own dx
        return (0, 0)


_OWNED = _calibrate_ownership()


def usable(blk, x):
    """Does this call go through the block entry points?  Training mode with autograd on, on the device, not under the launch-bracketing profiler."""
    if not (ENABLED and x.is_cuda and torch.is_grad_enabled() and not profiling._enabled and x.dtype == _F32 and x.dim() == 4):
        return False
    d = blk.__dict__
    ok = d.get("_prn_block_ok")
    if ok is None:
        ok = d["_prn_block_ok"] = supported(blk)
    if not ok:
        return False
    for m in (d.get("_prn_block_bns") or d.setdefault("_prn_block_bns", (blk.bn1, blk.bn2, blk.bn3) + ((blk.downsample[1],) if blk.downsample is not None else ()))):
        if not m.training:
            return False
    return x.shape[2] >= 3 and x.shape[3] >= 3


def invalidate(blk):
    """Forget everything derived from the block's parameter OBJECTS and addresses (Module._apply: .to() / .cuda() / .float() replace them)."""
    for k in ("_prn_block_states", "_prn_block_ok", "_prn_block_bns"):
        blk.__dict__.pop(k, None)


def bottleneck_train(blk, x, hand_back=False):
    """Bottleneck.forward in training mode as one node.  hand_back (stride-2 blocks with a downsample branch): -> (out, x handed back), see backbone.Bottleneck.forward."""
    key = (x.shape, ops._opts_entry()[2], _flags(), x.device.index)
    cache = blk.__dict__.get("_prn_block_states")
    if cache is None:
        cache = blk.__dict__["_prn_block_states"] = {}
    st = cache.get(key)
    if st is not None and (blk.conv1.weight is not st.ps[0] or blk.conv3.weight is not st.convw[1]):      # a parameter OBJECT was replaced
        invalidate(blk)
        cache = blk.__dict__["_prn_block_states"] = {}
        st = None
    if st is None:
        st = cache[key] = _state(blk, x, key)
    if hand_back and not st.hb:
        return _BottleneckFn.apply(x, blk, st, False, *st.ps), x
    return _BottleneckFn.apply(x, blk, st, bool(hand_back), *st.ps)
