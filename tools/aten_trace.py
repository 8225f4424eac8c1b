#!/usr/bin/env python
"""Which lines of this package issue the ATen operators of a training step?  A TorchDispatchMode logs every aten call of one
step with (forward) the innermost planerecnet_amd / bench frame or (backward) the autograd node that is running; view-only and
allocation ops are dropped.  Complements tools/aten_ops.py (which has the device times but no attribution on this build).

    python tools/aten_trace.py [--min 2]"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import ops, timer  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss, TargetPrefetcher  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--min", type=int, default=1)
args = ap.parse_args()
VIEWS = {"view", "_unsafe_view", "as_strided", "t", "transpose", "permute", "slice", "select", "expand", "unsqueeze", "squeeze", "detach", "alias",
         "empty", "empty_like", "empty_strided", "reshape", "unbind", "split", "split_with_sizes", "narrow", "unfold", "_reshape_alias", "size",
         "stride", "sym_size", "sym_numel", "numel", "is_same_size", "_local_scalar_dense", "lift_fresh", "new_empty", "new_empty_strided",
         "view_as_real", "chunk", "diagonal", "movedim", "flatten", "resize_", "set_", "record_stream", "is_pinned", "_has_compatible_shallow_copy_type"}

timer.disable_all()
torch.set_num_threads(4)
dev = torch.device("cuda:0")
set_cfg("PlaneRecNet_101_config")
torch.manual_seed(0)
net = PlaneRecNet(cfg)
net.init_head_weights()
net = net.to(dev).train()
crit = PlaneRecNetLoss().to(dev)
opt = __import__("planerecnet_amd.optim", fromlist=["FusedAdam"]).FusedAdam(net.parameters(), lr=1e-4)
images, inst, depths = bench.synth_batch(8, 480, 640, 1000, dev)
pf = TargetPrefetcher(crit)
pf.submit(inst, (480, 640))
ops.set_wgrad_async(True)
PHASE = ["?"]
LOG = collections.defaultdict(lambda: [0, 0])


class Trace(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        if name in VIEWS:
            return out
        node = torch._C._current_autograd_node()
        where = None
        if node is not None:
            where = "<%s>" % node.name()
        else:
            for fr in reversed(traceback.extract_stack(limit=24)):
                if ("planerecnet_amd/" in fr.filename or fr.filename.endswith("bench.py")) and "aten_trace" not in fr.filename:
                    where = "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
                    break
        t = out if isinstance(out, torch.Tensor) else next((a for a in args if isinstance(a, torch.Tensor)), None)
        n = t.numel() if isinstance(t, torch.Tensor) and t.is_cuda else -1
        if n < 0:
            return out                                          # host-side op
        if os.environ.get("ATEN_SHAPES") and n >= 1000000:     # large ops one by one: shape and strides of the first tensor argument
            a0 = next((a for a in args if isinstance(a, torch.Tensor)), t)
            where = "%s %s in%s%s" % (where or "(other)", tuple(t.shape), tuple(a0.shape), "" if a0.is_contiguous() else " strided%s" % (tuple(a0.stride()),))
        e = LOG[(PHASE[0], name, where or "(other)")]
        e[0] += 1
        e[1] += n
        return out


def step(trace=False):
    opt.zero_grad(set_to_none=True)
    PHASE[0] = "targets"
    t = pf.get(depths, dev)
    pf.submit(inst, (480, 640))
    PHASE[0] = "forward"
    out = net(images)
    PHASE[0] = "loss"
    losses = crit(net, *out, inst, depths, targets=t)
    tot = sum(losses.values()).sum()
    PHASE[0] = "backward"
    tot.backward()
    ops.wgrad_join()
    PHASE[0] = "adam"
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with Trace():
    step()
torch.cuda.synchronize()
tot = collections.Counter()
for (ph, name, where), (c, n) in LOG.items():
    tot[ph] += c
print("device-side aten calls in one step:", dict(tot), "total", sum(tot.values()))
print("%-9s %-26s %-46s %5s %12s" % ("phase", "op", "issued from", "calls", "elements"))
for (ph, name, where), (c, n) in sorted(LOG.items(), key=lambda kv: (-kv[1][1])):
    if c >= args.min:
        print("%-9s %-26s %-46s %5d %12d" % (ph, name, where, c, n))
pf.close()
