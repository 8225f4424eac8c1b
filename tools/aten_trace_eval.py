#!/usr/bin/env python
"""ATen operator calls of one eval-mode forward + post-process (PlaneRecNet_50, batch 1 by default), by source line."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from planerecnet_amd import timer  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402

B = int(os.environ.get("B", "1"))
timer.disable_all()
set_cfg("PlaneRecNet_50_config")
torch.manual_seed(0)
net = PlaneRecNet(cfg)
net.init_head_weights()
net = net.cuda().eval()
x = torch.randn(B, 3, 480, 640, device="cuda")
LOG = collections.Counter()
SKIP = {"view", "_unsafe_view", "as_strided", "t", "transpose", "permute", "slice", "select", "expand", "unsqueeze", "squeeze", "detach", "alias", "empty",
        "empty_like", "empty_strided", "reshape", "unbind", "split", "split_with_sizes", "narrow", "_reshape_alias", "new_empty", "flatten", "size", "stride"}


class Trace(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        if name in SKIP:
            return out
        where = "?"
        for fr in reversed(traceback.extract_stack(limit=30)):
            if "planerecnet_amd/" in fr.filename:
                where = "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
                break
        LOG[(name, where)] += 1
        return out


with torch.no_grad():
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    with Trace():
        net(x)
for (name, where), n in sorted(LOG.items(), key=lambda kv: -kv[1]):
    print("%4d  %-28s %s" % (n, name, where))
print("total", sum(LOG.values()))
