#!/usr/bin/env python
"""Debug aid: where do the kernel-tower gradients of the Winograd and the direct build part ways?  Runs the same train step
twice (ops.WINOGRAD on / off) and compares the gradient that reaches each stage of the instance head."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth  # noqa: E402
from planerecnet_amd import ops  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402

CN = sys.argv[1] if len(sys.argv) > 1 else "PlaneRecNet_50_config"
set_cfg(CN)
sd = synth.make_state_dict(CN, seed=3)
net = PlaneRecNet(cfg)
net.load_state_dict(sd)
net = net.cuda().train()
x, inst, gtd = synth.make_batch(2, 480, 640, seed=12)
crit = PlaneRecNetLoss().cuda()
inst_d = [{k: v.cuda() for k, v in g.items()} for g in inst]


def run(wino):
    ops.WINOGRAD = wino
    net.load_state_dict(sd)
    grads = {}
    hooks = []

    def grab(name):
        def h(mod, gin, gout):
            grads[name] = gout[0].detach().clone()
        return h
    for name, m in net.inst_head.named_modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.GroupNorm)):
            hooks.append(m.register_full_backward_hook(grab(name)))
    np.random.seed(13)
    out = net(x.cuda())
    acts = {"kern%d" % i: k.detach().clone() for i, k in enumerate(out[2])}
    kp_grads = {}
    for i, k in enumerate(out[2]):
        k.register_hook(lambda g, i=i: kp_grads.__setitem__("d_kern%d" % i, g.detach().clone()))
    losses = crit(net, *out, inst_d, gtd.cuda())
    net.zero_grad(set_to_none=True)
    sum(losses.values()).sum().backward()
    ops.wgrad_join()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    pg = {n: p.grad.detach().clone() for n, p in net.inst_head.named_parameters()}
    return acts, kp_grads, pg, {k: float(v) for k, v in losses.items()}


ops.set_wgrad_async(len(sys.argv) > 2 and sys.argv[2] == "async")
a1, k1, p1, l1 = run(True)
a0, k0, p0, l0 = run(False)
print("losses wino", l1)
print("losses direct", l0)


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


for k in a1:
    print("fwd %-10s rel %.2e" % (k, rel(a1[k], a0[k])))
for k in sorted(k1):
    print("grad at %-8s rel %.2e  max|g| %.2e  nonzero frac %.4f" % (k, rel(k1[k], k0[k]), k0[k].abs().max().item(), (k0[k] != 0).float().mean().item()))
for n in sorted(p1):
    print("param %-28s rel %.2e" % (n, rel(p1[n], p0[n])))
