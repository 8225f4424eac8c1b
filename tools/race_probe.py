#!/usr/bin/env python
"""Stress form of tools/determinism_probe.py for the configuration of tests/test_r101_train_gpu.py: PlaneRecNet_101, B = 2, the loss's own (inline) target
preparation, deferred weight gradients, a chosen GEMM arithmetic (ARITH = default | b8-plan | all-f16 | all-bf16 | fp32) -- REPS repetitions of the same step,
every gradient compared bit for bit with the first.  A repetition that differs names the tensors and how many elements moved: a stream-ordering bug (memory
re-used while the side stream still reads it) shows up as a sporadic difference in a few tensors; arithmetic never does.
    ARITH=all-f16 python tools/race_probe.py [repetitions=30]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth  # noqa: E402   (inputs and weights of the parity tests; the oracle's arithmetic is not used)
from planerecnet_amd import ops, timer  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ARITH = {"default": {}, "b8-plan": {"mode": 1, "kind": "f16", "min_tiles": 75, "min_gflop": 1.0}, "all-f16": {"mode": 2, "kind": "f16"},
         "all-bf16": {"mode": 2, "kind": "bf16"}, "fp32": {"mode": 0}}[os.environ.get("ARITH", "all-f16")]
timer.disable_all()
torch.set_num_threads(4)
CN = "PlaneRecNet_101_config"
set_cfg(CN)
sd = synth.make_state_dict(CN, seed=int(os.environ.get("WSEED", "4")))
net = PlaneRecNet(cfg)
net.load_state_dict(sd)
net = net.cuda().train()
x, inst, gtd = synth.make_batch(int(os.environ.get("BATCH", "2")), 480, 640, seed=13)
x, gtd = x.cuda(), gtd.cuda()
inst = [{k: v.cuda() for k, v in g.items()} for g in inst]
crit = PlaneRecNetLoss().cuda()
ops.set_split_gemm(**ARITH)
ops.WINOGRAD = os.environ.get("WINOGRAD", "1") == "1"
ops.set_wgrad_async(os.environ.get("ASYNC", "1") == "1")
params = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
start = {k: v.detach().clone() for k, v in net.state_dict().items()}
first, bad = None, 0
rng = np.random.RandomState(0)
junk = []
for rep in range(REPS):
    if os.environ.get("PERTURB", "1") == "1" and rep > 0:
        # a different allocator state for every repetition (what a long test session has and a steady loop has not): blocks of random sizes taken from and
        # returned to the pools of the compute stream, now and then the whole cache dropped -- which memory a step's tensors land in changes from repetition
        # to repetition, and with it what a missing stream dependency would corrupt
        for _ in range(rng.randint(0, 12)):
            junk.append(torch.empty(int(rng.randint(1, 64)) * (1 << int(rng.randint(10, 22))), device="cuda", dtype=torch.float32).fill_(float("nan")))
        rng.shuffle(junk)
        del junk[:rng.randint(0, len(junk) + 1)]
        if rng.rand() < 0.15:
            junk.clear()
            torch.cuda.empty_cache()
    with torch.no_grad():
        for k, v in net.state_dict().items():
            v.copy_(start[k])
    net.zero_grad(set_to_none=True)
    np.random.seed(14)
    out = net(x)
    losses = crit(net, *out, inst, gtd)
    sum(losses.values()).sum().backward()
    ops.wgrad_join()
    torch.cuda.synchronize()
    snap = {n: p.grad.detach().clone() for n, p in params if p.grad is not None}
    if first is None:
        first = snap
        continue
    diff = [(int((v != first[k]).sum()), v.numel(), float((v.double() - first[k].double()).norm() / (first[k].double().norm() + 1e-300)), k) for k, v in snap.items() if not torch.equal(v, first[k])]
    if diff:
        bad += 1
        print("repetition %d: %d tensors differ: %s" % (rep, len(diff), sorted(diff, reverse=True)[:6]), flush=True)
print("%d of %d repetitions differ from the first (ARITH=%s WINOGRAD=%s ASYNC=%s PRN_WGRAD_FAST=%s)" % (bad, REPS - 1, os.environ.get("ARITH", "all-f16"), ops.WINOGRAD,
      os.environ.get("ASYNC", "1"), os.environ.get("PRN_WGRAD_FAST", "1")))
