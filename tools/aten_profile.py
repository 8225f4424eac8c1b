#!/usr/bin/env python
"""torch.profiler view of one training step: which ATen ops (outside the HIP library) still launch kernels, and from where."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from planerecnet_amd import timer  # noqa: E402
from planerecnet_amd.config import cfg, set_cfg  # noqa: E402
from planerecnet_amd.losses import PlaneRecNetLoss, TargetPrefetcher  # noqa: E402
from planerecnet_amd.planerecnet import PlaneRecNet  # noqa: E402

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


def step():
    opt.zero_grad(set_to_none=True)
    t = pf.get(depths, dev)
    pf.submit(inst, (480, 640))
    out = net(images)
    losses = crit(net, *out, inst, depths, targets=t)
    sum(losses.values()).sum().backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=50, max_shapes_column_width=60))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=40))
print("==== callers of fill / zero / add / copy")
rows = [e for e in prof.key_averages(group_by_stack_n=10) if e.key in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::add", "aten::add_", "aten::copy_", "aten::zeros_like")]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    st = [l for l in e.stack if "planerecnet_amd" in l or "autograd" in l or "bench" in l or "tools" in l][:4]
    print(e.key, e.count, "%.1f us" % (e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total), " | ".join(s_.strip()[-90:] for s_ in st))
pf.close()
