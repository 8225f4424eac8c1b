#!/usr/bin/env python
"""Host / device cost of a one-rank RCCL all_reduce (the forced-exchange probe of bench.py)."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.randn(25 << 18, device="cuda")            # 25 MB
side = torch.cuda.Stream()
for _ in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
works = []
for _ in range(20):
    with torch.cuda.stream(side):
        works.append(dist.all_reduce(x, async_op=True))
t1 = time.perf_counter()
for w in works:
    with torch.cuda.stream(side):
        w.wait()
t2 = time.perf_counter()
torch.cuda.synchronize()
t3 = time.perf_counter()
print("enqueue %.1f us/call, wait %.1f us/call, drain %.2f ms" % ((t1 - t0) / 20 * 1e6, (t2 - t1) / 20 * 1e6, (t3 - t2) * 1e3))
dist.destroy_process_group()
