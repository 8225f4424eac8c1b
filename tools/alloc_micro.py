#!/usr/bin/env python
"""Cost of a caching-allocator malloc while blocks freed after record_stream() wait for their events (GPU parked)."""
import time
import torch

dev = torch.device("cuda:0")
x = torch.empty(10 << 20, device=dev)


def bench(n=2000):
    t = time.perf_counter()
    for _ in range(n):
        y = torch.empty_like(x)
        del y
    return (time.perf_counter() - t) / n * 1e6


print("malloc+free 40 MB, idle GPU, no pending events: %.1f us" % bench())
streams = [torch.cuda.Stream() for _ in range(8)]
torch.cuda.synchronize()
torch.cuda._sleep(int(3 * 2.4e9))                   # park the GPU ~3 s
for k in (1, 4, 8):
    for st in streams[:k]:
        st.wait_stream(torch.cuda.current_stream())     # the side stream's work (and its events) sits behind the park
        for _ in range(20):
            t_ = torch.empty(1 << 20, device=dev)
            with torch.cuda.stream(st):
                t_.add_(1.0)                        # queued behind the park... on another stream: make it wait for main
            t_.record_stream(st)
            del t_
    print("with record_stream'ed frees pending on %d streams: %.1f us" % (k, bench(300)))
torch.cuda.synchronize()
print("after sync: %.1f us" % bench())
