"""CPU, world_size 2, gloo: the data-parallel gradient exchange (planerecnet_amd/parallel.py) that bench.py / train.py run
over RCCL on the GPUs.  Checks the bucketed, hook-driven all-reduce against the definition: grad = mean over ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bucket_bytes, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from planerecnet_amd.parallel import GradAllReduce, all_reduce_mean_scalars
        torch.manual_seed(0)                                 # identical replicas
        net = torch.nn.Sequential(torch.nn.Linear(13, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                                  torch.nn.Linear(32, 7), torch.nn.Linear(7, 3))
        net[5].weight.requires_grad_(False)                  # a frozen parameter (frozen BN affine in the real model)
        unused = torch.nn.Parameter(torch.ones(5))           # a parameter that receives no gradient on ANY rank
        half = torch.nn.Parameter(torch.ones(4))             # ... and one that receives a gradient on rank 1 only
        params = list(net.parameters()) + [unused, half]
        ex = GradAllReduce(params, bucket_bytes=bucket_bytes)
        results = []
        for step in range(2):                                # two steps: bucket counters must reset
            g = torch.Generator().manual_seed(100 * step + rank)
            x = torch.randn(5, 13, generator=g)
            for p in params:
                p.grad = None
            loss = net(x).square().mean() * (rank + 1)
            if rank == 1:
                loss = loss + (half * torch.arange(4.0)).sum()
            loss.backward()
            local = [None if p.grad is None else p.grad.clone() for p in params]
            ex.finish()
            # how many ranks produced a gradient, per parameter (what optim.FusedAdam uses to skip a tensor like a .grad of None)
            cnt = {p: float(ex.presence[ex.index[p]]) for p in ex.params}
            assert cnt[unused] == 0.0 and cnt[half] == 1.0 and all(cnt[p] == float(world) for p in ex.params if p is not unused and p is not half)
            assert torch.allclose(half.grad, torch.arange(4.0) / world)          # the mean over ranks, zeros entering for rank 0
            # reference: plain all-reduce of the local gradients
            for p, l in zip(params, local):
                if p is half:
                    continue
                if l is None:
                    assert p.grad is None or float(p.grad.abs().max()) == 0
                    continue
                ref = l.clone()
                dist.all_reduce(ref)
                ref /= world
                assert torch.allclose(p.grad, ref, rtol=1e-6, atol=1e-7), "step %d" % step
            m = all_reduce_mean_scalars([loss.detach()], torch.device("cpu"))
            results.append(float(m[0]))
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [1 << 30, 2048, 64])
def test_grad_allreduce_equals_mean_over_ranks(bucket_bytes):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    out = dict(q.get(timeout=5) for _ in range(world))
    assert out[0] == out[1]                                  # both ranks log the same mean loss


def test_world_size_one_is_a_no_op():
    from planerecnet_amd.parallel import GradAllReduce
    lin = torch.nn.Linear(3, 2)
    ex = GradAllReduce(list(lin.parameters()))
    lin(torch.ones(1, 3)).sum().backward()
    g = lin.weight.grad.clone()
    ex.finish()
    assert torch.equal(lin.weight.grad, g) and ex.buckets == []
