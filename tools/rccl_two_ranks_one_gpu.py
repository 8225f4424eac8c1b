"""Does RCCL accept two ranks on ONE device (the only way to run a >1-rank RCCL collective on a one-GPU lease)?
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/rccl_two_ranks_one_gpu.py"""
import os

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
x = torch.full((1 << 20,), float(rank + 1), device="cuda")
try:
    dist.all_reduce(x, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    print("rank", rank, "all_reduce AVG over 2 ranks on one device:", x[0].item(), flush=True)
except Exception as e:  # noqa: BLE001
    print("rank", rank, "RCCL refused:", type(e).__name__, str(e).splitlines()[0][:200], flush=True)
dist.destroy_process_group()
