"""Can two ranks share ONE GPU under RCCL?  (The GPU box has a single device; the bench contract's N > 1 path is otherwise only
exercised over gloo.)  Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533
tools/rccl_one_gpu_probe.py"""
import os

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl")
    t = torch.ones(1024, device="cuda") * (rank + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_reduce over RCCL on one shared GPU -> {t[0].item()}", flush=True)
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: RCCL refused: {type(e).__name__}: {str(e)[:300]}", flush=True)
