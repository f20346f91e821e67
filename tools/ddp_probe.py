"""Probe (gloo, two ranks on one GPU): bucketed async all-reduce of a large CUDA buffer, as GradSync issues it."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd
from lightly_train_amd.parallel import GradSync
dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
n = int(sys.argv[1]); bucket = int(sys.argv[2]); mode = sys.argv[3]
g = torch.full((n,), float(rank + 1), device="cuda")
if mode == "async":
    s = GradSync(g, bucket_bytes=bucket); s.start(); s.finish()
else:
    for a in range(0, n, bucket // 4):
        dist.all_reduce(g[a:a + bucket // 4])
    g.mul_(0.5)
torch.cuda.synchronize()
print(rank, mode, n, bucket, float(g[0]), float(g[-1]), flush=True)
dist.destroy_process_group()
