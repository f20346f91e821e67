"""What plain streaming kernels reach on this box: copy, scale (read + write), read-only sum -- the yardstick for the bandwidth-bound
kernels of the step (LayerNorm, column sums, optimizer)."""
import torch

dev = "cuda"


def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    xb = x.bfloat16()
    tc = t(lambda: y.copy_(x)); ts = t(lambda: x.sum()); tm = t(lambda: torch.mul(x, 1.5, out=y)); tcast = t(lambda: xb.copy_(x))
    print(f"{mb:5d} MiB: copy {2 * n * 4 / tc / 1e12:5.2f} TB/s   read-only sum {n * 4 / ts / 1e12:5.2f} TB/s   scale {2 * n * 4 / tm / 1e12:5.2f} TB/s   f32->bf16 {n * 6 / tcast / 1e12:5.2f} TB/s")
