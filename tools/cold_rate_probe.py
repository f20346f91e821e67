"""Why does a ~100 MB streaming kernel run at 3 TB/s "cold" when a 1 GiB one reaches 5.9?  Same fp32 -> bf16 cast of 25216 x 768, timed after
(a) a 1 GiB WRITE pass (dirty lines in the Infinity Cache), (b) a 1 GiB READ-only pass (clean lines), (c) round-robin over 16 distinct
operand pairs (1.9 GB working set: always from HBM, nobody else's dirty lines), (d) the same buffer again (warm)."""
import torch

dev = "cuda"
T, D = 25216, 768
flush = torch.zeros(256 * 1024 * 1024, device=dev)
xs = [torch.randn(T, D, device=dev) for _ in range(16)]
ys = [torch.empty(T, D, device=dev, dtype=torch.bfloat16) for _ in range(16)]


def timed(pre, i=0, iters=20):
    ts = []
    for it in range(iters):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ys[i(it)].copy_(xs[i(it)]); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


byts = T * D * 6
for name, pre, idx in (("after a 1 GiB write pass", lambda: flush.add_(1.0), lambda it: 0),
                       ("after a 1 GiB read-only pass", lambda: flush.sum(), lambda it: 0),
                       ("round-robin over 16 buffers", lambda: None, lambda it: it % 16),
                       ("same buffer again (warm)", lambda: None, lambda it: 0)):
    t = timed(pre, idx)
    print(f"{name:32s}: {t:6.1f} us  {byts / t / 1e6:5.2f} TB/s")
