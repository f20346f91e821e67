"""Does a LOW-priority HIP stream for the weight-gradient GEMMs (hipStreamCreateWithPriority, wrapped as a torch ExternalStream) help the
chains the step waits for?  Same-process A/B, the side stream swapped step by step.  Measured: no -- 95.7 -> 109.9 ms."""
import ctypes as C, os, sys, time, statistics, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args
from lightly_train_amd.vit import ViTConfig
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(), C.c_int()
hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
print("priority range least", lo.value, "greatest", hi.value, "torch", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
def mk(prio):
    st = C.c_void_p()
    assert hip.hipStreamCreateWithPriority(C.byref(st), 0, prio) == 0
    return torch.cuda.ExternalStream(st.value)
dev = torch.device("cuda", 0)
cfg = ViTConfig(patch_size=16, img_size=224, init_values=1e-5, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0)
m = DINOv2(cfg, DINOv2Args(output_dim=65536), global_batch_size=128, total_steps=125_000, device=dev, seed=0)
g = torch.Generator().manual_seed(1234)
views = [torch.randn(128, 3, 224, 224, generator=g).to(dev) for _ in range(2)] + [torch.randn(128, 3, 98, 98, generator=g).to(dev) for _ in range(8)]
variants = {"normal": m.side_stream, "low": mk(lo.value)}
random.seed(100)
for _ in range(3):
    for k, s in variants.items():
        m.side_stream = s; m.train_step(views)
torch.cuda.synchronize()
t = {k: [] for k in variants}
for i in range(32):
    k = list(variants)[i % 2]
    m.side_stream = variants[k]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.train_step(views)
    torch.cuda.synchronize(); t[k].append((time.perf_counter() - t0) * 1e3)
for k in variants:
    print(f"side stream {k:7s}: median {statistics.median(t[k]):.2f} ms  min {min(t[k]):.2f}  max {max(t[k]):.2f}")
