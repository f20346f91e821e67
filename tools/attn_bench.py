"""Attention forward / backward timings on the step's shapes (ViT-B/16, per-GPU batch 128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd
from lightly_train_amd import ops

def bench(name, B, N, H=12, dh=64, iters=10):
    qkv = torch.randn(B, N, 3 * H * dh, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, N, H * dh, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, N, device="cuda")
    dout = torch.randn(B, N, H * dh, device="cuda").to(torch.bfloat16)
    ws = torch.empty(ops.attention_bwd_ws_floats(B, N, H, dh), device="cuda"); dqkv = torch.empty_like(qkv)
    res = {}
    for tag, fn in (("fwd", lambda: ops.attention_fwd(qkv, out, lse, B, N, H, dh, dh ** -0.5)),
                    ("bwd", lambda: ops.attention_bwd(qkv, out, dout, lse, ws, dqkv, B, N, H, dh, dh ** -0.5))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        res[tag] = e0.elapsed_time(e1) / iters * 1e3
    fl = 4.0 * B * H * N * N * dh
    print(f"{name:28s} B={B:5d} N={N:4d}: fwd {res['fwd']:7.1f} us ({fl / res['fwd'] / 1e6:6.1f} TF/s)   bwd {res['bwd']:7.1f} us ({2.5 * fl / res['bwd'] / 1e6:6.1f} TF/s)")

if len(sys.argv) > 1 and sys.argv[1] == "ab":   # python tools/attn_bench.py ab NAME v1 v2 ...: backward at the global-crop shape,
    # the per-call switch NAME alternating launch by launch in one process (medians of individually timed launches)
    import statistics
    name, vals = sys.argv[2], sys.argv[3:]
    B, N, H, dh = 256, 197, 12, 64
    qkv = torch.randn(B, N, 3 * H * dh, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, N, H * dh, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, N, device="cuda")
    dout = torch.randn(B, N, H * dh, device="cuda").to(torch.bfloat16)
    ws = torch.empty(ops.attention_bwd_ws_floats(B, N, H, dh), device="cuda"); dqkv = torch.empty_like(qkv)
    ops.attention_fwd(qkv, out, lse, B, N, H, dh, dh ** -0.5)
    t = {v: [] for v in vals}
    for i in range(25 * len(vals)):
        v = vals[i % len(vals)]
        os.environ[name] = v
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.attention_bwd(qkv, out, dout, lse, ws, dqkv, B, N, H, dh, dh ** -0.5)
        e1.record(); torch.cuda.synchronize()
        if i >= 5 * len(vals):
            t[v].append(e0.elapsed_time(e1) * 1e3)
    for v in vals:
        print(f"{name}={v}: median {statistics.median(t[v]):7.1f} us  min {min(t[v]):7.1f}")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "abf":   # python tools/attn_bench.py abf B N SPEC SPEC ...: FORWARD at one shape; a SPEC is a comma-separated list
    # of per-call switches (e.g. LT_ATTN_FWD_HPB=12,LT_ATTN_FWD_W=4; "-" = defaults) alternating launch by launch in one process; outputs compared
    # with the first SPEC's bit for bit
    import statistics
    B, N, specs = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4:]
    H, dh = 12, 64
    qkv = torch.randn(B, N, 3 * H * dh, device="cuda").to(torch.bfloat16)
    names = sorted({kv.split("=")[0] for sp in specs for kv in sp.split(",") if "=" in kv})
    def apply(sp):
        for n in names: os.environ.pop(n, None)
        for kv in sp.split(","):
            if "=" in kv: os.environ[kv.split("=")[0]] = kv.split("=")[1]
    outs = {}
    for sp in specs:
        apply(sp)
        out = torch.full((B, N, H * dh), float("nan"), device="cuda", dtype=torch.bfloat16); lse = torch.full((B, H, N), float("nan"), device="cuda")
        ops.attention_fwd(qkv, out, lse, B, N, H, dh, dh ** -0.5); torch.cuda.synchronize()
        outs[sp] = (out, lse)
    t = {sp: [] for sp in specs}
    out, lse = outs[specs[0]][0].clone(), outs[specs[0]][1].clone()
    for i in range(25 * len(specs)):
        sp = specs[i % len(specs)]
        apply(sp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.attention_fwd(qkv, out, lse, B, N, H, dh, dh ** -0.5)
        e1.record(); torch.cuda.synchronize()
        if i >= 5 * len(specs):
            t[sp].append(e0.elapsed_time(e1) * 1e3)
    for sp in specs:
        same = bool(torch.equal(outs[sp][0], outs[specs[0]][0]) and torch.equal(outs[sp][1], outs[specs[0]][1]))
        print(f"B={B} N={N} {sp:44s}: median {statistics.median(t[sp]):7.1f} us  min {min(t[sp]):7.1f}  bit-equal to the first: {same}")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "global":   # profiling runs: the global-crop shape only
    bench("global 224/16", 256, 197, iters=3)
    sys.exit(0)
bench("global 224/16", 256, 197)
bench("local 96/16", 1024, 37)
bench("local 98->112/16", 1024, 50)
bench("global 224/14", 256, 257)
bench("518/14 (ViT-L shape, H=16)", 8, 1370, H=16)
