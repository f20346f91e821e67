"""Diagnostic: which backward ops survive a stream capture (each in a fresh process: python tools/graph_op_probe.py OP)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa
from lightly_train_amd import ops
from lightly_train_amd.vit import split_k_plan, _split_k
op = sys.argv[1]
dev = "cuda"
T, D, hid, H, dh = 800, 384, 1536, 6, 64
bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
f = lambda *s: torch.randn(*s, device=dev)
scr = torch.empty(1 << 24, device=dev)
slab = torch.empty(32 * 1024 * 1024, device=dev)
def run():
    if op == "ln_bwd":
        ops.layernorm_bwd(f(T, D), f(D), f(T), f(T).abs() + 1, bf(T, D), f(T, D), f(T, D), f(D), f(D), T, D)
    elif op == "ln_bwd_fused":
        ops.layernorm_bwd(f(T, D), f(D), f(T), f(T).abs() + 1, bf(T, D), f(T, D), f(T, D), f(D), f(D), T, D, dnext=bf(T + 64, D)[:T], gamma_next=f(D),
                          rowscale_next=None, scale_next=1.0, dbias_next=f(D))
    elif op == "attn_bwd":
        for B, N in ((16, 50), (32, 10), (4, 197)):
            qkv = bf(B, N, 3 * H * dh); out = bf(B, N, H * dh); lse = f(B, H, N); dout = bf(B, N, H * dh)
            ws = torch.empty(ops.attention_bwd_ws_floats(B, N, H, dh), device=dev); dq = torch.empty_like(qkv)
            ops.attention_bwd(qkv, out, dout, lse, ws, dq, B, N, H, dh, dh ** -0.5)
    elif op == "wgrad":
        dy, x, gw, gb = bf(832, hid)[:T], bf(832, D)[:T], f(hid, D), f(hid)
        tiles = ((hid + 127) // 128) * ((D + 127) // 128)
        dyp = dy.as_strided((832, hid), (hid, 1)); xp = x.as_strided((832, D), (D, 1))
        dyp[T:].zero_(); xp[T:].zero_()
        ops.gemm(dyp, xp, gw, M=hid, N=D, K=832, trans_a=True, trans_b=True, epilogue=ops.EPI_F32_ACCUM, lda=hid, ldb=D, ldc=D, workspace=slab,
                 colsum=gb, **split_k_plan(hid, D, 832, True, _split_k(tiles, 832)))
    elif op == "dgrad_gelu":
        ops.gemm(bf(T, D), bf(D, hid), bf(T, hid), M=T, N=hid, K=D, trans_b=True, epilogue=ops.EPI_BF16_GELUGRAD, aux=bf(T, hid))
    elif op == "dgrad":
        ops.gemm(bf(T, hid), bf(hid, D), bf(T, D), M=T, N=D, K=hid, trans_b=True, epilogue=ops.EPI_BF16)
    elif op == "ledger":
        ops.reduce_begin(scr, 64)
        ops.layernorm_bwd(f(T, D), f(D), f(T), f(T).abs() + 1, bf(T, D), f(T, D), f(T, D), f(D), f(D), T, D)
        ops.reduce_flush()
torch.manual_seed(0)
# operands must outlive the graph: build them once per call inside run() would free them -- keep them alive through a list
keep = []
_bf, _f = bf, f
bf = lambda *s: (keep.append(_bf(*s)) or keep[-1])
f = lambda *s: (keep.append(_f(*s)) or keep[-1])
struct = sys.argv[2] if len(sys.argv) > 2 else "single"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
run(); run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cap = torch.cuda.current_stream()
    if struct == "single":
        run()
    elif struct == "forked":            # the op alone on a forked stream
        s2.wait_stream(cap)
        with torch.cuda.stream(s2): run()
        cap.wait_stream(s2)
    elif struct == "chain_side":        # chain on cap, the op on the side stream between two event edges
        s2.wait_stream(cap)
        keep[0].add_(1)
        s2.wait_event(cap.record_event())
        with torch.cuda.stream(s2):
            run()
            ev = s2.record_event()
        keep[1].add_(1)
        cap.wait_event(ev)
        keep[0].add_(1)
        cap.wait_stream(s2)
    elif struct == "two_chains_side":   # two chains (cap, s1) both feeding the side stream s2
        s1.wait_stream(cap); s2.wait_stream(cap)
        evs = []
        for st in (s1, cap, s1, cap):
            with torch.cuda.stream(st):
                keep[0 if st is cap else 1].add_(1)
                s2.wait_event(st.record_event())
            with torch.cuda.stream(s2):
                run()
                evs.append((st, s2.record_event()))
        for st, ev in evs:
            st.wait_event(ev)
            with torch.cuda.stream(st): keep[0 if st is cap else 1].add_(1)
        cap.wait_stream(s1); cap.wait_stream(s2)
g.replay(); torch.cuda.synchronize()
print("OK", op, struct)
