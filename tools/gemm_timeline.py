"""Where a token GEMM launch spends its time, per workgroup: a diagnostic build of gemm.hip (-DLT_GEMM_TIMING, linked as
lightly-train_amd/lib/liblt_amd_timing.so) stamps the 100 MHz wall clock at kernel entry, after the pipeline fill, after the K loop,
after the epilogue's last store was issued / acknowledged and when every wave is through, plus the CU the workgroup ran on.

  python lightly-train_amd/build.py --timing && python tools/gemm_timeline.py [lib]        (one line per shape: medians in us)
"""
import ctypes as C, os, sys, statistics, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa
from lightly_train_amd import _lib, ops

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(_lib.LIB_PATH), "liblt_amd_timing.so")
lib = C.CDLL(os.path.abspath(path))
lib.lt_last_error.restype = C.c_char_p
lib.lt_attention_bwd_ws_floats.restype = C.c_int64; lib.lt_attention_bwd_ws_floats.argtypes = [C.c_int] * 4
lib.lt_batchnorm_ws_floats.restype = C.c_int64; lib.lt_batchnorm_ws_floats.argtypes = [C.c_int]
for name, argtypes in _lib.SIGNATURES.items():
    fn = getattr(lib, name); fn.argtypes = argtypes; fn.restype = C.c_int
lib.lt_debug_gemm_timing.argtypes = [C.c_void_p, C.c_int64, C.c_int]
_lib.load()
_lib._lib = lib

def case(name, M, N, K, epi, trans_b=False):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(K, N, device="cuda") if trans_b else torch.randn(N, K, device="cuda")).to(torch.bfloat16)
    f32 = epi in (ops.EPI_RESID, ops.EPI_F32)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    kw = {}
    if epi == ops.EPI_BF16_GELU: kw = dict(bias=torch.zeros(N, device="cuda"), out2=torch.empty_like(out))
    if epi == ops.EPI_BF16: kw = dict(bias=torch.zeros(N, device="cuda"))
    if epi == ops.EPI_RESID: kw = dict(bias=torch.zeros(N, device="cuda"), gamma=torch.ones(N, device="cuda"), resid=torch.randn(M, N, device="cuda"))
    if epi == ops.EPI_BF16_GELUGRAD: kw = dict(aux=torch.randn(M, N, device="cuda").to(torch.bfloat16))
    return name, (lambda: ops.gemm(a, b, out, M=M, N=N, K=K, epilogue=epi, trans_b=trans_b, **kw)), ((M + 255) // 256) * ((N + 255) // 256), 2.0 * M * N * K

G = 50432
cases = [case("fc1 gelu+pre", G, 3072, 768, ops.EPI_BF16_GELU), case("qkv bf16", G, 2304, 768, ops.EPI_BF16), case("proj resid", G, 768, 768, ops.EPI_RESID),
         case("fc2 resid", G, 768, 3072, ops.EPI_RESID), case("dfc2 gelugrad", G, 3072, 768, ops.EPI_BF16_GELUGRAD, True),
         case("dfc1 f32", G, 768, 3072, ops.EPI_F32, True),
         case("resid 32 WGs", 8192, 256, 768, ops.EPI_RESID), case("resid 96 WGs", 8192, 768, 768, ops.EPI_RESID),
         case("resid 256 WGs", 65536, 256, 768, ops.EPI_RESID), case("f32 32 WGs", 8192, 256, 768, ops.EPI_F32, True)]
if os.environ.get("LT_GEMM_STAGGER"):
    print("start delays (diagnostic build):", os.environ["LT_GEMM_STAGGER"])
buf = np.zeros(8 * 16384, dtype=np.uint64)
print("shape            WGs  kernel   fill   loop(/Ktile)   epi-issue  +ack  all-waves   gap-on-CU  CUs  rounds   sum/CU   TF/s")
for name, fn, nwg, flops in cases:
    fn(); fn(); torch.cuda.synchronize()
    lib.lt_debug_gemm_timing(None, 0, 1)
    fn()
    lib.lt_debug_gemm_timing(buf.ctypes.data, 8 * nwg, 0)
    t = buf[:8 * nwg].reshape(nwg, 8).astype(np.int64)
    us = lambda x: x / 100.0
    t0, t1, t2, t3, t6, t7 = (t[:, i] for i in (0, 1, 2, 3, 6, 7))
    cu_key = (t[:, 5] & 0xf) * 100000 + (t[:, 4] & 0xffff00) // 256          # xcc, (cu/sh/se) bits of HW_ID
    per_cu = collections.defaultdict(list)
    for i in np.argsort(t0):
        per_cu[int(cu_key[i])].append(i)
    gaps = [us(t0[b_] - t7[a_]) for v in per_cu.values() for a_, b_ in zip(v, v[1:])]
    busy = [sum(us(t7[i] - t0[i]) for i in v) for v in per_cu.values()]
    span = us(t7.max() - t0.min())
    K = 3072 if name in ("fc2 resid", "dfc1 f32") else 768
    med = lambda x: float(np.median(x))
    # how many CUs are inside their epilogue at the same time (0.1 us grid)
    T0 = int(t0.min()); n_ = int(t7.max()) - T0 + 1
    cnt = np.zeros(n_ * 1 + 2, dtype=np.int32)
    for a_, b_ in zip(t2 - T0, t7 - T0):
        cnt[a_] += 1; cnt[b_] -= 1
    inepi = np.cumsum(cnt)[:n_]
    conc = f"in-epilogue CUs: mean {inepi.mean():5.1f} max {inepi.max():3d}  time with >128: {100.0 * (inepi > 128).mean():4.1f}%  with <16: {100.0 * (inepi < 16).mean():4.1f}%"
    print(f"{name:14s} {nwg:5d} {span:7.1f} {med(us(t1 - t0)):6.2f} {med(us(t2 - t1)):7.2f} ({med(us(t2 - t1)) / (K / 64):5.2f}) "
          f"{med(us(t3 - t2)):9.2f} {med(us(t6 - t3)):6.2f} {med(us(t7 - t2)):8.2f} {med(gaps) if gaps else 0:10.2f} {len(per_cu):5d} {nwg / len(per_cu):6.2f} "
          f"{med(busy):8.1f} {flops / span / 1e6:6.0f}   {conc}")
