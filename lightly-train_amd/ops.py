"""torch-tensor front-ends of the C ABI (include/lt_amd.h).  torch is plumbing only: it owns the HBM
allocations and the HIP stream; every arithmetic op below runs in liblt_amd.so."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Any, Any, Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import (EPI_BF16, EPI_BF16_GELU, EPI_BF16_GELUGRAD, EPI_F32, EPI_F32_ACCUM, EPI_RESID, GemmDesc, check)

__all__ = ["EPI_BF16", "EPI_BF16_GELU", "EPI_RESID", "EPI_F32", "EPI_BF16_GELUGRAD", "EPI_F32_ACCUM"]


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the handle itself: no Stream object per launch (~1000 launches / step)


def _stream() -> int:
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda, "lightly_train_amd ops need device tensors (no CPU fallback)"
    return t.data_ptr()


def _chk(t: Tensor, dtype: torch.dtype, name: str) -> None:
    if t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
        raise ValueError(f"{name}: expected contiguous device tensor of {dtype}, got {t.dtype} contiguous={t.is_contiguous()} cuda={t.is_cuda}")


# ------------------------------------------------------------------------------------------ launch plans
# The static part of a step (transformer blocks without stochastic-depth draws: the same kernels on the same buffers every step) is ~600 of
# the step's ~930 calls across the C ABI, each preceded by Python that re-derives what it derived the step before (workspace look-ups,
# descriptor fields, shape checks): 11 ms of launch-thread time per 80-ms step.  A LaunchPlan is the list of those calls as they were made
# once -- the bound library function with its marshalled arguments, and the cross-stream event edges between them -- replayed by a loop
# that does nothing else.  Unlike a HIP graph (measured: the ROCm 7.0 executor serialises the backward's three branches, +7 ms,
# profiles/r05_graph_replay_ab.log) a replay issues the very same launches on the very same streams as the eager step: the device sees
# no difference, results are bitwise equal by construction, and the library's host-side state (reduction ledger) runs as it does eagerly.
_QUERY_CALLS = frozenset(("lt_last_error", "lt_abi_version", "lt_device_info", "lt_attention_bwd_ws_floats", "lt_batchnorm_ws_floats",
                          "lt_reduce_overflows"))
_EV_RECORD, _EV_WAIT = torch.cuda.Event.record, torch.cuda.Event.wait


class LaunchPlan:
    """ops: (0, library function, argument tuple) | (1, event, stream): record | (2, event, stream): wait | (3, callable, None)."""
    __slots__ = ("ops",)

    def __init__(self) -> None:
        self.ops: list = []

    def counts(self) -> Dict[str, int]:
        c = {"launches": 0, "event_records": 0, "event_waits": 0, "callables": 0}
        for k, _, _ in self.ops:
            c[("launches", "event_records", "event_waits", "callables")[k]] += 1
        return c

    def replay(self) -> None:
        # (events: the logged objects are recorded again -- a fresh event per record, as the eager step makes them, measured no better:
        # profiles/r06l_plan_ab.log)
        for k, a, b in self.ops:
            if k == 0:
                rc = a(*b)
                if rc:
                    check(rc, a.__name__)
            elif k == 1:
                _EV_RECORD(a, b)
            elif k == 2:
                _EV_WAIT(a, b)
            else:
                a()


class _RecordingLib:
    """What `_lib.load()` returns while a plan is recorded: forwards every call to the library and logs it."""

    def __init__(self, lib: Any, plan: LaunchPlan) -> None:
        self._lib, self._plan, self._wrapped = lib, plan, {}

    def __getattr__(self, name: str) -> Any:
        fn = getattr(self._lib, name)
        if name in _QUERY_CALLS:
            return fn
        w = self._wrapped.get(name)
        if w is None:
            plan = self._plan

            def w(*args: Any, _fn: Any = fn) -> int:
                rc = _fn(*args)
                plan.ops.append((0, _fn, args))    # (ctypes.byref arguments keep their structure alive; ops.gemm builds a fresh one per call)
                return rc
            self._wrapped[name] = w
        return w


_active_plan: Optional[LaunchPlan] = None
plan_replay_enabled = True    # process-wide switch (instrumented runs that wrap ops.* functions must see every call: bench.py's roofline leg)


class record_plan:
    """with ops.record_plan() as plan: the launches made inside run as usual AND are logged: calls across the C ABI through `_lib.load()`,
    event records / waits through torch.cuda.Event (Stream.record_event / wait_event / wait_stream end there), other device work through
    `ops.recordable`.  The caller guarantees that nothing inside depends on the step (same buffers, same sizes, same streams)."""
    _saved: Any = (None, None)    # the two torch.cuda.Event methods as they were when the recording began

    def __enter__(self) -> LaunchPlan:
        global _active_plan
        if _active_plan is not None:      # a recording that an exception cut short: drop it
            record_plan.__exit__(self)
        plan = LaunchPlan()

        def rec(ev: Any, stream: Any = None) -> None:
            stream = torch.cuda.current_stream() if stream is None else stream
            _EV_RECORD(ev, stream)
            plan.ops.append((1, ev, stream))

        def wait(ev: Any, stream: Any = None) -> None:
            stream = torch.cuda.current_stream() if stream is None else stream
            _EV_WAIT(ev, stream)
            plan.ops.append((2, ev, stream))

        real = _lib.load()
        record_plan._saved = (torch.cuda.Event.record, torch.cuda.Event.wait)
        torch.cuda.Event.record, torch.cuda.Event.wait = rec, wait
        _lib._recording = _RecordingLib(real, plan)
        _active_plan = plan
        return plan

    def __exit__(self, *exc: Any) -> None:
        global _active_plan
        torch.cuda.Event.record, torch.cuda.Event.wait = record_plan._saved
        _lib._recording = None
        _active_plan = None


def recordable(fn: Any) -> None:
    """Device work that does not cross the C ABI (a torch fill of pad rows): run it, and log it when a plan is being recorded.  `fn` must
    capture the stream it runs on itself if that is not the current stream at replay."""
    fn()
    if _active_plan is not None:
        st = torch.cuda.current_stream() if torch.cuda.is_available() else None

        def on_stream(fn_: Any = fn, st_: Any = st) -> None:
            if st_ is None:
                fn_()
                return
            with torch.cuda.stream(st_):
                fn_()
        _active_plan.ops.append((3, on_stream, None))


def require_device(dev: torch.device, who: str) -> None:
    """The methods run on an MI355X only: there is no CPU path behind these wrappers."""
    if torch.device(dev).type != "cuda":
        raise RuntimeError(f"{who} runs on an MI355X only (no CPU fallback for the HIP kernels)")


class _PinnedRing:
    """Persistent pinned staging slots for `h2d`, handed out round-robin; a slot is taken again only after the event recorded behind its
    last copy has fired.  `Tensor.pin_memory()` costs the launch thread 0.6-2 ms per call once the host runs a few steps ahead of the
    device (every cached pinned block is still in flight, so each call is a fresh hipHostMalloc: tools/host_profile.py,
    profiles/r05_host_profile_dp02.log); a slot here is one memcpy into memory pinned once.  The ring is ONE pinned allocation made at
    its first use (slot size = the first request rounded up to a power of two, at least `min_bytes`): pinning slot by slot as the ring
    index advances spread hundreds of hipHostMalloc calls over the first steps of a run.  A request larger than the slot size gets a
    slot of its own (grown on demand)."""

    def __init__(self, slots: int, min_bytes: int) -> None:
        self.n = slots
        self.bufs: list = [None] * slots
        self.evs: list = [None] * slots
        self.i = 0
        self.min_bytes = min_bytes
        self.slot_bytes = 0

    def stage(self, t: Tensor):
        n = t.numel() * t.element_size()
        if self.slot_bytes == 0:
            sb = self.min_bytes
            while sb < n:
                sb *= 2
            arena = torch.empty(self.n * sb, dtype=torch.uint8, pin_memory=True)
            self.bufs = [arena[k * sb:(k + 1) * sb] for k in range(self.n)]
            self.slot_bytes = sb
        i = self.i
        self.i = (i + 1) % self.n
        if self.evs[i] is not None:
            self.evs[i].synchronize()      # (never waits in practice: the ring is deeper than the host's lead over the device)
        b = self.bufs[i]
        if b.numel() < n:
            b = torch.empty(n, dtype=torch.uint8, pin_memory=True)
            self.bufs[i] = b
        v = b[:n].view(t.dtype).view(t.shape)
        v.copy_(t)
        return v, i


_rings = {"small": _PinnedRing(512, 65536), "large": _PinnedRing(32, 1 << 20)}


def h2d(t: Tensor, dev: torch.device) -> Tensor:
    """A small host tensor to the device WITHOUT stalling the launch thread: through a pinned staging slot (a pageable source makes
    hipMemcpyAsync wait on its staging path -- 7.5 ms per copy once the host runs a few steps ahead of the device:
    tools/host_ahead_probe.py -- while a pinned source is a queued DMA; the slot is recycled only after the copy's event)."""
    if t.device.type == "cpu" and torch.device(dev).type == "cuda":
        if t.numel() == 0:
            return torch.empty(t.shape, dtype=t.dtype, device=dev)
        ring = _rings["small" if t.numel() * t.element_size() <= 65536 else "large"]
        v, i = ring.stage(t.contiguous())
        out = v.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))    # the stream of the device the copy was queued on (not the current device's)
        ring.evs[i] = ev
        return out
    return t.to(dev, non_blocking=True)


def device_info() -> dict:
    lib = _lib.load()
    name = C.create_string_buffer(128)
    cu, clk = C.c_int(0), C.c_int(0)
    check(lib.lt_device_info(name, 128, C.byref(cu), C.byref(clk)), "lt_device_info")
    return {"name": name.value.decode(), "compute_units": cu.value, "clock_khz": clk.value}


# ------------------------------------------------------------------------------------------ GEMM
def gemm(a: Tensor, b: Tensor, out: Tensor, *, M: int, N: int, K: int, trans_a: bool = False, trans_b: bool = False,
         epilogue: int = EPI_BF16, bias: Optional[Tensor] = None, gamma: Optional[Tensor] = None,
         resid: Optional[Tensor] = None, out2: Optional[Tensor] = None, aux: Optional[Tensor] = None,
         alpha: float = 1.0, split_k: int = 1, lda: Optional[int] = None, ldb: Optional[int] = None,
         ldc: Optional[int] = None, force_kernel: int = 0, workspace: Optional[Tensor] = None,
         rowscale: Optional[Tensor] = None, branch_scale: float = 1.0, batch: int = 1, stride_a: int = 0, stride_b: int = 0,
         stride_c: int = 0, colsum: Optional[Tensor] = None, ln: Optional[Dict[str, Any]] = None) -> Tensor:
    """out[M,N] = op(a) @ op(b)^T-like contraction, see lt_gemm_bf16 in include/lt_amd.h.  batch > 1: that many independent
    problems, operands `stride_*` elements apart (plain epilogues of the 128x128 kernel).  colsum (weight gradients, trans_a): f32 [M]
    += the column sums of the stored A = dY, i.e. the bias gradient of the same Linear, taken from the fragments the kernel holds."""
    _chk(a, torch.bfloat16, "gemm.a")
    _chk(b, torch.bfloat16, "gemm.b")
    d = GemmDesc()
    d.A, d.B, d.M, d.N, d.K = _p(a), _p(b), M, N, K
    d.lda = lda if lda is not None else (M if trans_a else K)
    d.ldb = ldb if ldb is not None else (N if trans_b else K)
    d.trans_a, d.trans_b, d.epilogue = int(trans_a), int(trans_b), epilogue
    want = torch.float32 if epilogue in (EPI_RESID, EPI_F32, EPI_F32_ACCUM) else torch.bfloat16
    _chk(out, want, "gemm.out")
    d.C, d.ldc = _p(out), (ldc if ldc is not None else N)
    if out2 is not None:
        _chk(out2, torch.bfloat16, "gemm.out2")
    d.C2, d.ldc2 = _p(out2), N
    for t, nm in ((bias, "bias"), (gamma, "gamma"), (resid, "resid")):
        if t is not None:
            _chk(t, torch.float32, "gemm." + nm)
    d.bias, d.gamma, d.resid, d.ldr = _p(bias), _p(gamma), _p(resid), N
    if aux is not None:
        _chk(aux, torch.bfloat16, "gemm.aux")
    d.aux, d.ldaux = _p(aux), N
    d.alpha, d.split_k, d.force_kernel = alpha, split_k, force_kernel
    d.rowscale, d.branch_scale = _p(rowscale), branch_scale
    d.workspace = _p(workspace)
    d.workspace_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    d.batch, d.stride_a, d.stride_b, d.stride_c = batch, stride_a, stride_b, stride_c
    if colsum is not None:
        _chk(colsum, torch.float32, "gemm.colsum")
    d.colsum = _p(colsum)
    if ln is not None:   # EPI_RESID: the next LayerNorm behind the GEMM -- dict(weight, bias, out bf16 [M, N], mean, rstd, eps); see lt_gemm_desc
        _chk(ln["out"], torch.bfloat16, "gemm.ln.out")
        d.ln_weight, d.ln_bias, d.ln_out, d.ln_mean, d.ln_rstd, d.ln_eps = _p(ln["weight"]), _p(ln["bias"]), _p(ln["out"]), _p(ln.get("mean")), _p(ln.get("rstd")), ln["eps"]
    check(_lib.load().lt_gemm_bf16(C.byref(d), _stream()), "lt_gemm_bf16")
    return out


def gemm_naive(a: Tensor, b: Tensor, M: int, N: int, K: int, trans_a: bool = False, trans_b: bool = False) -> Tensor:
    out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    check(_lib.load().lt_gemm_bf16_naive(_p(a), _p(b), _p(out), M, N, K, M if trans_a else K, N if trans_b else K, N,
                                         int(trans_a), int(trans_b), _stream()), "lt_gemm_bf16_naive")
    return out


def matmul_f32(a: Tensor, b: Tensor, out: Tensor, M: int, N: int, K: int, trans_a: bool = False, accumulate: bool = False) -> Tensor:
    for t in (a, b, out):
        _chk(t, torch.float32, "matmul_f32")
    check(_lib.load().lt_matmul_f32(_p(a), _p(b), _p(out), M, N, K, int(trans_a), int(accumulate), _stream()), "lt_matmul_f32")
    return out


# ------------------------------------------------------------------------------------------ tokens
def bicubic_taps(n_in: int, n_out: int):
    """4-tap (index, weight) table of torch's 1-D bicubic resize (align_corners=False), read off F.interpolate itself by
    pushing an identity basis through it (host, once per size pair) so the arithmetic is the reference's by construction."""
    import torch.nn.functional as F

    basis = torch.eye(n_in).view(1, 1, n_in, n_in)                        # rows = source index, cols = basis id
    m = F.interpolate(basis, size=(n_out, n_in), mode="bicubic", align_corners=False)[0, 0]  # [n_out, n_in]
    idx = torch.zeros(n_out, 4, dtype=torch.int32)
    wts = torch.zeros(n_out, 4, dtype=torch.float32)
    for o in range(n_out):
        nz = m[o].nonzero().flatten()
        assert nz.numel() <= 4, "bicubic row with more than 4 taps"
        idx[o, : nz.numel()] = nz.to(torch.int32)
        wts[o, : nz.numel()] = m[o, nz]
    return idx, wts


def resize_4tap(img: Tensor, iy: Tensor, wy: Tensor, ix: Tensor, wx: Tensor, Ho: int, Wo: int) -> Tensor:
    _chk(img, torch.float32, "resize.img")
    B, Cc, H, W = img.shape
    out = torch.empty(B, Cc, Ho, Wo, device=img.device, dtype=torch.float32)
    check(_lib.load().lt_resize_4tap(_p(img), _p(out), _p(iy), _p(wy), _p(ix), _p(wx), B * Cc, H, W, Ho, Wo, _stream()), "lt_resize_4tap")
    return out


def im2col(img: Tensor, p: int, kpad: int) -> Tensor:
    _chk(img, torch.float32, "im2col.img")
    B, Cc, H, W = img.shape
    cols = torch.empty(B * (H // p) * (W // p), kpad, device=img.device, dtype=torch.bfloat16)
    check(_lib.load().lt_im2col_bf16(_p(img), _p(cols), B, Cc, H, W, p, kpad, _stream()), "lt_im2col_bf16")
    return cols


def assemble_tokens(patch: Tensor, cls: Tensor, pos: Tensor, mask_token: Tensor, masks: Optional[Tensor], B: int, n_p: int,
                    D: int, out: Optional[Tensor] = None, reg: Optional[Tensor] = None, n_reg: int = 0) -> Tensor:
    x = out if out is not None else torch.empty(B, n_p + 1 + n_reg, D, device=patch.device, dtype=torch.float32)
    if masks is not None:
        _chk(masks, torch.uint8, "assemble.masks")
    check(_lib.load().lt_assemble_tokens(_p(patch), _p(cls), _p(pos), _p(mask_token), _p(masks), _p(reg), _p(x), B, n_p, n_reg, D,
                                         _stream()), "lt_assemble_tokens")
    return x


def assemble_tokens_bwd(dx: Tensor, masks: Optional[Tensor], dpatch: Tensor, dcls: Tensor, dpos: Tensor, dmask: Tensor, B: int,
                        n_p: int, D: int, dreg: Optional[Tensor] = None, n_reg: int = 0) -> None:
    check(_lib.load().lt_assemble_tokens_bwd(_p(dx), _p(masks), _p(dpatch), _p(dcls), _p(dpos), _p(dmask), _p(dreg), B, n_p, n_reg, D,
                                             _stream()), "lt_assemble_tokens_bwd")


# ------------------------------------------------------------------------------------------ norms
def layernorm_fwd(x: Tensor, w: Tensor, b: Tensor, rows: int, D: int, y_bf16: Optional[Tensor] = None,
                  y_f32: Optional[Tensor] = None, mean: Optional[Tensor] = None, rstd: Optional[Tensor] = None,
                  eps: float = 1e-6) -> None:
    _chk(x, torch.float32, "layernorm.x")
    check(_lib.load().lt_layernorm_fwd(_p(x), _p(w), _p(b), _p(y_bf16), _p(y_f32), _p(mean), _p(rstd), rows, D, eps, _stream()),
          "lt_layernorm_fwd")


def layernorm_bwd(x: Tensor, w: Tensor, mean: Tensor, rstd: Tensor, dy: Tensor, dres: Optional[Tensor], dx: Tensor, dw: Tensor,
                  db: Tensor, rows: int, D: int, ws: Optional[Tensor] = None, dnext: Optional[Tensor] = None,
                  gamma_next: Optional[Tensor] = None, rowscale_next: Optional[Tensor] = None, scale_next: float = 1.0,
                  dbias_next: Optional[Tensor] = None, ridx: Optional[Tensor] = None) -> None:
    """dx = dres + LN'(dy).  With `dnext` (bf16 [rows, D]) the kernel also emits the next branch's upstream gradient
    dx * gamma_next * scale_next * rowscale_next and adds its column sums to `dbias_next`.
    `ridx` (int64, device, no repeats): row r of (x, mean, rstd, dy) belongs to row ridx[r] of the gradient stream:
    dx[ridx[r]] = dres[ridx[r]] + LN'(dy[r]) (lt_layernorm_bwd_rows; dres may be dx)."""
    if ridx is not None:
        assert dnext is None and ws is None
        _chk(ridx, torch.int64, "layernorm_bwd.ridx")
        check(_lib.load().lt_layernorm_bwd_rows(_p(x), _p(w), _p(mean), _p(rstd), _p(dy), int(dy.dtype == torch.float32), _p(dres), _p(dx), _p(ridx),
                                                _p(dw), _p(db), rows, D, _stream()), "lt_layernorm_bwd_rows")
        return
    if dnext is not None:
        _chk(dnext, torch.bfloat16, "layernorm_bwd.dnext")
    check(_lib.load().lt_layernorm_bwd_fused(_p(x), _p(w), _p(mean), _p(rstd), _p(dy), int(dy.dtype == torch.float32), _p(dres), _p(dx),
                                             _p(dw), _p(db), _p(ws), ws.numel() if ws is not None else 0, _p(dnext), _p(gamma_next),
                                             _p(rowscale_next), scale_next, _p(dbias_next), rows, D, _stream()), "lt_layernorm_bwd")


def layerscale_dgamma(w_bf16: Tensor, dw: Tensor, bias: Optional[Tensor], dbias: Optional[Tensor], gamma: Tensor, dgamma: Tensor,
                      N: int, K: int) -> None:
    """dgamma += (rowdot(W, dW) + bias * dbias) / gamma  -- the LayerScale gradient from the weight gradient."""
    _chk(w_bf16, torch.bfloat16, "layerscale_dgamma.w")
    check(_lib.load().lt_layerscale_dgamma(_p(w_bf16), _p(dw), _p(bias), _p(dbias), _p(gamma), _p(dgamma), N, K, _stream()),
          "lt_layerscale_dgamma")


def layerscale_dgamma_batched(w_bf16: Tensor, dw: Tensor, bias: Optional[Tensor], dbias: Optional[Tensor], gamma: Tensor, dgamma: Tensor,
                              N: int, K: int, batch: int, stride: int) -> None:
    """`layerscale_dgamma` for `batch` layers in one launch; the arguments are layer 0's tensors, layer i's lie i * stride elements on."""
    _chk(w_bf16, torch.bfloat16, "layerscale_dgamma.w")
    check(_lib.load().lt_layerscale_dgamma_batched(_p(w_bf16), _p(dw), _p(bias), _p(dbias), _p(gamma), _p(dgamma), N, K, batch, stride, _stream()),
          "lt_layerscale_dgamma_batched")


def layerscale_bwd(dout: Tensor, y: Optional[Tensor], gamma: Optional[Tensor], dy: Tensor, dgamma: Optional[Tensor], rows: int,
                   D: int, dbias: Optional[Tensor] = None, rowscale: Optional[Tensor] = None, scale: float = 1.0,
                   ridx: Optional[Tensor] = None) -> None:
    """`ridx` (int64, device): row r of the branch takes its upstream gradient from row ridx[r] of `dout` (lt_layerscale_bwd_rows)."""
    if ridx is not None:
        _chk(ridx, torch.int64, "layerscale_bwd.ridx")
    check(_lib.load().lt_layerscale_bwd_rows(_p(dout), _p(ridx), _p(y), _p(gamma), _p(dy), _p(dgamma), _p(dbias), _p(rowscale), scale, rows, D,
                                             _stream()), "lt_layerscale_bwd_rows")


def colsum_bf16(x: Tensor, out: Tensor, rows: int, N: int) -> None:
    check(_lib.load().lt_colsum_bf16(_p(x), _p(out), rows, N, _stream()), "lt_colsum_bf16")


def colsum_f32(x: Tensor, out: Tensor, rows: int, N: int, accumulate: bool = False) -> None:
    check(_lib.load().lt_colsum_f32(_p(x), _p(out), rows, N, int(accumulate), _stream()), "lt_colsum_f32")


def gather_rows(src: Tensor, ld: int, idx: Tensor, M: int, D: int, out_bf16: Optional[Tensor] = None,
                out_f32: Optional[Tensor] = None) -> None:
    _chk(idx, torch.int64, "gather_rows.idx")
    check(_lib.load().lt_gather_rows(_p(src), ld, _p(idx), _p(out_bf16), _p(out_f32), M, D, _stream()), "lt_gather_rows")


def scatter_add_rows(src: Tensor, idx: Tensor, dst: Tensor, ld: int, M: int, D: int) -> None:
    _chk(idx, torch.int64, "scatter_add_rows.idx")
    check(_lib.load().lt_scatter_add_rows(_p(src), _p(idx), _p(dst), ld, M, D, _stream()), "lt_scatter_add_rows")


def cast_bf16(src: Tensor, dst: Tensor) -> None:
    check(_lib.load().lt_cast_f32_to_bf16(_p(src), _p(dst), src.numel(), _stream()), "lt_cast_f32_to_bf16")


def kl_fwd_bwd(s_logits: Tensor, t_logits: Tensor, ld: int, inv_temp: float, coef: float, loss: Tensor, dlogits: Optional[Tensor], ldd: int,
               rows: int, K: int) -> None:
    """loss[0] += coef * sum_rows KL(softmax(t/T) || softmax(s/T)); dlogits (bf16) = its gradient w.r.t. the student logits."""
    _chk(s_logits, torch.float32, "kl.s"); _chk(t_logits, torch.float32, "kl.t"); _chk(loss, torch.float32, "kl.loss")
    check(_lib.load().lt_kl_fwd_bwd(_p(s_logits), _p(t_logits), ld, inv_temp, coef, _p(loss), _p(dlogits), ldd, rows, K, _stream()), "lt_kl_fwd_bwd")


def symmetrize_bf16(d: Tensor, g: Tensor, batch: int, n: int, ld: int) -> None:
    _chk(d, torch.bfloat16, "symmetrize.d"); _chk(g, torch.bfloat16, "symmetrize.g")
    check(_lib.load().lt_symmetrize_bf16(_p(d), _p(g), batch, n, ld, _stream()), "lt_symmetrize_bf16")


def mixup(x: Tensor, index: Tensor, lam: float, out: Tensor) -> None:
    _chk(x, torch.float32, "mixup.x"); _chk(index, torch.int64, "mixup.index"); _chk(out, torch.float32, "mixup.out")
    check(_lib.load().lt_mixup(_p(x), _p(index), lam, _p(out), x.shape[0], x[0].numel(), _stream()), "lt_mixup")


def resample_tables(hs: int, ws_: int, ht: int, wt: int, mode: str = "bilinear"):
    """Sparse tap tables (forward and transposed) of F.interpolate((hs, ws) -> (ht, wt), mode, align_corners=False), read off
    F.interpolate itself on an identity basis (host, once per size pair)."""
    import torch.nn.functional as F

    n_s, n_t = hs * ws_, ht * wt
    basis = torch.eye(n_s).view(1, n_s, hs, ws_)
    R = F.interpolate(basis, size=(ht, wt), mode=mode, align_corners=False)[0].reshape(n_s, n_t).t().contiguous()   # [n_t, n_s]

    def sparse(M: Tensor):
        taps = max(int((M != 0).sum(1).max()), 1)
        idx = torch.zeros(M.shape[0], taps, dtype=torch.int32)
        wts = torch.zeros(M.shape[0], taps, dtype=torch.float32)
        for r in range(M.shape[0]):
            nz = M[r].nonzero().flatten()
            idx[r, : nz.numel()] = nz.to(torch.int32)
            wts[r, : nz.numel()] = M[r, nz]
        return idx, wts, taps

    return sparse(R), sparse(R.t().contiguous())


def resample_tokens(x: Tensor, idx: Tensor, w: Tensor, out: Tensor, B: int, n_in: int, n_out: int, D: int, taps: int) -> None:
    _chk(x, torch.float32, "resample.x"); _chk(idx, torch.int32, "resample.idx"); _chk(w, torch.float32, "resample.w")
    check(_lib.load().lt_resample_tokens(_p(x), _p(idx), _p(w), _p(out), B, n_in, n_out, D, taps, _stream()), "lt_resample_tokens")


def roi_resample_tokens(x: Tensor, src_image: Optional[Tensor], idx: Tensor, w: Tensor, B: int, img_stride: int, n_out: int, D: int,
                        out_bf16: Optional[Tensor] = None, out_f32: Optional[Tensor] = None) -> None:
    """Bilinear RoI resampling of token maps with per-image 4-tap tables (lt_roi_resample_tokens).  `x`: f32 view starting at the first
    patch token of image 0, images `img_stride` elements apart."""
    _chk(idx, torch.int32, "roi.idx"); _chk(w, torch.float32, "roi.w")
    assert x.dtype == torch.float32 and x.is_cuda
    check(_lib.load().lt_roi_resample_tokens(x.data_ptr(), _p(src_image), _p(idx), _p(w), _p(out_bf16), _p(out_f32), B, img_stride, n_out, D, _stream()),
          "lt_roi_resample_tokens")


def roi_resample_tokens_bwd(dout: Tensor, idx: Tensor, w: Tensor, din: Tensor, B: int, img_stride: int, n_in: int, n_out: int, D: int) -> None:
    """Gather-form backward of `roi_resample_tokens` into `din` (f32 view starting at the first patch token of image 0)."""
    _chk(dout, torch.float32, "roi_bwd.dout"); _chk(idx, torch.int32, "roi_bwd.idx"); _chk(w, torch.float32, "roi_bwd.w")
    assert din.dtype == torch.float32 and din.is_cuda
    check(_lib.load().lt_roi_resample_tokens_bwd(_p(dout), _p(idx), _p(w), din.data_ptr(), B, img_stride, n_in, n_out, D, _stream()),
          "lt_roi_resample_tokens_bwd")


def center_tokens(z: Tensor, B: int, n: int, C: int, out_bf16: Optional[Tensor] = None, out_f32: Optional[Tensor] = None) -> None:
    _chk(z, torch.float32, "center.z")
    check(_lib.load().lt_center_tokens(_p(z), _p(out_bf16), _p(out_f32), B, n, C, _stream()), "lt_center_tokens")


def cka_fwd_bwd(Ks: Tensor, Kt: Tensor, coef: Tensor, loss: Tensor, G: Optional[Tensor], B: int, n: int, ld: int, eps: float = 1e-8) -> None:
    """loss[0] += sum_b coef[b] * (1 - CKA(Ks[b], Kt[b])), G[b] (bf16) = its gradient with respect to Ks[b] (lt_cka_fwd_bwd)."""
    _chk(Ks, torch.float32, "cka.Ks"); _chk(Kt, torch.float32, "cka.Kt"); _chk(coef, torch.float32, "cka.coef")
    check(_lib.load().lt_cka_fwd_bwd(_p(Ks), _p(Kt), _p(coef), _p(loss), _p(G), B, n, ld, eps, _stream()), "lt_cka_fwd_bwd")


def rope_apply(qkv: Tensor, sin_t: Tensor, cos_t: Tensor, B: int, N: int, H: int, dh: int, prefix: int, inverse: bool = False) -> None:
    """DINOv3 rotary embedding, in place on q and k of the packed bf16 qkv activation (tokens >= prefix)."""
    _chk(qkv, torch.bfloat16, "rope.qkv"); _chk(sin_t, torch.float32, "rope.sin"); _chk(cos_t, torch.float32, "rope.cos")
    check(_lib.load().lt_rope_apply(_p(qkv), _p(sin_t), _p(cos_t), B, N, H, dh, prefix, int(inverse), _stream()), "lt_rope_apply")


def gelu_fwd(x: Tensor, y: Tensor, n: int) -> Tensor:
    """y[:n] = gelu(x[:n]) on flat bf16 storage (erf form, nn.GELU())."""
    _chk(x, torch.bfloat16, "gelu_fwd.x")
    _chk(y, torch.bfloat16, "gelu_fwd.y")
    check(_lib.load().lt_gelu_fwd_bf16(_p(x), _p(y), n, _stream()), "lt_gelu_fwd_bf16")
    return y


def gelu_bwd(dy: Tensor, x: Tensor, dx: Tensor, n: int) -> Tensor:
    """dx[:n] = dy[:n] * gelu'(x[:n]), x the saved pre-activation."""
    _chk(dy, torch.bfloat16, "gelu_bwd.dy")
    _chk(x, torch.bfloat16, "gelu_bwd.x")
    _chk(dx, torch.bfloat16, "gelu_bwd.dx")
    check(_lib.load().lt_gelu_bwd_bf16(_p(dy), _p(x), _p(dx), n, _stream()), "lt_gelu_bwd_bf16")
    return dx


def swiglu_fwd(x12: Tensor, out: Tensor, rows: int, H: int) -> None:
    """out[rows, H] = silu(x12[:, :H]) * x12[:, H:]  (bf16, reference swiglu_ffn.py:31-35)."""
    _chk(x12, torch.bfloat16, "swiglu.x12"); _chk(out, torch.bfloat16, "swiglu.out")
    check(_lib.load().lt_swiglu_fwd(_p(x12), _p(out), rows, H, _stream()), "lt_swiglu_fwd")


def swiglu_bwd(x12: Tensor, dh: Tensor, d12: Tensor, rows: int, H: int) -> None:
    _chk(x12, torch.bfloat16, "swiglu.x12"); _chk(dh, torch.bfloat16, "swiglu.dh"); _chk(d12, torch.bfloat16, "swiglu.d12")
    check(_lib.load().lt_swiglu_bwd(_p(x12), _p(dh), _p(d12), rows, H, _stream()), "lt_swiglu_bwd")


def cast_pad_rows(src: Tensor, dst: Tensor, R: int, Cc: int, Cpad: int) -> None:
    check(_lib.load().lt_cast_pad_rows(_p(src), _p(dst), R, Cc, Cpad, _stream()), "lt_cast_pad_rows")


def unpad_accumulate(src: Tensor, dst: Tensor, R: int, Cc: int, Cpad: int) -> None:
    check(_lib.load().lt_unpad_accumulate(_p(src), _p(dst), R, Cc, Cpad, _stream()), "lt_unpad_accumulate")


def scale_f32(dst: Tensor, alpha: float) -> None:
    check(_lib.load().lt_scale_f32(_p(dst), alpha, dst.numel(), _stream()), "lt_scale_f32")


def fill_f32(dst: Tensor, value: float) -> None:
    check(_lib.load().lt_fill_f32(_p(dst), value, dst.numel(), _stream()), "lt_fill_f32")


# ------------------------------------------------------------------------------------------ attention
def attention_fwd(qkv: Tensor, out: Tensor, lse: Tensor, B: int, N: int, H: int, dh: int, scale: float) -> None:
    _chk(qkv, torch.bfloat16, "attention.qkv")
    check(_lib.load().lt_attention_fwd(_p(qkv), _p(out), _p(lse), B, N, H, dh, scale, _stream()), "lt_attention_fwd")


def attention_bwd_ws_floats(B: int, N: int, H: int, dh: int) -> int:
    return int(_lib.load().lt_attention_bwd_ws_floats(B, N, H, dh))


def attention_bwd(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, ws: Tensor, dqkv: Tensor, B: int, N: int, H: int, dh: int,
                  scale: float) -> None:
    assert ws.numel() >= attention_bwd_ws_floats(B, N, H, dh)
    check(_lib.load().lt_attention_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(ws), _p(dqkv), B, N, H, dh, scale, _stream()),
          "lt_attention_bwd")


# ------------------------------------------------------------------------------------------ head pieces
def l2norm_fwd(x: Tensor, y: Tensor, inv: Tensor, rows: int, D: int, eps: float = 1e-12) -> None:
    check(_lib.load().lt_l2norm_fwd(_p(x), _p(y), _p(inv), rows, D, eps, _stream()), "lt_l2norm_fwd")


def l2norm_bwd(dy: Tensor, x: Tensor, inv: Tensor, dx: Tensor, rows: int, D: int) -> None:
    check(_lib.load().lt_l2norm_bwd(_p(dy), _p(x), _p(inv), _p(dx), rows, D, _stream()), "lt_l2norm_bwd")


def weightnorm_fwd(v: Tensor, g: Tensor, w: Tensor, K: int, D: int) -> None:
    check(_lib.load().lt_weightnorm_fwd(_p(v), _p(g), _p(w), K, D, _stream()), "lt_weightnorm_fwd")


def weightnorm_bwd(dw: Tensor, v: Tensor, g: Tensor, dv: Tensor, dg: Tensor, K: int, D: int) -> None:
    check(_lib.load().lt_weightnorm_bwd(_p(dw), _p(v), _p(g), _p(dv), _p(dg), K, D, _stream()), "lt_weightnorm_bwd")


# ------------------------------------------------------------------------------------------ losses
def softmax_center(logits: Tensor, center: Optional[Tensor], probs: Tensor, rows: int, K: int, inv_temp: float) -> None:
    check(_lib.load().lt_softmax_center(_p(logits), _p(center), _p(probs), rows, K, inv_temp, _stream()), "lt_softmax_center")


def center_ema(center: Tensor, colsum: Tensor, scale: float, momentum: float, K: int) -> None:
    check(_lib.load().lt_center_ema(_p(center), _p(colsum), scale, momentum, K, _stream()), "lt_center_ema")


def ce_fwd_bwd(s: Tensor, teacher: Tensor, ta: Tensor, tb: Optional[Tensor], row_weight: Optional[Tensor], scale: float,
               inv_temp: float, loss: Tensor, dlogits: Optional[Tensor], rows: int, K: int, slot: Optional[Tensor] = None) -> None:
    _chk(ta, torch.int32, "ce.ta")
    check(_lib.load().lt_ce_fwd_bwd(_p(s), _p(teacher), _p(ta), _p(tb), _p(row_weight), _p(slot), scale, inv_temp, _p(loss),
                                    _p(dlogits), rows, K, _stream()), "lt_ce_fwd_bwd")


def softmax_stats_colsum(logits: Tensor, center: Optional[Tensor], stats: Tensor, colsum: Tensor, rows: int, K: int, inv_temp: float,
                         scratch: Tensor) -> None:
    """stats[rows, 2] = (max, 1 / sum-exp) of (logits - center) * inv_temp per row, colsum[K] = column sums of the raw logits: the softmax
    centering of the teacher logits in one pass, without the probability matrix (lt_softmax_stats_colsum).  scratch: f32, >= K elements
    (256 * K for one workgroup per CU)."""
    _chk(stats, torch.float32, "softmax_stats.stats")
    _chk(scratch, torch.float32, "softmax_stats.scratch")
    fn = _lib.load().lt_softmax_stats_colsum_bf16 if logits.dtype == torch.bfloat16 else _lib.load().lt_softmax_stats_colsum   # bf16 logit rows: same arithmetic (fp32)
    check(fn(_p(logits) if rows else None, _p(center), _p(stats) if rows else None, _p(colsum), rows, K, inv_temp,
             _p(scratch), scratch.numel(), _stream()), "lt_softmax_stats_colsum")


def ce_fwd_bwd_logits(s: Tensor, t_logits: Tensor, t_stats: Tensor, center_a: Optional[Tensor], center_b: Optional[Tensor], split_row: int,
                      ta: Tensor, tb: Optional[Tensor], row_weight: Optional[Tensor], scale: float, inv_temp: float, inv_temp_t: float,
                      loss: Tensor, dlogits: Optional[Tensor], rows: int, K: int, slot: Optional[Tensor] = None) -> None:
    """`ce_fwd_bwd` against teacher probabilities rebuilt from the teacher logits, their row statistics and the centers
    (lt_ce_fwd_bwd_logits): teacher rows < split_row use center_a, the others center_b."""
    _chk(ta, torch.int32, "ce.ta")
    assert s.dtype == t_logits.dtype, "student and teacher logits share one element type"
    fn = _lib.load().lt_ce_fwd_bwd_logits_bf16 if s.dtype == torch.bfloat16 else _lib.load().lt_ce_fwd_bwd_logits
    check(fn(_p(s), _p(t_logits), _p(t_stats), _p(center_a), _p(center_b), split_row, _p(ta), _p(tb), _p(row_weight),
             _p(slot), scale, inv_temp, inv_temp_t, _p(loss), _p(dlogits), rows, K, _stream()), "lt_ce_fwd_bwd_logits")


def sk_exp(logits: Tensor, Q: Tensor, inv_temp: float) -> None:
    check(_lib.load().lt_sk_exp(_p(logits), _p(Q), logits.numel(), inv_temp, _stream()), "lt_sk_exp")


def sk_iter(Q: Tensor, colsum: Tensor, rows: int, K: int, n_total: float, final_mul: float) -> None:
    check(_lib.load().lt_sk_iter(_p(Q), _p(colsum), rows, K, n_total, final_mul, _stream()), "lt_sk_iter")


def koleo_fwd_bwd(x: Tensor, ld: int, loss: Tensor, dx: Tensor, ld_dx: int, n: int, D: int, weight: float, ws: Tensor, nn: Tensor,
                  eps: float = 1e-8) -> None:
    assert ws.numel() >= 2 * n * D + 2 * n and nn.dtype == torch.int32
    check(_lib.load().lt_koleo_fwd_bwd(_p(x), ld, _p(loss), _p(dx), ld_dx, n, D, eps, weight, _p(ws), _p(nn), _stream()),
          "lt_koleo_fwd_bwd")


def mse_fwd_bwd(s: Tensor, t: Tensor, ds: Optional[Tensor], n: int, scale: float, loss: Tensor) -> None:
    _chk(s, torch.float32, "mse.s")
    _chk(t, torch.float32, "mse.t")
    check(_lib.load().lt_mse_fwd_bwd(_p(s), _p(t), _p(ds), n, scale, _p(loss), _stream()), "lt_mse_fwd_bwd")


# ------------------------------------------------------------------------------------------ optimizer
def sumsq(g: Tensor, out: Tensor) -> None:
    check(_lib.load().lt_sumsq_f32(_p(g), _p(out), g.numel(), _stream()), "lt_sumsq_f32")


def adamw_flat(p: Tensor, g: Tensor, m: Tensor, v: Tensor, p_bf16: Optional[Tensor], seg_of_chunk: Tensor, seg_lr: Tensor,
               seg_wd_on: Tensor, seg_frozen: Tensor, freeze: int, lr_factor: float, wd: float, beta1: float, beta2: float,
               eps: float, step: int, sumsq_t: Optional[Tensor], max_norm: float) -> None:
    check(_lib.load().lt_adamw_flat(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), _p(seg_of_chunk), _p(seg_lr), _p(seg_wd_on),
                                    _p(seg_frozen), int(freeze), lr_factor, wd, beta1, beta2, eps, step, _p(sumsq_t), max_norm,
                                    _stream()), "lt_adamw_flat")


def lars_flat(p: Tensor, g: Tensor, buf: Optional[Tensor], p_bf16: Optional[Tensor], seg_of_chunk: Tensor, seg_chunk_begin: Tensor, seg_lr: Tensor,
              seg_wd_on: Tensor, ws: Tensor, seg_norms: Tensor, lr_factor: float, wd: float, momentum: float, dampening: float, nesterov: bool,
              trust: float, eps: float, first_step: bool, sumsq_t: Optional[Tensor], max_norm: float) -> None:
    """One LARS step on flat storage (lt_lars_norms + lt_lars_flat).  ws: 2 * numel / 1024 floats, seg_norms: [segments, 2]."""
    n, nseg = p.numel(), seg_lr.numel()
    assert ws.numel() >= 2 * n // 1024 and seg_norms.numel() >= 2 * nseg and seg_chunk_begin.numel() == nseg + 1
    lib = _lib.load()
    check(lib.lt_lars_norms(_p(p), _p(g), n, _p(seg_chunk_begin), nseg, _p(ws), _p(seg_norms), _stream()), "lt_lars_norms")
    check(lib.lt_lars_flat(_p(p), _p(g), _p(buf), _p(p_bf16), n, _p(seg_of_chunk), _p(seg_lr), _p(seg_wd_on), _p(seg_norms), lr_factor, wd, momentum,
                           dampening, int(nesterov), trust, eps, int(first_step), _p(sumsq_t), max_norm, _stream()), "lt_lars_flat")


def reduce_begin(scratch: Tensor, first_slot: int = 0) -> None:
    """Start deferring the cross-workgroup sums of the backward kernels into `scratch` (lt_reduce_begin_at: order-fixed reductions;
    `first_slot`: number range of this region's cached flush tables -- one range per region of a step, see include/lt_amd.h)."""
    _chk(scratch, torch.float32, "reduce_begin.scratch")
    check(_lib.load().lt_reduce_begin_at(_p(scratch), scratch.numel(), int(first_slot)), "lt_reduce_begin_at")


def reduce_flush() -> None:
    """Add everything recorded since the last flush, on the current stream (which must be ordered after the producers' streams)."""
    check(_lib.load().lt_reduce_flush(_stream()), "lt_reduce_flush")


def reduce_end() -> None:
    check(_lib.load().lt_reduce_end(_stream()), "lt_reduce_end")


def reduce_overflows() -> int:
    return int(_lib.load().lt_reduce_overflows())


def sgd_flat(p: Tensor, g: Tensor, buf: Optional[Tensor], p_bf16: Optional[Tensor], seg_of_chunk: Tensor, seg_lr: Tensor, seg_wd_on: Tensor,
             lr_factor: float, wd: float, momentum: float, dampening: float, nesterov: bool, first_step: bool, sumsq_t: Optional[Tensor],
             max_norm: float) -> None:
    """One torch.optim.SGD step on flat storage (lt_sgd_flat): coupled weight decay on the decayed segments, momentum buffer, clipping."""
    check(_lib.load().lt_sgd_flat(_p(p), _p(g), _p(buf), _p(p_bf16), p.numel(), _p(seg_of_chunk), _p(seg_lr), _p(seg_wd_on), lr_factor, wd, momentum,
                                  dampening, int(nesterov), int(first_step), _p(sumsq_t), max_norm, _stream()), "lt_sgd_flat")


def ema_flat(teacher: Tensor, student: Tensor, teacher_bf16: Optional[Tensor], m: float) -> None:
    check(_lib.load().lt_ema_flat(_p(teacher), _p(student), _p(teacher_bf16), teacher.numel(), m, _stream()), "lt_ema_flat")


# ------------------------------------------------------------------------------------------ convolutional student (NHWC bf16)
def conv_out_size(n: int, k: int, stride: int, pad: int) -> int:
    return (n + 2 * pad - k) // stride + 1


def im2col_nhwc(x: Tensor, cols: Tensor, B: int, H: int, W: int, Cc: int, KH: int, KW: int, stride: int, pad: int) -> Tensor:
    _chk(x, torch.bfloat16, "im2col.x")
    _chk(cols, torch.bfloat16, "im2col.cols")
    check(_lib.load().lt_im2col_nhwc_bf16(_p(x), _p(cols), B, H, W, Cc, KH, KW, stride, pad, cols.shape[-1], _stream()), "lt_im2col_nhwc_bf16")
    return cols


def col2im_nhwc(dcols: Tensor, dx: Tensor, B: int, H: int, W: int, Cc: int, KH: int, KW: int, stride: int, pad: int,
                add: Optional[Tensor] = None) -> Tensor:
    _chk(dcols, torch.bfloat16, "col2im.dcols")
    _chk(dx, torch.bfloat16, "col2im.dx")
    check(_lib.load().lt_col2im_nhwc_bf16(_p(dcols), _p(add), _p(dx), B, H, W, Cc, KH, KW, stride, pad, dcols.shape[-1], _stream()),
          "lt_col2im_nhwc_bf16")
    return dx


def im2col_nchw_f32(img: Tensor, cols: Tensor, KH: int, KW: int, stride: int, pad: int) -> Tensor:
    _chk(img, torch.float32, "im2col_nchw.img")
    _chk(cols, torch.bfloat16, "im2col_nchw.cols")
    B, Cin, H, W = img.shape
    check(_lib.load().lt_im2col_nchw_f32(_p(img), _p(cols), B, Cin, H, W, KH, KW, stride, pad, cols.shape[-1], _stream()), "lt_im2col_nchw_f32")
    return cols


def batchnorm_ws_floats(Cc: int) -> int:
    return int(_lib.load().lt_batchnorm_ws_floats(Cc))


def batchnorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, y: Tensor, mean: Tensor, rstd: Tensor, rows: int, Cc: int, ws: Tensor,
                  resid: Optional[Tensor] = None, running_mean: Optional[Tensor] = None, running_var: Optional[Tensor] = None,
                  eps: float = 1e-5, momentum: float = 0.1, relu: bool = False, sync: Optional[Any] = None) -> Tensor:
    """Training-mode BatchNorm.  sync: None, or a callable that adds a device tensor of doubles over the ranks in place (an all-reduce):
    SyncBatchNorm -- the statistics and the running estimates then come from the rows of ALL ranks."""
    _chk(x, torch.bfloat16, "batchnorm.x")
    _chk(y, torch.bfloat16, "batchnorm.y")
    assert ws.numel() >= batchnorm_ws_floats(Cc) and ws.dtype == torch.float32
    lib = _lib.load()
    if sync is None:
        check(lib.lt_batchnorm_fwd(_p(x), _p(gamma), _p(beta), _p(resid), _p(y), _p(mean), _p(rstd), _p(running_mean), _p(running_var), rows, Cc,
                                   eps, momentum, int(relu), _p(ws), _stream()), "lt_batchnorm_fwd")
        return y
    sums = torch.zeros(2 * Cc + 1, dtype=torch.float64, device=x.device)
    if rows > 0:
        check(lib.lt_batchnorm_stats(_p(x), rows, Cc, _p(ws), _p(sums), _stream()), "lt_batchnorm_stats")
    sync(sums)      # a rank without rows for this call still takes part in the collective
    if rows == 0:
        return y
    check(lib.lt_batchnorm_fwd_from_sums(_p(x), _p(sums), _p(gamma), _p(beta), _p(resid), _p(y), _p(mean), _p(rstd), _p(running_mean), _p(running_var),
                                         rows, Cc, eps, momentum, int(relu), _stream()), "lt_batchnorm_fwd_from_sums")
    return y


def batchnorm_apply(x: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor, beta: Tensor, y: Tensor, rows: int, Cc: int,
                    resid: Optional[Tensor] = None, relu: bool = False) -> Tensor:
    _chk(x, torch.bfloat16, "batchnorm_apply.x")
    check(_lib.load().lt_batchnorm_apply(_p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(resid), _p(y), rows, Cc, int(relu), _stream()),
          "lt_batchnorm_apply")
    return y


def batchnorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dx: Tensor, rows: int, Cc: int, ws: Tensor,
                  y: Optional[Tensor] = None, dz: Optional[Tensor] = None, dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None,
                  sync: Optional[Any] = None) -> Tensor:
    _chk(dy, torch.bfloat16, "batchnorm_bwd.dy")
    _chk(x, torch.bfloat16, "batchnorm_bwd.x")
    assert ws.numel() >= batchnorm_ws_floats(Cc) and ws.dtype == torch.float32
    lib = _lib.load()
    if sync is None:
        check(lib.lt_batchnorm_bwd(_p(dy), _p(y), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dz), _p(dx), _p(dgamma), _p(dbeta), rows, Cc, _p(ws),
                                   _stream()), "lt_batchnorm_bwd")
        return dx
    # SyncBatchNorm backward: the two means of the input gradient run over the rows of all ranks
    sums = torch.zeros(2 * Cc + 1, dtype=torch.float64, device=x.device)
    if rows > 0:
        check(lib.lt_batchnorm_bwd_sums(_p(dy), _p(y), _p(x), _p(mean), _p(rstd), _p(dz), _p(dgamma), _p(dbeta), rows, Cc, _p(ws), _p(sums), _stream()),
              "lt_batchnorm_bwd_sums")
    sync(sums)
    if rows == 0:
        return dx
    check(lib.lt_batchnorm_bwd_from_sums(_p(dz if dz is not None else dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(sums), _p(dx), rows, Cc, _p(ws),
                                         _stream()), "lt_batchnorm_bwd_from_sums")
    return dx


def maxpool3x3s2_fwd(x: Tensor, y: Tensor, idx: Tensor, B: int, H: int, W: int, Cc: int) -> None:
    _chk(x, torch.bfloat16, "maxpool.x")
    assert idx.dtype == torch.uint8
    check(_lib.load().lt_maxpool3x3s2_fwd(_p(x), _p(y), _p(idx), B, H, W, Cc, _stream()), "lt_maxpool3x3s2_fwd")


def maxpool3x3s2_bwd(dy: Tensor, idx: Tensor, dx: Tensor, B: int, H: int, W: int, Cc: int) -> None:
    _chk(dy, torch.bfloat16, "maxpool_bwd.dy")
    check(_lib.load().lt_maxpool3x3s2_bwd(_p(dy), _p(idx), _p(dx), B, H, W, Cc, _stream()), "lt_maxpool3x3s2_bwd")


def token_mean(x: Tensor, out: Tensor, B: int, n: int, Cc: int) -> Tensor:
    _chk(x, torch.bfloat16, "token_mean.x")
    _chk(out, torch.bfloat16, "token_mean.out")
    check(_lib.load().lt_token_mean_bf16(_p(x), _p(out), B, n, Cc, _stream()), "lt_token_mean_bf16")
    return out


def pool_bwd_add(d_tok: Optional[Tensor], d_pool: Optional[Tensor], out: Tensor, B: int, n: int, Cc: int) -> Tensor:
    _chk(out, torch.bfloat16, "pool_bwd_add.out")
    check(_lib.load().lt_pool_bwd_add(_p(d_tok), _p(d_pool), _p(out), B, n, Cc, _stream()), "lt_pool_bwd_add")
    return out


def add_bf16(a: Tensor, b: Tensor, out: Tensor) -> Tensor:
    check(_lib.load().lt_add_bf16(_p(a), _p(b), _p(out), a.numel(), _stream()), "lt_add_bf16")
    return out
