"""ViT backbone forward / backward on the HIP ops -- the MI355X replacement for
DinoVisionTransformer.forward_features (LT/_models/dinov2_vit/dinov2_vit_src/models/vision_transformer.py:307-384)
and Block / Attention / Mlp / LayerScale / PatchEmbed (layers/*.py), with an explicit backward pass (no autograd).

Parameter names equal the reference's DinoVisionTransformer.state_dict() keys (no register tokens, ffn "mlp").
Numerics follow the reference under `precision="bf16-mixed"`: fp32 master weights, bf16 MFMA operands,
fp32 accumulation, fp32 residual stream / LayerNorm / softmax statistics.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Any, Dict, Iterable, Iterator, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from . import ops
from .params import FlatParams


@dataclass
class ViTConfig:
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0
    patch_size: int = 16
    img_size: int = 224
    in_chans: int = 3
    init_values: Optional[float] = 1e-5
    interpolate_offset: float = 0.1
    interpolate_antialias: bool = False
    drop_path_rate: float = 0.0
    drop_path_uniform: bool = False
    num_register_tokens: int = 0
    ffn_layer: str = "mlp"   # "mlp" | "swiglu" | "swiglufused" (vision_transformer.py:179-185)
    ln_eps: float = 1e-6     # DINOv3 "layernormbf16" uses 1e-5
    rope_base: Optional[float] = None   # DINOv3: rotary embedding on q/k of the patch tokens instead of a learned pos_embed
                                        # (zero, frozen pos_embed in the state)
    rope_rescale: Optional[float] = None  # training-mode coordinate augmentation of a DINOv3 *student*: log-uniform rescale
                                          # in [1/r, r], drawn per block (rope_position_encoding.py:104-109; 2 for vits16..vitl16)
    mask_k_bias: bool = False           # DINOv3 LinearKMaskedBias: the K third of the qkv bias is held at zero (no gradient)
    block_chunks: int = 0               # checkpoint key naming only: the vitl14 / vitg14 YAMLs build `blocks.<chunk>.<i>.` (FSDP chunks)

    @property
    def swiglu(self) -> bool:
        return self.ffn_layer in ("swiglu", "swiglufused")

    @property
    def hidden(self) -> int:
        h = int(self.embed_dim * self.mlp_ratio)
        if self.swiglu:   # SwiGLUFFNFused: 2/3 of the MLP width rounded up to 8 (swiglu_ffn.py:61-63)
            h = (int(h * 2 / 3) + 7) // 8 * 8
        return h

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads


ARCHS: Dict[str, Dict[str, Any]] = {
    "_vit_test": dict(embed_dim=8, depth=3, num_heads=2, mlp_ratio=1.0),
    "vit_tiny": dict(embed_dim=192, depth=12, num_heads=3),
    "vit_small": dict(embed_dim=384, depth=12, num_heads=6),
    "vit_base": dict(embed_dim=768, depth=12, num_heads=12),
    "vit_large": dict(embed_dim=1024, depth=24, num_heads=16),
    "vit_giant2": dict(embed_dim=1536, depth=40, num_heads=24),
}


def vit_param_shapes(cfg: ViTConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    D, hid, p = cfg.embed_dim, cfg.hidden, cfg.patch_size
    n_p = (cfg.img_size // p) ** 2
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("cls_token", (1, 1, D)), ("pos_embed", (1, n_p + 1, D)),
    ]
    if cfg.num_register_tokens:
        out.append(("register_tokens", (1, cfg.num_register_tokens, D)))
    out += [("mask_token", (1, D)), ("patch_embed.proj.weight", (D, cfg.in_chans, p, p)), ("patch_embed.proj.bias", (D,))]
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        out += [(b + "norm1.weight", (D,)), (b + "norm1.bias", (D,)), (b + "attn.qkv.weight", (3 * D, D)),
                (b + "attn.qkv.bias", (3 * D,)), (b + "attn.proj.weight", (D, D)), (b + "attn.proj.bias", (D,))]
        if cfg.init_values:
            out.append((b + "ls1.gamma", (D,)))
        out += [(b + "norm2.weight", (D,)), (b + "norm2.bias", (D,))]
        if cfg.swiglu:
            out += [(b + "mlp.w12.weight", (2 * hid, D)), (b + "mlp.w12.bias", (2 * hid,)), (b + "mlp.w3.weight", (D, hid)),
                    (b + "mlp.w3.bias", (D,))]
        else:
            out += [(b + "mlp.fc1.weight", (hid, D)), (b + "mlp.fc1.bias", (hid,)), (b + "mlp.fc2.weight", (D, hid)),
                    (b + "mlp.fc2.bias", (D,))]
        if cfg.init_values:
            out.append((b + "ls2.gamma", (D,)))
    out += [("norm.weight", (D,)), ("norm.bias", (D,))]
    return out


def init_vit_state(cfg: ViTConfig, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
    """Random init with the reference's initialisers (vision_transformer.py:243-249, 489-495; Conv2d default)."""
    sd: Dict[str, Tensor] = {}
    for name, shape in vit_param_shapes(cfg):
        if name in ("cls_token", "register_tokens"):
            t = torch.empty(shape).normal_(std=1e-6, generator=generator)
        elif name == "pos_embed" or (name.endswith(".weight") and len(shape) == 2):
            t = torch.nn.init.trunc_normal_(torch.empty(shape), std=0.02, generator=generator)
        elif name == "patch_embed.proj.weight":
            bound = 1 / math.sqrt(shape[1] * shape[2] * shape[3])
            t = torch.empty(shape).uniform_(-bound, bound, generator=generator)
        elif name == "patch_embed.proj.bias":
            bound = 1 / math.sqrt(cfg.in_chans * cfg.patch_size ** 2)
            t = torch.empty(shape).uniform_(-bound, bound, generator=generator)
        elif name.endswith("gamma"):
            t = torch.full(shape, float(cfg.init_values))
        elif "norm" in name and name.endswith(".weight"):
            t = torch.ones(shape)
        else:
            t = torch.zeros(shape)
        sd[name] = t
    return sd


_DROP_RATES: Dict[Tuple[float, int], List[float]] = {}


def block_drop_rates(cfg: ViTConfig) -> List[float]:
    """Per-block stochastic-depth rate (vision_transformer.py:150-157): uniform or linspace(0, rate, depth)."""
    if cfg.drop_path_uniform:
        return [float(cfg.drop_path_rate)] * cfg.depth
    key = (float(cfg.drop_path_rate), int(cfg.depth))     # (asked for twice per step: the torch linspace + 12 .item() calls are cached)
    if key not in _DROP_RATES:
        _DROP_RATES[key] = [x.item() for x in torch.linspace(0, cfg.drop_path_rate, cfg.depth)]
    return list(_DROP_RATES[key])


def make_drop_plan(cfg: ViTConfig, batch: int, generator: Optional[torch.Generator] = None) -> Optional[List[Any]]:
    """Host-side draws for one training forward of the student (one entry per residual branch, attn then ffn):
    rate > 0.1 -> ("subset", randperm(b)[:max(int(b(1-rate)),1)])   (block.py:118-141, batch-subset stochastic depth)
    0 < rate <= 0.1 -> ("persample", bernoulli(keep)/keep)          (block.py:108-111 + drop_path.py:16-28)
    The reference draws from the device RNG; we draw from a host generator (same distribution, different stream)."""
    rates = block_drop_rates(cfg)
    if not any(r > 0 for r in rates):
        return None
    plan: List[Any] = []
    for r in rates:
        for _branch in range(2):
            if r == 0.0:
                plan.append(None)
            elif r > 0.1:
                s = max(int(batch * (1 - r)), 1)
                plan.append(("subset", torch.randperm(batch, generator=generator)[:s]))
            else:
                keep = 1 - r
                sc = torch.empty(batch).bernoulli_(keep, generator=generator)
                if keep > 0.0:
                    sc.div_(keep)
                plan.append(("persample", sc))
    return plan


class Workspace:
    """Named, shape-checked HBM buffers reused across steps (288 GB HBM: activations are kept, not recomputed)."""

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.bufs: Dict[str, Tensor] = {}
        self.zero_names: set = set()     # buffers whose never-written parts rely on the zero fill of their allocation (`zero=True`)
        self.generation = 0              # bumped by every (re)allocation: recorded launch plans / captured graphs hold raw addresses

    def get(self, name: str, shape: Tuple[int, ...], dtype: torch.dtype, zero: bool = False, pad_rows: int = 0) -> Tensor:
        """`zero`: zero-fill when the buffer is (re)allocated (padding that kernels never write must stay finite).
        `pad_rows`: allocate the first dimension rounded up to this multiple and return the leading `shape[0]` rows -- operands of the
        weight-gradient GEMMs, whose contraction over rows runs in whole 64-row tiles (`padded_rows`)."""
        full = tuple(shape)
        if pad_rows:
            full = (-(-shape[0] // pad_rows) * pad_rows,) + tuple(shape[1:])
        t = self.bufs.get(name)
        if t is None or tuple(t.shape) != full or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(full, dtype=dtype, device=self.device)
            self.bufs[name] = t
            self.generation += 1
        if zero:
            self.zero_names.add(name)
        return t[:shape[0]] if pad_rows else t

    def rezero(self) -> None:
        """Zero-fill every `zero=True` buffer again.  Those buffers are filled once, at allocation, and afterwards only hold what steps wrote
        into them (e.g. the last block's block-middle tensor: rows outside the loss rows keep the values of earlier steps and only have to
        be FINITE).  After a step that produced Inf / NaN activations they are not: whoever restores the weights (`load_state_dict`) calls
        this, otherwise 0 * NaN in the dense final LayerNorm keeps every later step's gradients NaN."""
        for name in self.zero_names:
            t = self.bufs.get(name)
            if t is not None:
                t.zero_()

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


def padded_rows(t: Tensor, rows: int, multiple: int = 64) -> Optional[Tensor]:
    """`t` (row-major [>= rows, C], possibly the leading slice of a larger allocation) seen with its row count rounded up to `multiple`,
    or None when the allocation ends before that.  A weight-gradient GEMM contracts over the rows: with the <= 63 pad rows zeroed in both
    operands it runs in whole K-tiles on the 256-row slab kernel (deterministic split-K) instead of the 128-row kernel's fp32 atomics."""
    kpad = -(-rows // multiple) * multiple
    if kpad == rows:
        return t[:rows]
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) != t.shape[1]:
        return None
    avail = (t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()) // t.shape[1]
    return t.as_strided((kpad, t.shape[1]), (t.shape[1], 1)) if avail >= kpad else None


DETERMINISTIC_SPLIT_K = os.environ.get("LT_DETERMINISTIC", "1") != "0"


def split_k_plan(M: int, N: int, K: int, trans_a: bool, split: int) -> Dict[str, int]:
    """`split_k` / `force_kernel` arguments of an accumulating GEMM with a split contraction.  The 256-row kernel reduces its K-slices
    through fp32 slabs in a fixed order; the dispatcher sends short contractions (< 4096 rows) to the 128-row kernel, whose slices meet in
    fp32 atomics -- arrival order, last bits differ from run to run.  For reproducible steps (LT_DETERMINISTIC, default on) every split
    contraction the slab kernel can take is pinned to it (force_kernel 11: the static-address form of the 256-row kernel, which falls back to
    the four-phase form 8 where it is not eligible -- same slabs, same bits), and the rest run unsplit."""
    if split <= 1 or not DETERMINISTIC_SPLIT_K:
        return dict(split_k=split)
    if K % 64 == 0 and N % 8 == 0 and N >= 128 and M >= 64 and (not trans_a or M % 8 == 0):
        return dict(split_k=split, force_kernel=11)
    return dict(split_k=1)


def _split_k(tiles: int, k: int, slots: int = 512) -> int:
    """split-K factor for wgrad GEMMs (few output tiles, very long contraction): pick the slice count whose
    workgroup count best fills whole waves of the chip (256 CUs x 2 resident 128x128 workgroups), keeping at
    least 8 k-tiles (512 rows) per slice; ties go to fewer slices (fewer fp32 atomics)."""
    best, best_eff = 2 if k >= 4096 else 1, 0.0   # >1 also enables the auto slice count of the 256-row kernel
    for s in range(best, 33):
        if s > 1 and k // s < 512:
            break
        blocks = tiles * s
        eff = blocks / (((blocks + slots - 1) // slots) * slots)
        if eff > best_eff + 0.02:
            best, best_eff = s, eff
    return best


class JointWgrad:
    """One weight-gradient GEMM for two backward passes that share the weights (the global- and the local-crop pass of the student).

    Both passes contract their own rows of dY with their own rows of X into the same gradient; run as two GEMMs, each writes and re-reads
    its own fp32 split-K slabs.  With the operands of the two passes ADJACENT in memory the pair is one contraction over T_a + T_b rows:
    half the slab traffic and half the launches.  `seed` places the operand buffers of both passes in joint allocations under the
    workspace names the passes ask for; `deposit` is called by each pass's `backward_iter` where it would have launched its own GEMM,
    and the second deposit of a pair launches the joint one on `side`.  A deposited buffer may be overwritten only after that launch:
    `backward_iter` rotates its upstream-gradient buffers so that this happens one block later (when both passes have deposited)."""

    X_NAMES = (("ln1", 1), ("att", 1), ("ln2", 1), ("act", 0))   # saved operands per block: (name, width = D (1) | hidden (0))

    def __init__(self, ws: Workspace, side: "torch.cuda.Stream", tags: Tuple[str, str]) -> None:
        self.ws, self.side, self.tags = ws, side, tags
        self.joint: Dict[int, Tensor] = {}      # data_ptr of the first pass's view -> (joint tensor, rows of the first, rows of the second)
        self.pending: Dict[str, Dict[str, Any]] = {}
        self.key: Optional[Tuple[Any, ...]] = None
        self.launched = 0

    def seed(self, depth: int, T: Tuple[int, int], D: int, hid: int, hid1: int) -> bool:
        """Joint allocations for the operands of both passes (row counts `T`, both multiples of 64).  Idempotent per geometry."""
        key = (depth, T, D, hid, hid1)
        if self.key == key:
            return True
        if T[0] % 64 or T[1] % 64:
            return False
        self.joint.clear()
        ta, tb = self.tags
        dev = self.ws.device

        def place(name_a: str, name_b: str, width: int) -> None:
            j = torch.empty((T[0] + T[1], width), dtype=torch.bfloat16, device=dev)
            self.ws.bufs[name_a], self.ws.bufs[name_b] = j[:T[0]], j[T[0]:]
            self.joint[j.data_ptr()] = j

        for i in range(depth):
            for n, is_d in self.X_NAMES:
                place(f"{ta}.b{i}.{n}", f"{tb}.b{i}.{n}", D if is_d else hid)
        for n, width in (("dD0", D), ("dD1", D), ("dD2r", D), ("dD3r", D), ("dH", hid1), ("dQ", 3 * D)):
            place(f"{ta}.{n}", f"{tb}.{n}", width)
        self.key, self.T = key, T
        self.ws.generation += 1
        return True

    def deposit(self, tag: str, wname: str, dy: Tensor, xin: Tensor, rows: int, run, consumed: Dict[int, Any]) -> None:
        """`run(dy, xin, K)` launches the accumulating GEMM (on the current stream); `consumed` is the depositing pass's map of
        buffer -> event after which the side stream no longer reads it."""
        main = torch.cuda.current_stream()
        me = dict(tag=tag, dy=dy, xin=xin, rows=rows, run=run, consumed=consumed, ev=main.record_event())
        other = self.pending.pop(wname, None)
        if other is None:
            self.pending[wname] = me
            consumed[dy.data_ptr()] = wname    # placeholder: `before_write` forces the launch if the partner has not deposited by then
            return
        a, b = (me, other) if me["tag"] == self.tags[0] else (other, me)
        jy, jx = self.joint.get(a["dy"].data_ptr()), self.joint.get(a["xin"].data_ptr())
        adjacent = (jy is not None and jx is not None and a["rows"] == self.T[0] and b["rows"] == self.T[1]
                    and b["dy"].data_ptr() == jy.data_ptr() + self.T[0] * jy.shape[1] * 2
                    and b["xin"].data_ptr() == jx.data_ptr() + self.T[0] * jx.shape[1] * 2)
        self.side.wait_event(a["ev"])
        self.side.wait_event(b["ev"])
        with torch.cuda.stream(self.side):
            if adjacent:
                me["run"](jy, jx, self.T[0] + self.T[1])
                self.launched += 1
            else:
                a["run"](a["dy"], a["xin"], a["rows"])
                b["run"](b["dy"], b["xin"], b["rows"])
            ev = self.side.record_event()
        a["consumed"][a["dy"].data_ptr()] = ev
        b["consumed"][b["dy"].data_ptr()] = ev

    def flush(self, only: Optional[str] = None) -> None:
        """Deposits whose partner never came (a pass that ran this layer on a row subset): launched on their own."""
        for wname, d in list(self.pending.items()):
            if only is not None and wname != only:
                continue
            self.side.wait_event(d["ev"])
            with torch.cuda.stream(self.side):
                d["run"](d["dy"], d["xin"], d["rows"])
                d["consumed"][d["dy"].data_ptr()] = self.side.record_event()
            del self.pending[wname]


class ViTEngine:
    """Runs one backbone (student or teacher) whose parameters live in a FlatParams under `prefix`."""

    def __init__(self, cfg: ViTConfig, params: FlatParams, prefix: str = "") -> None:
        self.cfg = cfg
        self.P = params
        self.prefix = prefix
        self.dev = params.device
        self._pos_maps: Dict[Tuple[int, int], Optional[Tensor]] = {}
        self._rope_ang: Dict[Tuple[int, int], Tensor] = {}
        self._rope: Dict[Tuple[int, int], Tuple[Tensor, Tensor]] = {}
        self._resize_taps: Dict[Tuple[int, int, int, int], Any] = {}
        self._fwd_graphs: Dict[Any, Dict[str, Any]] = {}
        self.graph_forward = os.environ.get("LT_GRAPH_FWD", "0") != "0"   # HIP-graph replay of the static forward blocks (`_graphed_blocks`)
        # the LayerNorm behind a residual GEMM handed to the GEMM call (lt_gemm_desc.ln_*; round 6: the library issues the LayerNorm launch, one
        # call across the C ABI instead of two).  LT_FUSE_LN=0: separate calls from here -- same launches, same bits
        self.fuse_ln = os.environ.get("LT_FUSE_LN", "1") != "0" and not self.graph_forward
        # launch-plan replay of the static forward blocks (round 6, ops.LaunchPlan): the calls across the C ABI that blocks 0 .. depth-2 make are
        # logged on the third pass of a geometry and replayed from then on -- same launches, same streams, none of the Python around them
        self.plan_forward = os.environ.get("LT_PLAN_FWD", "1") != "0" and not self.graph_forward
        self._fwd_plans: Dict[Any, Dict[str, Any]] = {}
        D = cfg.embed_dim
        kreal = cfg.in_chans * cfg.patch_size ** 2
        self.kreal = kreal
        self.kpad = (kreal + 7) // 8 * 8   # 16-byte rows for the MFMA GEMM (patch 14: 588 -> 592)
        self.wpe_pad: Optional[Tensor] = None
        if self.kpad != kreal:
            self.wpe_pad = torch.zeros(D, self.kpad, dtype=torch.bfloat16, device=self.dev)
            self.refresh_padded_weights()
        assert D % 8 == 0 and cfg.hidden % 8 == 0, "embed_dim and mlp hidden must be multiples of 8"
        if cfg.init_values is not None and float(cfg.init_values) == 0.0:
            # the LayerScale gradient is recovered from the weight gradient by dividing by gamma (lt_layerscale_dgamma): gamma == 0 has none
            raise ValueError("LayerScale init_values == 0 is not supported (use None for no LayerScale): its gradient is formed as (W . dW) / gamma")

    def _graphed_blocks(self, tag: str, x: Tensor, save: bool, n: int, run_block: Any) -> Tuple[Tensor, List[Dict[str, Any]]]:
        """Blocks 0 .. n-1 of a forward pass from the token buffer `x`: eagerly on the first two calls of a (pass, shape), then captured
        into a HIP graph (stream capture of the same launches) and replayed with one launch on the caller's stream."""
        key = (tag, tuple(x.shape), bool(save), n, x.data_ptr())
        ent = self._fwd_graphs.setdefault(key, {"calls": 0, "graph": None})

        def eager() -> Tuple[Tensor, List[Dict[str, Any]]]:
            xs, blks = x, []
            for i in range(n):
                xs, a_, m_ = run_block(i, xs, f"{tag}.b{i}." if save else f"{tag}.tmp.", save, save)
                blks.append({"attn": a_, "mlp": m_})
            return xs, blks

        if ent["graph"] is None:
            ent["calls"] += 1
            if ent["calls"] < 3:
                return eager()
            g = torch.cuda.CUDAGraph()
            cur = torch.cuda.current_stream()
            with torch.cuda.graph(g):        # (synchronises the device, captures on a side stream; nothing executes yet)
                xo, blks = eager()
            torch.cuda.set_stream(cur)
            ent.update(graph=g, out=xo, blocks=blks)
        ent["graph"].replay()
        return ent["out"], [{"attn": dict(b["attn"]), "mlp": dict(b["mlp"])} for b in ent["blocks"]]

    def refresh_padded_weights(self) -> None:
        """Re-derive the zero-padded bf16 patch-embedding matrix after the fp32 weights changed (optimizer / EMA)."""
        if self.wpe_pad is not None:
            ops.cast_pad_rows(self.w("patch_embed.proj.weight").view(self.cfg.embed_dim, -1), self.wpe_pad, self.cfg.embed_dim, self.kreal, self.kpad)

    def _patch_weight_bf16(self) -> Tensor:
        return self.wpe_pad if self.wpe_pad is not None else self.wb("patch_embed.proj.weight").view(self.cfg.embed_dim, -1)

    # ---- parameter access -------------------------------------------------------------------
    def w(self, name: str) -> Tensor:
        return self.P.p[self.prefix + name]

    def wb(self, name: str) -> Tensor:
        return self.P.b[self.prefix + name]

    def gw(self, name: str) -> Tensor:
        return self.P.g[self.prefix + name]

    def has(self, name: str) -> bool:
        return (self.prefix + name) in self.P.p

    # ---- positional embedding for a gh x gw grid ------------------------------------------------
    def _pos_map(self, gh: int, gw: int) -> Optional[Tensor]:
        """Linear map [gh*gw, M*M] with pos_grid = map @ pos_native, obtained by pushing an identity basis through
        the reference's bicubic F.interpolate call (vision_transformer.py:283-300) once on the host."""
        key = (gh, gw)
        if key in self._pos_maps:
            return self._pos_maps[key]
        n_native = self.w("pos_embed").shape[1] - 1
        m = int(math.sqrt(n_native))
        assert m * m == n_native
        if gh * gw == n_native and gh == gw:
            self._pos_maps[key] = None
            return None
        basis = torch.eye(n_native).reshape(1, m, m, n_native).permute(0, 3, 1, 2)  # [1, M*M chans, M, M]
        kw: Dict[str, Any] = {}
        if self.cfg.interpolate_offset:
            kw["scale_factor"] = (float(gh + self.cfg.interpolate_offset) / m, float(gw + self.cfg.interpolate_offset) / m)
        else:
            kw["size"] = (gh, gw)
        out = F.interpolate(basis, mode="bicubic", antialias=self.cfg.interpolate_antialias, **kw)
        assert out.shape[-2:] == (gh, gw)
        mp = out.permute(0, 2, 3, 1).reshape(gh * gw, n_native).contiguous().to(self.dev)
        self._pos_maps[key] = mp
        return mp

    def rope_tables_train(self, gh: int, gw: int) -> List[Tuple[Tensor, Tensor]]:
        """Per-block (sin, cos) tables of a DINOv3 student in training mode: the reference calls `rope_embed(H, W)` once per
        block (vision_transformer.py:269-271) and every call draws its own log-uniform rescale factor from torch's default
        generator (rope_position_encoding.py:104-109) -- same draws, same order here.  The tables of all blocks are built on
        the device in one go (angles are linear in the factor), so a step costs one small H2D copy instead of 2 per block."""
        cfg = self.cfg
        if cfg.rope_rescale is None:
            return [self._rope_tables(gh, gw)] * cfg.depth
        rmax = math.log(cfg.rope_rescale)
        muls = [float(torch.empty(1, dtype=torch.float32).uniform_(-rmax, rmax).exp()) for _ in range(cfg.depth)]
        base = self._rope_angles(gh, gw)                                                     # [P, head_dim] at factor 1
        ang = base.unsqueeze(0) * torch.tensor(muls, dtype=torch.float32).to(self.dev, non_blocking=True).view(-1, 1, 1)
        sin, cos = torch.sin(ang), torch.cos(ang)
        return [(sin[i], cos[i]) for i in range(cfg.depth)]

    def _rope_angles(self, gh: int, gw: int) -> Tensor:
        """Rotation angles f32 [gh*gw, head_dim] of DINOv3's RopePositionEmbedding (layers/rope_position_encoding.py:62-127,
        normalize_coords="separate", periods = base ** (2 i / (head_dim/2)), i < head_dim/4), on the device, cached per grid."""
        key = (gh, gw)
        if key not in self._rope_ang:
            dh = self.cfg.head_dim
            periods = float(self.cfg.rope_base) ** (2 * torch.arange(dh // 4, dtype=torch.float32) / (dh // 2))
            ch = torch.arange(0.5, gh, dtype=torch.float32) / gh
            cw = torch.arange(0.5, gw, dtype=torch.float32) / gw
            coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), dim=-1).flatten(0, 1)
            coords = 2.0 * coords - 1.0
            angles = (2 * math.pi * coords[:, :, None] / periods[None, None, :]).flatten(1, 2)
            self._rope_ang[key] = torch.cat((angles, angles), dim=-1).contiguous().to(self.dev)
        return self._rope_ang[key]

    def _rope_tables(self, gh: int, gw: int) -> Tuple[Tensor, Tensor]:
        """(sin, cos) f32 [gh*gw, head_dim] at rescale factor 1 (eval mode / frozen teacher), cached per grid."""
        key = (gh, gw)
        if key not in self._rope:
            ang = self._rope_angles(gh, gw)
            self._rope[key] = (torch.sin(ang), torch.cos(ang))
        return self._rope[key]

    def _pos_for_grid(self, ws: Workspace, tag: str, gh: int, gw: int) -> Tensor:
        D = self.cfg.embed_dim
        pe = self.w("pos_embed").view(-1, D)
        mp = self._pos_map(gh, gw)
        if mp is None:
            return pe
        pos = ws.get(tag + ".pos", (gh * gw + 1, D), torch.float32)
        pos[0].copy_(pe[0])
        ops.matmul_f32(mp, pe[1:], pos[1:], gh * gw, D, mp.shape[1])
        return pos

    # ---- forward --------------------------------------------------------------------------------
    def forward(self, ws: Workspace, tag: str, img: Tensor, masks: Optional[Tensor], save: bool,
                drop_plan: Optional[List[Any]] = None, rope_tables: Optional[List[Tuple[Tensor, Tensor]]] = None,
                checkpoint: bool = False, capture_layers: Optional[Iterable[int]] = None, capture_norm: bool = True,
                last_mlp_rows: Optional[Tuple[Tensor, int]] = None, cols: Optional[Tensor] = None) -> Dict[str, Any]:
        """img f32 [B,C,H,W] (H,W multiples of patch_size) -> ctx with "xn" f32 [B, N, D] (final-norm tokens).

        cols: the patch matrix of `img` (`ops.im2col(img, patch_size, kpad)`) when the caller already has it -- the teacher and the
        student's global-crop pass unfold the same images.

        drop_plan (training student only): 2*depth entries (attn, ffn branch per block) of None |
        ("subset", brange int64[s]) | ("persample", scale f32[B]) -- see make_drop_plan / layers/block.py:90-141.

        last_mlp_rows = (idx int64 [>= R] on the device, R): the caller reads the output only at these token rows (the DINOv2
        losses: cls rows and masked patch rows).  Tokens do not interact after the last attention, so the last block's MLP branch
        -- LayerNorm, fc1, GELU, fc2, LayerScale -- is evaluated on those R rows alone and added to them in place; every other row
        of the output is the branch-free residual and must not be read.  The saved operands are the compact R-row ones, so the
        backward of that branch runs on R rows as well (the same `mode == "subset"` path as batch-subset stochastic depth, at
        scale 1): the rows skipped there would have multiplied exact zeros."""
        cfg = self.cfg
        B, C, H, W = img.shape
        p, D, Hh, dh, hid = cfg.patch_size, cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden
        fc2 = "mlp.w3" if cfg.swiglu else "mlp.fc2"
        if H % p or W % p:
            # PatchEmbed.forward (layers/patch_embed.py:86-99): bicubic-resize to the next multiple of the patch size
            nh, nw = math.ceil(H / p) * p, math.ceil(W / p) * p
            key = (H, W, nh, nw)
            if key not in self._resize_taps:
                iy, wy = ops.bicubic_taps(H, nh)
                ix, wx = ops.bicubic_taps(W, nw)
                self._resize_taps[key] = tuple(t.to(self.dev) for t in (iy, wy, ix, wx))
            iy, wy, ix, wx = self._resize_taps[key]
            img = ops.resize_4tap(img.contiguous(), iy, wy, ix, wx, nh, nw)
            H, W = nh, nw
        gh, gw = H // p, W // p
        n_reg = cfg.num_register_tokens
        rope = None
        if cfg.rope_base is not None:   # one table per block: eval mode repeats the cached one
            rope = rope_tables if rope_tables is not None else [self._rope_tables(gh, gw)] * cfg.depth
        n_p, N = gh * gw, gh * gw + 1 + n_reg
        T = B * N
        scale = dh ** -0.5
        ctx: Dict[str, Any] = dict(B=B, N=N, n_p=n_p, gh=gh, gw=gw, T=T, masks=masks, tag=tag, rope=rope)

        if cols is None or tuple(cols.shape) != (B * n_p, self.kpad):
            cols = ops.im2col(img.contiguous(), p, self.kpad)
        patch = ws.get(tag + ".patch", (B * n_p, D), torch.float32)
        ops.gemm(cols, self._patch_weight_bf16(), patch, M=B * n_p, N=D, K=self.kpad,
                 epilogue=ops.EPI_F32, bias=self.w("patch_embed.proj.bias"))
        pos = self._pos_for_grid(ws, tag, gh, gw)
        x = ws.get(tag + ".x0" if save else tag + ".xa", (T, D), torch.float32)
        ops.assemble_tokens(patch, self.w("cls_token").view(D), pos, self.w("mask_token").view(D), masks, B, n_p, D, out=x,
                            reg=self.w("register_tokens").view(-1, D) if n_reg else None, n_reg=n_reg)
        ctx["cols"] = cols
        tok_np = np.arange(N, dtype=np.int64)
        # every stochastic-depth draw of this pass goes to the device in ONE upload per kind: the subset branches' token-row indices
        # (int64, (b * N + token) for the drawn images b) and the per-sample branches' row scales (f32, mask / keep per token row).  One
        # pinned copy each instead of one per branch: 24 branches per pass made the step host-bound (0.6 ms per pin_memory + 0.7 ms per
        # copy with the launch thread ahead of the device: profiles/r05_host_profile_dp02.log).  numpy on purpose: a torch CPU op over
        # more than 32 768 elements opens an OpenMP region, and waking the thread pool costs the launch thread ~0.5 ms per op.
        staged: Dict[int, Tensor] = {}
        if drop_plan is not None and self.dev.type == "cuda":
            sub = [(j, e[1]) for j, e in enumerate(drop_plan) if e is not None and e[0] == "subset"]
            per = [(j, e[1]) for j, e in enumerate(drop_plan) if e is not None and e[0] == "persample"]
            if sub:
                parts = [(v.numpy().astype(np.int64)[:, None] * N + tok_np[None, :]).reshape(-1) for _, v in sub]
                flat = ops.h2d(torch.from_numpy(np.concatenate(parts)), self.dev)
                o = 0
                for (j, _), prt in zip(sub, parts):
                    staged[j] = flat[o:o + prt.size]
                    o += prt.size
            if per:
                flat = ops.h2d(torch.from_numpy(np.concatenate([np.repeat(v.to(torch.float32).numpy(), N) for _, v in per])), self.dev)
                for k, (j, _) in enumerate(per):
                    staged[j] = flat[k * T:(k + 1) * T]

        def branch_setup(entry: Any, s: str, which: str, xin: Tensor, j: int = -1) -> Dict[str, Any]:
            """Decide the rows a residual branch runs on: all T rows, or the gathered rows of a batch subset."""
            br: Dict[str, Any] = {"mode": "plain", "rows": T, "nb": B, "x": xin, "rowscale": None, "scale": 1.0}
            if entry is None:
                return br
            kind, val = entry
            if kind == "rows":        # explicit token rows, branch at full scale (last_mlp_rows)
                ridx, Rr = val
                xs = ws.get(s + which + ".xs", (T, D), torch.float32)[:Rr]
                ops.gather_rows(xin, D, ridx, Rr, D, out_f32=xs)
                br.update(mode="subset", rows=Rr, nb=0, x=xs, idx=ridx, scale=1.0)
                return br
            if kind == "persample":   # DropPath: per-image mask/keep expanded over the image's tokens
                br["mode"] = "persample"
                br["rowscale"] = staged[j] if j in staged else ops.h2d(torch.from_numpy(np.repeat(val.to(torch.float32).numpy(), N)), self.dev)
                return br
            sb = int(val.numel())      # batch-subset stochastic depth
            Ts = sb * N
            idx = staged[j] if j in staged else ops.h2d(torch.from_numpy((val.numpy().astype(np.int64)[:, None] * N + tok_np[None, :]).reshape(-1)), self.dev)
            xs = ws.get(s + which + ".xs", (T, D), torch.float32)[:Ts]
            ops.gather_rows(xin, D, idx, Ts, D, out_f32=xs)
            br.update(mode="subset", rows=Ts, nb=sb, x=xs, idx=idx, scale=B / sb)
            return br

        def ln1_buffers(prefix: str) -> Dict[str, Any]:
            """Where a block keeps its first LayerNorm (operand of the qkv GEMM and of its weight gradient, row statistics for backward)."""
            return dict(out=ws.get(prefix + "ln1", (T, D), torch.bfloat16, pad_rows=64), mean=ws.get(prefix + "mean1", (T,), torch.float32),
                        rstd=ws.get(prefix + "rstd1", (T,), torch.float32))

        def run_block(i: int, x: Tensor, prefix: str, save: bool, keep_out: bool, ln1_done: bool = False, next_ln: Optional[Dict[str, Any]] = None):
            """One transformer block.  `prefix`: workspace names of its saved activations; `keep_out`: give the block output its
            own per-block buffer (it is the next block's input: the only activation kept under activation checkpointing).
            Round 6 -- the LayerNorm that follows a residual GEMM is handed to the GEMM call (lt_gemm_desc.ln_*: the library issues it; a kernel
            that normalised inside the GEMM was measured slower, profiles/r06_rowln_probe.md): norm2 behind the attention projection, and the
            NEXT block's norm1 behind fc2 (`next_ln`: that block's buffers and parameters; it is then called with `ln1_done`)."""
            s = prefix
            pre = f"blocks.{i}."
            g1 = self.w(pre + "ls1.gamma") if self.has(pre + "ls1.gamma") else None
            g2 = self.w(pre + "ls2.gamma") if self.has(pre + "ls2.gamma") else None
            e1 = drop_plan[2 * i] if drop_plan is not None else None
            e2 = drop_plan[2 * i + 1] if drop_plan is not None else None
            if i == cfg.depth - 1 and e2 is None and last_mlp_rows is not None and 0 < last_mlp_rows[1] < T:
                e2 = ("rows", last_mlp_rows)
            # ---------------- attention branch
            a = branch_setup(e1, s, "a", x, 2 * i)
            R, nb = a["rows"], a["nb"]
            lb1 = ln1_buffers(s)
            ln1 = lb1["out"]
            a["mean"], a["rstd"] = lb1["mean"], lb1["rstd"]
            if not ln1_done:
                ops.layernorm_fwd(a["x"], self.w(pre + "norm1.weight"), self.w(pre + "norm1.bias"), R, D, y_bf16=ln1, mean=a["mean"], rstd=a["rstd"], eps=cfg.ln_eps)
            else:
                assert a["mode"] == "plain"
            qkv = ws.get(s + "qkv", (T, 3 * D), torch.bfloat16)
            ops.gemm(ln1, self.wb(pre + "attn.qkv.weight"), qkv, M=R, N=3 * D, K=D, epilogue=ops.EPI_BF16, bias=self.w(pre + "attn.qkv.bias"))
            att = ws.get(s + "att", (T, D), torch.bfloat16, pad_rows=64)
            lse = ws.get(s + "lse", (B, Hh, N), torch.float32)
            if rope is not None:
                ops.rope_apply(qkv, rope[i][0], rope[i][1], nb, N, Hh, dh, 1 + n_reg)
            ops.attention_fwd(qkv, att, lse, nb, N, Hh, dh, scale)
            y1 = None   # the LayerScale gradient comes from the weight gradient (ops.layerscale_dgamma): no saved branch output
            ln2_fused: Optional[Dict[str, Any]] = None
            if a["mode"] == "subset":
                delta = ws.get(tag + ".delta", (T, D), torch.float32)
                ops.gemm(att, self.wb(pre + "attn.proj.weight"), delta, M=R, N=D, K=D, epilogue=ops.EPI_RESID, bias=self.w(pre + "attn.proj.bias"),
                         gamma=g1, resid=None, out2=y1, branch_scale=a["scale"])
                ops.scatter_add_rows(delta, a["idx"], x, D, R, D)   # x += (b/s) * g1 * branch on the subset rows, in place
                xm = x
            elif e2 is not None and e2[0] == "rows" and a["rowscale"] is None and os.environ.get("LT_SPARSE_LAST_PROJ", "1") != "0":
                # last block, output read at `ridx` only: the attention projection + LayerScale + residual are row-local too -- R rows of
                # them, written into a block-middle tensor that is FINITE everywhere else (the final LayerNorm still runs densely)
                ridx, Rr = e2[1]
                att_r = ws.get(s + "att_r", (T, D), torch.bfloat16, pad_rows=64)
                words = att.element_size() * D // 4    # a row as 32-bit words (bf16: D / 2): the row gather moves words, whatever they hold
                ops.gather_rows(att.view(torch.float32), words, ridx, Rr, words, out_f32=att_r.view(torch.float32))
                x_r = ws.get(tag + ".x_r", (T, D), torch.float32)
                ops.gather_rows(x, D, ridx, Rr, D, out_f32=x_r)
                xm_r = ws.get(tag + ".xm_r", (T, D), torch.float32)
                ops.gemm(att_r, self.wb(pre + "attn.proj.weight"), xm_r, M=Rr, N=D, K=D, epilogue=ops.EPI_RESID, bias=self.w(pre + "attn.proj.bias"),
                         gamma=g1, resid=x_r, out2=y1)
                # (finite, not zero, is what the other rows have to be: the buffer is zero-filled when it is allocated and only ever holds block
                # outputs after that -- a per-step fill of it was 0.09 ms of HBM writes per pass, and under this schedule every kernel's time
                # shows in the step one for one, profiles/r05_sensitivity.md; rows outside `ridx` give exact zeros in backward either way)
                xm = ws.get(s + "xm" if save else (tag + ".xb"), (T, D), torch.float32, zero=True)
                xm.index_copy_(0, ridx[:Rr], xm_r[:Rr])
                a["proj_rows"] = dict(idx=ridx, R=Rr, att_r=att_r)
            else:
                xm = ws.get(s + "xm" if save else (tag + ".xb"), (T, D), torch.float32)
                if e2 is None and self.fuse_ln:   # the MLP branch runs on every row: its LayerNorm rides behind the projection GEMM
                    ln2_fused = dict(weight=self.w(pre + "norm2.weight"), bias=self.w(pre + "norm2.bias"), eps=cfg.ln_eps,
                                     out=ws.get(s + "ln2", (T, D), torch.bfloat16, pad_rows=64), mean=ws.get(s + "mean2", (T,), torch.float32),
                                     rstd=ws.get(s + "rstd2", (T,), torch.float32))
                ops.gemm(att, self.wb(pre + "attn.proj.weight"), xm, M=T, N=D, K=D, epilogue=ops.EPI_RESID, bias=self.w(pre + "attn.proj.bias"),
                         gamma=g1, resid=x, out2=y1, rowscale=a["rowscale"], ln=ln2_fused)
            # ---------------- MLP branch
            m = branch_setup(e2, s, "m", xm, 2 * i + 1)
            R2 = m["rows"]
            ln2 = ws.get(s + "ln2", (T, D), torch.bfloat16, pad_rows=64)
            m["mean"], m["rstd"] = ws.get(s + "mean2", (T,), torch.float32), ws.get(s + "rstd2", (T,), torch.float32)
            if ln2_fused is None:
                ops.layernorm_fwd(m["x"], self.w(pre + "norm2.weight"), self.w(pre + "norm2.bias"), R2, D, y_bf16=ln2, mean=m["mean"], rstd=m["rstd"], eps=cfg.ln_eps)
            act = ws.get(s + "act", (T, hid), torch.bfloat16, pad_rows=64)
            if cfg.swiglu:   # w12 -> silu(x1) * x2 -> w3
                hpre = ws.get(s + "hpre", (T, 2 * hid), torch.bfloat16)
                ops.gemm(ln2, self.wb(pre + "mlp.w12.weight"), hpre, M=R2, N=2 * hid, K=D, epilogue=ops.EPI_BF16, bias=self.w(pre + "mlp.w12.bias"))
                ops.swiglu_fwd(hpre, act, R2, hid)
            else:
                hpre = ws.get(s + "hpre", (T, hid), torch.bfloat16) if save else None
                ops.gemm(ln2, self.wb(pre + "mlp.fc1.weight"), act, M=R2, N=hid, K=D, epilogue=ops.EPI_BF16_GELU, bias=self.w(pre + "mlp.fc1.bias"), out2=hpre)
            y2 = None
            if m["mode"] == "subset":
                delta = ws.get(tag + ".delta", (T, D), torch.float32)
                ops.gemm(act, self.wb(pre + fc2 + ".weight"), delta, M=R2, N=D, K=hid, epilogue=ops.EPI_RESID, bias=self.w(pre + fc2 + ".bias"),
                         gamma=g2, resid=None, out2=y2, branch_scale=m["scale"])
                ops.scatter_add_rows(delta, m["idx"], xm, D, R2, D)
                xo = xm
            else:
                xo = ws.get(f"{tag}.b{i}.xo" if keep_out else (tag + ".xa"), (T, D), torch.float32)
                ops.gemm(act, self.wb(pre + fc2 + ".weight"), xo, M=T, N=D, K=hid, epilogue=ops.EPI_RESID, bias=self.w(pre + fc2 + ".bias"),
                         gamma=g2, resid=xm, out2=y2, rowscale=m["rowscale"], ln=next_ln)
                next_ln = None     # consumed: the next block's norm1 is in its buffers
            assert next_ln is None, "the caller asked for the next block's LayerNorm behind an fc2 GEMM that does not run on every row"
            a.update(ln=ln1, qkv=qkv, att=att, lse=lse, y=y1)
            m.update(ln=ln2, act=act, hpre=hpre, y=y2)
            return xo, a, m
        ctx["run_block"] = run_block
        blocks: List[Dict[str, Any]] = []
        block_in: List[Tensor] = []
        # get_intermediate_layers (vision_transformer.py:386-480): outputs of the listed blocks, through the final norm
        cap_set = set(int(c) for c in capture_layers) if capture_layers is not None else set()
        captured: Dict[int, Tensor] = {}

        def capture(i: int, xb: Tensor) -> None:
            out = ws.get(f"{tag}.cap{i}", (B, N, D), torch.float32)
            if capture_norm:
                ops.layernorm_fwd(xb, self.w("norm.weight"), self.w("norm.bias"), T, D, y_f32=out, mean=ws.get(tag + ".capm", (T,), torch.float32),
                                  rstd=ws.get(tag + ".capr", (T,), torch.float32), eps=cfg.ln_eps)
            else:
                out.view(T, D).copy_(xb)
            captured[i] = out

        # HIP-graph replay of the leading STATIC blocks (no stochastic-depth draws, no rotary tables that change per step, not the last
        # block when it runs on the rows the losses read, no activation checkpointing, no intermediate captures): ~7 launches per block
        # become one graph launch per pass.  Buffers are the workspace's named allocations and the weights are views of the flat storage,
        # so every pointer a captured kernel holds stays valid; the first two calls of a shape run eagerly (they also allocate).
        n_graph = 0
        if (self.graph_forward and x.is_cuda and drop_plan is None and rope is None and not (save and checkpoint) and not cap_set):
            n_graph = cfg.depth - 1 if (last_mlp_rows is not None and 0 < last_mlp_rows[1] < T) else cfg.depth
        if n_graph > 0:
            x, gblocks = self._graphed_blocks(tag, x, save, n_graph, run_block)
            if save:
                blocks.extend(gblocks)
        ln1_done = False
        # launch-plan replay of the same leading static blocks (the shipped path: same eligibility as the graph, but the LayerNorms stay behind
        # their GEMM calls and the pass keeps its stream): eager on the first two passes of a geometry (they allocate), logged on the third
        n_plan, pent, recorder, rplan, gen_start = 0, None, None, None, -1
        if (self.plan_forward and ops.plan_replay_enabled and n_graph == 0 and x.is_cuda and drop_plan is None and rope is None and not (save and checkpoint) and not cap_set):
            n_plan = cfg.depth - 1 if (last_mlp_rows is not None and 0 < last_mlp_rows[1] < T) else cfg.depth
        if n_plan > 0:
            pkey = (tag, tuple(x.shape), bool(save), n_plan, x.data_ptr(), torch.cuda.current_stream().cuda_stream, self.fuse_ln)
            pent = self._fwd_plans.setdefault(pkey, {"calls": 0, "plan": None, "gen": -1, "replays": 0})
            if pent["plan"] is not None and pent["gen"] != ws.generation:     # a buffer moved since the log was written
                pent.update(calls=2, plan=None)
            if pent["plan"] is not None:
                pent["plan"].replay()
                pent["replays"] += 1
                x, ln1_done = pent["out"], pent["ln1_done"]
                if save:
                    blocks.extend({"attn": dict(b["attn"]), "mlp": dict(b["mlp"])} for b in pent["blocks"])
            else:
                pent["calls"] += 1
                if pent["calls"] >= 3:
                    gen_start = ws.generation
                    recorder = ops.record_plan()
                    rplan = recorder.__enter__()

        def end_recording(x_now: Tensor, ln1_now: bool) -> None:
            """The static blocks are through: keep their log with what the eager pass hands on (unless a buffer was allocated meanwhile)."""
            recorder.__exit__(None, None, None)
            if ws.generation == gen_start:
                pent.update(plan=rplan, gen=ws.generation, out=x_now, ln1_done=ln1_now,
                            blocks=[{"attn": dict(b["attn"]), "mlp": dict(b["mlp"])} for b in blocks] if save else [])

        first = n_plan if (pent is not None and pent["plan"] is not None) else n_graph
        for i in range(first, cfg.depth):
            if recorder is not None and i == n_plan:
                end_recording(x, ln1_done)
                recorder = None
            # the NEXT block's norm1 behind this block's fc2 GEMM: both on every row (no stochastic-depth draw on either branch, not the last
            # block's loss-row MLP), buffers of the next block's prefix
            nxt = None
            if (self.fuse_ln and not (save and checkpoint) and n_graph == 0 and i + 1 < cfg.depth and not cap_set
                    and (drop_plan is None or (drop_plan[2 * i + 1] is None and drop_plan[2 * (i + 1)] is None))
                    and not (i == cfg.depth - 1 and last_mlp_rows is not None)):
                nb_ = ln1_buffers(f"{tag}.b{i + 1}." if save else f"{tag}.tmp.")
                nxt = dict(weight=self.w(f"blocks.{i + 1}.norm1.weight"), bias=self.w(f"blocks.{i + 1}.norm1.bias"), eps=cfg.ln_eps, **nb_)
            if save and checkpoint:
                # activation checkpointing (reference _activation_checkpointing.py): keep only the block input, recompute the
                # block in backward.  Subset stochastic depth updates x in place, so the input is copied aside first.
                e1 = drop_plan[2 * i] if drop_plan is not None else None
                if e1 is not None and e1[0] != "persample":
                    xin = ws.get(f"{tag}.b{i}.xin", (T, D), torch.float32)
                    xin.copy_(x)
                    block_in.append(xin)
                else:
                    block_in.append(x)
                x, _, _ = run_block(i, x, f"{tag}.ck.", False, True)
            else:
                x, a_, m_ = run_block(i, x, f"{tag}.b{i}." if save else f"{tag}.tmp.", save, save, ln1_done=ln1_done, next_ln=nxt)
                ln1_done = nxt is not None
                if save:
                    blocks.append({"attn": a_, "mlp": m_})
            if i in cap_set:
                capture(i, x)
        if recorder is not None:     # every block was static
            end_recording(x, ln1_done)
            recorder = None
        ctx["block_in"] = block_in if (save and checkpoint) else None
        ctx["captured"] = captured
        xn = ws.get(tag + ".xn", (B, N, D), torch.float32)
        mean = ws.get(tag + ".meanf", (T,), torch.float32)
        rstd = ws.get(tag + ".rstdf", (T,), torch.float32)
        ops.layernorm_fwd(x, self.w("norm.weight"), self.w("norm.bias"), T, D, y_f32=xn, mean=mean, rstd=rstd, eps=cfg.ln_eps)
        ctx.update(blocks=blocks, x_last=x, meanf=mean, rstdf=rstd, xn=xn)
        return ctx

    # ---- backward -------------------------------------------------------------------------------
    def backward(self, ws: Workspace, ctx: Dict[str, Any], dxn: Tensor, side: Optional["torch.cuda.Stream"] = None) -> None:
        """Run `backward_iter` to completion on the current stream (the LayerScale gradients still need
        `finish_layerscale_grads` once all passes of the step are done)."""
        for _ in self.backward_iter(ws, ctx, dxn, side):
            pass

    def finish_layerscale_grads(self, blocks: Optional[Iterable[int]] = None, last_call: bool = True) -> None:
        """LayerScale gradients from the accumulated weight gradients: dgamma = (rowdot(W, dW) + b * db) / gamma
        (layer_scale.py:27-28 backward without saving the branch outputs).  Call once per step, after every backward pass
        (global and local crops) and its weight-gradient GEMMs have been enqueued / joined, before the optimizer.
        `blocks`: only these blocks (a data-parallel caller finishes a block as soon as both passes are through it, so that
        its gradient all-reduce can start during backward, `last_call=False`) -- every block exactly once per step."""
        cfg = self.cfg
        D, hid = cfg.embed_dim, cfg.hidden
        fc2 = "mlp.w3" if cfg.swiglu else "mlp.fc2"
        if cfg.rope_base is not None and last_call:
            self.gw("pos_embed").zero_()            # no positional embedding in a RoPE model: the zero table stays zero
        todo = list(range(cfg.depth) if blocks is None else blocks)
        if len(todo) == cfg.depth and cfg.depth > 1 and not cfg.mask_k_bias and self.has("blocks.0.ls1.gamma"):
            # every block at once: the blocks are laid out one after the other with equal tensor sizes, so layer i's tensors start a
            # constant number of elements behind layer i - 1's in the parameter, gradient and bf16-shadow storages alike -- two launches
            # instead of 2 x depth at the tail of the step, where nothing else is running
            off = self.P.offsets
            names = [t + s_ for t in ("attn.proj", fc2) for s_ in (".weight", ".bias")] + ["ls1.gamma", "ls2.gamma"]
            strides = [{off[self.prefix + f"blocks.{i + 1}." + n] - off[self.prefix + f"blocks.{i}." + n] for n in names} for i in range(cfg.depth - 1)]
            if all(len(s_) == 1 for s_ in strides) and len({next(iter(s_)) for s_ in strides}) == 1:
                stride = next(iter(strides[0]))
                for gname, lin, k_in in (("blocks.0.ls1.gamma", "blocks.0.attn.proj", D), ("blocks.0.ls2.gamma", "blocks.0." + fc2, hid)):
                    ops.layerscale_dgamma_batched(self.wb(lin + ".weight"), self.gw(lin + ".weight"), self.w(lin + ".bias"), self.gw(lin + ".bias"),
                                                  self.w(gname), self.gw(gname), D, k_in, cfg.depth, stride)
                return
        for i in todo:
            pre = f"blocks.{i}."
            if cfg.mask_k_bias:
                self.gw(pre + "attn.qkv.bias")[D:2 * D].zero_()   # LinearKMaskedBias: bias * mask => no gradient for the K third
            for gname, lin, k_in in ((pre + "ls1.gamma", pre + "attn.proj", D), (pre + "ls2.gamma", pre + fc2, hid)):
                if self.has(gname):
                    ops.layerscale_dgamma(self.wb(lin + ".weight"), self.gw(lin + ".weight"), self.w(lin + ".bias"), self.gw(lin + ".bias"),
                                          self.w(gname), self.gw(gname), D, k_in)

    def backward_iter(self, ws: Workspace, ctx: Dict[str, Any], dxn: Tensor, side: Optional["torch.cuda.Stream"] = None,
                      joint: Optional[JointWgrad] = None, stop_after: Optional[int] = None,
                      resume: Optional[Dict[str, Any]] = None) -> Iterator[str]:
        """dxn f32 [B,N,D] = dL/d(final-norm tokens).  Accumulates into the FlatParams grad views.

        Generator: yields "block" after enqueuing each transformer block and "tail" before the token-assembly /
        patch-embedding part, so that a caller can interleave the launches of two independent backward passes (global and
        local crops) on two streams.  Everything before "tail" only uses atomics or side-stream-ordered accumulations into
        the shared gradient buffer; the tail does plain read-modify-writes and must run after the other pass's tail.

        `side`: optional second HIP stream for the weight-gradient GEMMs and bias column sums.  They depend only on
        tensors the main (dgrad) chain has already produced and feed nothing but the optimizer, so running them beside
        the dgrad chain fills the CUs the 591-tile dgrad GEMMs leave idle in their last wave.  The caller must make the
        optimizer wait for `side`.

        `joint`: full-row weight gradients are deposited there instead of launched (`JointWgrad`); the upstream-gradient buffers then
        rotate through four allocations instead of two and the qkv data gradient gets a buffer of its own, so that no deposited
        operand is overwritten within the block iteration that deposited it.

        HIP-graph replay of the static blocks (the caller captures the launches of blocks depth-2 .. 0 once and replays them):
        `stop_after` = n ends the generator after n blocks; `resume` = the state a full run left in ctx["_bwd_state"] skips the final
        norm and the block loop and goes straight to "tail".  ctx["_bwd_consumed"] is the buffer -> event map of `before_write`, which
        the caller clears where it has joined the streams anyway (events recorded outside a capture must not be waited for inside it)."""
        cfg = self.cfg
        B, N, n_p, T, tag = ctx["B"], ctx["N"], ctx["n_p"], ctx["T"], ctx["tag"]
        D, Hh, dh, hid = cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden
        scale = dh ** -0.5
        dxa = ws.get(tag + ".dxa", (T, D), torch.float32)
        dxb = ws.get(tag + ".dxb", (T, D), torch.float32)
        if joint is None:
            dDs = [ws.get(tag + ".dD", (T, D), torch.bfloat16, pad_rows=64), ws.get(tag + ".dDb", (T, D), torch.bfloat16, pad_rows=64)]  # upstream grads of
            # consecutive branches alternate between two buffers: the LayerNorm backward of one branch writes the next one's
            dD3 = None
        else:
            dDs = [ws.get(tag + n, (T, D), torch.bfloat16, pad_rows=64) for n in (".dD0", ".dD1", ".dD2r", ".dD3r")]
            dD3 = ws.get(tag + ".dD3", (T, D), torch.bfloat16, pad_rows=64)   # qkv data gradient (not a weight-gradient operand)
        nring = len(dDs)
        dD2 = ws.get(tag + ".dD2", (T, D), torch.bfloat16, pad_rows=64)
        dH = ws.get(tag + ".dH", (T, 2 * hid if cfg.swiglu else hid), torch.bfloat16, pad_rows=64)
        dAct = ws.get(tag + ".dAct", (T, hid), torch.bfloat16, pad_rows=64) if cfg.swiglu else None
        fc1, fc2 = ("mlp.w12", "mlp.w3") if cfg.swiglu else ("mlp.fc1", "mlp.fc2")
        hid1 = 2 * hid if cfg.swiglu else hid
        dQ = ws.get(tag + ".dQ", (T, 3 * D), torch.bfloat16, pad_rows=64)
        aws = ws.get(tag + ".attn_ws", (ops.attention_bwd_ws_floats(B, N, Hh, dh),), torch.float32)

        blocks_ctx = ctx["blocks"]
        ckpt = ctx.get("block_in") is not None   # activation checkpointing: blocks are recomputed one at a time below

        def fuse_args(br: Optional[Dict[str, Any]], gname: str, bname: str, buf: Tensor) -> Dict[str, Any]:
            """LayerNorm-backward arguments that also produce the upstream gradient of branch `br` (not in subset mode, and
            not under checkpointing, where the next branch's bookkeeping does not exist yet)."""
            if br is None or br["mode"] == "subset":
                return {}
            return dict(dnext=buf, gamma_next=self.w(gname) if self.has(gname) else None, rowscale_next=br["rowscale"],
                        scale_next=float(br["scale"]), dbias_next=self.gw(bname))

        fc2n = "mlp.w3" if cfg.swiglu else "mlp.fc2"
        cur = 0   # index into dDs of the branch about to be processed
        last = f"blocks.{cfg.depth - 1}."
        slab = ws.get("wgrad.slabs", (32 * 1024 * 1024,), torch.float32)  # 128 MiB split-K scratch (deterministic reduction)
        main = torch.cuda.current_stream()
        consumed: Dict[int, Any] = {}  # buffer data_ptr -> event after which the side stream no longer reads it
        ctx["_bwd_consumed"] = consumed
        dx = dxa
        other = dxb
        have = False
        if resume is None:
            nxt = fuse_args(None if ckpt else blocks_ctx[-1]["mlp"], last + "ls2.gamma", last + fc2n + ".bias", dDs[cur])
            ops.layernorm_bwd(ctx["x_last"], self.w("norm.weight"), ctx["meanf"], ctx["rstdf"], dxn, None, dxa,
                              self.gw("norm.weight"), self.gw("norm.bias"), T, D, **nxt)
            have = bool(nxt)   # dDs[cur] already holds the upstream gradient of the branch about to be processed
        else:
            dx = resume["dx"]

        def before_write(buf: Tensor) -> None:
            ev = consumed.pop(buf.data_ptr(), None)
            if isinstance(ev, str):     # a joint deposit still waiting for the other pass: launch it alone
                joint.flush(only=ev)
                ev = consumed.pop(buf.data_ptr(), None)
            if ev is not None:
                main.wait_event(ev)

        def wgrad(dy: Tensor, xin: Tensor, wname: str, n_out: int, k_in: int, rows: int, bias: Optional[str] = None) -> None:
            tiles = ((n_out + 127) // 128) * ((k_in + 127) // 128)
            kpad = (rows + 63) // 64 * 64
            dyp, xp = padded_rows(dy, rows), padded_rows(xin, rows)
            if kpad != rows and dyp is not None and xp is not None:
                # zero the <=63 pad rows so the contraction can run in whole 64-row k-tiles (both operands: stale pad rows could hold NaN bit
                # patterns); torch fills, logged when a launch plan is being recorded
                ops.recordable(lambda a_=dyp[rows:kpad], b_=xp[rows:kpad]: (a_.zero_(), b_.zero_()))
                dy, xin = dyp, xp
            else:
                kpad = rows

            def run(dy_: Tensor = dy, xin_: Tensor = xin, k_: int = kpad) -> None:
                # the bias gradient (column sums of dy) rides the weight-gradient GEMM, which holds dy's fragments anyway (pad rows are zero)
                ops.gemm(dy_, xin_, self.gw(wname), M=n_out, N=k_in, K=k_, trans_a=True, trans_b=True, epilogue=ops.EPI_F32_ACCUM,
                         lda=n_out, ldb=k_in, ldc=k_in, workspace=slab, colsum=self.gw(bias) if bias is not None else None,
                         **split_k_plan(n_out, k_in, k_, True, _split_k(tiles, k_)))

            if side is None:
                run()
                return
            if joint is not None and rows == T and kpad == rows:
                joint.deposit(tag, wname, dy, xin, rows, run, consumed)
                return
            side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                run()
                consumed[dy.data_ptr()] = side.record_event()

        def ln_rows(br: Dict[str, Any], norm: str, dy: Tensor, R: int) -> None:
            """LayerNorm backward of a branch that ran on the rows br["idx"]: dx[idx] += LN'(dy), in place through the row index (the kernel's
            indexed form serves D <= 1024; wider models take the compact result + scatter-add pass).  Chosen on performance alone since round 6:
            round 5 kept the two-pass form for the last block's loss-row branch because the 100-step KoLeo-on trajectory tests asserted a
            hard 1e-3 that held for one rounding draw only (the two forms differ by one ulp in 4 % of the elements); those tests now assert the
            bound the data supports -- inside the reference's own bf16-autocast deviation (tests/test_gpu_step.py)."""
            if D <= 1024 and D % 4 == 0:
                ops.layernorm_bwd(br["x"], self.w(norm + ".weight"), br["mean"], br["rstd"], dy, dx, dx, self.gw(norm + ".weight"), self.gw(norm + ".bias"),
                                  R, D, ridx=br["idx"])
                return
            lng = ws.get(tag + ".lng", (T, D), torch.float32)[:R]
            ops.layernorm_bwd(br["x"], self.w(norm + ".weight"), br["mean"], br["rstd"], dy, None, lng, self.gw(norm + ".weight"), self.gw(norm + ".bias"), R, D)
            ops.scatter_add_rows(lng, br["idx"], dx, D, R, D)

        for i in (reversed(range(cfg.depth)) if resume is None else ()):
            main = torch.cuda.current_stream()   # (re-read per block: under a graph capture the chain runs on the capturing stream)
            if ckpt:
                if side is not None:
                    main.wait_stream(side)   # the previous block's weight-gradient GEMMs still read the shared activation buffers
                _, a, m = ctx["run_block"](i, ctx["block_in"][i], f"{tag}.ck.", True, False)
            else:
                blk = ctx["blocks"][i]
                a, m = blk["attn"], blk["mlp"]
            pre = f"blocks.{i}."
            g1 = self.w(pre + "ls1.gamma") if self.has(pre + "ls1.gamma") else None
            g2 = self.w(pre + "ls2.gamma") if self.has(pre + "ls2.gamma") else None
            # ---- MLP branch: xo = xm + scale * g2 * (fc2(gelu(fc1(ln2(rows)))))
            R2 = m["rows"]
            dD = dDs[cur]
            if not have:   # subset rows (read through their row index: no gathered copy), or the producing LayerNorm backward ran on a subset
                before_write(dD)
                ops.layerscale_bwd(dx, None, g2, dD, None, R2, D, dbias=self.gw(pre + fc2 + ".bias"), rowscale=m["rowscale"], scale=m["scale"],
                                   ridx=m["idx"] if m["mode"] == "subset" else None)
            wgrad(dD, m["act"], pre + fc2 + ".weight", D, hid, R2)
            before_write(dH)
            if cfg.swiglu:
                ops.gemm(dD, self.wb(pre + fc2 + ".weight"), dAct, M=R2, N=hid, K=D, trans_b=True, epilogue=ops.EPI_BF16)
                ops.swiglu_bwd(m["hpre"], dAct, dH, R2, hid)
            else:
                ops.gemm(dD, self.wb(pre + fc2 + ".weight"), dH, M=R2, N=hid, K=D, trans_b=True, epilogue=ops.EPI_BF16_GELUGRAD, aux=m["hpre"])
            wgrad(dH, m["ln"], pre + fc1 + ".weight", hid1, D, R2, bias=pre + fc1 + ".bias")
            ops.gemm(dH, self.wb(pre + fc1 + ".weight"), dD2, M=R2, N=D, K=hid1, trans_b=True, epilogue=ops.EPI_BF16)
            if m["mode"] == "subset":
                # dx[idx] += LN'(.) on the subset rows, in place through the row index (no compact result + scatter-add pass); identity path untouched
                ln_rows(m, pre + "norm2", dD2, R2)
                have = False
            else:
                before_write(dDs[(cur + 1) % nring])
                nxt = fuse_args(a, pre + "ls1.gamma", pre + "attn.proj.bias", dDs[(cur + 1) % nring])
                ops.layernorm_bwd(m["x"], self.w(pre + "norm2.weight"), m["mean"], m["rstd"], dD2, dx, other,
                                  self.gw(pre + "norm2.weight"), self.gw(pre + "norm2.bias"), T, D, **nxt)
                dx, other = other, dx
                have = bool(nxt)
            cur = (cur + 1) % nring
            # ---- attention branch: xm = x + scale * g1 * proj(attn(qkv(ln1(rows))))
            R1, nb = a["rows"], a["nb"]
            dD = dDs[cur]
            pr = a.get("proj_rows")
            if pr is not None:
                # the projection ran on `R` rows (forward): its gradients come from those rows of dx; d(att) is zero elsewhere
                assert not have
                Rr, ridx = pr["R"], pr["idx"]
                before_write(dD)
                ops.layerscale_bwd(dx, None, g1, dD, None, Rr, D, dbias=self.gw(pre + "attn.proj.bias"), rowscale=None, scale=1.0, ridx=ridx)
                wgrad(dD, pr["att_r"], pre + "attn.proj.weight", D, D, Rr)
                dAr = ws.get(tag + ".dAr", (T, D), torch.bfloat16, pad_rows=64)
                ops.gemm(dD, self.wb(pre + "attn.proj.weight"), dAr, M=Rr, N=D, K=D, trans_b=True, epilogue=ops.EPI_BF16)
                dD2.zero_()
                dD2.index_copy_(0, ridx[:Rr], dAr[:Rr])
            else:
                if not have:
                    before_write(dD)
                    ops.layerscale_bwd(dx, None, g1, dD, None, R1, D, dbias=self.gw(pre + "attn.proj.bias"), rowscale=a["rowscale"], scale=a["scale"],
                                       ridx=a["idx"] if a["mode"] == "subset" else None)
                wgrad(dD, a["att"], pre + "attn.proj.weight", D, D, R1)
                ops.gemm(dD, self.wb(pre + "attn.proj.weight"), dD2, M=R1, N=D, K=D, trans_b=True, epilogue=ops.EPI_BF16)
            before_write(dQ)
            ops.attention_bwd(a["qkv"], a["att"], dD2, a["lse"], aws, dQ, nb, N, Hh, dh, scale)
            if ctx.get("rope") is not None:   # gradients w.r.t. the un-rotated q / k: transposed rotation
                ops.rope_apply(dQ, ctx["rope"][i][0], ctx["rope"][i][1], nb, N, Hh, dh, 1 + cfg.num_register_tokens, inverse=True)
            wgrad(dQ, a["ln"], pre + "attn.qkv.weight", 3 * D, D, R1, bias=pre + "attn.qkv.bias")
            dX = dD if dD3 is None else dD3   # d(ln1 output): in place of the branch's upstream gradient, or its own buffer under `joint`
            before_write(dX)
            ops.gemm(dQ, self.wb(pre + "attn.qkv.weight"), dX, M=R1, N=D, K=3 * D, trans_b=True, epilogue=ops.EPI_BF16)
            if a["mode"] == "subset":
                ln_rows(a, pre + "norm1", dX, R1)
                have = False
            else:
                nxt = {}
                if i > 0:
                    pp = f"blocks.{i - 1}."
                    before_write(dDs[(cur + 1) % nring])
                    nxt = fuse_args(None if ckpt else blocks_ctx[i - 1]["mlp"], pp + "ls2.gamma", pp + fc2n + ".bias", dDs[(cur + 1) % nring])
                ops.layernorm_bwd(a["x"], self.w(pre + "norm1.weight"), a["mean"], a["rstd"], dX, dx, other,
                                  self.gw(pre + "norm1.weight"), self.gw(pre + "norm1.bias"), T, D, **nxt)
                dx, other = other, dx
                have = bool(nxt)
            cur = (cur + 1) % nring
            yield "block"
            if stop_after is not None and cfg.depth - i >= stop_after:
                return

        if resume is None:
            ctx["_bwd_state"] = dict(dx=dx)
        yield "tail"
        main = torch.cuda.current_stream()   # the tail may be resumed on another stream than the block loop
        # ---- token assembly + patch embedding
        dpatch = ws.get(tag + ".dpatch", (B * n_p, D), torch.bfloat16)
        mp = self._pos_map(ctx["gh"], ctx["gw"])
        gpos = self.gw("pos_embed").view(-1, D)
        if mp is None:
            dpos = gpos
        else:
            dpos = ws.get(tag + ".dpos", (n_p + 1, D), torch.float32)
            dpos.zero_()
        n_reg = cfg.num_register_tokens
        ops.assemble_tokens_bwd(dx, ctx["masks"], dpatch, self.gw("cls_token").view(D), dpos, self.gw("mask_token").view(D), B, n_p, D,
                                dreg=self.gw("register_tokens").view(-1, D) if n_reg else None, n_reg=n_reg)
        if mp is not None:
            gpos[0].add_(dpos[0])  # cls position row (plumbing: one D-vector add)
            ops.matmul_f32(mp, dpos[1:], gpos[1:], mp.shape[1], D, n_p, trans_a=True, accumulate=True)
        tiles = ((D + 127) // 128) * ((self.kpad + 127) // 128)

        def patch_wgrad() -> None:
            gview = self.gw("patch_embed.proj.weight").view(D, -1)
            target = gview
            if self.kpad != self.kreal:  # accumulate into a padded scratch, then fold the real columns into the gradient
                target = ws.get(tag + ".dwpe_pad", (D, self.kpad), torch.float32)
                target.zero_()
            ops.gemm(dpatch, ctx["cols"], target, M=D, N=self.kpad, K=B * n_p, trans_a=True,
                     trans_b=True, epilogue=ops.EPI_F32_ACCUM, lda=D, ldb=self.kpad,
                     **split_k_plan(D, self.kpad, B * n_p, True, max(2, _split_k(tiles, B * n_p))),
                     ldc=self.kpad, workspace=slab, colsum=self.gw("patch_embed.proj.bias"))
            if self.kpad != self.kreal:
                ops.unpad_accumulate(target, gview, D, self.kreal, self.kpad)

        if side is None:
            patch_wgrad()
        else:
            side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                patch_wgrad()
