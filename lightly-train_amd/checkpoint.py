"""Checkpoint / export formats of the reference for the flat-storage method objects.

What the reference writes and reads:
  * `checkpoint["state_dict"]` = `method.state_dict()` of the Lightning module (LT/_checkpoint.py:101-123): keys
    `{teacher,student}_embedding_model.wrapped_model._model.<vit key>`, `{teacher,student}_head.{dino_head,ibot_head}.<head key>`,
    `dino_loss.center`, `ibot_loss.center` (LT/_methods/dinov2/dinov2.py:196-257); models built with `block_chunks > 0`
    (vitl14 / vitg14 YAMLs) name their blocks `blocks.<chunk>.<i>.` instead of `blocks.<i>.`
    (dinov2_vit_src/models/vision_transformer.py:160-172);
  * `checkpoint["optimizer_states"][0]` = `torch.optim.AdamW.state_dict()` over the fused parameter groups of
    `get_optimizer_with_decay` / `get_fused_param_groups` (LT/_methods/dinov2/utils.py:191-273): parameter indices run over the
    groups in order, a group holds its members in `named_parameters()` order (student backbone first, then the head);
  * the exported model = `torch.save(get_model().state_dict())` of the (EMA) teacher backbone
    (LT/_models/dinov2_vit/dinov2_vit_package.py:146-162).

Everything here is name / layout bookkeeping on torch tensors (plumbing): it runs on any device and is unit-tested on CPU."""
from __future__ import annotations

import re
from typing import Any, Dict, List, Mapping, Optional, Tuple

import torch
from torch import Tensor

from .params import FlatParams

_CHUNKED = re.compile(r"^blocks\.(\d+)\.(\d+)\.")
_PLAIN = re.compile(r"^blocks\.(\d+)\.")


def vit_key_to_flat(key: str) -> str:
    """`blocks.<chunk>.<i>.x` (block_chunks > 0: the inner index is the global block index, chunks are padded with Identity)
    -> `blocks.<i>.x`; every other key unchanged."""
    m = _CHUNKED.match(key)
    return f"blocks.{m.group(2)}." + key[m.end():] if m else key


def vit_key_from_flat(key: str, depth: int, block_chunks: int) -> str:
    """Inverse of `vit_key_to_flat` for a model of `depth` blocks split into `block_chunks` chunks (0 = unchunked)."""
    if not block_chunks:
        return key
    m = _PLAIN.match(key)
    if not m:
        return key
    i = int(m.group(1))
    chunk = i // (depth // block_chunks)
    return f"blocks.{chunk}.{i}." + key[m.end():]


def method_key_to_flat(key: str, separate_ibot: bool) -> Optional[Tuple[str, str]]:
    """reference method.state_dict() key -> (role, flat name) with role in {"student", "teacher"}; None for the keys that are
    aliases of shared tensors (ibot_head.* when the iBOT head IS the DINO head) or not parameters of the flat storage."""
    for role in ("student", "teacher"):
        pre = f"{role}_embedding_model.wrapped_model._model."
        if key.startswith(pre):
            return role, "backbone." + vit_key_to_flat(key[len(pre):])
        pre = f"{role}_head.dino_head."
        if key.startswith(pre):
            return role, "head." + key[len(pre):]
        pre = f"{role}_head.ibot_head."
        if key.startswith(pre):
            return (role, "ihead." + key[len(pre):]) if separate_ibot else None
        pre = f"{role}_paka_head."            # DINOv31 (LT/_methods/dinov31/dinov31.py:126-146)
        if key.startswith(pre):
            return role, "paka." + key[len(pre):]
    return None


def method_state_dict(student: FlatParams, teacher: FlatParams, centers: Mapping[str, Tensor], separate_ibot: bool,
                      depth: int, block_chunks: int = 0,
                      buffers: Optional[Mapping[Tuple[str, str], Mapping[str, Tensor]]] = None) -> Dict[str, Tensor]:
    """The reference's `method.state_dict()` from the flat storages (key order: teacher backbone, student backbone, teacher heads,
    student heads, loss centers -- the registration order of dinov2.py:196-257)."""
    out: Dict[str, Tensor] = {}

    def emit(role: str, fp: FlatParams, n: str, src: str, dst: str) -> None:
        out[f"{role}_head.{dst}.{n[len(src):]}"] = fp.p[n].detach().clone()
        # BatchNorm1d buffers follow their module's parameters (weight, bias, running_mean, running_var, num_batches_tracked)
        mod = n[len(src):].rsplit(".", 1)[0]
        bufs = (buffers or {}).get((role, src), {})
        if n.endswith(".bias") and f"{mod}.running_mean" in bufs:
            for suffix in ("running_mean", "running_var", "num_batches_tracked"):
                out[f"{role}_head.{dst}.{mod}.{suffix}"] = bufs[f"{mod}.{suffix}"].detach().clone()

    for role, fp in (("teacher", teacher), ("student", student)):
        for n in fp.names:
            if n.startswith("backbone."):
                out[f"{role}_embedding_model.wrapped_model._model.{vit_key_from_flat(n[9:], depth, block_chunks)}"] = fp.p[n].detach().clone()
    for role, fp in (("teacher", teacher), ("student", student)):
        for n in fp.names:
            if n.startswith("head."):
                emit(role, fp, n, "head.", "dino_head")
        # a shared head is one module registered under two names: both prefixes appear in the reference's state_dict
        src = "ihead." if separate_ibot else "head."
        for n in fp.names:
            if n.startswith(src):
                emit(role, fp, n, src, "ibot_head")
    for k, v in centers.items():
        out[k] = v.detach().clone()
    for role, fp in (("student", student), ("teacher", teacher)):       # DINOv31's PaKA heads: registered last, student first
        for n in fp.names:
            if n.startswith("paka."):
                out[f"{role}_paka_head.{n[5:]}"] = fp.p[n].detach().clone()
    return out


def load_method_state_dict(sd: Mapping[str, Tensor], student: FlatParams, teacher: FlatParams, separate_ibot: bool,
                           strict: bool = True) -> Dict[str, Tensor]:
    """Copy a reference `method.state_dict()` into the flat storages (fp32 master copies AND their bf16 shadows).  Returns the
    non-parameter entries (`dino_loss.center`, `ibot_loss.center`) for the caller.  strict: every flat tensor must be present
    with its shape, and every key must be understood.  Everything is validated before the first copy: a load that raises leaves the
    model as it was."""
    seen = {"student": set(), "teacher": set()}
    extra: Dict[str, Tensor] = {}
    fps = {"student": student, "teacher": teacher}
    plan: List[Tuple[FlatParams, str, Tensor]] = []
    for k, v in sd.items():
        hit = method_key_to_flat(k, separate_ibot)
        if hit is None:
            if k in ("dino_loss.center", "ibot_loss.center"):
                extra[k] = v
            elif strict and not (k.endswith("_input_mean") or k.endswith("_input_std") or ".ibot_head." in k):
                raise KeyError(f"unexpected key in state_dict: {k}")
            continue
        role, name = hit
        fp = fps[role]
        if name.endswith(("running_mean", "running_var", "num_batches_tracked")) and "head." in name:
            extra[k] = v   # BatchNorm1d buffers of the projection heads: the caller hands them to the head engines
            continue
        if name not in fp.p:
            if strict:
                raise KeyError(f"unexpected key in state_dict: {k} (-> {name})")
            continue
        if tuple(v.shape) != tuple(fp.shapes[name]):
            raise ValueError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(fp.shapes[name])}")
        plan.append((fp, name, v))
        seen[role].add(name)
    if strict:
        for role, fp in fps.items():
            missing = [n for n in fp.names if n not in seen[role]]
            if missing:
                raise KeyError(f"missing keys for the {role}: {missing[:5]}{' ...' if len(missing) > 5 else ''}")
    for fp, name, v in plan:
        fp.p[name].copy_(v.to(fp.device, torch.float32))
    for fp in fps.values():
        fp.bf16.copy_(fp.data)
    return extra


def fused_groups(param_groups: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
    """get_fused_param_groups (utils.py:253-273) over per-parameter entries {"name", "flat", "lr", "weight_decay", "head",
    "last_layer"}: groups in order of first appearance, named after their first member, members in appearance order."""
    fused: Dict[Tuple[Any, ...], Dict[str, Any]] = {}
    for g in param_groups:
        key = (g["lr"], g["weight_decay"], g["head"], g["last_layer"])
        if key not in fused:
            fused[key] = dict(g, members=[g["flat"]])
        else:
            fused[key]["members"].append(g["flat"])
    return list(fused.values())


def _view(flat: Tensor, fp: FlatParams, name: str) -> Tensor:
    o = fp.offsets[name]
    return flat[o:o + fp.p[name].numel()].view(fp.shapes[name])


def optimizer_state_dict(student: FlatParams, exp_avg: Tensor, exp_avg_sq: Tensor, opt_step: int,
                         param_groups: List[Dict[str, Any]], hyper: Dict[str, Any],
                         prefix_steps: Optional[Mapping[str, int]] = None) -> Dict[str, Any]:
    """`torch.optim.AdamW.state_dict()` of the reference's optimizer: {"state": {idx: {step, exp_avg, exp_avg_sq}},
    "param_groups": [{..., "params": [idx, ...]}]}.  `param_groups`: one entry per flat tensor (see `fused_groups`), `hyper`:
    the group fields shared by all groups (betas, eps, ...); per-group "lr" / "weight_decay" are the values currently in effect.
    `prefix_steps`: flat-name prefix -> the Adam step count of THOSE parameters where it differs from `opt_step` (torch counts per parameter
    and skips parameters that had no gradient: DINOv31's PaKA head before `paka_start_step`); a count of 0 writes no state, as torch does."""
    state: Dict[int, Dict[str, Tensor]] = {}

    def step_of(n: str) -> int:
        for pre, st in (prefix_steps or {}).items():
            if n.startswith(pre):
                return int(st)
        return int(opt_step)

    groups: List[Dict[str, Any]] = []
    idx = 0
    for g in fused_groups(param_groups):
        ids = []
        for n in g["members"]:
            if step_of(n) > 0:   # torch creates the per-parameter state lazily at the first step
                state[idx] = {"step": torch.tensor(float(step_of(n))), "exp_avg": _view(exp_avg, student, n).detach().clone(),
                              "exp_avg_sq": _view(exp_avg_sq, student, n).detach().clone()}
            ids.append(idx)
            idx += 1
        grp = dict(hyper)
        grp.update(name=g["name"], lr=g["lr_now"], initial_lr=g["lr"], weight_decay=g["wd_now"], params=ids)
        groups.append(grp)
    return {"state": state, "param_groups": groups}


def load_optimizer_state_dict(osd: Mapping[str, Any], student: FlatParams, exp_avg: Tensor, exp_avg_sq: Tensor,
                              param_groups: List[Dict[str, Any]], prefix_steps: Optional[Dict[str, int]] = None) -> int:
    """Inverse of `optimizer_state_dict`: fills the flat moment buffers and returns the optimizer step count.  The group
    structure of the checkpoint must be the fused structure of this model (same sizes, same order).
    `prefix_steps` (in/out): flat-name prefixes whose parameters may carry a step count of their own (or no state at all); on return it
    holds the count found for each prefix (0 when the checkpoint has no state for them).  All other parameters must agree."""
    own: Dict[str, set] = {pre: set() for pre in (prefix_steps or {})}
    fg = fused_groups(param_groups)
    if len(osd["param_groups"]) != len(fg):
        raise ValueError(f"optimizer state has {len(osd['param_groups'])} parameter groups, this model has {len(fg)}")
    steps = set()
    plan: List[Tuple[Tensor, Tensor]] = []
    for g, og in zip(fg, osd["param_groups"]):
        if len(og["params"]) != len(g["members"]):
            raise ValueError(f"parameter group {og.get('name', '?')}: {len(og['params'])} parameters in the checkpoint vs {len(g['members'])}")
        for n, idx in zip(g["members"], og["params"]):
            st = osd["state"].get(idx)
            if st is None:
                continue
            for key, buf in (("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
                v = _view(buf, student, n)
                if tuple(st[key].shape) != tuple(v.shape):
                    raise ValueError(f"optimizer state of {n}: shape {tuple(st[key].shape)} vs {tuple(v.shape)}")
                plan.append((v, st[key]))
            pre = next((q for q in own if n.startswith(q)), None)
            (steps if pre is None else own[pre]).add(int(float(st["step"])))
    for pre, found in own.items():
        if len(found) > 1:
            raise ValueError(f"parameters under {pre!r} disagree on their optimizer step count: {sorted(found)}")
        prefix_steps[pre] = found.pop() if found else 0     # type: ignore[index]
    if len(steps) > 1:
        raise ValueError(f"parameters disagree on the optimizer step count: {sorted(steps)}")
    # validated: now replace the state.  Parameters without an entry (a step-0 or partial checkpoint) get zero moments, as
    # torch's load_state_dict drops their state -- not whatever this object had accumulated before.
    exp_avg.zero_()
    exp_avg_sq.zero_()
    for v, src in plan:
        v.copy_(src.to(v.device, torch.float32))
    return steps.pop() if steps else 0


# ---- the distillation methods (Distillation, DistillationV2, DistillationV3): model state in the reference's key names, optimizer state
# in this package's flat layout (their reference optimizer is LARS / AdamW over two unnamed groups: there is no per-parameter key to keep)
def distill_load_state_dict(m: Any, sd: Mapping[str, Tensor], head_map: Mapping[str, str], strict: bool = True) -> None:
    """Inverse of the methods' `state_dict()`: `student_embedding_model.wrapped_model._model.*` (ViT / DINOv3 students; chunked
    `blocks.<chunk>.<i>.` names accepted) or `._features.*` (torchvision ResNet students), the projection heads (`head_map`: reference
    prefix -> flat prefix) and `teacher_queue`; keys of the frozen teacher, which `on_save_checkpoint` drops anyway, are ignored.
    Keys and shapes are validated before the first copy; `strict` applies to backbone, head and BatchNorm-buffer keys alike."""
    fp: FlatParams = m.student
    seen = set()
    net = getattr(m, "s_net", None) or getattr(getattr(m, "s", None), "net", None)
    conv = hasattr(net, "w_stem")
    pre_vit, pre_conv = "student_embedding_model.wrapped_model._model.", "student_embedding_model.wrapped_model._features."
    bb = {k[len(pre_conv if conv else pre_vit):]: v for k, v in sd.items() if k.startswith(pre_conv if conv else pre_vit)}
    plan: List[Tuple[str, Tensor]] = []

    def want(n: str, key: str, v: Tensor) -> None:
        if n not in fp.p:
            if strict:
                raise KeyError(f"unexpected key in state_dict: {key}")
            return
        if tuple(v.shape) != tuple(fp.shapes[n]):
            raise ValueError(f"size mismatch for {key}: checkpoint {tuple(v.shape)} vs model {tuple(fp.shapes[n])}")
        plan.append((n, v)); seen.add(n)

    if conv:
        from .resnet import resnet_param_shapes, to_flat_layout

        params = {n for n, _ in resnet_param_shapes(net.cfg)}
        for k in net.buffers:                      # BatchNorm running estimates: required when strict
            if strict and k not in bb:
                raise KeyError(f"missing BatchNorm buffer in state_dict: {pre_conv}{k}")
        for k, v in bb.items():
            if k in params:
                want("backbone." + k, pre_conv + k, to_flat_layout(k, v.float()))
            elif k not in net.buffers and not k.startswith("fc.") and strict:
                raise KeyError(f"unexpected key in state_dict: {pre_conv}{k}")
    else:
        scfg = getattr(m, "scfg")
        bb = {vit_key_to_flat(k): v for k, v in bb.items()}
        if getattr(scfg, "rope_base", None) is not None:
            from .dinov3 import convert_dinov3_state
            bb = convert_dinov3_state(bb, scfg)
        for k, v in bb.items():
            want("backbone." + k, pre_vit + k, v)
    for rp, fpre in head_map.items():
        for k, v in sd.items():
            if k.startswith(rp):
                want(fpre + k[len(rp):], k, v)
    if strict:
        missing = [n for n in fp.names if n not in seen]
        if missing:
            raise KeyError(f"missing keys for the student: {missing[:5]}{' ...' if len(missing) > 5 else ''}")
    if "teacher_queue" in sd and tuple(sd["teacher_queue"].shape) != tuple(m.teacher_queue.shape):
        raise ValueError(f"size mismatch for teacher_queue: checkpoint {tuple(sd['teacher_queue'].shape)} vs model {tuple(m.teacher_queue.shape)}")
    for n, v in plan:
        fp.p[n].copy_(v.to(fp.device, torch.float32))
    if conv:
        net.load_buffers(bb)
    if "teacher_queue" in sd:
        m.teacher_queue.copy_(sd["teacher_queue"].to(m.teacher_queue.device, torch.float32))
    fp.bf16.copy_(fp.data)
    for e in (getattr(m, "s_vit", None), net):
        if e is not None:
            e.refresh_padded_weights()


def distill_optimizer_state(m: Any) -> Dict[str, Any]:
    out: Dict[str, Any] = dict(opt_step=int(m.opt_step), global_step=int(m.trainer.global_step))
    if m.lars is not None:
        out["lars"] = m.lars.state()
    else:
        out["exp_avg"], out["exp_avg_sq"] = m.exp_avg.detach().clone(), m.exp_avg_sq.detach().clone()
    return out


def distill_load_optimizer_state(m: Any, st: Mapping[str, Any]) -> None:
    m.opt_step, m.trainer.global_step = int(st["opt_step"]), int(st["global_step"])
    if m.lars is not None:
        m.lars.load_state(st["lars"])
    else:
        m.exp_avg.copy_(st["exp_avg"].to(m.exp_avg.device)); m.exp_avg_sq.copy_(st["exp_avg_sq"].to(m.exp_avg_sq.device))
