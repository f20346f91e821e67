"""Executable drop-in at the reference's plugin point: `DINOv2AMD(Method)` -- the class INTEGRATION.md section 3 describes.

The reference has no FFI; its boundary for this path is the `Method` protocol (LT/_methods/method.py:47-155: `training_step` ->
`training_step_impl(batch, batch_idx) -> TrainingStepResult`, the optimizer / clipping / EMA hooks) and the module containers Lightning
checkpoints (`teacher_embedding_model`, `student_embedding_model`, `teacher_head`, `student_head`, `dino_loss`, `ibot_loss`:
LT/_methods/dinov2/dinov2.py:176-257).  `DINOv2AMD` keeps EXACTLY those containers -- built by the reference's own constructors, so
`state_dict()` keys, `lightly_train.export()` and the pickled `CheckpointLightlyTrainModels{model, wrapped_model, embedding_model}`
envelope (LT/_checkpoint.py:85-123, read by LT/_commands/export.py:94,165-169) are the reference's -- and runs the step on
`lightly_train_amd.dinov2.DINOv2` (flat fp32 storage + HIP kernels, explicit backward, fused AdamW / EMA):

  * `training_step_impl` hands the batch to the HIP step, which also does backward, clipping, the optimizer step and the EMA
    (`automatic_optimization = False`: Lightning neither calls backward nor steps an optimizer, and -- what makes manual optimization the
    right mode here -- its DDP strategy then leaves the wrapper's reducer switched off, so the step's own RCCL exchange is the only one);
  * gradient accumulation (`gradient_accumulation_steps`, handed to Lightning as accumulate_grad_batches by LT/_commands/train_helpers.py:224-236,
    which Lightning refuses under manual optimization): `install_as()` wraps `train_helpers.get_trainer` so that the Trainer is built with
    accumulate_grad_batches = 1 and carries the window length as `trainer.lt_amd_accumulate_grad_batches`; the binding accumulates k
    micro-batches in the flat gradient buffer (loss / k, as Lightning scales it), steps on the k-th (or the epoch's last) and runs the EMA
    after every micro-batch, as the reference's `on_train_batch_end` does; `precision` values other than "bf16-mixed" raise;
  * `on_save_checkpoint` copies the flat storage back into the reference containers (`sync_to_containers`) and puts the reference-format
    `state_dict` / `optimizer_states` / `lr_schedulers` into the checkpoint dict; the reference's ModelCheckpoint callback then pickles
    the (now current) containers into the envelope -- a file written here is read by the reference's `Checkpoint.from_dict` and exported
    by `lightly_train.export()` unchanged;
  * `on_load_checkpoint` resumes the flat storage, moments and step counters from such a checkpoint (or one the reference wrote).

`DINOAMD` (`method="dino"`, ViT and ResNet wrappers) and `DistillationV3AMD` (`method="distillation"`, BASELINE configs[3]) bind their
methods the same way (`dino_amd_method_cls()`, `distillationv3_amd_method_cls()`).

The classes are created lazily (`dinov2_amd_method_cls()` ...), because they subclass the reference's `Method`, which needs `lightly_train`
importable.  `install_as("dinov2")` maps a method name to it in `method_helpers` for `lightly_train.train(method="dinov2", ...)`;
`get_method_cls` also accepts an instance (method_helpers.py:42-44).
"""
from __future__ import annotations

from typing import Any, Dict, Mapping, Optional

import torch
from torch import Tensor

from .dinov2 import DINOv2 as _HipDINOv2
from .dinov2 import DINOv2Args as _HipArgs
from .checkpoint import vit_key_to_flat
from .vit import ViTConfig

_CLS: Optional[type] = None

SUPPORTED_PRECISIONS = ("bf16-mixed", "bf16")


def accumulate_k(self: Any) -> int:
    """Micro-batches per optimizer step: `trainer.lt_amd_accumulate_grad_batches` (set by the `get_trainer` wrapper of `install_as`) or the
    method's `gradient_accumulation_steps` attribute (a caller driving the module itself)."""
    k = getattr(self.trainer, "lt_amd_accumulate_grad_batches", None) or getattr(self, "gradient_accumulation_steps", 1)
    return max(int(k or 1), 1)


def total_optimizer_steps(self: Any) -> int:
    """Optimizer steps of the whole run, counted the way the windows really close (`begin_micro_batch`: every k-th batch AND the last batch
    of every epoch, like Lightning's own accumulation): epochs * ceil(batches_per_epoch / k), capped by `max_steps` -- which, with the
    Trainer running at accumulate_grad_batches = 1 and ticking once per optimizer step (`_tick_lightning`), already counts optimizer steps.
    `ceil(all_batches / k)` undercounts whenever an epoch is not a multiple of k (10 batches, k = 4, 100 epochs: 300 steps, not 250), and
    the schedules raise once the step passes their total.  Without a per-epoch batch count (iterable datasets, a hand-driven module) the
    Trainer's estimate is all there is: `trainer.estimated_stepping_batches` counts batches there."""
    k = accumulate_k(self)
    tr = self.trainer
    est = int(tr.estimated_stepping_batches)
    if k == 1:
        return max(est, 1)
    nb, ep, ms = getattr(tr, "num_training_batches", None), getattr(tr, "max_epochs", None), getattr(tr, "max_steps", None)
    finite = isinstance(nb, (int, float)) and nb == nb and nb not in (float("inf"),) and nb > 0
    total: Optional[int] = None
    if finite and isinstance(ep, int) and ep > 0:
        total = ep * -(-int(nb) // k)
    if isinstance(ms, int) and ms > 0:
        total = ms if total is None else min(total, ms)
    if total is None:
        total = -(-est // k)
    return max(total, 1)


def begin_micro_batch(self: Any, batch_idx: int) -> Any:
    """The engine, told where this micro-batch sits in its accumulation window; returns (engine, is_boundary).  A window ends when
    (batch_idx + 1) % k == 0 or with the epoch (`trainer.is_last_batch`), as in Lightning's own accumulation."""
    m = self.impl()
    k = accumulate_k(self)
    if int(self.trainer.global_step) > m.trainer.global_step:   # (a resumed Trainer is ahead of a freshly built step object)
        m.trainer.global_step = int(self.trainer.global_step)
    boundary = (batch_idx + 1) % k == 0 or bool(getattr(self.trainer, "is_last_batch", False))
    if getattr(m, "supports_accumulation", False):
        m.accum_first, m.grad_scale, m.accum_last = self._micro == 0, 1.0 / k, boundary
    elif k > 1:
        raise NotImplementedError(f"{type(m).__name__}: gradient accumulation is implemented for the DINOv2 step only")
    self._micro = 0 if boundary else self._micro + 1
    return m, boundary


def check_precision(self: Any) -> None:
    p = getattr(self.trainer, "precision", None)
    if isinstance(p, str) and p not in SUPPORTED_PRECISIONS:
        raise ValueError(f"precision={p!r}: the MI355X step computes with bf16 MFMA operands, fp32 accumulation and fp32 master weights, i.e. "
                         f"the reference's 'bf16-mixed'; other precisions are not implemented (supported: {SUPPORTED_PRECISIONS})")


def vit_config_from_reference(model: Any, drop_path_rate: float = 0.0) -> ViTConfig:
    """`ViTConfig` of a reference `DinoVisionTransformer` (LT/_models/dinov2_vit/dinov2_vit_src/models/vision_transformer.py:75-183),
    read off the module's attributes and parameter shapes."""
    sd = model.state_dict()
    pre = "blocks.0.0." if getattr(model, "chunked_blocks", False) else "blocks.0."
    D = int(model.embed_dim)
    swiglu = pre + "mlp.w12.weight" in sd
    if swiglu:
        hidden = sd[pre + "mlp.w3.weight"].shape[1]
        mlp_ratio, ffn = 4.0, "swiglufused"
        assert (int(D * mlp_ratio * 2 / 3) + 7) // 8 * 8 == hidden, "SwiGLU width is not the fused default of mlp_ratio 4"
    else:
        mlp_ratio, ffn = sd[pre + "mlp.fc1.weight"].shape[0] / D, "mlp"
    n_p = sd["pos_embed"].shape[1] - 1
    img = int(round(n_p ** 0.5)) * int(model.patch_size)
    chunks = len(model.blocks) if getattr(model, "chunked_blocks", False) else 0
    ls = pre + "ls1.gamma"
    return ViTConfig(embed_dim=D, depth=int(model.n_blocks), num_heads=int(model.num_heads), mlp_ratio=mlp_ratio, patch_size=int(model.patch_size),
                     img_size=img, in_chans=int(sd["patch_embed.proj.weight"].shape[1]),
                     init_values=(float(sd[ls].flatten()[0]) if ls in sd else None), interpolate_offset=float(model.interpolate_offset),
                     interpolate_antialias=bool(model.interpolate_antialias), drop_path_rate=drop_path_rate,
                     num_register_tokens=int(model.num_register_tokens), ffn_layer=ffn, block_chunks=chunks)


def hip_args_from_reference(method_args: Any, args_cls: type = _HipArgs) -> Any:
    """The reference's `DINOv2Args` (dinov2.py:70-153; "auto" values already resolved) -> the HIP step's dataclass of the same fields."""
    import dataclasses

    kw = {}
    for f in dataclasses.fields(args_cls):
        if hasattr(method_args, f.name):
            v = getattr(method_args, f.name)
            if v is not None and not (isinstance(v, str) and v == "auto"):
                kw[f.name] = tuple(v) if isinstance(v, list) else v
    return args_cls(**kw)


def _strip(sd: Mapping[str, Tensor], prefix: str, backbone: bool = False) -> Dict[str, Tensor]:
    """Sub-dict of a state_dict without its prefix; backbone keys of chunked models (`blocks.<chunk>.<i>.`) flattened to `blocks.<i>.`."""
    return {(vit_key_to_flat(k[len(prefix):]) if backbone else k[len(prefix):]): v.detach().clone() for k, v in sd.items() if k.startswith(prefix)}


class DINOv2BindingMixin:
    """Everything of `DINOv2AMD` that does not need the reference package: the HIP engine built from the containers' weights on first use,
    the training-step hook (accumulation window, optimizer step, EMA), the flat-storage -> container sync and the checkpoint hooks.  The
    host class provides the module containers under the reference's names (`teacher_embedding_model.wrapped_model._model`,
    `student_head.dino_head`, `dino_loss.center`, ...: in production the reference's own `DINOv2(Method)`), `method_args`,
    `global_batch_size`, `trainer`, and the attributes `__init__` of `DINOv2AMD` sets.  tests/test_gpu_step.py runs this mixin over the
    real kernels on a container tree built from a reference-written fixture (the GPU box has no reference package)."""

    _result_cls: Any = None     # the TrainingStepResult class the host's `training_step` expects (the reference's in production)

    def _init_binding(self, device: Optional[torch.device], gradient_accumulation_steps: int) -> None:
        self.automatic_optimization = False     # the HIP step owns backward, clipping, AdamW and the EMA
        self.gradient_accumulation_steps = int(gradient_accumulation_steps)
        self._micro = 0                         # position inside the accumulation window
        self._impl: Optional[_HipDINOv2] = None
        self._impl_device = device
        self._pending_resume: Optional[Dict[str, Any]] = None

    # ---- the HIP step, built on first use (Lightning moves the module to its device only after __init__)
    def impl(self) -> _HipDINOv2:
        if self._impl is None:
            check_precision(self)
            dev = self._impl_device or next(self.parameters()).device
            sd = torch.nn.Module.state_dict(self)
            t_model = self.teacher_embedding_model.wrapped_model.get_model()
            a = self.method_args
            cfg = vit_config_from_reference(t_model)
            bb = "embedding_model.wrapped_model._model."
            engine_cls, args_cls, extra = self._engine(sd)
            self._impl = engine_cls(
                cfg, hip_args_from_reference(a, args_cls), global_batch_size=self.global_batch_size,
                total_steps=total_optimizer_steps(self), device=dev,
                backbone_state=_strip(sd, "student_" + bb, True), teacher_backbone_state=_strip(sd, "teacher_" + bb, True),
                student_head_state=_strip(sd, "student_head.dino_head."), teacher_head_state=_strip(sd, "teacher_head.dino_head."),
                student_ibot_head_state=_strip(sd, "student_head.ibot_head.") if a.ibot_separate_head else None,
                teacher_ibot_head_state=_strip(sd, "teacher_head.ibot_head.") if a.ibot_separate_head else None, **extra)
            self._impl.load_state_dict(sd)   # centers, BatchNorm buffers, chunked-block key names
            if self._pending_resume is not None:
                self._impl.load_checkpoint_dict(self._pending_resume)
                self._pending_resume = None
        return self._impl

    def _engine(self, sd: Mapping[str, Tensor]) -> Any:
        """(engine class, its argument dataclass, extra constructor keywords) -- DINOv2 here; the DINOv31 binding overrides it."""
        return _HipDINOv2, _HipArgs, {}

    def training_step_impl(self, batch: Mapping[str, Any], batch_idx: int) -> Any:
        m, boundary = begin_micro_batch(self, batch_idx)
        res = m.training_step_impl(batch, batch_idx)     # forward + explicit backward in HIP: gradients accumulate in m.student.grad
        if boundary:
            m.optimizer_step()                           # WD / lr schedules + freezes, clip 3.0, AdamW; global_step += 1 (dinov2.py:550-639)
            self._tick_lightning()
        m.on_train_batch_end()                           # EMA teacher at the engine's step, every micro-batch (dinov2.py:641-660)
        return self._result_cls(loss=res.loss, log_dict=res.log_dict)

    def _tick_lightning(self) -> None:
        """Manual optimization: Lightning advances `trainer.global_step` when a LightningOptimizer steps.  `configure_optimizers`
        returns a no-op SGD over a tensor that is not part of the module, stepped here once per optimizer step, so that max_steps, the
        checkpoint callback's step counter and `checkpoint["global_step"]` keep their meaning.  Only a module that is not attached to
        a Lightning Trainer (this repo's CPU tests drive the hooks by hand) has no optimizers to step; any other failure propagates."""
        strategy = getattr(self.trainer, "strategy", None)
        if strategy is None:      # no Lightning Trainer behind `self.trainer`: the caller advances its own counter
            return
        o = self.optimizers()
        for one in (o if isinstance(o, (list, tuple)) else [o]):
            one.step()

    def configure_optimizers(self) -> Any:   # manual optimization: a counter for Lightning's progress tracking only (_tick_lightning)
        return torch.optim.SGD([torch.zeros((), requires_grad=True)], lr=0.0)

    def configure_gradient_clipping(self, *a: Any, **k: Any) -> None:
        return None

    def on_before_optimizer_step(self, *a: Any, **k: Any) -> None:
        return None

    def on_train_batch_end(self, outputs: Any, batch: Any, batch_idx: int) -> None:
        parent = getattr(super(), "on_train_batch_end", None)     # `Method.on_train_batch_end`: batch timing only -- the EMA has run
        if parent is not None:
            parent(outputs=outputs, batch=batch, batch_idx=batch_idx)

    # ---- checkpoints
    def sync_to_containers(self) -> None:
        """Flat HIP storage -> the reference module containers (what `state_dict()`, the export commands and the pickled envelope read)."""
        if self._impl is not None:
            sd = {k: v.detach().to("cpu") for k, v in self._impl.state_dict().items()}
            torch.nn.Module.load_state_dict(self, sd, strict=True)

    def state_dict(self, *a: Any, **k: Any) -> Any:
        self.sync_to_containers()
        return torch.nn.Module.state_dict(self, *a, **k)

    def on_save_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        self.sync_to_containers()
        checkpoint["state_dict"] = torch.nn.Module.state_dict(self)
        if self._impl is not None:
            ck = self._impl.checkpoint_dict()
            checkpoint["optimizer_states"] = ck["optimizer_states"]
            checkpoint["lr_schedulers"] = ck["lr_schedulers"]

    def on_load_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        ck = {k: checkpoint[k] for k in ("state_dict", "optimizer_states", "global_step") if k in checkpoint}
        if self._impl is not None:
            self._impl.load_checkpoint_dict(ck)
        else:
            self._pending_resume = ck
        # Lightning restores optimizers AFTER this hook from the same dict (`restore_optimizers`: optimizer.load_state_dict(
        # checkpoint["optimizer_states"][i])): what it finds must fit the progress-counter SGD of `configure_optimizers`, not the
        # reference-format AdamW / SGD state (many parameter groups) the engine has just taken -- for checkpoints written here and by
        # the reference alike.  No scheduler is registered with Lightning (the engine owns the schedule): nothing to restore there.
        if "optimizer_states" in checkpoint:
            checkpoint["optimizer_states"] = [self.configure_optimizers().state_dict()]
        if "lr_schedulers" in checkpoint:
            checkpoint["lr_schedulers"] = []


def dinov2_amd_method_cls() -> type:
    """The `Method` subclass (needs the reference package importable)."""
    global _CLS
    if _CLS is not None:
        return _CLS
    from lightly_train._methods.dinov2.dinov2 import DINOv2 as RefDINOv2
    from lightly_train._methods.method import Method, TrainingStepResult

    class DINOv2AMD(DINOv2BindingMixin, RefDINOv2):   # type: ignore[misc, valid-type]
        """`method="dinov2"` on MI355X.  Subclasses the reference method for its constructor (the module containers, built by the
        reference's own code), its static class hooks (`method_args_cls`, `optimizer_args_cls`, `transform_cls`) and `Method.training_step`
        (logging with sync_dist); everything that computes is replaced."""

        def __init__(self, method_args: Any, optimizer_args: Any, embedding_model: Any, global_batch_size: int, num_input_channels: int,
                     device: Optional[torch.device] = None, gradient_accumulation_steps: int = 1) -> None:
            super().__init__(method_args=method_args, optimizer_args=optimizer_args, embedding_model=embedding_model,
                             global_batch_size=global_batch_size, num_input_channels=num_input_channels)
            self._init_binding(device, gradient_accumulation_steps)

    DINOv2AMD._result_cls = TrainingStepResult
    DINOv2AMD.__qualname__ = "DINOv2AMD"
    _CLS = DINOv2AMD
    return _CLS


_V31_CLS: Optional[type] = None


def dinov31_amd_method_cls() -> type:
    """`DINOv31AMD(Method)`: `method="dinov31"` (LT/_methods/dinov31/dinov31.py: DINOv2 + the PaKA term) on `lightly_train_amd.dinov31.DINOv31`
    -- the DINOv2 binding with another engine: the reference's constructor also builds `student_paka_head` / `teacher_paka_head`, whose weights
    start the engine's PaKA heads; the hooks (accumulation is refused by this engine, resume, checkpoints) are the mixin's."""
    global _V31_CLS
    if _V31_CLS is not None:
        return _V31_CLS
    from lightly_train._methods.dinov31.dinov31 import DINOv31 as RefDINOv31
    from lightly_train._methods.method import TrainingStepResult

    from .dinov31 import DINOv31 as HipDINOv31
    from .dinov31 import DINOv31Args as HipDINOv31Args

    class DINOv31AMD(DINOv2BindingMixin, RefDINOv31):   # type: ignore[misc, valid-type]
        def __init__(self, method_args: Any, optimizer_args: Any, embedding_model: Any, global_batch_size: int, num_input_channels: int,
                     device: Optional[torch.device] = None, gradient_accumulation_steps: int = 1) -> None:
            super().__init__(method_args=method_args, optimizer_args=optimizer_args, embedding_model=embedding_model,
                             global_batch_size=global_batch_size, num_input_channels=num_input_channels)
            self._init_binding(device, gradient_accumulation_steps)

        def _engine(self, sd: Mapping[str, Tensor]) -> Any:
            return HipDINOv31, HipDINOv31Args, dict(paka_head_state=_strip(sd, "student_paka_head."), teacher_paka_head_state=_strip(sd, "teacher_paka_head."))

        def on_train_batch_end(self, outputs: Any, batch: Any, batch_idx: int) -> None:
            from lightly_train._methods.method import Method
            Method.on_train_batch_end(self, outputs=outputs, batch=batch, batch_idx=batch_idx)    # batch timing only: both EMAs ran in the flat update

    DINOv31AMD._result_cls = TrainingStepResult
    DINOv31AMD.__qualname__ = "DINOv31AMD"
    _V31_CLS = DINOv31AMD
    return _V31_CLS


_DINO_CLS: Optional[type] = None


def dino_amd_method_cls() -> type:
    """`DINOAMD(Method)`: the same binding for `method="dino"` (LT/_methods/dino/dino.py:221-480) on `lightly_train_amd.dino.DINO`.
    ViT backbones (the DINOv2 ViT wrapper); the reference's containers are `teacher_embedding_model`, `teacher_projection_head`,
    `student_embedding_model`, `student_projection_head`, `criterion.center`."""
    global _DINO_CLS
    if _DINO_CLS is not None:
        return _DINO_CLS
    from lightly_train._methods.dino.dino import DINO as RefDINO
    from lightly_train._methods.method import Method, TrainingStepResult
    from lightly_train._optim.optimizer_type import OptimizerType

    from .dino import DINO as HipDINO
    from .dino import DINOArgs as HipDINOArgs

    base = dinov2_amd_method_cls()

    class DINOAMD(RefDINO):   # type: ignore[misc, valid-type]
        def __init__(self, method_args: Any, optimizer_args: Any, embedding_model: Any, global_batch_size: int, num_input_channels: int,
                     device: Optional[torch.device] = None, gradient_accumulation_steps: int = 1) -> None:
            super().__init__(method_args=method_args, optimizer_args=optimizer_args, embedding_model=embedding_model,
                             global_batch_size=global_batch_size, num_input_channels=num_input_channels)
            self.automatic_optimization = False
            self.gradient_accumulation_steps = int(gradient_accumulation_steps)
            self._micro = 0
            self._impl: Optional[HipDINO] = None
            self._impl_device = device
            self._pending_resume: Optional[Dict[str, Any]] = None

        def impl(self) -> HipDINO:
            if self._impl is None:
                import dataclasses

                check_precision(self)
                dev = self._impl_device or next(self.parameters()).device
                sd = Method.state_dict(self)
                a, oa = self.method_args, self.optimizer_args
                kw = {f.name: getattr(a, f.name) for f in dataclasses.fields(HipDINOArgs)
                      if hasattr(a, f.name) and getattr(a, f.name) is not None and getattr(a, f.name) != "auto"}
                # the deprecated epoch spellings (resolve_auto leaves *_steps at None when they are set): converted as dino.py:301-312 / :450-468
                total = int(self.trainer.estimated_stepping_batches)
                spe = -(-total // int(self.trainer.max_epochs or 1))
                if getattr(a, "warmup_teacher_temp_steps", None) is None:
                    if getattr(a, "warmup_teacher_temp_epochs", None) is None:
                        raise ValueError("Either warmup_teacher_temp_epochs or warmup_teacher_temp_steps must be set.")
                    kw["warmup_teacher_temp_steps"] = min(int(a.warmup_teacher_temp_epochs * spe), int(total * a.warmup_teacher_temp_max_steps_fraction))
                if getattr(a, "student_freeze_last_layer_steps", None) is None:
                    if getattr(a, "student_freeze_last_layer_epochs", None) is None:
                        raise ValueError("Either student_freeze_last_layer_epochs or student_freeze_last_layer_steps must be set.")
                    kw["student_freeze_last_layer_steps"] = int(a.student_freeze_last_layer_epochs * spe)
                is_sgd = oa.type() == OptimizerType.SGD
                kw.update(optimizer="sgd" if is_sgd else "adamw", lr=float(oa.lr), weight_decay=float(oa.weight_decay))
                if is_sgd:
                    kw["momentum"] = float(oa.momentum)
                else:
                    kw.update(betas=tuple(oa.betas), eps=float(oa.eps))
                wrapped = self.teacher_embedding_model.wrapped_model
                common = dict(global_batch_size=self.global_batch_size, total_steps=int(self.trainer.estimated_stepping_batches), device=dev,
                              student_head_state=_strip(sd, "student_projection_head."), teacher_head_state=_strip(sd, "teacher_projection_head."))
                if hasattr(wrapped, "_features"):       # ResNetModelWrapper: convolutional backbone
                    from .dino import DINOResNet

                    bb = "embedding_model.wrapped_model._features."
                    self._impl = DINOResNet(resnet_config_from_reference(wrapped._features), HipDINOArgs(**kw),
                                            backbone_state=_strip(sd, "student_" + bb), teacher_backbone_state=_strip(sd, "teacher_" + bb), **common)
                else:
                    bb = "embedding_model.wrapped_model._model."
                    self._impl = HipDINO(vit_config_from_reference(wrapped.get_model()), HipDINOArgs(**kw),
                                         backbone_state=_strip(sd, "student_" + bb, True), teacher_backbone_state=_strip(sd, "teacher_" + bb, True), **common)
                self._impl.load_state_dict(sd)
                if self._pending_resume is not None:
                    self._impl.load_checkpoint_dict(self._pending_resume)
                    self._pending_resume = None
            return self._impl

        def training_step_impl(self, batch: Mapping[str, Any], batch_idx: int) -> Any:
            m, _ = begin_micro_batch(self, batch_idx)         # (k > 1 raises: accumulation is bound for the DINOv2 step)
            res = m.training_step_impl(batch, batch_idx)     # EMA first, then forward + explicit backward (dino.py:273-316)
            m.optimizer_step()                               # WD schedule, last-layer freeze, clip 3.0, SGD / AdamW (dino.py:330-477)
            self._tick_lightning()
            return TrainingStepResult(loss=res.loss, log_dict=res.log_dict)

    for name in ("_tick_lightning", "configure_optimizers", "configure_gradient_clipping", "on_before_optimizer_step", "on_train_batch_end",
                 "sync_to_containers", "state_dict", "on_save_checkpoint", "on_load_checkpoint"):
        setattr(DINOAMD, name, getattr(base, name))
    DINOAMD.__qualname__ = "DINOAMD"
    _DINO_CLS = DINOAMD
    return _DINO_CLS


def dinov3_config_from_reference(model: Any, student: bool = False) -> ViTConfig:
    """`ViTConfig` of a reference DINOv3 `DinoVisionTransformer` (LT/_models/dinov3/dinov3_src/models/vision_transformer.py:75-230) from its
    attributes and state_dict.  `student`: keep the training-mode RoPE coordinate rescaling (a frozen teacher runs in eval())."""
    from .dinov3 import dinov3_vit_config

    sd = model.state_dict()
    D = int(model.embed_dim)
    swiglu = "blocks.0.mlp.w1.weight" in sd
    if swiglu:
        raise NotImplementedError("DINOv3 SwiGLU feed-forward blocks are not bound (the hub's ViT-S/B/L recipes use the MLP)")
    rope = model.rope_embed
    ls = "blocks.0.ls1.gamma"
    return dinov3_vit_config(D, int(model.n_blocks), int(model.num_heads), patch_size=int(model.patch_size), img_size=224,
                             ffn_ratio=sd["blocks.0.mlp.fc1.weight"].shape[0] / D, n_storage_tokens=int(model.n_storage_tokens),
                             layerscale_init=(float(sd[ls].flatten()[0]) if ls in sd else None), rope_base=float(rope.base),
                             ln_eps=float(model.norm.eps), rope_rescale=(float(rope.rescale_coords) if (student and rope.rescale_coords) else None),
                             mask_k_bias=any(k.endswith("qkv.bias_mask") for k in sd))


_DV3_CLS: Optional[type] = None


def distillationv3_amd_method_cls() -> type:
    """`DistillationV3AMD(Method)`: the binding for `method="distillation"` (LT/_methods/distillationv3/distillationv3.py:171-440, BASELINE
    configs[3]) on `lightly_train_amd.distillationv3.DistillationV3`: frozen DINOv3 / DINOv2 ViT teacher, DINOv2-ViT / DINOv3-ViT / torchvision
    ResNet student, the two Linear projection heads, the teacher queue.  Same hooks as `DINOv2AMD`; the teacher's weights stay out of
    checkpoints exactly as in the reference (`on_save_checkpoint` / `on_load_checkpoint` of the base class still run)."""
    global _DV3_CLS
    if _DV3_CLS is not None:
        return _DV3_CLS
    from lightly_train._methods.distillationv3.distillationv3 import DistillationV3 as RefDV3
    from lightly_train._methods.method import Method, TrainingStepResult
    from lightly_train._optim.optimizer_type import OptimizerType

    from .dinov3 import convert_dinov3_state
    from .distillationv3 import DistillationV3 as HipDV3
    from .distillationv3 import DistillationV3Args as HipDV3Args
    from .lars import LARSArgs

    base = dinov2_amd_method_cls()

    class DistillationV3AMD(RefDV3):   # type: ignore[misc, valid-type]
        def __init__(self, method_args: Any, optimizer_args: Any, embedding_model: Any, global_batch_size: int, num_input_channels: int,
                     device: Optional[torch.device] = None, gradient_accumulation_steps: int = 1) -> None:
            super().__init__(method_args=method_args, optimizer_args=optimizer_args, embedding_model=embedding_model,
                             global_batch_size=global_batch_size, num_input_channels=num_input_channels)
            self.automatic_optimization = False
            self.gradient_accumulation_steps = int(gradient_accumulation_steps)
            self._micro = 0
            self._impl: Optional[HipDV3] = None
            self._impl_device = device
            self._pending_resume: Optional[Dict[str, Any]] = None

        def impl(self) -> HipDV3:
            if self._impl is None:
                dev = self._impl_device or next(self.parameters()).device
                a, oa = self.method_args, self.optimizer_args
                check_precision(self)
                kw: Dict[str, Any] = dict(queue_size=int(a.queue_size), temperature_global=float(a.temperature_global), temperature_local=float(a.temperature_local),
                                          lr_scale_method=a.lr_scale_method, reference_batch_size=int(a.reference_batch_size),
                                          loss_local_weight=float(a.loss_local_weight))
                if oa.type() == OptimizerType.LARS:
                    kw.update(optimizer="lars", lars=LARSArgs(lr=float(oa.lr), momentum=float(oa.momentum), dampening=float(oa.dampening),
                                                              weight_decay=float(oa.weight_decay), nesterov=bool(oa.nesterov),
                                                              trust_coefficient=float(oa.trust_coefficient), eps=float(oa.eps)))
                else:
                    kw.update(optimizer="adamw", lr=float(oa.lr), betas=tuple(oa.betas), eps=float(oa.eps), weight_decay=float(oa.weight_decay))
                t_model = self.teacher_embedding_model.get_model()
                t_sd = {k: v.detach().clone() for k, v in t_model.state_dict().items()}
                if hasattr(t_model, "rope_embed"):
                    tcfg = dinov3_config_from_reference(t_model)
                    t_state = convert_dinov3_state(t_sd, tcfg)
                else:
                    tcfg = vit_config_from_reference(t_model)
                    t_state = {vit_key_to_flat(k): v for k, v in t_sd.items()}
                wrapped = self.student_embedding_model.wrapped_model
                if hasattr(wrapped, "_features"):
                    scfg: Any = resnet_config_from_reference(wrapped._features)
                    s_state = {k: v.detach().clone() for k, v in wrapped._features.state_dict().items()}
                else:
                    s_model = wrapped.get_model()
                    s_sd = {k: v.detach().clone() for k, v in s_model.state_dict().items()}
                    if hasattr(s_model, "rope_embed"):
                        scfg = dinov3_config_from_reference(s_model, student=True)
                        s_state = convert_dinov3_state(s_sd, scfg)
                    else:
                        scfg = vit_config_from_reference(s_model)
                        s_state = {vit_key_to_flat(k): v for k, v in s_sd.items()}
                lin = lambda m_: {k: v.detach().clone() for k, v in m_.state_dict().items()}    # noqa: E731
                self._impl = HipDV3(scfg, tcfg, HipDV3Args(**kw), global_batch_size=self.global_batch_size,
                                    total_steps=int(self.trainer.estimated_stepping_batches), max_epochs=int(self.trainer.max_epochs or 1), device=dev,
                                    student_state=s_state, teacher_state=t_state, proj_global_state=lin(self.student_projection_head_global),
                                    proj_local_state=lin(self.student_projection_head_local))
                self._impl.teacher_queue.copy_(self.teacher_queue.to(dev))
                if self._pending_resume is not None:
                    self._impl.load_state_dict(self._pending_resume["state_dict"], strict=False)
                    if self._pending_resume.get("amd_optimizer_state") is not None:
                        self._impl.load_optimizer_state(self._pending_resume["amd_optimizer_state"])
                    self._pending_resume = None
            return self._impl

        def training_step_impl(self, batch: Mapping[str, Any], batch_idx: int) -> Any:
            m, _ = begin_micro_batch(self, batch_idx)
            res = m.training_step_impl(batch, batch_idx)     # frozen teacher, mixup, student forward + explicit backward (:245-374)
            m.optimizer_step()                               # clip 1.0, AdamW / LARS, generic warmup-cosine schedule
            self._tick_lightning()
            return TrainingStepResult(loss=res.loss, log_dict=dict(res.log_dict))

        def sync_to_containers(self) -> None:
            if self._impl is not None:
                sd = {k: v.detach().to("cpu") for k, v in self._impl.state_dict().items()}
                Method.load_state_dict(self, sd, strict=False)      # (the frozen teacher is not part of the step's state_dict)

        def state_dict(self, *a: Any, **k: Any) -> Any:
            self.sync_to_containers()
            return Method.state_dict(self, *a, **k)

        def on_save_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
            self.sync_to_containers()
            checkpoint["state_dict"] = Method.state_dict(self)
            if self._impl is not None:
                checkpoint["amd_optimizer_state"] = self._impl.optimizer_state()   # flat moments / LARS buffer + counters (package layout)
            RefDV3.on_save_checkpoint(self, checkpoint)                            # drops the teacher's keys, as the reference does

        def on_load_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
            ck = {"state_dict": dict(checkpoint["state_dict"]), "amd_optimizer_state": checkpoint.get("amd_optimizer_state")}
            RefDV3.on_load_checkpoint(self, checkpoint)                            # re-adds the teacher's keys for Lightning's strict load
            if "optimizer_states" in checkpoint:                                   # see DINOv2AMD.on_load_checkpoint
                checkpoint["optimizer_states"] = [self.configure_optimizers().state_dict()]
            if "lr_schedulers" in checkpoint:
                checkpoint["lr_schedulers"] = []
            if self._impl is not None:
                self._impl.load_state_dict(ck["state_dict"], strict=False)
                if ck["amd_optimizer_state"] is not None:
                    self._impl.load_optimizer_state(ck["amd_optimizer_state"])
            else:
                self._pending_resume = ck

    for name in ("_tick_lightning", "configure_optimizers", "configure_gradient_clipping", "on_before_optimizer_step", "on_train_batch_end"):
        setattr(DistillationV3AMD, name, getattr(base, name))
    DistillationV3AMD.__qualname__ = "DistillationV3AMD"
    _DV3_CLS = DistillationV3AMD
    return _DV3_CLS


_D12_CLS: Dict[str, type] = {}


def distillation12_amd_method_cls(kind: str) -> type:
    """`DistillationAMD` (kind "v1": LT/_methods/distillation/distillation.py:155-360, pooled feature against a queue, KL) and
    `DistillationV2AMD` (kind "v2": LT/_methods/distillationv2/distillationv2.py:155-377, patch features of the last teacher blocks, MSE)
    on `lightly_train_amd.distillation`.  In both the reference keeps the teacher as the bare ViT (`teacher_embedding_model`), the
    student as the `EmbeddingModel`, one `student_projection_head`; "auto" optimizer = LARS."""
    if kind in _D12_CLS:
        return _D12_CLS[kind]
    import importlib

    from lightly_train._methods.method import Method, TrainingStepResult
    from lightly_train._optim.optimizer_type import OptimizerType

    from . import distillation as HD
    from .dinov3 import convert_dinov3_state
    from .lars import LARSArgs

    if kind == "v1":
        Ref = importlib.import_module("lightly_train._methods.distillation.distillation").Distillation
        Hip, HipArgs = HD.Distillation, HD.DistillationArgs
    else:
        Ref = importlib.import_module("lightly_train._methods.distillationv2.distillationv2").DistillationV2
        Hip, HipArgs = HD.DistillationV2, HD.DistillationV2Args
    base = dinov2_amd_method_cls()

    class DistillationAMD(Ref):   # type: ignore[misc, valid-type]
        def __init__(self, method_args: Any, optimizer_args: Any, embedding_model: Any, global_batch_size: int, num_input_channels: int,
                     device: Optional[torch.device] = None, gradient_accumulation_steps: int = 1) -> None:
            super().__init__(method_args=method_args, optimizer_args=optimizer_args, embedding_model=embedding_model,
                             global_batch_size=global_batch_size, num_input_channels=num_input_channels)
            self.automatic_optimization = False
            self.gradient_accumulation_steps = int(gradient_accumulation_steps)
            self._micro = 0
            self._impl: Optional[Any] = None
            self._impl_device = device
            self._pending_resume: Optional[Dict[str, Any]] = None

        def impl(self) -> Any:
            if self._impl is None:
                dev = self._impl_device or next(self.parameters()).device
                a, oa = self.method_args, self.optimizer_args
                check_precision(self)
                kw: Dict[str, Any] = dict(lr_scale_method=a.lr_scale_method, reference_batch_size=int(a.reference_batch_size))
                if kind == "v1":
                    kw.update(queue_size=int(a.queue_size), temperature=float(a.temperature))
                else:
                    kw.update(n_teacher_blocks=int(a.n_teacher_blocks), n_projection_layers=int(a.n_projection_layers),
                              projection_hidden_dim=int(a.projection_hidden_dim))
                if oa.type() == OptimizerType.LARS:
                    kw.update(optimizer="lars", lars=LARSArgs(lr=float(oa.lr), momentum=float(oa.momentum), dampening=float(oa.dampening),
                                                              weight_decay=float(oa.weight_decay), nesterov=bool(oa.nesterov),
                                                              trust_coefficient=float(oa.trust_coefficient), eps=float(oa.eps)))
                else:
                    kw.update(optimizer="adamw", lr=float(oa.lr), betas=tuple(oa.betas), eps=float(oa.eps), weight_decay=float(oa.weight_decay))
                t_model = self.teacher_embedding_model
                t_sd = {k: v.detach().clone() for k, v in t_model.state_dict().items()}
                if hasattr(t_model, "rope_embed"):
                    tcfg = dinov3_config_from_reference(t_model)
                    t_state = convert_dinov3_state(t_sd, tcfg)
                else:
                    tcfg = vit_config_from_reference(t_model)
                    t_state = {vit_key_to_flat(k): v for k, v in t_sd.items()}
                wrapped = self.student_embedding_model.wrapped_model
                if hasattr(wrapped, "_features"):
                    scfg: Any = resnet_config_from_reference(wrapped._features)
                    s_state = {k: v.detach().clone() for k, v in wrapped._features.state_dict().items()}
                else:
                    s_model = wrapped.get_model()
                    if hasattr(s_model, "rope_embed"):
                        raise NotImplementedError("DINOv3 students are bound for DistillationV3 only")
                    scfg = vit_config_from_reference(s_model)
                    s_state = {vit_key_to_flat(k): v.detach().clone() for k, v in s_model.state_dict().items()}
                head = {k: v.detach().clone() for k, v in self.student_projection_head.state_dict().items()}
                self._impl = Hip(scfg, tcfg, HipArgs(**kw), global_batch_size=self.global_batch_size,
                                 total_steps=int(self.trainer.estimated_stepping_batches), max_epochs=int(self.trainer.max_epochs or 1), device=dev,
                                 student_state=s_state, teacher_state=t_state, head_state=head)
                if kind == "v1":
                    self._impl.teacher_queue.copy_(self.teacher_queue.to(dev))
                if self._pending_resume is not None:
                    self._impl.load_state_dict(self._pending_resume["state_dict"], strict=False)
                    if self._pending_resume.get("amd_optimizer_state") is not None:
                        self._impl.load_optimizer_state(self._pending_resume["amd_optimizer_state"])
                    self._pending_resume = None
            return self._impl

        def training_step_impl(self, batch: Mapping[str, Any], batch_idx: int) -> Any:
            m, _ = begin_micro_batch(self, batch_idx)
            res = m.training_step_impl(batch, batch_idx)
            m.optimizer_step()
            self._tick_lightning()
            return TrainingStepResult(loss=res.loss, log_dict=dict(res.log_dict))

        def sync_to_containers(self) -> None:
            if self._impl is not None:
                sd = {k: v.detach().to("cpu") for k, v in self._impl.state_dict().items()}
                Method.load_state_dict(self, sd, strict=False)

        def state_dict(self, *a: Any, **k: Any) -> Any:
            self.sync_to_containers()
            return Method.state_dict(self, *a, **k)

        def on_save_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
            self.sync_to_containers()
            checkpoint["state_dict"] = Method.state_dict(self)
            if self._impl is not None:
                checkpoint["amd_optimizer_state"] = self._impl.optimizer_state()
            Ref.on_save_checkpoint(self, checkpoint)

        def on_load_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
            ck = {"state_dict": dict(checkpoint["state_dict"]), "amd_optimizer_state": checkpoint.get("amd_optimizer_state")}
            Ref.on_load_checkpoint(self, checkpoint)
            if "optimizer_states" in checkpoint:
                checkpoint["optimizer_states"] = [self.configure_optimizers().state_dict()]
            if "lr_schedulers" in checkpoint:
                checkpoint["lr_schedulers"] = []
            if self._impl is not None:
                self._impl.load_state_dict(ck["state_dict"], strict=False)
                if ck["amd_optimizer_state"] is not None:
                    self._impl.load_optimizer_state(ck["amd_optimizer_state"])
            else:
                self._pending_resume = ck

    for name in ("_tick_lightning", "configure_optimizers", "configure_gradient_clipping", "on_before_optimizer_step", "on_train_batch_end"):
        setattr(DistillationAMD, name, getattr(base, name))
    DistillationAMD.__qualname__ = DistillationAMD.__name__ = "DistillationAMD" if kind == "v1" else "DistillationV2AMD"
    _D12_CLS[kind] = DistillationAMD
    return DistillationAMD


def resnet_config_from_reference(features: Any) -> Any:
    """`ResNetConfig` of the `_features` container of a reference `ResNetModelWrapper` (LT/_models/torchvision/resnet.py): stage depths and
    the stem width read off the state_dict."""
    from .resnet import ResNetConfig

    sd = features.state_dict()
    layers = []
    for li in range(1, 5):
        n = 0
        while f"layer{li}.{n}.conv1.weight" in sd:
            n += 1
        layers.append(n)
    return ResNetConfig(layers=tuple(layers), width=int(sd["conv1.weight"].shape[0]), in_chans=int(sd["conv1.weight"].shape[1]))


def install_as(name: str = "dinov2") -> type:
    """Map a method name of `lightly_train.train(method=...)` to the MI355X class (method_helpers.py:54-69 builds its table per call)."""
    from lightly_train._methods import method_helpers

    cls = {"dino": dino_amd_method_cls, "dinov31": dinov31_amd_method_cls, "distillation": distillationv3_amd_method_cls, "distillationv3": distillationv3_amd_method_cls,
           "distillationv1": lambda: distillation12_amd_method_cls("v1"), "distillationv2": lambda: distillation12_amd_method_cls("v2")}.get(
        name, dinov2_amd_method_cls)()
    orig = method_helpers._method_name_to_cls

    def patched() -> Dict[str, type]:
        m = orig()
        m[name] = cls
        return m

    method_helpers._method_name_to_cls = patched
    _AMD_CLASSES.add(cls)
    _wrap_get_trainer()
    return cls


_AMD_CLASSES: set = set()             # the classes `install_as` has put into the method table of this process
_LAST_TRAINER: Dict[str, Any] = {}    # the Trainer the wrapped `get_trainer` built last, and the k it was asked for


def _wrap_get_trainer() -> None:
    """`lightly_train.train(gradient_accumulation_steps=k)` reaches Lightning as Trainer(accumulate_grad_batches=k)
    (LT/_commands/train_helpers.py:208-247), which Lightning rejects for a manual-optimization module.  Build the Trainer with 1 and let it
    carry k as `lt_amd_accumulate_grad_batches`: the binding accumulates in its flat gradient buffer (`begin_micro_batch`).
    `train()` builds the Trainer BEFORE it resolves the method class (LT/_commands/train.py:433 / :476), so the rewrite cannot look at the
    method; it is undone in the wrapped `get_method_cls` when the class that comes out is not one of the installed MI355X classes -- a
    `simclr` run in the same process keeps Lightning's own accumulation."""
    from lightly_train._commands import train_helpers
    from lightly_train._methods import method_helpers

    if getattr(train_helpers.get_trainer, "_lt_amd_wrapped", False):
        return
    orig = train_helpers.get_trainer

    def get_trainer(*args: Any, **kwargs: Any) -> Any:
        import inspect

        _LAST_TRAINER.clear()    # a train() that aborted between get_trainer and get_method_cls must not leave its Trainer to the next run
        bound = inspect.signature(orig).bind(*args, **kwargs)
        k = int(bound.arguments.get("gradient_accumulation_steps", 1) or 1)
        bound.arguments["gradient_accumulation_steps"] = 1
        trainer = orig(*bound.args, **bound.kwargs)
        trainer.lt_amd_accumulate_grad_batches = k
        _LAST_TRAINER.update(trainer=trainer, k=k)
        return trainer

    orig_get_cls = method_helpers.get_method_cls

    def get_method_cls(method: Any) -> Any:
        cls = orig_get_cls(method)
        tr, k = _LAST_TRAINER.get("trainer"), _LAST_TRAINER.get("k", 1)
        if tr is not None and cls not in _AMD_CLASSES:
            # not ours: hand the window back to Lightning (Trainer.accumulate_grad_batches is a plain attribute the epoch loop reads)
            tr.accumulate_grad_batches = k
            if hasattr(tr, "lt_amd_accumulate_grad_batches"):
                del tr.lt_amd_accumulate_grad_batches
        _LAST_TRAINER.clear()
        return cls

    get_trainer._lt_amd_wrapped = True   # type: ignore[attr-defined]
    train_helpers.get_trainer = get_trainer
    method_helpers.get_method_cls = get_method_cls
