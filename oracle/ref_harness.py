"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Loads the *reference's own* DINOv2 hot-path modules from /root/reference on CPU
so the restatement in oracle/dinov2_oracle.py can be pinned against them and so
that golden fixtures (tests/golden/) can be generated (oracle/make_golden.py).

`import lightly_train` fails in this image (cv2 / torchvision / pytorch_lightning
/ lightly / omegaconf / albumentations are absent, no network), see SURVEY.md
section 8(c).  We therefore
  * pre-register `lightly_train` as a namespace-like module pointing at the
    reference source dir (skips lightly_train/__init__.py and its `import cv2`),
  * install a MetaPathFinder that fabricates permissive stub modules for the
    absent third-party packages,
  * provide a minimal `LightningModule(nn.Module)` and restate the three
    un-vendored LightlySSL helpers the method calls (KoLeoLoss, cosine_schedule,
    CosineWarmupScheduler, update_param_groups) -- the reference does not vendor
    them (pyproject `lightly>=1.5.26`), their published algorithm is restated in
    oracle/dinov2_oracle.py and injected here so the reference class runs.

Nothing here travels to the GPU box: /root/reference does not exist there.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types
from typing import Any

import torch
from torch import nn

REFERENCE_SRC = "/root/reference/src"

_STUB_ROOTS = {
    "pytorch_lightning", "lightning_fabric", "lightly", "torchvision", "albumentations",
    "cv2", "omegaconf", "lightning_utilities", "tensorboard", "matplotlib", "mlflow",
    "wandb", "torchmetrics", "pycocotools", "pynvml", "timm", "xformers", "PIL",
    "pydicom", "onnx", "onnxruntime", "tensorrt", "ultralytics", "super_gradients",
    "rfdetr", "posthog", "eval_type_backport", "lightning",
}


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "lightly_train"))


class _DummyMeta(type):
    def __getattr__(cls, name: str) -> Any:
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        val = _DummyMeta(name, (_Dummy,), {})
        setattr(cls, name, val)
        return val


class _Dummy(metaclass=_DummyMeta):
    """Permissive placeholder: callable, subscriptable, attribute-able."""

    def __init__(self, *a: Any, **k: Any) -> None:
        pass

    def __call__(self, *a: Any, **k: Any) -> Any:
        # used as decorator -> return the function unchanged
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Dummy()

    def __getattr__(self, name: str) -> Any:
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def __getitem__(self, item: Any) -> Any:
        return _Dummy

    def __class_getitem__(cls, item: Any) -> Any:
        return cls

    def __iter__(self):  # type: ignore[no-untyped-def]
        return iter(())

    def __bool__(self) -> bool:
        return False

    def __mro_entries__(self, bases):  # type: ignore[no-untyped-def]
        return (_Dummy,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name: str) -> Any:
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        val = _DummyMeta(name, (_Dummy,), {})
        setattr(self, name, val)
        return val


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):  # type: ignore[no-untyped-def]
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):  # type: ignore[no-untyped-def]
        m = _StubModule(spec.name)
        m.__path__ = []  # type: ignore[attr-defined]
        return m

    def exec_module(self, module):  # type: ignore[no-untyped-def]
        pass


class LightningModuleShim(nn.Module):
    """Just enough of pytorch_lightning.LightningModule for Method/DINOv2."""

    def __init__(self, *a: Any, **k: Any) -> None:
        super().__init__()
        self.trainer: Any = None

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def global_step(self) -> int:
        return int(self.trainer.global_step)

    def log(self, *a: Any, **k: Any) -> None:
        pass

    def log_dict(self, *a: Any, **k: Any) -> None:
        pass

    def clip_gradients(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):  # type: ignore[no-untyped-def]
        params = [p for g in optimizer.param_groups for p in g["params"]]
        torch.nn.utils.clip_grad_norm_(params, gradient_clip_val)


_INSTALLED = False


def install() -> None:
    """Make `lightly_train._methods.dinov2...` importable from the reference tree."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present at /root/reference")
    os.environ.setdefault("XFORMERS_DISABLED", "1")
    sys.meta_path.insert(0, _StubFinder())

    pkg = types.ModuleType("lightly_train")
    pkg.__path__ = [os.path.join(REFERENCE_SRC, "lightly_train")]  # type: ignore[attr-defined]
    sys.modules["lightly_train"] = pkg

    # concrete pieces the reference really executes from the stubs
    pl = importlib.import_module("pytorch_lightning")
    pl.LightningModule = LightningModuleShim  # type: ignore[attr-defined]
    plu = importlib.import_module("pytorch_lightning.utilities")
    plu.rank_zero_only = lambda f: f  # type: ignore[attr-defined]
    lui = importlib.import_module("lightning_utilities.core.imports")

    class RequirementCache:  # noqa: D401
        def __init__(self, *a: Any, **k: Any) -> None:
            pass

        def __bool__(self) -> bool:
            return False

    lui.RequirementCache = RequirementCache  # type: ignore[attr-defined]

    from oracle import dinov2_oracle as O

    ll = importlib.import_module("lightly.loss")
    ll.KoLeoLoss = O.KoLeoLoss  # type: ignore[attr-defined]
    lus = importlib.import_module("lightly.utils.scheduler")
    lus.cosine_schedule = O.cosine_schedule  # type: ignore[attr-defined]
    lus.CosineWarmupScheduler = O.CosineWarmupScheduler  # type: ignore[attr-defined]
    luo = importlib.import_module("lightly.utils.optim")
    luo.update_param_groups = O.update_param_groups  # type: ignore[attr-defined]
    # the LightlySSL pieces of the DINO method (LT/_methods/dino/dino.py:15-17), restated in oracle/dino_oracle.py (parity unpinned)
    from oracle import dino_oracle as ODN

    ll.DINOLoss = ODN.DINOLoss  # type: ignore[attr-defined]
    lmh = importlib.import_module("lightly.models.modules.heads")
    lmh.DINOProjectionHead = ODN.DINOProjectionHead  # type: ignore[attr-defined]
    lmu = importlib.import_module("lightly.models.utils")
    lmu.get_weight_decay_parameters = ODN.get_weight_decay_parameters  # type: ignore[attr-defined]
    lu = importlib.import_module("lightly.utils")
    lu.optim = luo  # type: ignore[attr-defined]
    from oracle import lars_oracle

    lul = importlib.import_module("lightly.utils.lars")
    lul.LARS = lars_oracle.LARS  # type: ignore[attr-defined]
    # the two LightlySSL pieces of DINOv31 (LT/_methods/dinov31/dinov31.py:55), restated in oracle/dinov31_oracle.py (parity unpinned)
    from oracle import dinov31_oracle as O31

    ll.PatchKernelAlignmentLoss = O31.PatchKernelAlignmentLoss  # type: ignore[attr-defined]
    ll.roi_resample_to_grid = O31.roi_resample_to_grid  # type: ignore[attr-defined]
    # torchvision is not installed: the convolutional student runs on the restated ResNet (oracle/resnet_oracle.py), registered
    # where the reference's ResNetModelWrapper imports it from (LT/_models/torchvision/resnet.py:9-10)
    from oracle import resnet_oracle as OR

    tvm = importlib.import_module("torchvision.models")
    tvm.ResNet = OR.ResNet  # type: ignore[attr-defined]
    tvu = importlib.import_module("torchvision.models._utils")
    tvu.IntermediateLayerGetter = OR.IntermediateLayerGetter  # type: ignore[attr-defined]
    ltu = importlib.import_module("lightly.transforms.utils")
    ltu.IMAGENET_NORMALIZE = {"mean": [0.485, 0.456, 0.406], "std": [0.229, 0.224, 0.225]}  # type: ignore[attr-defined]
    _INSTALLED = True


class MockTrainer:
    def __init__(self, total_steps: int, global_step: int = 0) -> None:
        self.global_step = global_step
        self.max_epochs = 1
        self.estimated_stepping_batches = total_steps
        self.train_dataloader = None


def build_reference_method(
    arch: str = "_vit_test",
    patch_size: int = 16,
    img_size: int = 224,
    model_kwargs: dict | None = None,
    method_kwargs: dict | None = None,
    global_batch_size: int = 16,
    total_steps: int = 100,
    seed: int = 0,
    method_cls=None,
    method_cls_kwargs: dict | None = None,
):
    """Instantiate the reference DINOv2 Method (CPU, fp32) around a reference ViT.  `method_cls`: a subclass to instantiate instead
    (lightly_train_amd.integration.DINOv2AMD), built by the same constructor calls from the same seed = identical initial weights."""
    install()
    import random

    from lightly_train._methods.dinov2.dinov2 import DINOv2, DINOv2AdamWViTArgs, DINOv2Args
    from lightly_train._models.dinov2_vit.dinov2_vit import DINOv2ViTModelWrapper
    from lightly_train._models.dinov2_vit.dinov2_vit_src.models import vision_transformer as vits
    from lightly_train._models.embedding_model import EmbeddingModel

    torch.manual_seed(seed)
    random.seed(seed)
    kw = dict(
        img_size=img_size, patch_size=patch_size, init_values=1e-5, drop_path_rate=0.0,
        ffn_layer="mlp", block_chunks=0, interpolate_offset=0.1,
    )
    kw.update(model_kwargs or {})
    model = getattr(vits, arch)(**kw)
    wrapped = DINOv2ViTModelWrapper(model)
    emb = EmbeddingModel(wrapped_model=wrapped)
    margs = DINOv2Args(**(method_kwargs or {}))
    oargs = DINOv2AdamWViTArgs()
    margs.resolve_auto(scaling_info=None, optimizer_args=oargs, wrapped_model=wrapped)  # type: ignore[arg-type]
    method = (method_cls or DINOv2)(
        method_args=margs, optimizer_args=oargs, embedding_model=emb,
        global_batch_size=global_batch_size, num_input_channels=3, **(method_cls_kwargs or {}),
    )
    method.trainer = MockTrainer(total_steps)
    return method


class ReferenceRunner:
    """Drives the reference Method through Lightning's documented hook order
    (SURVEY.md 3.1): training_step_impl -> backward -> on_before_optimizer_step ->
    configure_gradient_clipping -> optimizer.step -> lr_scheduler.step ->
    global_step += 1 -> on_train_batch_end (EMA)."""

    def __init__(self, method) -> None:  # type: ignore[no-untyped-def]
        self.method = method
        [self.optim], [sched] = method.configure_optimizers()
        self.sched = sched["scheduler"]
        # Method.on_train_batch_end also logs batch timing; only the EMA matters here.
        method._log_time_batch_end = lambda *a, **k: None  # type: ignore[attr-defined]
        method._batch_timing_tracker = None

    def split_state(self):  # type: ignore[no-untyped-def]
        sd = self.method.state_dict()
        out = {"student_backbone": {}, "teacher_backbone": {}, "student_head": {}, "teacher_head": {},
               "student_ibot_head": {}, "teacher_ibot_head": {}}
        for k, v in sd.items():
            for role in ("student", "teacher"):
                pre = f"{role}_embedding_model.wrapped_model._model."
                if k.startswith(pre):
                    out[f"{role}_backbone"][k[len(pre):]] = v.detach().clone()
                pre = f"{role}_head.dino_head."
                if k.startswith(pre):
                    out[f"{role}_head"][k[len(pre):]] = v.detach().clone()
                pre = f"{role}_head.ibot_head."
                if k.startswith(pre) and self.method.student_head.ibot_head is not self.method.student_head.dino_head:
                    out[f"{role}_ibot_head"][k[len(pre):]] = v.detach().clone()
        return out

    def train_step(self, views):  # type: ignore[no-untyped-def]
        m = self.method
        res = m.training_step_impl({"views": views, "filename": []}, 0)
        res.loss.backward()
        m.on_before_optimizer_step(self.optim)
        params = [p for g in self.optim.param_groups for p in g["params"]]
        gnorm = torch.nn.utils.clip_grad_norm_(params, m.method_args.gradient_clip_val)
        self.optim.step()
        self.optim.zero_grad(set_to_none=True)
        self.sched.step()
        m.trainer.global_step += 1
        try:
            m.on_train_batch_end(None, {"views": views, "filename": []}, 0)
        except Exception:
            # base-class timing/logging hooks need a real Trainer; EMA has already run.
            pass
        out = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}
        out["loss"] = float(res.loss.detach())
        out["grad_norm"] = float(gnorm)
        return out
