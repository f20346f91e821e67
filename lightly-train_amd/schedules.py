"""Host-side scalar schedules of the DINOv2 method (pure Python; mirrors the call sites in
LT/_methods/dinov2/dinov2.py:261-266,576-586,600-660 and the un-vendored LightlySSL helpers they call)."""
from __future__ import annotations

import math


def cosine_schedule(step: int, max_steps: int, start_value: float, end_value: float) -> float:
    """lightly.utils.scheduler.cosine_schedule (no period): start -> end over max_steps."""
    if step < 0 or max_steps < 1:
        raise ValueError("invalid step / max_steps")
    if step > max_steps:
        raise ValueError(f"step {step} > max_steps {max_steps}")
    if max_steps == 1 or step == max_steps:
        return end_value
    return end_value - (end_value - start_value) * (math.cos(math.pi * step / (max_steps - 1)) + 1) / 2


def warmup_cosine_lr_factor(step: int, warmup_steps: int, max_steps: int, end_value: float) -> float:
    """Multiplier applied by lightly's CosineWarmupScheduler (interval='step')."""
    if step < warmup_steps:
        return (step + 1) / warmup_steps
    if step >= max_steps:
        return end_value
    return cosine_schedule(step - warmup_steps, max_steps - warmup_steps, 1.0, end_value)


def linear_warmup_schedule(step: int, warmup_steps: int, start_value: float, end_value: float) -> float:
    """LT/_methods/dinov2/scheduler.py:13-34 (same argument validation / errors)."""
    if warmup_steps < 0:
        raise ValueError(f"Warmup steps {warmup_steps} can't be negative.")
    if step < 0:
        raise ValueError(f"Current step number {step} can't be negative.")
    if start_value < 0:
        raise ValueError(f"Start value {start_value} can't be negative.")
    if end_value <= 0:
        raise ValueError(f"End value {end_value} can't be non-positive.")
    if start_value > end_value:
        raise ValueError(f"Start value {start_value} must be less than or equal to end value {end_value}.")
    if step < warmup_steps:
        return start_value + step / warmup_steps * (end_value - start_value)
    return end_value
