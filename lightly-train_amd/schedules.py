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
    """Teacher-temperature ramp (reference LT/_methods/dinov2/scheduler.py:13-34): linear from `start_value` to `end_value`
    over `warmup_steps`, constant afterwards; the same argument domain is enforced (ValueError outside it)."""
    domain = (
        (warmup_steps >= 0, f"warmup_steps must be >= 0, got {warmup_steps}"),
        (step >= 0, f"step must be >= 0, got {step}"),
        (start_value >= 0, f"start_value must be >= 0, got {start_value}"),
        (end_value > 0, f"end_value must be > 0, got {end_value}"),
        (start_value <= end_value, f"start_value {start_value} exceeds end_value {end_value}"),
    )
    for ok, msg in domain:
        if not ok:
            raise ValueError(msg)
    frac = min(step / warmup_steps, 1.0) if warmup_steps > 0 else 1.0
    return start_value + frac * (end_value - start_value) if frac < 1.0 else end_value
