"""ops.LaunchPlan / ops.record_plan (round 6): the recorder behind the launch-plan replay of a step's static blocks, without a GPU.
The library is replaced by a stand-in that logs what it is called with; events are stand-ins too (torch.cuda.Event needs a device)."""
import ctypes as C

import pytest
import torch

import lightly_train_amd  # noqa: F401
from lightly_train_amd import _lib, ops


class FakeLib:
    def __init__(self):
        self.log = []
        self.fail_next = False

    def _fn(self, name):
        def f(*args):
            self.log.append((name, tuple(a.value if isinstance(a, C.c_int) else a for a in args)))
            if self.fail_next:
                self.fail_next = False
                return 3
            return 0
        f.__name__ = name
        return f

    def __getattr__(self, name):
        if name == "lt_last_error":
            return lambda: b"stand-in failure"
        return self._fn(name)


@pytest.fixture
def fake_lib(monkeypatch):
    lib = FakeLib()
    monkeypatch.setattr(_lib, "_lib", lib)
    yield lib
    assert _lib._recording is None and ops._active_plan is None


def test_recorded_calls_are_replayed_with_the_same_arguments_in_the_same_order(fake_lib):
    with ops.record_plan() as plan:
        assert _lib.load() is not fake_lib                       # the logging proxy
        _lib.load().lt_layernorm_fwd(1, 2, 3, 4, None, 5, 6, 128, 768, 1e-6, 99)
        _lib.load().lt_attention_bwd_ws_floats(1, 2, 3, 64)      # a query, not a launch: forwarded, not logged
        ops.recordable(lambda: fake_lib.log.append(("torch fill", ())))
        _lib.load().lt_reduce_flush(99)
    assert _lib.load() is fake_lib
    assert [k for k, _, _ in plan.ops] == [0, 3, 0] and plan.counts() == {"launches": 2, "event_records": 0, "event_waits": 0, "callables": 1}
    made = [e for e in fake_lib.log if e[0] != "lt_attention_bwd_ws_floats"]
    fake_lib.log.clear()
    plan.replay()
    assert fake_lib.log == made
    plan.replay()
    assert fake_lib.log == made + made


def test_a_failing_replayed_call_raises_like_the_eager_call(fake_lib):
    with ops.record_plan() as plan:
        _lib.load().lt_ema_flat(1, 2, 3, 4, 0.5, 0.5, 7)
    fake_lib.fail_next = True
    with pytest.raises(_lib.LtAmdError, match="stand-in failure"):
        plan.replay()


def test_event_edges_are_logged_and_the_event_class_is_restored(fake_lib):
    rec0, wait0 = torch.cuda.Event.record, torch.cuda.Event.wait
    seen = []

    class Ev:
        pass

    # stand-ins for the two torch methods the recorder wraps (a real torch.cuda.Event needs a device)
    ops._EV_RECORD, ops._EV_WAIT = (lambda ev, st: seen.append(("record", ev, st))), (lambda ev, st: seen.append(("wait", ev, st)))
    try:
        e, s1, s2 = Ev(), object(), object()
        with ops.record_plan() as plan:
            assert torch.cuda.Event.record is not rec0
            torch.cuda.Event.record(e, s1)
            _lib.load().lt_reduce_flush(5)
            torch.cuda.Event.wait(e, s2)
        assert torch.cuda.Event.record is rec0 and torch.cuda.Event.wait is wait0
        assert seen == [("record", e, s1), ("wait", e, s2)]
        assert plan.counts() == {"launches": 1, "event_records": 1, "event_waits": 1, "callables": 0}
        seen.clear()
        plan.replay()
        assert seen == [("record", e, s1), ("wait", e, s2)]
    finally:
        ops._EV_RECORD, ops._EV_WAIT = rec0, wait0


def test_a_recording_cut_short_by_an_exception_does_not_block_the_next_one(fake_lib):
    r = ops.record_plan()
    r.__enter__()
    _lib.load().lt_reduce_flush(1)             # ... and the step raised here: __exit__ never ran
    with ops.record_plan() as plan:
        _lib.load().lt_reduce_flush(2)
    assert [b for _, _, b in plan.ops] == [(2,)]
    assert torch.cuda.Event.record is ops._EV_RECORD
