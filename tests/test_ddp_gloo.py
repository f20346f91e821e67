"""CPU, world_size 2 over gloo: the N>1 data-parallel path (gradient mean in buckets, batch sharding rule)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.parallel import GradSync, per_rank_batch, world_size

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert world_size() == 2 and per_rank_batch(256, world) == 128
        g = torch.Generator().manual_seed(100 + rank)
        grad = torch.randn(10_000, generator=g)
        mine = grad.clone()
        sync = GradSync(grad, bucket_bytes=4096 * 4)  # 3 buckets: 4096 + 4096 + 1808
        assert len(sync.ranges) == 3 and sync.ranges[-1] == (8192, 10_000)
        sync.start(5000, 9200)   # a range that became final during backward (split at the bucket size: 4096 + 104)
        sync.start(100, 200)
        assert sync.uncovered(0, 10_000) == [(0, 100), (200, 5000), (9200, 10_000)] and len(sync.handles) == 3
        sync.start(4000, 6000)   # overlapping request: only the part not reduced yet
        assert sync.uncovered(0, 10_000) == [(0, 100), (200, 4000), (9200, 10_000)]
        sync.finish()            # the rest, then wait + 1/world
        assert sync.covered == [] and sync.handles == []
        others = [torch.randn(10_000, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        expect = sum(others) / world
        assert torch.allclose(grad, expect, atol=1e-6)
        assert torch.equal(others[rank], mine)
        # center sums: async all-reduce consumed later (dinov2_loss.py:139-160 semantics)
        cs = torch.full((8,), float(rank + 1))
        h = dist.all_reduce(cs, async_op=True)
        h.wait()
        assert torch.equal(cs, torch.full((8,), 3.0))
        # logging scalars: one coalesced all-reduce instead of one per scalar (method.py:131-144, sync_dist=True)
        from lightly_train_amd.parallel import coalesced_mean
        vals = coalesced_mean([torch.tensor(float(rank)), torch.tensor(10.0 + rank), torch.tensor(2.0)])
        assert [round(float(v), 6) for v in vals] == [0.5, 10.5, 2.0]
        torch.save(grad, os.path.join(out_dir, f"grad{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_grad_sync_two_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / "grad0.pt")
    b = torch.load(tmp_path / "grad1.pt")
    assert torch.equal(a, b), "ranks disagree after the gradient all-reduce"


def test_batch_must_divide():
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.parallel import bucket_ranges, per_rank_batch

    with pytest.raises(ValueError):
        per_rank_batch(100, 8)
    assert bucket_ranges(10, 4) == [(0, 4), (4, 8), (8, 10)]


def _rsag_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.parallel import GradSync

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 3 * GradSync.RSAG_MIN_ELEMS + 1001            # odd length: tails that do not divide by the world size
        base = torch.randn(n, generator=torch.Generator().manual_seed(7 + rank))
        out = {}
        for mode in ("", "rsag"):
            os.environ["LT_GRAD_COMM_MODE"] = mode
            grad = base.clone()
            sync = GradSync(grad, bucket_bytes=(GradSync.RSAG_MIN_ELEMS + 333) * 4)
            assert sync.rsag == (mode == "rsag")
            sync.start(GradSync.RSAG_MIN_ELEMS, 2 * GradSync.RSAG_MIN_ELEMS + 50)   # a range that became final early (split at the bucket size)
            sync.start(10, 300)                                                      # a short one: stays an all-reduce
            sync.finish()
            assert not sync.handles and not sync._shards and not sync.covered
            out[mode] = grad
        torch.save(out, os.path.join(out_dir, f"rsag{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_grad_sync_reduce_scatter_all_gather_mode_equals_all_reduce(tmp_path):
    """LT_GRAD_COMM_MODE=rsag: the large calls of the gradient exchange as reduce-scatter + all-gather.  Two ranks over gloo:
    the mean is the all-reduce path's bit for bit (two addends commute), the ranks end identical, tails that do not divide by the world
    size and short calls are covered."""
    mp.spawn(_rsag_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rsag0.pt")
    r1 = torch.load(tmp_path / "rsag1.pt")
    for mode in ("", "rsag"):
        assert torch.equal(r0[mode], r1[mode]), mode
    assert torch.equal(r0[""], r0["rsag"])
    n = r0[""].numel()
    want = sum(torch.randn(n, generator=torch.Generator().manual_seed(7 + r)) for r in range(2)) / 2
    assert torch.allclose(r0["rsag"], want, atol=1e-6)
