"""CPU, two ranks over gloo, exact arithmetic (tests/tools/ops_emu.py): the data-parallel DINOv2 step -- each rank with ITS OWN images and
masks -- against ONE process on the concatenated batch.  With equal per-rank batches the reference's semantics make the two the same
model: the per-rank mean losses average to the global mean, the DINO center sums over 2B * world rows, the iBOT center is the mean of the
per-rank means, the Sinkhorn column sums and totals run over all ranks, the gradients are averaged, and (batch_norm=True) SyncBatchNorm
takes the statistics of every head call over all ranks' rows.  Every one of those factors / collectives is pinned here to fp32 round-off:
loss, parameters after two optimizer steps, EMA teacher, centers, BatchNorm running estimates.  KoLeo (a nearest-neighbour term inside
the LOCAL batch) is off.  The early per-block gradient all-reduce under backward needs streams and is exercised on the GPU
(tests/test_gpu_ddp.py); here the gradients are reduced in one go before the optimizer."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _masks_for(seed: int, n_crops: int, n_p: int):
    """Random block-free masks in the collated format of create_collated_masks (utils.py:120-152): half of the crops masked, at positions
    of the rank's own but with the same COUNT per crop on every rank -- the iBOT center is the mean over ranks of the per-rank means over
    the masked patches (dinov2_loss.py:274-282), which is the global mean only when the ranks hold equally many."""
    g = torch.Generator().manual_seed(seed)
    cm = torch.zeros(n_crops, n_p, dtype=torch.bool)
    for c in range(0, n_crops, 2):
        k = 2 + (c // 2) % max(1, n_p // 2 - 2)
        cm[c, torch.randperm(n_p, generator=g)[:k]] = True
    return _collate(cm)


def _collate(cm):
    w = (1.0 / cm.sum(-1).clamp(min=1.0)).unsqueeze(-1).expand_as(cm)[cm]
    return {"collated_masks": cm, "mask_indices_list": cm.flatten().nonzero().flatten(), "masks_weight": w,
            "upperbound": int(cm.sum()), "n_masked_patches": torch.tensor([int(cm.sum())])}


def _views(seed, b, g_size, l_size, n_local):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(b, 3, g_size, g_size, generator=g) for _ in range(2)] + [torch.randn(b, 3, l_size, l_size, generator=g) for _ in range(n_local)]


def _run(name: str, ranks, world: int, n_steps: int = 2):
    """The steps of the ranks in `ranks`: one entry = this process is that rank of `world`; both = one process on the concatenated batch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import ops_emu
    import test_dinov2_method_cpu as T
    from lightly_train_amd import ops

    torch.cuda.current_stream = lambda *a, **k: T._NoStream()
    torch.cuda.set_stream = lambda s: None
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    b, n_p = fx["b"], (fx["g_size"] // fx["cfg"]["patch_size"]) ** 2
    fx = dict(fx, b=b * 2)          # global batch (LR scale) = two ranks' worth in every variant
    with ops_emu.emulate(ops):
        m = T.build_exact(fx, koleo_loss_weight=0.0)
        assert m.world == world
        losses = []
        for s in range(n_steps):
            vs = [_views(500 + 10 * s + r, b, fx["g_size"], fx["l_size"], fx["n_local"]) for r in ranks]
            ms = [_masks_for(900 + 10 * s + r, 2 * b, n_p) for r in ranks]
            if len(ranks) == 1:
                views, masks = vs[0], ms[0]
            else:   # crop-major layout of the concatenated batch: view 0 of all images, then view 1 of all images
                views = [torch.cat([v[i] for v in vs]) for i in range(len(vs[0]))]
                cm = torch.cat([ms[0]["collated_masks"][:b], ms[1]["collated_masks"][:b], ms[0]["collated_masks"][b:], ms[1]["collated_masks"][b:]])
                masks = _collate(cm)
            res = m.train_step(views, masks=masks)
            losses.append(float(res.loss))
        m._apply_center_updates()
        bufs = {k: v for h in (m.s_head, m.t_head) for k, v in h.buffer_state().items()}
        return dict(loss=losses, student=m.student.data.clone(), teacher=m.teacher.data.clone(), dino_center=m.dino_center.clone(),
                    ibot_center=m.ibot_center.clone(), bufs_s=m.s_head.buffer_state(), bufs_t=m.t_head.buffer_state(), names=list(m.student.names),
                    offsets=dict(m.student.offsets))


def _worker(rank: int, world: int, port: int, out_dir: str, name: str, env: dict = None, tag: str = "") -> None:
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.update(env or {})
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"all_reduce": 0}
    real = dist.all_reduce

    def counted(*a, **k):
        calls["all_reduce"] += 1
        return real(*a, **k)

    dist.all_reduce = counted
    try:
        out = _run(name, [rank], world)
        out["all_reduce_calls"] = calls["all_reduce"]
        torch.save(out, os.path.join(out_dir, f"{tag}r{rank}.pt"))
    finally:
        dist.all_reduce = real
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["step_d64_softmax", "step_vittest_sinkhorn", "step_d64_bn"])
def test_two_ranks_with_their_own_data_equal_one_process_on_all_of_it(tmp_path, name):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), name), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=False) for i in range(2)]
    one = _run(name, [0, 1], 1)
    bn = "bn" in name
    for s in range(len(one["loss"])):
        assert 0.5 * (r[0]["loss"][s] + r[1]["loss"][s]) == pytest.approx(one["loss"][s], rel=3e-5), s
    assert torch.equal(r[0]["student"], r[1]["student"]) and torch.equal(r[0]["teacher"], r[1]["teacher"])     # replicas stay identical
    for key in ("student", "teacher"):
        d = (r[0][key] - one[key]).abs()
        if bn:   # gradient-free biases in front of BatchNorm move by round-off-driven +-lr steps (tests/test_dinov2_method_cpu.py)
            for n_ in one["names"]:
                if n_.endswith(("mlp.0.bias", "mlp.3.bias")) or n_ == "backbone.norm.bias":
                    o = one["offsets"][n_]
                    d[o:o + 1024] = 0
        assert d.max().item() < 5e-6, (key, d.max().item())
    assert torch.allclose(r[0]["dino_center"], one["dino_center"], atol=1e-6) and torch.allclose(r[0]["ibot_center"], one["ibot_center"], atol=1e-6)
    for which in ("bufs_s", "bufs_t"):
        for k, v in one[which].items():
            assert torch.equal(r[0][which][k], r[1][which][k]), k
            assert torch.allclose(r[0][which][k].float(), v.float(), atol=5e-5 if k.endswith("running_mean") else 2e-6), (which, k)


def test_joint_sinkhorn_allreduce_is_bit_identical_to_one_head_at_a_time(tmp_path):
    """Sinkhorn-Knopp centering under data parallelism (dinov2_loss.py:97-106,200-215): the DINO and the iBOT head iterate in lockstep and
    each iteration's prototype sums of BOTH heads cross the ranks in one all-reduce (`DINOv2._sinkhorn_joint`) -- 3 collectives per step
    where one head at a time (`LT_SINKHORN_JOINT=0`) takes 6.  Two gloo ranks with their own data, two optimizer steps: every parameter,
    the EMA teacher and the losses are bit-identical between the two schedules, and the collective count drops by 3 per step."""
    outs = {}
    for tag, env in (("joint_", {"LT_SINKHORN_JOINT": "1"}), ("sep_", {"LT_SINKHORN_JOINT": "0"})):
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "step_vittest_sinkhorn", env, tag), nprocs=2, join=True)
        outs[tag] = [torch.load(tmp_path / f"{tag}r{i}.pt", weights_only=False) for i in range(2)]
    j, s_ = outs["joint_"], outs["sep_"]
    for r in range(2):
        assert j[r]["loss"] == s_[r]["loss"]
        assert torch.equal(j[r]["student"], s_[r]["student"]) and torch.equal(j[r]["teacher"], s_[r]["teacher"])
    assert torch.equal(j[0]["student"], j[1]["student"])
    n_steps = len(j[0]["loss"])
    assert s_[0]["all_reduce_calls"] - j[0]["all_reduce_calls"] == 3 * n_steps, (s_[0]["all_reduce_calls"], j[0]["all_reduce_calls"])


def _run_dino_v1(ranks, world: int, n_steps: int = 3):
    """DINO v1 (lightly_train_amd/dino.py): the EMA-first step with the center all-reduce of lightly's `Center.update` and the gradient
    mean, per-rank views; concatenated crop-major for the single process."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import ops_emu
    import test_dino_v1_cpu as TD
    import test_dinov2_method_cpu as T
    from lightly_train_amd import ops

    torch.cuda.current_stream = lambda *a, **k: T._NoStream()
    torch.cuda.set_stream = lambda s: None
    torch.cuda.Stream = T._NoStream
    fx = torch.load(os.path.join(GOLD, "dino_v1_d64.pt"), weights_only=False)
    b = fx["b"]
    fx = dict(fx, b=2 * b)          # global batch (LR scale) = two ranks' worth in every variant
    with ops_emu.emulate(ops):
        m = TD.exact(TD.build(fx))
        assert m.world == world
        losses = []
        for s in range(n_steps):
            vs = [_views(700 + 10 * s + r, b, fx["g_size"], fx["l_size"], fx["n_local"]) for r in ranks]
            views = vs[0] if len(ranks) == 1 else [torch.cat([v[i] for v in vs]) for i in range(len(vs[0]))]
            losses.append(float(m.train_step(views).loss))
        return dict(loss=losses, student=m.student.data.clone(), teacher=m.teacher.data.clone(), center=m.center.clone())


def _worker_dino_v1(rank: int, world: int, port: int, out_dir: str) -> None:
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.save(_run_dino_v1([rank], world), os.path.join(out_dir, f"d{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_dino_v1_two_ranks_with_their_own_data_equal_one_process_on_all_of_it(tmp_path):
    """Three steps (the last layer unfreezes after two): per-rank mean losses average to the global mean (every (t, s) pair term is a mean
    over the batch), the center moves by the mean over views, batch AND ranks, the SGD step sees the rank-averaged gradient."""
    mp.spawn(_worker_dino_v1, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"d{i}.pt", weights_only=False) for i in range(2)]
    one = _run_dino_v1([0, 1], 1)
    for s in range(len(one["loss"])):
        assert 0.5 * (r[0]["loss"][s] + r[1]["loss"][s]) == pytest.approx(one["loss"][s], rel=3e-5), s
    assert torch.equal(r[0]["student"], r[1]["student"]) and torch.equal(r[0]["teacher"], r[1]["teacher"])
    for key in ("student", "teacher"):
        assert (r[0][key] - one[key]).abs().max().item() < 5e-6, key
    assert torch.allclose(r[0]["center"], one["center"], atol=1e-6)
