"""-m gpu: the N > 1 data-parallel step end to end on real kernels.  The GPU boxes have one device, so two ranks share
cuda:0 and talk over gloo (RCCL refuses two ranks on one GPU); every collective of the step is exercised: async center
all-reduces (softmax centering), synchronous Sinkhorn row sums, the bucketed gradient mean.

Both ranks are fed the SAME views and masks.  Then every all-reduce averages identical values and the result must equal
the single-process run with the same per-rank batch and the same `global_batch_size` (LR scale) -- which pins the
scaling factors (1/world on gradients, /(2B*world) on the DINO center, per-rank mean/world on the iBOT center, the
Sinkhorn totals) and the ordering of async handles across streams."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_steps(center_method: str, n_steps: int = 3):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_step as T

    fx = torch.load(os.path.join(ROOT, "tests", "golden", "step_d64_softmax.pt"), weights_only=False)
    m = T.build(fx, koleo_loss_weight=0.0, center_method=center_method)
    losses = []
    for s in range(n_steps):
        rec = fx["steps"][min(s, len(fx["steps"]) - 1)]
        views = T.synth_views(rec["view_seed"] + 10 * s, fx["b"], fx["g_size"], fx["l_size"], fx["n_local"])
        res = m.train_step(views, masks=rec["masks"])
        losses.append(float(res.loss))
    torch.cuda.synchronize()
    return losses, m.student.data.detach().cpu().clone(), m.teacher.data.detach().cpu().clone(), m.dino_center.cpu().clone()


def _worker(rank: int, world: int, port: int, out_dir: str, center_method: str) -> None:
    import sys
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        out = _run_steps(center_method)
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("center_method", ["softmax", "sinkhorn_knopp"])
def test_two_ranks_match_single_process(tmp_path, center_method):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), center_method), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    single = _run_steps(center_method)
    for a, b in ((r0, r1), (r0, single)):
        assert a[0] == pytest.approx(b[0], rel=2e-4), "loss per step"
        for i, nm in ((1, "student"), (2, "teacher"), (3, "dino center")):
            d = (a[i] - b[i]).abs().max().item()
            assert d <= 2e-4 * max(1.0, b[i].abs().max().item()), (nm, d)


def _overlap_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import sys
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import test_gpu_step as T

        fx = torch.load(os.path.join(ROOT, "tests", "golden", "step_d64_softmax.pt"), weights_only=False)
        out = {}
        for key, overlap, streams in (("overlap", True, True), ("overlap_one_stream", True, False), ("after", False, True)):
            m = T.build(fx, koleo_loss_weight=0.0)
            m.overlap_grad_reduce = overlap
            m.overlap_streams = streams
            early = []
            for s in range(3):
                rec = fx["steps"][min(s, len(fx["steps"]) - 1)]
                views = T.synth_views(rec["view_seed"] + 10 * s + 1000 * rank, fx["b"], fx["g_size"], fx["l_size"], fx["n_local"])  # per-rank data
                m.training_step_impl({"views": views}, 0, masks=rec["masks"])
                early.append(sum(b - a for a, b in m._grad_sync.covered) if m._grad_sync is not None else 0)
                m.optimizer_step()
                m.on_train_batch_end()
            torch.cuda.synchronize()
            out[key] = (m.student.data.cpu().clone(), early, m.student.numel)
        torch.save(out, os.path.join(out_dir, f"o{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_overlapped_with_backward(tmp_path):
    """Different images on the two ranks: after three steps the parameters are bit-identical across ranks only if every
    element of the gradient went through the all-reduce exactly once; starting the head / per-block all-reduces during
    backward (the default) gives the same parameters as one all-reduce after it, and covers nearly all of the buffer early."""
    mp.spawn(_overlap_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "o0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "o1.pt", weights_only=False)
    for key in ("overlap", "overlap_one_stream", "after"):
        bad = (r0[key][0] != r1[key][0]).nonzero().flatten()
        assert bad.numel() == 0, f"ranks diverged ({key}): {bad.numel()} elements, first at {bad[:4].tolist()}, last {bad[-1].item()}, max diff {(r0[key][0] - r1[key][0]).abs().max().item()}"
    b = r0["after"][0]
    for key in ("overlap", "overlap_one_stream"):
        assert (r0[key][0] - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item()), key
        early, numel = r0[key][1], r0[key][2]
        assert all(e > 0.5 * numel for e in early), (key, early, numel)   # heads + every block were in flight before backward ended
    assert all(e == 0 for e in r0["after"][1])


def _joint_ddp_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import random
    import sys
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import test_gpu_step as T

        fx = torch.load(os.path.join(ROOT, "tests", "golden", "step_d64_softmax.pt"), weights_only=False)
        B = 32       # 2 x 32 x 37 and 2 x 32 x 10 token rows: whole 64-row K-tiles, the condition for the joint weight-gradient launches
        out = {}
        for key, overlap, joint, head_side in (("joint_overlap", True, 1, 1), ("separate_after", False, 0, 0)):
            m = T.build(fx, koleo_loss_weight=0.0)
            m.overlap_grad_reduce, m.joint_wgrad, m.head_side = overlap, joint, head_side
            early = []
            for s in range(3):
                views = T.synth_views(4000 + 10 * s + 1000 * rank, B, fx["g_size"], fx["l_size"], fx["n_local"])   # per-rank data
                random.seed(50 + s)          # the same masks on both schedules (drawn by the step from the global generator)
                m.training_step_impl({"views": views}, 0)
                early.append(sum(b - a for a, b in m._grad_sync.covered) if m._grad_sync is not None else 0)
                m.optimizer_step()
                m.on_train_batch_end()
            torch.cuda.synchronize()
            out[key] = (m.student.data.cpu().clone(), early, m.student.numel, m._joint.launched if m._joint is not None else 0)
        torch.save(out, os.path.join(out_dir, f"j{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_joint_weight_gradients_and_head_side_stream_under_data_parallel(tmp_path):
    """The round-4 schedule changes in the data-parallel step: joint weight gradients of the two student passes (their GEMMs are launched
    when the SECOND pass deposits, and a block's all-reduce starts only after them) and the heads' weight gradients on the side stream
    (the head span's all-reduce is ordered after that stream).  Two ranks, different images: bit-identical parameters across ranks, equal
    to the separate / one-all-reduce-at-the-end schedule to fp32 summation order, and the all-reduces still start during backward."""
    mp.spawn(_joint_ddp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "j0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "j1.pt", weights_only=False)
    for key in ("joint_overlap", "separate_after"):
        assert torch.equal(r0[key][0], r1[key][0]), key
    a, b = r0["joint_overlap"], r0["separate_after"]
    assert a[3] >= 3 * (4 * 2 - 3) and b[3] == 0                   # joint launches happened (all but the last block's row-subset layers), and only there
    assert (a[0] - b[0]).abs().max().item() <= 2e-5 * max(1.0, b[0].abs().max().item())
    assert all(e > 0.5 * a[2] for e in a[1]) and all(e == 0 for e in b[1])


def _run_distill(n_steps: int = 3):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_distill as TD

    fx = torch.load(os.path.join(ROOT, "tests", "golden", "distill_v3_d64.pt"), weights_only=False)
    m = TD.build(fx)
    losses = []
    for rec in fx["steps"][:n_steps]:
        x = torch.randn(fx["b"], 3, 64, 64, generator=torch.Generator().manual_seed(rec["x_seed"]))
        losses.append(float(m.train_step(x, mix=(rec["lam"], rec["index"])).loss))
    torch.cuda.synchronize()
    return losses, m.student.data.detach().cpu().clone(), m.teacher_queue.cpu().clone()


def _distill_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import sys
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        torch.save(_run_distill(), os.path.join(out_dir, f"d{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _distill_overlap_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import sys
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import test_gpu_distill as TD

        fx = torch.load(os.path.join(ROOT, "tests", "golden", "distill_v3_d64.pt"), weights_only=False)
        out = {}
        for overlap in (True, False):
            m = TD.build(fx)
            m.overlap_grad_reduce = overlap
            for rec in fx["steps"][:3]:
                x = torch.randn(fx["b"], 3, 64, 64, generator=torch.Generator().manual_seed(rec["x_seed"] + 1000 * rank))  # per-rank images
                m.train_step(x, mix=(rec["lam"], rec["index"]))
            torch.cuda.synchronize()
            out[overlap] = m.student.data.cpu().clone()
        torch.save(out, os.path.join(out_dir, f"do{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_distillation_gradient_allreduce_overlapped(tmp_path):
    """DistillationV3 with different images per rank: replicas stay bit-identical, overlapped == after-backward reduction."""
    mp.spawn(_distill_overlap_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "do0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "do1.pt", weights_only=False)
    for overlap in (True, False):
        assert torch.equal(r0[overlap], r1[overlap]), f"ranks diverged (overlap={overlap})"
    assert (r0[True] - r0[False]).abs().max().item() <= 1e-5 * max(1.0, r0[False].abs().max().item())


def test_distillation_two_ranks_match_single_process(tmp_path):
    """DistillationV3 under data parallelism: per-GPU queues, gradient mean -- same data on both ranks == single process."""
    mp.spawn(_distill_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "d0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "d1.pt", weights_only=False)
    single = _run_distill()
    for a, b in ((r0, r1), (r0, single)):
        assert a[0] == pytest.approx(b[0], rel=2e-4)
        for i in (1, 2):
            assert (a[i] - b[i]).abs().max().item() <= 2e-4 * max(1.0, b[i].abs().max().item())


def test_bench_contract_two_ranks():
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), with the two ranks
    folded onto cuda:0 over gloo (LT_BENCH_BACKEND): every rank has to take part in every step that contains collectives
    (incl. the instrumented roofline step), rank 0 prints ONE JSON line with whole-job throughput."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, LT_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--model", "vit_small", "--batch", "8", "--out-dim", "4096", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["scaling"] == "weak"
    assert out["value"] == pytest.approx(16 * out["steps"] / (out["ms_per_step"] * out["steps"] / 1e3), rel=1e-2)
    assert out["roofline"]["launches_per_step"] > 0 and out["cpu_baseline"] is None
    assert out["comm"]["exposed_allreduce_ms_per_step"] >= 0 and out["comm"]["gradient_bytes_per_step"] > 0


def _resnet_case(rows_of_rank=None, sync=True):
    """Forward + backward of a small bottleneck ResNet on (a slice of) one fixed batch; returns features, parameter gradients, buffers."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lightly_train_amd  # noqa: F401
    import test_gpu_distill as TD
    from lightly_train_amd.params import FlatParams
    from lightly_train_amd.resnet import ResNetConfig, ResNetEngine, flat_named
    from lightly_train_amd.vit import Workspace

    cfg = ResNetConfig(layers=(1, 1), width=16)
    g = torch.Generator().manual_seed(11)
    sd = TD._perturbed_resnet_state(cfg, g)
    B, S = 8, 64
    x = torch.randn(B, 3, S, S, generator=g)
    fp = FlatParams(flat_named(cfg, sd), "cuda", True)
    eng = ResNetEngine(cfg, fp, "", buffers=sd)
    if not sync:
        eng.bn_sync = None
    sl = slice(0, B) if rows_of_rank is None else rows_of_rank
    ws = Workspace(torch.device("cuda"))
    ctx = eng.forward(ws, "r", x[sl].cuda(), save=True, train=True)
    hw = ctx["h"] * ctx["w"]
    d_all = torch.randn(B * hw, cfg.feature_dim, generator=g) * 0.1
    n = (sl.stop - sl.start) * hw
    dfeat = torch.zeros_like(ctx["feat"])
    dfeat[:n] = d_all[sl.start * hw: sl.stop * hw].to(torch.bfloat16).cuda()
    fp.grad.zero_()
    eng.backward(ws, ctx, dfeat)
    torch.cuda.synchronize()
    return (ctx["feat"][:n].float().cpu(), {k: fp.g[k].float().cpu().clone() for k in fp.names},
            {k: v.float().cpu().clone() for k, v in eng.buffers.items()})


def _syncbn_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import sys
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        torch.save(_resnet_case(slice(4 * rank, 4 * rank + 4)), os.path.join(out_dir, f"bn{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_equal_one_process_on_the_whole_batch(tmp_path):
    """SyncBatchNorm in the convolutional student (the reference trains with sync_batchnorm=True on GPUs, train_helpers.py:223): two ranks
    holding half of a batch each produce the feature rows, the (rank-summed) parameter gradients and the running estimates of one process
    on the whole batch -- up to the bf16 rounding flips a BatchNorm network amplifies -- while per-rank statistics (the control) do not."""
    port = _free_port()
    mp.spawn(_syncbn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"bn{i}.pt", weights_only=False) for i in range(2)]
    feat_w, grad_w, buf_w = _resnet_case()
    ctl = [_resnet_case(slice(4 * i, 4 * i + 4), sync=False) for i in range(2)]        # one process, per-part statistics

    def fro(a, b):
        return ((a - b).norm() / (b.norm() + 1e-20)).item()

    e_sync = fro(torch.cat([r[0][0], r[1][0]]), feat_w)
    e_ctl = fro(torch.cat([ctl[0][0], ctl[1][0]]), feat_w)
    assert e_sync < 2e-2 and e_ctl > 4 * e_sync, (e_sync, e_ctl)
    errs = {k: fro(r[0][1][k] + r[1][1][k], grad_w[k]) for k in grad_w}
    errs_ctl = {k: fro(ctl[0][1][k] + ctl[1][1][k], grad_w[k]) for k in grad_w}
    import statistics
    assert statistics.median(errs.values()) < 4e-2 and max(errs.values()) < 0.2, sorted(errs.items(), key=lambda t: -t[1])[:5]
    assert statistics.median(errs_ctl.values()) > 3 * statistics.median(errs.values()), (statistics.median(errs_ctl.values()), statistics.median(errs.values()))
    for k in buf_w:
        assert torch.equal(r[0][2][k], r[1][2][k]), k                              # both ranks end with the same running estimates
        if k.endswith("running_var"):
            assert fro(r[0][2][k], buf_w[k]) < 2e-2, k
        elif k.endswith("running_mean"):
            assert (r[0][2][k] - buf_w[k]).abs().max().item() < 2e-2, k


def _rccl_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    """ONE rank on a real RCCL communicator (two ranks cannot share the box's single GPU: profiles/r03g_rccl_one_gpu.log), with the step
    told that it is one of two: every collective of the data-parallel path (async center sums, Sinkhorn row sums, the head / per-block
    gradient all-reduces started during backward) is issued as a real RCCL operation on RCCL's own stream -- identity in value, but with
    the stream semantics gloo does not have."""
    import random
    import sys
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import lightly_train_amd  # noqa: F401
        from lightly_train_amd import parallel
        from lightly_train_amd.dinov2 import DINOv2, DINOv2Args
        from lightly_train_amd.vit import ViTConfig

        parallel.world_size = lambda: 2
        DINOv2.world = property(lambda self: 2)
        cfg = ViTConfig(embed_dim=384, depth=3, num_heads=6, mlp_ratio=4.0, patch_size=16, img_size=224, init_values=1e-2, drop_path_rate=0.1)
        B = 8
        g = torch.Generator().manual_seed(0)
        views = [torch.randn(B, 3, 224, 224, generator=g) for _ in range(2)] + [torch.randn(B, 3, 96, 96, generator=g) for _ in range(8)]
        out = {}
        for key, center, overlap in (("overlap", "softmax", True), ("overlap_again", "softmax", True), ("after", "softmax", False),
                                     ("sk_overlap", "sinkhorn_knopp", True), ("sk_after", "sinkhorn_knopp", False), ("abi", "softmax", True)):
            # "abi": the gradient all-reduces through the library's own communicator handle (lt_comm_*, include/lt_amd.h) instead of
            # torch.distributed's -- a second RCCL communicator of this process on the same device
            os.environ["LT_GRAD_COMM"] = "abi" if key == "abi" else ""
            if key == "abi":      # (a one-rank communicator: the step merely believes it is one of two)
                parallel.AbiComm._instance = parallel.AbiComm(0, 1, parallel.AbiComm.unique_id())
            m = DINOv2(cfg, DINOv2Args(output_dim=8192, hidden_dim=512, dino_bottleneck_dim=256, center_method=center), global_batch_size=2 * B,
                       total_steps=100, device="cuda", seed=3)
            m.overlap_grad_reduce = overlap
            early, losses = [], []
            for s in range(3):
                random.seed(50 + s)
                res = m.training_step_impl({"views": views}, 0)
                early.append(sum(b - a for a, b in m._grad_sync.covered) if m._grad_sync is not None else 0)
                m.optimizer_step()
                m.on_train_batch_end()
                losses.append(float(res.loss))
            torch.cuda.synchronize()
            out[key] = (m.student.data.cpu().clone(), m.teacher.data.cpu().clone(), m.dino_center.cpu().clone(), early, m.student.numel, losses)
        torch.save(out, os.path.join(out_dir, "rccl.pt"))
    finally:
        dist.destroy_process_group()


def test_collectives_over_rccl_overlapped_equal_sequential(tmp_path):
    """The data-parallel code path over RCCL itself (one rank, told it is one of two).  With the order-fixed reductions a step is
    bitwise reproducible, so the comparison is exact: all-reduces started during backward on RCCL's stream == one all-reduce after it
    == the same run again, for both centering methods; and the early ranges cover most of the gradient buffer."""
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = torch.load(tmp_path / "rccl.pt", weights_only=False)
    for a, b in (("overlap", "overlap_again"), ("overlap", "after"), ("sk_overlap", "sk_after"), ("overlap", "abi")):
        for i, what in enumerate(("student", "teacher", "center")):
            assert torch.equal(r[a][i], r[b][i]), (a, b, what, (r[a][i] - r[b][i]).abs().max().item())
        assert r[a][5] == r[b][5], (a, b, "losses")
    assert all(e > 0.5 * r["overlap"][4] for e in r["overlap"][3]) and all(e == 0 for e in r["after"][3])
    assert all(x == x for x in r["overlap"][5])     # finite losses
