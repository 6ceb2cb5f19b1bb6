"""SURVEY.md §8f-4 on CPU: Lovasz-Softmax against a per-class restatement of the published algorithm, IoU bookkeeping,
and the bucketed gradient all-reduce with gloo (world_size 2)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import torch

from lattice_net_amd.losses import LovaszSoftmax, Scores, nll_loss_gather


def lovasz_per_class_reference(logp: np.ndarray, target: np.ndarray, ignore: int) -> float:
    p = np.exp(logp.astype(np.float64))
    losses = []
    for c in range(p.shape[1]):
        if c == ignore:
            continue
        fg = (target == c).astype(np.float64)
        if fg.sum() == 0:
            continue
        err = np.abs(fg - p[:, c])
        order = np.argsort(-err, kind="stable")
        err_s, fg_s = err[order], fg[order]
        inter = fg_s.sum() - np.cumsum(fg_s)
        union = fg_s.sum() + np.cumsum(1 - fg_s)
        jac = 1 - inter / union
        jac[1:] = jac[1:] - jac[:-1]
        losses.append(float(err_s @ jac))
    return float(np.mean(losses))


def test_lovasz_softmax_matches_per_class_algorithm_and_is_differentiable():
    rng = np.random.default_rng(0)
    n, c = 500, 6
    logits = torch.tensor(rng.standard_normal((n, c)), dtype=torch.float64, requires_grad=True)
    target = rng.integers(0, c - 1, n)  # class c-1 never occurs: it must be skipped
    logp = torch.log_softmax(logits, 1)
    loss = LovaszSoftmax(ignore_index=0)(logp, torch.from_numpy(target))
    ref = lovasz_per_class_reference(logp.detach().numpy(), target, ignore=0)
    assert abs(loss.item() - ref) < 1e-10
    loss.backward()
    assert torch.isfinite(logits.grad).all() and logits.grad.abs().sum() > 0
    # perfect prediction -> zero loss
    perfect = torch.full((n, c), -50.0, dtype=torch.float64)
    perfect[torch.arange(n), torch.from_numpy(target)] = 0.0
    assert float(LovaszSoftmax(ignore_index=0)(perfect, torch.from_numpy(target))) < 1e-12


def test_scores_iou():
    s = Scores()
    gt = torch.tensor([0, 1, 1, 2, 2, 2, 3])
    pred = torch.tensor([0, 1, 2, 2, 2, 1, 3])
    probs = torch.nn.functional.one_hot(pred, 5).float()
    s.accumulate_scores(probs, gt, unlabeled_idx=0)
    ious = s.iou_per_class()
    assert 0 not in ious  # unlabeled
    assert abs(ious[1] - 1 / 3) < 1e-12 and abs(ious[2] - 2 / 4) < 1e-12 and ious[3] == 1.0
    assert abs(s.avg_class_iou() - (1 / 3 + 0.5 + 1) / 3) < 1e-12
    s.update_best()
    assert s.best_iou == s.avg_class_iou()


WORKER = textwrap.dedent("""
    import torch, sys
    from lattice_net_amd import sharding
    dist = sharding.init("gloo")
    world, rank, _ = sharding.env_world()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    sharding.broadcast_parameters(dist, net.parameters())
    x = torch.full((4, 5), float(rank + 1))
    net(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    sharding.allreduce_gradients(dist, net.parameters(), bucket_bytes=64)  # tiny buckets: several collectives
    # reference: gather every rank's local gradient and average
    for p, g in zip(net.parameters(), local):
        outs = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(outs, g)
        assert torch.allclose(p.grad, sum(outs) / world, atol=1e-6), "bucketed all-reduce != mean of local gradients"
    from lattice_net_amd.losses import Scores
    s = Scores()
    s.accumulate_scores(torch.eye(3)[[0, 1, 2]], torch.tensor([0, 1, 1 + rank % 2]), None)
    s.all_reduce(dist)
    assert int(s.intersection_per_class.sum()) == 5, s.intersection_per_class
    dist.barrier()
    print(f"rank{rank}-ok", flush=True)
""")


def test_gradient_allreduce_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket() as sock:  # a free port: parallel CI jobs or a leftover worker must not collide on a fixed one
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
           str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("-ok") == 2 and "rank0" in r.stdout and "rank1" in r.stdout


def test_nll_loss_gather_matches_torch_nll_loss():
    torch.manual_seed(0)
    logp = torch.log_softmax(torch.randn(300, 7, dtype=torch.float64, requires_grad=True), 1)
    target = torch.randint(0, 7, (300,))
    for ignore in (None, 0, 3):
        ref = torch.nn.functional.nll_loss(logp, target, ignore_index=-100 if ignore is None else ignore)
        assert torch.allclose(nll_loss_gather(logp, target, ignore), ref, atol=1e-12)


def test_generalized_soft_dice_loss_matches_the_one_hot_formula():
    from lattice_net_amd.losses import GeneralizedSoftDiceLoss
    rng = np.random.default_rng(5)
    n, c = 400, 7
    logp = torch.log_softmax(torch.from_numpy(rng.standard_normal((n, c))).double(), 1).requires_grad_(True)
    target = torch.from_numpy(rng.integers(0, c, n))
    loss = GeneralizedSoftDiceLoss(ignore_index=0)(logp, target)
    # diceloss.py:172-209 spelled out with the one-hot matrix
    p = logp.detach().exp()
    onehot = torch.zeros((n, c), dtype=torch.float64)
    onehot[torch.arange(n), target] = 1
    dice = 2 * (p * onehot).sum(0) / ((p + onehot).sum(0) + 1e-6)
    w = torch.ones(c, dtype=torch.float64)
    w[0] = 0
    expect = (w * (1 - dice)).sum() / c
    assert abs(float(loss) - float(expect)) < 1e-12
    loss.backward()
    assert torch.isfinite(logp.grad).all() and float(logp.grad.abs().sum()) > 0
    perfect = torch.log(onehot.clamp(min=1e-12))
    assert float(GeneralizedSoftDiceLoss(ignore_index=0)(perfect, target)) < 1e-5


DP_WORKER = textwrap.dedent("""
    import os, sys, numpy as np, torch
    from lattice_net_amd import sharding
    from lattice_net_amd.losses import LovaszSoftmax, nll_loss_gather
    from lattice_net_amd.synthetic import box_surface_cloud
    from tests.test_oracle_network import make_oracle_case
    out_dir = sys.argv[1]
    dist = sharding.init("gloo")
    world, rank, _ = sharding.env_world()
    # every rank starts from different parameters; rank 0's are broadcast (ln_train.py builds one model per process)
    net, lattice, _, _, _ = make_oracle_case(n=8, seed=10 + rank)
    sharding.broadcast_parameters(dist, list(net.parameters()) + list(net.buffers()))
    n, c = 500, 6
    pos = torch.from_numpy(box_surface_cloud(n, 100 + rank))             # cloud `rank` of the batch
    target = torch.from_numpy(np.random.default_rng(100 + rank).integers(0, c, n))
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-4, amsgrad=True)   # ln_train.py:165
    logp, _ = net(lattice, pos, torch.zeros((n, 1), dtype=torch.float64))
    loss = 0.5 * LovaszSoftmax(ignore_index=0)(logp, target) + 0.5 * nll_loss_gather(logp, target, ignore_index=0)   # ln_train.py:156-158
    opt.zero_grad()
    loss.backward()
    sharding.allreduce_gradients(dist, net.parameters())
    opt.step()
    torch.save({k: v.clone() for k, v in net.state_dict().items()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_data_parallel_step_equals_single_process_step_on_both_clouds(tmp_path):
    """SURVEY 8f-4 as a checked computation (ln_train.py:156-189 with the gradient exchange of sharding.allreduce_gradients): two gloo
    ranks x one cloud each, one AdamW step, against ONE process that runs both clouds and steps on the mean of the two losses — the
    whole LNN in float64 on the CPU oracle lattice; every parameter of both ranks equals the single-process result to 1e-9."""
    from lattice_net_amd.losses import LovaszSoftmax
    from lattice_net_amd.synthetic import box_surface_cloud
    from tests.test_oracle_network import make_oracle_case
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
           str(port), str(script), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout + r.stderr
    # the single process: rank 0's initial parameters, both clouds, gradient of the mean loss
    net, lattice, _, _, _ = make_oracle_case(n=8, seed=10)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-4, amsgrad=True)
    opt.zero_grad()
    n, c = 500, 6
    for k in range(2):
        pos = torch.from_numpy(box_surface_cloud(n, 100 + k))
        target = torch.from_numpy(np.random.default_rng(100 + k).integers(0, c, n))
        logp, _ = net(lattice, pos, torch.zeros((n, 1), dtype=torch.float64))
        loss = 0.5 * LovaszSoftmax(ignore_index=0)(logp, target) + 0.5 * nll_loss_gather(logp, target, ignore_index=0)
        (loss / 2).backward()  # gradients accumulate: d(mean of the two losses)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    opt.step()
    want = net.state_dict()
    got = [torch.load(tmp_path / f"rank{k}.pt") for k in range(2)]
    moved = 0
    for k, w in want.items():
        for g in got:
            assert torch.allclose(g[k], w, rtol=1e-9, atol=1e-12), k
        moved += int(not torch.equal(before[k], w))
    assert moved >= 60, "the optimizer step must have changed (nearly) every tensor"


def test_losses_and_scores_match_the_references_own_python_on_f13(golden):
    """F13 (tests/golden/make_losses_fixture.py): the reference's LovaszSoftmax (lovasz_loss.py:23), GeneralizedSoftDiceLoss
    (diceloss.py:8) and Scores (callbacks/scores.py:8-110) executed in this container on seeded inputs — plain, an ignore class,
    classes absent from the cloud, every point of one class, one point, 20 confident classes; three clouds accumulated for the
    scores.  The reference's losses only run in float32 (they build float32 one-hot tensors): this package's are evaluated in
    float32 against them (values and gradients 2e-6 of the largest entry) and in float64 against the same numbers at the
    resolution of the reference's float32 cumulative sums."""
    from lattice_net_amd.losses import GeneralizedSoftDiceLoss
    f = golden("F13_losses")
    for name in [str(x) for x in f["case_names"]]:
        logp64, labels, ignore = torch.from_numpy(f[f"{name}/logp"]), torch.from_numpy(f[f"{name}/labels"]), int(f[f"{name}/ignore"])
        for dtype, rtol, gtol in ((torch.float32, 2e-6, 2e-6), (torch.float64, 5e-6, 3e-5)):  # (the reference's own float32 cumsums)
            for red in ("mean", "sum"):
                x = logp64.to(dtype).clone().requires_grad_(True)
                loss = LovaszSoftmax(ignore_index=ignore, reduction=red)(x, labels)
                ref = float(f[f"{name}/lovasz_{red}"])
                assert abs(float(loss) - ref) <= rtol * max(abs(ref), 1.0), (name, red, dtype, float(loss), ref)
                loss.backward()
                g_ref = f[f"{name}/lovasz_{red}_grad"].astype(np.float64)
                assert np.max(np.abs(x.grad.double().numpy() - g_ref)) <= gtol * max(np.max(np.abs(g_ref)), 1e-30), (name, red, dtype)
            per_class = LovaszSoftmax(ignore_index=ignore, reduction="none")(logp64.to(dtype), labels)
            ref_pc = f[f"{name}/lovasz_none"].astype(np.float64)
            assert per_class.shape[0] == ref_pc.shape[0] and np.allclose(per_class.double().numpy(), ref_pc, rtol=5e-6, atol=5e-7), name
            x = logp64.to(dtype).clone().requires_grad_(True)
            loss = GeneralizedSoftDiceLoss(ignore_index=ignore)(x, labels)
            ref = float(f[f"{name}/dice"])
            assert abs(float(loss) - ref) <= 5e-6 * max(abs(ref), 1.0), (name, dtype, float(loss), ref)
            loss.backward()
            g_ref = f[f"{name}/dice_grad"].astype(np.float64)
            assert np.max(np.abs(x.grad.double().numpy() - g_ref)) <= 5 * gtol * max(np.max(np.abs(g_ref)), 1e-30), (name, dtype)
    s = Scores()
    unl = int(f["scores/unlabeled_idx"])
    for k in range(3):
        s.accumulate_scores(torch.from_numpy(f[f"scores/{k}/probs"]), torch.from_numpy(f[f"scores/{k}/gt"]), unl)
        avg, d = s.compute_stats()
        assert sorted(d) == f[f"scores/{k}/iou_classes"].tolist()
        assert np.allclose([d[i] for i in sorted(d)], f[f"scores/{k}/iou_values"], rtol=0, atol=1e-15)
        assert abs(avg - float(f[f"scores/{k}/avg_iou"])) < 1e-15
        s.update_best()
        assert abs(s.best_iou - float(f[f"scores/{k}/best_iou"])) < 1e-15
