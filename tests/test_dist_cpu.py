"""world_size-2 gloo tests of the sharding + metric reduction (CPU, no GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_pred(idx, rng):
    n = len(idx)
    gt = rng.normal(0, 0.05, (n, 42, 3)).astype(np.float32)
    pred = gt + rng.normal(0, 0.005, (n, 42, 3)).astype(np.float32)
    return dict(pred_cam_params=np.zeros((n, 3), np.float32), pred_shape_params=np.zeros((n, 20), np.float32),
                pred_pose_params=np.zeros((n, 96), np.float32), pred_hand_trans=np.zeros((n, 1, 3), np.float32),
                pred_joints_3d=pred, gt_joints_3d=np.concatenate([gt, np.ones((n, 42, 1), np.float32)], 2),
                collision_loss_origin_scale=np.abs(rng.normal(0, 1e-3, (n, 1556))).astype(np.float32))


def _all_preds(num):
    rng = np.random.RandomState(0)
    return _fake_pred(np.arange(num), rng)


def _worker(rank, world, port, num, bs, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ihmr_amd import dist as D
    from ihmr_amd.evaluator import Evaluator
    r, w = D.init_dist("gloo")
    assert (r, w) == (rank, world)
    idx, pad = D.shard_indices(num, bs, r, w)
    full = _all_preds(num)
    ev = Evaluator()
    for s in range(0, len(idx), bs):
        sel = idx[s:s + bs]
        keep = ~pad[s:s + bs]
        pr = {k: v[sel] for k, v in full.items()}
        ev.update(sel, pr)
        new = ev.pred_results[-len(sel):]
        ev.pred_results = ev.pred_results[:-len(sel)] + [p for p, k in zip(new, keep) if k]   # mask padding duplicates
    total = D.reduce_metrics(ev.metric_sums())
    if rank == 0:
        np.save(out, total)
    torch.distributed.destroy_process_group()


def test_two_rank_metric_reduction_matches_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    from ihmr_amd.evaluator import Evaluator
    num, bs, world = 13, 4, 2          # 13 samples pad to 16 = 2 ranks x 2 batches of 4
    out = str(tmp_path / "sums.npy")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num, bs, out), nprocs=world, join=True)
    got = np.load(out)
    ev = Evaluator()
    ev.update(np.arange(num), _all_preds(num))
    ref = ev.metric_sums()
    assert np.allclose(got, ref, rtol=1e-10, atol=1e-12), (got, ref)   # float64 sums, different association
    m = Evaluator.metrics_from_sums(got)
    assert m["mpjpe_3d"] > 0 and m["collision_max"] >= m["collision_ave"]


def _grad_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ihmr_amd import dist as D
    D.init_dist("gloo")
    g = torch.from_numpy(np.random.RandomState(100 + rank).normal(size=4099).astype(np.float32))   # this rank's flat gradient
    scale = D.all_reduce_gradients(g)
    if rank == 0:
        np.save(out, (g * scale).numpy())
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_all_reduce_is_the_mean(tmp_path):
    """the training step's data-parallel exchange (ihmr_amd.dist.all_reduce_gradients): sum over ranks x 1/world = mean"""
    out = str(tmp_path / "grad.npy")
    mp.spawn(_grad_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ref = sum(np.random.RandomState(100 + r).normal(size=4099).astype(np.float32) for r in range(2)) * np.float32(0.5)
    assert np.allclose(np.load(out), ref, rtol=0, atol=1e-7)
    from ihmr_amd.dist import all_reduce_gradients
    assert all_reduce_gradients(torch.ones(3)) == 1.0                     # single process: identity


def _bucket_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ihmr_amd import dist as D
    D.init_dist("gloo")
    n = 1000
    spans = [(900, 1000), (700, 900), (650, 700), (300, 650), (0, 300)]          # completion order: end of the buffer first
    g = torch.zeros(n)
    red = D.OverlappedGradientReducer(g, spans, bucket_bytes=4 * 250)
    src = torch.from_numpy(np.random.RandomState(200 + rank).normal(size=n).astype(np.float32))
    for k, (lo, hi) in enumerate(spans):                                          # the "backward pass": fill a span, report it
        g[lo:hi] = src[lo:hi]
        red.ready(k)
    scale = red.finish()
    if rank == 0:
        np.save(out, (g * scale).numpy())
    torch.distributed.destroy_process_group()


def test_overlapped_bucketed_gradient_reduction(tmp_path):
    """the encoder trainer's data-parallel exchange: buckets fired while the "backward pass" is still filling the buffer"""
    from ihmr_amd.dist import plan_gradient_buckets
    spans = [(900, 1000), (700, 900), (650, 700), (300, 650), (0, 300)]
    plan = plan_gradient_buckets(spans, 250)
    assert plan == {1: (700, 1000), 3: (300, 700), 4: (0, 300)}                  # >= 250 floats each, the remainder last
    assert plan_gradient_buckets(spans, 10 ** 9) == {4: (0, 1000)}               # one bucket when the buffer is small
    out = str(tmp_path / "bucket.npy")
    mp.spawn(_bucket_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ref = sum(np.random.RandomState(200 + r).normal(size=1000).astype(np.float32) for r in range(2)) * np.float32(0.5)
    assert np.allclose(np.load(out), ref, rtol=0, atol=1e-7)


def test_shard_indices_cover_and_pad():
    sys.path.insert(0, ROOT)
    from ihmr_amd.dist import shard_indices
    seen, pads = [], 0
    for r in range(4):
        idx, pad = shard_indices(70, 8, r, 4)
        assert len(idx) == 24                      # 70 -> 96 = 4 ranks x 3 batches of 8
        seen += idx[~pad].tolist()
        pads += int(pad.sum())
        assert np.all(idx[pad] == 0)
    assert sorted(seen) == list(range(70)) and pads == 26


def test_evaluator_matches_reference_metrics():
    """the product's host evaluator against the golden vectors captured from the reference's metric_utils"""
    sys.path.insert(0, ROOT)
    from ihmr_amd import evaluator as E
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "metrics.npz")))
    for i in range(4):
        a = E.get_single_joints_error(g[f"pred_{i}"], g[f"gt_{i}"], g[f"valid_{i}"], float(g[f"scale_{i}"]))
        assert np.allclose(a, g[f"j3d_err_{i}"], atol=1e-9)
        p = E.get_single_pa_inter_joints_error(g[f"pred_{i}"], g[f"gt_{i}"], g[f"valid_{i}"], float(g[f"scale_{i}"]))
        assert np.allclose(p, g[f"pa_err_{i}"], atol=1e-7)


def _single_worker(rank, world, port, out):
    import ihmr_amd.dist as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", TORCHELASTIC_RUN_ID="t")
    r, w = D.init_dist("gloo")
    assert (r, w) == (0, 1) and torch.distributed.is_initialized()
    sums = D.reduce_metrics(np.arange(9, dtype=np.float64))
    g = torch.arange(8, dtype=torch.float32)
    f = D.all_reduce_gradients(g)
    np.save(out, np.concatenate([sums, [f], g.numpy()]))
    torch.distributed.destroy_process_group()


def test_one_rank_group_goes_through_the_backend(tmp_path):
    """A ONE-rank launch under torch.distributed.run initialises a real process group and the reductions run through it
    (what tests/test_gpu_multirank.py does with RCCL on the single GPU of a test box)."""
    out = str(tmp_path / "one.npy")
    mp.spawn(_single_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    got = np.load(out)
    assert np.array_equal(got[:9], np.arange(9)) and got[9] == 1.0 and np.array_equal(got[10:], np.arange(8))


def _fake_sysfs(root, bdf, node, cpulist):
    d = root / "bus/pci/devices" / bdf
    d.mkdir(parents=True)
    (d / "numa_node").write_text(f"{node}\n")
    if node >= 0:
        n = root / f"devices/system/node/node{node}"
        n.mkdir(parents=True)
        (n / "cpulist").write_text(cpulist + "\n")


def test_numa_pinning_reads_the_topology_and_leaves_no_trace(tmp_path):
    """dist.pin_to_gpu_numa_node (INTEGRATION.md "Multi-GPU launch") against a FAKE sysfs tree and a given bus id: no GPU runtime is
    touched, and the process's affinity is restored afterwards (on a two-socket GPU box the real call narrows it for good)."""
    import os
    from ihmr_amd import dist as D
    before = os.sched_getaffinity(0)
    try:
        cpus = sorted(before)
        # a node that holds two of this process's CPUs and others it does not have (ranges and singles; upper-case bus id)
        want = {cpus[0], cpus[-1]}
        _fake_sysfs(tmp_path / "a", "0000:c1:00.0", 1, f"{cpus[0]},{cpus[-1]},100000-100003")
        assert D.numa_cpus_of_pci_device("0000:C1:00.0", str(tmp_path / "a")) == (1, want | set(range(100000, 100004)))
        msg = D.pin_to_gpu_numa_node(3, sysfs_root=str(tmp_path / "a"), bdf="0000:C1:00.0")
        assert msg is not None and "NUMA node 1" in msg and f"{len(want)} cores" in msg
        assert os.sched_getaffinity(0) == want
        os.sched_setaffinity(0, before)
        # nothing is changed: node -1 (single socket), a missing device, a node without any of our CPUs, a garbled file
        _fake_sysfs(tmp_path / "b", "0000:05:00.0", -1, "")
        _fake_sysfs(tmp_path / "c", "0000:05:00.0", 0, "100000-100007")
        _fake_sysfs(tmp_path / "d", "0000:05:00.0", 0, "x-y")
        for root, bdf in ((tmp_path / "b", "0000:05:00.0"), (tmp_path / "b", "0000:06:00.0"), (tmp_path / "c", "0000:05:00.0"),
                          (tmp_path / "d", "0000:05:00.0"), (tmp_path / "nowhere", "0000:05:00.0")):
            assert D.pin_to_gpu_numa_node(0, sysfs_root=str(root), bdf=bdf) is None
            assert os.sched_getaffinity(0) == before
    finally:
        os.sched_setaffinity(0, before)


def test_bench_refuses_a_launcher_whose_world_size_differs_from_gpus():
    """`--gpus N` is a guarantee: under a launcher with another WORLD_SIZE bench.py exits non-zero (before anything touches a GPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and not r.stdout.strip()
