"""The N > 1 code paths on ONE GPU: two ranks share the device (IHMR_DIST_BACKEND=gloo, CUDA tensors through gloo), launched
exactly as the driver launches them (``python -m torch.distributed.run --nproc-per-node 2 ...``).  RCCL itself needs a
multi-GPU node; what is covered here is everything around it: rank / seed handling, barriers, the MAX / SUM reductions, rank-0
output, the gradient exchange of the training loops."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600, nproc=2, backend="gloo"):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, IHMR_DIST_BACKEND=backend, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_with_two_ranks_prints_one_line():
    lines = _run(["bench.py", "--gpus", "2", "--steps", "8", "--warmup", "4"])
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["steps"] == 8 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["cpu_baseline"] is None and d["roofline"]["frac"] > 0


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python3 bench.py --gpus 2 ...` as the driver's 1-GPU command is shaped, with NO launcher around it: bench.py itself must start
    two ranks (a child `torch.distributed.run`, before the parent touches the GPU) and pass their one line through -- n_gpus 2, the
    whole job's batch.  On this one-GPU box the two ranks share the device over gloo and the line says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "IHMR_DIST_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "8", "--warmup", "4"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["steps"] == 8 and d["warmup"] == 4 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    import torch
    if torch.cuda.device_count() < 2:
        assert d["config"]["ranks_per_device"] == 2 and d["config"]["dist_backend"] == "gloo"
    else:
        assert d["config"]["ranks_per_device"] == 1 and d["config"]["dist_backend"] == "nccl"


def test_bench_line_keeps_the_latency_figures_in_config():
    """The metric's second half (ms per refinement iteration at batch 64, one batch in flight) and the per-stage figures sit in
    `config`, which the driver's record keeps."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-work-counters"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    c = d["config"]
    assert c["latency_ms_per_refine_iter"] == d["latency"]["ms_per_refine_iter"] and 0.0 < c["latency_ms_per_refine_iter"] < 0.2
    assert c["latency_images_per_s"] == d["latency"]["images_per_s"] > 1000.0
    assert len(c["latency_stage_us_per_refine_iter"]) == 4 and all(10.0 < x < 300.0 for x in c["latency_stage_us_per_refine_iter"])
    assert c["distinct_batches"] == max(g for q in c["launch_sequences_timed"] for g in q) == 2 and d["parity"]["translated_reuse"] == "on"
    print(f"[bench] latency {c['latency_ms_per_refine_iter']:.4f} ms per iteration, stages {c['latency_stage_us_per_refine_iter']}")


def test_run_optimize_two_ranks_equals_one_process():
    from ihmr_amd import run_optimize
    two = _run(["-m", "ihmr_amd.run_optimize", "--num_samples", "128", "--batchSize", "32", "--opt_epoch", "9"])[-1]
    one = run_optimize.main(["--num_samples", "128", "--batchSize", "32", "--opt_epoch", "9"])
    for k in ("mpjpe_3d", "inter_mpjpe_3d", "collision_ave", "collision_max"):
        assert abs(two[k] - one[k]) <= 1e-12 * max(1.0, abs(one[k])), (k, two[k], one[k])


def test_bench_with_eight_ranks_runs_config_5_code_path():
    """BASELINE.json configs[4] (bash/optimize.sh:11,22-23: 512 samples over the 8 processes of one node) without an 8-GPU node:
    EIGHT ranks launched as the driver launches them, sharing the one GPU over gloo -- per-rank seeds, the barrier + MAX
    reduction of the step time over 8 ranks, rank-0-only output with the whole job's batch."""
    # (the one TIMING bound below holds on a warm box: the first eight-rank run on a fresh box -- eight processes paging the image in and
    # capturing their graphs at once -- has read 7.8 ms where every later run reads 0.3 - 0.6; the run is repeated once before the bound counts.
    # Everything else is asserted on every run.)
    for attempt in range(2):
        lines = _run(["bench.py", "--gpus", "8", "--steps", "8", "--warmup", "3"], nproc=8, timeout=1500)
        assert len(lines) == 1
        d = lines[0]
        assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 512 and d["steps"] == 8 and d["warmup"] == 3 and d["scaling"] == "weak"
        assert d["config"]["parallelism"].startswith("dp8") and d["value"] > 0 and d["cpu_baseline"] is None
        assert abs(d["value"] - 512 * 8 / (d["ms_per_step"] * 8 / 1000.0)) < 1e-6 * d["value"]     # whole-job images / max-over-ranks time
        print(f"[bench] 8 ranks on one GPU, run {attempt}: host submit {d['host_submit_ms_per_step']:.3f} ms per step of {d['ms_per_step']:.3f} ms")
        if 0.0 < d["host_submit_ms_per_step"] < 2.7:
            break
    # host submission cost, max over the 8 ranks (round 5): what a rank's launch thread spends per step must stay below the GPU time
    # of a step on a GPU of its own (2.7 ms, BENCH_r04) -- here with the eight ranks contending for one GPU's queues AND one host
    assert 0.0 < d["host_submit_ms_per_step"] < 2.7, d["host_submit_ms_per_step"]


def test_host_submission_cost_is_a_fraction_of_a_step():
    """One rank: the launch thread's work per step (graph launches, input staging, export requests; bench.py:
    host_submit_ms_per_step) against the step's GPU time."""
    for attempt in range(2):       # (a timing bound: repeated once, see the eight-rank test)
        r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras", "--no-work-counters"],
                           cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        print(f"[bench] host submit {d['host_submit_ms_per_step']:.3f} ms per step of {d['ms_per_step']:.3f} ms")
        if 0.0 < d["host_submit_ms_per_step"] < 0.5 * d["ms_per_step"]:
            break
    assert 0.0 < d["host_submit_ms_per_step"] < 0.5 * d["ms_per_step"]


def test_run_optimize_eight_ranks_with_padding_equals_one_process():
    """500 samples at batch 64 on 8 ranks: the list is padded to 512 with 12 copies of sample 0 (opt_dataset.py:38-51), every rank
    refines one batch of 64, the padding is masked out of the metric sums (evaluator.py:137-146) and ONE float64 all-reduce
    combines them -- equal to the single-process run over the same 500 samples to 1e-12."""
    from ihmr_amd import dist as D
    from ihmr_amd import run_optimize
    pads = [int(D.shard_indices(500, 64, r, 8)[1].sum()) for r in range(8)]
    assert pads == [0] * 7 + [12] and all(len(D.shard_indices(500, 64, r, 8)[0]) == 64 for r in range(8))
    args = ["--num_samples", "500", "--batchSize", "64", "--opt_epoch", "9"]
    eight = _run(["-m", "ihmr_amd.run_optimize"] + args, nproc=8, timeout=1500)[-1]
    one = run_optimize.main(args)
    assert eight["world"] == 8 and eight["num_samples"] == 500
    for k in ("mpjpe_3d", "inter_mpjpe_3d", "collision_ave", "collision_max"):
        assert abs(eight[k] - one[k]) <= 1e-12 * max(1.0, abs(one[k])), (k, eight[k], one[k])


def test_training_loops_with_two_ranks():
    log = _run(["-m", "ihmr_amd.run_train_mlp", "--num_samples", "64", "--batchSize", "32", "--epochs", "4", "--stages", "1"])
    assert log[-1]["steps"] == 8 and log[-1]["loss_last"] < log[-1]["loss_first"]
    log = _run(["-m", "ihmr_amd.run_train_baseline", "--num_samples", "8", "--batchSize", "8", "--total_epoch", "6", "--lr", "1e-4"])
    assert len(log) == 6 and log[-1]["loss_last"] < log[0]["loss_first"]


def test_rccl_backend_runs_the_collectives_with_one_rank():
    """RCCL itself (backend "nccl") on the one GPU a test box has: ONE rank launched exactly as the driver launches N
    (``torch.distributed.run``), so the process group is real -- ``init_process_group("nccl", device_id=...)``, the barrier, the
    MAX all-reduce of the step time in bench.py and the float64 metric all-reduce of ``run_optimize`` on a device tensor all
    go through RCCL (two ranks cannot share a GPU in one RCCL communicator; a multi-GPU node only adds peers)."""
    from ihmr_amd import run_optimize
    log = _run(["-m", "ihmr_amd.run_optimize", "--num_samples", "64", "--batchSize", "32", "--opt_epoch", "4"], nproc=1, backend="nccl")[-1]
    one = run_optimize.main(["--num_samples", "64", "--batchSize", "32", "--opt_epoch", "4"])
    assert log["world"] == 1
    for k in ("mpjpe_3d", "inter_mpjpe_3d", "collision_ave", "collision_max"):
        assert abs(log[k] - one[k]) <= 1e-12 * max(1.0, abs(one[k])), (k, log[k], one[k])
    lines = _run(["bench.py", "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-extras", "--rccl-selftest"],
                 nproc=1, backend="nccl")
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["value"] > 0
    st = lines[0]["rccl_selftest"]
    assert st["backend"] == "nccl" and st["all_reduce_sum_f64_ok"] and st["all_reduce_max_ok"] and st["all_gather_ok"] and st["gradient_bucket_ok"]
