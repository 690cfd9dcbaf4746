"""Multi-GPU plumbing: one process per GPU, samples sharded with no data-path collective, one all-reduce of
the metric sums at the end.

Replaces ``src/utils/init_utils.py:10-18`` (``init_dist``: env-var rendezvous, ``'nccl'`` = RCCL on ROCm,
``torch.cuda.set_device(rank % num_gpus)``) and the pickle-file gather + ``barrier`` of
``src/optimize.py:78-89``.  The padding rule follows ``data/opt_dataset.py:38-51``: the sample list is padded
with copies of sample 0 up to a multiple of ``batch x world`` and the duplicates are masked out of the metrics.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def launched_by_torchrun() -> bool:
    """True under ``python -m torch.distributed.run`` (also with ONE rank: the process group is then initialised and every
    collective really goes through the backend -- how the RCCL path is exercised on a single-GPU box)."""
    return "TORCHELASTIC_RUN_ID" in os.environ and "RANK" in os.environ


def init_dist(backend: str | None = None):
    """Returns (rank, world_size).  No-op for plain single-process runs (not launched by torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not launched_by_torchrun():
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:     # IHMR_DIST_BACKEND=gloo: ranks may share one GPU (the single-GPU check of the multi-rank code paths)
        backend = os.environ.get("IHMR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend != "nccl" and torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    if backend == "nccl":
        torch.cuda.set_device(local % torch.cuda.device_count())
    if not dist.is_initialized():
        if backend == "nccl":     # bind the communicator to this rank's GPU up front (eager init, no device guessing at first use)
            dist.init_process_group(backend, device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend)
    return rank, world


def gpu_pci_address(device_index: int):
    """``dddd:bb:dd.0`` of a visible GPU from torch's device properties (the runtime torch itself has loaded: no second copy of
    libamdhip64.so is opened), or None without a GPU."""
    if not torch.cuda.is_available():
        return None
    p = torch.cuda.get_device_properties(int(device_index))
    return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"


def numa_cpus_of_pci_device(bdf: str, sysfs_root: str = "/sys"):
    """(NUMA node, set of its CPUs) of the PCI device ``bdf`` as sysfs describes it -- ``<root>/bus/pci/devices/<bdf>/numa_node`` ->
    ``<root>/devices/system/node/node<k>/cpulist`` -- or None when either file is missing, unreadable or the node is -1 (a
    single-socket box)."""
    try:
        path = os.path.join(sysfs_root, "bus/pci/devices", str(bdf).lower(), "numa_node")
        if not os.path.isfile(path):
            return None
        with open(path) as fh:
            node = int(fh.read().strip())
        if node < 0:
            return None
        cpus = set()
        with open(os.path.join(sysfs_root, f"devices/system/node/node{node}/cpulist")) as fh:
            for part in fh.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        return node, cpus
    except (OSError, ValueError):
        return None


def pin_to_gpu_numa_node(device_index: int, sysfs_root: str = "/sys", bdf: str = None):
    """Best-effort CPU affinity for a one-process-per-GPU rank: restrict this process to the cores of the NUMA node its GPU hangs off.
    A rank's host work is graph launches and small copies (bench.py: ``host_submit_ms_per_step``); on a two-socket node eight
    unpinned ranks migrate across sockets and their submission latency (PCIe doorbells, pinned-buffer reads) goes through the
    inter-socket link.  Returns a short description, or None when the topology cannot be read (nothing is changed then).
    ``bdf`` / ``sysfs_root`` exist for the CPU test (a fake topology under tmp_path); INTEGRATION.md "Multi-GPU launch" shows the
    equivalent ``numactl`` line for launchers that prefer to pin from outside."""
    try:
        bdf = bdf or gpu_pci_address(device_index)
        found = numa_cpus_of_pci_device(bdf, sysfs_root) if bdf else None
        if found is None:
            return None
        node, cpus = found
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return f"GPU {device_index} ({bdf}) -> NUMA node {node}: {len(cpus)} cores"
    except Exception:       # (no permission, exotic torch build: run unpinned)
        return None


def shard_indices(num_samples: int, batch_size: int, rank: int, world: int):
    """Contiguous per-rank slices of the padded index list; returns (indices, is_padding) for this rank."""
    per_round = batch_size * world
    padded = ((num_samples + per_round - 1) // per_round) * per_round
    idx = np.arange(padded)
    is_pad = idx >= num_samples
    idx = np.where(is_pad, 0, idx)          # opt_dataset.py:49-51 pads with copies of sample 0
    per_rank = padded // world
    sl = slice(rank * per_rank, (rank + 1) * per_rank)
    return idx[sl], is_pad[sl]


def reduce_metrics(sums: np.ndarray, device=None) -> np.ndarray:
    """all_reduce(SUM) of the additive metric vector (float64); identity when no process group exists (a one-rank group still
    goes through the backend)."""
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(sums, dtype=np.float64)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor(np.asarray(sums, dtype=np.float64), device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def all_reduce_gradients(flat_grads: torch.Tensor) -> float:
    """Data-parallel gradient exchange of a training step: ONE all_reduce(SUM) over the flat gradient buffer of the
    model (RCCL over xGMI for CUDA tensors, gloo in the CPU tests) -- what ``DistributedDataParallel`` does for the
    reference's sub-networks (``models/mlp_model.py:383-385``), as a single bucket: the largest IHMR-MLP head is
    0.75 M parameters = 3 MB, a fraction of a millisecond on one xGMI link.  Returns the factor the optimizer has to
    apply to the summed gradient (1 / world size: DDP averages)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


def plan_gradient_buckets(spans, bucket_floats):
    """Bucketing of a flat gradient buffer whose parts become final from the END towards the start during the backward pass
    (the heads first, the stem last).  ``spans`` = the (lo, hi) ranges in the order they complete, each ending where the
    previous one starts (hi_k == lo_{k-1}); returns ``{index of the span after which to fire: (lo, hi)}``: contiguous slices of
    at least ``bucket_floats`` floats (the last one takes the remainder) that together cover the buffer exactly once."""
    fire, hi = {}, None
    for k, (lo, h) in enumerate(spans):
        if hi is None:
            hi = h
        assert h == (spans[k - 1][0] if k else h), "spans must be contiguous, descending"
        if hi - lo >= bucket_floats or k == len(spans) - 1:
            fire[k] = (lo, hi)
            hi = lo
    return fire


class OverlappedGradientReducer:
    """DistributedDataParallel's bucketed gradient all-reduce for a flat gradient buffer: ``ready(k)`` is called by the backward
    pass when span k is final and starts an asynchronous all-reduce (RCCL: on its own stream, behind the kernels already queued)
    of every bucket that has become complete, so the exchange of the deep layers' gradients runs under the backward pass of the
    shallow ones; ``finish()`` waits and returns the factor for the optimizer (1 / world size).  25 MB buckets by default: four
    for the 102 MB of ResNet-50 -- on a ring over xGMI a bucket of that size is bandwidth-bound, smaller ones pay the ring
    latency of 2 (N - 1) steps per bucket."""

    def __init__(self, flat_grads: torch.Tensor, spans, bucket_bytes: int = 25 << 20):
        self.grads, self.plan, self.handles = flat_grads, plan_gradient_buckets(spans, bucket_bytes // 4), []
        self.active = dist.is_available() and dist.is_initialized()

    def ready(self, k: int):
        if self.active and k in self.plan:
            lo, hi = self.plan[k]
            self.handles.append(dist.all_reduce(self.grads[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self) -> float:
        for h in self.handles:
            h.wait()
        self.handles = []
        return 1.0 / dist.get_world_size() if self.active else 1.0
