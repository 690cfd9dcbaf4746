"""bench.py's in-line IHMR-MLP figure (123 k) against the standalone one (135 k): does the IHMR-Baseline leg that runs before it matter?
usage (GPU box): python3 scripts/experiments/mlp_after_baseline.py [gc]"""
import gc, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

def show(tag):
    d = bench.secondary("mlp", with_cpu=False)
    print(f"{tag}: {d['value'] / 1e3:.1f} k images/s, wall {d['ms_per_step']:.3f} ms, GPU {d['gpu_ms_per_batch']:.3f} ms per batch", flush=True)

show("fresh")
b = bench.secondary("baseline", with_cpu=len(sys.argv) > 1 and "cpu" in sys.argv[1:])
print(f"baseline leg: {b['value'] / 1e3:.2f} k images/s", flush=True)
show("after the IHMR-Baseline leg")
gc.collect(); torch.cuda.empty_cache()
show("after gc.collect() + empty_cache()")
