"""The encoder on 64 images as ONE pass of 64 against TWO passes of 32 on two streams (two instances, the same weights): batch 64 sits 2 %
past a workgroup-generation boundary on every one-tile-per-workgroup layer; do two half passes fill each other's last
generations?  usage (GPU box): python3 scripts/experiments/encoder_split_batch.py"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ihmr_amd.networks import InterHandEncoder

opt = types.SimpleNamespace(total_params_dim=122)
torch.manual_seed(0)
img = torch.rand(64, 3, 224, 224, device="cuda") * 2 - 1


def make(B):
    e = InterHandEncoder(opt, torch.zeros(B, 122)).cuda()
    return e


full = make(64)
halves = [make(32), make(32)]
for h in halves:
    h.load_state_dict(full.state_dict())
quarters = [make(16) for _ in range(4)]
for q in quarters:
    q.load_state_dict(full.state_dict())
streams = [torch.cuda.Stream() for _ in range(2)]


def run_full():
    return full(img)


def run_split(models, ns):
    outs = []
    n = 64 // len(models)
    for i, m in enumerate(models):
        with torch.cuda.stream(streams[i % ns]):
            outs.append(m(img[i * n:(i + 1) * n]))
    return outs


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


ref = run_full()
torch.cuda.synchronize()
sp = run_split(halves, 2)
torch.cuda.synchronize()
ref_t = ref[0] if isinstance(ref, (tuple, list)) else ref
sp_t = torch.cat([(o[0] if isinstance(o, (tuple, list)) else o) for o in sp])
print(f"max |split - full| over the outputs: {(sp_t - ref_t).abs().max().item():.3e} (max |full| {ref_t.abs().max().item():.3e})")
for tag, fn in (("1 x 64, one stream", run_full), ("2 x 32 on two streams", lambda: run_split(halves, 2)), ("2 x 32 on ONE stream", lambda: run_split(halves, 1)),
                ("4 x 16 on two streams", lambda: run_split(quarters, 2))):
    print(f"{tag}: {timeit(fn):.3f} ms per 64 images", flush=True)
