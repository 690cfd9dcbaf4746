#!/usr/bin/env python3
"""Phase times of sdf_dist_kernel's work items from an experiment build (-DSDF_STAMPS=1 list search / =2 full search):
   hipcc ... -DSDF_STAMPS=1 ihmr_hip.hip -o build/stamps1.so ;  IHMR_HIP_LIBRARY=build/stamps1.so python3 scripts/sdf_stamps.py [fuse]
Runs one stage of the refinement at `fuse` x 64 samples per launch with the counters on and prints the mean shader-clock cycles per
item and phase (summed per wave by the kernel into the spare counter slots)."""
import ctypes as C, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ihmr_amd import hip, two_hand
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = 64
o = types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122,
                          cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="", strategy="opt_default",
                          save_mid_freq=10, optimizer="adam", opt_epoch=49, fuse_batches=G)
m = OptimizeModel(o)
fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
bs = [synthetic_opt_batch(B, fwd, seed=1234 + 1000 * i) for i in range(G)]
inp = {k: torch.cat([b[k] for b in bs]).cuda() for k in bs[0]}
m.set_input(inp); m.init_optimize(); m.optimize()          # warm: lists built, graphs captured
torch.cuda.synchronize()
import numpy as np
L = hip.lib()
L.ihmr_debug_stamps.argtypes = [C.c_void_p, C.c_int]
names = ["items", "front (loads + staging)", "(list: bound; full: sphere passes)", "(list: walk)", "refine", "exact", "wave lifetime"]
for stage_id in (1, 3):
    m.set_input(inp); m.init_optimize()
    for s in m.strategy[:stage_id]:
        m.run_stage(s)
    L.ihmr_debug_stamps(None, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter() if "time" in dir() else None
    import time
    t0 = time.perf_counter()
    m.run_stage(m.strategy[stage_id]); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    raw = np.zeros(4096 * 4 * 12 + 4096 * 8, np.int64)
    L.ihmr_debug_stamps(raw.ctypes.data, 0)
    buf, span = raw[:4096 * 4 * 8].reshape(-1, 8), raw[4096 * 4 * 8:4096 * 4 * 12].reshape(-1, 4)
    prep = raw[4096 * 4 * 12:].reshape(-1, 8)
    prep = prep[prep[:, 7] > 0]
    if len(prep):
        ph = prep[:, :6] / prep[:, 7:8] / 2400.0
        tot = ph.sum(1)
        print('   prep per hand [us]: ' + '; '.join(f'{n} {ph[:, k].mean():.2f} (max {ph[:, k].max():.2f})' for k, n in enumerate(['loads+box', 'normalise+needed', 'list state', 'records', 'ray parity', 'publish'])) + f'; total mean {tot.mean():.2f} p90 {np.percentile(tot, 90):.2f} max {tot.max():.2f}')
    sp = span[span[:, 1] > 0]
    # the shader clocks of the XCDs are not aligned with each other: times relative to the first wave start of the same XCD
    xcd = (sp[:, 2] % 8).astype(int)
    t0x = sp[:, 0].min()
    st, en = (sp[:, 0] - t0x) / 100.0, (sp[:, 1] - t0x) / 100.0
    print(f"   last launch: {len(sp)} waves; wave start p10/p50/p90/max {np.percentile(st, 10):.1f}/{np.percentile(st, 50):.1f}/"
          f"{np.percentile(st, 90):.1f}/{st.max():.1f} us; end max per XCD {[round(float(en[xcd == x].max()), 1) for x in range(8)]} us; "
          f"lifetime mean {np.mean(en - st):.2f} p90 {np.percentile(en - st, 90):.2f} max {(en - st).max():.2f} us")
    print("      live waves (chip) at " + "; ".join(f"{t} us: {int(((st <= t) & (en > t)).sum())}" for t in (2, 5, 10, 20, 30, 40, 50, 60)))
    print(f"      work units per workgroup: mean {sp[:, 3].mean():.2f} max {sp[:, 3].max()}")
    live = buf[buf[:, 7] > 0]
    nl = 50
    busy = live[live[:, 0] > 0]
    print(f"stage {stage_id}: wall {1e3 * wall:.2f} ms for 50 iterations; waves of this search per launch {len(live)}, with work {len(busy)}; "
          f"wave-items per launch {busy[:, 0].sum() / nl:.0f}")
    it = busy[:, 0].sum()
    print("   mean cycles per wave-item: " + "; ".join(f"{n}: {busy[:, 1 + k].sum() / it:.0f}" for k, n in enumerate(names[1:6])) +
          f"; sum {busy[:, 1:6].sum() / it:.0f} (= {busy[:, 1:6].sum() / it / 2400:.2f} us)")
    print(f"   mean wave lifetime {live[:, 6].sum() / live[:, 7].sum():.0f} cycles (busy waves {busy[:, 6].sum() / busy[:, 7].sum():.0f}); "
          f"items per busy wave and launch {it / busy[:, 7].sum():.2f}")
