#!/usr/bin/env python3
"""After ONE forward (prep + dist + non-fused sampler) at fuse x 64 samples: read qcell, the bitmaps, the boxes and the vertices back and
recompute every cell word on the host.  usage: IHMR_HIP_LIBRARY=build/qcheck.so python3 scripts/debug_qcell.py [fuse]"""
import ctypes as C, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ihmr_amd import hip, two_hand
from ihmr_amd.optimize_model import OptimizeModel
from ihmr_amd.synthetic import synthetic_opt_batch
G = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = 64
o = types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, num_joints=42, total_params_dim=122,
                          cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="", strategy="opt_default",
                          save_mid_freq=10, optimizer="adam", opt_epoch=3, fuse_batches=G, no_fused_tail=True)
m = OptimizeModel(o); m.use_graphs = False
fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
bs = [synthetic_opt_batch(B, fwd, seed=1234 + 1000 * i) for i in range(G)]
inp = {k: torch.cat([b[k] for b in bs]).cuda() for k in bs[0]}
m.set_input(inp); m.init_optimize()
for rep in range(2):
    if os.environ.get("STAGE"):
        m.run_stage(m.strategy[int(os.environ["STAGE"])]) if rep == 0 else m.forward_losses(m.default_loss_weights)
    else:
        m.forward_losses(m.default_loss_weights)
    torch.cuda.synchronize()
    BB = m.batch_size
    L = hip.lib()
    ptr = (C.c_void_p * 4)()
    L.ihmr_debug_sdf_ptrs.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.ihmr_debug_sdf_ptrs(C.byref(m.io), BB, ptr)
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    def dl(p, nbytes):
        buf = np.zeros(nbytes, np.uint8)
        assert rt.hipMemcpy(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(p), ctypes.c_size_t(nbytes), 2) == 0
        return buf
    qc = dl(ptr[0], 2 * BB * 778 * 4).view(np.uint32).reshape(BB, 2, 778)
    ib = dl(ptr[1], 2 * BB * 1024 * 4).view(np.uint32).reshape(2, BB, 1024)
    box = dl(ptr[2], 2 * BB * 16).view(np.float32).reshape(2, BB, 4)
    verts = m.buf["verts"].cpu().numpy()          # (2, BB, 778, 3)
    bad_cell = bad_mask = tot = 0
    for hnd in range(2):
        c, s = box[hnd][:, None, :3], box[hnd][:, None, 3:4]
        q = (verts[1 - hnd] - c) / s
        ix = ((q + 1.0) * 32 - 1.0) / 2.0
        f = np.floor(ix)
        ing = np.all((f >= -1) & (f <= 31), axis=-1)
        fi = f.astype(np.int64)
        want = np.where(ing, 0x80000000 | (fi[..., 0] + 1) | ((fi[..., 1] + 1) << 6) | ((fi[..., 2] + 1) << 12), 0).astype(np.uint32)
        got = qc[:, hnd, :]
        bad_cell += int(((got & 0x8003ffff) != want).sum()); tot += want.size
        # mask from the bitmap
        mask = np.zeros_like(want)
        i0, j0, k0 = fi[..., 0], fi[..., 1], fi[..., 2]
        for c4 in range(4):
            j, k = j0 + (c4 & 1), k0 + (c4 >> 1)
            ok = ing & (j >= 0) & (j < 32) & (k >= 0) & (k < 32)
            w = ib[hnd][np.arange(BB)[:, None], np.clip(k, 0, 31) * 32 + np.clip(j, 0, 31)]
            b0 = np.where(ok & (i0 >= 0), (w >> (i0 & 31).astype(np.uint32)) & 1, 0)
            b1 = np.where(ok & (i0 + 1 < 32), (w >> ((i0 + 1) & 31).astype(np.uint32)) & 1, 0)
            mask |= (b0 << (2 * c4)).astype(np.uint32) | (b1 << (2 * c4 + 1)).astype(np.uint32)
        bad_mask += int((((got >> 18) & 0xff) != mask).sum())
        if rep == 1 and hnd == 0:
            w_ = np.argwhere((got & 0x8003ffff) != want)[:5]
            for b_, v_ in w_:
                print("  bad cell at sample", b_, "vertex", v_, hex(got[b_, v_]), "want", hex(want[b_, v_]))
    print(f"forward {rep}: entries {tot}, wrong cells {bad_cell} (float rounding at cell borders possible), wrong masks {bad_mask}, "
          f"high bits set {(qc >> 26 & 31 != 0).sum()}, finite verts {np.isfinite(verts).all()}")
