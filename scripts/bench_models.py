#!/usr/bin/env python3
"""Secondary workloads (BASELINE.json configs[1], configs[2]): IHMR-Baseline B=64 and IHMR-MLP B=128 inference.
Not the driver's bench line (that is bench.py = IHMR-OPT); prints one JSON line per workload."""
import json, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch

def opt(B, **kw):
    d = dict(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, input_nc=3, num_joints=42, total_params_dim=122,
             cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3, model_root="", mean_param_file="mean_mano_params.pkl",
             checkpoints_dir="./checkpoints", strategy="mlp_default")
    d.update(kw); return types.SimpleNamespace(**d)

def timeit(fn, steps, warmup):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

def main():
    from ihmr_amd import two_hand
    from ihmr_amd.baseline_model import InterHandModel
    from ihmr_amd.mlp_model import MLPModel
    from ihmr_amd.strategies import make_mlp_strategy
    from ihmr_amd.synthetic import synthetic_opt_batch
    which = sys.argv[1:] or ["baseline", "mlp", "train"]
    if "baseline" in which:
        B = 64
        m = InterHandModel(opt(B)); m.eval()
        fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
        batch = {k: v.cuda() for k, v in synthetic_opt_batch(B, fwd, seed=1234, with_image=True).items()}
        def step():
            m.set_input(batch); m.test(); return m.get_pred_result()
        dt_block = timeit(step, 10, 3)
        pend = []
        def step_async():        # the export of batch i is collected while batch i + 1 runs (get_pred_result_async)
            m.set_input(batch); m.test(); pend.append(m.get_pred_result_async())
            if len(pend) > 1: pend.pop(0).wait()
        dt = timeit(step_async, 20, 3)
        # encoder alone
        enc_dt = timeit(lambda: m.encoder(batch["img"]), 10, 3)
        print(json.dumps(dict(workload="IHMR-Baseline (ResNet-50 + MANO regress) batch=64 inference", images_per_s=B / dt, ms_per_batch=dt * 1e3, blocking_export_ms_per_batch=dt_block * 1e3,
                              encoder_ms_per_batch=enc_dt * 1e3, encoder_tflops=8.2e9 * B / enc_dt / 1e12, encoder_frac_of_fp32_mfma_peak=8.2e9 * B / enc_dt / 157.3e12)))
    if "mlp" in which:
        from helpers import seeded_state_dict
        B = 128
        strat = make_mlp_strategy()
        m = MLPModel(opt(B)); m.set_update_info(strat, B)
        for i in range(len(strat)):
            m.add_new_network(i); net = m.sub_network_list[i]; net.load_state_dict(seeded_state_dict(net, 900 + i, last_scale=0.02))
        m.eval()
        fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
        b = synthetic_opt_batch(B, fwd, seed=1234, with_feat=True)
        b["init_hand_trans"] = b["init_hand_trans"][:, 0, :3].contiguous(); b["img"] = torch.zeros(B, 3, 8, 8)
        batch = {k: v.cuda() for k, v in b.items()}
        def step():
            m.set_input(batch); m.test(); return m.get_pred_result()
        dt = timeit(step, 10, 3)
        pend = []
        def step_async():        # the export of batch i is collected while batch i + 1 runs (get_pred_result_async)
            m.set_input(batch); m.test(); pend.append(m.get_pred_result_async())
            if len(pend) > 1: pend.pop(0).wait()
        dta = timeit(step_async, 20, 3)
        print(json.dumps(dict(workload="IHMR-MLP refinement head batch=128 inference (6 stages: 8 MANO+SDF evaluations, 6 MLPs)", images_per_s=B / dta,
                              ms_per_batch=dta * 1e3, blocking_export_ms_per_batch=dt * 1e3, blocking_export_images_per_s=B / dt)))

    if "train" in which:          # the two training steps: python -m ihmr_amd.run_train_mlp / run_train_baseline
        from ihmr_amd import run_train_baseline, run_train_mlp
        log = run_train_mlp.main(["--num_samples", "512", "--batchSize", "128", "--epochs", "10", "--stages", "6"])
        ms = float(np.mean([r["ms_per_step"] for r in log[1:]]))
        print(json.dumps(dict(workload="IHMR-MLP training step batch=128 (mean over stages 1-5)", ms_per_step=ms, samples_per_s=128 / ms * 1e3)))
        log = run_train_baseline.main(["--num_samples", "256", "--batchSize", "64", "--total_epoch", "3"])
        print(json.dumps(dict(workload="IHMR-Baseline training step batch=64 (ResNet-50 train mode + MANO + losses + Adam)",
                              ms_per_step=log[-1]["ms_per_step"], images_per_s=log[-1]["images_per_s"])))


if __name__ == "__main__":
    main()
