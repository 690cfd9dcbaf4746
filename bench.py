#!/usr/bin/env python3
"""Benchmark of the IHMR hot path on MI355X: IHMR-OPT, 200 refinement iterations, batch 64 per GPU
(BASELINE.json metric / configs[3]).

A "step" is one pass of the hot path over one synthetic batch, exactly the body of the reference's
loop ``src/optimize.py:61-71``: ``set_input -> init_optimize -> optimize -> get_pred_result`` with the
strategy ``opt_default`` at ``epoch=49`` (4 stages x 50 = 200 forward+backward+Adam iterations + the
final forward) and ``save_mid_freq=10`` (``bash/optimize.sh:33``).  Inputs are resident in HBM when the
timed region starts.  One process per GPU; for N>1 the driver launches this file under
``torch.distributed.run`` and the ranks shard the global batch (independent samples, no data-path
collective); timing = max over ranks of K steps bracketed by barrier + synchronize.

Prints ONE JSON line (rank 0) with the driver contract plus ``roofline`` (dominant kernel, HIP-event
timed on the launch stream inside this run) and ``cpu_baseline`` (the CPU oracle, a port of the
reference's PyTorch op graph, timed on a bounded sample on this host's cores).
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np
import torch

FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: FP32 vector == FP32 (f32-input) MFMA dense peak
# Shader cycles one wave64 vector instruction occupies its SIMD's issue port, calibrated (round 5; scripts/microbench_issue,
# profiles/r5_issue_rates.txt, four waves per SIMD): v_fma / v_mul / v_add_f32 and v_add_u32 2.4-2.5, shifts / 3-operand logic / packed
# fp32 4.2, compare + select 3.1, rcp / sqrt 8.2; weighted with the static instruction mix of the three large kernels
# (scripts/isa_mix.py: 2.84 / 3.17 / 3.1-3.2) = 3.0.  Round 4 priced every vector instruction at 4 (a 16-lane SIMD): its issue_slots
# fractions were a third too high.  A LONE wave issues one vector instruction per ~5 cycles: a SIMD needs two or more waves that are
# not waiting to reach this rate.
VALU_ISSUE_CYCLES = 3.0
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec; ~6.3 TB/s achievable)


def make_opt(B, epoch, freq, rank, asset="mitten"):
    """`asset`: "mitten" = the default synthetic MANO-shaped model (every test, every headline number); "fingers" = the same model on
    a mesh with a palm and five finger tubes (ihmr_amd/assets.py: the geometry-sensitivity asset)."""
    return types.SimpleNamespace(isTrain=False, dist=False, process_rank=rank, batchSize=B, inputSize=224, num_joints=42,
                                 total_params_dim=122, cam_params_dim=3, pose_params_dim=96, shape_params_dim=20,
                                 trans_params_dim=3, model_root="" if asset == "mitten" else f"synthetic:{asset}", strategy="opt_default",
                                 save_mid_freq=freq, optimizer="adam", opt_epoch=epoch)


def geometry_report(asset, B, epoch, freq, steps=24, fuse=6):
    """Throughput and collision-work statistics of the refinement on one synthetic asset, measured the same way for every asset
    (two launch sequences of `fuse` batches on two streams, `steps` batches in all; NOT the headline's schedule): images/s, inside
    voxels per sample and iteration, share of inside voxels searched in full, voxels refused a candidate list (list overflow: more
    than 192 triangles within the list bound) per sample and iteration.  "fingers" uses the interlocked batch generator."""
    from ihmr_amd import two_hand
    from ihmr_amd.optimize_model import OptimizeModel
    from ihmr_amd.synthetic import synthetic_opt_batch
    inter = asset != "mitten"
    def model(g):
        o = make_opt(B, epoch, freq, -1, asset)
        o.fuse_batches = g
        return OptimizeModel(o)
    one = model(1)
    fwd = lambda p, s_, t: two_hand.forward_from_packed(one.mano_models["right"], p.cuda(), s_.cuda(), t.cuda())[2]
    bs = [synthetic_opt_batch(B, fwd, seed=1234 + 1000 * i, interlock=inter) for i in range(fuse)]
    inp = {k: torch.cat([b[k] for b in bs]).cuda() for k in bs[0]}
    streams = [torch.cuda.Stream() for _ in range(2)]
    mdls = [model(fuse) for _ in streams]
    def pass_():
        hs = []
        for m, st in zip(mdls, streams):
            with torch.cuda.stream(st):
                m.set_input(inp); m.init_optimize()
        for stage in one.strategy:
            for m, st in zip(mdls, streams):
                with torch.cuda.stream(st):
                    m.run_stage(stage)
        for m, st in zip(mdls, streams):
            with torch.cuda.stream(st):
                m.forward_losses(m.default_loss_weights); hs.append(m.get_pred_result_async())
        return [h.wait() for h in hs]
    pass_(); pass_()
    torch.cuda.synchronize()
    reps = max(1, steps // (2 * fuse))
    t0 = time.perf_counter()
    for _ in range(reps):
        res = pass_()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # work counters: one batch, untimed
    one.set_input({k: v[:B] for k, v in inp.items()}); one.init_optimize()
    one.sdf_counters_start()
    for stage in one.strategy:
        one.run_stage(stage)
    c = one.sdf_counters_stop()
    n_it = 4 * (epoch + 1)
    per = lambda k: c[k] / (B * n_it)
    return dict(asset=asset, batch_generator="interlocked" if inter else "default", images_per_s=2 * fuse * reps * B / dt,
                inside_voxels_per_sample_iteration=per("inside_voxels"), full_search_share=c["voxels_full_search"] / max(c["inside_voxels"], 1),
                refused_lists_per_sample_iteration=per("lists_refused"), refused_share_of_full_searches=c["lists_refused"] / max(c["voxels_full_search"], 1),
                rebuilt_share=c["voxels_rebuilt"] / max(c["inside_voxels"], 1),
                mean_penetration_depth_m=float(np.mean(res[0]["collision_loss_origin_scale"])),
                schedule=f"2 streams x {fuse} batches of {B}, {reps} pass(es)")


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_topology():
    """(physical cores, hardware threads) of this host: distinct (physical id, core id) pairs of /proc/cpuinfo and os.cpu_count()."""
    threads = os.cpu_count() or 1
    cores = set()
    try:
        with open("/proc/cpuinfo") as fh:
            phys = core = None
            for line in fh:
                k, _, v = line.partition(":")
                k = k.strip()
                if k == "physical id":
                    phys = v.strip()
                elif k == "core id":
                    core = v.strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
            if phys is not None and core is not None:
                cores.add((phys, core))
    except OSError:
        pass
    return (len(cores) or threads), threads


def cpu_baseline(batch_cpu, epoch_full, freq, n_samples=32, iters_per_stage=8):
    """The oracle (kind "port": same op graph as the reference -- torch LBS + losses + torch.optim.Adam,
    dense 32^3 voxel SDF in C/OpenMP) on the first `n_samples` samples of the same batch for
    `iters_per_stage` iterations per stage, extrapolated linearly to the full iteration count
    (per-iteration cost is constant within a stage).  Timed twice: torch at 32 threads (the small
    LBS / loss tensors are fastest there) and torch at every core (a much shorter sample, see below); the
    dense voxel SDF (C/OpenMP, > 95 % of the CPU time) uses every core both times.  `value` is the better of the two."""
    from ihmr_amd.assets import synthetic_mano
    from ihmr_amd.strategies import make_opt_strategy
    from oracle.opt_ref import OptimizeRef
    cores = os.cpu_count() or 1
    sub = {k: v[:n_samples].clone() for k, v in batch_cpu.items()}
    n_fwd = 4 * iters_per_stage + 1           # + final forward (no backward; counted as a full iteration: conservative)
    full_iters = 4 * (epoch_full + 1) + 1
    runs, oracle_out = {}, None
    for threads in sorted({min(32, cores), cores}):
        torch.set_num_threads(threads)
        # every core: 256 torch threads on 2.3 KB tensors spend their time in fork / join (~10 s per iteration measured on a
        # 256-thread host), so that figure is taken on ONE iteration of stage 0 (+ the final forward) -- stated, not the baseline
        short = threads > 32
        strat = make_opt_strategy(0)[:1] if short else make_opt_strategy(iters_per_stage - 1)
        evals = 2 if short else n_fwd
        orc = OptimizeRef(synthetic_mano(True), synthetic_mano(False), n_samples, strat, save_mid_freq=1)
        orc.set_input(sub)
        orc.init_optimize()
        t0 = time.perf_counter()
        orc.optimize()
        t_total = time.perf_counter() - t0
        if not short:
            oracle_out = dict(result=orc.get_pred_result(), selected=[np.asarray(x).copy() for x in orc.selected], n_samples=n_samples,
                              iters_per_stage=iters_per_stage)
            # the float64 ARBITER of the same sample (untimed, not part of the baseline): the same loop in double precision -- how far is
            # each float32 implementation from the exact trajectory?  (parity.vs_f64)
            arb = OptimizeRef(synthetic_mano(True), synthetic_mano(False), n_samples, strat, save_mid_freq=1, dtype=torch.float64)
            arb.set_input(sub); arb.init_optimize(); arb.optimize()
            oracle_out["result_f64"] = arb.get_pred_result()
            oracle_out["selected_f64"] = [np.asarray(x).copy() for x in arb.selected]
        runs[threads] = dict(seconds=t_total, forward_backward_evaluations=evals, images_per_s=n_samples / (t_total / evals * full_iters),
                             ms_per_refine_iter=1000.0 * t_total / evals)
    best = max(runs, key=lambda k: runs[k]["images_per_s"])
    phys, hw = cpu_topology()
    # `cores` = the threads the timed leg actually used (the driver contract's field): torch at `best` threads; the dense voxel SDF
    # (C/OpenMP, > 95 % of the time) on every hardware thread of the host
    return dict(value=runs[best]["images_per_s"], unit="images/s", cores=cores, physical_cores=phys, hardware_threads=hw,
                threads_used=dict(torch=best, openmp_sdf=cores), cpu_model=cpu_model_string(), kind="port",
                torch_threads=best, by_torch_threads={str(k): v for k, v in runs.items()},
                sample=f"{n_samples} samples x {4 * iters_per_stage} refine iterations (+1 forward) measured in {runs[best]['seconds']:.1f}s, "
                       f"extrapolated linearly to {full_iters - 1} iterations",
                ms_per_refine_iter=runs[best]["ms_per_refine_iter"]), oracle_out


def parity_vs_oracle(batch_cpu, oracle_out, rank):
    """The HIP path on exactly the sample the `cpu_baseline` leg pushed through the oracle (same `n_samples` samples, same
    4 x `iters_per_stage` refinement iterations, a snapshot per iteration) -> worst differences, so that the line carries an
    oracle value beside every parity number (BASELINE.json: MPJPE / penetration depth within 1e-4)."""
    from ihmr_amd.optimize_model import OptimizeModel
    n, it = oracle_out["n_samples"], oracle_out["iters_per_stage"]
    m = OptimizeModel(make_opt(n, it - 1, 1, rank))
    m.set_input({k: v[:n].cuda() for k, v in batch_cpu.items()})
    m.init_optimize()
    m.optimize()
    torch.cuda.synchronize()
    got, ref = m.get_pred_result(), oracle_out["result"]
    sel = [x.cpu().numpy() for x in m.selected_history]
    agree = float(np.mean([np.mean(a == b) for a, b in zip(sel, oracle_out["selected"])]))
    err = lambda k: float(np.abs(got[k].astype(np.float64) - ref[k].astype(np.float64)).max())
    verts = max(err("pred_right_hand_verts"), err("pred_left_hand_verts"))
    pen_h, pen_o = float(np.mean(got["collision_loss_origin_scale"])), float(np.mean(ref["collision_loss_origin_scale"]))
    # MPJPE of both sides against the synthetic annotation, the reference's metric (root-aligned joints are what both export)
    mp = lambda r: float(np.mean(np.linalg.norm(r["pred_joints_3d"] - r["gt_joints_3d"][..., :3], axis=-1)))
    vs_f64 = None
    if "result_f64" in oracle_out:
        a = oracle_out["result_f64"]
        same = np.all(np.stack(sel) == np.stack(oracle_out["selected_f64"]), axis=0) & np.all(np.stack(sel) == np.stack(oracle_out["selected"]), axis=0)
        dist = lambda x, k: np.abs(x[k][same].astype(np.float64) - a[k][same])
        vs_f64 = dict(note="float64 arbiter: the oracle's loop in double precision on the same sample; distances over the samples whose "
                           "selections agree in all three runs", samples_compared=int(same.sum()))
        for name, k in (("joints", "pred_joints_3d"), ("right_verts", "pred_right_hand_verts"), ("left_verts", "pred_left_hand_verts"),
                        ("pen_depth", "collision_loss_origin_scale")):
            dh, do = dist(got, k), dist(ref, k)
            vs_f64[name] = dict(hip_max_m=float(dh.max()), hip_mean_m=float(dh.mean()), oracle32_max_m=float(do.max()), oracle32_mean_m=float(do.mean()))
    return dict(sample=f"{n} samples x {4 * it} refine iterations (+ final forward), snapshot every iteration: the cpu_baseline sample",
                vs_f64=vs_f64,
                max_abs_joint_m=err("pred_joints_3d"), max_abs_vertex_m=verts, max_abs_pen_depth_m=err("collision_loss_origin_scale"),
                selection_agreement=agree, mean_penetration_depth_m=dict(hip=pen_h, oracle=pen_o, abs_diff=abs(pen_h - pen_o)),
                mpjpe_m=dict(hip=mp(got), oracle=mp(ref), abs_diff=abs(mp(got) - mp(ref))), tolerance_m=1e-4,
                within_tolerance=bool(err("pred_joints_3d") < 1e-4 and verts < 1e-4 and err("collision_loss_origin_scale") < 1e-4 and agree == 1.0))


RIDGE_FLOP_PER_BYTE = FP32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)     # 19.7: below it a kernel is priced against HBM

# Per-unit models of the three kernels (DESIGN.md section 5.3 spells the arithmetic out).  Bytes: what the algorithm has to move
# once, not what the implementation moves (that is `traffic`, from the counters).
HAND_TABLE_BYTES = 1538 * (16 + 4) + 778 * 16                 # a hand's triangle records (circle 16 B + packed normal 4 B) + normalised vertices
HAND_VERT_BYTES = 778 * 12
KERNEL_SLOTS = (("sdf_dist_kernel", 1), ("sdf_prep_kernel", 0), ("opt_tail_kernel", 2))      # (name prefix in the profiles, ihmr_kernel_timer slot)


def kernel_models(kernel, hands, st):
    """(algorithmic bytes, executed-or-modelled flops) of ONE launch of `kernel` over `hands` hands; st = per-launch work counters."""
    samples = hands / 2.0
    if kernel == "sdf_dist_kernel":
        # every hand's tables once + per EVALUATED inside voxel (`inside_voxels`: what the prep kernel hands over -- a static hand hands
        # over only its new voxels) its work-list entry, map word (8 B since round 5: list | nearest triangle + that triangle's corner
        # ids), candidate list (384 B) and result;
        # flops = what the kernel executed: 11 per bounding-sphere test, 30 per plane + circle test, 75 per exact distance
        return (hands * HAND_TABLE_BYTES + st["inside_voxels"] * (4 + 8 + 384 + 4),
                11.0 * st["sphere_tests"] + 30.0 * st["plane_tests"] + 75.0 * st["dist_evals"])
    if kernel == "sdf_prep_kernel":
        # a hand that is rebuilt reads both hands' vertices (own: box + normalisation, other: needed-voxel mask) and the lists' reference
        # pose + bitmap (9.3 + 4 KB), writes its tables and box and the 778 cell words of the other hand's queries; per needed voxel
        # 4 B (phi = 0 or a work-list entry).  Round 6: the model follows what opt_default really asks of the kernel (round 5 charged
        # every hand the full set, 78 KB, and the counters came out BELOW the model): in the translation stage -- a quarter of the
        # iterations -- both hands are static in their own frames and move only the other hand's vertices, the cell words and the box
        # (12.5 KB); in the other stages 22 % of the hands hand the distance kernel no voxel and skip their triangle records (30.8 KB)
        # flops: ~20 per vertex pair (normalisation, voxel index), ~130 per triangle record, 14 per (triangle, column) ray test
        # + 20 per hit-mask evaluation (counted together in ray_tests: (u, v) tests + needed voxels of the hit columns)
        full = 2 * HAND_VERT_BYTES + 9336 + 4096 + HAND_TABLE_BYTES + 16 + 778 * 4
        static = HAND_VERT_BYTES + 16 + 778 * 4
        return (hands * (0.25 * static + 0.75 * (full - 0.22 * 1538 * 20)) + st["needed_voxels"] * 4,
                hands * (778 * 20.0 + 1538 * 130.0) + st["ray_tests"] * 17.0)
    # opt_tail_kernel, per sample: sampling reads the 1556 cell words (round 6: they carry the corner masks; the two 4 KB bitmaps are no longer read) and -- for the queries that
    # touch an inside voxel: priced as all of them, the model of round 4 -- both hands' vertices + one phi value per needed voxel; the LBS backward reads v_posed of
    # both hands (+ the 3 KB skeleton records); the translation / orientation form also skins the next vertices (vertices out).  The
    # gradients between the phases stay in LDS and the per-vertex depths are not written inside the loop: no bytes.  flops: 60 per
    # sampled vertex (trilinear value + gradient), LBS backward over the four non-zero weights (778 x 4 x 24 x 2 per hand), skinning
    # 778 x 4 x 24 per hand
    skin = 0.5            # the translation / orientation stages (half the iterations of opt_default) also skin: vertices out
    return (samples * (2 * HAND_VERT_BYTES + 1556 * 4 + 2 * (HAND_VERT_BYTES + 3136) + skin * 2 * HAND_VERT_BYTES) + st["needed_voxels"] * 4,
            samples * (1556 * 60.0 + 2 * 778 * 4 * 24 * 2.0 + skin * 2 * 778 * 4 * 24))


def kernel_rooflines(args, B, plan, instance, inputs, hip):
    import ctypes as C
    sizes_run = {}
    for q in plan(args.steps):
        for g in q:
            sizes_run[g] = sizes_run.get(g, 0) + 1
    g_main = max(sizes_run, key=lambda g: g * sizes_run[g])       # the launch size that carried most of the timed region's work
    per_kernel = {k: dict(by_launch_size=[], tot_ms=0.0, tot_bytes=0.0, tot_flops=0.0, tot_full=0.0) for k, _ in KERNEL_SLOTS}
    for g in sorted(sizes_run):
        mdl = instance(0, g)
        timer = hip.KernelTimer()
        hip.lib().ihmr_set_kernel_timer(C.byref(timer))
        graphs = mdl.use_graphs
        mdl.use_graphs = False   # event records cannot sit inside a captured graph
        mdl.set_input(inputs[g]); mdl.init_optimize(); mdl.optimize(0, 1)
        torch.cuda.synchronize()
        hip.lib().ihmr_flush_kernel_timer()
        hip.lib().ihmr_set_kernel_timer(None)
        # work EXECUTED per launch, from the kernels' own counters (DESIGN.md "Measurement"): the same refinement once more,
        # untimed (the counters are global atomics), one sdf_prep_kernel + one sdf_dist_kernel launch per iteration
        n_launch = max(int(timer.n[hip.TIMED_SDF_DIST]), 1)
        stats = None
        if not args.no_work_counters:
            mdl.set_input(inputs[g]); mdl.init_optimize()
            mdl.sdf_counters_start()
            mdl.optimize(0, 1)
            cnt = mdl.sdf_counters_stop()
            stats = dict(sphere_tests=cnt["sphere_tests"] / n_launch, plane_tests=cnt["plane_tests"] / n_launch, dist_evals=cnt["dist_evals"] / n_launch,
                         voxels_from_lists=cnt["voxels_from_lists"] / n_launch,
                         voxels_full_search=cnt["voxels_full_search"] / n_launch,
                         inside_voxels=cnt["inside_voxels"] / n_launch, needed_voxels=cnt["needed_voxels"] / n_launch,
                         ray_tests=cnt["ray_tests"] / n_launch)
        mdl.use_graphs = graphs
        for kernel, slot in KERNEL_SLOTS:
            ms = timer.launch_ms(slot)
            if ms is None or ms <= 0:
                continue
            ab, fl = kernel_models(kernel, 2.0 * g * B, stats) if stats else (None, None)
            e = dict(batches_per_launch=g, launch_sequences_in_timed_region=sizes_run[g], avg_launch_ms=ms, launches_timed=int(timer.n[slot]),
                     algorithmic_bytes_per_launch=ab, flops_per_launch=fl)
            if kernel == "sdf_dist_kernel":
                e["work_per_launch"] = stats
                e["full_search_flops_per_launch"] = (1538 * 11.0 * stats["inside_voxels"] + 75.0 * stats["dist_evals"]) if stats else None
            pk = per_kernel[kernel]
            pk["by_launch_size"].append(e)
            if stats:
                pk["tot_ms"] += ms * sizes_run[g]; pk["tot_bytes"] += ab * sizes_run[g]; pk["tot_flops"] += fl * sizes_run[g]
                if kernel == "sdf_dist_kernel":
                    pk["tot_full"] += e["full_search_flops_per_launch"] * sizes_run[g]
    prefix, meta = committed_profile(g_main, "opt") if (B == 64 and args.epoch == 49) else (None, None)
    kernels = []
    for kernel, _ in KERNEL_SLOTS:
        pk = per_kernel[kernel]
        t = pk["tot_ms"] * 1e-3
        ach_bw = pk["tot_bytes"] / t / 1e9 if t > 0 else None
        ach_fl = pk["tot_flops"] / t / 1e12 if t > 0 else None
        ai = pk["tot_flops"] / pk["tot_bytes"] if pk["tot_bytes"] > 0 else None
        main_ms = next((p["avg_launch_ms"] for p in pk["by_launch_size"] if p["batches_per_launch"] == g_main), None)
        # counters of the committed profile of THIS code (source hash checked) at the main launch size: HBM bytes, vector instructions
        traffic = prof_us = n_inst = share = None
        if prefix is not None:
            tr, ks, sq = profile_rows(prefix, "pmc_traffic", kernel), profile_rows(prefix, "kernel_stats", kernel), profile_rows(prefix, "pmc_sq", kernel)
            traffic = float(tr[0]["hbm_bytes_per_launch"]) if tr else None
            prof_us = float(ks[0]["avg_us"]) if ks else None
            share = sum(float(r["percent"]) for r in ks) if ks and "percent" in ks[0] else None
            n_inst = float(sq[0]["SQ_INSTS_VALU_per_launch"]) if sq and sq[0].get("SQ_INSTS_VALU_per_launch") else None
        issue = None
        if n_inst and main_ms:
            # vector-ALU ISSUE slots: SQ_INSTS_VALU x the calibrated cycles per wave64 instruction of this instruction mix
            # (VALU_ISSUE_CYCLES above) / (1024 SIMDs x the cycles of this run's launch at the 2.4 GHz peak clock): how busy the vector
            # pipes are with instructions of ANY kind (a lower bound: the sustained clock is below the peak clock)
            issue = dict(valu_wave_instructions_per_launch=n_inst, simds=1024, cycles_per_wave_instruction=VALU_ISSUE_CYCLES, clock_ghz=2.4,
                         calibration="profiles/r5_issue_rates.txt (scripts/microbench_issue) x the kernels' static instruction mix (scripts/isa_mix.py)",
                         frac=n_inst * VALU_ISSUE_CYCLES / (1024.0 * main_ms * 1e-3 * 2.4e9))
        ctr_bw = traffic / (main_ms * 1e-3) / 1e9 if (traffic and main_ms) else None
        main_bytes = next((p["algorithmic_bytes_per_launch"] for p in pk["by_launch_size"] if p["batches_per_launch"] == g_main), None)
        kernels.append(dict(
            kernel=kernel, bound="hbm" if (ai is None or ai < RIDGE_FLOP_PER_BYTE) else "valu", arithmetic_intensity_flop_per_byte=ai,
            hbm=dict(achieved=ach_bw, peak=HBM_PEAK_GBS, unit="GB/s", frac=(ach_bw / HBM_PEAK_GBS) if ach_bw else None),
            valu=dict(achieved=ach_fl, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=(ach_fl / FP32_PEAK_TFLOPS) if ach_fl else None),
            issue_slots=issue, traffic=traffic, traffic_unit="bytes/launch",
            traffic_over_algorithmic=(traffic / main_bytes) if (traffic and main_bytes) else None,
            counter_traffic_rate_gbs=ctr_bw, profile_avg_launch_us=prof_us, share_of_gpu_time_pct=share,
            avg_launch_ms_main_size=main_ms, by_launch_size=pk["by_launch_size"]))
    dom = kernels[0]                              # sdf_dist_kernel: the largest share of GPU time in profiles/
    dpk = per_kernel["sdf_dist_kernel"]
    ach_full = dpk["tot_full"] / (dpk["tot_ms"] * 1e-3) / 1e12 if dpk["tot_ms"] > 0 and dpk["tot_full"] > 0 else None
    src = ("profiles/" + os.path.basename(prefix)) if prefix else None
    return dict(
        # headline = the dominant kernel against the roof its arithmetic intensity selects (AI = flops / algorithmic bytes against the
        # ridge 157.3 TFLOP/s / 8 TB/s = 19.7 flop/B) -- fixed by the model, NOT "whichever fraction is larger"
        bound=dom["bound"], kernel="sdf_dist_kernel",
        achieved=dom[dom["bound"]]["achieved"], peak=dom[dom["bound"]]["peak"], unit=dom[dom["bound"]]["unit"], frac=dom[dom["bound"]]["frac"],
        traffic=dom["traffic"], traffic_unit="bytes/launch", traffic_batches_per_launch=g_main,
        traffic_source=(src + "_pmc_traffic.csv") if src and dom["traffic"] else None,
        ridge_flop_per_byte=RIDGE_FLOP_PER_BYTE, valu=dict(dom["valu"], issue_slots=dom["issue_slots"]), hbm=dom["hbm"],
        kernels=kernels, profile=src,
        full_search_equivalent=dict(achieved=ach_full, frac=(ach_full / FP32_PEAK_TFLOPS) if ach_full else None,
                                    note="sdf_dist_kernel priced with the work of searching all 1538 triangles for every inside voxel "
                                         "(the kernel without its candidate lists; rounds 1 and early 2 were priced this way)"),
        history_basis="`frac` of earlier rounds is NOT one series: r1 = executed flops / the fp32 MFMA-VALU peak priced as a FULL search "
                      "(bound 'mfma', 0.139), r2 = executed flops of the culled search / the vector peak (bound 'valu', 0.054), r3 = "
                      "algorithmic bytes / HBM peak because that fraction was the larger one (bound 'hbm', 0.178; on r2's basis 0.043).  "
                      "From r4 on the roof is fixed by arithmetic intensity (all three kernels: HBM) and `valu` / `issue_slots` are always "
                      "given beside it; compare rounds through kernels[].avg_launch_ms_main_size and `value`.",
        traffic_note="HBM bytes per launch from the committed rocprofv3 PMC summary of this command at --streams 1, the same launch size and "
                     "the SAME library source hash (counters cannot be read from inside the process); null when no profile of the loaded "
                     "library is committed",
        note="times = HIP events around each kernel's in-loop launches on their stream, single-stream graph-less passes, one per launch "
             "size of the timed region, aggregated by the number of launch sequences of that size; algorithmic bytes / flops per launch "
             "from the collision kernels' own work counters x the per-unit models of DESIGN.md section 5.3")


def committed_profile(batches_per_launch=None, tag=None):
    """Newest committed profile set of THIS code: `profiles/<name>_meta.json` (written by scripts/profile_round.sh and its
    siblings) records the source hash of the library that was profiled; a set is used only when that hash equals the hash of
    the library this process loads (ihmr_amd/libihmr_hip.srchash) -- a profile of older code is never quoted.
    Returns (prefix, meta) or (None, None)."""
    import glob
    from ihmr_amd import hip
    cur = hip.loaded_source_hash()
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_meta.json")):
        try:
            with open(f) as fh:
                meta = json.load(fh)
        except (OSError, ValueError):
            continue
        if cur is None or meta.get("srchash") != cur:
            continue
        if batches_per_launch is not None and meta.get("batches_per_launch") != batches_per_launch:
            continue
        if tag is not None and meta.get("config") != tag:
            continue
        if best is None or meta.get("unix_time", 0) > best[1].get("unix_time", 0):
            best = (f[:-len("_meta.json")], meta)
    return best if best else (None, None)


def profile_rows(prefix, suffix, kernel):
    """Rows of `<prefix>_<suffix>.csv` whose kernel name starts with `kernel`, most-launched shape first."""
    import csv
    try:
        with open(f"{prefix}_{suffix}.csv", newline="") as fh:
            rows = [r for r in csv.DictReader(fh) if r["kernel"].startswith(kernel)]
    except OSError:
        return []
    key = "launches" if rows and "launches" in rows[0] else "calls"
    return sorted(rows, key=lambda r: -int(r[key]))


def pmc_traffic(kernel, batches_per_launch, tag="opt"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of the SAME code at THIS launch size (two separate
    `--pmc` passes, FETCH_SIZE doubled per the gfx950 correction; scripts/profile_round.sh).  Counters cannot be read from inside
    the process, so this is the profile's figure; (None, None, None) when no profile of the loaded library exists."""
    prefix, meta = committed_profile(batches_per_launch, tag)
    if prefix is None:
        return None, None, None
    rows = profile_rows(prefix, "pmc_traffic", kernel)
    if not rows:
        return None, None, None
    kt = profile_rows(prefix, "kernel_stats", kernel)
    return float(rows[0]["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(prefix) + "_pmc_traffic.csv", (float(kt[0]["avg_us"]) if kt else None)


def encoder_traffic():
    """HBM bytes of ONE encoder forward (64 images) from the committed PMC profile of `bench.py --config baseline` of the loaded
    library: sum over the convolution kernels' rows of bytes x launches, divided by the number of forwards in the profiled run
    (= launches of the stem, the only layer on the generic-gather variant).  (None, None) without such a profile."""
    prefix, meta = committed_profile(None, "baseline")
    if prefix is None:
        return None, None
    rows = profile_rows(prefix, "pmc_traffic", "conv_")
    # the stem is the only layer on the 4-channel gather: conv_igemm_kernel<BM, BN, MODE = CONV_C4 (2)>
    stem = [r for r in rows if r["kernel"].startswith("conv_igemm_kernel") and r["kernel"].rstrip().endswith(", 2>")]
    if not rows or not stem:
        return None, None
    fwd = sum(int(r["launches"]) for r in stem)
    return sum(float(r["hbm_bytes_per_launch"]) * int(r["launches"]) for r in rows) / fwd, "profiles/" + os.path.basename(prefix) + "_pmc_traffic.csv"


def encoder_mfma_busy():
    """MFMA-busy share of every convolution kernel AND launch shape (= groups of layers) of one encoder forward, from the committed SQ
    counter profile of `bench.py --config baseline` of the loaded library: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / 32 (the profile's
    `mfma_busy_frac` column / 32, the normalisation of DESIGN.md section 9), with the launches per forward and the kernel-trace
    duration; longest groups first.  [] without such a profile."""
    prefix, meta = committed_profile(None, "baseline")
    if prefix is None:
        return []
    sq = profile_rows(prefix, "pmc_sq", "conv_")
    kt = {(r["kernel"], r["workgroups"]): r for r in profile_rows(prefix, "kernel_stats", "conv_")}
    stem = [r for r in sq if r["kernel"].startswith("conv_igemm_kernel") and r["kernel"].rstrip().endswith(", 2>")]
    fwd = sum(int(r["launches"]) for r in stem) or 1
    out = []
    for r in sq:
        if not r.get("mfma_busy_frac"):
            continue
        k = kt.get((r["kernel"], r["workgroups"]))
        out.append(dict(kernel=r["kernel"], workgroups=int(r["workgroups"]), launches_per_forward=int(r["launches"]) / fwd,
                        mfma_busy=float(r["mfma_busy_frac"]) / 32.0, avg_us=float(k["avg_us"]) if k else None))
    return sorted(out, key=lambda d: -(d["avg_us"] or 0) * d["launches_per_forward"])


def secondary(config, with_cpu=True):
    """BASELINE.json configs[1] / configs[2] (parity-test cases, NOT the driver's metric): IHMR-Baseline batch 64 and
    IHMR-MLP batch 128 inference on one MI355X with the CPU oracle timed beside them on a bounded sample (BASELINE.md
    section 3: "reported per config").  Returned as a dict: the default run attaches both to its line under
    `secondary_configs` (so that a driver record exists for them); `python bench.py --config baseline|mlp` prints one alone."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import seeded_state_dict
    from ihmr_amd import two_hand
    from ihmr_amd.assets import synthetic_mano
    from ihmr_amd.strategies import make_mlp_strategy
    from ihmr_amd.synthetic import synthetic_opt_batch
    torch.cuda.set_device(0)
    cores = os.cpu_count() or 1
    torch.set_num_threads(min(32, cores))
    # a clean slate between independent legs of the line: the instances of the leg before (reference cycles through their captured
    # graphs and streams) and the allocator's cached blocks cost the NEXT leg host time -- IHMR-MLP read 123 k images/s behind the
    # IHMR-Baseline leg and 135.7 k alone or after this (scripts/experiments/mlp_after_baseline.py)
    import gc
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

    def timeit(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def opt(B):
        return types.SimpleNamespace(isTrain=False, dist=False, process_rank=-1, batchSize=B, inputSize=224, input_nc=3, num_joints=42,
                                     total_params_dim=122, cam_params_dim=3, pose_params_dim=96, shape_params_dim=20, trans_params_dim=3,
                                     model_root="", mean_param_file="mean_mano_params.pkl", checkpoints_dir="./checkpoints",
                                     strategy="mlp_default")
    if config == "baseline":
        from ihmr_amd.baseline_model import InterHandModel
        from oracle.encoder_ref import InterHandEncoderRef
        from oracle.mano_ref import ManoRef
        from oracle.sdf_ref import SDFLossRef
        B, Bc = 64, 64
        m = InterHandModel(opt(B)); m.eval()
        fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
        cpu_batch = synthetic_opt_batch(B, fwd, seed=1234, with_image=True)
        batch = {k: v.cuda() for k, v in cpu_batch.items()}
        pend = []
        def step():              # the export of batch i is collected while batch i + 1 runs (get_pred_result_async)
            m.set_input(batch); m.test(); pend.append(m.get_pred_result_async())
            if len(pend) > 1:
                pend.pop(0).wait()
        dt = timeit(step, 20, 3)
        enc_dt_eager = timeit(lambda: m.encoder(batch["img"]), 10, 3)
        # the encoder as test() runs it: replayed from a captured graph (InterHandModel.use_test_graph)
        enc_graph, img_static = torch.cuda.CUDAGraph(), batch["img"].clone()
        torch.cuda.synchronize()
        with torch.cuda.graph(enc_graph):
            m.encoder(img_static)
        enc_dt = timeit(enc_graph.replay, 40, 5)
        # two batches of 64 in flight: a second model instance on a second stream (the tails of one stream's convolutions --
        # 784 workgroups on 256 CUs -- are filled by the other's)
        m2 = InterHandModel(opt(B)); m2.eval()
        m2.encoder.load_state_dict(m.encoder.state_dict())
        st2 = [torch.cuda.Stream(), torch.cuda.Stream()]
        pend2 = []
        def step2():
            hs = []
            for mm, st in ((m, st2[0]), (m2, st2[1])):
                with torch.cuda.stream(st):
                    mm.set_input(batch); mm.test(); hs.append(mm.get_pred_result_async())
            while pend2:
                pend2.pop(0).wait()
            pend2.extend(hs)
        torch.cuda.synchronize()
        dt2 = timeit(step2, 12, 3) / 2
        del m2
        enc_traffic, enc_traffic_src = encoder_traffic()
        out = dict(metric="images/sec, IHMR-Baseline (ResNet-50 + MANO regress) batch=64 inference", value=B / dt, unit="images/s", n_gpus=1,
                   ms_per_step=dt * 1e3, dtype="f32", data="synthetic", higher_is_better=True,
                   config=dict(workload="BASELINE.json configs[1]: InterHandModel.test() + get_pred_result(), batch 64, 224x224"),
                   two_batches_in_flight=dict(images_per_s=B / dt2, ms_per_batch=dt2 * 1e3, note="two model instances on two HIP streams"),
                   roofline=dict(bound="mfma", kernel="conv_streamk_kernel + conv_igemm_kernel (whole encoder)", achieved=8.2e9 * B / enc_dt / 1e12,
                                 clock_note="peak = 157.3 TFLOP/s at 2.4 GHz; the shader clock read inside these kernels (shader-clock counter against the "
                                            "100 MHz wall clock, scripts/experiments/encoder_clock_in_pass.py with a -DCONV_STAMPS build) ramps from 2.0-2.1 GHz "
                                            "in the first milliseconds of a burst to 2.33-2.34 GHz in a sustained run (this figure: 45 passes back to back)",
                                 peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=8.2e9 * B / enc_dt / 1e12 / FP32_PEAK_TFLOPS,
                                 traffic=enc_traffic, traffic_unit="bytes per encoder forward (64 images)", traffic_source=enc_traffic_src,
                                 mfma_busy_by_layer_group=encoder_mfma_busy(),
                                 hbm=dict(achieved=(enc_traffic / enc_dt / 1e9) if enc_traffic else None, peak=HBM_PEAK_GBS, unit="GB/s",
                                          frac=(enc_traffic / enc_dt / 1e9 / HBM_PEAK_GBS) if enc_traffic else None,
                                          note="counter bytes of the profile / this run's encoder time"),
                                 encoder_ms_per_batch=enc_dt * 1e3, encoder_ms_per_batch_eager=enc_dt_eager * 1e3,
                                 timing="40 replays of the captured encoder pass (as InterHandModel.test() replays it); eager launches: encoder_ms_per_batch_eager"),
                   cpu_baseline=None)
        if not with_cpu:
            return out
        # CPU oracle on the first Bc images: encoder + two MANO evaluations (prediction, annotation) + the collision metric
        right, left = synthetic_mano(True), synthetic_mano(False)
        ref = InterHandEncoderRef(m.mean_params[:Bc].clone()); ref.load_state_dict({k: v.cpu() for k, v in m.encoder.state_dict().items()}); ref.eval()
        mr, ml = ManoRef(right), ManoRef(left)
        sdf = SDFLossRef(right["faces"], left["faces"])
        def two(pose, shape, trans):
            o = {}
            for name, mm, ps, bs in (("right", mr, 0, 0), ("left", ml, 48, 10)):
                r = mm(global_orient=pose[:, ps:ps + 3], hand_pose=pose[:, ps + 3:ps + 48], betas=shape[:, bs:bs + 10])
                o[name] = (r.vertices, r.joints)
            shift = trans.reshape(-1, 1, 3) + (o["right"][1][:, 0:1] - o["left"][1][:, 0:1])
            return o["right"][0], o["left"][0] + shift
        t0 = time.perf_counter()
        with torch.no_grad():
            fp, _ = ref(cpu_batch["img"][:Bc])
            rv, lv = two(fp[:, 3:99], fp[:, 99:119], fp[:, 119:122])
            sdf(torch.stack([rv, lv], 1), return_per_vert_loss=True, return_origin_scale_loss=True)
            two(cpu_batch["mano_pose"][:Bc], cpu_batch["mano_betas"][:Bc], cpu_batch["hand_trans"][:Bc, 0, :3])
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=Bc / tc, unit="images/s", cores=cores, kind="port",
                                   sample=f"{Bc} images through the oracle (encoder + 2 x two-hand MANO + dense voxel SDF) in {tc:.1f}s")
    else:
        from ihmr_amd.mlp_model import MLPModel
        from oracle.mlp_ref import MLPRef
        B, Bc = 128, 64
        strat = make_mlp_strategy()
        m = MLPModel(opt(B)); m.set_update_info(strat, B)
        orc = MLPRef(synthetic_mano(True), synthetic_mano(False), Bc, strat, num_data=B)
        for i in range(len(strat)):
            m.add_new_network(i)
            sd = seeded_state_dict(orc.nets[i], 900 + i, last_scale=0.02)
            orc.nets[i].load_state_dict(sd); m.sub_network_list[i].load_state_dict(sd)
        m.eval()
        fwd = lambda p, s, t: two_hand.forward_from_packed(m.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
        b = synthetic_opt_batch(B, fwd, seed=1234, with_feat=True)
        b["init_hand_trans"] = b["init_hand_trans"][:, 0, :3].contiguous(); b["img"] = torch.zeros(B, 3, 8, 8)
        batch = {k: v.cuda() for k, v in b.items()}
        pend = []
        def step():
            m.set_input(batch); m.test(); pend.append(m.get_pred_result_async())
            if len(pend) > 1:
                pend.pop(0).wait()
        runs = [timeit(step, 20, 3) for _ in range(5)]            # spread over five runs of 20 batches
        dt = float(np.median(runs))
        # GPU time of test() alone (events on its stream, nothing else in flight) beside the wall time per batch
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        m.set_input(batch); torch.cuda.synchronize()
        gpu = []
        for _ in range(10):
            ev[0].record(); m.test(); ev[1].record(); torch.cuda.synchronize()
            gpu.append(ev[0].elapsed_time(ev[1]))
        gpu_ms = float(np.median(gpu))
        # two batches in flight: a second instance on a second stream (as the Baseline config does)
        m2 = MLPModel(opt(B)); m2.set_update_info(strat, B)
        for i in range(len(strat)):
            m2.add_new_network(i); m2.sub_network_list[i].load_state_dict(m.sub_network_list[i].state_dict())
        m2.eval()
        st2 = [torch.cuda.Stream(), torch.cuda.Stream()]
        pend2 = []
        def step2():
            hs = []
            for mm, st in ((m, st2[0]), (m2, st2[1])):
                with torch.cuda.stream(st):
                    mm.set_input(batch); mm.test(); hs.append(mm.get_pred_result_async())
            while pend2:
                pend2.pop(0).wait()
            pend2.extend(hs)
        torch.cuda.synchronize()
        runs2 = [timeit(step2, 12, 3) / 2 for _ in range(5)]
        dt2 = float(np.median(runs2))
        del m2
        # dominant kernel of the config (largest share of GPU time in the committed profile: sdf_dist_kernel at 256 hands, the
        # full search -- a single-shot caller has no candidate lists): HIP-event time of its in-loop launches in a graph-less pass,
        # work from its own counters in a second pass
        import ctypes as C
        from ihmr_amd import hip
        timer = hip.KernelTimer()
        core_graphs = m._core.use_graphs
        m._core.use_graphs = False
        hip.lib().ihmr_set_kernel_timer(C.byref(timer))
        m.set_input(batch); m._test_eager(); torch.cuda.synchronize()
        hip.lib().ihmr_flush_kernel_timer(); hip.lib().ihmr_set_kernel_timer(None)
        m._core.sdf_counters_start(); m.set_input(batch); m._test_eager(); cnt = m._core.sdf_counters_stop()
        m._core.use_graphs = core_graphs
        nl = max(int(timer.n[hip.TIMED_SDF_DIST]), 1)
        k_ms = timer.launch_ms(hip.TIMED_SDF_DIST)
        flops = (11.0 * cnt["sphere_tests"] + 30.0 * cnt["plane_tests"] + 75.0 * cnt["dist_evals"]) / nl
        abytes = 2.0 * B * (1538 * (16 + 4) + 778 * 16) + cnt["inside_voxels"] / nl * 8
        traffic, tsrc, prof_us = pmc_traffic("sdf_dist_kernel", None, "mlp")
        ach, bw = flops / (k_ms * 1e-3) / 1e12, abytes / (k_ms * 1e-3) / 1e9
        hbm_binds = flops / abytes < RIDGE_FLOP_PER_BYTE        # the roof is fixed by arithmetic intensity
        out = dict(metric="images/sec, IHMR-MLP refinement head batch=128 inference", value=B / dt, unit="images/s", n_gpus=1, ms_per_step=dt * 1e3,
                   dtype="f32", data="synthetic", higher_is_better=True,
                   config=dict(workload="BASELINE.json configs[2]: MLPModel.test() (6 stages, 6 MLPs; of the reference's 8 MANO + SDF evaluations the camera-only stage's is "
                                        "replaced by its exact shortcut, ihmr_mlp_camera_select) + export, batch 128; "
                                        "test() replayed as one captured hipGraph per instance"),
                   runs_images_per_s=[B / r for r in runs], spread_images_per_s=[B / max(runs), B / min(runs)],
                   gpu_ms_per_batch=gpu_ms, wall_ms_per_batch=dt * 1e3,
                   two_batches_in_flight=dict(images_per_s=B / dt2, ms_per_batch=dt2 * 1e3, runs_images_per_s=[B / r for r in runs2],
                                              note="two model instances on two HIP streams"),
                   roofline=dict(bound="hbm" if hbm_binds else "valu", kernel="sdf_dist_kernel (full search, 256 hands per launch)",
                                 achieved=bw if hbm_binds else ach, peak=HBM_PEAK_GBS if hbm_binds else FP32_PEAK_TFLOPS,
                                 unit="GB/s" if hbm_binds else "TFLOP/s", frac=(bw / HBM_PEAK_GBS) if hbm_binds else (ach / FP32_PEAK_TFLOPS),
                                 valu=dict(achieved=ach, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP32_PEAK_TFLOPS),
                                 hbm=dict(achieved=bw, peak=HBM_PEAK_GBS, unit="GB/s", frac=bw / HBM_PEAK_GBS),
                                 avg_launch_ms=k_ms, launches_per_batch=nl, kernel_ms_per_batch=k_ms * nl, share_of_gpu_time=k_ms * nl / gpu_ms,
                                 algorithmic_flops_per_launch=flops, algorithmic_bytes_per_launch=abytes,
                                 traffic=traffic, traffic_unit="bytes/launch", traffic_source=tsrc, profile_avg_launch_us=prof_us),
                   cpu_baseline=None)
        if not with_cpu:
            return out
        orc.set_input({k: v[:Bc].clone() for k, v in b.items()})
        t0 = time.perf_counter()
        orc.test()
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=Bc / tc, unit="images/s", cores=cores, kind="port",
                                   sample=f"{Bc} samples through the oracle's MLPRef.test() in {tc:.1f}s")
    out["cpu_baseline"]["cpu_model"] = cpu_model_string()
    out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this file under `torch.distributed.run` as a CHILD process
    (never an exec, and nothing in this parent has initialised the GPU: `torch.cuda.device_count()` does not on this image) and
    return its exit code; stdout / stderr are inherited, so rank 0's JSON line is this command's JSON line.  A box with fewer
    than N GPUs cannot give RCCL one device per rank: the ranks then share the devices round-robin over gloo (the single-GPU check
    of the N-rank code path, `config.ranks_per_device` > 1 in the line says so) unless IHMR_DIST_BACKEND is set."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    if ndev < n:
        env.setdefault("IHMR_DIST_BACKEND", "gloo")
        print(f"bench.py: --gpus {n} on a box with {ndev} GPU(s): ranks share devices over {env['IHMR_DIST_BACKEND']}", file=sys.stderr)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=64, help="samples per batch (the reference's batchSize: the batch-mean losses run over it)")
    ap.add_argument("--epoch", type=int, default=49, help="opt_default epoch per stage (49 -> 200 iterations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed-by-the-driver extras of the line: single-batch latency, H2D-inclusive rate, batch-512 run, "
                         "secondary configs (profiling runs: keeps the kernel summary to the main workload)")
    ap.add_argument("--no-single-batch-roofline", action="store_true", help="(kept for old scripts; implied by --no-extras)")
    ap.add_argument("--no-work-counters", action="store_true",
                    help="skip the untimed pass that reads the collision kernels' work counters (global atomics: they would distort a "
                         "rocprofv3 profile of this command); roofline.achieved is then null, the launch times stay")
    ap.add_argument("--fuse", type=int, default=10,
                    help="batches of --batch samples carried by ONE launch sequence (opt.fuse_batches; per-sample arithmetic "
                         "identical to separate batches); batches in flight = streams x fuse")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent launch sequences in flight per GPU (each step is still one full pass over one batch of --batch samples)")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="under torch.distributed.run: run the package's collectives (metric all-reduce, MAX, all-gather, bucketed gradient "
                         "all-reduce) on device tensors through the process group's backend and report the checks in the line")
    ap.add_argument("--asset", type=str, default="mitten", choices=["mitten", "fingers"],
                    help="synthetic MANO-shaped model of the run: the default blob, or the five-finger mesh with interlocked batches "
                         "(geometry sensitivity; the default run reports both under `geometry`)")
    ap.add_argument("--config", type=str, default="opt", choices=["opt", "baseline", "mlp"],
                    help="opt = the driver's bench line (IHMR-OPT); baseline / mlp = one of the secondary BASELINE.json configs alone")
    args = ap.parse_args()
    if args.config != "opt":
        assert torch.cuda.is_available(), "bench.py needs an MI355X: the hot path has no CPU fallback"
        print(json.dumps(secondary(args.config, with_cpu=not args.no_cpu_baseline)))
        return

    # --gpus N is a guarantee, not a hint: N ranks run, or the command fails.  Under a launcher (WORLD_SIZE set: the driver's
    # `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) the two must agree; without one and N > 1 this
    # process becomes the launcher -- BEFORE it has touched the GPU -- and passes the ranks' one JSON line through.
    if "WORLD_SIZE" in os.environ:
        if int(os.environ["WORLD_SIZE"]) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    elif args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or ("TORCHELASTIC_RUN_ID" in os.environ and "RANK" in os.environ):   # (one rank under torchrun: RCCL with a 1-rank group)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank %= max(torch.cuda.device_count(), 1)     # one rank per GPU on the driver's node; ranks may share a GPU in the
        torch.cuda.set_device(local_rank)                   # single-GPU check of this code path (IHMR_DIST_BACKEND=gloo)
        # RCCL needs a device per rank; more ranks than devices (a one-GPU box checking the N-rank code path) share them over gloo
        backend = os.environ.get("IHMR_DIST_BACKEND") or ("gloo" if world > max(torch.cuda.device_count(), 1) else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        torch.cuda.set_device(0)
    assert torch.cuda.is_available(), "bench.py needs an MI355X: the hot path has no CPU fallback"
    # one process per GPU on one node: keep each rank's launch thread on the cores next to its GPU (IHMR_PIN_NUMA=0 switches it off)
    numa_pin = None
    if world > 1 and os.environ.get("IHMR_PIN_NUMA", "1") != "0":
        from ihmr_amd import dist as _D
        numa_pin = _D.pin_to_gpu_numa_node(torch.cuda.current_device())

    import ctypes as C
    from ihmr_amd import hip, two_hand
    from ihmr_amd.optimize_model import OptimizeModel
    from ihmr_amd.synthetic import synthetic_opt_batch

    B, freq = args.batch, 10
    S, G = max(1, args.streams), max(1, args.fuse)
    n_iters = 4 * (args.epoch + 1)
    # Batches in flight = S x G.  The kernels of one 64-sample batch are latency-bound and fill at most half of the
    # 256 CUs, so the work is made large first: one launch sequence carries up to G = 10 batches (opt.fuse_batches keeps every
    # sample's arithmetic that of a 64-sample batch, tests/test_gpu_parity.py), and three such sequences run on three HIP streams
    # so that one's per-sample kernels overlap another's collision kernels (the driver's 20 steps run as sequences of 7 + 7 + 6
    # batches: 20.0 k images/s against 19.4 k for 2 x 10 and 16.5 k for 4 x 5, scripts/sweep_flight.sh).  No runtime knobs
    # (hardware-queue counts etc.) involved.
    # --batch 512 --streams 1 --fuse 1 is the reference's recipe verbatim: ONE sequence over a real batch of 512.
    def make_model(fuse, batch=B):
        o = make_opt(batch, args.epoch, freq, rank if world > 1 else -1, args.asset)
        o.fuse_batches = fuse
        return OptimizeModel(o)

    _prio = [int(x) for x in os.environ.get("IHMR_STREAM_PRIORITIES", "").split(",") if x.strip()]      # (experiment knob: per-stream priorities)
    streams = [torch.cuda.Stream(priority=_prio[i % len(_prio)]) if _prio else torch.cuda.Stream() for i in range(S)]
    model = make_model(1)                        # single-batch instance: latency figures, size-1 jobs of stream 0
    pool = {(0, 1): model}                       # (stream, batches per launch sequence) -> instance, built on demand
    fwd = lambda p, s, t: two_hand.forward_from_packed(model.mano_models["right"], p.cuda(), s.cuda(), t.cuda())[2]
    # G DISTINCT synthetic batches (seed 1234 + 1000 i + rank): a launch sequence that carries g batches carries batches 0..g-1
    batches_cpu = [synthetic_opt_batch(B, fwd, seed=1234 + 1000 * i + rank, first_index=(rank * G + i) * B, interlock=args.asset != "mitten")
                   for i in range(G)]
    batch_cpu = batches_cpu[0]
    inputs, inputs_host = {}, {}
    for g in range(1, G + 1):
        cat = {k: torch.cat([bc[k] for bc in batches_cpu[:g]], dim=0) for k in batch_cpu}
        inputs_host[g] = {k: v.pin_memory() for k, v in cat.items()}
        inputs[g] = {k: v.cuda() for k, v in cat.items()}      # resident in HBM before timing
    batch = inputs[1]
    torch.cuda.synchronize()

    host_submit = [0.0]                          # seconds of host submission work per step of the last run_steps()

    def plan(n):
        """n batches -> per stream a list of job sizes (batches fused into one launch sequence, <= G): the streams get
        equal shares (+-1), a share is cut into the fewest jobs of the most equal sizes.  Any n keeps all S streams busy
        to the end (n = 20, S = 4, G = 3: every stream runs a 3 and a 2)."""
        out = []
        for i in range(S):
            c = n // S + (1 if i < n % S else 0)
            k = -(-c // G) if c else 0
            out.append([c // k + (1 if j < c % k else 0) for j in range(k)] if k else [])
        return out

    def instance(i, g):
        if (i, g) not in pool:
            pool[(i, g)] = make_model(g)
            with torch.cuda.stream(streams[i]):      # untimed first pass: captures the stage graphs, allocates the pinned buffers
                m = pool[(i, g)]
                m.set_input(inputs[g]); m.init_optimize(); m.optimize()
                m.get_pred_result_async().wait(); m.get_pred_result_async().wait()
        return pool[(i, g)]

    def run_steps(n, src=inputs):
        """n full passes over one batch each (set_input -> init_optimize -> optimize -> get_pred_result), S streams x up to
        G fused batches in flight.  `src` = the device-resident inputs, or the pinned host copies (then every step's 15
        input tensors cross PCIe inside the step, as in the reference's loop body)."""
        res = None
        sizes = plan(n)
        pending = []                               # export handles of the previous round
        t_wait = 0.0                               # host time spent WAITING for the GPU (export handles): everything else is submission
        t_begin = time.perf_counter()
        for r in range(max((len(q) for q in sizes), default=0) + 1):
            jobs = [(instance(i, q[r]), streams[i], src[q[r]]) for i, q in enumerate(sizes) if r < len(q)]
            for mdl, st, inp in jobs:
                with torch.cuda.stream(st):
                    mdl.set_input(inp)
                    mdl.init_optimize()
            for stage in model.strategy:          # interleave the stages so the host keeps every stream fed
                for mdl, st, _ in jobs:
                    with torch.cuda.stream(st):
                        mdl.run_stage(stage)
            handles = []
            for mdl, st, _ in jobs:
                with torch.cuda.stream(st):
                    mdl.forward_losses(mdl.default_loss_weights)
                    # device -> host export of every batch, as the reference's loop does; queued behind the refinement
                    # on its stream and collected one round later, so the host never leaves the GPU without work
                    handles.append(mdl.get_pred_result_async())
            tw = time.perf_counter()
            for h in pending:
                res = h.wait()
            t_wait += time.perf_counter() - tw
            pending = handles
        # host-side submission cost of the run: wall time of this thread minus the time it sat in the export waits (graph launches,
        # input staging, export requests; a queue that is full blocks inside a launch and counts -- the upper bound is the honest one)
        host_submit[0] = (time.perf_counter() - t_begin - t_wait) / max(n, 1)
        return res

    extras = rank == 0 and world == 1 and not args.no_extras
    needs_single = extras or any(g == 1 for n in (max(args.warmup, 0), args.steps) for q in plan(n) for g in q)
    if needs_single:                                 # the pre-built single-batch instance gets its untimed first pass too (profiling
        with torch.cuda.stream(streams[0]):          # runs without a one-batch job skip it: its launches would blur the per-kernel means)
            model.set_input(batch); model.init_optimize(); model.optimize()
            model.get_pred_result_async().wait(); model.get_pred_result_async().wait()
    # every instance the warm-up and the timed run will use is built (graphs captured, pinned buffers allocated) now
    for n in (max(args.warmup, 0), args.steps):
        for i, q in enumerate(plan(n)):
            for g in q:
                instance(i, g)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, src):
        barrier()
        t0 = time.perf_counter()
        r = run_steps(n, src)
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, r

    run_steps(max(args.warmup, 0))
    # ---- host submission cost (what decides whether N ranks x S streams scale on one node's CPUs): the SAME K steps with the GPU
    #      synchronised out of the picture as far as the loop allows -- the thread's wall time minus its waits for export handles.
    #      Measured on a pass of its own so that the timed region below stays exactly the contract's
    run_steps(args.steps)
    torch.cuda.synchronize()
    submit_ms = 1000.0 * host_submit[0]
    if dist is not None:
        t = torch.tensor([submit_ms], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        submit_ms = float(t.item())
    # (experiment builds only, -DIHMR_TIMELINE + IHMR_TIMELINE_OUT=<file.npy>: workgroup records of the timed region, scripts/timeline_wg.py)
    tl_out = os.environ.get("IHMR_TIMELINE_OUT") if hasattr(hip.lib(), "ihmr_debug_timeline") else None
    if tl_out:
        hip.lib().ihmr_debug_timeline.restype = C.c_long
        hip.lib().ihmr_debug_timeline.argtypes = [C.c_void_p, C.c_long]
        assert hip.lib().ihmr_debug_timeline(None, 4 << 20) == 0
    elapsed, res = timed(args.steps, inputs)
    if tl_out:
        rec = np.zeros((4 << 20, 3), np.uint64)
        n_rec = hip.lib().ihmr_debug_timeline(rec.ctypes.data, 0)
        np.save(tl_out, rec[:max(n_rec, 0)])
    ms_per_step = 1000.0 * elapsed / args.steps
    value = world * B * args.steps / elapsed
    # the same K steps with the per-step host-to-device copy of the inputs inside the timed region (never `value`)
    h2d = None
    if extras:
        run_steps(min(args.warmup, 8), inputs_host)
        e2, _ = timed(args.steps, inputs_host)
        h2d = dict(value=world * B * args.steps / e2, unit="images/s", ms_per_step=1000.0 * e2 / args.steps,
                   note="inputs start in pinned host memory: the 15 input tensors of every step (~40 KB per 64 samples) are copied "
                        "inside the step, as in the reference's loop body (optimize.py:61-71)")

    # ---- the three large kernels of an iteration (sdf_dist_kernel, sdf_prep_kernel, opt_tail_kernel: ~80 % of the GPU time): HIP
    #      events on the launch stream in separate single-stream, graph-less passes of the same workload, one per launch size the
    #      timed region actually ran (with several sequences in flight the kernels share the GPU and a per-launch time is
    #      meaningless); work and algorithmic bytes from the collision kernels' own counters in a second, untimed pass
    roofline = None
    if rank == 0:
        roofline = kernel_rooflines(args, B, plan, instance, inputs, hip)

    # ---- single-batch latency: ONE batch of --batch samples, one stream, nothing else in flight.
    #      ms/refine-iter = stage-loop wall time / iterations (excludes set_input and the export); images/s = batch /
    #      total per-batch wall time (set_input -> init_optimize -> optimize -> blocking get_pred_result)
    latency = None
    if extras:
        loops, totals = [], []
        with torch.cuda.stream(streams[0]):
            for rep in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model.set_input(batch); model.init_optimize()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for stage in model.strategy:
                    model.run_stage(stage)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                model.forward_losses(model.default_loss_weights)
                model.get_pred_result()
                t3 = time.perf_counter()
                loops.append(t2 - t1); totals.append(t3 - t0)
            # per stage (a pass of its own: a synchronize between the stages): microseconds per refinement iteration
            per_stage = [[] for _ in model.strategy]
            for rep in range(5):
                model.set_input(batch); model.init_optimize()
                for si, stage in enumerate(model.strategy):
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                    model.run_stage(stage)
                    torch.cuda.synchronize()
                    per_stage[si].append(time.perf_counter() - ts)
        loop, tot = float(np.median(loops)), float(np.median(totals))
        stage_us = [1e6 * float(np.median(p)) / (args.epoch + 1) for p in per_stage]
        latency = dict(batch=B, streams=1, batches_per_launch=1, ms_per_refine_iter=1000.0 * loop / n_iters, ms_per_batch=1000.0 * tot,
                       images_per_s=B / tot, stage_us_per_refine_iter=stage_us,
                       note="median of 7 passes; stage loop replayed from its hipGraphs; per stage: median of 5 passes of their own")

    # ---- the reference's own recipe (bash/optimize.sh:11,33): batch 512 per process as ONE launch sequence on one stream
    large = None
    if extras and B == 64:
        BL, nL = 512, 4
        big = make_model(1, BL)
        big_in = {k: v.cuda() for k, v in synthetic_opt_batch(BL, fwd, seed=4321 + rank).items()}
        for _ in range(2):
            big.set_input(big_in); big.init_optimize(); big.optimize(); big.get_pred_result_async().wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = None
        for _ in range(nL):
            big.set_input(big_in); big.init_optimize(); big.optimize()
            h = big.get_pred_result_async()
            if pend is not None:
                pend.wait()
            pend = h
        pend.wait()
        torch.cuda.synchronize()
        tL = (time.perf_counter() - t0) / nL
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        big.set_input(big_in); big.init_optimize()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for stage in big.strategy:
            big.run_stage(stage)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # two such batches in flight on two streams
        big2 = make_model(1, BL)
        with torch.cuda.stream(streams[1 % S]):
            big2.set_input(big_in); big2.init_optimize(); big2.optimize(); big2.get_pred_result_async().wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = []
        for _ in range(nL):
            hs = []
            for mdl, st in ((big, streams[0]), (big2, streams[1 % S])):
                with torch.cuda.stream(st):
                    mdl.set_input(big_in); mdl.init_optimize()
            for stage in big.strategy:
                for mdl, st in ((big, streams[0]), (big2, streams[1 % S])):
                    with torch.cuda.stream(st):
                        mdl.run_stage(stage)
            for mdl, st in ((big, streams[0]), (big2, streams[1 % S])):
                with torch.cuda.stream(st):
                    mdl.forward_losses(mdl.default_loss_weights)
                    hs.append(mdl.get_pred_result_async())
            for h in pend:
                h.wait()
            pend = hs
        for h in pend:
            h.wait()
        torch.cuda.synchronize()
        t2s = (time.perf_counter() - t0) / (2 * nL)
        large = dict(batch=BL, streams=1, batches_per_launch=1, images_per_s=BL / tL, ms_per_batch=1000.0 * tL,
                     ms_per_refine_iter=1000.0 * (t2 - t1) / n_iters, steps=nL, two_streams_images_per_s=BL / t2s,
                     note="one OptimizeModel of batchSize 512 (batch-mean losses over 512), one stream, one launch sequence per stage; "
                          "two_streams: two such batches in flight")
        del big2
        del big, big_in

    # ---- geometry sensitivity: the same refinement on the blob and on the five-finger mesh with interlocked hands, one schedule for both
    geometry = None
    if extras and B == 64:
        geometry = [geometry_report(a, B, args.epoch, freq) for a in ("mitten", "fingers")]
        geometry.append(dict(throughput_ratio_fingers_over_mitten=geometry[1]["images_per_s"] / geometry[0]["images_per_s"]))

    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.asset == "mitten":
        cpu, oracle_out = cpu_baseline(batch_cpu, args.epoch, freq)
        parity = parity_vs_oracle(batch_cpu, oracle_out, -1)
    second = None
    if extras and not args.no_cpu_baseline:
        del pool, model
        torch.cuda.empty_cache()
        second = dict(baseline=secondary("baseline"), mlp=secondary("mlp"))

    selftest = None
    if dist is not None and args.rccl_selftest:
        from ihmr_amd import dist as D
        dev = torch.device("cuda", torch.cuda.current_device())
        v = np.arange(9, dtype=np.float64) + rank
        tot = D.reduce_metrics(v)                           # float64 9-vector, summed on a device tensor
        exp = sum(np.arange(9, dtype=np.float64) + r for r in range(world))
        mx = torch.tensor([float(rank)], device=dev, dtype=torch.float64)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        parts = [torch.zeros(4, device=dev) for _ in range(world)]
        dist.all_gather(parts, torch.full((4,), float(rank), device=dev))
        flat = torch.ones(1 << 20, device=dev) * (rank + 1)
        red = D.OverlappedGradientReducer(flat, [(1 << 19, 1 << 20), (0, 1 << 19)], bucket_bytes=1 << 20)
        red.ready(0); red.ready(1)
        f = red.finish()
        torch.cuda.synchronize()
        selftest = dict(backend=dist.get_backend(), world=world, all_reduce_sum_f64_ok=bool(np.array_equal(tot, exp)),
                        all_reduce_max_ok=bool(float(mx.item()) == world - 1),
                        all_gather_ok=bool(all(float(p[0].item()) == i for i, p in enumerate(parts))),
                        gradient_bucket_ok=bool(torch.all(flat == world * (world + 1) / 2).item() and abs(f - 1.0 / world) < 1e-12))

    if rank == 0:
        amortised = ms_per_step / (n_iters + 1)
        out = dict(
            metric="images/sec, IHMR-OPT 200-iter refinement batch=64 (ms/refine-iter: latency.ms_per_refine_iter; amortised: "
                   "ms_per_refine_iter_amortised)",
            value=value, unit="images/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step,
            # throughput figure: whole-run time / (steps x iterations) with streams x fuse batches in flight -- NOT the duration of
            # one refinement iteration (that is latency.ms_per_refine_iter)
            ms_per_refine_iter_amortised=amortised, higher_is_better=True, scaling="weak", vs_baseline=None,
            # multi-GPU readiness measurable on one GPU: the launch thread's work per step (max over ranks).  A rank keeps its GPU fed
            # as long as this stays below the GPU time of a step (ms_per_step of a one-GPU run: on 8 GPUs every rank has its own)
            host_submit_ms_per_step=submit_ms, host_submit_fraction_of_step=submit_ms / ms_per_step if world == 1 else None,
            host_cpu_affinity=numa_pin,
            dtype="f32", data="synthetic",
            config=dict(workload=f"IHMR-OPT opt_default epoch={args.epoch} ({n_iters} refine iterations + final forward), "
                                 f"save_mid_freq={freq}, batch {B}/GPU, synthetic MANO-shaped asset seed 0"
                                 + ("" if args.asset == "mitten" else f" [{args.asset} mesh, interlocked batches: NOT the headline workload]"),
                        global_batch=world * B, refine_iters=n_iters, batches_in_flight_per_gpu=S * G, launch_streams=S,
                        max_batches_per_launch_sequence=G,
                        # what the timed region really ran: job sizes per stream, and how many DISTINCT synthetic batches they touched
                        # (a launch sequence of g batches carries batches 0..g-1 of this rank's G generated ones)
                        launch_sequences_timed=plan(args.steps), distinct_batches=max((g for q in plan(args.steps) for g in q), default=0),
                        distinct_batches_generated=G,
                        devices_visible=torch.cuda.device_count(), ranks_per_device=-(-world // max(torch.cuda.device_count(), 1)),
                        dist_backend=(dist.get_backend() if dist is not None else None),
                        # the metric's second half -- ms per refinement iteration at batch 64 with ONE batch in flight -- copied
                        # here from `latency` because the driver's record keeps `config` (null in --no-extras / multi-rank runs)
                        latency_ms_per_refine_iter=latency["ms_per_refine_iter"] if latency else None,
                        latency_images_per_s=latency["images_per_s"] if latency else None,
                        latency_stage_us_per_refine_iter=latency["stage_us_per_refine_iter"] if latency else None,
                        translated_hand_reuse="on (default; rounding-level, DESIGN 5.0)",
                        stage_list_reuse="on (default; exact: candidate lists kept across stage boundaries, ihmr_opt_stage.keep_lists)",
                        parallelism=f"dp{world} (independent samples, no collective)"),
            roofline=roofline, cpu_baseline=cpu, latency=latency, h2d_inclusive=h2d, large_batch=large, secondary_configs=second,
            geometry=geometry,
            # LBS + losses + Adam are nominally HBM work -- 78 KB of algorithmic traffic per sample and
            # iteration -- and in practice bound by the dependent kernel boundaries of an iteration (3 / 3 / 4 / 7 launches by stage)
            lbs_losses_adam=dict(bound="hbm", algorithmic_bytes_per_sample_iteration=78e3, kernel_launches_per_iteration=4.25,
                                 achieved=78e3 * B * world / (amortised * 1e-3) / 1e9, peak=6300.0, unit="GB/s",
                                 frac=78e3 * B * world / (amortised * 1e-3) / 1e9 / (6300.0 * world),
                                 note="whole-iteration rate at the bench's concurrency: the part is latency-, not bandwidth-bound"),
            parity=dict(translated_reuse="on", timed_run_mean_penetration_depth_m=float(np.mean(res["collision_loss_origin_scale"])), vs_oracle=parity),
        )
        if selftest is not None:
            out["rccl_selftest"] = selftest
        if cpu is not None:
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(out))
    if dist is not None:
        # rank 0's single-stream roofline passes run while the other ranks wait HERE (a barrier they expect), not inside
        # destroy_process_group's teardown with its shorter patience
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
