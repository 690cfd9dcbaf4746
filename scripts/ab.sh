#!/bin/bash
# A/B of library builds on the GPU box (via gpurun).  ONE script for what ab_*.sh / abl*.sh / cmp_bench.sh did in rounds 1-3.
#
#   [MODE=kernels] [FUSE=8] [BATCH=64] [REPS=2] [KERNELS='sdf_|opt_tail'] scripts/ab.sh lib1.so lib2.so ... product
#
# Every argument is a library build (IHMR_HIP_LIBRARY); "product" = the in-tree library -- which the GPU box REBUILDS when the
# sources differ from its build record: compare against an explicit build of HEAD, not against "product", while the tree is modified.
#   MODE=kernels   per-kernel means from one single-stream kernel trace per build at FUSE batches of BATCH samples per launch
#                  (KERNELS = regex of kernel-name prefixes to print; rows with fewer than MINCALLS launches are dropped)
#   MODE=bench     whole-loop throughput: the driver's command (--steps 20 --warmup 5), REPS runs per build, alternating
#   MODE=both      kernels, then bench
#   MODE=baseline  bench.py --config baseline (encoder ms, TFLOP/s, images/s), REPS runs per build
#   MODE=mlp       bench.py --config mlp (images/s, GPU ms per batch), REPS runs per build
#   MODE=train     IHMR-Baseline training step (ms per step), REPS runs per build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
MODE=${MODE:-kernels}; F=${FUSE:-8}; B=${BATCH:-64}; REPS=${REPS:-2}; K=${KERNELS:-sdf_|opt_tail}; MINC=${MINCALLS:-150}
use() { if [ "$1" = "product" ]; then unset IHMR_HIP_LIBRARY; else export IHMR_HIP_LIBRARY=$(realpath "$1"); fi; }
kernels() {
  rm -rf gpurun_out/ab
  timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/ab -o ab -- python3 bench.py --steps $F --warmup $F --streams 1 --fuse $F --batch $B \
      --no-cpu-baseline --no-extras --no-work-counters > /dev/null 2>&1
  python3 scripts/rocprof_summary.py gpurun_out/ab/ab_results.db /tmp/ab.csv > /dev/null
  python3 - "$K" "$MINC" <<'PY'
import csv, re, sys
pat, minc = re.compile("^(?:" + sys.argv[1] + ")"), int(sys.argv[2])
for r in csv.DictReader(open("/tmp/ab.csv")):
    if pat.match(r["kernel"]) and int(r["calls"]) >= minc:
        print(f"   {r['kernel'][:40]:40s} wg {r['workgroups']:>8s} calls {r['calls']:>5s} avg {float(r['avg_us']):8.2f} us  {float(r['percent']):5.1f} %")
PY
  rm -rf gpurun_out/ab
}
value() { python3 -c "import sys,json; print(round(json.loads(sys.stdin.readline())['value']))"; }
if [ "$MODE" = kernels ] || [ "$MODE" = both ]; then
  for lib in "$@"; do use "$lib"; echo "== $lib (kernels, $F x $B samples per launch)"; kernels; done
fi
case "$MODE" in
  kernels) ;;
  bench|both)
    for r in $(seq $REPS); do for lib in "$@"; do use "$lib"
      echo "$lib run $r: $(timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-work-counters 2>/dev/null | value) images/s"
    done; done ;;
  baseline)
    for r in $(seq $REPS); do for lib in "$@"; do use "$lib"
      timeout 300 python3 bench.py --config baseline --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$lib run $r: baseline', round(d['value']), 'img/s; encoder', round(d['roofline']['encoder_ms_per_batch'],3), 'ms =', round(d['roofline']['achieved'],1), 'TFLOP/s; two in flight', round(d['two_batches_in_flight']['images_per_s']))"
    done; done ;;
  mlp)
    for r in $(seq $REPS); do for lib in "$@"; do use "$lib"
      timeout 300 python3 bench.py --config mlp --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$lib run $r: mlp', round(d['value']), 'img/s; GPU', round(d['gpu_ms_per_batch'],3), 'ms per batch; two in flight', round(d['two_batches_in_flight']['images_per_s']))"
    done; done ;;
  train)
    for r in $(seq $REPS); do for lib in "$@"; do use "$lib"
      timeout 300 python3 -m ihmr_amd.run_train_baseline --num_samples 256 --batchSize 64 --total_epoch 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib run $r:', round(d['ms_per_step'],2), 'ms/step')"
    done; done ;;
  *) echo "unknown MODE=$MODE"; exit 2 ;;
esac
