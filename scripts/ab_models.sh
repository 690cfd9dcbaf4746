#!/bin/bash
# Baseline workload (encoder time) per library build: usage ab_models.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in "$@"; do
  export IHMR_HIP_LIBRARY=$(realpath $lib); echo -n "$lib run $r: "
  timeout 300 python3 scripts/bench_models.py baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['encoder_ms_per_batch'],3), 'ms encoder,', round(d['images_per_s']), 'img/s')"
done; done
