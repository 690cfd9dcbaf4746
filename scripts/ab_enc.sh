#!/bin/bash
# A/B of library builds on the Baseline config: usage scripts/ab_enc.sh lib1.so ... ("product" = in-tree)
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" = "product" ]; then unset IHMR_HIP_LIBRARY; else export IHMR_HIP_LIBRARY=$(realpath $lib); fi
  echo "== $lib"
  python3 bench.py --config baseline --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('baseline', round(d['value']), 'img/s; encoder', round(d['roofline']['encoder_ms_per_batch'],3), 'ms =', round(d['roofline']['achieved'],1), 'TFLOP/s; two in flight', round(d['two_batches_in_flight']['images_per_s']))"
done
