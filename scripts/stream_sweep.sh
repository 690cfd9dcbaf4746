#!/bin/bash
# throughput vs batches in flight (and hardware queues): usage stream_sweep.sh
cd $GRAFT_REPO_ROOT
for q in 8; do
for s in 1 2 3 4 5 6; do
  echo -n "queues=$q streams=$s: "
  GPU_MAX_HW_QUEUES=$q timeout 200 python3 bench.py --steps 24 --warmup $s --streams $s --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'img/s', round(d['ms_per_refine_iter']*1000,1), 'us/iter')"
done
done
