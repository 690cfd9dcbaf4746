#!/bin/bash
# throughput of the default bench under runtime knobs: usage knob_sweep.sh
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 | streams=$2 fuse=$3: "; env $1 timeout 300 python3 bench.py --steps 64 --warmup 16 --streams $2 --fuse $3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'img/s')" 2>&1 | tail -1; }
run GPU_MAX_HW_QUEUES=8 4 4
run GPU_MAX_HW_QUEUES=16 8 2
run GPU_MAX_HW_QUEUES=16 6 3
run "GPU_MAX_HW_QUEUES=8 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" 4 4
run "GPU_MAX_HW_QUEUES=8 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" 4 4
run "GPU_MAX_HW_QUEUES=4" 4 4
run "GPU_MAX_HW_QUEUES=8 HIP_FORCE_DEV_KERNARG=1" 4 4
run "GPU_MAX_HW_QUEUES=8 HSA_ENABLE_INTERRUPT=0" 4 4
run "GPU_MAX_HW_QUEUES=8" 4 6
run "GPU_MAX_HW_QUEUES=8" 3 6
run "GPU_MAX_HW_QUEUES=8" 2 8
