#!/bin/bash
# round 6, GPU job 1: where does the single-batch iteration's time go OUTSIDE the workgroups?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r6_job1.log
: > $O
SB="python3 bench.py --steps 1 --warmup 1 --streams 1 --fuse 1 --no-cpu-baseline --no-extras --no-work-counters"
echo "== microbench_gaps (wall)" >> $O
timeout 300 ./scripts/microbench_gaps >> $O 2>&1
echo "== microbench_gaps under the kernel trace" >> $O
rm -rf gpurun_out/mg; timeout 300 rocprofv3 --kernel-trace -d gpurun_out/mg -o mg -- ./scripts/microbench_gaps > gpurun_out/r6_mg_traced.log 2>&1
python3 scripts/gap_table.py gpurun_out/mg/mg_results.db gpurun_out/r6_microbench_gaps_trace.csv >> $O 2>&1
rm -rf gpurun_out/mg
echo "== single-batch latency, product" >> $O
timeout 300 python3 scripts/latency.py 64 9 >> $O 2>&1
for knob in "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "GPU_MAX_HW_QUEUES=1" "HSA_ENABLE_INTERRUPT=0"; do
  echo "== single-batch latency, $knob" >> $O
  env $knob timeout 300 python3 scripts/latency.py 64 7 >> $O 2>&1
done
echo "== timeline of one batch (in-kernel clocks)" >> $O
IHMR_HIP_LIBRARY=$PWD/build/timeline.so IHMR_TIMELINE_OUT=/tmp/tl.npy timeout 300 $SB > gpurun_out/r6_tl_bench.json 2>> $O
python3 scripts/gap_timeline.py /tmp/tl.npy gpurun_out/r6_f1_gaps.csv >> $O 2>&1
echo "== kernel trace of one batch: the trace's own gaps" >> $O
rm -rf gpurun_out/kt; timeout 400 rocprofv3 --kernel-trace -d gpurun_out/kt -o kt -- $SB > gpurun_out/r6_kt.log 2>&1
python3 scripts/gap_table.py gpurun_out/kt/kt_results.db gpurun_out/r6_f1_trace_gaps.csv >> $O 2>&1
rm -rf gpurun_out/kt
echo "== new GPU tests" >> $O
timeout 1500 python3 -m pytest tests/test_gpu_multirank.py -x -q -k "without_a_launcher or latency_figures" >> $O 2>&1
timeout 900 python3 -m pytest tests/test_gpu_models.py -x -q -k "sync_export or baseline_model_matches_oracle" >> $O 2>&1
tail -5 $O
