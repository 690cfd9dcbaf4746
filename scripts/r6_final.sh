#!/bin/bash
# round 6, final measurement series of the committed code: profiles (three launch sizes + the two secondary configs), the gap tables
# of one batch of 64, the microbenchmark of the launch boundary, and the full bench line.
# usage (GPU box): bash scripts/r6_final.sh <tag>      e.g. r6_v4
tag=${1:-r6_v4}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out build
# the instrumented library of the gap table (every workgroup records kind / start / end) and the trivial-kernel chain, built from this tree
[ build/timeline.so -nt ihmr_amd/libihmr_hip.so ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -Iinclude -DIHMR_TIMELINE \
    ihmr_amd/csrc/ihmr_hip.hip -o build/timeline.so > gpurun_out/${tag}_build.log 2>&1
[ -x scripts/microbench_gaps ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench_gaps.hip -o scripts/microbench_gaps >> gpurun_out/${tag}_build.log 2>&1
for f in 7 1 8; do FUSE=$f bash scripts/profile_round.sh $tag > gpurun_out/${tag}_f${f}_profile.log 2>&1; done
bash scripts/profile_config.sh $tag baseline > gpurun_out/${tag}_baseline_profile.log 2>&1
bash scripts/profile_config.sh $tag mlp > gpurun_out/${tag}_mlp_profile.log 2>&1
SB="python3 bench.py --steps 1 --warmup 1 --streams 1 --fuse 1 --no-cpu-baseline --no-extras --no-work-counters"
IHMR_HIP_LIBRARY=$PWD/build/timeline.so IHMR_TIMELINE_OUT=/tmp/tl.npy timeout 300 $SB > /dev/null 2> gpurun_out/${tag}_tl.err
python3 scripts/gap_timeline.py /tmp/tl.npy gpurun_out/${tag}_f1_gaps.csv > /dev/null 2>&1
rm -rf gpurun_out/kt; timeout 400 rocprofv3 --kernel-trace -d gpurun_out/kt -o kt -- $SB > /dev/null 2>&1
python3 scripts/gap_table.py gpurun_out/kt/kt_results.db gpurun_out/${tag}_f1_trace_gaps.csv > /dev/null 2>&1
rm -rf gpurun_out/kt
timeout 300 ./scripts/microbench_gaps > gpurun_out/${tag}_microbench_gaps.txt 2>&1
timeout 300 python3 scripts/latency.py 64 9 > gpurun_out/${tag}_latency.txt 2>/dev/null
timeout 1500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
timeout 600 python3 bench.py --gpus 2 --steps 8 --warmup 4 > gpurun_out/${tag}_bench_gpus2.json 2> gpurun_out/${tag}_bench_gpus2.err
cat gpurun_out/${tag}_latency.txt; cut -c1-400 gpurun_out/${tag}_bench_line.json; ls gpurun_out | grep "^$tag" | wc -l
