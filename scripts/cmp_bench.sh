#!/bin/bash
# A/B of whole-loop throughput (bench default: 2 streams) for experiment libraries: scripts/cmp_bench.sh lib1.so lib2.so ... (alternating twice)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "$@"; do
  export IHMR_HIP_LIBRARY=$(realpath $lib)
  v=$(python3 bench.py --no-extras --no-cpu-baseline --no-work-counters 2>/dev/null | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['value']))")
  echo "$lib: $v images/s"
done
done
