#!/bin/bash
# default bench.py, 3 runs per library build: usage ab_bench.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for lib in "$@"; do
  export IHMR_HIP_LIBRARY=$(realpath $lib); echo -n "$lib run $r: "
  timeout 200 python3 bench.py --no-cpu-baseline --no-extras --steps 64 --warmup 16 2>&1 | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['value']))"
done; done
