#!/bin/bash
# batches in flight: streams x fuse sweep of the bench's timed region (steps = streams x fuse x 1 so that every stream runs one sequence)
cd $GRAFT_REPO_ROOT
for cfg in "2 10" "2 8" "3 8" "2 16" "3 10" "4 6" "1 20" "2 20"; do
  set -- $cfg
  python3 bench.py --steps $(( $1 * $2 )) --warmup $(( $1 * $2 )) --streams $1 --fuse $2 --no-cpu-baseline --no-extras --no-work-counters 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('streams $1 x fuse $2:', round(d['value']), 'images/s')"
done
