#!/bin/bash
# one short run of `bench.py --workload hier --pilot net` (the commander step with the pilot networks in the loop):
#   bash tools/variants_rates2.sh <variants|sides> <HH_POLICY_W or -> <streams> [arenas] [extra bench flags...]
# e.g. through gpurun:  for c in "variants - 1" "variants - 2" "variants - 4" "sides - 4"; do bash tools/variants_rates2.sh $c; done
cd $GRAFT_REPO_ROOT
ROWS=$1; W=$2; K=$3; N=${4:-8192}; shift; shift; shift; [ $# -gt 0 ] && shift
[ "$W" != "-" ] && export HH_POLICY_W=$W
python bench.py --workload hier --pilot net --pilot-rows $ROWS --streams $K --arenas $N --steps 12 --warmup 3 --no-cpu-baseline "$@" 2>&1 | grep '^{"metric' | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); print('$ROWS W=$W K=$K N=$N $*', '%.3e'%l['value'], 'commander-steps/s,', round(l['ms_per_step'],3), 'ms per commander step')"
