#!/bin/bash
# one short run of bench.py --workload hier --pilot net:  bash tools/variants_rates2.sh <variants|sides> <HH_POLICY_W or -> <streams> [arenas]
cd $GRAFT_REPO_ROOT
[ "$2" != "-" ] && export HH_POLICY_W=$2
python bench.py --workload hier --pilot net --pilot-rows $1 --streams $3 --arenas ${4:-8192} --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); print('$1 W=$2 K=$3 N=${4:-8192}', '%.3e'%l['value'], 'ms', round(l['ms_per_step'],3))"
