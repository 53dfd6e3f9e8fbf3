#!/bin/bash
# bash tools/variants_rates3.sh <streams> [extra bench flags...]: one short run of the variant-row commander step
cd $GRAFT_REPO_ROOT
K=$1; shift
python bench.py --workload hier --pilot net --pilot-rows variants --streams $K --steps 12 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); print('K=$K $*', '%.3e'%l['value'], 'ms', round(l['ms_per_step'],3))"
