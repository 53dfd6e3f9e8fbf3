#!/bin/bash
# A/B of the commander step with the networks in the loop: bash tools/nets_ab.sh "K..." "ENV=VAL ..." (each env setting a separate run), 40 timed steps each
cd $GRAFT_REPO_ROOT
for K in $1; do for E in "${@:2}"; do
  env $E python bench.py --workload hier --pilot net --streams $K --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{"metric' | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); print('K=$K $E', '%.3e'%l['value'], round(l['ms_per_step'],3), 'ms')"
done; done
