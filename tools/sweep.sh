#!/bin/bash
# Tuning probe: env-steps/s of the 2-vs-2 rollout over arena counts and kernel variants.
for N in ${SWEEP_N:-4096 16384 32768 65536 262144}; do
  for W in 1 2; do
    for Q in 0 1; do
      v=$(HH_FORCE_W=$W HH_NO_QUAD=$Q python bench.py --arenas $N --steps 8 --warmup 2 --no-cpu-baseline --no-extra | python -c "import json,sys; print('%.1f' % (json.loads(sys.stdin.readline())['value']/1e6))")
      echo "N=$N W=$W no_quad=$Q : $v M env-steps/s"
    done
  done
done
