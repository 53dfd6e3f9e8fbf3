#!/bin/bash
# A/B of the policy kernel's instances in one gpurun call: LIBS="a.so b.so" TILES="32 64 328" bash tools/ab_policy.sh
for lib in ${LIBS:-""}; do
  echo "== lib=$lib"
  HH_WORLD_LIB=$lib python tools/policy_bench.py 32768 ${TILES:-32 64 328}
  HH_WORLD_LIB=$lib python tools/policy_bench.py 16384 ${TILES:-32 64 328}
  HH_WORLD_LIB=$lib python tools/policy_bench.py 8192 ${TILES:-32 64 328}
done
