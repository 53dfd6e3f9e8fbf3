#!/bin/bash
# A/B of alternative builds of libhh_world.so (compiler flags, experiments) in ONE gpurun call, so that both sides see the same box:
#   LIBS="hhmarl_2d_amd/lib/ab_x.so hhmarl_2d_amd/lib/ab_y.so" bash tools/ab_lib.sh      ("" = the default build)
for lib in ${LIBS:-""}; do
  echo "== lib=$lib"
  for a in 4096 16384 262144; do HH_WORLD_LIB=$lib python bench.py --arenas $a --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('low', $a, '%.4g'%d['value'], d['ms_per_step'])"; done
  HH_WORLD_LIB=$lib python bench.py --workload hier --steps 50 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hier tape', '%.4g'%d['value'], d['ms_per_step'])"
  [ -n "$QUICK" ] && continue
  HH_WORLD_LIB=$lib python bench.py --workload rollout --steps 200 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rollout', '%.4g'%d['value'], d['ms_per_step'])"
  HH_WORLD_LIB=$lib python tools/policy_bench.py 32768 32
done
