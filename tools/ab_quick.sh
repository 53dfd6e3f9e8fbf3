#!/bin/bash
# A/B of library builds on ONE box, the workloads the 2-vs-2 kernel serves:  LIBS="base.so new.so" bash tools/ab_quick.sh   ("product" = the default build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in ${LIBS:-product}; do
  L=$lib; [ "$lib" = "product" ] && L=""
  for a in ${ARENAS:-4096 8192 262144}; do
    ch=500; [ $a -gt 100000 ] && ch=125
    HH_WORLD_LIB=$L python bench.py --arenas $a --chunk $ch --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib low', $a, '%.4g'%d['value'], round(d['roofline']['avg_launch_ms'],4), 'ms/launch')"
  done
done
done
