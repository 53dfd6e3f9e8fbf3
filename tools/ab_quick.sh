#!/bin/bash
# quick A/B of library builds on one box: LIBS="product hhmarl_2d_amd/lib/abl_x.so" bash tools/ab_quick.sh   (2-vs-2 at 4096 / 8192 / 262144 arenas, 3-vs-3 tape at 8192 / 65536)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in ${LIBS:-product}; do
  L=$lib; [ "$lib" = "product" ] && L=""
  for a in 4096 8192 262144; do HH_WORLD_LIB=${L:+$PWD/$L} python bench.py --arenas $a --chunk $([ $a = 262144 ] && echo 125 || echo 500) --steps 12 --warmup 3 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib low', $a, '%.4g'%d['value'], round(d['ms_per_step'],4), 'ms/launch')"; done
  HH_WORLD_LIB=${L:+$PWD/$L} python bench.py --workload hier --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib hier tape 8192', '%.4g'%d['value'], round(d['ms_per_step'],4))"
  HH_WORLD_LIB=${L:+$PWD/$L} python bench.py --workload hier --arenas 65536 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=$lib hier tape 65536', '%.4g'%d['value'], round(d['ms_per_step'],4))"
done; done
