for lib in $LIBS; do
  echo "== lib=$lib"
  HH_WORLD_LIB=$lib python bench.py --workload hier --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hier tape', '%.4g'%d['value'], d['ms_per_step'])"
  HH_WORLD_LIB=$lib python bench.py --workload hier --pilot net --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hier nets', '%.4g'%d['value'], d['ms_per_step'])"
  HH_WORLD_LIB=$lib python bench.py --workload hier --arenas 65536 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hier tape 65536', '%.4g'%d['value'], d['ms_per_step'])"
done
