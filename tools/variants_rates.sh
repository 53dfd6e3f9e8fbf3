cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/var
for cfg in "variants 1" "variants 2" "variants 4" "variants 8" "sides 4" "sides 2"; do
  set -- $cfg
  python bench.py --workload hier --pilot net --pilot-rows $1 --streams $2 --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.readline()); print('$1 K=$2', '%.3e'%l['value'], 'ms', round(l['ms_per_step'],3), 'gpu_ms', round(l['gpu_ms_per_step'],3), 'launches', l['launches_per_step'])" 2>&1 | tee -a gpurun_out/var/rates.txt
done
