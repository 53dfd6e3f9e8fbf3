cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
{
echo "== hier nets sweep (variants)"
for c in "variants - 2" "variants - 3" "variants - 1" "variants 2 2" "variants 2 3" "variants - 4"; do bash tools/variants_rates2.sh $c; done
echo "== 3v3 tape macro step: product vs ablations"
for lib in "" hhmarl_2d_amd/lib/abl_noevt.so hhmarl_2d_amd/lib/abl_noendtab.so; do
  HH_WORLD_LIB=$lib python bench.py --workload hier --steps 60 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hier tape lib=$lib', '%.4g'%d['value'], round(d['gpu_ms_per_step']*1e3,2), 'us per commander step')"
done
echo "== policy epilogue ablation (no tanh / split behind the shared layer's column groups: wrong results, timing only)"
for lib in "" hhmarl_2d_amd/lib/abl_noepi.so; do
  for W in 2 3; do echo "lib=$lib W=$W: $(HH_WORLD_LIB=$lib HH_POLICY_W=$W python tools/policy_bench.py 32768 0 2>&1 | tail -1)"; done
  echo "lib=$lib ppo: $(HH_WORLD_LIB=$lib python tools/ppo_bench.py 16384 16 2>&1 | tail -3 | tr '\n' ' ')"
done
} > gpurun_out/r6b/exp1.log 2>&1
cat gpurun_out/r6b/exp1.log
