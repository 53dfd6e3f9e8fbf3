#!/bin/bash
# one gpurun call: the rocprofv3 evidence of round 4 — headline kernel (kernel-trace stats + PMC passes), the policy kernel forms
# (hh_k_policy_w16 greedy, hh_k_policy_ppo sampler) with their HBM traffic, the networks-in-the-loop commander step trace, the driver-style bench line
R=$GRAFT_REPO_ROOT
TAG=r04_low ARENAS_PER_WAVE=8 bash $R/tools/prof_pmc.sh > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for form in w16 ppo; do
  OUT=$R/gpurun_out/r04_policy_$form; rm -rf $OUT; mkdir -p $OUT
  if [ $form = ppo ]; then CMD="python $R/tools/ppo_bench.py 16384 8"; K=hh_k_policy_w16_ppo; else CMD="python $R/tools/policy_bench.py 32768 0"; K="hh_k_policy_w16<8>"; fi
  rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1
  python $R/tools/rocpd_summary.py --kernel "$K" --min-us 20 $OUT/stats/stats_results.db $OUT/pmc1/pmc1_results.db $OUT/pmc2/pmc2_results.db $OUT/pmc3/pmc3_results.db $OUT/pmc4/pmc4_results.db > $OUT/summary.txt 2>&1
  python $R/tools/rocpd_summary.py --traffic $OUT/pmc3/pmc3_results.db $OUT/pmc4/pmc4_results.db "$K" 16384 1 > $OUT/traffic_raw.json
  python -c "
import json, sys
d = json.load(open('$OUT/traffic_raw.json'))
for k in ('arenas', 'ticks_per_launch', 'hbm_bytes_per_arena_tick'): d.pop(k, None)
d.update(rows=32768, hbm_bytes_per_row=d['hbm_bytes_per_launch'] / 32768, command='$CMD'.split('tools/')[-1].join(['tools/', '']))
json.dump(d, open('$OUT/traffic.json', 'w'))"     # -> profiles/latest_policy_traffic.json (w16) | latest_policy_ppo_traffic.json (ppo): bench.py quotes them
  tail -2 $OUT/stats.log >> $OUT/summary.txt
  find $OUT -name "*.db" -delete
done
bash $R/tools/prof_hier_net.sh > /dev/null 2>&1
cd $R && python bench.py --steps 20 --warmup 5 > $R/gpurun_out/r04_bench_line.json 2> $R/gpurun_out/r04_bench_line.err
for t in r04_low r04_policy_w16 r04_policy_ppo; do echo "#### $t"; head -70 $R/gpurun_out/$t/summary.txt; cat $R/gpurun_out/$t/traffic.json; done
cat $R/gpurun_out/r04_low/pmc.json; head -40 $R/gpurun_out/prof_hier_net/summary.txt; cat $R/gpurun_out/prof_hier_net/gaps.txt | tail -12
cut -c1-400 $R/gpurun_out/r04_bench_line.json
