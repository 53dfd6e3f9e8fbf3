#!/bin/bash
# rocprofv3 passes for the bench kernel: kernel-trace stats, then PMC counter passes (separate runs).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 1000 --warmup 250 --no-cpu-baseline --arenas ${ARENAS:-4096} ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc1 -- python $R/bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- python $R/bench.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python $R/bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python $R/bench.py $ARGS > $OUT/pmc4.log 2>&1
python $R/tools/rocpd_summary.py --kernel ${KERNEL:-hh_k_world_quad} --min-us 1000 $OUT/stats/stats_results.db $OUT/pmc1/pmc1_results.db $OUT/pmc2/pmc2_results.db $OUT/pmc3/pmc3_results.db $OUT/pmc4/pmc4_results.db > $OUT/summary.txt 2>&1
python $R/tools/rocpd_summary.py --traffic $OUT/pmc3/pmc3_results.db $OUT/pmc4/pmc4_results.db ${KERNEL:-hh_k_world_quad} ${ARENAS:-4096} 250 > $OUT/traffic.json
cat $OUT/summary.txt $OUT/traffic.json
