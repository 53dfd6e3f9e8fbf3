#!/bin/bash
# rocprofv3 passes for the bench kernel: kernel-trace stats, then PMC counter passes (separate runs: gpurun refuses --pmc
# together with trace domains other than --kernel-trace, and the TCC counters cannot share a pass).
# env: ARENAS (4096), CHUNK (500), BENCH_ARGS, KERNEL (hh_k_world_quad), TAG (name under gpurun_out/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG:-prof}
rm -rf $OUT; mkdir -p $OUT
A=${ARENAS:-4096}; C=${CHUNK:-500}; K=${KERNEL:-hh_k_world_quad}
ARGS="--steps 8 --warmup 2 --spinup 0.3 --no-cpu-baseline --no-extra --arenas $A --chunk $C ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc1 -- python $R/bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- python $R/bench.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU -d $OUT/pmc5 -o pmc5 -- python $R/bench.py $ARGS > $OUT/pmc5.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python $R/bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc4 -- python $R/bench.py $ARGS > $OUT/pmc4.log 2>&1
MINUS=${MIN_US:-500}
python $R/tools/rocpd_summary.py --kernel $K --min-us $MINUS $OUT/stats/stats_results.db $OUT/pmc1/pmc1_results.db $OUT/pmc2/pmc2_results.db $OUT/pmc5/pmc5_results.db $OUT/pmc3/pmc3_results.db $OUT/pmc4/pmc4_results.db > $OUT/summary.txt 2>&1
python $R/tools/rocpd_summary.py --traffic $OUT/pmc3/pmc3_results.db $OUT/pmc4/pmc4_results.db $K $A $C > $OUT/traffic.json
python $R/tools/rocpd_summary.py --pmcjson $OUT/pmc1/pmc1_results.db $K $A $C ${ARENAS_PER_WAVE:-16} $OUT/pmc5/pmc5_results.db $OUT/pmc2/pmc2_results.db > $OUT/pmc.json
tail -1 $OUT/stats.log > $OUT/bench_line.json
cat $OUT/summary.txt $OUT/traffic.json $OUT/pmc.json
# the rocpd databases are tens of MB each and gpurun_out/ is capped at 64 MiB: keep the text summaries only
find $OUT -name "*.db" -delete
