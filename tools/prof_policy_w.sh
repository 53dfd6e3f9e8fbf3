#!/bin/bash
# SQ counters of the policy kernel alone (tools/policy_bench.py): env ROWS (32768), KERNEL (hh_k_policy_w), HH_POLICY_W (1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${TAG:-polw}; rm -rf $OUT; mkdir -p $OUT
export HH_POLICY_W=${HH_POLICY_W:-1}
K=${KERNEL:-hh_k_policy_w}
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/tools/policy_bench.py ${ROWS:-32768} 0 > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc1 -- python $R/tools/policy_bench.py ${ROWS:-32768} 0 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc2 -- python $R/tools/policy_bench.py ${ROWS:-32768} 0 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY -d $OUT/pmc3 -o pmc3 -- python $R/tools/policy_bench.py ${ROWS:-32768} 0 > $OUT/pmc3.log 2>&1
python $R/tools/rocpd_summary.py --kernel $K --min-us 20 $OUT/stats/stats_results.db $OUT/pmc1/pmc1_results.db $OUT/pmc2/pmc2_results.db $OUT/pmc3/pmc3_results.db > $OUT/summary.txt 2>&1
cat $OUT/summary.txt; tail -3 $OUT/pmc2.log $OUT/pmc3.log
find $OUT -name "*.db" -delete
