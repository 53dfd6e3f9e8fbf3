#!/bin/bash
# PMC counters of the phase kernel inside the networks-in-the-loop commander step (eager launches)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_hier_net_pmc; rm -rf $OUT; mkdir -p $OUT
ARGS="--workload hier --pilot net --arenas ${ARENAS:-8192} --steps 6 --warmup 2 --spinup 0.2 --no-graph --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc1 -- python $R/bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH_LEVEL -d $OUT/pmc2 -o pmc2 -- python $R/bench.py $ARGS > $OUT/pmc2.log 2>&1
python $R/tools/rocpd_summary.py --kernel hh_k_hier_oct $OUT/pmc1/pmc1_results.db $OUT/pmc2/pmc2_results.db > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.db" -delete
