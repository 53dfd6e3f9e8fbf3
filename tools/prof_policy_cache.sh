#!/bin/bash
# cache counters of the policy kernel alone (tools/policy_bench.py): how much of the weight stream the per-CU L1 absorbs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_policy_cache; rm -rf $OUT; mkdir -p $OUT
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -d $OUT/$n -o p -- python $R/tools/policy_bench.py ${ROWS:-32768} 32 > $OUT/$n.log 2>&1
  python $R/tools/rocpd_summary.py --kernel hh_k_policy_h $OUT/$n/p_results.db 2>&1 | grep -v "^==" 
done
find $OUT -name "*.db" -delete
