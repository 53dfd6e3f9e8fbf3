#!/bin/bash
# rocprofv3 kernel-trace statistics of `bench.py --workload hier --pilot net` in the variant-row form (eager launches), per stream count
#   env: KS (default "1 2 4"), ARENAS (8192)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_hier_var; rm -rf $OUT; mkdir -p $OUT
for K in ${KS:-1 2 4}; do
  rocprofv3 --kernel-trace --stats -d $OUT/k$K -o stats -- python $R/bench.py --workload hier --pilot net --pilot-rows ${ROWS:-variants} --streams $K --arenas ${ARENAS:-8192} --steps 12 --warmup 3 --no-graph --no-cpu-baseline > $OUT/k$K.log 2>&1
  echo "== streams $K: $(tail -1 $OUT/k$K.log | cut -c1-120)" >> $OUT/summary.txt
  python $R/tools/rocpd_summary.py $OUT/k$K/stats_results.db 2>&1 | head -8 >> $OUT/summary.txt
  find $OUT/k$K -name "*.db" -delete
done
cat $OUT/summary.txt
