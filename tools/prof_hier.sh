#!/bin/bash
# rocprofv3 kernel-trace statistics of the HighLevelEnv bench (eager launches so that every phase launch is traced).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_hier; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --workload hier --arenas ${ARENAS:-8192} --steps 60 --warmup 10 --no-graph --no-cpu-baseline > $OUT/stats.log 2>&1
python $R/tools/rocpd_summary.py $OUT/stats/stats_results.db > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -20; tail -1 $OUT/stats.log | cut -c1-200
