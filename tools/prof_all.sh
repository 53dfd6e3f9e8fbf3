#!/bin/bash
# one gpurun call: the rocprofv3 evidence of the round (kernel-trace stats + separate PMC passes) for the three kernels
#   hh_k_world_quad (bench default, configs[1]), hh_k_hier_macro (configs[3], tape), hh_k_policy (configs[2])
R=$GRAFT_REPO_ROOT
TAG=prof_low ARENAS_PER_WAVE=8 bash $R/tools/prof_pmc.sh   # 4096 arenas run the 8-arenas-per-wave form > /dev/null 2>&1
TAG=prof_hier ARENAS=8192 KERNEL=hh_k_hier_macro_oct BENCH_ARGS="--workload hier" ARENAS_PER_WAVE=8 MIN_US=50 bash $R/tools/prof_pmc.sh > /dev/null 2>&1
TAG=prof_policy ARENAS=16384 KERNEL=hh_k_policy_h BENCH_ARGS="--workload rollout --steps 100" ARENAS_PER_WAVE=16 MIN_US=50 bash $R/tools/prof_pmc.sh > /dev/null 2>&1
for t in prof_low prof_hier prof_policy; do echo "#### $t"; cat $R/gpurun_out/$t/summary.txt | head -60; cat $R/gpurun_out/$t/traffic.json $R/gpurun_out/$t/pmc.json; grep "^{" $R/gpurun_out/$t/stats.log | tail -1 | cut -c1-300; done
