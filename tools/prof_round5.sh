#!/bin/bash
# one gpurun call: the rocprofv3 evidence of round 5 (profiles/r05_*)
#   r05_low   headline kernel hh_k_world_quad<1, 1, true, 8, true> at 4096 arenas: kernel-trace stats + PMC passes (SQ mix, F64 classes, FETCH / WRITE)
#   r05_sat   the saturated configuration: hh_k_world_quad<2, 1, false, 16, false> at 262144 arenas (bench.py extra.configs1_saturated)
#   r05_hier  hh_k_hier_macro_oct (configs[3], tape) refreshed
#   prof_hier_net   kernel trace of the commander step with the pilot networks in the loop
#   r05_bench_line.json   the driver's command on this build
R=$GRAFT_REPO_ROOT
TAG=r05_low ARENAS=4096 CHUNK=500 ARENAS_PER_WAVE=8 bash $R/tools/prof_pmc.sh > /dev/null 2>&1
TAG=r05_sat ARENAS=262144 CHUNK=125 ARENAS_PER_WAVE=16 MIN_US=2000 bash $R/tools/prof_pmc.sh > /dev/null 2>&1
TAG=r05_hier ARENAS=8192 CHUNK=1 KERNEL=hh_k_hier_macro_oct BENCH_ARGS="--workload hier --steps 40" ARENAS_PER_WAVE=8 MIN_US=50 bash $R/tools/prof_pmc.sh > /dev/null 2>&1
bash $R/tools/prof_hier_net.sh > /dev/null 2>&1
cd $R && python bench.py --steps 20 --warmup 5 > $R/gpurun_out/r05_bench_line.json 2> $R/gpurun_out/r05_bench_line.err
python tools/vector_env_rates.py $R/gpurun_out/r05_vector_env_rates.json > /dev/null 2>&1
for t in r05_low r05_sat r05_hier; do echo "#### $t"; head -40 $R/gpurun_out/$t/summary.txt; cat $R/gpurun_out/$t/traffic.json $R/gpurun_out/$t/pmc.json; done
head -14 $R/gpurun_out/prof_hier_net/summary.txt; cat $R/gpurun_out/prof_hier_net/gaps.txt | tail -8
cut -c1-300 $R/gpurun_out/r05_bench_line.json
