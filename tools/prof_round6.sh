#!/bin/bash
# one gpurun call: the rocprofv3 evidence of round 6 (profiles/r06_*)
#   r06_low   headline kernel hh_k_world_quad<1, 1, true, 8, true, true> at 4096 arenas: kernel-trace stats + PMC passes (SQ mix, F64 classes, FETCH / WRITE)
#   r06_sat   the saturated configuration: hh_k_world_quad<2, 1, false, 16, false, true> at 262144 arenas (bench.py extra.configs1_saturated)
#   r06_hier  hh_k_hier_macro_oct (configs[3], tape)
#   r06_nets  kernel trace of the commander step with the pilot networks in the loop (four sub-worlds, one HIP graph each) + the workgroup timeline of the same step
#   r06_bench_line.json   the driver's command on this build
R=$GRAFT_REPO_ROOT
TAG=r06_low ARENAS=4096 CHUNK=500 ARENAS_PER_WAVE=8 bash $R/tools/prof_pmc.sh > /dev/null 2>&1
TAG=r06_sat ARENAS=262144 CHUNK=125 ARENAS_PER_WAVE=16 MIN_US=2000 bash $R/tools/prof_pmc.sh > /dev/null 2>&1
TAG=r06_hier ARENAS=8192 CHUNK=1 KERNEL=hh_k_hier_macro_oct BENCH_ARGS="--workload hier --steps 40" ARENAS_PER_WAVE=8 MIN_US=50 bash $R/tools/prof_pmc.sh > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/r06_nets; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o stats -- python $R/bench.py --workload hier --pilot net --steps 12 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1
echo "== bench.py --workload hier --pilot net (under rocprofv3 --kernel-trace --stats: the profiler serialises the streams; unprofiled rate below): $(grep '^{"metric' $OUT/trace.log | tail -1 | python3 -c 'import sys,json; l=json.loads(sys.stdin.readline()); print("%.3e commander-steps/s, %.3f ms per commander step of %s arenas, %d streams, %d launches per step" % (l["value"], l["ms_per_step"], l["config"]["arenas_per_gpu"], l["streams"], l["launches_per_step"]))')" > $OUT/summary.txt
python $R/tools/rocpd_summary.py $OUT/trace/stats_results.db 2>&1 | head -8 >> $OUT/summary.txt
python - <<PY >> $OUT/summary.txt
import sqlite3
db = sqlite3.connect("$OUT/trace/stats_results.db")
rows = list(db.execute("select name, start, end from kernels order by start"))
rows = rows[len(rows) // 3:]
pol = sorted(e - s for nm, s, e in rows if "hh_k_policy" in nm)
ph = sorted(e - s for nm, s, e in rows if "hh_k_hier_oct_v" in nm)
print(f"  policy calls: n={len(pol)} p10 {pol[len(pol) // 10] / 1e3:.1f} median {pol[len(pol) // 2] / 1e3:.1f} p90 {pol[9 * len(pol) // 10] / 1e3:.1f} us;  world phases: n={len(ph)} p10 {ph[len(ph) // 10] / 1e3:.1f} median {ph[len(ph) // 2] / 1e3:.1f} p90 {ph[9 * len(ph) // 10] / 1e3:.1f} us")
PY
find $OUT -name "*.db" -delete
cd $R
for i in 1 2 3; do python bench.py --workload hier --pilot net --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | tail -1 | python3 -c 'import sys,json; l=json.loads(sys.stdin.readline()); print("unprofiled: %.3e commander-steps/s, %.3f ms per commander step, %d streams (%s)" % (l["value"], l["ms_per_step"], l["streams"], l["config"]["parallelism"]))' >> $OUT/summary.txt; done
python bench.py --workload hier --pilot net --joined --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | tail -1 | python3 -c 'import sys,json; l=json.loads(sys.stdin.readline()); print("unprofiled --joined: %.3e commander-steps/s, %.3f ms per commander step, %d streams (%s)" % (l["value"], l["ms_per_step"], l["streams"], l["config"]["parallelism"]))' >> $OUT/summary.txt
python bench.py --workload hier --pilot net --joined --streams 2 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | tail -1 | python3 -c 'import sys,json; l=json.loads(sys.stdin.readline()); print("unprofiled --joined --streams 2 (the round-5 form of the graph): %.3e commander-steps/s, %.3f ms per commander step" % (l["value"], l["ms_per_step"]))' >> $OUT/summary.txt
if [ -f $R/hhmarl_2d_amd/lib/abl_timeline.so ]; then
  HH_WORLD_LIB=$R/hhmarl_2d_amd/lib/abl_timeline.so python tools/timeline.py 8192 4 > $OUT/timeline_k4.txt 2>&1
  HH_TL_JOINED=1 HH_WORLD_LIB=$R/hhmarl_2d_amd/lib/abl_timeline.so python tools/timeline.py 8192 4 > $OUT/timeline_k4_joined.txt 2>&1
fi
python bench.py --steps 20 --warmup 5 > $R/gpurun_out/r06_bench_line.json 2> $R/gpurun_out/r06_bench_line.err
for t in r06_low r06_sat r06_hier; do echo "#### $t"; head -40 $R/gpurun_out/$t/summary.txt; cat $R/gpurun_out/$t/traffic.json $R/gpurun_out/$t/pmc.json; done
cat $OUT/summary.txt
python tools/show_line.py $R/gpurun_out/r06_bench_line.json
