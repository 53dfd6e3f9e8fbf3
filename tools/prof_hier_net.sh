#!/bin/bash
# rocprofv3 kernel-trace statistics of the HighLevelEnv bench with the pilot networks in the loop, as the driver's line runs it (variant rows, 2 sub-worlds on 2 streams,
# one HIP graph) and, for the A/B, the two-calls-per-sub-step form on 4 streams
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_hier_net; rm -rf $OUT; mkdir -p $OUT
for mode in variants sides; do
  rocprofv3 --kernel-trace --stats -d $OUT/$mode -o stats -- python $R/bench.py --workload hier --pilot net --pilot-rows $mode --arenas ${ARENAS:-8192} --steps 12 --warmup 3 --no-cpu-baseline > $OUT/$mode.log 2>&1
  echo "== bench.py --workload hier --pilot net --pilot-rows $mode: $(grep '^{"metric' $OUT/$mode.log | tail -1 | python3 -c 'import sys,json; l=json.loads(sys.stdin.readline()); print("%.3e commander-steps/s, %.3f ms per commander step of %s arenas, %d streams, %d launches per step" % (l["value"], l["ms_per_step"], l["config"]["arenas_per_gpu"], l["streams"], l["launches_per_step"]))')" >> $OUT/summary.txt
  python $R/tools/rocpd_summary.py $OUT/$mode/stats_results.db 2>&1 | head -7 >> $OUT/summary.txt
  python - <<PY >> $OUT/summary.txt
import sqlite3
db = sqlite3.connect("$OUT/$mode/stats_results.db")
rows = list(db.execute("select name, start, end from kernels order by start"))
rows = rows[len(rows) // 3:]          # past the spin-up
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
pol = sorted(e - s for nm, s, e in rows if "hh_k_policy" in nm)
ph = sorted(e - s for nm, s, e in rows if "hh_k_hier_oct" in nm)
print(f"  last two thirds of the run: {len(rows)} kernels, span {span / 1e6:.2f} ms, sum of kernel durations {busy / 1e6:.2f} ms = {busy / span:.2f} kernels in flight on average")
print(f"  policy calls: n={len(pol)} p10 {pol[len(pol) // 10] / 1e3:.1f} median {pol[len(pol) // 2] / 1e3:.1f} p90 {pol[9 * len(pol) // 10] / 1e3:.1f} us;  world phases: n={len(ph)} p10 {ph[len(ph) // 10] / 1e3:.1f} median {ph[len(ph) // 2] / 1e3:.1f} p90 {ph[9 * len(ph) // 10] / 1e3:.1f} us")
PY
  find $OUT/$mode -name "*.db" -delete
done
cat $OUT/summary.txt
