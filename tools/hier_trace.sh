#!/bin/bash
# Tuning probe: per-dispatch durations of one HighLevelEnv macro step (rocprofv3 kernel trace, eager launches).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/hier; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/tr -o tr -- python $R/bench.py --workload hier --arenas ${ARENAS:-8192} --steps 6 --warmup 2 --no-graph --no-cpu-baseline > $OUT/log.txt 2>&1
python - <<PY
import sqlite3
db = sqlite3.connect("$OUT/tr/tr_results.db")
rows = list(db.execute("select name, start, end from kernels order by start"))
t_last = max(i for i, r in enumerate(rows) if "hh_k_hier" in r[0])
# walk back over the last macro step: 34 hier launches
idx = [i for i, r in enumerate(rows) if "hh_k_hier" in r[0]][-34:]
first = idx[0]
for i in range(first, t_last + 1):
    n, s, e = rows[i]
    print(f"{(s - rows[first][1]) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n[:70]}")
tot = sum(rows[i][2] - rows[i][1] for i in idx)
print("hier kernel time in the step: %.1f us of %.1f us wall" % (tot / 1e3, (rows[t_last][2] - rows[first][1]) / 1e3))
PY
