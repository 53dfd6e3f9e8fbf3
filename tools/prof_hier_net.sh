#!/bin/bash
# rocprofv3 kernel-trace statistics of the HighLevelEnv bench with the pilot networks in the loop (eager launches: every launch traced)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_hier_net; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --workload hier --pilot net --arenas ${ARENAS:-8192} --steps 12 --warmup 3 --no-graph --no-cpu-baseline > $OUT/stats.log 2>&1
python $R/tools/rocpd_summary.py $OUT/stats/stats_results.db > $OUT/summary.txt 2>&1
python - <<PY > $OUT/gaps.txt
import sqlite3
db = sqlite3.connect("$OUT/stats/stats_results.db")
rows = list(db.execute("select name, start, end from kernels order by start"))
rows = rows[len(rows)//3:]          # past the spin-up
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
gaps = [rows[i+1][1] - rows[i][2] for i in range(len(rows) - 1)]
import statistics
# durations by phase: ... P H(act) P H(tick) ... P H(tick 16) H(end) [..] H(begin) P ...
H = [(i, e - s) for i, (nm, s, e) in enumerate(rows) if "hh_k_hier" in nm]
names = [nm for nm, _, _ in rows]
ph = {"begin": [], "act": [], "tick": [], "end": []}
state = None
for k, (i, d) in enumerate(H):
    prev_is_h = k > 0 and all("hh_k_policy" not in names[j] for j in range(H[k-1][0] + 1, i))
    if prev_is_h and state == "tick": state = "end"
    elif prev_is_h and state == "end": state = "begin"
    elif state == "begin": state = "act"
    elif state == "act": state = "tick"
    elif state == "tick": state = "act"
    else: state = None if not prev_is_h else state
    if state is None and prev_is_h: state = "end"
    if state: ph[state].append(d)
for k, v in ph.items():
    if v: print(f"  hier phase {k:6s} n={len(v):5d} avg {sum(v)/len(v)/1e3:7.2f} us  min {min(v)/1e3:7.2f}  max {max(v)/1e3:7.2f}")
P = sorted(e - s for nm, s, e in rows if "hh_k_policy_h" in nm)
print(f"  policy: n={len(P)} p10 {P[len(P)//10]/1e3:.1f} median {P[len(P)//2]/1e3:.1f} p90 {P[9*len(P)//10]/1e3:.1f} us")
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %)  median gap {statistics.median(gaps)/1e3:.2f} us  mean gap {sum(gaps)/len(gaps)/1e3:.2f} us")
PY
cat $OUT/summary.txt | head -12; cat $OUT/gaps.txt; tail -1 $OUT/stats.log | cut -c1-200
find $OUT -name "*.db" -delete
