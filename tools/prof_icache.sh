#!/bin/bash
# round 6: does the instruction cache matter?  SQC_ICACHE_* and SQ_IFETCH of the persistent kernels, hinted (product) against unhinted layout (abl_norare.so)
#   one PMC pass per library and workload (rocprofv3 --pmc only), max over dispatches of each counter per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/icache; rm -rf $OUT; mkdir -p $OUT
run() { # tag lib args...
  tag=$1; lib=$2; shift; shift
  HH_WORLD_LIB=${lib:+$R/$lib} rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/$tag -o p -- python $R/bench.py --no-cpu-baseline --no-extra --spinup 0.3 "$@" > $OUT/$tag.log 2>&1
  python - $OUT/$tag/p_results.db $tag <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select kernel_name, counter_name, max(value), count(*) from counters_collection group by kernel_name, counter_name"))
best = {}
for k, c, v, n in rows:
    best.setdefault(k, {})[c] = v
for k, d in best.items():
    if d.get("SQ_INSTS_VALU", 0) < 1e7: continue
    print(sys.argv[2], k[:48], " ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())), "miss_rate=%.4f" % (d.get("SQC_ICACHE_MISSES", 0) / max(1.0, d.get("SQC_ICACHE_REQ", 1))))
PY
  find $OUT/$tag -name "*.db" -delete
}
for lib in ${LIBS_IC:-product}; do
  [ "$lib" = "product" ] && lib=""
  n=$( [ -z "$lib" ] && echo product || basename $lib .so )
  run q4096_$n   "$lib" --arenas 4096 --chunk 500 --steps 6 --warmup 2
  run q262144_$n "$lib" --arenas 262144 --chunk 125 --steps 6 --warmup 2
  run h8192_$n   "$lib" --workload hier --arenas 8192 --steps 30 --warmup 5
  run h65536_$n  "$lib" --workload hier --arenas 65536 --steps 12 --warmup 3
  run nets_$n    "$lib" --workload hier --pilot net --arenas 8192 --steps 8 --warmup 2
done 2>&1 | tee $OUT/summary.txt
