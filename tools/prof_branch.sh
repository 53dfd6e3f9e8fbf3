#!/bin/bash
# one PMC pass: total / branch / skipped instruction counts of the bench kernel (env: ARENAS, CHUNK, KERNEL, LIB)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/branch; rm -rf $OUT; mkdir -p $OUT
A=${ARENAS:-4096}; C=${CHUNK:-250}; K=${KERNEL:-hh_k_world_quad}
[ -n "$LIB" ] && export HH_WORLD_LIB=$R/$LIB
rocprofv3 --pmc SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VSKIPPED SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES -d $OUT/p -o p -- python $R/bench.py --steps 8 --warmup 2 --spinup 0.3 --no-cpu-baseline --no-extra --arenas $A --chunk $C ${BENCH_ARGS} > $OUT/log 2>&1
python $R/tools/rocpd_summary.py --kernel $K --min-us ${MIN_US:-500} $OUT/p/p_results.db
find $OUT -name "*.db" -delete
