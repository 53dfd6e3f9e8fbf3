#!/usr/bin/env python
"""Summarise rocprofv3 sqlite outputs (rocpd) into text: per-kernel duration stats and, when
present, per-kernel PMC counter averages.
Usage: rocpd_summary.py [--kernel SUBSTR] [--min-us X] <results.db> [...]
--min-us drops short dispatches (e.g. the reset launch) from duration and counter averages."""
import sqlite3
import sys


def cols(db, t):
    return [r[1] for r in db.execute(f"pragma table_info({t})")]


def main():
    args = sys.argv[1:]
    filt, min_us = None, 0.0
    while args and args[0].startswith("--"):
        if args[0] == "--kernel":
            filt = args[1]
        elif args[0] == "--min-us":
            min_us = float(args[1])
        args = args[2:]
    for path in args:
        db = sqlite3.connect(path)
        print(f"== {path.split('gpurun_out/')[-1]}")
        q = ("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
             f"where (end-start) >= {min_us * 1e3} group by name order by sum(end-start) desc")
        print(f"{'kernel':56s} {'calls':>6s} {'avg_us':>11s} {'min_us':>11s} {'max_us':>11s} {'total_ms':>10s}")
        for n, c, a, mn, mx, s in db.execute(q):
            if filt and filt not in str(n):
                continue
            print(f"{str(n)[:56]:56s} {c:6d} {a/1e3:11.2f} {mn/1e3:11.2f} {mx/1e3:11.2f} {s/1e6:10.3f}")
        pc = cols(db, "counters_collection")
        if pc and "counter_name" in pc:
            dur = "(end-start)" if "end" in pc and "start" in pc else None
            where = f"where {dur} >= {min_us * 1e3}" if dur and min_us else ""
            try:
                rows = list(db.execute(f"select kernel_name, counter_name, count(*), avg(value), max(value) from counters_collection {where} group by kernel_name, counter_name"))
            except sqlite3.Error:
                rows = list(db.execute("select kernel_name, counter_name, count(*), avg(value), max(value) from counters_collection group by kernel_name, counter_name"))
            rows = [r for r in rows if not filt or filt in str(r[0])]
            if rows:
                print(f"{'kernel':40s} {'counter':22s} {'dispatches':>10s} {'avg/dispatch':>18s} {'max/dispatch':>18s}")
                for k, cname, n, a, mx in rows:
                    print(f"{str(k)[:40]:40s} {cname:22s} {n:10d} {a:18.1f} {mx:18.1f}")


def traffic_json(fetch_db, write_db, kernel, min_us, arenas, ticks):
    """HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB; gfx950 FETCH_SIZE counts half of wide reads)"""
    import json
    vals = {}
    for name, path in (("FETCH_SIZE", fetch_db), ("WRITE_SIZE", write_db)):
        db = sqlite3.connect(path)
        rows = list(db.execute("select kernel_name, avg(value), max(value) from counters_collection where counter_name=? group by kernel_name", (name,)))
        rows = sorted((r for r in rows if kernel in str(r[0])), key=lambda r: -r[2])
        vals[name] = rows[0][2] if rows else None   # the dominant matching kernel; max over dispatches = a full-length launch
        if rows:
            vals["full"] = str(rows[0][0])[:60]
    b = None if None in (vals["FETCH_SIZE"], vals["WRITE_SIZE"]) else int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
    print(json.dumps({"arenas": arenas, "ticks_per_launch": ticks, "kernel": kernel, "kernel_full": vals.get("full"), "fetch_size_kb": vals["FETCH_SIZE"],
                      "write_size_kb": vals["WRITE_SIZE"], "hbm_bytes_per_launch": b,
                      "hbm_bytes_per_arena_tick": None if b is None else b / (arenas * ticks),
                      "note": "2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, separate passes"}))


def pmc_json(db_path, kernel, arenas, ticks, arenas_per_wave, f64_db=None, more_dbs=()):
    """instruction counters per wave-tick (one wave-tick = `arenas_per_wave` arenas x 1 tick) for bench.py's fp64 roofline block;
    f64_db: the pass with SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 (the instructions that are FP64 arithmetic)"""
    import json
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select kernel_name, counter_name, max(value) from counters_collection group by kernel_name, counter_name"))
    for extra in ([f64_db] if f64_db else []) + list(more_dbs):
        rows += list(sqlite3.connect(extra).execute("select kernel_name, counter_name, max(value) from counters_collection group by kernel_name, counter_name"))
    rows = [r for r in rows if kernel in str(r[0])]
    if not rows:
        print(json.dumps({"error": "kernel not found"}))
        return
    top = max(rows, key=lambda r: r[2] if r[1] == "SQ_INSTS_VALU" else -1)[0]   # the instance doing the most work
    v = {r[1]: r[2] for r in rows if r[0] == top}
    wt = (arenas + arenas_per_wave - 1) // arenas_per_wave * ticks
    f64 = {k[len("SQ_INSTS_VALU_"):]: v[k] / wt for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64") if k in v}
    print(json.dumps({"arenas": arenas, "ticks_per_launch": ticks, "kernel": str(top)[:60], "insts_f64_per_wave_tick": f64 or None,
                      "insts_valu_per_wave_tick": v.get("SQ_INSTS_VALU", 0) / wt, "insts_salu_per_wave_tick": v.get("SQ_INSTS_SALU", 0) / wt,
                      "insts_lds_per_wave_tick": v.get("SQ_INSTS_LDS", 0) / wt, "sq_waves": v.get("SQ_WAVES"),
                      "wave_cycles_per_wave_tick": v.get("SQ_WAVE_CYCLES", 0) * 4 / wt,
                      # where the waves' cycles go (quad-cycles of ALL waves of the kernel, incl. the output wave of the two-wave form):
                      # issuing an instruction / parked at s_waitcnt or a barrier / stalled at issue (dependency, pipe busy)
                      "sq_quad_cycles": {k: v[k] for k in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                                            "SQ_WAIT_INST_LDS") if k in v},
                      "insts_per_wave_tick": {k[len("SQ_INSTS_"):]: v[k] / wt for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                                                                                       "SQ_INSTS_SMEM") if k in v},
                      "wave_tick": f"{arenas_per_wave} arenas x 1 tick", "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ... (tools/prof_pmc.sh)"}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic":
        traffic_json(sys.argv[2], sys.argv[3], sys.argv[4], 0, int(sys.argv[5]), int(sys.argv[6]))
    elif len(sys.argv) > 1 and sys.argv[1] == "--pmcjson":
        pmc_json(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7] if len(sys.argv) > 7 else None, sys.argv[8:])
    else:
        main()
