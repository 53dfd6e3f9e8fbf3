#!/usr/bin/env python
"""Summarise rocprofv3 sqlite outputs (rocpd) into text: per-kernel duration stats and, when
present, per-kernel PMC counter averages.  Usage: rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def cols(db, t):
    return [r[1] for r in db.execute(f"pragma table_info({t})")]


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        print(f"== {path}")
        kc = cols(db, "kernels")
        name = "name" if "name" in kc else kc[0]
        q = (f"select {name}, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
             f"group by {name} order by sum(end-start) desc")
        print(f"{'kernel':60s} {'calls':>7s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'total_ms':>12s}")
        for n, c, a, mn, mx, s in db.execute(q):
            print(f"{str(n)[:60]:60s} {c:7d} {a/1e3:12.2f} {mn/1e3:12.2f} {mx/1e3:12.2f} {s/1e6:12.3f}")
        try:
            pc = cols(db, "counters_collection")
            if pc:
                kn = "kernel_name" if "kernel_name" in pc else ("name" if "name" in pc else None)
                cn = "counter_name" if "counter_name" in pc else None
                vn = "value" if "value" in pc else ("counter_value" if "counter_value" in pc else None)
                if kn and cn and vn:
                    rows = list(db.execute(f"select {kn}, {cn}, count(*), avg({vn}), sum({vn}) from counters_collection group by {kn}, {cn}"))
                    if rows:
                        print(f"{'kernel':44s} {'counter':24s} {'dispatches':>10s} {'avg/dispatch':>18s}")
                        for k, cname, n, a, s in rows:
                            print(f"{str(k)[:44]:44s} {cname:24s} {n:10d} {a:18.1f}")
                else:
                    print("counters_collection columns:", pc)
        except sqlite3.Error as e:
            print("pmc query failed:", e)


if __name__ == "__main__":
    main()
