"""prints the main numbers of a bench.py line (file argument or stdin)"""
import json, sys
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
d = json.loads([l for l in src if l.startswith("{")][-1])
print("headline", "%.4g" % d["value"], d["unit"], round(d["ms_per_step"], 4), "ms/step; roofline frac", round(d["roofline"]["frac"], 4), "; cpu_baseline", d.get("cpu_baseline", {}).get("value"))
for k, v in d.get("extra", {}).items():
    print(" ", k, v.get("value") and "%.4g" % v["value"], v.get("ms_per_step") and round(v["ms_per_step"], 4), v.get("error") or v.get("skipped") or "", v.get("child_seconds"))
