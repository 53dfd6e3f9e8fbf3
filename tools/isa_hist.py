"""static instruction histogram of one kernel in the disassembly tools/kernel_meta.sh writes (DIS=1):  python tools/isa_hist.py <name regex> [k.s]"""
import collections
import re
import sys

pat = re.compile(sys.argv[1])
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/hh_kernel_meta/k.s"
ops, total, inside = collections.Counter(), 0, False
for line in open(path):
    m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
    if m:
        inside = bool(pat.search(m.group(1)))
        continue
    if inside:
        m = re.match(r"\s+([a-z_0-9]+)", line)
        if m:
            ops[m.group(1)] += 1
            total += 1
print(total, "instructions")
for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"{k:40s}{v}")
