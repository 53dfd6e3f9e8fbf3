#!/usr/bin/env python3
"""Which source lines do a kernel's branches come from?  (tuning aid for the HH_RARE / HH_USUAL block placement hints)
  hipcc ... -gline-tables-only -S --cuda-device-only hh_world.hip -o hh_world_g.s ; python tools/branch_map.py hh_world_g.s <kernel name regex> [--loop]
Prints every conditional branch of the kernel with the .loc (file:line, inlined-at chain not resolved) in force where it stands, whether its target lies
ahead (a skip over an in-line body: taken when the body is not wanted) or behind (a loop back edge), and the distance in instructions."""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    files = {}
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m and re.search(pat, m.group(1)):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    for ln in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    labels, ins_no, n = {}, {}, 0
    for i in range(start, end + 1):
        s = lines[i].strip()
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = n
        elif s and not s.startswith((";", ".")) and not s.endswith(":"):
            ins_no[i] = n
            n += 1
    loc = "?"
    out = []
    for i in range(start, end + 1):
        s = lines[i].strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            loc = f"{files.get(int(m.group(1)), m.group(1))}:{m.group(2)}"
        m = re.match(r"(s_cbranch_\w+)\s+(\.LBB\w+)", s)
        if m and i in ins_no:
            d = labels.get(m.group(2), -1) - ins_no[i]
            out.append((ins_no[i], m.group(1), d, loc))
    print(f"{n} instructions, {len(out)} conditional branches")
    for at, op, d, loc in out:
        print(f"{at:6d} {op:18s} {'ahead' if d > 0 else 'BACK '} {d:6d}  {loc}")


if __name__ == "__main__":
    main()
