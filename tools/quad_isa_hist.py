"""Per-opcode instruction histogram of hh_k_world_quad's two loops — the simulation wave's tick loop and the output wave's row loop —
from the disassembly tools/kernel_meta.sh writes (DIS=1).  Static counts of the loop bodies (every path once: the rare regions — launch
bookkeeping, kill resolution, Karney fallback, reset — are listed with the rest; the PMC counters of profiles/ give the dynamic totals),
grouped into the classes the round-3 review asked for: FP64 arithmetic, moves, selects, compares, DPP / lane exchange, conversions,
integer / logic, transcendental, LDS, global memory, scalar, control.
    DIS=1 bash tools/kernel_meta.sh && python tools/quad_isa_hist.py 'hh_k_world_quadILi1ELi1ELb1ELi8ELb1E' """
import collections
import re
import sys

pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else r"hh_k_world_quadILi1ELi1ELb1ELi8ELb1E")
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/hh_kernel_meta/k.s"
lines, inside = [], False
for line in open(path):
    m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
    if m:
        inside = bool(pat.search(m.group(1)))
        continue
    if inside and re.match(r"\s+[a-z_0-9]+", line):
        lines.append(line.split("//")[0].strip() + " //" + line.split("//")[1].split(":")[0] if "//" in line else line.strip())
addr = lambda l: int(l.split("//")[1], 16)
body = lines
loops = []
for i, l in enumerate(body):
    m = re.match(r"s_c?branch\S*\s+(\d+)", l)
    if m and int(m.group(1)) > 32767:
        tgt = addr(l) + 4 + (int(m.group(1)) - 65536) * 4
        j = [k for k, x in enumerate(body) if addr(x) == tgt]
        if j:
            loops.append((j[0], i))
# outermost loops only
outer = [lp for lp in loops if not any(o[0] <= lp[0] and lp[1] <= o[1] and o != lp for o in loops)]


def klass(op, full):
    if "dpp" in full or op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane", "ds_swizzle", "ds_bpermute", "v_mov_b32_dpp")):
        return "DPP / lane exchange"
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_ldexp_f64", "v_fract_f64", "v_floor_f64", "v_trunc_f64", "v_rndne_f64", "v_ceil_f64", "v_div_")):
        return "FP64 arithmetic"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rcp_f32", "v_exp", "v_log", "v_sqrt_f32", "v_rsq_f32", "v_sin", "v_cos")):
        return "transcendental"
    if op.startswith("v_cndmask"):
        return "selects (v_cndmask)"
    if op.startswith(("v_cmp", "v_cmpx")):
        return "compares"
    if op.startswith(("v_mov", "v_accvgpr", "v_pk_mov", "v_swap")):
        return "moves"
    if op.startswith("v_cvt"):
        return "conversions"
    if op.startswith(("ds_",)):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "global / scratch memory"
    if op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_waitcnt", "s_nop", "s_endpgm", "s_setpc", "s_swappc", "s_getpc", "s_sleep", "s_setprio")):
        return "control / waits"
    if op.startswith("s_"):
        return "scalar ALU / moves"
    if op.startswith("v_"):
        return "other VALU (f32, int, logic)"
    return "other"


print(f"{len(body)} static instructions in the kernel; outermost loops: {[(b - a + 1) for a, b in outer]}")
for (a, b) in sorted(outer, key=lambda lp: lp[0] - lp[1])[:2]:
    seg = body[a:b + 1]
    ops = collections.Counter(x.split()[0] for x in seg)
    kl = collections.Counter(klass(x.split()[0], x) for x in seg)
    n = len(seg)
    which = "simulation wave: tick loop" if n > 1500 else "output wave: row loop"
    print(f"==== {which}: {n} static instructions in the loop body")
    for k, v in kl.most_common():
        print(f"   {k:32s} {v:6d}  {100.0 * v / n:5.1f} %")
    print("   -- opcodes")
    for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 45):
        print(f"   {k:40s} {v}")
