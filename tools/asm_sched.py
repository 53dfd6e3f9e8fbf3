#!/usr/bin/env python3
"""Post-register-allocation scheduling pass over hipcc's gfx950 assembly (build step of libhh_world.so, __graft_entry__.build).

Why.  tools/ubench/pred.hip (one wave on a SIMD, raw instruction streams): a vector instruction that writes a scalar register — v_cmp (lane mask to
VCC / an SGPR pair), v_readlane, v_div_scale, v_mad_u64_u32's carry — keeps the SCALAR unit from issuing for ~20 cycles: `v_cmp; s_and_b64` takes 24.3 cycles
instead of 8, whatever the s_ instruction is (the dependent s_and_b64 of a predicate chain, or an unrelated s_mov_b32 of the next f64 literal), while four
vector instructions placed between the two hide the wait completely (v_cmp, 4 x v_mov, s_and: 24.3 cycles for six instructions).  hipcc's scheduler does not
model this: the persistent world kernels' tick loops hold one such pair every seventeenth instruction (278 sites in the 4 707 instructions of
hh_k_world_quad<1, 1, true, 8, true, true>'s loop, 217 of them back to back), and at one wave per SIMD nothing else fills the gap.

What.  Inside basic blocks only, two moves that change no value:
  * a literal move `s_mov_b32 / s_mov_b64 sX, <constant>` that would be the first scalar instruction behind such a vector instruction is hoisted above it
    (past instructions that do not mention sX);
  * the vector instruction itself (plain v_cmp forms) is hoisted up past independent non-scalar instructions until four of them separate it from the scalar
    instruction that follows.
Both moves only cross instructions that share no register with the moved one (any overlap of any operand counts, reads included), never cross labels, branches,
waits, barriers, s_nop pads, lane-access / DPP-sensitive / matrix instructions, PC-relative address sequences or inline-asm blocks, and a moved vector
instruction never ends up directly behind an instruction that writes one of its operands unless it already stood there (trans-use and similar one-slot hazards keep
the distance the compiler gave them).  Results are bit-identical by construction (same instructions, same operands, dependencies preserved); tests/ and the soaks
compare the built library with the CPU oracle as before.

Usage: asm_sched.py in.s out.s [--stats] [--only REGEX]
"""
import re
import sys

REG_RE = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc_lo|vcc_hi|vcc|exec_lo|exec_hi|exec|m0|scc)\b")
LABEL_RE = re.compile(r"^[A-Za-z_.$][\w.$@]*:")
MOVES = "abc"       # a: literal s_mov hoist, b: compare hoist, c: filler pull-up (--moves: bisecting a miscompare)
WINDOW = 4          # instruction slots between a scalar-register-writing vector instruction and the next scalar instruction that hide the wait


def regs_of(text):
    out = set()
    for m in REG_RE.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        elif m.group(3):
            for k in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), k))
        else:
            g = m.group(6)
            if g.startswith("vcc"):
                out.update({("vcc", 0), ("vcc", 1)} if g == "vcc" else {("vcc", 0 if g.endswith("lo") else 1)})
            elif g.startswith("exec"):
                out.update({("exec", 0), ("exec", 1)} if g == "exec" else {("exec", 0 if g.endswith("lo") else 1)})
            else:
                out.add((g, 0))
    return out


SMEM_PREFIX = ("s_load", "s_store", "s_buffer_load", "s_buffer_store", "s_dcache", "s_memtime", "s_memrealtime", "s_atc", "s_scratch")
NOT_SALU = ("s_nop", "s_waitcnt", "s_barrier", "s_sleep", "s_endpgm", "s_branch", "s_cbranch", "s_setprio", "s_sethalt", "s_trap", "s_code_end", "s_icache_inv", "s_wakeup", "s_ttracedata")
HARD = ("s_waitcnt", "s_barrier", "s_nop", "s_sleep", "s_endpgm", "s_branch", "s_cbranch", "s_setprio", "s_sethalt", "s_trap", "s_setreg", "s_getreg", "s_getpc", "s_setpc", "s_swappc", "s_call",
        "s_rfe", "s_memtime", "s_memrealtime", "s_icache_inv", "s_dcache", "s_sendmsg", "s_code_end", "s_wakeup", "s_ttracedata", "s_movrel", "s_set_gpr_idx",
        "v_cmpx", "v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_mfma", "v_smfmac", "v_accvgpr", "v_swap", "v_movrel", "v_interp", "v_nop", "v_pipeflush", "s_cbranch_g_fork")
SGPR_WRITING_VALU = ("v_cmp_", "v_readlane", "v_readfirstlane", "v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32", "v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co")


class Ins:
    __slots__ = ("text", "op", "kind", "regs", "dst", "hard", "salu", "wsgpr", "lit_mov", "movable_cmp", "wexec")

    def __init__(self, text):
        self.text = text
        body = text.split(";", 1)[0].strip()
        parts = body.split(None, 1)
        self.op = parts[0] if parts else ""
        args = parts[1] if len(parts) > 1 else ""
        op = self.op
        self.kind = "ins"
        self.regs = regs_of(args)
        self.dst = set()
        first = args.split(",", 1)[0] if args else ""
        self.salu = op.startswith("s_") and not op.startswith(NOT_SALU)
        self.hard = op.startswith(HARD) or "@" in body or "dpp" in body or "sdwa" in body or "quad_perm" in body or "row_" in body or "wave_" in body or "bank_mask" in body
        self.wsgpr = op.startswith(SGPR_WRITING_VALU)
        # may change EXEC (s_..._saveexec writes it without naming it): no vector or memory instruction is ever moved across one of these
        self.wexec = "saveexec" in op or op.startswith("v_cmpx") or (op.startswith("s_") and re.search(r"\bexec(_lo|_hi)?\b", first) is not None) or op.startswith(("s_cmov", "s_wqm", "s_quadmask"))
        if op.startswith("v_"):
            # (every vector instruction reads EXEC; its writers are scalar instructions or v_cmpx, which nothing vector is ever moved across)
            if op.startswith(("v_div_fmas", "v_cndmask_b32_e32", "v_addc_co_u32_e32", "v_subb_co_u32_e32", "v_subbrev_co_u32_e32")):
                self.regs |= {("vcc", 0), ("vcc", 1)}
        if self.salu:
            if not op.startswith(("s_mov_b32", "s_mov_b64", "s_movk_i32")):
                self.regs.add(("scc", 0))   # written or read: any overlap blocks a move
        if op.startswith(SMEM_PREFIX):
            self.salu = True      # issued by the scalar unit: waits like an s_ ALU instruction (for detection); never moved, never crossed by a literal move's register
        # a literal move: s_mov_b32 / b64 sX, <no register operand>
        self.lit_mov = False
        if op in ("s_mov_b32", "s_mov_b64") and "," in args and not self.hard:
            d, src = args.split(",", 1)
            dreg = regs_of(d)
            if dreg and all(r[0] == "s" for r in dreg) and re.match(r"^-?(0x[0-9a-fA-F]+|\d+(\.\d+)?([eE][-+]?\d+)?)$", src.strip()):
                self.lit_mov = True
                self.dst = dreg
        # a plain vector compare whose mask goes to VCC or an SGPR pair (its first operand)
        self.movable_cmp = False
        if op.startswith("v_cmp_") and not self.hard:
            self.dst = regs_of(first)
            self.movable_cmp = bool(self.dst)


def conflict(a, b):
    """any register both mention (reads included: the coarse test keeps every dependency, true or not)"""
    return not a.regs.isdisjoint(b.regs)


def _window_writers(block, i):
    """indices of the scalar-register-writing vector instructions inside the WINDOW slots in front of the scalar instruction at i, none of them with
    another scalar instruction between (the FIRST scalar instruction behind such a vector instruction takes the wait), farthest first"""
    found = []
    k = i - 1
    while k >= 0 and i - k - 1 < WINDOW:
        b = block[k]
        if b.salu or b.op.startswith(("s_cbranch", "s_branch")):
            break
        if b.wsgpr:
            found.append(k)
        k -= 1
    return sorted(found)


def _hoist_cmp(block, idx, need):
    """move the compare at idx up by at most `need` slots; returns the slots gained"""
    w = block[idx]
    p = idx
    moved = 0
    while p > 0 and moved < need:
        above = block[p - 1]
        if above.hard or above.salu or above.wexec or above.wsgpr or above.op.startswith(SMEM_PREFIX) or conflict(above, w):
            break
        if p - 2 >= 0:   # never end up directly behind an instruction that mentions one of w's operands unless w already stood there (one-slot hazards keep their distance)
            pred = block[p - 2]
            if not pred.regs.isdisjoint(w.regs) and not pred.salu:
                break
        p -= 1
        moved += 1
    if moved:
        block.insert(p, block.pop(idx))
    return moved


def _hoist_mov(block, j, first_writer):
    """move the literal s_mov at j up above the vector instructions it would wait for; True if it found a place whose WINDOW holds none of them"""
    s = block[j]
    p = j
    while p > 0:
        above = block[p - 1]
        if above.hard or above.op.startswith(("s_cbranch", "s_branch")) or above.op.startswith(SMEM_PREFIX) or not above.regs.isdisjoint(s.dst):
            return False
        p -= 1
        if p <= first_writer and not _window_writers(block[:p] + [s], p):
            block.insert(p, block.pop(j))
            return True
        if j - p > 24:
            return False
    return False


TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_", "v_frexp_", "v_ldexp_")
MEM = ("ds_", "global_", "buffer_", "scratch_", "flat_", "s_load", "s_store", "s_buffer", "image_", "tbuffer_")


def _sensitive(b):
    """an instruction that is one end of a hardware wait-state rule on this target (the compiler satisfied those rules for ITS instruction order: nothing is
    taken out of, or put into, the five slots on either side of one)"""
    return b.hard or b.op.startswith(TRANS) or b.op.startswith(MEM) or b.op.startswith("v_div_fmas") or b.op.startswith("v_mfma") or "op_sel" in b.text


def _plain_valu(b):
    return b.op.startswith("v_") and not b.hard and not b.wsgpr and not b.salu and not b.op.startswith(TRANS) and not b.op.startswith("v_div_fmas") and "op_sel" not in b.text


def _pull_fillers(block, j, need, lookahead=32):
    """independent plain vector instructions from behind the scalar instruction at j, moved in front of it (each one slot of the wait hidden).  A candidate
    crosses only instructions it shares no register with, stands at least five slots from any wait-state-sensitive instruction where it is taken from, and lands
    behind two instructions it shares no register with.  Returns the number moved."""
    got = 0
    q = j + 1
    while got < need and q < len(block) and q - j <= lookahead:
        x = block[q]
        if x.op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")) or x.hard and x.op.startswith(("s_waitcnt", "s_barrier")):
            break
        if _plain_valu(x):
            lo, hi = max(0, q - 5), min(len(block), q + 6)
            if not any(_sensitive(block[k]) for k in range(lo, hi) if k != q) and not any(_sensitive(block[k]) for k in range(max(0, j - 5), min(len(block), j + 1))):
                crossed = block[j:q]
                if any(c.wexec for c in crossed):
                    break          # nothing behind an EXEC write may come in front of it
                if all(x.regs.isdisjoint(c.regs) for c in crossed) and all(x.regs.isdisjoint(block[k].regs) for k in range(max(0, j - 2), j)):
                    block.insert(j, block.pop(q))
                    j += 1
                    got += 1
                    q += 1
                    continue
        q += 1
    return got


def schedule_block(block, stats):
    """block: the instructions between two labels / branches / comment lines.  In place, one forward pass over its scalar instructions."""
    if len(block) < 3:
        return
    i = 0
    while i < len(block):
        s = block[i]
        if not s.salu:
            i += 1
            continue
        found = _window_writers(block, i)
        if not found:
            i += 1
            continue
        stats["sites"] += 1
        stats["stall_slots_before"] += WINDOW - (i - found[-1] - 1)
        if "a" in MOVES and s.lit_mov and _hoist_mov(block, i, found[0]):
            stats["mov_hoisted"] += 1
            continue       # another instruction stands at i now: examine it
        for n_done, idx in enumerate(found):           # farthest first, so that a group of compares moves up together
            w = block[idx]
            if not w.movable_cmp or "b" not in MOVES:
                continue
            gap = i - idx - 1
            got = _hoist_cmp(block, idx, WINDOW - gap)
            if got:
                stats["cmp_hoisted"] += 1
        left = _window_writers(block, i)
        if left and "c" in MOVES:   # still short: fill the gap with independent vector instructions from behind the scalar instruction
            need = WINDOW - (i - left[-1] - 1)
            got = _pull_fillers(block, i, need)
            if got:
                stats["fillers"] += got
                i += got           # the scalar instruction moved down by as many slots
                left = _window_writers(block, i)
        stats["stall_slots_after"] += (WINDOW - (i - left[-1] - 1)) if left else 0
        i += 1


def process(lines, only=None, stats=None):
    out = []
    block = []
    fn = None
    active = True
    in_asm = False

    def flush():
        nonlocal block
        if block:
            if active and not in_asm:
                schedule_block(block, stats)
            out.extend(x.text for x in block)
            block = []

    for ln in lines:
        s = ln.strip()
        if s.startswith(";;#ASMSTART") or s.startswith(";#ASMSTART"):
            flush()
            in_asm = True
            out.append(ln)
            continue
        if s.startswith(";;#ASMEND") or s.startswith(";#ASMEND"):
            in_asm = False
            out.append(ln)
            continue
        if not s or s.startswith(";") and not block:
            out.append(ln)
            continue
        if s.startswith(";"):      # a comment line inside a block: keep the block's instructions together, drop nothing
            flush()
            out.append(ln)
            continue
        if s.startswith(".") and not LABEL_RE.match(s) or LABEL_RE.match(s):
            flush()
            m = re.match(r"^(_Z[\w$]+|[A-Za-z_][\w$]*):", s)
            if m and not s.startswith(".L"):
                fn = m.group(1)
                active = only is None or re.search(only, fn) is not None
            out.append(ln)
            continue
        if in_asm:
            out.append(ln)
            continue
        ins = Ins(ln)
        block.append(ins)
        if ins.op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            flush()
    flush()
    return out


def main():
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    only = None
    if "--only" in sys.argv:
        only = sys.argv[sys.argv.index("--only") + 1]
        a = [x for x in a if x != only]
    global MOVES
    if "--moves" in sys.argv:
        MOVES = sys.argv[sys.argv.index("--moves") + 1]
        a = [x for x in a if x != MOVES]
    src, dst = a[0], a[1]
    stats = {"sites": 0, "mov_hoisted": 0, "cmp_hoisted": 0, "fillers": 0, "stall_slots_before": 0, "stall_slots_after": 0}
    with open(src) as f:
        lines = f.read().split("\n")
    out = process(lines, only, stats)
    with open(dst, "w") as f:
        f.write("\n".join(out))
    if "--stats" in sys.argv:
        print("asm_sched:", stats, file=sys.stderr)


if __name__ == "__main__":
    main()
