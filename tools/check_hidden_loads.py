#!/usr/bin/env python3
"""Lint the gfx950 assembly of the HIP extension for the one thing the hand-placed action-row loads rely on (ActionRow / ActionWord,
split_kernel.hpp): the load is an `asm volatile` the compiler does not know to be asynchronous -- it believes the destination registers
hold the row from the asm statement on, while the data only lands before the matching `s_waitcnt vmcnt(0)` (take()). That is sound as
long as NOTHING touches those registers between the two: a copy the register allocator placed there (live-range split, spill to an AGPR
or to scratch) would read registers the load has not written yet. This walks the control-flow graph of every kernel from each such load
to the asm waits it can reach and reports any instruction on the way that names one of the destination registers.
    check_hidden_loads.py file.s   -> prints offending instructions, exit 1 if any."""
import re
import sys

LABEL = re.compile(r"^([.\w$]+):")
# The walk stops at the first label after the load: past a join the registers may legitimately hold another wave role's values (the
# kernels branch on the wave's role, and the role that never uses the row reuses its registers), which a static walk cannot tell from
# a copy of the row. What the allocator does to a value it wants out of the way -- a spill or a copy right behind its definition -- sits
# in front of that join and is caught; every kernel form with such a load is also held to the oracle by a test whose outcome depends on
# the loaded values (random control indices / weapon bits).
STOP_AT_JOINS = True
BR = re.compile(r"^\s+(s_cbranch_\w+|s_branch)\s+([.\w$]+)")
LOAD = re.compile(r"^\s+global_load_dword(?:x[234])?\s+(v\[\d+:\d+\]|v\d+),")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    text = text.split(";")[0].split("//")[0]
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def functions(lines):
    """(name, first line, last line) of every function body (.type name,@function ... .Lfunc_end)."""
    cur, start = None, None
    for i, l in enumerate(lines):
        m = LABEL.match(l)
        if m and not m.group(1).startswith(".L") and not l.startswith("\t"):
            cur, start = m.group(1), i
        if l.startswith(".Lfunc_end") and cur is not None:
            yield cur, start, i
            cur = None


def scan(path):
    lines = open(path).read().split("\n")
    bad, checked = [], 0
    for name, lo, hi in functions(lines):
        body = lines[lo:hi]
        labels = {}
        for i, l in enumerate(body):
            m = LABEL.match(l)
            if m:
                labels[m.group(1)] = i
        # inline-asm statements: index of the first line inside, its text
        asm_at = {}
        i = 0
        while i < len(body):
            if "#ASMSTART" in body[i]:
                j = i + 1
                txt = []
                while j < len(body) and "#ASMEND" not in body[j]:
                    txt.append(body[j]); j += 1
                asm_at[i] = (j, "\n".join(txt))
                i = j
            i += 1
        for i, (j, txt) in asm_at.items():
            m = LOAD.search("\n" + txt) or LOAD.match(txt)
            if not m:
                continue
            dst = frozenset(regs_of(m.group(1)))
            checked += 1
            # Forward walk over the control-flow graph from the line after the asm statement, carrying the set of destination registers
            # that still hold "the row" as far as the compiler is concerned. A READ of one of them before the take() is the hazard; a
            # WRITE means the compiler considers the row dead on this path (the paths of the other wave roles, where the row is never
            # used, reuse the registers at once) and drops the register from the set.
            seen, work, reached_wait = set(), [(j + 1, dst)], False
            while work:
                k, hold = work.pop()
                while k < len(body) and hold:
                    if (k, hold) in seen:
                        break
                    seen.add((k, hold))
                    l = body[k]
                    if STOP_AT_JOINS and LABEL.match(l) and k > j + 1:
                        reached_wait = True              # (beyond the first join the paths of the other wave roles cannot be told apart statically)
                        break
                    if k in asm_at:                      # another asm statement: a take() ends the path
                        e, t = asm_at[k]
                        if "s_waitcnt vmcnt(0)" in t:
                            reached_wait = True
                            break
                        k = e + 1
                        continue
                    if l.startswith("\t") or l.startswith(" "):
                        ins = l.strip()
                        if ins and not ins.startswith((";", ".", "//")):
                            mnem = ins.split()[0]
                            ops = ins[len(mnem):].split(";")[0]
                            parts = [o.strip() for o in ops.split(",")]
                            has_vdst = (mnem.startswith(("v_", "ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append", "global_load", "scratch_load",
                                                         "buffer_load", "flat_load", "global_atomic", "ds_add_rtn", "ds_min_rtn", "ds_max_rtn"))
                                        and not mnem.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_accvgpr_write")) and parts and parts[0].startswith("v"))
                            wr = regs_of(parts[0]) if has_vdst else set()
                            rd = regs_of(",".join(parts[1:])) if has_vdst else regs_of(ops)
                            if has_vdst and any(t in mnem for t in ("fmac", "mac_", "_dpp", "dot2c", "dot4c", "dot8c", "movrel", "mfma", "swap")):
                                rd |= wr
                            if rd & hold:
                                bad.append((name, lo + k + 1, ins, sorted(hold)))
                                hold = hold - rd
                            hold = hold - wr
                            if ins.startswith("s_endpgm"):
                                break
                            b = BR.match(l)
                            if b:
                                tgt = labels.get(b.group(2))
                                if tgt is not None:
                                    work.append((tgt, hold))
                                if b.group(1) == "s_branch":
                                    break
                    k += 1
            if not reached_wait:
                bad.append((name, lo + i + 1, "no asm `s_waitcnt vmcnt(0)` reachable from this load", sorted(dst)))
    return bad, checked


if __name__ == "__main__":
    bad, checked = scan(sys.argv[1])
    for name, ln, ins, dst in bad:
        print(f"{name}: line {ln}: `{ins}` touches v{dst[0]}..v{dst[-1]} of a hand-placed load still in flight")
    print(f"{checked} hand-placed loads checked, {len(bad)} problems")
    sys.exit(1 if bad else 0)
