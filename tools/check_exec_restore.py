#!/usr/bin/env python3
"""Lint the gfx950 assembly of the HIP extension for a code-generation hazard seen with ROCm 7.2's LLVM on very large kernels:
a join block whose `s_or_b64 exec, exec, s[a:b]` (re-activating the lanes that skipped a divergent region) is preceded, in the
same basic block, by vector instructions. Those instructions run under the narrowed mask, so register copies the allocator
placed there (spill-to-AGPR, live-range splits) are lost for every lane that skipped the region. Usage:
    check_exec_restore.py file.s   -> prints offending blocks, exit 1 if any."""
import re
import sys

VEC = re.compile(r"^\s+(v_|global_|scratch_|ds_|buffer_|flat_)")
LABEL = re.compile(r"^(\.LBB\d+_\d+|[A-Za-z_][\w$.]*):")
SAVE = re.compile(r"^\s+s_and_saveexec_b64 (s\[\d+:\d+\]),")
SKIP = re.compile(r"^\s+s_cbranch_execz (\.LBB\d+_\d+)")
RESTORE = re.compile(r"^\s+s_or_b64 exec, exec, (s\[\d+:\d+\])")
ENDBLOCK = re.compile(r"^\s+(s_cbranch|s_branch|s_endpgm|s_setpc)")


def scan(path):
    """A hazard is: `s_and_saveexec_b64 S, cond ; s_cbranch_execz L` ... `L:` <vector instructions> `s_or_b64 exec, exec, S`.
    Lanes that skipped the region arrive at L with exec still narrowed (possibly empty), so the vector instructions do not run
    for them. (Without the skip branch the same shape is legitimate straight-line predication and is not reported.)"""
    lines = open(path).read().split("\n")
    bad = []
    func = None
    skip_mask = {}          # label -> saved-exec register pair of the region it closes
    last_save = None
    for ln, line in enumerate(lines, 1):
        m = LABEL.match(line)
        if m and not m.group(1).startswith(".L"):
            func = m.group(1)
        m = SAVE.match(line)
        if m:
            last_save = (m.group(1), ln)
            continue
        m = SKIP.match(line)
        if m and last_save and ln - last_save[1] <= 3:
            skip_mask[(func, m.group(1))] = last_save[0]
    func = None
    i = 0
    while i < len(lines):
        line = lines[i]
        m = LABEL.match(line)
        if m and not m.group(1).startswith(".L"):
            func = m.group(1)
        if m and (func, m.group(1)) in skip_mask:
            want = skip_mask[(func, m.group(1))]
            vecs = []
            j = i + 1
            while j < len(lines):
                t = lines[j]
                if LABEL.match(t) or ENDBLOCK.match(t):
                    break
                r = RESTORE.match(t)
                if r and r.group(1) == want:
                    if vecs:
                        bad.append((func, (m.group(1), i + 1), j + 1, vecs))
                    break
                if re.match(r"^\s+s_\w*saveexec|^\s+s_\w+ exec,", t):
                    break            # an else-flip or another region begins: what follows is predicated on purpose
                if VEC.match(t) and not re.match(r"^\s+v_(writelane|readlane|readfirstlane)_b32", t):
                    vecs.append((j + 1, t.strip()))      # (lane-indexed SGPR spill moves ignore EXEC)
                j += 1
        i += 1
    return bad


def main():
    total = 0
    for path in sys.argv[1:]:
        for func, label, ln, vecs in scan(path):
            total += 1
            print(f"{path}:{ln}: {func}: block {label[0]} (line {label[1]}) runs {len(vecs)} vector instruction(s) before restoring exec:")
            for l, t in vecs[:6]:
                print(f"    {l}: {t}")
    print(f"{total} hazard(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
