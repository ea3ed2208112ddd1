#!/usr/bin/env python3
"""Source lint for the shuffle discipline of the step kernels: a `__shfl*` on the right-hand side of `&&`, `||` or
`?:` is only executed by the lanes that get there; the other lane of the pair is masked off and the reader gets its OWN value back.
Every shuffle must be fetched into a variable in env-uniform control flow first. Usage: check_shuffle_discipline.py files..."""
import re
import sys

PAT = re.compile(r"(\|\||&&|\?)[^;]*__shfl")


def main():
    bad = 0
    for path in sys.argv[1:]:
        for ln, line in enumerate(open(path), 1):
            code = line.split("//")[0]
            if PAT.search(code):
                bad += 1
                print(f"{path}:{ln}: shuffle on the right of a short-circuit / conditional operator: {code.strip()[:140]}")
    print(f"{bad} violation(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
