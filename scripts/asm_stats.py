#!/usr/bin/env python3
"""Instruction mix of one kernel in the `make asm` listing: asm_stats.py <mangled-name-substring> [listing]."""
import collections
import sys

path = sys.argv[2] if len(sys.argv) > 2 else "nf-isam_amd/csrc/_asm/nsf_kernels.s"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if sys.argv[1] in l and l.rstrip().endswith(tuple([":"])) is False and l.startswith("_Z") and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
c = collections.Counter()
for l in lines[start + 1:end]:
    t = l.strip()
    if t and not t.startswith((";", ".")) and not t.endswith(":"):
        c[t.split()[0]] += 1
print(lines[start].split(":")[0], "instructions:", sum(c.values()))
for k, v in c.most_common(40):
    print(f"  {k:32s} {v}")
