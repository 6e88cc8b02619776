#!/usr/bin/env python3
"""Development aid: objdump of one function of probe_prof with the sampler's hit counts per instruction.
usage: annotate.py samples.txt probe_prof <function substring> [min_count]"""
import collections, re, subprocess, sys
samples, binary, func = sys.argv[1], sys.argv[2], sys.argv[3]
hits = collections.Counter(int(l.split()[0], 16) for l in open(samples) if l.strip())
total = sum(hits.values())
dis = subprocess.run(["objdump", "-d", "-C", "--no-show-raw-insn", binary], capture_output=True, text=True).stdout
on = False
acc = 0
for line in dis.split("\n"):
    m = re.match(r"^([0-9a-f]+) <(.*)>:$", line)
    if m:
        on = func in m.group(2)
        if on: print(line)
        continue
    if not on: continue
    m = re.match(r"^\s*([0-9a-f]+):\s+(.*)$", line)
    if not m: continue
    a = int(m.group(1), 16)
    h = hits.get(a, 0)
    acc += h
    print("%6s  %x: %s" % (h if h else "", a, m.group(2)))
print("total in function:", acc, "of", total)
