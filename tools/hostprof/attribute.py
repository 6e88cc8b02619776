#!/usr/bin/env python3
"""Development aid: attributes the samples written by libsampler.so
("pc tid thread" lines + .maps) to thread names, mapped files and - by nearest
dynamic symbol - functions.  usage: attribute.py samples.txt [more samples...]"""
import bisect, collections, subprocess, sys
rows = []
for path in sys.argv[1:]:
    maps = []
    for line in open(path + ".maps"):
        f = line.split()
        lo, hi = (int(x, 16) for x in f[0].split("-"))
        maps.append((lo, hi, int(f[2], 16), f[5] if len(f) > 5 else "[anon]"))
    maps.sort()
    los = [m[0] for m in maps]
    for line in open(path):
        f = line.split(None, 2)
        if not f: continue
        pc = int(f[0], 16); th = f[2].strip() if len(f) > 2 else "?"
        i = bisect.bisect_right(los, pc) - 1
        if i >= 0 and pc < maps[i][1]: rows.append((th, maps[i][3], pc - maps[i][0] + maps[i][2]))
        else: rows.append((th, "?", pc))
n = len(rows)
print(n, "samples")
print("-- by thread name")
for k, v in collections.Counter(r[0] for r in rows).most_common(12): print("%6d %5.1f%%  %s" % (v, 100.0 * v / n, k))
print("-- by mapped file")
for k, v in collections.Counter(r[1].split("/")[-1] for r in rows).most_common(10): print("%6d %5.1f%%  %s" % (v, 100.0 * v / n, k))
symtab = {}
def sym(lib, off):
    if lib not in symtab:
        s = []
        for l in subprocess.run(["nm", "-D", "-C", "--defined-only", lib], capture_output=True, text=True).stdout.split("\n"):
            f = l.split(None, 2)
            if len(f) == 3:
                try: s.append((int(f[0], 16), f[2]))
                except ValueError: pass
        s.sort(); symtab[lib] = (s, [x[0] for x in s])
    s, keys = symtab[lib]
    i = bisect.bisect_right(keys, off) - 1
    return s[i][1][:90] if i >= 0 else "?"
print("-- outside libhwang_amd.so: thread / file / nearest dynamic symbol")
c = collections.Counter()
for th, lib, off in rows:
    if "libhwang_amd" in lib or lib == "?": continue
    c[(th, lib.split("/")[-1], sym(lib, off))] += 1
for k, v in c.most_common(25): print("%6d %5.1f%%  %s" % (v, 100.0 * v / n, " | ".join(k)))
