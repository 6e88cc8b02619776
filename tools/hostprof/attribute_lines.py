#!/usr/bin/env python3
"""Development aid: samples of libsampler.so that fall into libhwang_amd.so,
attributed to functions and source lines of a build of the library WITH debug
info (same sources, -g).  usage: attribute_lines.py samples.txt lib_with_debug.so"""
import bisect, collections, subprocess, sys
path, lib = sys.argv[1], sys.argv[2]
import os
ADDR2LINE = "/opt/rocm/lib/llvm/bin/llvm-addr2line" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-addr2line") else "addr2line"
segs = []
for l in subprocess.run(["readelf", "-lW", lib], capture_output=True, text=True).stdout.split("\n"):
    f = l.split()
    if f and f[0] == "LOAD": segs.append((int(f[1], 16), int(f[2], 16), int(f[4], 16)))
maps = []
for line in open(path + ".maps"):
    f = line.split(); lo, hi = (int(x, 16) for x in f[0].split("-"))
    maps.append((lo, hi, int(f[2], 16), f[5] if len(f) > 5 else "[anon]"))
maps.sort(); los = [m[0] for m in maps]
addrs = []; n = 0
for line in open(path):
    f = line.split(None, 2)
    if not f: continue
    n += 1
    pc = int(f[0], 16); i = bisect.bisect_right(los, pc) - 1
    if i >= 0 and pc < maps[i][1] and "libhwang" in maps[i][3]:
        off = pc - maps[i][0] + maps[i][2]
        for o, v, sz in segs:
            if o <= off < o + sz: addrs.append(off - o + v); break
out = subprocess.run([ADDR2LINE, "-f", "-C", "-e", lib] + [hex(a) for a in addrs], capture_output=True, text=True).stdout.split("\n")
fn = collections.Counter(); ln = collections.Counter()
for name, loc in zip(out[0::2], out[1::2]):
    name = name.replace("hwang::h264::Parser::Impl::", "").replace("hwang::h264::(anonymous namespace)::", "").split("(")[0]
    fn[name] += 1
    ln[loc.split("/")[-1].split(" ")[0]] += 1
print(n, "samples,", len(addrs), "in the library")
for k, v in fn.most_common(24): print("%5d %5.1f%%  %s" % (v, 100.0 * v / n, k))
print("-- hottest lines")
for k, v in ln.most_common(40): print("%5d %5.1f%%  %s" % (v, 100.0 * v / n, k))
