#!/usr/bin/env python3
"""Development aid: llvm-symbolizer output (--output-style=LLVM --inlining) of the instruction samples -> innermost function,
the chain of callers inside SliceData, and hottest source lines."""
import collections, sys
blocks = open(sys.argv[1]).read().split('\n\n')
fn = collections.Counter(); ln = collections.Counter(); incl = collections.Counter()
tot = 0
for b in blocks:
    l = [x for x in b.split('\n') if x]
    if len(l) < 2: continue
    tot += 1
    frames = [(l[i], l[i + 1].split('/')[-1].rsplit(':', 1)[0]) for i in range(0, len(l) - 1, 2)]
    fn[frames[0][0]] += 1
    ln[frames[0][1]] += 1
    for f in set(x[0] for x in frames): incl[f] += 1
print(tot, 'samples')
print('-- innermost function (exclusive)')
for k, v in fn.most_common(40): print('%6d %5.1f%%  %s' % (v, 100 * v / tot, k))
print('-- inclusive')
for k, v in incl.most_common(40): print('%6d %5.1f%%  %s' % (v, 100 * v / tot, k))
print('-- lines')
for k, v in ln.most_common(60): print('%6d %5.1f%%  %s' % (v, 100 * v / tot, k))
