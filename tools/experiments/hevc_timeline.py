"""Reads the per-decoder picture timelines HWANG_HEVC_TIMELINE wrote (hevc_video_decoder.cpp) and says where a lane's time
goes: host parse, waiting for a staging set, and on the GPU side how long a picture takes from submission to its end, how
much of that is the GPU working on it (start -> end) and how much is waiting to start."""
import glob
import sys

import numpy as np

rows = []
for k, path in enumerate(sorted(glob.glob(sys.argv[1] + "*.csv"))):
    a = np.loadtxt(path, delimiter=",", skiprows=1, ndmin=2)
    if len(a):
        rows.append(np.concatenate([a, np.full((len(a), 1), k)], axis=1))
a = np.concatenate(rows)
begin, stall_end, parsed, submitted, g0, g1, intra, dec = a.T
t_hi = g1.max()
sel = begin > t_hi - float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else np.ones(len(a), bool)
print("pictures", sel.sum(), "decoders", len(rows), "span ms", (g1[sel].max() - begin[sel].min()) / 1e3)
for name, m in (("intra pictures", sel & (intra == 1)), ("other pictures", sel & (intra == 0))):
    if not m.any():
        continue
    f = lambda v: "mean %8.1f  p50 %8.1f  p90 %8.1f  max %8.1f" % (v[m].mean(), np.percentile(v[m], 50), np.percentile(v[m], 90), v[m].max())
    print(name, int(m.sum()))
    print("  host: wait for a staging set  us", f(stall_end - begin))
    print("  host: parse                   us", f(parsed - stall_end))
    print("  host: submit                  us", f(submitted - parsed))
    print("  gpu: submit -> stream starts  us", f(np.maximum(g0 - parsed, 0)))
    print("  gpu: start -> end             us", f(g1 - g0))
    print("  gpu: submitted -> end         us", f(g1 - submitted))
# per decoder: the fraction of the span its stream is busy, its host is parsing, its host is stalled
span = g1[sel].max() - begin[sel].min()
print("per decoder averages: stream busy %.3f, host parsing %.3f, host stalled %.3f of the span" % (
    (g1 - g0)[sel].sum() / span / len(rows), (parsed - stall_end)[sel].sum() / span / len(rows), (stall_end - begin)[sel].sum() / span / len(rows)))
