#!/bin/bash
# GPU box: kernel trace of the perf probe, summarised per kernel and batch width.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf /tmp/trace_out
FRAMES=${FRAMES:-1200} REPEAT=${REPEAT:-90} LANES=${LANES:-128} PROF_LANES=${LANES:-128} timeout 300 python tools/perf_probe.py > /dev/null 2>&1  # corpus + warm caches
FRAMES=${FRAMES:-1200} REPEAT=${REPEAT:-90} LANES=${LANES:-128} PROF_LANES=${LANES:-128} timeout 400 rocprofv3 --kernel-trace -d /tmp/trace_out -o probe --output-format csv -- python3 tools/perf_probe.py 2>&1 | grep -E "lanes|launches" 
python3 tools/analyze_trace.py /tmp/trace_out ${AT:-0.6} ${NEV:-120}
