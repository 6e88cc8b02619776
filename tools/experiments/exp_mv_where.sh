#!/bin/bash
# Round 4: where does k_mv's time go?  Variants of the kernel (wrong vectors, timing only) with one lane (a picture's
# latency through the kernel).  Build on the CPU box first: see the loop below.  Answer: DESIGN section 3 "k_mv".
cd "$GRAFT_REPO_ROOT" || exit 1
PL="1" bash tools/experiments/ab_libs.sh _ab/lib_mv_base.so _ab/lib_mv_nowait.so _ab/lib_mv_nostore.so _ab/lib_mv_nowait_nostore.so _ab/lib_mv_nowait_nostore_noexport.so _ab/lib_mv_nobody.so
