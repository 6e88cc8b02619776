# A/B of host-only builds of the HEVC parser (tools/experiments/src/hevc_parse_samp.cpp, product flags) on a GPU box's CPU:
# the build container's timings move by +-10 % from run to run.  usage: builds in _ab/hevc_parse_<name>, data in _ab/hv
for r in 1 2 3; do for b in "$@"; do echo -n "$b "; taskset -c 8 _ab/hevc_parse_$b _ab/hv/hvcc _ab/hv/blob _ab/hv/sizes 6 | awk '{print $3}' | sort -n | head -1; done; done
echo "intra picture only:"
for r in 1 2; do for b in "$@"; do echo -n "$b "; taskset -c 8 _ab/hevc_parse_$b _ab/hv/hvcc _ab/hv/blob _ab/hv/sizes1 30 | awk '{print $3}' | sort -n | head -1; done; done
