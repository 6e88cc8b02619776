# SIGPROF samples of the HEVC parser alone on a GPU box's CPU (symbolised in the build container: same binary)
cd _ab && taskset -c 8 ./hevc_parse_prof hv/hvcc hv/blob hv/sizes1 120 | tail -1; cp /tmp/hevc_samples.txt ../gpurun_out/r06_hevc_samples_intra.txt
taskset -c 8 ./hevc_parse_prof hv/hvcc hv/blob hv/sizes 40 | tail -1; cp /tmp/hevc_samples.txt ../gpurun_out/r06_hevc_samples_gop.txt
wc -l ../gpurun_out/r06_hevc_samples_*.txt
